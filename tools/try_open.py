"""Diagnostic: a cosmology the reference itself refuses (Omega_k = +0.1: 'Bessels need to be interpolated outside the range in which they
have been computed') - what this backend does with it."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from classpp_public_amd import classy
for ok in (0.1, 0.2):
    c = classy.Class({"h": 0.72, "omega_b": 0.0223, "omega_cdm": 0.115, "Omega_k": ok, "YHe": 0.245, "z_reio": 9., "output": "tCl,pCl,mPk",
                      "l_max_scalars": 700, "P_k_max_h/Mpc": 2.})
    try:
        cl = c.compute().raw_cl()
        print("Omega_k=%g: finite=%s tt[2:6]=%s" % (ok, np.all(np.isfinite(cl["tt"])), cl["tt"][2:6]))
    except classy.CosmoError as e:
        print("Omega_k=%g: %s: %s" % (ok, type(e).__name__, str(e)[:200]))
    c.struct_cleanup()
