"""Diagnostic: kernel time and C_l parity of an alternative build of the library (python tools/lib_try.py <lib.so> [config ...])."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from classpp_public_amd import capi
capi.LIB_PATH = os.path.abspath(sys.argv[1])
from classpp_public_amd.backend import Backend
from classpp_public_amd.inputs import Inputs
for cfg in sys.argv[2:] or ["lcdm"]:
    inp = Inputs(cfg)
    be = Backend(inp)
    for rep in range(3):
        _, stats, status = be.perturb_solve(want_sources=False)
    ms = be.kernel_ms(0)[0]
    cl = be.cl(be.transfer(None)).cpu().numpy()
    ref = inp.d["sp.cl_table"]; sp = inp.spectra
    err = max(np.max(np.abs(cl[:, i] / ref[:, i] - 1)) for i in (sp.index_ct_tt, sp.index_ct_ee))
    print("%s: perturb %.2f ms, steps %d, status ok %s, C_l TT/EE max err %.1e" % (cfg, ms, sum(s.steps for s in stats), not status.any(), err), flush=True)
    be.close()
