"""Diagnostic: delta_m(k, tau0) of a configuration against the reference's, per k."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from classpp_public_amd.backend import Backend
from classpp_public_amd.inputs import Inputs

inp = Inputs(sys.argv[1] if len(sys.argv) > 1 else "long_full")
be = Backend(inp)
src, stats, status = be.perturb_solve()
got = src.cpu().numpy()
dm, ref = got[inp.config.index_tp_delta_m, -1, :], inp.d["pt.delta_m_today"]
err = np.abs(dm / ref - 1)
print("max", err.max(), "at k index", err.argmax(), "k", inp.k[err.argmax()])
im = int(err.argmax())
for i in list(range(max(im - 4, 0), min(im + 5, inp.nk))):
    print("near max: %4d k=%.4e err=%.2e steps=%d failed=%d fevals=%d lus=%d regimes=%d" % (i, inp.k[i], err[i], stats[i].steps, stats[i].failed, stats[i].fevals, stats[i].factorisations, stats[i].n_regimes))
for i in range(0, inp.nk, max(inp.nk // 10, 1)):
    print("%4d k=%.4e err=%.2e steps=%d regimes=%d" % (i, inp.k[i], err[i], stats[i].steps, stats[i].n_regimes))
