"""Diagnostic: per-component comparison of the GPU RHS with the oracle's for one (k, tau, regime)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib
from classpp_public_amd.backend import Backend
from classpp_public_amd.inputs import Inputs

inp = Inputs("small")
be = Backend(inp)
k, tau = float(sys.argv[1]), float(sys.argv[2])
flags = tuple(int(x) for x in sys.argv[3:6])
rng = np.random.default_rng(1)
y = rng.normal(size=64)
want = oracle_lib.derivs(inp, k, tau, *flags, y)
got = be.dbg_derivs(k, tau, *flags, y[: want.size])
for i, (a, b) in enumerate(zip(got, want)):
    print("%2d % .15e % .15e  %.2e" % (i, a, b, abs(a - b) / max(abs(b), 1e-300)))
