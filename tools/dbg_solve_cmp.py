"""Diagnostic: (I - hg J) x = b through the kernel's structured factorisation vs a dense numpy solve, per component."""
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
k, tau, hg = (float(x) for x in sys.argv[1:4])
flags = tuple(int(x) for x in sys.argv[4:7])
n = oracle_lib.derivs(inp, k, tau, *flags, np.zeros(64)).size
J = np.zeros((n, n))
for j in range(n):
    e = np.zeros(64)
    e[j] = 1.0
    J[:, j] = oracle_lib.derivs(inp, k, tau, *flags, e)
A = np.eye(n) - hg * J
b = np.random.default_rng(2).normal(size=n)
want = np.linalg.solve(A, b)
got = be.dbg_solve(k, tau, *flags, hg, b)
res = A @ got - b
for i in range(n):
    print("%2d % .12e % .12e  res % .2e" % (i, got[i], want[i], res[i]))
