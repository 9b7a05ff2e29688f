"""Diagnostic: perturbation kernel time on N-fold densified k grids (python tools/dense_try.py <config> <N> ...), to place the
crossover between the full-register and the half-register build (CPT_WAVES_PER_SIMD / CPT_NCDM_WAVES_PER_SIMD override it)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from classpp_public_amd.backend import Backend
from classpp_public_amd.inputs import Inputs
from classpp_public_amd.sharded import densify_k

cfg = sys.argv[1]
inp = Inputs(cfg)
be = Backend(inp)
for n in [int(a) for a in sys.argv[2:]]:
    k = densify_k(inp.k, n)
    for rep in range(2):
        be.perturb_solve(k=k, want_sources=False)
    print("%s x%d: %d k-modes, kernel %.2f ms" % (cfg, n, k.size, be.kernel_ms(0)[0]), flush=True)
be.close()
