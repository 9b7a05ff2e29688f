"""Diagnostic: time per wave-cooperative table lookup (k_dbg_lookup over n increasing tau's)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from classpp_public_amd.backend import Backend
from classpp_public_amd.inputs import Inputs
import torch
inp = Inputs("lcdm")
be = Backend(inp)
for n, lo, hi in [(20000, 1.0, 14000.0), (20000, 250.0, 400.0), (20000, 3000.0, 3001.0)]:
    taus = np.linspace(lo, hi, n)
    be.dbg_lookup(taus[:10])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    be.dbg_lookup(taus)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("n=%d tau in [%g,%g]: %.3f ms total, %.0f ns per lookup (~%.0f ticks at 2.4 GHz)" % (n, lo, hi, dt * 1e3, dt / n * 1e9, dt / n * 2.4e9))
