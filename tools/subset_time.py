"""Diagnostic: is the perturbation kernel's time the time of its heaviest k-mode alone?  The kernel on the full k grid, on the n
largest wavenumbers only, and on the largest one alone (python tools/subset_time.py [config])."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from classpp_public_amd.backend import Backend
from classpp_public_amd.inputs import Inputs

cfg = sys.argv[1] if len(sys.argv) > 1 else "explanatory_mpk"
full = Inputs(cfg).k.size
for n in (full, 512, 256, 128, 32, 1):
    if n > full:
        continue
    inp = Inputs(cfg)
    inp.k = np.ascontiguousarray(inp.k[full - n:])
    inp.k_size_cl = min(inp.k_size_cl, n) if hasattr(inp, "k_size_cl") else n
    be = Backend(inp)
    ms = []
    for i in range(4):
        _, stats, status = be.perturb_solve(want_sources=False)
        ms.append(be.kernel_ms(0)[0])
    print("%-16s %4d largest k: kernel ms %s | most steps %d, status %d" % (cfg, n, " ".join("%.2f" % m for m in ms), max(s.steps for s in stats), int(np.abs(status).max())), flush=True)
    be.close()
