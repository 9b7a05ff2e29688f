"""Diagnostic: what a caller of the classy surface pays per new cosmology (python tools/classy_time.py [config]): Class.set ->
compute -> lensed_cl -> pk at 100 wavenumbers, one parameter changed every round so that nothing is reused; wall time per
round and the Python-side profile of the last one."""
import cProfile
import os
import pstats
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from classpp_public_amd import classy  # noqa: E402
from classpp_public_amd.inputs import GOLDEN  # noqa: E402
from classpp_public_amd.pipeline import read_ini  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "explanatory_mpk"
pars = {k: v for k, v in read_ini(os.path.join(GOLDEN, cfg + ".ini")).items() if k != "threads"}
ks = np.logspace(-3, -0.5, 100)


def one(c, h):
    c.set({"h": repr(h)}) if "h" in pars else c.set({"H0": repr(100. * h)})
    c.compute()
    cl = c.lensed_cl() if "lCl" in pars.get("output", "") else c.raw_cl()
    pk = [c.pk(k, 0.) for k in ks] if "mPk" in pars.get("output", "") else None
    return cl, pk


c = classy.Class(pars)
one(c, 0.67)
ms = []
for i in range(6):
    t = time.perf_counter()
    one(c, 0.67 + 0.001 * (i + 1))
    ms.append((time.perf_counter() - t) * 1e3)
print("classy round (set, compute, lensed_cl, pk x100) ms:", " ".join("%.1f" % m for m in ms), flush=True)
pr = cProfile.Profile()
pr.enable()
one(c, 0.68)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
