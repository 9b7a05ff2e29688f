"""Diagnostic: perturbation-kernel time of a configuration with a given build of the library (A/B of two .so files on one box).
    python tools/ab_run.py <lib> <config> [repeats]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from classpp_public_amd import capi

capi.LIB_PATH = os.path.abspath(sys.argv[1])
from classpp_public_amd.backend import Backend
from classpp_public_amd.inputs import Inputs

inp = Inputs(sys.argv[2])
be = Backend(inp)
ms = []
for i in range(int(sys.argv[3]) if len(sys.argv) > 3 else 6):
    be.perturb_solve(want_sources=False)
    ms.append(be.kernel_ms(0)[0])
print(os.path.basename(sys.argv[1]), sys.argv[2], "min %.2f ms  median %.2f ms" % (min(ms[1:]), sorted(ms[1:])[len(ms[1:]) // 2]))
