"""Diagnostic: perturbation-kernel time of an alternative build of libcpt.so (python tools/kernel_time.py <lib> [config])."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from classpp_public_amd import capi

capi.LIB_PATH = os.path.abspath(sys.argv[1])
from classpp_public_amd.backend import Backend
from classpp_public_amd.inputs import Inputs

inp = Inputs(sys.argv[2] if len(sys.argv) > 2 else "lcdm")
be = Backend(inp)
ms = []
for i in range(4):
    be.perturb_solve(want_sources=False)
    ms.append(be.kernel_ms(0)[0])
print(os.path.basename(sys.argv[1]), "kernel ms", ["%.2f" % m for m in ms])
