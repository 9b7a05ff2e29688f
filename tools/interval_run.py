"""Diagnostic: steps and cycles per approximation interval of the heaviest k-mode (needs a libcpt built with -DCPT_PROFILE_INTERVALS).
    python tools/interval_run.py <lib> [config]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from classpp_public_amd import capi

capi.LIB_PATH = os.path.abspath(sys.argv[1])
from classpp_public_amd.backend import Backend
from classpp_public_amd.inputs import Inputs

inp = Inputs(sys.argv[2] if len(sys.argv) > 2 else "lcdm")
be = Backend(inp)
src, stats, status = be.perturb_solve(want_sources=False)
print("kernel ms", be.kernel_ms(0))
