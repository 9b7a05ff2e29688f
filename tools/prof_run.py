"""Diagnostic: in-kernel cycle breakdown of the heaviest k-mode (needs a libcpt built with -DCPT_PROFILE).
    hipcc ... -DCPT_PROFILE -o classpp_public_amd/csrc/libcpt_prof.so ...;  python tools/prof_run.py <lib> [config]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from classpp_public_amd import capi

capi.LIB_PATH = os.path.abspath(sys.argv[1])
from classpp_public_amd.backend import Backend
from classpp_public_amd.inputs import Inputs

inp = Inputs(sys.argv[2] if len(sys.argv) > 2 else "lcdm")
be = Backend(inp)
for i in range(2):
    src, stats, status = be.perturb_solve(want_sources=False)
print("kernel ms", be.kernel_ms(0))
os.environ["CPT_PROFILE_HELPER"] = "1"
out = (C.c_ulonglong * 32)()
L = capi.lib()
L.cpt_dbg_profile.argtypes = [C.POINTER(C.c_ulonglong)]
L.cpt_dbg_profile(out)
names = ["rhs(newton)", "lu_solve", "factorise", "jacobian+init", "sampling", "newton-control", "schedule", "total"]
tot = max(out[7], 1)
s = stats[len(stats) - 1]
print("heaviest mode: steps", s.steps, "fevals", s.fevals, "lus", s.factorisations, "solves", s.solves, "jacs", s.jacobians)
for n, v in zip(names, out):
    print("%-12s %12d cycles %5.1f%%" % (n, v, 100.0 * v / tot))
h = out[16:32]
print("helper wave: look-ups %d cycles / %d = %.0f each; inversions %d / %d = %.0f each; samples %d / %d = %.0f each; idle turns %d; alive %d cycles" % (
    h[0], h[1], h[0] / max(h[1], 1), h[2], h[3], h[2] / max(h[3], 1), h[4], h[5], h[4] / max(h[5], 1), h[6], h[7]))
print("inside every rhs call (all slots): take-row %d  gather+bcast %d  algebra %d  combine %d   [calls %d]" % (out[8], out[9], out[10], out[11], s.fevals))
print("post_step %d  new_step %d  errtest+accept %d  wait for the inverse %d" % (out[12], out[13], out[14], out[15]))
print("cycles/step %.0f  rhs cycles/call %.0f  solve cycles/call %.0f  factorise cycles/call %.0f  jac cycles/call %.0f" % (
    tot / s.steps, out[0] / s.solves, out[1] / s.solves, out[2] / max(s.factorisations, 1), out[3] / max(s.jacobians, 1)))
