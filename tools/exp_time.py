"""Diagnostic: perturbation-kernel time and the integrator statistics of the heaviest k-mode for alternative builds of libcpt.so
(python tools/exp_time.py <config> <lib> [<lib> ...]; one process per library)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
if sys.argv[1] == "--one":
    sys.path.insert(0, os.path.dirname(HERE))
    from classpp_public_amd import capi
    capi.LIB_PATH = os.path.abspath(sys.argv[3])
    from classpp_public_amd.backend import Backend
    from classpp_public_amd.inputs import Inputs
    be = Backend(Inputs(sys.argv[2]))
    ms = []
    for i in range(5):
        _, stats, _ = be.perturb_solve(want_sources=False)
        ms.append(be.kernel_ms(0)[0])
    big = max(stats, key=lambda s: s.steps)
    print("%-28s kernel ms %s | heaviest mode: steps %d failed %d fevals %d jacs %d lus %d solves %d" % (
        os.path.basename(sys.argv[3]), " ".join("%.2f" % m for m in ms), big.steps, big.failed, big.fevals, big.jacobians, big.factorisations, big.solves), flush=True)
else:
    for lib in sys.argv[2:]:
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "--one", sys.argv[1], lib])
