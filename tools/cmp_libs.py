"""Diagnostic: source functions of two builds of libcpt.so against each other (python tools/cmp_libs.py <config> <libA> <libB>):
max over (k, tau) of |A - B| relative to the maximum over tau of each (type, k) column, per source type."""
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
if sys.argv[1] == "--one":
    sys.path.insert(0, os.path.dirname(HERE))
    from classpp_public_amd import capi
    capi.LIB_PATH = os.path.abspath(sys.argv[3])
    from classpp_public_amd.backend import Backend
    from classpp_public_amd.inputs import Inputs
    be = Backend(Inputs(sys.argv[2]))
    src, stats, status = be.perturb_solve()
    np.save(sys.argv[4], src.cpu().numpy())
else:
    cfg, a, b = sys.argv[1:4]
    for lib, out in ((a, "/tmp/cmp_a.npy"), (b, "/tmp/cmp_b.npy")):
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "--one", cfg, lib, out])
    A, B = np.load("/tmp/cmp_a.npy"), np.load("/tmp/cmp_b.npy")
    scale = np.max(np.abs(A), axis=1, keepdims=True) + 1e-300
    d = np.abs(A - B) / scale
    print(cfg, "source types", A.shape[0], "max relative difference per type:", " ".join("%.2e" % x for x in d.reshape(A.shape[0], -1).max(axis=1)))
    for tp in range(A.shape[0]):
        it, ik = np.unravel_index(np.argmax(d[tp]), d[tp].shape)
        perk = d[tp].max(axis=0)
        print("  type %d: worst at tau index %d, k index %d: A %.6e B %.6e column max %.3e; k-modes above 1e-3: %d, above 1e-4: %d of %d" % (
            tp, it, ik, A[tp, it, ik], B[tp, it, ik], scale[tp, 0, ik], int((perk > 1e-3).sum()), int((perk > 1e-4).sum()), perk.size))
    if len(sys.argv) > 4:      # ... and both against the CPU restatement (tests/oracle_lib.py) on the k-modes named
        sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tests"))
        sys.path.insert(0, os.path.dirname(HERE))
        import oracle_lib
        from classpp_public_amd.inputs import Inputs
        inp = Inputs(cfg)
        ks = [int(x) for x in sys.argv[4:]]
        O, ostats, _, _ = oracle_lib.perturb(inp, k=inp.k[ks])
        for j, ik in enumerate(ks):
            sc = np.max(np.abs(O[:, :, j]), axis=1, keepdims=True) + 1e-300
            ea, eb = np.abs(A[:, :, ik] - O[:, :, j]) / sc, np.abs(B[:, :, ik] - O[:, :, j]) / sc
            print("  k index %d (oracle steps %d): A vs oracle %s | B vs oracle %s" % (ik, ostats[j].steps, " ".join("%.1e" % x for x in ea.max(axis=1)), " ".join("%.1e" % x for x in eb.max(axis=1))))
