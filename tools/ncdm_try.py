#!/usr/bin/env python3
"""Diagnostic: integrate an ncdm fixture on the GPU and compare the sources with the reference's (tests/golden)."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from classpp_public_amd.backend import Backend
from classpp_public_amd.inputs import Inputs

cfg = sys.argv[1] if len(sys.argv) > 1 else "ncdm_small"
nk = int(sys.argv[2]) if len(sys.argv) > 2 else 0
inp = Inputs(cfg)
be = Backend(inp)
ks = np.arange(inp.nk) if nk == 0 else np.unique(np.round(np.linspace(0, inp.nk - 1, nk)).astype(int))
t0 = time.time()
try:
    src, stats, status = be.perturb_solve(k=inp.k[ks])
except Exception as e:
    print("FAILED:", e)
    sys.exit(1)
print("status", status[:10], "wall %.3f s" % (time.time() - t0), "kernel ms", be.kernel_ms(0))
got = src.cpu().numpy()
ref = inp.d["pt.sources"][:, :, ks] if "pt.sources" in inp.d else None
if ref is None:
    kk = inp.d["pt.sources_k_index"]; ref = inp.d["pt.sources_subset"]; got = got[:, :, kk]
print("steps", sum(s.steps for s in stats), "fevals", sum(s.fevals for s in stats), "regimes", [s.n_regimes for s in stats][:8])
for tp in range(ref.shape[0]):
    sc = np.abs(ref[tp]).max(axis=0, keepdims=True); sc[sc == 0] = 1
    err = np.abs(got[tp] - ref[tp]) / sc
    print("tp", tp, "max %.2e" % err.max(), "at k idx", int(err.max(axis=0).argmax()), "finite", bool(np.isfinite(got[tp]).all()))
for j in range(got.shape[2]):
    bad = ~np.isfinite(got[:, :, j]).all(axis=0)
    first = int(np.argmax(bad)) if bad.any() else -1
    ok = slice(0, first if first >= 0 else None)
    errs = []
    for tp in range(ref.shape[0]):
        sc = np.abs(ref[tp, :, j]).max() or 1
        errs.append(np.abs(got[tp, ok, j] - ref[tp, ok, j]).max() / sc if (first != 0) else np.nan)
    print("k[%d]=%.3e first bad it %d (tau %.1f) steps %d regimes %d; err before: %s" % (j, inp.k[ks[j]], first, inp.tau[first] if first >= 0 else -1, stats[j].steps, stats[j].n_regimes,
          " ".join("%.1e" % e for e in errs)))
