"""Diagnostic: status, step counts and source parity of a configuration with a given build of the library."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from classpp_public_amd import capi

capi.LIB_PATH = os.path.abspath(sys.argv[1])
from classpp_public_amd.backend import Backend, CptError
from classpp_public_amd.inputs import Inputs
from test_oracle_perturb import check_sources

inp = Inputs(sys.argv[2])
be = Backend(inp)
tag = "%s %s compact=%s split=%s" % (os.path.basename(sys.argv[1]), sys.argv[2], os.environ.get("CPT_NCDM_COMPACT", "-"), os.environ.get("CPT_NCDM_SPLIT", "-"))
for rep in range(2):
    try:
        src, stats, status = be.perturb_solve()
        got = src.cpu().numpy()
        ks = inp.d["pt.sources_k_index"]
        try:
            check_sources(inp.config, got[:, :, ks], inp.d["pt.sources_subset"])
            print(tag, "run", rep, "ok, sources match; total steps", sum(s.steps for s in stats))
        except AssertionError as e:
            ref = inp.d["pt.sources_subset"]
            bad = [int(ks[j]) for j in range(len(ks)) if np.max(np.abs(got[:, :, ks[j]] - ref[:, :, j])) > 1e-2 * np.max(np.abs(ref[:, :, j]))]
            print(tag, "run", rep, "SOURCES DIFFER", str(e)[:80], "bad k indices", bad, "steps there", [stats[i].steps for i in bad], "regimes", [stats[i].n_regimes for i in bad])
    except CptError as e:
        print(tag, "run", rep, "FAILED:", str(e)[:150])
