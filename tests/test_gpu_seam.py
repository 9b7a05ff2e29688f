"""The reference-side seam, RUNNING: the reference's own Cosmology / PrimordialModule / NonlinearModule / SpectraModule / LensingModule on the
sources_ and transfer_ tables the GPU backend filled.

oracle/_ref/ref_driver_cpt is the unmodified reference plus the patch of include/reference_side/cpt_seam.cpp (ten inserted lines, two edited loop
headers; oracle/apply_seam.py + `make -C oracle seam`, built in the container where /root/reference exists, travels to the GPU box as a binary):
with CPT_BACKEND=mi355x its PerturbationsModule and TransferModule constructors build their index maps and grids as always, then hand the k loop
(pm.cpp:668-718) and the q loop (tm.cpp:287-318) to cpt::PerturbationsModule / cpt::TransferModule through the adapter (cpt_adapter.h).
Everything downstream is the reference's code reading the reference's classes.  Compared with the golden vectors of the unpatched reference:
sources within the bands of tests/bands.py, C_l / lensed C_l / P(k) / sigma8 at the contract's 1e-4 (3e-4 for the coarse `small` precision file)."""
import os
import subprocess
import sys

import numpy as np
import pytest

import bands
from classpp_public_amd.inputs import Inputs
from test_oracle_perturb import check_sources

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "oracle", "_ref", "ref_driver_cpt")


def run_patched_reference(cfg, tmp_path, backend=True):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from make_fixtures import load_bin
    out = str(tmp_path / (cfg + ".bin"))
    env = dict(os.environ)
    if backend:
        env["CPT_BACKEND"] = "mi355x"
    else:
        env.pop("CPT_BACKEND", None)
    p = subprocess.run([DRIVER, "dump", os.path.join(ROOT, "tests", "golden", cfg + ".ini"), out], cwd=os.path.join(ROOT, "tests", "golden"),
                       capture_output=True, text=True, env=env)
    assert p.returncode == 0, p.stdout + p.stderr
    return load_bin(out)


@pytest.mark.parametrize("cfg", ["small", "explanatory", "ncdm_small", "iso_cdi"])
def test_reference_modules_run_on_the_gpu_backend(cfg, tmp_path):
    if not os.path.exists(DRIVER):
        pytest.skip("oracle/_ref/ref_driver_cpt (the patched reference, built where /root/reference exists) is not on this box")
    inp = Inputs(cfg)
    ref = inp.d
    got = run_patched_reference(cfg, tmp_path)
    # the modules' tables, in the reference's own layout
    if "pt.sources" in ref:
        check_sources(inp.config, got["pt.sources"], ref["pt.sources"])
    else:
        check_sources(inp.config, got["pt.sources"][:, :, ref["pt.sources_k_index"]], ref["pt.sources_subset"])
    tol = 1e-4 if cfg == "explanatory" else 3e-4
    worst = {}
    sp = inp.spectra
    a, b = got["sp.cl_table"], ref["sp.cl_table"]
    for name, idx, rel in (("tt", sp.index_ct_tt, True), ("ee", sp.index_ct_ee, True), ("pp", sp.index_ct_pp, True), ("te", sp.index_ct_te, False)):
        if idx >= 0:
            worst[name] = float(np.max(np.abs(a[:, idx] / b[:, idx] - 1)) if rel else np.max(np.abs(a[:, idx] - b[:, idx])) / np.max(np.abs(b[:, idx])))
    if "le.cl_lens" in ref:   # the reference's LensingModule on the reference's SpectraModule on the GPU's transfer table
        sel = ref["le.l"].astype(int) <= int(ref["le.l_lensed_max"][0])
        for name, idx in (("lensed tt", sp.index_ct_tt), ("lensed ee", sp.index_ct_ee), ("lensed bb", sp.index_ct_bb)):
            if idx >= 0:
                worst[name] = float(np.max(np.abs(got["le.cl_lens"][sel, idx] / ref["le.cl_lens"][sel, idx] - 1)))
    if "nl.pk_lin_z0" in ref:   # the reference's NonlinearModule on the GPU's sources_
        worst["pk"] = float(np.max(np.abs(got["nl.pk_lin_z0"] / ref["nl.pk_lin_z0"] - 1)))
        worst["sigma8"] = float(abs(got["nl.sigma8"][0] / ref["nl.sigma8"][0] - 1))
    print("\n[seam %s] reference modules on GPU tables, max errors vs the unpatched reference: %s" % (cfg, ", ".join("%s %.1e" % kv for kv in worst.items())))
    for name, err in worst.items():
        assert err < tol, (name, err)
    # exact zeros where the reference neglects a transfer function (integer decisions)
    if "tr.transfer" in ref:
        assert np.array_equal(got["tr.transfer"] == 0, ref["tr.transfer"] == 0)


def test_patched_reference_without_the_backend_is_the_reference(tmp_path):
    """the patch changes nothing unless the backend is asked for: bit-identical tables"""
    if not os.path.exists(DRIVER):
        pytest.skip("oracle/_ref/ref_driver_cpt is not on this box")
    got = run_patched_reference("small", tmp_path, backend=False)
    ref = Inputs("small").d
    assert np.array_equal(got["sp.cl_table"], ref["sp.cl_table"]) and np.array_equal(got["pt.sources"], ref["pt.sources"])
    assert np.array_equal(got["tr.transfer"], ref["tr.transfer"])
