"""The reference-side binding of INTEGRATION.md - include/reference_side/cpt_adapter.h, the file a maintainer adds to the reference
tree - compiled against the UNMODIFIED reference headers (oracle/Makefile, target `adapter`) and run on the reference's own
InputModule / BackgroundModule / ThermodynamicsModule: the cpt::Inputs it produces must equal, field by field, the one this repository
builds from the committed fixtures (which were dumped from the same reference by oracle/ref_driver.cpp).  That pins the adapter's
reading of every struct member, its index maps (perturb_indices_of_perturbs, transfer_indices_of_transfers) and the table pointers."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from classpp_public_amd import hostlib  # noqa: E402
from classpp_public_amd.inputs import Inputs  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
EXE = os.path.join(ROOT, "oracle", "_ref", "adapter_check")
pytestmark = pytest.mark.skipif(not os.path.exists(EXE), reason="oracle/_ref/adapter_check is built where /root/reference is present (__graft_entry__.build)")


def _run(cfg):
    out = subprocess.run([EXE, os.path.join(GOLDEN, cfg + ".ini")], capture_output=True, text=True, cwd=GOLDEN, check=True).stdout
    return {ln.split()[0]: ln.split()[1] for ln in out.splitlines() if len(ln.split()) == 2}


def _checksum(a):
    a = np.ascontiguousarray(a, dtype=np.float64).reshape(-1)
    return float(np.sum(a * (1. + (np.arange(a.size) % 97) / 97.)))


def _same(a, b, tol=1e-15):
    return a == b or abs(a - b) <= tol * max(abs(a), abs(b))     # (scalars: %.17g round trip = equal)


def _same_sum(a, b):
    return _same(a, b, 1e-10)                                     # (checksums: numpy sums pairwise, the driver sequentially)


def _compare_struct(tag, got, struct, skip=()):
    bad = []
    for name, ctype in struct._fields_:
        key = "%s.%s" % (tag, name)
        if name in skip or key not in got:
            continue
        if isinstance(ctype, type) and issubclass(ctype, C.Array):      # (printed element by element: "<key>.<i> <value>")
            for i, want in enumerate(getattr(struct, name)):
                if int(got["%s.%d" % (key, i)]) != int(want):
                    bad.append(("%s.%d" % (key, i), int(got["%s.%d" % (key, i)]), int(want)))
            continue
        want = getattr(struct, name)
        have = float(got[key]) if ctype is C.c_double else int(got[key])
        if not _same(float(have), float(want)):
            bad.append((key, have, want))
    return bad


@pytest.mark.parametrize("cfg", ["lcdm", "explanatory_mpk", "small", "curved", "open", "newt", "iso_cdi", "iso_nid", "tens", "tens_curved",
                                 "ncdm_small", "ncdm3", "ncdm3_tens", "small_tk", "newt_tk", "ncdm_small_tk", "ncdm3_small_tk"])
def test_adapter_inputs_equal_the_fixture_inputs(cfg):
    got = _run(cfg)
    inp = Inputs(cfg)
    c = inp.config
    seen = [n for n, _ in c._fields_ if "config." + n in got]
    assert len(seen) == len(c._fields_), sorted(set(n for n, _ in c._fields_) - set(seen))     # the driver prints every member of cpt_config
    # (fixtures dumped by an older ref_driver lack a few entries that their mode does not read; Inputs then holds a placeholder)
    d = inp.d
    skip = [f for f, key in (("transfer_neglect_delta_k_T_t2", "ppr.transfer_neglect_delta_k_T_t2"), ("transfer_neglect_delta_k_T_e", "ppr.transfer_neglect_delta_k_T_e"),
                             ("transfer_neglect_delta_k_T_b", "ppr.transfer_neglect_delta_k_T_b"), ("tol_ncdm_initial_w", "ppr.tol_ncdm_initial_w"),
                             ("tensor_method", "ppt.tensor_method"), ("entropy_ini", "ppr.entropy_ini"), ("index_tp_delta_cb", "pt.index_tp_delta_cb"), ("index_tp_delta_ncdm1", "pt.index_tp_delta_ncdm1"), ("index_tp_theta_ncdm1", "pt.index_tp_theta_ncdm1"),
                             ("index_tt_b", "tr.index_tt_b")) if key not in d]
    bad = _compare_struct("config", got, c, skip)
    assert not bad, bad
    assert int(got["with_tensors"]) == 0 and int(got["n_ic"]) == 1 and int(got["ic.0"]) == c.ic
    # tables: sizes, index map, and the arrays behind the pointers
    t = inp.t
    tb = inp.tables
    skip = {n for n, ct in tb._fields_ if not (ct is C.c_int or ct is C.c_double)}
    if not c.has_ncdm:
        skip |= {"index_bg_rho_ncdm1", "index_bg_p_ncdm1", "index_bg_pseudo_p_ncdm1"}   # (only read with non-cold species)
    bad = _compare_struct("tables", got, tb, skip)
    assert not bad, bad
    for key, arr in (("tau_table", t["bg.tau_table"]), ("background_table", t["bg.background_table"]),
                     ("d2background_dtau2_table", t["bg.d2background_dtau2_table"]), ("z_table", t["th.z_table"]),
                     ("thermodynamics_table", t["th.thermodynamics_table"]), ("d2thermodynamics_dz2_table", t["th.d2thermodynamics_dz2_table"])):
        assert _same_sum(float(got["sum." + key]), _checksum(arr)), key
    for n in range(c.N_ncdm if c.has_ncdm else 0):
        assert int(got["ncdm.%d.q_size" % n]) == t["ncdm.q_%d" % n].size
        assert _same(float(got["ncdm.%d.M" % n]), float(t["ncdm.M"][n])) and _same(float(got["ncdm.%d.factor" % n]), float(t["ncdm.factor"][n]))
        for a, b in (("sum_q", "ncdm.q_%d"), ("sum_w", "ncdm.w_%d"), ("sum_dlnf0", "ncdm.dlnf0_dlnq_%d")):
            assert _same_sum(float(got["ncdm.%d.%s" % (n, a)]), _checksum(t[b % n])), (n, a)
    g = hostlib.grid_params(inp)
    bad = _compare_struct("grid", got, g, skip=("l_tensor_max",) if "ppt.l_tensor_max" not in inp.d else ())
    assert not bad, bad


def test_adapter_modes_and_initial_conditions():
    """modes = s,t -> with_tensors and the tensor mode's own index maps; several initial conditions -> ic[] in the reference's order"""
    got = _run("sc_st_lens")
    assert int(got["with_tensors"]) == 1 and int(got["config.mode"]) == 0 and int(got["config_tensors.mode"]) == 1
    assert (int(got["config_tensors.index_tp_t2"]), int(got["config_tensors.index_tp_p"]), int(got["config_tensors.tp_size"])) == (0, 1, 2)
    assert (int(got["config_tensors.index_tt_t2"]), int(got["config_tensors.index_tt_e"]), int(got["config_tensors.index_tt_b"]),
            int(got["config_tensors.tt_size"])) == (0, 1, 2, 3)
    assert int(got["config_tensors.index_tp_t0"]) == -1 and int(got["config_tensors.index_tt_lcmb"]) == -1
    assert int(got["config_tensors.evolve_tensor_ur"]) == 1 and int(got["config.evolve_tensor_ur"]) == 0
    tens = Inputs("tens").config   # the tensors-only fixture of the same cosmology: same index maps
    for f in ("tp_size", "index_tp_t2", "index_tp_p", "tt_size", "index_tt_t2", "index_tt_e", "index_tt_b", "evolve_tensor_ur"):
        assert int(got["config_tensors." + f]) == getattr(tens, f), f
    ini = os.path.join(GOLDEN, "_adapter_two_ic.ini")
    try:
        with open(os.path.join(GOLDEN, "iso_cdi.ini")) as f:
            text = f.read().replace("ic = cdi", "ic = ad,cdi,niv")
        with open(ini, "w") as f:
            f.write(text)
        got = _run("_adapter_two_ic")
    finally:
        if os.path.exists(ini):
            os.remove(ini)
    assert int(got["n_ic"]) == 3 and [int(got["ic.%d" % i]) for i in range(3)] == [0, 2, 4] and int(got["config.ic"]) == 0
