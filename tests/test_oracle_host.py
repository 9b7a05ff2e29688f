"""ORACLE PINNING: the bit-exact restatement of the reference's background / thermodynamics modules (oracle/restate/host/) against the
tables dumped from the unmodified reference (tests/golden/tables_*.npz).  Same integration variable, integrators and spline
recurrences => required bit for bit.  This restatement is the checker of the product's own host numerics (tests/test_host_cosmo.py)."""
import os

import numpy as np
import pytest

import oracle_lib
from classpp_public_amd import hostlib
from classpp_public_amd.inputs import Inputs


@pytest.mark.parametrize("cfg", ["lcdm", "curved", "open", "ncdm_small", "ncdm3_small"])
def test_oracle_background_table_bit_exact(cfg):
    inp = Inputs(cfg)
    t = inp.t
    bg = oracle_lib.host_background(inp)
    assert bg["bg.bt_size"] == int(t["bg.bt_size"][0]) and bg["bg.bg_size"] == int(t["bg.bg_size"][0])
    for key in ("bg.tau_table", "bg.z_table", "bg.background_table", "bg.d2background_dtau2_table"):
        assert np.array_equal(bg[key], t[key]), key
    assert bg["bg.conformal_age"] == float(t["bg.conformal_age"][0]) and bg["bg.Omega0_m"] == float(t["bg.Omega0_m"][0])


@pytest.mark.parametrize("cfg", ["lcdm", "curved", "open"])
def test_oracle_thermodynamics_table_bit_exact(cfg):
    inp = Inputs(cfg)
    t = inp.t
    th = oracle_lib.host_thermodynamics(inp)
    assert th["th.tt_size"] == int(t["th.tt_size"][0]) and th["th.th_size"] == int(t["th.th_size"][0])
    for key in ("th.z_table", "th.thermodynamics_table", "th.d2thermodynamics_dz2_table"):
        assert np.array_equal(th[key], t[key]), key
    for key in ("tau_ini", "YHe", "n_e", "z_rec", "tau_rec", "rs_rec", "ra_rec", "angular_rescaling", "tau_free_streaming", "tau_cut",
                "z_reionization"):
        assert th["th." + key] == float(t["th." + key][0]), key


def test_oracle_reionization_from_optical_depth():
    """tau_reio given instead of z_reio: the bisection of th.cpp:2222-2318 (fixture lcdm_taureio = lcdm.ini with tau_reio = 0.0925)"""
    inp = Inputs("lcdm")
    ref = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "lcdm_taureio.npz")))
    tp = hostlib.thermo_params(inp)
    tp.reio_from_tau = 1
    tp.tau_reio = float(ref["pth.tau_reio"][0])
    tp.z_reio = 0.
    th = oracle_lib.host_thermodynamics(inp, tp=tp)
    assert th["th.z_reionization"] == float(ref["th.z_reionization"][0])
    rows = ref["th.row_index"]
    assert np.array_equal(th["th.z_table"][rows], ref["th.z_table_rows"])
    assert np.array_equal(th["th.thermodynamics_table"][rows], ref["th.thermodynamics_table_rows"])
