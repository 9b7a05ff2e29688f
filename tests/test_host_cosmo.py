"""Host-side background / thermodynamics tables (include/cpt_host.h, classpp_public_amd/host/cpt_cosmo.cpp): the
reference's MODEL with this project's own NUMERICS (one Dormand-Prince 5(4) integrator at 1e-10 / 1e-9, one factorise-once spline
solver).  Checked three ways:
  * against the tables dumped from the unmodified reference, at the accuracy of the REFERENCE's integration (its evolver runs the
    background at rtol 1e-6: conformal time 3e-6, everything that does not involve an integral: round-off);
  * against the bit-exact restatement of the reference (oracle/restate/host/, pinned by tests/test_oracle_host.py) run at a tolerance
    of 1e-11: the limit the reference's own table converges to - agreement 3e-8 or better;
  * downstream: C_l and P(k) from parameters alone within 1e-4 of the reference (tests/test_gpu_end_to_end.py, on the GPU).
"""
import os

import numpy as np
import pytest

import oracle_lib
from classpp_public_amd import hostlib
from classpp_public_amd.inputs import Inputs

REF_INTEGRATION_ERROR = 5e-6     # of the reference's background evolver (rtol 1e-6, measured 2.7e-6 on tau)


def _colmax_err(a, b):
    scale = np.maximum(np.abs(b).max(axis=0), 1e-300)
    return (np.abs(a - b) / scale).max(axis=0)


@pytest.mark.parametrize("cfg", ["lcdm", "curved", "open", "ncdm_small", "ncdm3_small"])
def test_background_table_against_the_reference(cfg):
    """flat, closed and open LambdaCDM + massless neutrinos, one / three massive neutrino species (25 / 33 columns)"""
    inp = Inputs(cfg)
    t = inp.t
    bg = hostlib.background(inp)
    assert bg["bg.bt_size"] == int(t["bg.bt_size"][0]) and bg["bg.bg_size"] == int(t["bg.bg_size"][0])
    for key in t.keys():
        if key.startswith("bg.index_bg_"):
            assert bg[key] == int(t[key][0]), key
    assert np.array_equal(bg["bg.z_table"], t["bg.z_table"])
    assert np.max(np.abs(bg["bg.tau_table"] / t["bg.tau_table"] - 1)) < REF_INTEGRATION_ERROR
    err = _colmax_err(bg["bg.background_table"], t["bg.background_table"])
    integrated = [bg["bg.index_bg_" + n] for n in ("conf_distance", "ang_distance", "lum_distance", "time", "rs", "D", "f")]
    for c in range(err.size):
        assert err[c] < (REF_INTEGRATION_ERROR if c in integrated else 1e-14), (c, err[c])
    assert abs(bg["bg.conformal_age"] / float(t["bg.conformal_age"][0]) - 1) < 2e-7
    assert abs(bg["bg.Omega0_m"] / float(t["bg.Omega0_m"][0]) - 1) < 1e-14
    # second derivatives: not comparable node by node (the reference's tau nodes carry uncorrelated 1e-6 errors, which a second
    # difference over a 7e-3 step in ln a turns into O(1)); what matters is the interpolant between the nodes
    rng = np.random.default_rng(0)
    tau_ref = t["bg.tau_table"]
    u = np.sort(rng.uniform(np.log(tau_ref[1]), np.log(tau_ref[-2]), 4000))
    for col in ("a", "H", "rho_g", "rho_b"):
        c = bg["bg.index_bg_" + col]
        mine = _spline(bg["bg.tau_table"], bg["bg.background_table"][:, c], bg["bg.d2background_dtau2_table"][:, c], np.exp(u))
        ref = _spline(tau_ref, t["bg.background_table"][:, c], t["bg.d2background_dtau2_table"][:, c], np.exp(u))
        assert np.max(np.abs(mine / ref - 1)) < 3 * REF_INTEGRATION_ERROR, col   # H ~ tau^-2 etc.: a few times the error in tau


def _spline(x, y, m, v):
    i = np.clip(np.searchsorted(x, v, side="right") - 1, 0, x.size - 2)
    h = x[i + 1] - x[i]
    b = (v - x[i]) / h
    a = 1 - b
    return a * y[i] + b * y[i + 1] + ((a ** 3 - a) * m[i] + (b ** 3 - b) * m[i + 1]) * h * h / 6


@pytest.mark.parametrize("cfg", ["lcdm", "open", "ncdm3_small"])
def test_background_is_the_limit_the_reference_converges_to(cfg):
    """the bit-exact restatement of the reference's integration with its tolerance tightened from 1e-6 to 1e-11 lands on this
    library's table: the 3e-6 distance to the reference's own table is the reference's integration error, not ours"""
    inp = Inputs(cfg)
    mine = hostlib.background(inp)
    tight = oracle_lib.host_background(inp, rtol=1e-11)
    assert np.max(np.abs(mine["bg.tau_table"] / tight["bg.tau_table"] - 1)) < 3e-8   # (two orders below the reference's own error)
    assert np.max(_colmax_err(mine["bg.background_table"], tight["bg.background_table"])) < 3e-8
    loose = oracle_lib.host_background(inp)
    assert np.max(np.abs(loose["bg.tau_table"] / tight["bg.tau_table"] - 1)) > 1e-6     # (and the reference's table is NOT there)


def test_background_rejects_what_it_does_not_know():
    inp = Inputs("lcdm")
    p = hostlib.cosmo_params(inp)
    p.has_fld = 1
    with pytest.raises(ValueError, match="only photons, baryons, cdm"):
        hostlib.background(inp, p)
    p = hostlib.cosmo_params(inp)
    p.a_ini_over_a_today_default = 1e-3   # not radiation dominated (the reference's class_test, background_module.cpp:1654)
    with pytest.raises(ValueError, match="not close enough to 1"):
        hostlib.background(inp, p)


# tolerance per thermodynamics column, relative to the column's largest entry.  kappa''' is the second derivative of a spline through
# kappa' - a difference quotient of 1e-6 noise - and carries no information at this level; it only enters the sampling-rate heuristic.
_TH_TOL = {"xe": 3e-6, "dkappa": 3e-6, "tau_d": 1e-6, "ddkappa": 1e-4, "exp_m_kappa": 1e-6, "g": 1e-6, "dg": 1e-6, "ddg": 3e-5, "Tb": 1e-6,
           "wb": 1e-6, "cb2": 1e-5, "rate": 1e-5}


@pytest.mark.parametrize("cfg", ["lcdm", "curved", "open"])
def test_thermodynamics_table_against_the_reference(cfg):
    """RECFAST 1.5 with the smoothed Saha / rate-equation hand-overs, CAMB-like reionization sampled adaptively, baryon temperature,
    merged table, opacity and visibility columns, smoothed rate, second derivatives in z, and every scalar the hot path reads.
    Same nodes as the reference (the redshift grid and the adaptive reionization sampling are part of the model)."""
    inp = Inputs(cfg)
    t = inp.t
    th = hostlib.thermodynamics(inp)
    assert th["th.tt_size"] == int(t["th.tt_size"][0]) and th["th.th_size"] == int(t["th.th_size"][0])
    assert np.max(np.abs(th["th.z_table"] - t["th.z_table"])) < 1e-8
    err = _colmax_err(th["th.thermodynamics_table"], t["th.thermodynamics_table"])
    for name, tol in _TH_TOL.items():
        assert err[th["th.index_th_" + name]] < tol, (name, err[th["th.index_th_" + name]])
    for key in t.keys():
        if key.startswith("th.index_th_"):
            assert th[key] == int(t[key][0]), key
    for key in ("tau_ini", "z_rec", "tau_rec", "rs_rec", "ra_rec", "angular_rescaling", "tau_free_streaming", "tau_cut"):
        assert abs(th["th." + key] / float(t["th." + key][0]) - 1) < 3e-7, key
    for key in ("YHe", "n_e", "z_reionization"):
        assert abs(th["th." + key] / float(t["th." + key][0]) - 1) < 1e-14, key


def test_reionization_from_optical_depth():
    """tau_reio given instead of z_reio: bisection on the optical depth (fixture lcdm_taureio = lcdm.ini with tau_reio = 0.0925)"""
    inp = Inputs("lcdm")
    ref = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "lcdm_taureio.npz")))
    tp = hostlib.thermo_params(inp)
    tp.reio_from_tau = 1
    tp.tau_reio = float(ref["pth.tau_reio"][0])
    tp.z_reio = 0.
    th = hostlib.thermodynamics(inp, tp=tp)
    # the bisection stops at a relative width of reionization_optical_depth_tol = 1e-4 in tau: z_reio is defined to that accuracy
    assert abs(th["th.z_reionization"] / float(ref["th.z_reionization"][0]) - 1) < 2e-4
    assert abs(th["th.tau_reionization"] - tp.tau_reio) < 1e-4 * tp.tau_reio
    for key in ("tau_rec", "rs_rec", "tau_free_streaming", "tau_cut"):
        assert abs(th["th." + key] / float(ref["th." + key][0]) - 1) < 3e-7, key


def test_thermodynamics_errors():
    inp = Inputs("lcdm")
    tp = hostlib.thermo_params(inp)
    tp.YHe = 0.7
    with pytest.raises(ValueError, match="out of bounds"):
        hostlib.thermodynamics(inp, tp=tp)
    tp = hostlib.thermo_params(inp)
    tp.reio_parametrization = 2   # reio_bins_tanh
    with pytest.raises(ValueError, match="none and camb only"):
        hostlib.thermodynamics(inp, tp=tp)
    tp = hostlib.thermo_params(inp)
    tp.z_reio = 60.               # would start above reionization_z_start_max
    with pytest.raises(ValueError, match="reionization_z_start_max"):
        hostlib.thermodynamics(inp, tp=tp)


@pytest.mark.parametrize("cfg", ["small", "lcdm", "curved", "open", "tens", "tens_curved", "ncdm_small", "ncdm", "ncdm3", "ncdm3_tens", "ncdm_k3000"])
def test_parameter_inputs_reproduce_the_fixture_inputs(cfg):
    """classpp_public_amd/pipeline.py: tables and grids computed on the host from parameters alone against what the reference handed
    over.  The grid RULES are the reference's (tests/test_host_grids.py checks them bit for bit on the reference's tables); fed with this
    library's own tables they give the same number of points and values within the reference's integration error."""
    from classpp_public_amd.pipeline import ParameterInputs
    a, b = ParameterInputs(cfg), Inputs(cfg)
    for name in ("k", "tau", "q"):
        x, y = getattr(a, name), getattr(b, name)
        assert x.shape == y.shape, name
        assert np.max(np.abs(x / y - 1)) < REF_INTEGRATION_ERROR, name
    assert a.k_size_cl == b.k_size_cl and np.array_equal(a.l, b.l)
    for field, _ in type(a.config)._fields_:
        x, y = getattr(a.config, field), getattr(b.config, field)
        if isinstance(x, float):
            assert abs(x - y) <= REF_INTEGRATION_ERROR * abs(y), field
        elif hasattr(x, "__len__"):          # (ctypes arrays: cpt_config::index_tp_transfer)
            assert list(x) == list(y), field
        else:
            assert x == y, field


def test_replaced_cosmology_reaches_the_config():
    """ParameterInputs(name, cosmology=...) without `params`: the replaced densities, H0 and curvature must be the ones the device
    configuration carries (they were once re-read from the committed fixture while the tables followed the new cosmology)."""
    from classpp_public_amd.pipeline import ParameterInputs
    base = Inputs("small")
    inp = ParameterInputs("small", cosmology=dict(h=0.60, omega_b=0.0200, omega_cdm=0.14, Omega_k=-0.02), YHe=0.25, z_reio=8.5, n_s=0.95)
    c = inp.config
    H0 = 0.60 * 1e5 / 2.99792458e8
    assert abs(c.H0 / H0 - 1) < 1e-14 and c.H0 != base.config.H0
    assert c.sgnK == 1 and c.has_curvature == 1 and abs(c.K / (0.02 * H0 * H0) - 1) < 1e-12
    assert c.tau0 == float(inp.t["bg.conformal_age"][0]) and c.tau0 != base.config.tau0
    assert abs(float(inp.d["pba.H0"][0]) / H0 - 1) < 1e-14


def test_host_tables_repeat_bit_for_bit():
    """The look-up hints and the integrators' kept stages are state of one call: two calls on the same parameters return the same bits."""
    inp = Inputs("lcdm")
    a, b = hostlib.thermodynamics(inp), hostlib.thermodynamics(inp)
    assert a.keys() == b.keys()
    for key in a:
        assert np.array_equal(np.asarray(a[key]), np.asarray(b[key])), key
    a, b = hostlib.background(inp), hostlib.background(inp)
    for key in a:
        assert np.array_equal(np.asarray(a[key]), np.asarray(b[key])), key
