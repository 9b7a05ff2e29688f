"""GPU parity tests of hot path A (classpp_public_amd/csrc/cpt_perturb.hip) through the C ABI.

Checkers: the oracle (oracle/restate/perturb_oracle.cpp, pinned to the reference by tests/test_oracle_perturb.py)
for single device functions, and the golden sources_ dumped from the unmodified reference for the whole stage.
Tolerances for sources are those of tests/test_oracle_perturb.py (the reference's own rtol=1e-5 noise floor,
documented there); single-function checks are at round-off level.
"""
import numpy as np
import pytest
import torch

import bands
import oracle_lib
from classpp_public_amd.inputs import Inputs
from test_oracle_perturb import check_sources

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def small():
    from classpp_public_amd.backend import Backend
    inp = Inputs("small")
    return inp, Backend(inp)


def test_lookup_matches_oracle(small):
    inp, be = small
    t = inp.t["bg.tau_table"]
    # random order on purpose: forward walks, backward jumps, table nodes, both ends, z above the thermo table
    rng = np.random.default_rng(0)
    tau = np.concatenate([np.exp(rng.uniform(np.log(t[0] * 1.0001), np.log(t[-1]), 400)), t[[0, 1, 17, 2000, -2, -1]],
                          np.sort(np.exp(rng.uniform(np.log(1.0), np.log(t[-1]), 300)))])
    got = be.dbg_lookup(tau)
    want = oracle_lib.lookup(inp, tau)
    scale = np.maximum(np.abs(want), 1e-300)
    assert np.max(np.abs(got - want) / scale) < 1e-12


@pytest.mark.parametrize("flags", [(1, 0, 0), (0, 0, 0), (0, 0, 1), (0, 1, 1), (1, 0, 1)])
def test_derivs_match_oracle(small, flags):
    inp, be = small
    rng = np.random.default_rng(1)
    for k, tau in [(1e-4, 50.0), (0.03, 150.0), (0.03, 290.0), (0.2, 3000.0), (0.5, 13000.0)]:
        y = rng.normal(size=64)
        want = oracle_lib.derivs(inp, k, tau, *flags, y)
        got = be.dbg_derivs(k, tau, *flags, y[: want.size])
        assert got.size == want.size
        scale = np.max(np.abs(want))
        assert np.max(np.abs(got - want)) < 1e-11 * scale, (k, tau, np.max(np.abs(got - want)) / scale)


@pytest.mark.parametrize("flags", [(1, 0, 0), (0, 0, 0), (0, 0, 1), (0, 1, 1), (1, 0, 1)])
def test_structured_solve_matches_dense(small, flags):
    """(I - hg J) x = b through the kernel's structured factorisation (tridiagonal tails by continued fractions + dense
    register-resident core) against a dense numpy solve with the Jacobian assembled from the oracle's RHS."""
    inp, be = small
    rng = np.random.default_rng(2)
    for k, tau, hg in [(0.03, 150.0, 0.4), (0.03, 290.0, 2.5), (0.2, 3000.0, 1.0), (0.5, 900.0, 30.0), (1e-3, 5000.0, 500.0)]:
        n = oracle_lib.derivs(inp, k, tau, *flags, np.zeros(64)).size
        J = np.zeros((n, n))
        for j in range(n):
            e = np.zeros(64)
            e[j] = 1.0
            J[:, j] = oracle_lib.derivs(inp, k, tau, *flags, e)
        A = np.eye(n) - hg * J
        b = rng.normal(size=n)
        want = np.linalg.solve(A, b)
        got = be.dbg_solve(k, tau, *flags, hg, b)
        assert np.all(np.isfinite(got))
        # componentwise backward error (rows with hg*kappa' ~ 1e10 entries cancel to O(1)) and the solution itself
        assert np.all(np.abs(A @ got - b) <= 1e-11 * (np.abs(A) @ np.abs(got) + np.abs(b)))
        assert np.max(np.abs(got - want)) < 1e-8 * np.max(np.abs(want)), (k, tau, hg)


@pytest.mark.parametrize("flags", [(1, 0, 0), (0, 0, 0), (0, 0, 1), (0, 1, 1), (1, 0, 1)])
def test_product_form_of_the_core_solve(small, flags, monkeypatch):
    """The same systems with the core block solved as x = A_cc^-1 b (the inverse the helper wave forms from the integrator's factors,
    one column per lane) instead of by the two triangular sweeps: the form every Newton iteration but the first after a factorisation
    takes.  An explicit inverse is not backward stable; what the simplified Newton iteration needs is a forward error far below its
    convergence rate, and the two forms must agree to that."""
    inp, be = small
    rng = np.random.default_rng(2)
    ran = 0
    for k, tau, hg in [(0.03, 150.0, 0.4), (0.03, 290.0, 2.5), (0.2, 3000.0, 1.0), (0.5, 900.0, 30.0), (1e-3, 5000.0, 500.0)]:
        n = oracle_lib.derivs(inp, k, tau, *flags, np.zeros(64)).size
        b = rng.normal(size=n)
        monkeypatch.setenv("CPT_DBG_SOLVE_INVERSE", "0")
        lu = be.dbg_solve(k, tau, *flags, hg, b, full=True)
        monkeypatch.setenv("CPT_DBG_SOLVE_INVERSE", "1")
        got = be.dbg_solve(k, tau, *flags, hg, b, full=True)
        assert lu[63] == 0.0                        # (the kernel says which form ran: a factorisation that had to exchange rows keeps
        ran += int(got[63] == 1.0)                  #  the triangular sweeps - the tightly coupled regime at large hg kappa')
        lu, got = lu[:n], got[:n]
        assert np.all(np.isfinite(got))
        assert np.max(np.abs(got - lu)) < 1e-9 * np.max(np.abs(lu)), (k, tau, hg, np.max(np.abs(got - lu)) / np.max(np.abs(lu)))
    assert ran >= (1 if flags[0] else 4), ran


def test_perturb_small_all_modes(small):
    inp, be = small
    src, stats, status = be.perturb_solve()
    assert not status.any()
    got = src.cpu().numpy()
    check_sources(inp.config, got, inp.d["pt.sources"])
    osrc, ostats, _, _ = oracle_lib.perturb(inp)
    # same algorithm => nearly the same amount of work as the CPU restatement
    gs, os_ = sum(s.steps for s in stats), sum(s.steps for s in ostats)
    assert abs(gs - os_) < 0.02 * os_, (gs, os_)
    assert np.allclose([s.tau_ini for s in stats], [s.tau_ini for s in ostats], rtol=1e-9)
    assert [s.n_regimes for s in stats] == [s.n_regimes for s in ostats]
    ms, n = be.kernel_ms(0)
    print("\n[small] perturb kernel %.3f ms for %d modes, %d steps" % (ms, inp.nk, gs))


@pytest.mark.parametrize("cfg", ["small_tk", "newt_tk", "lcdm_tk", "ncdm_small_tk", "ncdm3_small_tk"])
def test_density_and_velocity_transfer_sources(cfg):
    """output = mTk, vTk (pm.cpp:1000-1050, 6930-6975, 7017-7200): delta_tot, delta_g, delta_b, delta_cdm, delta_ur, theta_tot, theta_g,
    theta_b, theta_cdm (Newtonian gauge), theta_ur, phi, psi as sources of their own, in the slots the reference gives them
    (cpt_config::index_tp_transfer), synchronous and Newtonian gauge, against the reference's sources_ table; bands of tests/bands.py
    (at most twice what the reference moves its own by at rtol / 2, tests/golden/noise_lcdm_tk.npz).  With massive neutrinos (one and three
    species; with a helper wave and without): delta_ncdm, theta_ncdm of every species as well, and the species in the totals."""
    from classpp_public_amd.backend import Backend
    import os
    if cfg.startswith("ncdm"):
        for helper in ("0", "1"):
            os.environ["CPT_SETS_HELPER"] = helper
            try:
                inp = Inputs(cfg)
                be = Backend(inp)
                src, stats, status = be.perturb_solve()
                assert not status.any()
                check_sources(inp.config, src.cpu().numpy()[:, :, inp.d["pt.sources_k_index"]], inp.d["pt.sources_subset"])
                be.close()
            finally:
                del os.environ["CPT_SETS_HELPER"]
        return
    inp = Inputs(cfg)
    assert inp.config.has_transfers and inp.config.tp_size >= 16
    be = Backend(inp)
    src, stats, status = be.perturb_solve()
    assert not status.any()
    got = src.cpu().numpy()
    assert np.all(np.isfinite(got))
    if "pt.sources_k_index" in inp.d:
        check_sources(inp.config, got[:, :, inp.d["pt.sources_k_index"]], inp.d["pt.sources_subset"])
    else:
        check_sources(inp.config, got, inp.d["pt.sources"])
    # phi + psi is the sum of the two potentials where all three are asked for
    c = inp.config
    if c.index_tp_phi_plus_psi >= 0:
        tk = list(c.index_tp_transfer)
        s = got[tk[10]] + got[tk[11]]
        assert np.max(np.abs(s - got[c.index_tp_phi_plus_psi])) < 1e-12 * np.max(np.abs(s))
    be.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", ["lcdm", "explanatory", "iso_cdi", "iso_nid", "newt", "ncdm", "ncdm3", "ncdm_k3000", "long_full", "ncdm_permille"])
def test_perturb_full_size(cfg):
    """BASELINE configs 1-2: every k-mode integrated on the GPU; the 16 golden columns are compared with the reference."""
    from classpp_public_amd.backend import Backend
    inp = Inputs(cfg)
    be = Backend(inp)
    src, stats, status = be.perturb_solve()
    assert not status.any()
    got = src.cpu().numpy()
    assert np.all(np.isfinite(got))
    ks = inp.d["pt.sources_k_index"]
    # (long_full, l_max = 50: delta_m / phi+psi within 5e-5 - the reference moves its own by 3.2e-5 when its rtol is halved,
    #  tests/golden/noise_long_full.npz; tests/bands.py allows twice that)
    check_sources(inp.config, got[:, :, ks], inp.d["pt.sources_subset"], **({"dm_tol": bands.LONG_DM_BAND} if cfg in ("long_full", "ncdm_permille") else {}))
    if inp.config.index_tp_delta_m >= 0:  # delta_m(k, tau0) for every k: the P(k) input
        dm, ref_dm = got[inp.config.index_tp_delta_m, -1, :], inp.d["pt.delta_m_today"]
        if inp.config.ic == 0:
            # (long_full: 5e-5, see above)
            err_dm = np.max(np.abs(dm / ref_dm - 1))
            print("\n[%s] delta_m today vs reference: max %.2e" % (cfg, err_dm))
            assert err_dm < (1e-5 if cfg not in ("long_full", "ncdm_permille") else bands.LONG_DM_BAND[0])
            assert np.median(np.abs(dm / ref_dm - 1)) < 1e-6
        else:  # isocurvature delta_m(k) changes sign: relative to the column maximum
            assert np.max(np.abs(dm - ref_dm)) < 1e-5 * np.max(np.abs(ref_dm))
    ms, n = be.kernel_ms(0)
    steps = np.array([s.steps for s in stats])
    print("\n[%s] perturb kernel %.2f ms, %d modes -> %.0f k-modes/s; steps total %d max %d; fevals %d, LU %d, solves %d" % (
        cfg, ms, inp.nk, inp.nk / ms * 1e3, steps.sum(), steps.max(), sum(s.fevals for s in stats),
        sum(s.factorisations for s in stats), sum(s.solves for s in stats)))
    # chained: transfer from the resident sources, compared with the reference's transfer_ golden subsets.
    # Tolerance 1e-4 of the per-(type,l) max: the sources carry the rtol-level noise discussed in test_oracle_perturb.
    tr = be.transfer(None).cpu().numpy()
    ls = inp.d["tr.transfer_l_index"]
    want = inp.d["tr.transfer_at_l"]
    scale = np.max(np.abs(want), axis=-1, keepdims=True)
    scale[scale == 0] = 1
    err = np.max(np.abs(tr[:, ls, :] - want) / scale)
    print("[%s] chained transfer: max err rel. to row max %.2e" % (cfg, err))
    assert err < bands.transfer_band(cfg), (err, bands.transfer_band(cfg))
    be.close()


# ---- non-flat space (BASELINE configs[4]): K != 0 enters through the s_l factors of the multipole ladders, k cotK(tau)
# in the truncation, the Einstein constraints, the tight-coupling slip and the super-horizon series (pm.cpp:2530-2533,
# 5856-5971, 7969-7979, 9488-9501, 4838-4941).  tests/golden/curved.ini: Omega_k = -0.01 (closed).
@pytest.fixture(scope="module", params=["curved", "open"])
def curved(request):
    """closed (Omega_k = -0.01) and open (Omega_k = +0.01) universes"""
    from classpp_public_amd.backend import Backend
    inp = Inputs(request.param)
    assert inp.config.K != 0 and inp.config.sgnK == (1 if request.param == "curved" else -1)
    be = Backend(inp)
    yield inp, be
    be.close()


@pytest.mark.parametrize("flags", [(1, 0, 0), (0, 0, 0), (0, 0, 1), (0, 1, 1), (1, 0, 1)])
def test_curved_derivs_match_oracle(curved, flags):
    inp, be = curved
    rng = np.random.default_rng(3)
    for k, tau in [(2e-4, 50.0), (0.03, 150.0), (0.03, 290.0), (0.2, 3000.0), (0.5, 9000.0)]:
        y = rng.normal(size=64)
        want = oracle_lib.derivs(inp, k, tau, *flags, y)
        got = be.dbg_derivs(k, tau, *flags, y[: want.size])
        scale = np.max(np.abs(want))
        assert np.max(np.abs(got - want)) < 1e-11 * scale, (k, tau, np.max(np.abs(got - want)) / scale)


def test_curved_perturb_matches_reference(curved):
    inp, be = curved
    src, stats, status = be.perturb_solve()
    assert not status.any()
    got = src.cpu().numpy()
    assert np.all(np.isfinite(got))
    check_sources(inp.config, got, inp.d["pt.sources"])
    dm = got[inp.config.index_tp_delta_m, -1, :]
    assert np.max(np.abs(dm / inp.d["pt.delta_m_today"] - 1)) < 1e-5


def test_closed_transfer_matches_reference_and_oracle(curved):
    """closed space: per-q hyperspherical tables (nu integer) + the flat-rescaling approximation above
    hyper_flat_approximation_nu (lowered to 1500 in curved.ini so that both branches are exercised), from the reference's
    own sources: transfer_ of the reference and of the oracle to round-off"""
    inp, be = curved
    src = inp.d["pt.sources"]
    got = be.transfer(torch.from_numpy(src).cuda()).cpu().numpy()
    ref = inp.d["tr.transfer"]
    orc, work = oracle_lib.transfer(inp, src)
    assert np.array_equal(got == 0, ref == 0)
    scale = np.max(np.abs(ref), axis=-1, keepdims=True)
    scale[scale == 0] = 1
    # the first node of every per-q table sits at chi = hyper_x_min = 1e-5 where the Hermite derivatives carry factors
    # 1/sin^2(chi) ~ 1e10: an ulp of difference in Phi (device sin/tan/sqrt vs libm) becomes ~1e-7 for the two lowest nu
    assert np.max(np.abs(got - orc) / scale) < 1e-6
    assert np.max(np.abs(got - ref) / scale) < 1e-6
    e = np.abs(got - ref) / scale
    first = 20 if inp.config.sgnK == 1 else 120   # (open space: nu is not integer and starts at 0.3: more low-nu tables)
    assert np.max(e[:, :, first:]) < 1e-9  # beyond the lowest nu: round-off
    ints, tsamp, fused = be.transfer_work()
    assert (ints, tsamp) == work


def test_closed_cl_end_to_end(curved):
    inp, be = curved
    be.perturb_solve(want_sources=False)
    cl = be.cl(be.transfer(None)).cpu().numpy()
    ref = inp.d["sp.cl_table"]
    sp = inp.spectra
    for name, idx in (("tt", sp.index_ct_tt), ("ee", sp.index_ct_ee), ("pp", sp.index_ct_pp)):
        err = np.max(np.abs(cl[:, idx] / ref[:, idx] - 1))
        assert err < 3e-4, (name, err)   # coarse precision file (same as `small`)
    for name, idx in (("te", sp.index_ct_te), ("tp", sp.index_ct_tp), ("ep", sp.index_ct_ep)):
        err = np.max(np.abs(cl[:, idx] - ref[:, idx])) / np.max(np.abs(ref[:, idx]))
        assert err < 3e-4, (name, err)
    # and the quadrature weights against the sequential algorithm on the reference's own table
    want = oracle_lib.cl_table(inp, inp.d["tr.transfer"])
    got = be.cl(torch.from_numpy(np.ascontiguousarray(inp.d["tr.transfer"])).cuda()).cpu().numpy()
    scale = np.max(np.abs(want), axis=0, keepdims=True)
    scale[scale == 0] = 1
    assert np.max(np.abs(got - want) / scale) < 1e-11


# ---- Newtonian gauge (pm.cpp:5869-5897, 8049-8074, 8228-8243, 9549-9592, 6849-6860, 5095-5198): phi dynamic in the eta lane,
# theta_cdm in its own core lane, metric_euler = k^2 psi in every Euler equation
@pytest.fixture(scope="module")
def newt():
    from classpp_public_amd.backend import Backend
    inp = Inputs("newt")
    assert inp.config.gauge == 0
    be = Backend(inp)
    yield inp, be
    be.close()


@pytest.mark.parametrize("flags", [(1, 0, 0), (0, 0, 0), (0, 0, 1), (0, 1, 1), (1, 0, 1)])
def test_newtonian_derivs_match_oracle(newt, flags):
    inp, be = newt
    rng = np.random.default_rng(4)
    for k, tau in [(1e-4, 50.0), (0.03, 150.0), (0.03, 290.0), (0.2, 3000.0), (0.5, 13000.0)]:
        y = rng.normal(size=64)
        want = oracle_lib.derivs(inp, k, tau, *flags, y)
        got = be.dbg_derivs(k, tau, *flags, y[: want.size])
        assert got.size == want.size
        scale = np.max(np.abs(want))
        assert np.max(np.abs(got - want)) < 1e-11 * scale, (k, tau, np.max(np.abs(got - want)) / scale)


@pytest.mark.parametrize("flags", [(1, 0, 0), (0, 0, 0), (0, 1, 1)])
def test_newtonian_structured_solve_matches_dense(newt, flags):
    inp, be = newt
    rng = np.random.default_rng(5)
    for k, tau, hg in [(0.03, 150.0, 0.4), (0.2, 3000.0, 1.0), (1e-3, 5000.0, 500.0)]:
        n = oracle_lib.derivs(inp, k, tau, *flags, np.zeros(64)).size
        J = np.zeros((n, n))
        for j in range(n):
            e = np.zeros(64)
            e[j] = 1.0
            J[:, j] = oracle_lib.derivs(inp, k, tau, *flags, e)
        A = np.eye(n) - hg * J
        b = rng.normal(size=n)
        want = np.linalg.solve(A, b)
        got = be.dbg_solve(k, tau, *flags, hg, b)
        assert np.all(np.abs(A @ got - b) <= 1e-11 * (np.abs(A) @ np.abs(got) + np.abs(b)))
        assert np.max(np.abs(got - want)) < 1e-8 * np.max(np.abs(want)), (k, tau, hg)


# ---- tensor modes (pm.cpp:3519-3586, 9045-9215, 7243-7280): tests/golden/tens.ini is a tensors-only reference run (modes = t)
@pytest.fixture(scope="module", params=["tens", "tens_curved"])
def tens(request):
    from classpp_public_amd.backend import Backend
    inp = Inputs(request.param)
    assert inp.config.mode == 1
    be = Backend(inp)
    yield inp, be
    be.close()


@pytest.mark.parametrize("flags", [(1, 0, 0), (0, 0, 0), (0, 1, 0)])
def test_tensor_derivs_match_oracle(tens, flags):
    inp, be = tens
    rng = np.random.default_rng(6)
    for k, tau in [(1e-4, 50.0), (0.03, 150.0), (0.03, 290.0), (0.2, 3000.0), (0.5, 13000.0)]:
        y = rng.normal(size=64)
        want = oracle_lib.derivs(inp, k, tau, *flags, y)
        got = be.dbg_derivs(k, tau, *flags, y[: want.size])
        assert got.size == want.size
        scale = np.max(np.abs(want))
        assert np.max(np.abs(got - want)) < 1e-11 * scale, (k, tau, np.max(np.abs(got - want)) / scale)


@pytest.mark.parametrize("flags", [(1, 0, 0), (0, 0, 0)])
def test_tensor_structured_solve_matches_dense(tens, flags):
    inp, be = tens
    rng = np.random.default_rng(7)
    for k, tau, hg in [(0.03, 150.0, 0.4), (0.2, 3000.0, 1.0), (1e-3, 5000.0, 500.0)]:
        n = oracle_lib.derivs(inp, k, tau, *flags, np.zeros(64)).size
        J = np.zeros((n, n))
        for j in range(n):
            e = np.zeros(64)
            e[j] = 1.0
            J[:, j] = oracle_lib.derivs(inp, k, tau, *flags, e)
        A = np.eye(n) - hg * J
        b = rng.normal(size=n)
        want = np.linalg.solve(A, b)
        got = be.dbg_solve(k, tau, *flags, hg, b)
        assert np.all(np.abs(A @ got - b) <= 1e-11 * (np.abs(A) @ np.abs(got) + np.abs(b)))
        assert np.max(np.abs(got - want)) < 1e-8 * np.max(np.abs(want)), (k, tau, hg)


def test_tensor_sources_match_reference(tens):
    inp, be = tens
    src, stats, status = be.perturb_solve()
    assert not status.any()
    got = src.cpu().numpy()
    ref = inp.d["pt.sources"]
    for tp, tol in ((inp.config.index_tp_t2, 2e-4), (inp.config.index_tp_p, 2e-4)):
        scale = np.max(np.abs(ref[tp]), axis=0, keepdims=True)
        scale[scale == 0] = 1
        assert np.max(np.abs(got[tp] - ref[tp]) / scale) < tol, tp
    osrc, ostats, _, _ = oracle_lib.perturb(inp)
    gs, os_ = sum(s.steps for s in stats), sum(s.steps for s in ostats)
    assert abs(gs - os_) < 0.02 * os_, (gs, os_)


# ---- massive neutrinos (BASELINE configs 3-4): 1 + NW wavefronts per k-mode, the momentum hierarchies Psi_l(q) in the chain waves,
# bordered Newton system (pm.cpp:8725-8879, 6317-6432, 5229-5256, 4479-4517).  ncdm_small / ncdm3_small hold the reference's full
# sources_ for one / three species of 0.06 eV at the coarse precision of `small`.
def test_first_order_mb_tight_coupling(small):
    """tight_coupling_approximation = first_order_MB (pm.cpp:9351-9361) against the reference (fixture tca_mb = small.ini with that scheme);
    the scheme matters at the level the test resolves: the default scheme's t0 sources miss this fixture by more than its tolerance"""
    from classpp_public_amd.backend import Backend
    inp = Inputs("tca_mb")
    assert inp.config.tight_coupling_approximation == 0
    be = Backend(inp)
    src, stats, status = be.perturb_solve()
    assert not status.any()
    check_sources(inp.config, src.cpu().numpy(), inp.d["pt.sources"])
    be.close()
    with pytest.raises(AssertionError):
        check_sources(inp.config, small[0].d["pt.sources"], inp.d["pt.sources"])


def test_hierarchies_longer_than_one_wavefront():
    """l_max_g = l_max_pol_g = l_max_ur = 50 (cl_permille-class): 14 + 3 x 48 = 158 equations per k-mode.  The three l >= 3 tails run on
    chain wavefronts of their own (cpt_perturb.hip "long tails"): sources against the reference (fixture long_small), work against the
    dense CPU restatement, and - the tails being longer versions of the same ladders - close to the default-hierarchy sources."""
    from classpp_public_amd.backend import Backend
    inp = Inputs("long_small")
    assert inp.config.l_max_g == 50 and inp.config.l_max_pol_g == 50 and inp.config.l_max_ur == 50
    be = Backend(inp)
    src, stats, status = be.perturb_solve()
    assert not status.any()
    got = src.cpu().numpy()
    assert np.all(np.isfinite(got))
    # (delta_m, phi + psi: 5e-5 with l_max = 50, twice the move of the reference against itself - tests/golden/noise_long_full.npz;
    #  the default hierarchies are held to 1e-5)
    check_sources(inp.config, got, inp.d["pt.sources"], dm_tol=bands.LONG_DM_BAND)
    ks = np.arange(0, inp.nk, 9)
    _, ostats, _, _ = oracle_lib.perturb(inp, k=inp.k[ks])
    gs, os_ = sum(stats[i].steps for i in ks), sum(s.steps for s in ostats)
    assert abs(gs - os_) < 0.02 * os_, (gs, os_)
    assert [stats[i].n_regimes for i in ks] == [s.n_regimes for s in ostats]
    assert np.allclose([stats[i].tau_ini for i in ks], [s.tau_ini for s in ostats], rtol=1e-9)
    ms, n = be.kernel_ms(0)
    print("\n[long_small] perturb kernel %.1f ms for %d modes, %d steps" % (ms, inp.nk, sum(s.steps for s in stats)))
    be.close()


def test_long_hierarchy_kernel_equals_the_one_wave_kernel_on_default_hierarchies(monkeypatch):
    """the same configuration (default l_max: 12 / 10 / 17) through the one-wavefront kernel and - forced by CPT_LONG_TAILS=1 - through
    the long-hierarchy kernel (tails on chain waves, bordered Newton system): two implementations of one system of equations"""
    from classpp_public_amd.backend import Backend
    inp = Inputs("small")
    be = Backend(inp)
    a, sa, status = be.perturb_solve()
    a = a.cpu().numpy()
    be.close()
    monkeypatch.setenv("CPT_LONG_TAILS", "1")
    be = Backend(inp)
    b, sb, status = be.perturb_solve()
    assert not status.any()
    b = b.cpu().numpy()
    be.close()
    check_sources(inp.config, b, inp.d["pt.sources"])
    check_sources(inp.config, b, a)
    dm = inp.config.index_tp_delta_m
    err = np.abs(b[dm, -1, :] / a[dm, -1, :] - 1)
    print("\n[long vs one-wave kernel, small] delta_m today: max %.2e median %.2e; steps %d vs %d" % (err.max(), np.median(err), sum(s.steps for s in sb), sum(s.steps for s in sa)))
    assert err.max() < 3e-5 and np.median(err) < 1e-6
    assert abs(sum(s.steps for s in sb) - sum(s.steps for s in sa)) < 0.02 * sum(s.steps for s in sa)


@pytest.mark.parametrize("cfg", ["ncdm_small", "ncdm3_small"])
def test_ncdm_sources_match_reference(cfg):
    from classpp_public_amd.backend import Backend
    inp = Inputs(cfg)
    be = Backend(inp)
    src, stats, status = be.perturb_solve()
    assert not status.any()
    got = src.cpu().numpy()
    assert np.all(np.isfinite(got))
    check_sources(inp.config, got, inp.d["pt.sources"])
    ks = np.arange(0, inp.nk, 7)
    _, ostats, _, _ = oracle_lib.perturb(inp, k=inp.k[ks])
    gs, os_ = sum(stats[i].steps for i in ks), sum(s.steps for s in ostats)
    assert abs(gs - os_) < 0.02 * os_, (gs, os_)          # same algorithm => same amount of work as the dense CPU restatement
    assert [stats[i].n_regimes for i in ks] == [s.n_regimes for s in ostats]   # (5 = tca, ufa, ncdmfa, rsa switches)
    assert max(s.n_regimes for s in stats) == 5
    ms, n = be.kernel_ms(0)
    print("\n[%s] perturb kernel %.1f ms for %d modes, %d steps" % (cfg, ms, inp.nk, sum(s.steps for s in stats)))
    be.close()


def test_tensor_modes_with_massive_neutrinos():
    """tensors with three ncdm species in the massless approximation (rho_relativistic = rho_ur + 3 sum p_ncdm, pm.cpp:6640-6657;
    start-time condition on the ncdm equation of state, pm.cpp:2574-2603): sources and chained transfer vs the reference"""
    from classpp_public_amd.backend import Backend
    inp = Inputs("ncdm3_tens")
    be = Backend(inp)
    src, stats, status = be.perturb_solve()
    assert not status.any()
    got = src.cpu().numpy()
    ks = inp.d["pt.sources_k_index"]
    ref = inp.d["pt.sources_subset"]
    for tp in (inp.config.index_tp_t2, inp.config.index_tp_p):
        scale = np.max(np.abs(ref[tp]), axis=0, keepdims=True)
        assert np.max(np.abs(got[tp][:, ks] - ref[tp]) / scale) < 2e-4
    tr = be.transfer(None).cpu().numpy()
    ls = inp.d["tr.transfer_l_index"]
    want = inp.d["tr.transfer_at_l"]
    scale = np.max(np.abs(want), axis=-1, keepdims=True)
    scale[scale == 0] = 1
    assert np.max(np.abs(tr[:, ls, :] - want) / scale) < 2e-4
    be.close()


def test_perturb_error_paths(small):
    """what the reference reports through class_test / ErrorMsg, through the C ABI: invalid arguments before any launch
    (CPT_ERR_INVALID -> CptInputError), failures inside the kernel per mode (CPT_ERR_RUNTIME -> CptError with the mode's k)"""
    from classpp_public_amd.backend import CptError, CptInputError
    inp, be = small
    with pytest.raises(CptInputError, match="division by zero"):
        be.perturb_solve(k=np.array([1e-3, 0.0]))                       # pm.cpp:2524
    with pytest.raises(CptInputError, match="strictly increasing"):
        be.perturb_solve(tau=np.array([100., 90., 200.]))
    with pytest.raises(CptInputError, match="exceeds the conformal age"):
        be.perturb_solve(tau=np.array([100., 2. * inp.config.tau0]))
    # a wavenumber so large that even the first line of the background table is too late to start it (pm.cpp:2562-2573)
    with pytest.raises(CptError, match="too late for this k"):
        be.perturb_solve(k=np.array([1e-3, 1e9]))
    # the backend stays usable after a failed call
    src, stats, status = be.perturb_solve(k=inp.k[::20])
    assert not status.any() and np.all(np.isfinite(src.cpu().numpy()))


def test_perturb_ragged_and_empty_inputs(small):
    """one k-mode, one sample time, an empty batch: the shapes at the edges of the batched entry point"""
    from classpp_public_amd.backend import CptInputError
    inp, be = small
    full, _, _ = be.perturb_solve()
    full = full.cpu().numpy()
    one, stats, status = be.perturb_solve(k=inp.k[7:8])
    assert one.shape == (inp.config.tp_size, inp.tau.size, 1) and not status.any()
    assert np.array_equal(one.cpu().numpy()[:, :, 0], full[:, :, 7])          # k-modes are independent units: bit for bit
    last, _, status = be.perturb_solve(k=inp.k[::9], tau=inp.tau[-2:])
    assert last.shape == (inp.config.tp_size, 2, inp.k[::9].size) and not status.any()
    ref = full[:, -2:, ::9]
    assert np.max(np.abs(last.cpu().numpy() - ref)) <= 1e-4 * np.max(np.abs(ref))   # (other start of the sampling, same integration)
    with pytest.raises(CptInputError, match="at least"):
        be.perturb_solve(k=np.zeros(0))
    with pytest.raises(CptInputError, match="at least"):
        be.perturb_solve(tau=inp.tau[-1:])
    src, _, status = be.perturb_solve(k=inp.k[:3])                               # still usable
    assert not status.any()


def test_perturb_is_linear_in_the_initial_amplitude_at_full_size():
    """size-independent property at BASELINE's full size: the system is linear, so doubling the primordial curvature doubles every
    source function - up to the step control, which is not scale-free (absolute floor 1e-15 of the error weights, ev.cpp:367-374:
    the high multipoles start below it), i.e. within the band two valid step sequences differ by (check_sources)"""
    from classpp_public_amd.backend import Backend
    inp = Inputs("lcdm")
    be = Backend(inp)
    s1, st1, status = be.perturb_solve()
    assert not status.any()
    s1 = s1.cpu().numpy()
    be.close()
    inp2 = Inputs("lcdm")
    inp2.config.curvature_ini = 2.0 * inp.config.curvature_ini
    be2 = Backend(inp2)
    s2, st2, status = be2.perturb_solve()
    assert not status.any()
    s2 = s2.cpu().numpy()
    be2.close()
    n1, n2 = sum(s.steps for s in st1), sum(s.steps for s in st2)
    assert abs(n1 - n2) < 0.02 * n1
    c = inp.config
    for tp, tol in ((c.index_tp_t0, 1e-2), (c.index_tp_t1, 1e-2), (c.index_tp_t2, 5e-4), (c.index_tp_p, 5e-4), (c.index_tp_delta_m, 3e-5)):
        scale = np.max(np.abs(s1[tp]), axis=0, keepdims=True)
        assert np.max(np.abs(0.5 * s2[tp] - s1[tp]) / scale) < tol, tp   # (t0: 5.5e-3 measured = two step sequences of the same tolerance)


def test_row_layout_and_packed_layout_agree(small, monkeypatch):
    """Two lane maps of the same equations: every l >= 3 tail in a 16-lane row of its own with log-depth (cyclic reduction) tail
    solves - the default whenever the tails fit - and the packed map with sequential sweeps (CPT_TAIL_ROWS=0, or hierarchies longer
    than 16 as in tests/golden/sc_prec.ini).  Same Newton matrix, another elimination order: same sources to the band of two
    valid step sequences, same amount of work."""
    inp, be = small
    rows, st_rows, status = be.perturb_solve()
    assert not status.any()
    monkeypatch.setenv("CPT_TAIL_ROWS", "0")
    packed, st_packed, status = be.perturb_solve()
    assert not status.any()
    monkeypatch.delenv("CPT_TAIL_ROWS")
    n1, n2 = sum(s.steps for s in st_rows), sum(s.steps for s in st_packed)
    assert abs(n1 - n2) < 0.02 * n1
    check_sources(inp.config, rows.cpu().numpy(), packed.cpu().numpy())
    check_sources(inp.config, rows.cpu().numpy(), inp.d["pt.sources"])
    check_sources(inp.config, packed.cpu().numpy(), inp.d["pt.sources"])


@pytest.mark.parametrize("cfg", ["ncdm_small", "ncdm3_small", "long_small", "ncdm_permille_small", "ncdm3", "long_full"])
def test_register_set_kernels_repeat_bit_for_bit(cfg):
    """More than 64 equations per k-mode (momentum bins of massive neutrinos, hierarchies longer than a wavefront): ONE wavefront owns the
    mode and runs ONE copy of the step control (cpt_perturb_sets.inc) - there is no second copy that could drift, no barrier a wave could
    miss.  (Round 2 spread such a mode over up to six waves, each with its own copy: results then changed from run to run once.)  Two
    launches on one handle and one on a fresh handle must agree bit for bit, statistics included - with helper waves and without
    (CPT_SETS_HELPER: two instantiations of the kernel, each deterministic; between them the compiler may fuse a multiply-add differently,
    so they agree as two valid step sequences do, not bit for bit)."""
    from classpp_public_amd.backend import Backend
    inp = Inputs(cfg)
    be = Backend(inp)
    a, sa, _ = be.perturb_solve()
    a = a.clone()
    b, sb, _ = be.perturb_solve()
    assert torch.equal(a, b) and [s.steps for s in sa] == [s.steps for s in sb]
    be.close()
    be2 = Backend(inp)
    c, sc, _ = be2.perturb_solve()
    assert torch.equal(a, c) and [s.fevals for s in sa] == [s.fevals for s in sc]
    be2.close()
    if cfg.endswith("_small"):
        import os
        for helper in ("0", "1"):
            os.environ["CPT_SETS_HELPER"] = helper
            try:
                be3 = Backend(inp)
                d, sd, _ = be3.perturb_solve()
                d = d.clone()
                d2, sd2, _ = be3.perturb_solve()
                assert torch.equal(d, d2) and [s.steps for s in sd] == [s.steps for s in sd2], "helper=%s" % helper
                check_sources(inp.config, d.cpu().numpy(), a.cpu().numpy())
                be3.close()
            finally:
                del os.environ["CPT_SETS_HELPER"]


def test_long_hierarchies_with_massive_neutrinos_all_modes():
    """l_max_g = 25, l_max_pol_g = 20, l_max_ur = 35, l_max_ncdm = 28 with one massive species (pm.cpp:3302-3481 at any l_max, :3412-3440,
    :8832-8884): 88 photon / ur lanes AND five momentum hierarchies of 29 multipoles - the three l >= 3 tails and three momentum-bin sets
    beside the core set of the one wavefront per k-mode; after the ncdm fluid switch the fluids join the core while the tails stay sets.
    Every mode and every source type against the reference's full table; step counts against the dense CPU restatement."""
    from classpp_public_amd.backend import Backend
    inp = Inputs("ncdm_permille_small")
    c = inp.config
    assert (c.l_max_g, c.l_max_pol_g, c.l_max_ur, c.l_max_ncdm, c.N_ncdm) == (25, 20, 35, 28, 1)
    be = Backend(inp)
    src, stats, status = be.perturb_solve()
    assert not status.any()
    got = src.cpu().numpy()
    assert np.all(np.isfinite(got))
    check_sources(c, got, inp.d["pt.sources"], dm_tol=bands.LONG_DM_BAND)
    ks = np.arange(0, inp.nk, 9)
    _, ostats, _, _ = oracle_lib.perturb(inp, k=inp.k[ks])
    gs, os_ = sum(stats[i].steps for i in ks), sum(s.steps for s in ostats)
    assert abs(gs - os_) < 0.02 * os_, (gs, os_)
    ms, n = be.kernel_ms(0)
    print("\n[ncdm_permille_small] perturb kernel %.1f ms for %d modes, %d steps" % (ms, inp.nk, sum(s.steps for s in stats)))
    be.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", ["lcdm", "tens", "curved", "newt"])
def test_two_wave_kernels_repeat_bit_for_bit(cfg):
    """The integrator wave and its helper talk through counters in LDS without a barrier (table rows by request number, the inverse of
    the core block of the Newton matrix): WHEN the helper's answers arrive depends on timing, WHICH answer the integrator uses and which
    form of the solve it takes does not (the first solve after a factorisation is the triangular one, every later one waits for the
    inverse of exactly that factorisation).  Three launches - two on one handle, one on a fresh handle - must agree bit for bit,
    statistics included."""
    from classpp_public_amd.backend import Backend
    inp = Inputs(cfg)
    be = Backend(inp)
    a, sa, _ = be.perturb_solve()
    a = a.clone()
    b, sb, _ = be.perturb_solve()
    assert torch.equal(a, b) and [(s.steps, s.fevals, s.solves) for s in sa] == [(s.steps, s.fevals, s.solves) for s in sb]
    be.close()
    be2 = Backend(inp)
    c, sc, _ = be2.perturb_solve()
    assert torch.equal(a, c) and [(s.steps, s.factorisations) for s in sa] == [(s.steps, s.factorisations) for s in sc]
    be2.close()



