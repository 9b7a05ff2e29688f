"""Device lensing (cpt_lensing_batch, classpp_public_amd/csrc/cpt_lensing.hip) against the oracle and the reference's own
lensed table; and the whole chain k-modes -> ... -> lensed C_l against the reference (north star: 1e-4)."""
import numpy as np
import pytest
import torch

import oracle_lib
from classpp_public_amd.inputs import Inputs

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def expl():
    from classpp_public_amd.backend import Backend
    inp = Inputs("explanatory")
    be = Backend(inp)
    yield inp, be
    be.close()


def _cmp(inp, got, want, tol_rel, tol_abs):
    sp = inp.spectra
    for name, idx in (("tt", sp.index_ct_tt), ("ee", sp.index_ct_ee), ("bb", sp.index_ct_bb), ("pp", sp.index_ct_pp)):
        err = np.max(np.abs(got[:, idx] / want[:, idx] - 1))
        assert err < tol_rel, (name, err)
    for name, idx in (("te", sp.index_ct_te), ("tp", sp.index_ct_tp), ("ep", sp.index_ct_ep)):
        err = np.max(np.abs(got[:, idx] - want[:, idx])) / np.max(np.abs(want[:, idx]))
        assert err < tol_abs, (name, err)


@pytest.mark.parametrize("accurate", [False, True])
def test_lensing_matches_oracle(expl, accurate):
    inp, be = expl
    d = inp.d
    lmax = int(d["le.l_unlensed_max"][0])
    cl = np.ascontiguousarray(d["sp.cl_table"])
    got = be.lensed_cl(torch.from_numpy(cl).to(be.device), lmax, 500, accurate=accurate, tol_gauss_legendre=1e-14).cpu().numpy()
    want = oracle_lib.lensing(inp, cl, lmax, 500, int(accurate), 70, 1e-14)
    assert got.shape == want.shape
    # accurate mode integrates the full correlation function (no difference trick): cancellation amplifies the
    # different summation order of the device reductions to a few 1e-9
    _cmp(inp, got, want, 1e-7 if accurate else 1e-9, 1e-8 if accurate else 1e-10)


def test_lensing_matches_reference_table(expl):
    inp, be = expl
    d = inp.d
    cl = np.ascontiguousarray(d["sp.cl_table"])
    got = be.lensed_cl(torch.from_numpy(cl).to(be.device), int(d["le.l_unlensed_max"][0]), int(d["le.delta_l_max"][0])).cpu().numpy()
    _cmp(inp, got, d["le.cl_lens"], 1e-9, 1e-10)


def test_full_chain_lensed_cl_matches_reference(expl):
    """tables -> k-modes -> sources -> transfer -> C_l -> lensed C_l, all on the GPU, vs the reference's lensed table."""
    inp, be = expl
    d = inp.d
    be.perturb_solve(want_sources=False)
    cl = be.cl(be.transfer(None))
    got = be.lensed_cl(cl, int(d["le.l_unlensed_max"][0]), int(d["le.delta_l_max"][0])).cpu().numpy()
    want = d["le.cl_lens"]
    sel = d["le.l"] <= int(d["le.l_lensed_max"][0])
    sp = inp.spectra
    worst = {}
    for name, idx in (("tt", sp.index_ct_tt), ("ee", sp.index_ct_ee), ("bb", sp.index_ct_bb), ("pp", sp.index_ct_pp)):
        worst[name] = np.max(np.abs(got[sel, idx] / want[sel, idx] - 1))
        assert worst[name] < 1e-4, (name, worst[name])
    idx = sp.index_ct_te
    worst["te"] = np.max(np.abs(got[sel, idx] - want[sel, idx])) / np.max(np.abs(want[sel, idx]))
    assert worst["te"] < 1e-4
    print("\n[explanatory, lensed] max errors vs reference: %s" % ", ".join("%s %.1e" % kv for kv in worst.items()))


def test_lensing_rejects_bad_input(expl):
    from classpp_public_amd.backend import CptInputError
    inp, be = expl
    cl = torch.zeros((inp.l.size, inp.spectra.ct_size), dtype=torch.float64, device=be.device)
    with pytest.raises(CptInputError):
        be.lensed_cl(cl, 100000)          # beyond the l grid
    with pytest.raises(CptInputError):
        be.lensed_cl(cl, 3000, 3000)      # delta_l_max >= l_max


def test_config5_closed_space_lensed_cl_matches_reference():
    """BASELINE configs[4]: Omega_k = -0.01 (closed) with lensing, default precision - the whole chain on the GPU"""
    from classpp_public_amd.backend import Backend
    inp = Inputs("curved_full")
    d = inp.d
    be = Backend(inp)
    be.perturb_solve(want_sources=False)
    cl = be.cl(be.transfer(None))
    got = be.lensed_cl(cl, int(d["le.l_unlensed_max"][0]), int(d["le.delta_l_max"][0])).cpu().numpy()
    want = d["le.cl_lens"]
    sel = d["le.l"] <= int(d["le.l_lensed_max"][0])
    sp = inp.spectra
    worst = {}
    for name, idx in (("tt", sp.index_ct_tt), ("ee", sp.index_ct_ee), ("bb", sp.index_ct_bb), ("pp", sp.index_ct_pp)):
        worst[name] = np.max(np.abs(got[sel, idx] / want[sel, idx] - 1))
        assert worst[name] < 1e-4, (name, worst[name])
    worst["te"] = np.max(np.abs(got[sel, sp.index_ct_te] - want[sel, sp.index_ct_te])) / np.max(np.abs(want[sel, sp.index_ct_te]))
    assert worst["te"] < 1e-4
    print("\n[curved_full, lensed] max errors vs reference: %s; kernels: perturb %.2f ms, los %.2f ms" % (
        ", ".join("%s %.1e" % kv for kv in worst.items()), be.kernel_ms(0)[0], be.kernel_ms(1)[0]))
    be.close()
