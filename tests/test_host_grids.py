"""Host-side sampling grids against the grids dumped from the unmodified reference:
k_[md], k_size_cl_, tau_sampling_, l_, q_.  These are integer/closed-form constructions: required bit-exact."""
import numpy as np
import pytest

from classpp_public_amd import hostlib
from classpp_public_amd.inputs import Inputs


@pytest.mark.parametrize("cfg", ["small", "lcdm", "explanatory", "newt", "iso_cdi", "tens", "curved", "open", "curved_full", "tens_curved",
                                 "ncdm_small", "ncdm", "ncdm3", "ncdm3_tens", "ncdm_k3000"])
def test_grids_bit_exact(cfg):
    """scalars and tensors (pm.cpp:2007-2105: linear grid only, l_tensor_max), flat / open / closed space (k_min of pm.cpp:1677-1691,
    integer-nu q grid of tm.cpp:1003-1040), massive neutrinos (same builders, other cosmology)"""
    inp = Inputs(cfg)
    k, k_size_cl, k_size_cmb = hostlib.k_list(inp)
    assert np.array_equal(k, inp.d["pt.k"])
    assert k_size_cl == int(inp.d["pt.k_size_cl"][0]) and k_size_cmb == int(inp.d["pt.k_size_cmb"][0])
    tau = hostlib.tau_sampling(inp)
    assert np.array_equal(tau, inp.d["pt.tau_sampling"])
    l = hostlib.l_list(inp)
    assert np.array_equal(l, inp.d["tr.l"])
    q = hostlib.q_list(inp, k[0], k[k_size_cl - 1])
    assert np.array_equal(q, inp.d["tr.q"])


def test_grid_errors_are_reported():
    inp = Inputs("small")
    g = hostlib.grid_params(inp)
    g.k_step_transition = 0.0
    with pytest.raises(ValueError, match="division by zero"):
        hostlib.k_list(inp, g)
    g = hostlib.grid_params(inp)
    g.start_sources_at_tau_c_over_tau_h = 1e-9   # earlier than the thermodynamics table
    with pytest.raises(ValueError, match="inappropriate"):
        hostlib.tau_sampling(inp, g)


def test_ln_tau_tail_and_tau_of_z_match_reference():
    """z_max_pk = 3 (lcdm_zpk.ini): the tail of the sampling kept for P(k, z) (pm.cpp:1554-1592) has the reference's length and values,
    and tau(z) at the requested redshifts is the reference's background_tau_of_z"""
    from classpp_public_amd import hostlib
    inp = Inputs("lcdm_zpk")
    d = inp.d
    zmax = float(d["ppt.z_max_pk"][0])
    tau = hostlib.tau_sampling(inp)
    assert np.array_equal(tau, d["pt.tau_sampling"])
    n = hostlib.ln_tau_size(tau, hostlib.tau_of_z(inp, zmax))
    assert n == d["pt.ln_tau"].size
    assert np.array_equal(np.log(tau[tau.size - n:]), d["pt.ln_tau"])
    for z, want in zip(d["nl.z_pk"], d["nl.tau_of_z_pk"]):
        assert abs(hostlib.tau_of_z(inp, z) / want - 1) < 1e-12
    assert hostlib.ln_tau_size(tau, 0.) == 1
    with pytest.raises(Exception, match="smaller than or equal to the first possible value"):
        hostlib.ln_tau_size(tau, 0.5 * tau[0])
