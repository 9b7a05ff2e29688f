"""Non-cold species on the host (classpp_public_amd/host/cpt_ncdm.cpp, include/cpt_host.h cpt_host_ncdm): momentum samplings, d ln f0 /
d ln q, mass <-> density, against what the reference's NonColdDarkMatter object held when the fixtures were dumped
(tools/non_cold_dark_matter.cpp:202-790, tools/quadrature.c:69-360).  The nodes must be THE SAME rule the reference picked (5 nodes
for the perturbations at tol_ncdm_synchronous = 1e-3, 11 for the background at tol_ncdm_bg = 1e-5) to round-off; d ln f0 / d ln q is
analytic here and a five-point numerical derivative there (7e-8 apart)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from classpp_public_amd import hostlib  # noqa: E402
from classpp_public_amd.pipeline import ncdm_from_ini, read_ini  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


@pytest.mark.parametrize("cfg,tables", [("ncdm", "tables_ncdm1.npz"), ("ncdm3", "tables_ncdm3.npz")])
def test_species_from_the_ini_equal_the_reference(cfg, tables):
    ini = read_ini(os.path.join(GOLDEN, cfg + ".ini"))
    d = np.load(os.path.join(GOLDEN, cfg + ".npz"))
    t = np.load(os.path.join(GOLDEN, tables))
    out, Omega, m_eV = ncdm_from_ini(ini, float(d["pba.T_cmb"][0]), float(d["pba.h"][0]), int(d["ppt.gauge"][0]), dict(d))
    n = int(d["pba.N_ncdm"][0])
    assert len(Omega) == n
    for i in range(n):
        for key, tol in (("ncdm.q_%d", 1e-14), ("ncdm.w_%d", 1e-12), ("ncdm.q_bg_%d", 1e-14), ("ncdm.w_bg_%d", 1e-12), ("ncdm.dlnf0_dlnq_%d", 2e-7)):
            a, b = out[key % i], t[key % i]
            assert a.shape == b.shape, key % i                  # the same rule: 5 / 11 nodes
            assert np.max(np.abs(a / b - 1)) < tol, (key % i, np.max(np.abs(a / b - 1)))
    assert np.max(np.abs(out["ncdm.M"] / t["ncdm.M"] - 1)) < 1e-12 and np.max(np.abs(out["ncdm.factor"] / t["ncdm.factor"] - 1)) < 1e-14
    # the density budget closes on the reference's Omega_Lambda
    rest = float(d["pba.Omega0_g"][0] + d["pba.Omega0_b"][0] + d["pba.Omega0_ur"][0] + d["pba.Omega0_cdm"][0] + d["pba.Omega0_k"][0])
    assert abs((1. - rest - sum(Omega)) / float(d["pba.Omega0_lambda"][0]) - 1) < 1e-12
    if "m_ncdm" in ini:
        assert np.allclose(m_eV, [float(x) for x in ini["m_ncdm"].split(",")], rtol=0, atol=0)


def test_mass_density_round_trip_and_degeneracy():
    h, T = 0.7, 2.7255
    out1, Om, m = hostlib.ncdm_species(T, h, m_ncdm=[0.1, 0.3])
    out2, Om2, m2 = hostlib.ncdm_species(T, h, Omega_ncdm=Om)          # Omega -> M by Newton iteration on rho(M), tol_M_ncdm = 1e-7
    assert np.max(np.abs(np.array(m2) / np.array(m) - 1)) < 1e-7 and np.allclose(Om2, Om, rtol=1e-15)
    # heavier is denser, and in the non-relativistic limit Omega h^2 = m / 93.14 eV (T_ncdm = 0.71611 is tuned to that number)
    assert Om[1] > Om[0] and abs(Om[1] * h * h * 93.14 / 0.3 - 1) < 2e-3
    # mass and density both given: the degeneracy absorbs the ratio (ncdm.cpp:774-779)
    out3, Om3, _ = hostlib.ncdm_species(T, h, m_ncdm=[0.1], Omega_ncdm=[2 * Om[0]])
    assert abs(out3["ncdm.factor"][0] / out1["ncdm.factor"][0] - 2) < 1e-12 and abs(Om3[0] / Om[0] - 2) < 1e-15
    # a chemical potential adds density; a tighter tolerance adds nodes
    out4, Om4, _ = hostlib.ncdm_species(T, h, m_ncdm=[0.1], ksi_ncdm=[0.5])
    assert Om4[0] > Om[0]
    out5, _, _ = hostlib.ncdm_species(T, h, m_ncdm=[0.1], tol_ncdm=1e-5)
    assert out5["ncdm.q_0"].size > out1["ncdm.q_0"].size and out5["ncdm.q_0"].size == out1["ncdm.q_bg_0"].size


def test_refusals():
    with pytest.raises(ValueError, match="less than for a massless species"):
        hostlib.ncdm_species(2.7255, 0.7, Omega_ncdm=[1e-7])
    with pytest.raises(ValueError, match="positive m_ncdm or Omega_ncdm"):
        hostlib.ncdm_species(2.7255, 0.7, m_ncdm=[0.0])
    with pytest.raises(ValueError, match="at most 3"):
        hostlib.ncdm_species(2.7255, 0.7, m_ncdm=[0.1] * 4)
    with pytest.raises(ValueError, match="files"):
        ncdm_from_ini({"N_ncdm": "1", "m_ncdm": "0.1", "use_ncdm_psd_files": "1"}, 2.7255, 0.7, 1)
    with pytest.raises(ValueError, match="3 values for N_ncdm = 2"):
        ncdm_from_ini({"N_ncdm": "2", "m_ncdm": "0.1, 0.1, 0.1"}, 2.7255, 0.7, 1)
