"""From cosmological parameters to C_l and P(k) with nothing handed over from the reference (SURVEY S8f-1 closed):

    parameters --host--> background table (ndf15 in ln a) --host--> thermodynamics table (RECFAST + reionization)
               --host--> k, tau, l, q grids --GPU--> sources S(k,tau) --GPU--> Delta_l(q) --GPU--> C_l (+ lensing), P(k)

`ParameterInputs(name)` reads ONLY parameters and flags of a configuration (the values of the reference's input structs, plus YHe
and the reionization redshift / optical depth from the .ini) and computes every table and grid the hot path consumes with
classpp_public_amd/host/ (libcpt_host.so).  What stays outside: .ini parsing, the BBN helium table, shooting, non-cold species in
the background (their momentum quadrature), HyRec.
"""
import os

import numpy as np

from . import hostlib
from .inputs import GOLDEN, Inputs


def read_ini(path):
    out = {}
    for line in open(path):
        line = line.split("#")[0].strip()
        if "=" in line:
            k, v = line.split("=", 1)
            out[k.strip()] = v.strip()
    return out


def _list(ini, key, n, default=None):
    """comma-separated entry of an .ini / parameter dictionary -> n floats (input_module.cpp:1040-1075: one value per species)"""
    if key not in ini:
        return None if default is None else [default] * n
    v = ini[key]
    vals = [float(x) for x in str(v).replace("[", "").replace("]", "").split(",")] if isinstance(v, str) else [float(x) for x in np.atleast_1d(v)]
    if len(vals) != n:
        raise ValueError("%s: %d values for N_ncdm = %d species" % (key, len(vals), n))
    return vals


def ncdm_from_ini(ini, T_cmb, h, gauge, d=None):
    """The non-cold species of an .ini / parameter dictionary (N_ncdm, m_ncdm | Omega_ncdm | omega_ncdm, T_ncdm, ksi_ncdm, deg_ncdm;
    input_module.cpp:1014-1110) -> (ncdm.* table entries, Omega0 per species, mass in eV per species).  gauge: 1 synchronous, 0 newtonian
    (the momentum sampling of the perturbations is finer in the Newtonian gauge, input_module.cpp:1088-1092)."""
    n = int(float(ini.get("N_ncdm", 0)))
    if n < 1:
        raise ValueError("N_ncdm must be given (>= 1) to describe non-cold species")
    for key in ("ncdm_psd_filenames", "use_ncdm_psd_files", "Number of momentum bins", "Quadrature strategy", "Maximum q"):
        if key in ini and str(ini[key]).strip() not in ("", "0", "no"):
            raise ValueError("%s: distributions from files and manual momentum samplings are outside this package" % key)
    m = _list(ini, "m_ncdm", n)
    Om = _list(ini, "Omega_ncdm", n)
    om = _list(ini, "omega_ncdm", n)
    if Om is not None and om is not None:
        raise ValueError("In input, you can only enter one of Omega_ncdm or omega_ncdm, choose one")
    if om is not None:
        Om = [x / h / h for x in om]
    if m is None and Om is None:
        raise ValueError("non-cold species need m_ncdm or Omega_ncdm / omega_ncdm")

    def prec(name, default):
        if name in ini:
            return float(ini[name])
        return float(d["ppr." + name][0]) if d is not None and ("ppr." + name) in d else default
    tol = prec("tol_ncdm_synchronous", 1e-3) if gauge == 1 else prec("tol_ncdm_newtonian", 1e-5)
    return hostlib.ncdm_species(T_cmb, h, m_ncdm=m, Omega_ncdm=Om, T_ncdm=_list(ini, "T_ncdm", n, 0.71611), ksi_ncdm=_list(ini, "ksi_ncdm", n, 0.),
                                deg_ncdm=_list(ini, "deg_ncdm", n, 1.), tol_ncdm=tol, tol_ncdm_bg=prec("tol_ncdm_bg", 1e-5), tol_M_ncdm=prec("tol_M_ncdm", 1e-7))


def density_parameters(h, omega_b=None, omega_cdm=None, Omega_k=0.0, N_ur=3.046, T_cmb=2.7255, Omega_b=None, Omega_cdm=None, gauge_synchronous=True,
                       Omega_ncdm=0.0):
    """The budget equation of the reference's input module for LambdaCDM + massless neutrinos (source/input_module.cpp:593-603, 702,
    786, 1191 and the closure Omega_Lambda = 1 - Omega_k - sum): -> dict of the struct background entries the host modules read."""
    c, G, k_B, h_P, Mpc = 2.99792458e8, 6.67428e-11, 1.3806504e-23, 6.62606896e-34, 3.085677581282e22
    sigma_B = 2. * np.pi ** 5 * k_B ** 4 / 15. / h_P ** 3 / c ** 2
    H0 = h * 1.e5 / c
    Omega0_g = (4. * sigma_B / c * T_cmb ** 4) / (3. * c * c * 1.e10 * h * h / Mpc / Mpc / 8. / np.pi / G)
    Omega0_ur = N_ur * 7. / 8. * (4. / 11.) ** (4. / 3.) * Omega0_g
    # baryons / cdm: Omega or omega as given; when neither is, the default *Omega0* stays (0.022032 / 0.12038 over the default h squared,
    # :3191-3192, whatever h is), and a vanishing cdm density is raised to Omega0_cdm_min_synchronous in the synchronous gauge (:872)
    Omega0_b = Omega_b if Omega_b is not None else (omega_b / h / h if omega_b is not None else 0.022032 / (0.67556 * 0.67556))
    Omega0_cdm = Omega_cdm if Omega_cdm is not None else (omega_cdm / h / h if omega_cdm is not None else 0.12038 / (0.67556 * 0.67556))
    if gauge_synchronous and Omega0_cdm == 0.:
        Omega0_cdm = 1.e-10
    # same accumulation order as the reference's Omega_tot (:725-874, 1238; the non-cold species come last, :1110)
    Omega0_lambda = 1. - Omega_k - ((((Omega0_g + Omega0_b) + Omega0_ur) + Omega0_cdm) + Omega_ncdm)
    K = -Omega_k * H0 ** 2
    return {"H0": H0, "h": h, "T_cmb": T_cmb, "Omega0_g": Omega0_g, "Omega0_ur": Omega0_ur, "Omega0_b": Omega0_b, "Omega0_cdm": Omega0_cdm,
            "Omega0_lambda": Omega0_lambda, "Omega0_k": Omega_k, "K": K, "sgnK": 0 if K == 0 else (1 if K > 0 else -1)}


class ParameterInputs(Inputs):
    """Inputs whose spline tables and sampling grids are computed on the host from the cosmological parameters.

    `name` selects the precision / output settings of a committed configuration; `cosmology` (a dict for density_parameters),
    `YHe`, `z_reio` / `tau_reio` and `A_s`, `n_s` replace its cosmological parameters - any LambdaCDM + massless-neutrino
    cosmology runs, not just the fixtures' ones."""

    def __init__(self, name, golden_dir=GOLDEN, cosmology=None, YHe=None, z_reio=None, tau_reio=None, A_s=None, n_s=None, params=None, ini=None):
        """params / ini: the parameter entries and the (YHe, z_reio | tau_reio, l_max_tensors) strings given directly (classy.py)
        instead of the committed <name>.npz / <name>.ini"""
        self.name = name
        self.d = dict(params) if params is not None else dict(np.load(os.path.join(golden_dir, name + ".npz")))
        ini = dict(ini) if ini is not None else read_ini(os.path.join(golden_dir, name + ".ini"))
        if cosmology is not None:
            for k, v in density_parameters(**cosmology).items():
                self.d["pba." + k] = np.array([v], dtype=np.int32 if k == "sgnK" else np.float64)
            self.d["pba.has_curvature"] = np.array([int(self.d["pba.sgnK"][0] != 0)], dtype=np.int32)
        if YHe is not None:
            ini["YHe"] = repr(YHe)
        if z_reio is not None:
            ini.pop("tau_reio", None); ini["z_reio"] = repr(z_reio)
        if tau_reio is not None:
            ini.pop("z_reio", None); ini["tau_reio"] = repr(tau_reio)
        ncdm = None
        if int(self.d["pba.has_ncdm"][0]):
            # non-cold species from their physical parameters (hostlib.ncdm_species: momentum samplings, mass <-> density)
            ncdm, _, _ = ncdm_from_ini(ini, float(self.d["pba.T_cmb"][0]), float(self.d["pba.h"][0]), int(self.d["ppt.gauge"][0]), self.d)
            if len(ncdm["ncdm.M"]) != int(self.d["pba.N_ncdm"][0]):
                raise ValueError("N_ncdm = %d but %d species described" % (int(self.d["pba.N_ncdm"][0]), len(ncdm["ncdm.M"])))
        cp = hostlib.cosmo_params(self, ncdm=ncdm)            # struct background values (pba.* of the dump)
        tp = hostlib.CptThermoParams()
        hostlib.lib().cpt_host_thermo_defaults.argtypes = [hostlib.C.POINTER(hostlib.CptThermoParams)]
        hostlib.lib().cpt_host_thermo_defaults.restype = None
        hostlib.lib().cpt_host_thermo_defaults(hostlib.C.byref(tp))
        tp.YHe = float(ini["YHe"])
        tp.reio_parametrization = int(self.d["pth.reio_parametrization"][0])
        if "tau_reio" in ini:
            tp.reio_from_tau, tp.tau_reio = 1, float(ini["tau_reio"])
        else:
            tp.reio_from_tau, tp.z_reio = 0, float(ini["z_reio"])
        tables = {}
        tables.update(hostlib.background(self, cp))
        tables.update(hostlib.thermodynamics(self, cp, tp))
        if ncdm is not None:
            tables.update(ncdm)
        # (self.d, not the committed file again: it carries the replaced cosmological parameters)
        super().__init__(name, golden_dir, tables=tables, params=self.d)
        self.l_tensor_max = int(ini["l_max_tensors"]) if "l_max_tensors" in ini else None
        if A_s is not None:
            self.spectra.A_s = A_s
        if n_s is not None:
            self.spectra.n_s = n_s
        # the sampling grids, rebuilt from the tables just computed (never read from the fixture)
        self.k, self.k_size_cl, _ = hostlib.k_list(self)
        self.tau = hostlib.tau_sampling(self)
        if self.has_cls:
            self.l = hostlib.l_list(self)
            self.q = hostlib.q_list(self, self.k[0], self.k[self.k_size_cl - 1])
