"""classy-compatible Python surface over the MI355X hot path ("classy surface"; reference: classy.pyx:127-380, 454-560,
644-710, 744-825, 1093-1180).

    from classpp_public_amd.classy import Class
    cosmo = Class()
    cosmo.set({"output": "tCl,pCl,lCl,mPk", "lensing": "yes", "h": 0.67556, "omega_b": 0.022032, "omega_cdm": 0.12038, "YHe": 0.2453})
    cosmo.compute()
    cl, lcl, pk, s8 = cosmo.raw_cl(2500), cosmo.lensed_cl(2500), cosmo.pk(0.1, 0.), cosmo.sigma8()

What is behind it: `set()` collects the parameter dictionary like the reference's wrapper does (classy.pyx:244-249); `compute()`
turns it into the flat parameter / flag / precision entries the reference's input module would produce for it (input_module.cpp:549-3148,
for the subset listed in _KNOWN), builds background, thermodynamics and the sampling grids on the host (pipeline.ParameterInputs ->
libcpt_host.so) and runs perturbations -> transfer -> C_l / P(k) (-> lensing) on the GPU through the C ABI (backend.Backend ->
libcpt_hip.so).  `modes = s,t` runs one device handle per mode and sums the spectra, as the reference's spectra module does.

Outside this surface (raises CosmoSevereError, the reference's class for input errors, classy.pyx:71-80): non-cold species (their
momentum quadrature is an input of the path), isocurvature mixtures, z_pk > 0, non-linear corrections, number counts / lensing-potential
shear spectra, the BBN helium table (give YHe as a number), HyRec.  Errors of the compute path surface as CosmoComputationError
(classy.pyx:82-101).  There is no CPU fallback: compute() beyond the 'thermodynamics' level needs the HIP library and a GPU.
"""
import numpy as np

from . import hostlib
from .defaults import DEFAULT_PRECISION
from .pipeline import ParameterInputs, density_parameters, ncdm_from_ini


class CosmoError(Exception):
    def __init__(self, message=""):
        self.message = message
        super().__init__(message)


class CosmoSevereError(CosmoError):
    """wrong input (reference: std::invalid_argument -> CosmoSevereError, classy.pyx:88-101)"""


class CosmoComputationError(CosmoError):
    """a module failed for this parameter point (reference: std::runtime_error -> CosmoComputationError)"""


_VERBOSE = tuple(m + "_verbose" for m in ("input", "background", "thermodynamics", "perturbations", "transfer", "primordial", "spectra",
                                          "nonlinear", "lensing", "output"))
_KNOWN = {"h", "H0", "T_cmb", "omega_b", "Omega_b", "omega_cdm", "Omega_cdm", "Omega_k", "N_ur", "N_eff", "YHe", "recombination",
          "reio_parametrization", "z_reio", "tau_reio", "output", "lensing", "modes", "ic", "gauge", "P_k_ini type", "k_pivot", "A_s",
          "ln10^{10}A_s", "n_s", "alpha_s", "r", "n_t", "alpha_t", "l_max_scalars", "l_max_tensors", "P_k_max_h/Mpc", "P_k_max_1/Mpc", "z_pk",
          "z_max_pk", "non linear", "threads", "class_dir", "N_ncdm", "m_ncdm", "Omega_ncdm", "omega_ncdm", "T_ncdm", "ksi_ncdm", "deg_ncdm",
          "tensor method", "delta_l_max", "accurate_lensing", "num_mu_minus_lmax"} | set(_VERBOSE) | {
              p + "_" + ic for ic in ("bi", "cdi", "nid", "niv") for p in ("f", "n", "alpha")} | {
              p + "_" + a + "_" + b for a, b in (("ad", "bi"), ("ad", "cdi"), ("ad", "nid"), ("ad", "niv"), ("bi", "cdi"), ("bi", "nid"), ("bi", "niv"),
                                                 ("cdi", "nid"), ("cdi", "niv"), ("nid", "niv")) for p in ("c", "n", "alpha")}
_LEVELS = ("background", "thermodynamics", "perturb", "primordial", "nonlinear", "transfer", "spectra", "lensing")


def _arr(v, integer=False):
    return np.array([v], dtype=np.int32 if integer else np.float64)


def _yes(v):
    return str(v).strip().lower() in ("yes", "y", "true", "1")


_IC_NAMES = ("ad", "bi", "cdi", "nid", "niv")     # the reference's order (perturbations_module.cpp:1153-1170); codes = CPT_IC_*


def initial_conditions(pars):
    """'ic' of a classy dictionary -> the requested scalar initial conditions in the reference's order (input_module.cpp:1832-1870)"""
    want = [t.strip().lower() for t in str(pars.get("ic", "ad")).replace("&", ",").split(",") if t.strip()]
    full = {"ad": "ad", "bi": "bi", "cdi": "cdi", "nid": "nid", "niv": "niv"}
    if not want or any(t not in full for t in want):
        raise CosmoSevereError("ic: a list of ad, bi, cdi, nid, niv")
    return [n for n in _IC_NAMES if n in want]


def primordial_pairs(pars, ics):
    """(amplitude, tilt, running) of the primordial spectrum of every pair of initial conditions - diagonal: A_s f_ic^2, n_ic, alpha_ic;
    off-diagonal: c_12 sqrt(A_11 A_22), (n_11 + n_22) / 2 + n_12, (alpha_11 + alpha_22) / 2 + alpha_12 (input_module.cpp:1880-1960,
    primordial_module.cpp:716-890) -> {(i, j): (A, n, alpha)}, i <= j indices into ics; vanishing cross-correlations are left out"""
    def num(key, default):
        try:
            return float(pars[key]) if key in pars else default
        except (TypeError, ValueError):
            raise CosmoSevereError("could not read a number for '%s' (got %r)" % (key, pars[key]))
    A_s = num("A_s", 2.215e-9) if "ln10^{10}A_s" not in pars else np.exp(num("ln10^{10}A_s", 3.)) * 1e-10
    n_s, alpha_s = num("n_s", 0.9619), num("alpha_s", 0.)
    out = {}
    for i, a in enumerate(ics):
        if a == "ad":
            out[(i, i)] = (A_s, n_s, alpha_s)
        else:
            f = num("f_" + a, 1.)
            if f == 0.:
                raise CosmoSevereError("f_%s = 0: remove %s from ic instead" % (a, a))
            out[(i, i)] = (A_s * f * f, num("n_" + a, 1.), num("alpha_" + a, 0.))
    for i, a in enumerate(ics):
        for j in range(i + 1, len(ics)):
            b = ics[j]
            c = num("c_%s_%s" % (a, b), 0.)
            if abs(c) > 1.:
                raise CosmoSevereError("c_%s_%s must lie in [-1, 1]" % (a, b))
            if c != 0.:
                out[(i, j)] = (np.sqrt(out[(i, i)][0] * out[(j, j)][0]) * c, 0.5 * (out[(i, i)][1] + out[(j, j)][1]) + num("n_%s_%s" % (a, b), 0.),
                               0.5 * (out[(i, i)][2] + out[(j, j)][2]) + num("alpha_%s_%s" % (a, b), 0.))
    return out


def build_parameters(pars, mode, ic="ad"):
    """classy-style dictionary -> (entries, ini) for one mode ('s' | 't'): entries keyed like the reference's input structs
    (pba.*, pth.*, ppt.*, ppr.*, ppm.*, ptr.*, pt.index_tp_*, tr.index_tt_*, sp.index_ct_*; input_module.cpp:549-3148 for the defaults
    and the derived values, perturbations_module.cpp:250-420 / transfer_module.cpp:560-640 / spectra_module.cpp:470-560 for the index
    order), ini = the few strings pipeline.ParameterInputs reads besides them."""
    unknown = [k for k in pars if k not in _KNOWN and k not in DEFAULT_PRECISION]
    if unknown:
        raise CosmoSevereError("Class did not read input parameter(s): %s\n" % ", ".join(unknown))

    def num(key, default):
        try:
            return float(pars[key]) if key in pars else default
        except (TypeError, ValueError):
            raise CosmoSevereError("could not read a number for '%s' (got %r)" % (key, pars[key]))

    N_ncdm = int(num("N_ncdm", 0))
    if N_ncdm < 0 or N_ncdm > 3:
        raise CosmoSevereError("N_ncdm: 0 to 3 non-cold species")
    if N_ncdm == 0 and any(k in pars for k in ("m_ncdm", "Omega_ncdm", "omega_ncdm", "T_ncdm", "ksi_ncdm", "deg_ncdm")):
        raise CosmoSevereError("Class did not read input parameter(s): non-cold species parameters without N_ncdm\n")
    if str(pars.get("non linear", "")).strip().lower() not in ("", "none", "no"):
        raise CosmoSevereError("non-linear corrections are outside the accelerated path")
    if str(pars.get("recombination", "RECFAST")).strip().upper() != "RECFAST":
        raise CosmoSevereError("recombination = RECFAST is the code built here (HyRec is outside the path)")
    if str(pars.get("P_k_ini type", "analytic_Pk")).strip() != "analytic_Pk":
        raise CosmoSevereError("P_k_ini type = analytic_Pk only")
    ics_all = initial_conditions(pars)
    if ic not in ics_all and mode == "s":
        raise CosmoSevereError("initial condition %s is not among ic = %s" % (ic, ",".join(ics_all)))
    # z_max_pk: given, else the largest entry of the z_pk list (input_module.cpp:2715-2749)
    try:
        z_pk = [float(x) for x in str(pars.get("z_pk", "0")).replace("[", "").replace("]", "").split(",") if str(x).strip() != ""]
    except ValueError:
        raise CosmoSevereError("could not read a list of numbers for 'z_pk' (got %r)" % (pars.get("z_pk"),))
    z_max_pk = num("z_max_pk", max(z_pk + [0.]))
    if z_max_pk < 0. or min(z_pk + [0.]) < 0.:
        raise CosmoSevereError("asked for negative redshift z=%e" % min(z_pk + [z_max_pk]))
    for a, b in (("h", "H0"), ("omega_b", "Omega_b"), ("omega_cdm", "Omega_cdm"), ("z_reio", "tau_reio"), ("A_s", "ln10^{10}A_s"),
                 ("N_ur", "N_eff"), ("P_k_max_h/Mpc", "P_k_max_1/Mpc")):
        if a in pars and b in pars:
            raise CosmoSevereError("In input, you can only enter one of %s or %s, choose one" % (a, b))

    # ---- background budget (input_module.cpp:593-603, 702, 786, 1191)
    h = num("h", 0.67556) if "H0" not in pars else num("H0", 67.556) / 100.
    N_ur = num("N_ur", 3.046) if "N_eff" not in pars else num("N_eff", 3.046)
    gauge = str(pars.get("gauge", "synchronous")).strip().lower()
    if gauge not in ("synchronous", "newtonian"):
        raise CosmoSevereError("gauge: synchronous or newtonian")
    opt = lambda key: num(key, 0.) if key in pars else None
    # non-cold species (input_module.cpp:1014-1110): the momentum samplings and the mass <-> density relation, on the host
    ncdm_ini, Omega_ncdm_tot = {}, 0.
    if N_ncdm:
        if gauge != "synchronous":
            raise CosmoSevereError("non-cold species are integrated in the synchronous gauge only")
        ncdm_ini = {k: pars[k] for k in ("N_ncdm", "m_ncdm", "Omega_ncdm", "omega_ncdm", "T_ncdm", "ksi_ncdm", "deg_ncdm", "tol_ncdm_synchronous",
                                         "tol_ncdm_bg", "tol_M_ncdm") if k in pars}
        try:
            _, Om_species, _ = ncdm_from_ini(ncdm_ini, num("T_cmb", 2.7255), h, 1)
        except ValueError as e:
            raise CosmoSevereError(str(e))
        Omega_ncdm_tot = float(sum(Om_species))
    dens = density_parameters(h, opt("omega_b"), opt("omega_cdm"), num("Omega_k", 0.), N_ur, num("T_cmb", 2.7255),
                              Omega_b=opt("Omega_b"), Omega_cdm=opt("Omega_cdm"), gauge_synchronous=(gauge == "synchronous"),
                              Omega_ncdm=Omega_ncdm_tot)
    omega_b = dens["Omega0_b"] * h * h
    if h <= 0 or dens["Omega0_b"] <= 0 or dens["Omega0_cdm"] < 0 or N_ur < 0:
        raise CosmoSevereError("h and omega_b must be positive, omega_cdm and N_ur non-negative")
    d = {}
    for k, v in dens.items():
        d["pba." + k] = _arr(v, integer=(k == "sgnK"))
    d["pba.a_today"] = _arr(1.)
    d["pba.has_cdm"] = _arr(int(dens["Omega0_cdm"] != 0.), True)
    d["pba.has_ur"] = _arr(int(dens["Omega0_ur"] != 0.), True)
    d["pba.has_curvature"] = _arr(int(dens["sgnK"] != 0), True)
    for f, v in (("has_ncdm", int(N_ncdm > 0)), ("has_lambda", 1), ("has_fld", 0), ("N_ncdm", N_ncdm)):
        d["pba." + f] = _arr(v, True)

    # ---- thermodynamics switches
    reio = str(pars.get("reio_parametrization", "reio_camb")).strip()
    if reio not in ("reio_camb", "reio_none"):
        raise CosmoSevereError("reio_parametrization: reio_camb or reio_none")
    d["pth.reio_parametrization"] = _arr(1 if reio == "reio_camb" else 0, True)
    ini = {k: (v if isinstance(v, str) else ", ".join(repr(float(x)) for x in np.atleast_1d(v))) for k, v in ncdm_ini.items()}
    if "YHe" in pars and str(pars["YHe"]).strip().upper() != "BBN":
        ini["YHe"] = repr(num("YHe", 0.))
    elif abs(omega_b / 0.022032 - 1.) < 1e-12 and N_ur == 3.046:
        ini["YHe"] = "0.2452539925130077"   # what the reference's BBN interpolation returns at its default (omega_b, N_ur)
    else:
        raise CosmoSevereError("YHe = BBN needs the reference's BBN table, which is not part of this package: give YHe as a number")
    if "tau_reio" in pars:
        ini["tau_reio"] = repr(num("tau_reio", 0.))
    else:
        ini["z_reio"] = repr(num("z_reio", 11.357))

    # ---- outputs and modes (input_module.cpp:1700-1830, 2960-3000)
    out = [s.strip() for s in str(pars.get("output", "")).replace(" ", ",").split(",") if s.strip()]
    bad = [s for s in out if s.lower() not in ("tcl", "pcl", "lcl", "mpk", "mtk", "dtk", "vtk")]
    if bad:
        raise CosmoSevereError("output %s is outside the accelerated path (tCl, pCl, lCl, mPk, mTk / dTk, vTk)" % ", ".join(bad))
    low = [s.lower() for s in out]
    has_t, has_p, has_l, has_pk = "tcl" in low, "pcl" in low, "lcl" in low, "mpk" in low
    # density / velocity transfer functions (input_module.cpp: mTk = dTk -> has_density_transfers, vTk -> has_velocity_transfers)
    has_dtk, has_vtk = (("mtk" in low) or ("dtk" in low)) and mode != "t", ("vtk" in low) and mode != "t"
    lensing = _yes(pars.get("lensing", "no"))
    if lensing and not (has_l and (has_t or has_p)):
        raise CosmoSevereError("Lensed Cls only possible if you ask for lensing potential Cls and temperature or polarisation Cls (output must contain lCl and tCl or pCl)")
    tens = mode == "t"
    if tens and not (has_t or has_p):
        raise CosmoSevereError("tensor modes need tCl or pCl in output")
    ppr_delta_l_max = int(num("delta_l_max", 500))
    d["ppt.gauge"] = _arr(1 if gauge == "synchronous" else 0, True)
    d["ppt.has_scalars"] = _arr(int(not tens), True); d["ppt.has_tensors"] = _arr(int(tens), True)
    # (one device handle integrates one initial condition: the entries describe the run of `ic` alone)
    for f in _IC_NAMES:
        d["ppt.has_" + f] = _arr(int(f == (ic if not tens else "ad")), True)
    d["ppt.has_cl_cmb_temperature"] = _arr(int(has_t), True)
    d["ppt.has_cl_cmb_polarization"] = _arr(int(has_p), True)
    d["ppt.has_cl_cmb_lensing_potential"] = _arr(int(has_l and not tens), True)
    d["ppt.has_pk_matter"] = _arr(int(has_pk and not tens), True)
    all_modes = [m.strip() for m in str(pars.get("modes", "s")).split(",")]
    # (each l_max is read only when its mode is requested, input_module.cpp:2975-3000)
    l_max_scalars = int(num("l_max_scalars", 2500)) if "s" in all_modes else 2500
    l_max_tensors = int(num("l_max_tensors", 500)) if "t" in all_modes else 500
    d["ppt.l_scalar_max"] = _arr(l_max_scalars + (ppr_delta_l_max if lensing else 0), True)
    d["ppt.l_tensor_max"] = _arr(l_max_tensors, True)
    ini["l_max_tensors"] = str(l_max_tensors)
    if has_pk and "P_k_max_h/Mpc" in pars:
        kmax = num("P_k_max_h/Mpc", 1.) * h
    elif has_pk and "P_k_max_1/Mpc" in pars:
        kmax = num("P_k_max_1/Mpc", 1.)
    else:
        kmax = 1.
    d["ppt.k_max_for_pk"] = _arr(kmax); d["ppt.z_max_pk"] = _arr(z_max_pk)
    for f in ("switch_sw", "switch_eisw", "switch_lisw", "switch_dop", "switch_pol"):
        d["ppt." + f] = _arr(1, True)
    d["ppt.eisw_lisw_split_z"] = _arr(120.); d["ppt.three_ceff2_ur"] = _arr(1.); d["ppt.three_cvis2_ur"] = _arr(1.); d["ppt.G_eff_ur"] = _arr(0.)
    tmeth = str(pars.get("tensor method", "massless")).strip().lower()
    if tmeth not in ("massless", "photons"):
        raise CosmoSevereError("tensor method: massless or photons (exact needs non-cold species)")
    d["ppt.tensor_method"] = _arr(1 if tmeth == "massless" else 0, True)
    d["pt.mode_tensors"] = _arr(int(tens), True)
    d["pt.evolve_tensor_ur"] = _arr(int(tens and tmeth == "massless" and (dens["Omega0_ur"] != 0. or N_ncdm > 0)), True)   # pm.cpp:590-611
    d["ptr.lcmb_rescale"] = _arr(1.); d["ptr.lcmb_tilt"] = _arr(0.); d["ptr.lcmb_pivot"] = _arr(0.1)

    # ---- primordial (input_module.cpp:2380-2470; tensors: A_t = r A_s, 'scc' = the self-consistency conditions)
    A_s = num("A_s", 2.215e-9) if "ln10^{10}A_s" not in pars else np.exp(num("ln10^{10}A_s", 3.)) * 1e-10
    n_s, alpha_s, k_pivot = num("n_s", 0.9619), num("alpha_s", 0.), num("k_pivot", 0.05)
    if "s" not in all_modes:
        # a run without scalar modes does not read the scalar tilt and running (input_module.cpp:1972-2012): they keep their defaults,
        # also inside the self-consistency conditions for n_t and alpha_t below
        n_s, alpha_s = 0.9619, 0.
    d["ppm.A_s"] = _arr(A_s); d["ppm.n_s"] = _arr(n_s); d["ppm.alpha_s"] = _arr(alpha_s); d["ppm.k_pivot"] = _arr(k_pivot)
    if tens:
        r = num("r", 1.)
        if r <= 0:
            raise CosmoSevereError("r must be positive for tensor modes")
        scc = lambda key: key not in pars or str(pars[key]).strip().lower() == "scc"
        n_t = -r / 8. * (2. - r / 8. - n_s) if scc("n_t") else num("n_t", 0.)
        alpha_t = r / 8. * (r / 8. + n_s - 1.) if scc("alpha_t") else num("alpha_t", 0.)
        d["ppm.amplitude0"] = _arr(r * A_s); d["ppm.tilt0"] = _arr(n_t + 1.); d["ppm.running0"] = _arr(alpha_t)
    else:
        amp, tilt, run = primordial_pairs(pars, [ic])[(0, 0)]
        d["ppm.amplitude0"] = _arr(amp); d["ppm.tilt0"] = _arr(tilt); d["ppm.running0"] = _arr(run)

    # ---- index maps, in the order the modules define them
    tp, n = {}, 0
    has_cdm_, has_ur_, newt_ = dens["Omega0_cdm"] != 0., dens["Omega0_ur"] != 0., gauge != "synchronous"
    for name, on in (("t2", has_t or has_p), ("p", has_p), ("t0", has_t and not tens), ("t1", has_t and not tens),
                     ("delta_m", has_pk and not tens), ("delta_cb", has_pk and not tens and N_ncdm > 0),
                     # (pm.cpp:1107-1140: the transfer sources sit between delta_cb and phi+psi, psi after it)
                     ("delta_tot", has_dtk), ("delta_g", has_dtk), ("delta_b", has_dtk), ("delta_cdm", has_dtk and has_cdm_), ("delta_ur", has_dtk and has_ur_),
                     ("delta_ncdm1", has_dtk and N_ncdm > 0),
                     ("theta_tot", has_vtk), ("theta_g", has_vtk), ("theta_b", has_vtk), ("theta_cdm", has_vtk and has_cdm_ and newt_),
                     ("theta_ur", has_vtk and has_ur_), ("theta_ncdm1", has_vtk and N_ncdm > 0), ("phi", has_dtk),
                     ("phi_plus_psi", has_l and not tens), ("psi", has_dtk)):
        tp[name] = n if on else -1
        n += (N_ncdm if name in ("delta_ncdm1", "theta_ncdm1") else 1) * int(on)     # (one slot per species)
    for name, idx in tp.items():
        d["pt.index_tp_" + name] = _arr(idx, True)
    d["pt.tp_size"] = _arr(n, True)
    tt, n = {}, 0
    for name, on in (("t2", has_t), ("e", has_p), ("t0", has_t and not tens), ("t1", has_t and not tens), ("b", has_p and tens),
                     ("lcmb", has_l and not tens)):
        tt[name] = n if on else -1
        n += int(on)
    for name, idx in tt.items():
        d["tr.index_tt_" + name] = _arr(idx, True)
    d["tr.tt_size"] = _arr(n, True)
    ct, n = {}, 0
    for name, on in (("tt", has_t), ("ee", has_p), ("te", has_t and has_p), ("bb", has_p), ("pp", has_l and not tens),
                     ("tp", has_t and has_l and not tens), ("ep", has_p and has_l and not tens)):
        ct[name] = n if on else -1
        n += int(on)
    for name, idx in ct.items():
        d["sp.index_ct_" + name] = _arr(idx, True)
    d["sp.ct_size"] = _arr(n, True)
    d["sp.l_max_tot"] = _arr(int(d["ppt.l_tensor_max"][0]) if tens else int(d["ppt.l_scalar_max"][0]), True)

    # ---- precision (include/precisions.h defaults, overridable by name like in a .pre file)
    for name, v in DEFAULT_PRECISION.items():
        d["ppr." + name] = np.array([pars.get(name, v)], dtype=np.int32 if isinstance(v, int) else np.float64)
    d["le.delta_l_max"] = _arr(ppr_delta_l_max, True)
    d["le.accurate_lensing"] = _arr(int(num("accurate_lensing", 0)), True)
    d["le.num_mu_minus_lmax"] = _arr(int(num("num_mu_minus_lmax", 70)), True)
    d["le.has_lensed_cls"] = _arr(int(lensing), True)
    return d, ini


def spline_to_integer_l(l, table, lmax):
    """C_l on the multipole grid l[nl] (table [nl][ncol]) -> every integer l <= lmax, [ncol][lmax+1], zero below l[0]: the cubic
    spline in l with end derivatives estimated from the first / last three nodes that the reference's cl_output() applies
    (spectra_module.cpp:250-330, tools/arrays.c array_spline_table_lines with _SPLINE_EST_DERIV_).  Host post-processing."""
    x = np.asarray(l, dtype=np.float64)
    y = np.asarray(table, dtype=np.float64)
    n = x.size
    y2 = np.zeros_like(y); u = np.zeros_like(y)
    d10, d20, d21 = x[1] - x[0], x[2] - x[0], x[2] - x[1]
    dy0 = (d20 * d20 * (y[1] - y[0]) - d10 * d10 * (y[2] - y[0])) / (d20 * d10 * d21)
    y2[0] = -0.5
    u[0] = 3. / d10 * ((y[1] - y[0]) / d10 - dy0)
    for i in range(1, n - 1):
        sig = (x[i] - x[i - 1]) / (x[i + 1] - x[i - 1])
        p = sig * y2[i - 1] + 2.
        y2[i] = (sig - 1.) / p
        u[i] = (y[i + 1] - y[i]) / (x[i + 1] - x[i]) - (y[i] - y[i - 1]) / (x[i] - x[i - 1])
        u[i] = (6. * u[i] / (x[i + 1] - x[i - 1]) - sig * u[i - 1]) / p
    e1, e2, e12 = x[n - 2] - x[n - 1], x[n - 3] - x[n - 1], x[n - 3] - x[n - 2]
    dyn = (e2 * e2 * (y[n - 2] - y[n - 1]) - e1 * e1 * (y[n - 3] - y[n - 1])) / (e2 * e1 * e12)
    un = 3. / (x[n - 1] - x[n - 2]) * (dyn - (y[n - 1] - y[n - 2]) / (x[n - 1] - x[n - 2]))
    y2[n - 1] = (un - 0.5 * u[n - 2]) / (0.5 * y2[n - 2] + 1.)
    for k in range(n - 2, -1, -1):
        y2[k] = y2[k] * y2[k + 1] + u[k]
    L = np.arange(lmax + 1, dtype=np.float64)
    out = np.zeros((y.shape[1], lmax + 1))
    sel = (L >= x[0]) & (L <= x[-1])
    Ls = L[sel]
    hi = np.clip(np.searchsorted(x, Ls, side="left"), 1, n - 1)
    lo = hi - 1
    hh = x[hi] - x[lo]
    b = (Ls - x[lo]) / hh
    a = 1. - b
    out[:, sel] = (a[:, None] * y[lo] + b[:, None] * y[hi] + ((a ** 3 - a)[:, None] * y2[lo] + (b ** 3 - b)[:, None] * y2[hi]) * (hh * hh)[:, None] / 6.).T
    return out


class _ModeRun:
    """one device handle (one mode): perturbations -> transfer -> C_l table on the multipole grid, P(k) on the k grid"""

    def __init__(self, pars, mode, device, level, ic="ad", keep=False):
        """keep: several initial conditions are combined afterwards - hold on to the transfer table and delta_m(k, tau_0)"""
        from .backend import Backend, CptError, CptInputError
        self.mode, self.ic = mode, ic
        self.tr = self.dm = None
        d, ini = build_parameters(pars, mode, ic)
        try:
            self.inp = ParameterInputs("classy-" + mode, params=d, ini=ini)
        except ValueError as e:   # libcpt_host.so refused the point (e.g. tau_reio out of reach, unphysical densities)
            raise CosmoComputationError(str(e))
        self.be = self.cl = self.pk = None
        if level in ("background", "thermodynamics"):
            return
        try:
            self.be = Backend(self.inp, device=device)
            src = self.be.perturb_solve(want_sources=keep)[0]
            if self.inp.has_cls and level not in ("perturb", "primordial", "nonlinear"):
                tr = self.be.transfer(None)
                self.cl = self.be.cl(tr)
                if keep:
                    self.tr = tr
            if self.inp.config.index_tp_delta_m >= 0:
                self.pk = self.be.pk_linear().cpu().numpy()
                if keep:
                    self.dm = src[self.inp.config.index_tp_delta_m, -1, :].cpu().numpy().copy()
        except CptInputError as e:
            raise CosmoSevereError(str(e))
        except CptError as e:
            raise CosmoComputationError(str(e))

    def close(self):
        if self.be is not None:
            self.be.close()
            self.be = None


class Class:
    """The reference's `classy.Class` for the outputs of the accelerated path (classy.pyx:127-380)."""

    def __init__(self, input_parameters=None, device="cuda:0"):
        self._pars = dict(input_parameters or {})
        self._device = device
        self._runs = {}
        self._level = None
        self.parameters_changed = True
        self._cache = {}

    # -- legacy life-cycle calls (classy.pyx:236-273)
    def struct_cleanup(self):
        for r in self._runs.values():
            r.close()
        self._runs, self._level, self._cache = {}, None, {}
        self._ics, self._pairs = ["ad"], {}
        self.parameters_changed = True

    def empty(self):
        self._pars = {}
        self.struct_cleanup()
        return self

    def set(self, *args, **kwargs):
        new = dict(args[0]) if args else {}
        new.update(kwargs)
        if all(k in self._pars and self._pars[k] == v for k, v in new.items()):
            return self
        self._pars.update(new)
        self.struct_cleanup()
        return self

    @property
    def pars(self):
        return self._pars

    def _modes(self):
        modes = [m.strip() for m in str(self._pars.get("modes", "s")).split(",") if m.strip()]
        if not modes or any(m not in ("s", "t") for m in modes):
            raise CosmoSevereError("modes: s, t or s,t (vector modes are outside the accelerated path)")
        return sorted(set(modes))

    def compute(self, level=None):
        level = (level or ["lensing"])[0].lower()
        if level not in _LEVELS:
            raise CosmoSevereError("unknown level %r" % level)
        if self._runs and not self.parameters_changed and _LEVELS.index(self._level) >= _LEVELS.index(level):
            return self
        self.struct_cleanup()
        ics = initial_conditions(self._pars)
        self._ics, self._pairs = ics, primordial_pairs(self._pars, ics)
        for m in self._modes():
            if m == "s":
                # one device handle per initial condition (independent integrations); "s" = the first, "s:<ic>" the others
                for i, name in enumerate(ics):
                    self._runs["s" if i == 0 else "s:" + name] = _ModeRun(self._pars, "s", self._device, level, ic=name, keep=len(ics) > 1)
            else:
                self._runs[m] = _ModeRun(self._pars, m, self._device, level)
        self._level = level
        self.parameters_changed = False
        return self

    def _need(self, level):
        if not self._runs or self.parameters_changed or _LEVELS.index(self._level) < _LEVELS.index(level):
            self.compute([level])
        return next(iter(self._runs.values()))

    # -- C_l (classy.pyx:305-380)
    def _total_unlensed(self, lmax):
        """per-mode tables splined to every l and summed (spectra_module.cpp cl_output: the sum over modes) -> {name: [lmax+1]}"""
        tot = {}

        def add(r, table, factor):
            sp = r.inp.spectra
            full = spline_to_integer_l(r.inp.l, table.cpu().numpy(), min(lmax, int(r.inp.l[-1])))
            for name in ("tt", "ee", "te", "bb", "pp", "tp", "ep"):
                idx = getattr(sp, "index_ct_" + name)
                if idx >= 0:
                    acc = tot.setdefault(name, np.zeros(lmax + 1))
                    acc[: full.shape[1]] += factor * full[idx]
        for r in self._runs.values():
            add(r, r.cl, 1.)
        # correlated initial conditions: twice the cross spectra (spectra_module.cpp cl_output: sum over the symmetric ic x ic matrix)
        for (i, j), (amp, tilt, run) in self._pairs.items():
            if i != j:
                ri, rj = self._scalar_run(i), self._scalar_run(j)
                add(ri, ri.be.cl_cross(ri.tr, rj.tr, amp, tilt, run), 2.)
        return tot

    def _scalar_run(self, i):
        return self._runs["s" if i == 0 else "s:" + self._ics[i]]

    def _l_max_tot(self):
        return max(int(r.inp.d["sp.l_max_tot"][0]) for r in self._runs.values())

    def raw_cl(self, lmax=-1):
        r0 = self._need("spectra")
        if not r0.inp.has_cls or r0.cl is None:
            raise CosmoSevereError("No Cls computed")
        top = self._l_max_tot()
        lmax = top if lmax == -1 else lmax
        if lmax > top:
            raise CosmoSevereError("Can only compute up to lmax=%d" % top)
        out = self._total_unlensed(lmax)
        out["ell"] = np.arange(lmax + 1)
        return out

    def lensed_cl(self, lmax=-1):
        import torch
        self._need("lensing")
        if "s" not in self._runs or self._runs["s"].cl is None:
            raise CosmoSevereError("No Cls computed")
        r = self._runs["s"]
        d, sp = r.inp.d, r.inp.spectra
        if not int(d["le.has_lensed_cls"][0]):
            raise CosmoSevereError("Lensing Cls not computed, add 'lensing':'yes' to your input.")
        l_unlensed_max = int(d["ppt.l_scalar_max"][0])
        delta = int(d["le.delta_l_max"][0])
        top = l_unlensed_max - delta
        lmax = top if lmax == -1 else lmax
        if lmax > top:
            raise CosmoSevereError("Can only compute up to lmax=%d" % top)
        if "lensed" not in self._cache:
            table = r.cl
            if len(self._runs) > 1:   # s,t and / or several initial conditions: the total unlensed spectra on the scalar multipole grid
                tot = self._total_unlensed(int(r.inp.l[-1]))
                table = r.cl.clone()
                for name in ("tt", "ee", "te", "bb", "pp", "tp", "ep"):
                    idx = getattr(sp, "index_ct_" + name)
                    if idx >= 0:
                        table[:, idx] = torch.as_tensor(tot[name][r.inp.l], device=table.device)
            from .backend import CptError
            try:
                got = r.be.lensed_cl(table, l_unlensed_max, delta, accurate=bool(int(d["le.accurate_lensing"][0])),
                                     num_mu_minus_lmax=int(d["le.num_mu_minus_lmax"][0])).cpu().numpy()
            except CptError as e:
                raise CosmoComputationError(str(e))
            self._cache["lensed"] = (r.inp.l[: got.shape[0]].copy(), got)
        le_l, got = self._cache["lensed"]
        full = spline_to_integer_l(le_l, got, lmax)
        out = {name: full[getattr(sp, "index_ct_" + name)] for name in ("tt", "ee", "te", "bb", "pp", "tp", "ep")
               if getattr(sp, "index_ct_" + name) >= 0}
        out["ell"] = np.arange(lmax + 1)
        return out

    def lensed_cl_computed(self):
        return bool(self._runs) and "s" in self._runs and bool(int(self._runs["s"].inp.d["le.has_lensed_cls"][0])) and self._level == "lensing"

    # -- P(k), sigma (classy.pyx:454-560, 644-710, 805-815)
    def _pk_run(self):
        self._need("nonlinear")
        r = self._runs.get("s")
        if r is None or r.pk is None:
            raise CosmoSevereError("Power spectrum not computed. You must add mPk to the list of outputs.")
        return r

    def _pk_total(self):
        """P(k) on the k grid, summed over the initial conditions: sum_i P_ii + 2 sum_{i<j} P_ij with
        P_ij = 2 pi^2 / k^3 calP_ij(k) delta_i(k) delta_j(k) (nonlinear_module.cpp:1886-2040; one initial condition: the device's P(k))"""
        r = self._pk_run()
        if len(self._ics) == 1:
            return r.pk
        if "pk_total" not in self._cache:
            k = r.inp.k
            tot = np.zeros_like(r.pk)
            kp = float(r.inp.d["ppm.k_pivot"][0])
            for (i, j), (amp, tilt, run) in self._pairs.items():
                ri, rj = self._scalar_run(i), self._scalar_run(j)
                if i == j:
                    tot += ri.pk
                else:
                    lk = np.log(k / kp)
                    tot += 2. * (2. * np.pi ** 2 / k ** 3) * amp * np.exp((tilt - 1.) * lk + 0.5 * run * lk * lk) * ri.dm * rj.dm
            self._cache["pk_total"] = tot
        return self._cache["pk_total"]

    def _sigma_total(self, R):
        r = self._pk_run()
        if len(self._ics) == 1:
            return r.be.sigma(float(R))
        import ctypes as C
        from . import capi
        k, pk = np.ascontiguousarray(r.inp.k), np.ascontiguousarray(self._pk_total())
        out = C.c_double()
        pd = C.POINTER(C.c_double)
        if capi.lib().cpt_sigma_of_pk(k.ctypes.data_as(pd), pk.ctypes.data_as(pd), k.size, float(R), 80., C.byref(out)) != 0:
            raise CosmoComputationError("sigma(R): the total P(k) is not positive everywhere")
        return out.value

    def pk_lin(self, k, z=0.):
        """linear total-matter P(k, z) [Mpc^3] at k [1/Mpc], 0 <= z <= z_max_pk: natural cubic spline of ln P in ln k over the k grid
        (nonlinear_module.cpp:2041-2212 nonlinear_pk_at_k_and_z; z > 0: the spline in ln tau of nonlinear_pk_at_z first)"""
        r = self._pk_run()
        kk = r.inp.k
        if not (kk[0] <= k <= kk[-1]):
            raise CosmoSevereError("k=%e out of bounds [%e:%e]" % (k, kk[0], kk[-1]))
        key = ("lnpk", float(z))
        if key not in self._cache:
            from scipy.interpolate import CubicSpline
            self._cache[key] = CubicSpline(np.log(kk), np.log(self._pk_total() if z == 0. else self._pk_at_z(z)), bc_type="natural")
        return float(np.exp(self._cache[key](np.log(k))))

    def _late_times(self, z):
        """(tau(z), ln_tau_size) for 0 < z <= z_max_pk: the redshift's conformal time and the length of the tail of the sampling the sources
        were kept on (pm.cpp:1554-1592; nonlinear_module.cpp:129-146 for the two errors)"""
        from . import hostlib
        r = self._pk_run()
        if z < 0.:
            raise CosmoSevereError("asked for negative redshift z=%e" % z)
        zmax = float(r.inp.d["ppt.z_max_pk"][0])
        if zmax == 0.:
            raise CosmoComputationError("You are asking for the matter power spectrum at z=%e but the code was asked to store it only at z=0. You probably "
                                        "forgot to pass the input parameter z_max_pk (see explanatory.ini)" % z)
        if "ln_tau_size" not in self._cache:
            self._cache["ln_tau_size"] = hostlib.ln_tau_size(r.inp.tau, hostlib.tau_of_z(r.inp, zmax))
        return hostlib.tau_of_z(r.inp, z), self._cache["ln_tau_size"]

    def _pk_at_z(self, z):
        """linear P(k, z) on the k grid, 0 < z <= z_max_pk: ln P(k, tau_i) splined in ln tau over the tail of the sampling, on the device
        (Backend.pk_at_tau / cpt_pk_at_tau; NonlinearModule::nonlinear_pk_at_z)"""
        if len(self._ics) != 1:
            raise CosmoSevereError("P(k, z > 0) with several correlated initial conditions is outside this package")
        r = self._pk_run()
        key = ("pk_z", float(z))
        if key not in self._cache:
            tau_z, n = self._late_times(z)
            try:
                self._cache[key] = r.be.pk_at_tau(tau_z, n).cpu().numpy()
            except Exception as e:
                raise CosmoComputationError(str(e))
        return self._cache[key]

    pk = pk_lin

    # -- baryons + cold dark matter alone (classy.pyx:493-560, 675-708, 811-816): only with non-cold species, otherwise P_cb = P_m is not stored
    def _cb_run(self):
        r = self._pk_run()
        if int(r.inp.config.index_tp_delta_cb) < 0:
            raise CosmoSevereError("P_cb not computed (probably because there are no massive neutrinos) so you cannot ask for it")
        if len(self._ics) != 1:
            raise CosmoSevereError("P_cb with several correlated initial conditions is outside this package")
        return r

    def pk_cb_lin(self, k, z=0.):
        r = self._cb_run()
        kk = r.inp.k
        if not (kk[0] <= k <= kk[-1]):
            raise CosmoSevereError("k=%e out of bounds [%e:%e]" % (k, kk[0], kk[-1]))
        key = ("lnpk_cb", float(z))
        if key not in self._cache:
            from scipy.interpolate import CubicSpline
            try:
                if z == 0.:
                    pk = r.be.pk_linear(cb=True).cpu().numpy()
                else:
                    tau_z, n = self._late_times(z)
                    pk = r.be.pk_at_tau(tau_z, n, cb=True).cpu().numpy()
            except (CosmoSevereError, CosmoComputationError):
                raise
            except Exception as e:
                raise CosmoComputationError(str(e))
            self._cache[key] = CubicSpline(np.log(kk), np.log(pk), bc_type="natural")
        return float(np.exp(self._cache[key](np.log(k))))

    pk_cb = pk_cb_lin

    def sigma_cb(self, R, z=0.):
        r = self._cb_run()
        if float(r.inp.d["ppt.k_max_for_pk"][0]) < self.h():
            raise CosmoSevereError("In order to get sigma(R,z) you must set 'P_k_max_h/Mpc' to 1 or bigger, in order to have k_max > 1 h/Mpc.")
        try:
            if z == 0.:
                return r.be.sigma(float(R), cb=True)
            tau_z, n = self._late_times(z)
            return r.be.sigma_at_tau(float(R), tau_z, n, cb=True)
        except (CosmoSevereError, CosmoComputationError):
            raise
        except Exception as e:
            raise CosmoComputationError(str(e))

    def sigma8_cb(self):
        return self.sigma_cb(8. / self.h())

    def _pk_grid(self, fn, k, z, k_size, z_size, mu_size):
        """P on a [k_size][z_size][mu_size] array of wavenumbers (classy.pyx:562-640: the fast loops of the likelihoods)"""
        k = np.asarray(k, dtype=np.float64)
        out = np.zeros((k_size, z_size, mu_size))
        for ik in range(k_size):
            for iz in range(z_size):
                for im in range(mu_size):
                    out[ik, iz, im] = fn(float(k[ik, iz, im]), float(z[iz]))
        return out

    def get_pk(self, k, z, k_size, z_size, mu_size): return self._pk_grid(self.pk, k, z, k_size, z_size, mu_size)
    def get_pk_lin(self, k, z, k_size, z_size, mu_size): return self._pk_grid(self.pk_lin, k, z, k_size, z_size, mu_size)
    def get_pk_cb(self, k, z, k_size, z_size, mu_size): return self._pk_grid(self.pk_cb, k, z, k_size, z_size, mu_size)
    def get_pk_cb_lin(self, k, z, k_size, z_size, mu_size): return self._pk_grid(self.pk_cb_lin, k, z, k_size, z_size, mu_size)

    def get_pk_cb_array(self, k, z, k_size, z_size, nonlinear):
        if nonlinear:
            raise CosmoSevereError("non-linear corrections are outside the accelerated path")
        return np.array([[self.pk_cb_lin(float(kk), float(zz)) for zz in np.asarray(z)[:z_size]] for kk in np.asarray(k)[:k_size]]).ravel()

    def get_pk_array(self, k, z, k_size, z_size, nonlinear):
        if nonlinear:
            raise CosmoSevereError("non-linear corrections are outside the accelerated path")
        return np.array([[self.pk_lin(float(kk), float(zz)) for zz in np.asarray(z)[:z_size]] for kk in np.asarray(k)[:k_size]]).ravel()

    def get_pk_and_k(self):
        """the k grid [1/Mpc] and P(k) [Mpc^3] on it, as computed on the device (no interpolation)"""
        r = self._pk_run()
        return self._pk_total().copy(), r.inp.k.copy()

    def sigma(self, R, z=0.):
        r = self._pk_run()
        if float(r.inp.d["ppt.k_max_for_pk"][0]) < self.h():
            raise CosmoSevereError("In order to get sigma(R,z) you must set 'P_k_max_h/Mpc' to 1 or bigger, in order to have k_max > 1 h/Mpc.")
        if z == 0.:
            return self._sigma_total(R)
        if len(self._ics) != 1:
            raise CosmoSevereError("sigma(R, z > 0) with several correlated initial conditions is outside this package")
        tau_z, n = self._late_times(z)   # (classy.pyx:644-676 -> nonlinear_sigmas_at_z, nonlinear_module.cpp:927-963)
        try:
            return r.be.sigma_at_tau(float(R), tau_z, n)
        except Exception as e:
            raise CosmoComputationError(str(e))

    def sigma8(self):
        """the nonlinear module's sigma8_ (no k_max check here, classy.pyx:805-809)"""
        return self._sigma_total(8. / self.h())

    # -- scalars (classy.pyx:744-825, 1079-1092, 1771-1776)
    def _t(self, key, level="thermodynamics"):
        return float(np.asarray(self._need(level).inp.t[key]).reshape(-1)[0])

    def _b(self, key):
        return float(np.asarray(self._need("background").inp.d["pba." + key]).reshape(-1)[0])

    def h(self): return self._b("h")
    def T_cmb(self): return self._b("T_cmb")
    def Omega_g(self): return self._b("Omega0_g")
    def Omega_b(self): return self._b("Omega0_b")
    def omega_b(self): return self._b("Omega0_b") * self.h() ** 2
    def Omega_Lambda(self): return self._b("Omega0_lambda")
    def Omega0_k(self): return self._b("Omega0_k")
    def Omega0_cdm(self): return self._b("Omega0_cdm")
    def Omega_m(self): return self._t("bg.Omega0_m", "background")
    def Omega0_m(self): return self._t("bg.Omega0_m", "background")
    def Omega_r(self): return self._t("bg.Omega0_r", "background")
    def Neff(self): return self._t("bg.Neff", "background")
    def age(self): return self._t("bg.age", "background")
    def conformal_age(self): return self._t("bg.conformal_age", "background")
    def n_s(self): return float(self._need("background").inp.d["ppm.n_s"][0])
    def A_s(self): return float(self._need("background").inp.d["ppm.A_s"][0])
    def tau_reio(self): return self._t("th.tau_reionization")
    def z_reio(self): return self._t("th.z_reionization")
    def z_rec(self): return self._t("th.z_rec")
    def theta_s_100(self): return 100. * self._t("th.rs_rec") / self._t("th.ra_rec")

    def rs_drag(self):
        """comoving sound horizon at baryon drag (thermodynamics_module.cpp:1170-1184): the background's r_s at z_d"""
        return self._bg_value_at_z(self._t("th.z_d"), "rs")

    def theta_star_100(self):
        """100 r_s(z_*) / r_a(z_*) (thermodynamics_module.cpp:1138-1154)"""
        zs = self._t("th.z_star")
        return 100. * self._bg_value_at_z(zs, "rs") / (self._bg_value_at_z(zs, "ang_distance") * (1. + zs))

    def k_eq(self):
        """a H at radiation / matter equality (background_module.cpp:1690-1742: bisection in tau on Omega_m / Omega_r = 1)"""
        t = self._need("background").inp.t
        tau, tab, d2 = np.asarray(t["bg.tau_table"]), np.asarray(t["bg.background_table"]), np.asarray(t["bg.d2background_dtau2_table"])
        col = lambda n: int(np.asarray(t["bg.index_bg_" + n]).reshape(-1)[0])
        im, ir, ia, iH = col("Omega_m"), col("Omega_r"), col("a"), col("H")
        lo, hi = 0, tau.size - 1
        while hi - lo > 1:
            mid = (lo + hi) // 2
            if tab[mid, im] / tab[mid, ir] > 1.: hi = mid
            else: lo = mid
        tlo, thi, row = tau[lo], tau[hi], tab[lo]
        while thi - tlo > 1e-6:      # (precision parameter tol_tau_eq)
            tm = 0.5 * (tlo + thi)
            row = self._spline_row(tau, tab, d2, tm)
            if row[im] / row[ir] > 1.: thi = tm
            else: tlo = tm
        return float(row[ia] * row[iH])

    # -- values at a redshift (classy.pyx:825-1080: background_tau_of_z + background_at_tau with every column, thermodynamics_at_z)
    @staticmethod
    def _spline_row(x, tab, d2, v):
        """row of a table splined in its abscissa x at v (array_interpolate_spline, tools/arrays.c:1558-1640): every column at once"""
        n = x.size
        up = x[-1] > x[0]
        lo, hi = min(x[0], x[-1]), max(x[0], x[-1])
        if lo - 1e-9 * abs(lo) <= v <= hi + 1e-9 * abs(hi):
            v = min(max(v, lo), hi)          # (tau(z = 0) is the last abscissa up to rounding)
        if (v < lo) or (v > hi):
            raise CosmoSevereError("value %e outside the tabulated range [%e, %e]" % (v, min(x[0], x[-1]), max(x[0], x[-1])))
        i = int(np.searchsorted(x, v, side="right") - 1) if up else int(n - 1 - np.searchsorted(x[::-1], v, side="left"))
        i = min(max(i, 0), n - 2)
        h = x[i + 1] - x[i]
        b = (v - x[i]) / h
        a = 1. - b
        return a * tab[i] + b * tab[i + 1] + ((a * a * a - a) * d2[i] + (b * b * b - b) * d2[i + 1]) * h * h / 6.

    def _background_at_z(self, z):
        from . import hostlib
        r = self._need("background")
        if z < 0.:
            raise CosmoSevereError("asked for negative redshift z=%e" % z)
        t = r.inp.t
        try:
            tau = hostlib.tau_of_z(r.inp, float(z))
        except Exception as e:
            raise CosmoSevereError(str(e))
        return self._spline_row(np.asarray(t["bg.tau_table"]), np.asarray(t["bg.background_table"]), np.asarray(t["bg.d2background_dtau2_table"]), tau), tau

    def _bg_value_at_z(self, z, name):
        row, _ = self._background_at_z(z)
        idx = int(np.asarray(self._need("background").inp.t["bg.index_bg_" + name]).reshape(-1)[0])
        if idx < 0:
            raise CosmoSevereError("the background table holds no column %s" % name)
        return float(row[idx])

    def _th_value_at_z(self, z, name):
        t = self._need("thermodynamics").inp.t
        zt = np.asarray(t["th.z_table"])
        if z < 0.:
            raise CosmoSevereError("asked for negative redshift z=%e" % z)
        if z >= zt[-1]:
            raise CosmoSevereError("z=%e beyond the thermodynamics table (z_max = %e): only the perturbation kernel continues it analytically" % (z, zt[-1]))
        row = self._spline_row(zt, np.asarray(t["th.thermodynamics_table"]), np.asarray(t["th.d2thermodynamics_dz2_table"]), float(z))
        return float(row[int(np.asarray(t["th.index_th_" + name]).reshape(-1)[0])])

    def Hubble(self, z): return self._bg_value_at_z(z, "H")
    def Om_m(self, z): return self._bg_value_at_z(z, "Omega_m")
    def angular_distance(self, z): return self._bg_value_at_z(z, "ang_distance")
    def luminosity_distance(self, z): return self._bg_value_at_z(z, "lum_distance")
    def scale_independent_growth_factor(self, z): return self._bg_value_at_z(z, "D")
    def scale_independent_growth_factor_f(self, z): return self._bg_value_at_z(z, "f")
    def ionization_fraction(self, z): return self._th_value_at_z(z, "xe")
    def baryon_temperature(self, z): return self._th_value_at_z(z, "Tb")

    def z_of_tau(self, tau):
        t = self._need("background").inp.t
        row = self._spline_row(np.asarray(t["bg.tau_table"]), np.asarray(t["bg.background_table"]), np.asarray(t["bg.d2background_dtau2_table"]), float(tau))
        return 1. / float(row[int(np.asarray(t["bg.index_bg_a"]).reshape(-1)[0])]) - 1.

    def z_of_r(self, z_array):
        """(r(z), dz/dr) for an array of redshifts: comoving distance and H(z) (classy.pyx:395-443)"""
        z_array = np.asarray(z_array, dtype=np.float64)
        r = np.array([self._bg_value_at_z(float(z), "conf_distance") for z in z_array])
        dzdr = np.array([self._bg_value_at_z(float(z), "H") for z in z_array])
        return r, dzdr

    def get_current_derived_parameters(self, names):
        table = {"h": self.h, "H0": lambda: 100. * self.h(), "Omega_Lambda": self.Omega_Lambda, "Omega0_lambda": self.Omega_Lambda,
                 "Omega_m": self.Omega_m, "Neff": self.Neff, "age": self.age, "conformal_age": self.conformal_age, "tau_reio": self.tau_reio,
                 "z_reio": self.z_reio, "z_rec": self.z_rec, "tau_rec": lambda: self._t("th.tau_rec"), "rs_rec": lambda: self._t("th.rs_rec"),
                 "ra_rec": lambda: self._t("th.ra_rec"), "100*theta_s": self.theta_s_100, "YHe": lambda: self._t("th.YHe"),
                 "n_e": lambda: self._t("th.n_e"), "A_s": self.A_s, "ln10^{10}A_s": lambda: np.log(1e10 * self.A_s()), "n_s": self.n_s,
                 "sigma8": self.sigma8, "z_star": lambda: self._t("th.z_star"), "z_d": lambda: self._t("th.z_d")}
        out = {}
        for n in names:
            if n not in table:
                raise CosmoSevereError("%s was not recognized as a derived parameter" % n)
            out[n] = table[n]()
        return out

    # -- tables (classy.pyx:1093-1180): column titles follow the reference's output files for the columns this package computes
    def get_background(self):
        t = self._need("background").inp.t
        tab = t["bg.background_table"]
        out = {"z": np.asarray(t["bg.z_table"]).copy(), "conf. time [Mpc]": np.asarray(t["bg.tau_table"]).copy()}
        for key, title in (("a", "a"), ("H", "H [1/Mpc]"), ("rho_g", "(.)rho_g"), ("rho_b", "(.)rho_b"), ("rho_cdm", "(.)rho_cdm"),
                           ("rho_lambda", "(.)rho_lambda"), ("rho_ur", "(.)rho_ur"), ("rho_crit", "(.)rho_crit"), ("time", "proper time [Mpc]"),
                           ("rs", "comov.snd.hrz."), ("conf_distance", "comov. dist."), ("ang_distance", "ang.diam.dist."),
                           ("lum_distance", "lum. dist."), ("D", "gr.fac. D"), ("f", "gr.fac. f")):
            idx = t.get("bg.index_bg_" + key, -1)
            if isinstance(idx, np.ndarray):
                idx = int(idx.reshape(-1)[0])
            if idx is not None and idx >= 0:
                out[title] = tab[:, idx].copy()
        return out

    def get_thermodynamics(self):
        t = self._need("thermodynamics").inp.t
        tab = t["th.thermodynamics_table"]
        out = {"z": np.asarray(t["th.z_table"]).copy()}
        for key, title in (("xe", "x_e"), ("dkappa", "kappa' [Mpc^-1]"), ("exp_m_kappa", "exp(-kappa)"), ("g", "g [Mpc^-1]"), ("Tb", "Tb [K]"),
                           ("cb2", "c_b^2"), ("tau_d", "tau_d")):
            idx = int(np.asarray(t["th.index_th_" + key]).reshape(-1)[0])
            out[title] = tab[:, idx].copy()
        return out

    def get_transfer(self, z=0., output_format="class"):
        """density and velocity transfer functions at redshift z (classy.pyx:1303-1388, PerturbationsModule::perturb_output_data,
        pm.cpp:130-330): {'k (h/Mpc)', 'd_g', 'd_b', 'd_cdm', 'd_ur', 'd_tot', 'phi', 'psi', 't_g', 't_b', 't_cdm', 't_ur', 't_tot'} for the
        columns the run holds (output = mTk / dTk, vTk); z = 0: the last time sample, 0 < z <= z_max_pk: the sources splined in ln tau
        over the tail of the sampling, as perturb_sources_at_tau does; output_format = 'camb': the eight fixed columns -T_x / k^2"""
        if output_format not in ("class", "camb"):
            raise CosmoSevereError("get_transfer: output_format is 'class' or 'camb'")
        self._need("perturb")
        r = self._runs.get("s")
        c = r.inp.config if r is not None else None
        if r is None or not c.has_transfers:
            raise CosmoSevereError("No density or velocity transfer functions computed. You must add mTk and / or vTk to the list of outputs.")
        from .capi import TK_NAMES
        S = r.be.get_sources(r.inp.tau.size, r.inp.k.size).cpu().numpy()       # [tp][tau][k]
        if z == 0.:
            at = S[:, -1, :]
        else:
            tau_z, n = self._late_times_of(r, z)
            lt = np.log(r.inp.tau[-n:])
            if np.log(tau_z) < lt[0]:
                raise CosmoSevereError("Asking sources at a z bigger than z_max_pk, something probably went wrong")
            # (array_spline_table_lines with _SPLINE_EST_DERIV_, tools/arrays.c:514-640: a clamped cubic spline whose end slopes are the
            #  derivatives of the parabolas through the first / last three points)
            from scipy.interpolate import CubicSpline
            y = S[:, -n:, :]
            d0 = ((lt[2] - lt[0]) ** 2 * (y[:, 1] - y[:, 0]) - (lt[1] - lt[0]) ** 2 * (y[:, 2] - y[:, 0])) / ((lt[2] - lt[0]) * (lt[1] - lt[0]) * (lt[2] - lt[1]))
            d1 = ((lt[-3] - lt[-1]) ** 2 * (y[:, -2] - y[:, -1]) - (lt[-2] - lt[-1]) ** 2 * (y[:, -3] - y[:, -1])) / \
                 ((lt[-3] - lt[-1]) * (lt[-2] - lt[-1]) * (lt[-3] - lt[-2]))
            at = CubicSpline(lt, y, axis=1, bc_type=((1, d0), (1, d1)))(np.log(tau_z))
        out = {"k (h/Mpc)": r.inp.k / self.h()}
        if output_format == "camb":
            # the CMBFAST / CAMB convention (pm.cpp:289-300, titles :377-389): minus the density transfer functions over k^2, fixed
            # columns; absent species read zero and of several non-cold species only the first is written
            k2 = r.inp.k * r.inp.k

            def scaled(index):
                return -at[index] / k2 if index >= 0 else np.zeros_like(k2)
            for title, name in (("-T_cdm/k2", "delta_cdm"), ("-T_idm_dr/k2", None), ("-T_b/k2", "delta_b"), ("-T_g/k2", "delta_g"),
                                ("-T_ur/k2", "delta_ur"), ("-T_idr/k2", None), ("-T_ncdm/k2", "ncdm"), ("-T_tot/k2", "delta_tot")):
                if name == "ncdm":
                    out[title] = scaled(int(c.index_tp_delta_ncdm1) if c.has_ncdm else -1)
                else:
                    out[title] = scaled(int(c.index_tp_transfer[TK_NAMES.index(name)]) if name else -1)
            return out
        titles = {"delta_g": "d_g", "delta_b": "d_b", "delta_cdm": "d_cdm", "delta_ur": "d_ur", "delta_tot": "d_tot", "phi": "phi", "psi": "psi",
                  "theta_g": "t_g", "theta_b": "t_b", "theta_cdm": "t_cdm", "theta_ur": "t_ur", "theta_tot": "t_tot"}
        order = ("delta_g", "delta_b", "delta_cdm", "delta_ur", "delta_tot", "phi", "psi", "theta_g", "theta_b", "theta_cdm", "theta_ur", "theta_tot")
        for name in order:
            idx = int(c.index_tp_transfer[TK_NAMES.index(name)])
            if idx >= 0:
                out[titles[name]] = at[idx].copy()
            if c.has_ncdm and name in ("delta_ur", "theta_ur"):      # (pm.cpp:248-251, 276-279: the species follow ur)
                first = int(c.index_tp_delta_ncdm1 if name == "delta_ur" else c.index_tp_theta_ncdm1)
                if first >= 0:
                    for i in range(int(c.N_ncdm)):
                        out["%s_ncdm[%d]" % ("d" if name == "delta_ur" else "t", i)] = at[first + i].copy()
        return out

    def _late_times_of(self, r, z):
        from . import hostlib
        if z < 0.:
            raise CosmoSevereError("asked for negative redshift z=%e" % z)
        zmax = float(r.inp.d["ppt.z_max_pk"][0])
        if zmax == 0. or z > zmax:
            raise CosmoSevereError("get_transfer at z=%e needs z_max_pk >= z in the input (the sources are kept up to z_max_pk = %e)" % (z, zmax))
        n = hostlib.ln_tau_size(r.inp.tau, hostlib.tau_of_z(r.inp, zmax))
        return hostlib.tau_of_z(r.inp, z), n

    def get_sources(self):
        """(sources, k, tau): the source functions S(k, tau) of the scalar run, {name: [k_size][tau_size]} like the reference's
        get_sources (classy.pyx; perturbations_module.h:11-40 for the types)"""
        self._need("perturb")
        r = self._runs.get("s") or next(iter(self._runs.values()))
        c = r.inp.config
        S = r.be.get_sources(r.inp.tau.size, r.inp.k.size).cpu().numpy()
        out = {}
        for name in ("t0", "t1", "t2", "p", "delta_m", "phi_plus_psi"):
            idx = getattr(c, "index_tp_" + name)
            if idx >= 0:
                out[name] = S[idx].T.copy()
        if c.has_transfers:
            from .capi import TK_NAMES
            for i, name in enumerate(TK_NAMES):
                if c.index_tp_transfer[i] >= 0:
                    out[name] = S[int(c.index_tp_transfer[i])].T.copy()
        return out, r.inp.k.copy(), r.inp.tau.copy()

    def __del__(self):
        try:
            self.struct_cleanup()
        except Exception:
            pass
