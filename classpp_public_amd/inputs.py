"""Hot-path inputs for a named configuration.

The backend's inputs (background / thermodynamics spline tables, k / tau / q / l grids, precision and physics
parameters) are deterministic functions of an .ini, produced in the reference by modules that sit UPSTREAM of the
hot path (InputModule, BackgroundModule, ThermodynamicsModule).  Two sources: the committed fixtures tests/golden/*.npz
dumped from the unmodified reference by oracle/make_fixtures.py (every configuration), or - for LambdaCDM with massless
neutrinos - the backend's own host-side background + RECFAST + grid builders (classpp_public_amd/pipeline.py).
"""
import ctypes as C
import os

import numpy as np

from .capi import CptConfig, CptSpectraParams, CptTables

GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


# committed table fixture of every committed configuration (several configurations share one cosmology)
TABLES_OF = {
    "small": "tables_lcdm.npz", "lcdm": "tables_lcdm.npz", "explanatory": "tables_lcdm.npz", "explanatory_mpk": "tables_lcdm.npz",
    "iso_cdi": "tables_lcdm.npz", "iso_nid": "tables_lcdm.npz", "newt": "tables_lcdm.npz", "tens": "tables_lcdm.npz",
    "tca_mb": "tables_lcdm.npz", "lcdm_zpk": "tables_lcdm.npz", "lcdm_tk": "tables_lcdm.npz", "lcdm_zpk_tk": "tables_lcdm.npz", "small_tk": "tables_lcdm.npz", "newt_tk": "tables_lcdm.npz", "long_small": "tables_lcdm.npz", "long_full": "tables_lcdm.npz", "newt_full": "tables_lcdm.npz", "iso_bi_full": "tables_lcdm.npz", "iso_niv_full": "tables_lcdm.npz", "tens_full": "tables_lcdm.npz",
    "curved": "tables_curved.npz", "curved_full": "tables_curved.npz", "tens_curved": "tables_curved.npz", "open": "tables_open.npz",
    "ncdm": "tables_ncdm1.npz", "ncdm_small": "tables_ncdm1.npz", "ncdm_small_tk": "tables_ncdm1.npz", "ncdm3_small_tk": "tables_ncdm3.npz", "ncdm_k3000": "tables_ncdm1.npz",
    "ncdm_permille": "tables_ncdm1.npz", "ncdm_permille_small": "tables_ncdm1.npz",
    "ncdm3": "tables_ncdm3.npz", "ncdm3_small": "tables_ncdm3.npz", "ncdm3_tens": "tables_ncdm3.npz",
}


def _check_tables_match(name, d, t):
    """the parameters of the configuration and the table file must describe one cosmology: H0, and the densities today"""
    bg = t["bg.background_table"]
    last = bg[-1]
    H0 = float(d["pba.H0"].reshape(-1)[0])

    def col(f):
        return float(last[int(t["bg.index_bg_" + f].reshape(-1)[0])])

    checks = [("H0", col("H"), H0), ("Omega0_g", col("rho_g") / H0 ** 2, float(d["pba.Omega0_g"].reshape(-1)[0])),
              ("Omega0_b", col("rho_b") / H0 ** 2, float(d["pba.Omega0_b"].reshape(-1)[0]))]
    if int(d["pba.has_cdm"].reshape(-1)[0]):
        checks.append(("Omega0_cdm", col("rho_cdm") / H0 ** 2, float(d["pba.Omega0_cdm"].reshape(-1)[0])))
    for what, got, want in checks:
        if not abs(got - want) <= 1e-6 * abs(want):
            raise ValueError("configuration %r and its table file disagree on %s: %.10g (tables) vs %.10g (parameters)" % (name, what, got, want))


def _s(d, key):
    v = d[key]
    return v.reshape(-1)[0]


class Inputs:
    """config + tables + grids for one configuration (`small`, `lcdm`, `explanatory`)."""

    def __init__(self, name, golden_dir=GOLDEN, tables=None, params=None):
        """tables: dict keyed like the reference's table dump (bg.*, th.*) to use instead of the committed table fixture,
        e.g. the output of the host background / thermodynamics modules (classpp_public_amd/pipeline.py)"""
        self.name = name
        # params: the parameter / flag entries (pba.*, ppt.*, ppr.*, index maps ...) given directly instead of a committed fixture
        self.d = dict(params) if params is not None else dict(np.load(os.path.join(golden_dir, name + ".npz")))
        if tables is not None:
            self.t = {k: np.atleast_1d(np.asarray(v)) for k, v in tables.items()}
        else:
            tname = TABLES_OF.get(name)
            if tname is None or not os.path.exists(os.path.join(golden_dir, tname)):
                raise FileNotFoundError("no committed background/thermodynamics tables for configuration %r: pass tables= "
                                        "(classpp_public_amd/pipeline.py computes them) or add it to inputs.TABLES_OF" % (name,))
            self.t = dict(np.load(os.path.join(golden_dir, tname)))
            _check_tables_match(name, self.d, self.t)
        d, t = self.d, self.t
        c = CptConfig()
        c.H0 = _s(d, "pba.H0"); c.K = _s(d, "pba.K"); c.sgnK = int(_s(d, "pba.sgnK"))
        c.has_cdm = int(_s(d, "pba.has_cdm")); c.has_ur = int(_s(d, "pba.has_ur"))
        c.has_ncdm = int(_s(d, "pba.has_ncdm")); c.has_fld = int(_s(d, "pba.has_fld"))
        c.has_curvature = int(_s(d, "pba.has_curvature"))
        c.T_cmb = _s(d, "pba.T_cmb"); c.a_today = _s(d, "pba.a_today")
        c.YHe = _s(t, "th.YHe"); c.n_e = _s(t, "th.n_e"); c.tau0 = _s(t, "bg.conformal_age")
        c.tau_rec = _s(t, "th.tau_rec"); c.tau_free_streaming = _s(t, "th.tau_free_streaming")
        c.tau_cut = _s(t, "th.tau_cut"); c.angular_rescaling = _s(t, "th.angular_rescaling")
        c.gauge = int(_s(d, "ppt.gauge"))
        for f in ("switch_sw", "switch_eisw", "switch_lisw", "switch_dop", "switch_pol"):
            setattr(c, f, int(_s(d, "ppt." + f)))
        c.eisw_lisw_split_z = _s(d, "ppt.eisw_lisw_split_z")
        c.three_ceff2_ur = _s(d, "ppt.three_ceff2_ur"); c.three_cvis2_ur = _s(d, "ppt.three_cvis2_ur")
        c.tp_size = int(_s(d, "pt.tp_size"))
        for f in ("t0", "t1", "t2", "p", "delta_m", "phi_plus_psi"):
            setattr(c, "index_tp_" + f, int(_s(d, "pt.index_tp_" + f)))
        for f in ("start_small_k_at_tau_c_over_tau_h", "start_large_k_at_tau_h_over_tau_k",
                  "tight_coupling_trigger_tau_c_over_tau_h", "tight_coupling_trigger_tau_c_over_tau_k",
                  "radiation_streaming_trigger_tau_over_tau_k", "ur_fluid_trigger_tau_over_tau_k", "curvature_ini",
                  "tol_perturb_integration", "tol_tau_approx", "smallest_allowed_variation",
                  "hyper_x_min", "hyper_sampling_flat", "hyper_phi_min_abs", "hyper_sampling_curved_low_nu",
                  "hyper_sampling_curved_high_nu", "hyper_nu_sampling_step", "hyper_flat_approximation_nu",
                  "transfer_neglect_delta_k_S_t0", "transfer_neglect_delta_k_S_t1",
                  "transfer_neglect_delta_k_S_t2", "transfer_neglect_delta_k_S_e",
                  "transfer_neglect_late_source", "l_switch_limber"):
            setattr(c, f, float(_s(d, "ppr." + f)))
        for f in ("tight_coupling_approximation", "radiation_streaming_approximation", "ur_fluid_approximation",
                  "l_max_g", "l_max_pol_g", "l_max_ur"):
            setattr(c, f, int(_s(d, "ppr." + f)))
        self.has_cls = ("tr.q" in d) or ("tr.tt_size" in d and int(_s(d, "tr.tt_size")) > 0)
        if self.has_cls:
            c.tt_size = int(_s(d, "tr.tt_size"))
            for f in ("t0", "t1", "t2", "e", "lcmb"):
                setattr(c, "index_tt_" + f, int(_s(d, "tr.index_tt_" + f)))
        c.lcmb_rescale = _s(d, "ptr.lcmb_rescale"); c.lcmb_tilt = _s(d, "ptr.lcmb_tilt")
        c.lcmb_pivot = _s(d, "ptr.lcmb_pivot")
        # tensor modes: one mode per handle (a tensors-only reference run dumps pt.mode_tensors = 1)
        c.mode = int(_s(d, "pt.mode_tensors")) if "pt.mode_tensors" in d else 0
        c.l_max_g_ten = int(_s(d, "ppr.l_max_g_ten")); c.l_max_pol_g_ten = int(_s(d, "ppr.l_max_pol_g_ten"))
        c.gw_ini = float(_s(d, "ppr.gw_ini"))
        c.evolve_tensor_ur = int(_s(d, "pt.evolve_tensor_ur")) if "pt.evolve_tensor_ur" in d else 0
        c.index_tt_b = int(_s(d, "tr.index_tt_b")) if "tr.index_tt_b" in d else -1
        for f in ("t2", "e", "b"):
            key = "ppr.transfer_neglect_delta_k_T_" + f
            setattr(c, "transfer_neglect_delta_k_T_" + f, float(_s(d, key)) if key in d else 0.0)
        # non-cold dark matter (massive neutrinos): momentum grids travel in the tables struct below
        c.N_ncdm = int(_s(d, "pba.N_ncdm")) if c.has_ncdm else 0
        c.l_max_ncdm = int(_s(d, "ppr.l_max_ncdm")); c.ncdm_fluid_approximation = int(_s(d, "ppr.ncdm_fluid_approximation"))
        c.ncdm_fluid_trigger_tau_over_tau_k = float(_s(d, "ppr.ncdm_fluid_trigger_tau_over_tau_k"))
        c.tol_ncdm_initial_w = float(_s(d, "ppr.tol_ncdm_initial_w")) if "ppr.tol_ncdm_initial_w" in d else 1e-3
        c.index_tp_delta_cb = int(_s(d, "pt.index_tp_delta_cb")) if "pt.index_tp_delta_cb" in d else -1
        c.tensor_method = int(_s(d, "ppt.tensor_method")) if "ppt.tensor_method" in d else 1
        from .capi import TK_NAMES
        for i, name in enumerate(TK_NAMES):      # density / velocity transfer sources (output = mTk, vTk)
            c.index_tp_transfer[i] = int(_s(d, "pt.index_tp_" + name)) if ("pt.index_tp_" + name) in d else -1
        c.index_tp_delta_ncdm1 = int(_s(d, "pt.index_tp_delta_ncdm1")) if "pt.index_tp_delta_ncdm1" in d else -1
        c.index_tp_theta_ncdm1 = int(_s(d, "pt.index_tp_theta_ncdm1")) if "pt.index_tp_theta_ncdm1" in d else -1
        c.has_transfers = int(any(c.index_tp_transfer[i] >= 0 for i in range(len(TK_NAMES))))
        # initial condition: one mode per handle (ad unless the fixture says otherwise)
        c.ic = 0
        for code, key in ((1, "ppt.has_bi"), (2, "ppt.has_cdi"), (3, "ppt.has_nid"), (4, "ppt.has_niv")):
            if key in d and int(_s(d, key)) and not int(_s(d, "ppt.has_ad")):
                c.ic = code
        c.entropy_ini = float(_s(d, "ppr.entropy_ini")) if "ppr.entropy_ini" in d else 1.0
        self.config = c

        # tables (keep numpy arrays alive: the struct only holds raw pointers)
        self._keep = []

        def ptr(a):
            a = np.ascontiguousarray(a, dtype=np.float64)
            self._keep.append(a)
            return a.ctypes.data_as(C.POINTER(C.c_double))

        tb = CptTables()
        tb.bt_size = int(_s(t, "bg.bt_size")); tb.bg_size = int(_s(t, "bg.bg_size"))
        tb.tau_table = ptr(t["bg.tau_table"]); tb.background_table = ptr(t["bg.background_table"])
        tb.d2background_dtau2_table = ptr(t["bg.d2background_dtau2_table"])
        for f in ("a", "H", "H_prime", "rho_g", "rho_b", "rho_cdm", "rho_ur"):
            setattr(tb, "index_bg_" + f, int(_s(t, "bg.index_bg_" + f)))
        tb.tt_size = int(_s(t, "th.tt_size")); tb.th_size = int(_s(t, "th.th_size"))
        tb.z_table = ptr(t["th.z_table"]); tb.thermodynamics_table = ptr(t["th.thermodynamics_table"])
        tb.d2thermodynamics_dz2_table = ptr(t["th.d2thermodynamics_dz2_table"])
        for f in ("xe", "dkappa", "tau_d", "ddkappa", "dddkappa", "exp_m_kappa", "g", "dg", "cb2", "rate"):
            setattr(tb, "index_th_" + f, int(_s(t, "th.index_th_" + f)))
        if c.has_ncdm:
            for f in ("rho_ncdm1", "p_ncdm1", "pseudo_p_ncdm1"):
                setattr(tb, "index_bg_" + f, int(_s(t, "bg.index_bg_" + f)))
            for n in range(c.N_ncdm):
                tb.q_size_ncdm[n] = t["ncdm.q_%d" % n].size
                tb.q_ncdm[n] = ptr(t["ncdm.q_%d" % n]); tb.w_ncdm[n] = ptr(t["ncdm.w_%d" % n])
                tb.dlnf0_dlnq_ncdm[n] = ptr(t["ncdm.dlnf0_dlnq_%d" % n])
                tb.M_ncdm[n] = float(t["ncdm.M"][n]); tb.factor_ncdm[n] = float(t["ncdm.factor"][n])
        self.tables = tb

        # primordial spectrum + C_l slots (struct primordial / SpectraModule index_ct_*)
        sp = CptSpectraParams()
        sp.A_s = _s(d, "ppm.A_s"); sp.n_s = _s(d, "ppm.n_s"); sp.alpha_s = _s(d, "ppm.alpha_s"); sp.k_pivot = _s(d, "ppm.k_pivot")
        if "ppm.amplitude0" in d:  # effective power law of the (single) initial condition, e.g. A_s f_cdi^2, n_cdi
            sp.A_s = _s(d, "ppm.amplitude0"); sp.n_s = _s(d, "ppm.tilt0"); sp.alpha_s = _s(d, "ppm.running0")
        sp.ct_size = int(_s(d, "sp.ct_size")) if "sp.ct_size" in d else 0
        for f in ("tt", "ee", "te", "pp", "tp", "ep"):
            setattr(sp, "index_ct_" + f, int(_s(d, "sp.index_ct_" + f)) if ("sp.index_ct_" + f) in d else -1)
        used = {getattr(sp, "index_ct_" + f) for f in ("tt", "ee", "te", "pp", "tp", "ep")}
        sp.index_ct_bb = int(_s(d, "sp.index_ct_bb")) if "sp.index_ct_bb" in d else -1
        for i in range(sp.ct_size):  # older fixtures: the remaining slot of a scalar run is BB (identically zero)
            if i not in used and sp.index_ct_bb < 0:
                sp.index_ct_bb = i
        self.spectra = sp

        # grids
        if "pt.k" not in d:   # parameters only: the grids are built by the caller (classpp_public_amd/pipeline.py)
            self.k = self.tau = self.q = self.l = None
            self.k_size_cl = 0
            return
        self.k = np.ascontiguousarray(d["pt.k"], dtype=np.float64)
        self.k_size_cl = int(_s(d, "pt.k_size_cl"))
        self.tau = np.ascontiguousarray(d["pt.tau_sampling"], dtype=np.float64)
        if self.has_cls:
            self.q = np.ascontiguousarray(d["tr.q"], dtype=np.float64)
            self.l = np.ascontiguousarray(d["tr.l"], dtype=np.int32)

    @property
    def nk(self):
        return self.k.size

    @property
    def ntau(self):
        return self.tau.size
