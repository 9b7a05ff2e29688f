"""ctypes mirror of include/cpt.h (the C ABI of the backend) and loader of the in-tree HIP library.

The product path: Python -> ctypes -> classpp_public_amd/csrc/libcpt.so (hand-written HIP for gfx950).
There is no CPU fallback: if the shared library is missing, importing `lib()` raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libcpt.so")

CPT_OK, CPT_ERR_INVALID, CPT_ERR_RUNTIME, CPT_ERR_NO_DEVICE, CPT_ERR_UNSUPPORTED = 0, 1, 2, 3, 4

_d, _i = C.c_double, C.c_int


class CptConfig(C.Structure):
    """struct cpt_config (include/cpt.h) -- field order must match exactly."""
    _fields_ = [
        ("H0", _d), ("K", _d), ("sgnK", _i),
        ("has_cdm", _i), ("has_ur", _i), ("has_ncdm", _i), ("has_fld", _i), ("has_curvature", _i),
        ("T_cmb", _d), ("a_today", _d),
        ("YHe", _d), ("n_e", _d), ("tau0", _d), ("tau_rec", _d), ("tau_free_streaming", _d), ("tau_cut", _d),
        ("angular_rescaling", _d),
        ("gauge", _i),
        ("switch_sw", _i), ("switch_eisw", _i), ("switch_lisw", _i), ("switch_dop", _i), ("switch_pol", _i),
        ("eisw_lisw_split_z", _d), ("three_ceff2_ur", _d), ("three_cvis2_ur", _d),
        ("tp_size", _i),
        ("index_tp_t0", _i), ("index_tp_t1", _i), ("index_tp_t2", _i), ("index_tp_p", _i),
        ("index_tp_delta_m", _i), ("index_tp_phi_plus_psi", _i),
        ("start_small_k_at_tau_c_over_tau_h", _d), ("start_large_k_at_tau_h_over_tau_k", _d),
        ("tight_coupling_trigger_tau_c_over_tau_h", _d), ("tight_coupling_trigger_tau_c_over_tau_k", _d),
        ("tight_coupling_approximation", _i),
        ("radiation_streaming_approximation", _i), ("radiation_streaming_trigger_tau_over_tau_k", _d),
        ("ur_fluid_approximation", _i), ("ur_fluid_trigger_tau_over_tau_k", _d),
        ("l_max_g", _i), ("l_max_pol_g", _i), ("l_max_ur", _i),
        ("curvature_ini", _d),
        ("tol_perturb_integration", _d), ("tol_tau_approx", _d), ("smallest_allowed_variation", _d),
        ("tt_size", _i),
        ("index_tt_t0", _i), ("index_tt_t1", _i), ("index_tt_t2", _i), ("index_tt_e", _i), ("index_tt_lcmb", _i),
        ("lcmb_rescale", _d), ("lcmb_tilt", _d), ("lcmb_pivot", _d),
        ("hyper_x_min", _d), ("hyper_sampling_flat", _d), ("hyper_phi_min_abs", _d),
        ("transfer_neglect_delta_k_S_t0", _d), ("transfer_neglect_delta_k_S_t1", _d),
        ("transfer_neglect_delta_k_S_t2", _d), ("transfer_neglect_delta_k_S_e", _d),
        ("transfer_neglect_late_source", _d), ("l_switch_limber", _d),
        ("ic", _i), ("entropy_ini", _d),
        ("mode", _i), ("l_max_g_ten", _i), ("l_max_pol_g_ten", _i), ("gw_ini", _d), ("evolve_tensor_ur", _i), ("index_tt_b", _i),
        ("transfer_neglect_delta_k_T_t2", _d), ("transfer_neglect_delta_k_T_e", _d), ("transfer_neglect_delta_k_T_b", _d),
        ("hyper_sampling_curved_low_nu", _d), ("hyper_sampling_curved_high_nu", _d), ("hyper_nu_sampling_step", _d),
        ("hyper_flat_approximation_nu", _d),
        ("N_ncdm", _i), ("l_max_ncdm", _i), ("ncdm_fluid_approximation", _i), ("ncdm_fluid_trigger_tau_over_tau_k", _d),
        ("tol_ncdm_initial_w", _d), ("index_tp_delta_cb", _i), ("tensor_method", _i),
        ("has_transfers", _i), ("index_tp_transfer", _i * 12), ("index_tp_delta_ncdm1", _i), ("index_tp_theta_ncdm1", _i),
    ]


# cpt_config.index_tp_transfer is indexed by these (enum CPT_TK_* of include/cpt.h; names as in the reference's index_tp_<name>_)
TK_NAMES = ("delta_tot", "delta_g", "delta_b", "delta_cdm", "delta_ur", "theta_tot", "theta_g", "theta_b", "theta_cdm", "theta_ur", "phi", "psi")


_pd = C.POINTER(_d)


MAX_NCDM = 3


class CptTables(C.Structure):
    """struct cpt_tables (include/cpt.h)."""
    _fields_ = [
        ("bt_size", _i), ("bg_size", _i),
        ("tau_table", _pd), ("background_table", _pd), ("d2background_dtau2_table", _pd),
        ("index_bg_a", _i), ("index_bg_H", _i), ("index_bg_H_prime", _i), ("index_bg_rho_g", _i),
        ("index_bg_rho_b", _i), ("index_bg_rho_cdm", _i), ("index_bg_rho_ur", _i),
        ("tt_size", _i), ("th_size", _i),
        ("z_table", _pd), ("thermodynamics_table", _pd), ("d2thermodynamics_dz2_table", _pd),
        ("index_th_xe", _i), ("index_th_dkappa", _i), ("index_th_tau_d", _i), ("index_th_ddkappa", _i),
        ("index_th_dddkappa", _i), ("index_th_exp_m_kappa", _i), ("index_th_g", _i), ("index_th_dg", _i),
        ("index_th_cb2", _i), ("index_th_rate", _i),
        ("index_bg_rho_ncdm1", _i), ("index_bg_p_ncdm1", _i), ("index_bg_pseudo_p_ncdm1", _i),
        ("q_size_ncdm", _i * MAX_NCDM), ("q_ncdm", _pd * MAX_NCDM), ("w_ncdm", _pd * MAX_NCDM),
        ("dlnf0_dlnq_ncdm", _pd * MAX_NCDM), ("M_ncdm", _d * MAX_NCDM), ("factor_ncdm", _d * MAX_NCDM),
    ]


class CptStepstat(C.Structure):
    _fields_ = [("steps", _i), ("failed", _i), ("fevals", _i), ("jacobians", _i), ("factorisations", _i),
                ("solves", _i), ("n_regimes", _i), ("tau_ini", _d)]


class CptSpectraParams(C.Structure):
    _fields_ = [("A_s", _d), ("n_s", _d), ("alpha_s", _d), ("k_pivot", _d), ("ct_size", _i),
                ("index_ct_tt", _i), ("index_ct_ee", _i), ("index_ct_te", _i), ("index_ct_bb", _i), ("index_ct_pp", _i),
                ("index_ct_tp", _i), ("index_ct_ep", _i)]


class CptLensingParams(C.Structure):
    _fields_ = [("l_unlensed_max", _i), ("delta_l_max", _i), ("accurate_lensing", _i), ("num_mu_minus_lmax", _i),
                ("tol_gauss_legendre", _d)]


class CptStepIo(C.Structure):
    """struct cpt_step_io (include/cpt.h): arguments of the fused cpt_step"""
    _fields_ = [("k", _pd), ("nk", _i), ("k_size_cl", _i), ("tau_sampling", _pd), ("ntau", _i), ("q", _pd), ("nq", _i),
                ("l", C.POINTER(_i)), ("nl", _i), ("sp", C.POINTER(CptSpectraParams)), ("lp", C.POINTER(CptLensingParams)),
                ("transfer_dev", C.c_void_p), ("cl_dev", C.c_void_p), ("cl_lensed_dev", C.c_void_p), ("pk_dev", C.c_void_p),
                ("stats", C.POINTER(CptStepstat)), ("status", C.POINTER(_i))]


# every symbol include/cpt.h declares (tests check that the built library exports all of them)
EXPORTS = [
    "cpt_create", "cpt_destroy", "cpt_last_error", "cpt_create_error", "cpt_perturb_solve_batch",
    "cpt_transfer_batch", "cpt_get_sources", "cpt_last_kernel_ms", "cpt_last_transfer_work",
    "cpt_dbg_lookup", "cpt_dbg_derivs", "cpt_dbg_solve", "cpt_dbg_bessel", "cpt_cl_batch", "cpt_cl_cross_batch", "cpt_sigma_of_pk", "cpt_pk_linear", "cpt_sigma", "cpt_pk_cb_linear", "cpt_sigma_cb",
    "cpt_lensing_l_size", "cpt_lensing_batch", "cpt_step", "cpt_pk_at_tau", "cpt_sigma_at_tau",
    "cpt_comm_get_unique_id", "cpt_comm_init", "cpt_comm_destroy", "cpt_allgather_sources", "cpt_gather_transfer", "cpt_gather_cl",
    "cpt_dbg_pad_rows", "cpt_dbg_uninterleave",
]

_lib = None


def lib():
    """Load libcpt.so (built in-tree by __graft_entry__.build()). Fails loudly when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "HIP extension %s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback for the product path)" % LIB_PATH)
    # PyTorch-ROCm ships its own HIP runtime: load it first so that libcpt.so binds to the SAME libamdhip64
    # (two HIP runtimes in one process do not share the device context / streams / allocations).
    import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    vp, pi, ll = C.c_void_p, C.POINTER(_i), C.POINTER(C.c_longlong)
    L.cpt_create.argtypes = [C.POINTER(CptConfig), C.POINTER(CptTables), C.POINTER(vp)]
    L.cpt_create.restype = _i
    L.cpt_destroy.argtypes = [vp]
    L.cpt_destroy.restype = None
    L.cpt_last_error.argtypes = [vp]
    L.cpt_last_error.restype = C.c_char_p
    L.cpt_create_error.argtypes = []
    L.cpt_create_error.restype = C.c_char_p
    L.cpt_perturb_solve_batch.argtypes = [vp, _pd, _i, _pd, _i, vp, C.POINTER(CptStepstat), pi]
    L.cpt_perturb_solve_batch.restype = _i
    L.cpt_transfer_batch.argtypes = [vp, vp, _pd, _i, _i, _pd, _i, _pd, _i, pi, _i, vp]
    L.cpt_transfer_batch.restype = _i
    L.cpt_get_sources.argtypes = [vp, vp]
    L.cpt_get_sources.restype = _i
    L.cpt_last_kernel_ms.argtypes = [vp, _i, _pd, pi]
    L.cpt_last_kernel_ms.restype = _i
    L.cpt_last_transfer_work.argtypes = [vp, ll, ll, ll]
    L.cpt_last_transfer_work.restype = _i
    L.cpt_cl_batch.argtypes = [vp, C.POINTER(CptSpectraParams), vp, _pd, _i, _i, vp]
    L.cpt_cl_batch.restype = _i
    L.cpt_cl_cross_batch.argtypes = [vp, C.POINTER(CptSpectraParams), vp, vp, _pd, _i, _i, vp]
    L.cpt_cl_cross_batch.restype = _i
    L.cpt_sigma_of_pk.argtypes = [_pd, _pd, _i, C.c_double, C.c_double, C.POINTER(C.c_double)]
    L.cpt_sigma_of_pk.restype = _i
    L.cpt_pk_linear.argtypes = [vp, C.POINTER(CptSpectraParams), _pd, _i, vp]
    L.cpt_pk_linear.restype = _i
    L.cpt_sigma.argtypes = [vp, C.POINTER(CptSpectraParams), _pd, _i, _d, _d, C.POINTER(_d)]
    L.cpt_sigma.restype = _i
    L.cpt_pk_cb_linear.argtypes = L.cpt_pk_linear.argtypes
    L.cpt_pk_cb_linear.restype = _i
    L.cpt_sigma_cb.argtypes = L.cpt_sigma.argtypes
    L.cpt_sigma_cb.restype = _i
    L.cpt_pk_at_tau.argtypes = [vp, C.POINTER(CptSpectraParams), _pd, _i, _i, _d, _i, vp]
    L.cpt_pk_at_tau.restype = _i
    L.cpt_sigma_at_tau.argtypes = [vp, C.POINTER(CptSpectraParams), _pd, _i, _i, _d, _i, _d, _d, C.POINTER(_d)]
    L.cpt_sigma_at_tau.restype = _i
    L.cpt_lensing_l_size.argtypes = [pi, _i, C.POINTER(CptLensingParams)]
    L.cpt_lensing_l_size.restype = _i
    L.cpt_lensing_batch.argtypes = [vp, C.POINTER(CptSpectraParams), C.POINTER(CptLensingParams), pi, _i, vp, vp]
    L.cpt_lensing_batch.restype = _i
    L.cpt_step.argtypes = [vp, C.POINTER(CptStepIo)]
    L.cpt_step.restype = _i
    L.cpt_comm_get_unique_id.argtypes = [vp]
    L.cpt_comm_get_unique_id.restype = _i
    L.cpt_comm_init.argtypes = [vp, vp, _i, _i]
    L.cpt_comm_init.restype = _i
    L.cpt_comm_destroy.argtypes = [vp]
    L.cpt_comm_destroy.restype = _i
    L.cpt_allgather_sources.argtypes = [vp, _i, _i]
    L.cpt_allgather_sources.restype = _i
    L.cpt_gather_transfer.argtypes = [vp, vp, _i, _i, vp]
    L.cpt_gather_transfer.restype = _i
    L.cpt_gather_cl.argtypes = [vp, vp, _i, _i, vp]
    L.cpt_gather_cl.restype = _i
    L.cpt_dbg_pad_rows.argtypes = [vp, vp, vp, _i, _i, _i, _i]
    L.cpt_dbg_pad_rows.restype = _i
    L.cpt_dbg_uninterleave.argtypes = [vp, vp, vp, _i, _i, _i, _i, _i]
    L.cpt_dbg_uninterleave.restype = _i
    L.cpt_dbg_lookup.argtypes = [vp, _pd, _i, _pd]
    L.cpt_dbg_lookup.restype = _i
    L.cpt_dbg_derivs.argtypes = [vp, _d, _d, _i, _i, _i, _pd, _pd, pi]
    L.cpt_dbg_derivs.restype = _i
    L.cpt_dbg_solve.argtypes = [vp, _d, _d, _i, _i, _i, _d, _pd, _pd]
    L.cpt_dbg_solve.restype = _i
    L.cpt_dbg_bessel.argtypes = [vp, pi, _i, _d, pi, _pd, _pd, _pd, _i]
    L.cpt_dbg_bessel.restype = _i
    _lib = L
    return L
