/* cpt.h -- C ABI of the MI355X-native perturbations -> transfer backend ("cpt").
 *
 * This is the drop-in boundary for ONE hot path of CLASS++ (AarhusCosmology/CLASSpp_public):
 *   - the per-k stiff ODE integration of PerturbationsModule (loop body at
 *     source/perturbations_module.cpp:668-718, i.e. perturb_solve :2463-2787 driving tools/evolver_ndf15.cpp:62-705)
 *   - the line-of-sight Bessel quadrature of TransferModule (loop body at source/transfer_module.cpp:287-318,
 *     i.e. transfer_compute_for_each_q :1488-1715).
 * The reference has no FFI for this path; the seam is its C++ class API.  The two batched entry
 * points below replace the BODIES of those two loops; thin C++ shim classes (classpp_public_amd/host/) reproduce
 * the constructors / public data of PerturbationsModule / TransferModule on top of them.
 *
 * Conventions
 *   - plain C, no exceptions cross the ABI: every call returns CPT_OK (0) or a nonzero error code and leaves a
 *     message retrievable with cpt_last_error(); per-item status arrays report per-k-mode / per-q failures
 *     (mirrors _SUCCESS_/_FAILURE_ + ErrorMsg, include/common.h:140-330).
 *   - all arithmetic is IEEE double ("f64"); integers only for indices and flags.
 *   - pointers named *_dev are DEVICE (HBM) pointers, all others are HOST pointers.  The library never takes
 *     ownership of caller memory.
 *   - a handle is bound to the HIP device current at cpt_create(): every entry point makes that device current for its
 *     duration and restores the caller's device on return.  Calls on one handle must be serialised by the caller (the
 *     reference's module constructors are single-caller too, source/cosmology.cpp:16-86).
 *   - streams: every handle owns one non-blocking HIP stream; each entry point enqueues there and returns with that stream
 *     drained (one synchronisation per call, at its end), so outputs are complete on return.  Device INPUT buffers must be complete before the call: a caller that
 *     produced them on another stream (an RCCL collective, a framework's stream) synchronises that stream first
 *     (classpp_public_amd/backend.py::Backend._fence does exactly that for torch).
 *   - there is NO CPU fallback: without a usable HIP device every compute entry point fails with CPT_ERR_NO_DEVICE.
 */
#ifndef CPT_H
#define CPT_H

#ifdef __cplusplus
extern "C" {
#endif

#define CPT_OK 0
#define CPT_ERR_INVALID 1     /* inconsistent / unsupported argument (reference: std::invalid_argument)      */
#define CPT_ERR_RUNTIME 2     /* failure inside the computation (reference: std::runtime_error)              */
#define CPT_ERR_NO_DEVICE 3   /* no HIP device / HIP runtime error                                           */
#define CPT_ERR_UNSUPPORTED 4 /* physics branch of the reference that this backend does not implement (yet)  */

/* gauges, as enum possible_gauges in source/perturbations.h */
#define CPT_GAUGE_NEWTONIAN 0
#define CPT_GAUGE_SYNCHRONOUS 1

/* tight_coupling_approximation values, as enum tca_method in source/perturbations.h */
#define CPT_TCA_FIRST_ORDER_MB 0
#define CPT_TCA_FIRST_ORDER_CAMB 1
#define CPT_TCA_FIRST_ORDER_CLASS 2
#define CPT_TCA_SECOND_ORDER_CRS 3
#define CPT_TCA_SECOND_ORDER_CLASS 4
#define CPT_TCA_COMPROMISE_CLASS 5

/* radiation_streaming_approximation (enum rsa_method): 0 rsa_null, 1 rsa_MD, 2 rsa_MD_with_reio, 3 rsa_none */
#define CPT_RSA_NULL 0
#define CPT_RSA_MD 1
#define CPT_RSA_MD_WITH_REIO 2
#define CPT_RSA_NONE 3
/* ur_fluid_approximation (enum ufa_method): 0 ufa_mb, 1 ufa_hu, 2 ufa_CLASS, 3 ufa_none */
#define CPT_UFA_MB 0
#define CPT_UFA_HU 1
#define CPT_UFA_CLASS 2
#define CPT_UFA_NONE 3
/* ncdm_fluid_approximation (enum ncdmfa_method, source/perturbations.h:55) */
#define CPT_NCDMFA_MB 0
#define CPT_NCDMFA_HU 1
#define CPT_NCDMFA_CLASS 2
#define CPT_NCDMFA_NONE 3
/* tensor_method (enum tensor_methods, source/perturbations.h:56) */
#define CPT_TM_PHOTONS_ONLY 0
#define CPT_TM_MASSLESS_APPROXIMATION 1
#define CPT_TM_EXACT 2
#define CPT_MAX_NCDM 3       /* species per handle */
#define CPT_MAX_Q_NCDM 8     /* momentum bins per species */

/* Flat POD with the physics flags / derived scalars / precision parameters the path reads
 * (struct background / thermo / perturbs / precision / transfers of the reference).  */
enum { CPT_MODE_SCALARS = 0, CPT_MODE_TENSORS = 1 };
enum { CPT_IC_AD = 0, CPT_IC_BI = 1, CPT_IC_CDI = 2, CPT_IC_NID = 3, CPT_IC_NIV = 4 };

/* density / velocity transfer sources (PerturbationsModule::index_tp_delta_tot_ ... index_tp_psi_, perturbations_module.h:93-122) */
enum { CPT_TK_DELTA_TOT = 0, CPT_TK_DELTA_G, CPT_TK_DELTA_B, CPT_TK_DELTA_CDM, CPT_TK_DELTA_UR,
       CPT_TK_THETA_TOT, CPT_TK_THETA_G, CPT_TK_THETA_B, CPT_TK_THETA_CDM, CPT_TK_THETA_UR, CPT_TK_PHI, CPT_TK_PSI, CPT_NTK };

typedef struct cpt_config {
  /* --- background (source/background.h) --- */
  double H0;      /* [1/Mpc] */
  double K;       /* curvature; only K == 0 is implemented                                     */
  int sgnK;
  int has_cdm, has_ur, has_ncdm, has_fld, has_curvature;
  double T_cmb, a_today;
  /* --- thermodynamics scalars (source/thermodynamics_module.h) --- */
  double YHe, n_e;              /* used by the z > z_table_max analytic extrapolation, th.cpp:128-219 */
  double tau0;                  /* conformal_age_                                                     */
  double tau_rec;               /* tau_rec_                                                           */
  double tau_free_streaming;    /* tau_free_streaming_                                                */
  double tau_cut;               /* tau_cut_                                                           */
  double angular_rescaling;     /* angular_rescaling_                                                 */
  /* --- perturbation flags (source/perturbations.h:105-222) --- */
  int gauge;
  int switch_sw, switch_eisw, switch_lisw, switch_dop, switch_pol;
  double eisw_lisw_split_z;
  double three_ceff2_ur, three_cvis2_ur;
  /* source types to produce and their slot in the sources table (reference order: t2,p,t0,t1,delta_m,phi+psi);
     -1 = not requested (pm.cpp:1107-1150) */
  int tp_size;
  int index_tp_t0, index_tp_t1, index_tp_t2, index_tp_p, index_tp_delta_m, index_tp_phi_plus_psi;
  /* --- precision (include/precisions.h) --- */
  double start_small_k_at_tau_c_over_tau_h, start_large_k_at_tau_h_over_tau_k;
  double tight_coupling_trigger_tau_c_over_tau_h, tight_coupling_trigger_tau_c_over_tau_k;
  int tight_coupling_approximation;
  int radiation_streaming_approximation;
  double radiation_streaming_trigger_tau_over_tau_k;
  int ur_fluid_approximation;
  double ur_fluid_trigger_tau_over_tau_k;
  int l_max_g, l_max_pol_g, l_max_ur;
  double curvature_ini;
  double tol_perturb_integration, tol_tau_approx, smallest_allowed_variation;
  /* --- transfer (source/transfer.h, include/precisions.h:335-395) --- */
  /* transfer types and their slot in the transfer table (reference order: t2,e,t0,t1,lcmb); -1 = absent */
  int tt_size;
  int index_tt_t0, index_tt_t1, index_tt_t2, index_tt_e, index_tt_lcmb;
  double lcmb_rescale, lcmb_tilt, lcmb_pivot;
  double hyper_x_min, hyper_sampling_flat, hyper_phi_min_abs;
  double transfer_neglect_delta_k_S_t0, transfer_neglect_delta_k_S_t1, transfer_neglect_delta_k_S_t2,
      transfer_neglect_delta_k_S_e;
  double transfer_neglect_late_source;
  double l_switch_limber;
  /* --- initial condition of the (single) mode integrated by this handle (pm.cpp:4846-5083); appended last so that
   *     zero-initialised older callers get the adiabatic mode --- */
  int ic;               /* CPT_IC_AD (0), CPT_IC_BI, CPT_IC_CDI, CPT_IC_NID, CPT_IC_NIV */
  double entropy_ini;   /* ppr->entropy_ini (isocurvature normalisation; default 1) */
  /* --- tensor modes (one mode per handle; pm.cpp:3519-3586, 9045-9215, 7243-7280; tm.cpp:3494-3529) --- */
  int mode;                   /* CPT_MODE_SCALARS (0) or CPT_MODE_TENSORS (1) */
  int l_max_g_ten, l_max_pol_g_ten;   /* precision: multipoles of the tensor photon hierarchies (default 5, 5) */
  double gw_ini;              /* precision: initial gravitational-wave amplitude (default 1) */
  int evolve_tensor_ur;       /* PerturbationsModule::evolve_tensor_ur_: massless neutrinos source the gravitational waves */
  int index_tt_b;             /* transfer slot of the B-mode polarisation type (tensors; -1 = absent) */
  double transfer_neglect_delta_k_T_t2, transfer_neglect_delta_k_T_e, transfer_neglect_delta_k_T_b;
  /* --- non-flat space: per-q hyperspherical tables (tm.cpp:3777-3887; include/precisions.h:341-346) --- */
  double hyper_sampling_curved_low_nu, hyper_sampling_curved_high_nu, hyper_nu_sampling_step, hyper_flat_approximation_nu;
  /* --- non-cold dark matter: massive neutrinos etc. (read only when has_ncdm != 0; pm.cpp:3441-3466, 8725-8879, 6317-6432) --- */
  int N_ncdm;                         /* number of species, <= CPT_MAX_NCDM; momentum grids travel in cpt_tables          */
  int l_max_ncdm;                     /* precision: multipoles per momentum bin (default 17)                               */
  int ncdm_fluid_approximation;       /* CPT_NCDMFA_MB / HU / CLASS / NONE (source/perturbations.h:55)                     */
  double ncdm_fluid_trigger_tau_over_tau_k;
  double tol_ncdm_initial_w;          /* start-time condition |p/rho - 1/3| of every species (pm.cpp:2574-2603)           */
  int index_tp_delta_cb;              /* slot of the cdm+baryon density source (requested together with delta_m when ncdm is
                                         present, pm.cpp:996); -1 = absent                                                 */
  int tensor_method;                  /* CPT_TM_PHOTONS_ONLY / MASSLESS_APPROXIMATION / EXACT (pm.cpp:590-611): with ncdm and
                                         the massless approximation the tensor ur hierarchy carries rho_ur + 3 sum p_ncdm  */
  int has_transfers;                  /* 0: the array below is ignored (a zero-initialised struct asks for none)               */
  int index_tp_transfer[CPT_NTK];     /* slots of the density / velocity transfer sources (output = mTk, vTk: pm.cpp:1000-1050,
                                         6930-7200), indexed by CPT_TK_*; -1 = absent.  Scalar modes.                          */
  int index_tp_delta_ncdm1, index_tp_theta_ncdm1;   /* first of N_ncdm consecutive slots each: delta / theta of every non-cold species
                                         (pm.cpp:1122, 1137); read with has_transfers, -1 = absent                            */
} cpt_config;

/* Spline tables the RHS samples (all HOST pointers, row-major [n_lines][n_columns], copied to HBM by cpt_create):
 *   background:     BackgroundModule::tau_table_, background_table_, d2background_dtau2_table_
 *                   (source/background_module.h:166-178), columns located by the index_bg_* map;
 *   thermodynamics: ThermodynamicsModule::z_table_, thermodynamics_table_, d2thermodynamics_dz2_table_
 *                   (source/thermodynamics_module.h:120-125), columns located by the index_th_* map.          */
typedef struct cpt_tables {
  int bt_size, bg_size;
  const double* tau_table;
  const double* background_table;
  const double* d2background_dtau2_table;
  int index_bg_a, index_bg_H, index_bg_H_prime, index_bg_rho_g, index_bg_rho_b, index_bg_rho_cdm, index_bg_rho_ur;
  int tt_size, th_size;
  const double* z_table;
  const double* thermodynamics_table;
  const double* d2thermodynamics_dz2_table;
  int index_th_xe, index_th_dkappa, index_th_tau_d, index_th_ddkappa, index_th_dddkappa, index_th_exp_m_kappa,
      index_th_g, index_th_dg, index_th_cb2;
  int index_th_rate; /* only read by the host-side time sampling (include/cpt_host.h), never by the kernels */
  /* non-cold dark matter (read only when cfg->has_ncdm): background columns of the first species (others contiguous,
   * source/background_module.h:55-57) and the momentum grids NonColdDarkMatter::q_ncdm_, w_ncdm_, dlnf0_dlnq_ncdm_, M_ncdm_,
   * factor_ncdm_ (tools/non_cold_dark_matter.h:70-79), host pointers */
  int index_bg_rho_ncdm1, index_bg_p_ncdm1, index_bg_pseudo_p_ncdm1;
  int q_size_ncdm[CPT_MAX_NCDM];
  const double* q_ncdm[CPT_MAX_NCDM];
  const double* w_ncdm[CPT_MAX_NCDM];
  const double* dlnf0_dlnq_ncdm[CPT_MAX_NCDM];
  double M_ncdm[CPT_MAX_NCDM], factor_ncdm[CPT_MAX_NCDM];
} cpt_tables;

/* per-k-mode work counters = the evolver's stepstat[6] (tools/evolver_ndf15.cpp:29-37) summed over regimes */
typedef struct cpt_stepstat {
  int steps, failed, fevals, jacobians, factorisations, solves;
  int n_regimes;             /* number of constant-approximation intervals integrated     */
  double tau_ini;            /* start time found by the bisection of pm.cpp:2545-2635      */
} cpt_stepstat;

typedef struct cpt_handle cpt_handle;

/* Create a handle: validates cfg (unsupported physics -> CPT_ERR_UNSUPPORTED), uploads the tables. */
int cpt_create(const cpt_config* cfg, const cpt_tables* tabs, cpt_handle** out);
void cpt_destroy(cpt_handle* h);
/* message of the last failing call on this handle ("" if none); valid until the next call on the handle */
const char* cpt_last_error(const cpt_handle* h);
/* message for a failure that happened before a handle existed (cpt_create) */
const char* cpt_create_error(void);

/* Hot path A: integrate nk scalar adiabatic k-modes (one wavefront per mode) and sample the source functions.
 * Replaces the task body of PerturbationsModule::perturb_init (pm.cpp:686-707 -> perturb_solve).
 *   k[nk], tau_sampling[ntau]         host arrays (reference: k_[md], tau_sampling_)
 *   sources_dev                       device, reference layout sources_[ic*tp+tp][index_tau*k_size+index_k]
 *                                     flattened as [tp_size][ntau][nk]; may be NULL (results stay resident in the
 *                                     handle, k-major, for cpt_transfer_batch(..., NULL, ...))
 *   stats[nk], status[nk]             host, may be NULL                                                       */
int cpt_perturb_solve_batch(cpt_handle* h, const double* k, int nk, const double* tau_sampling, int ntau,
                            double* sources_dev, cpt_stepstat* stats, int* status);

/* Hot path B: all Delta_l^X(q) for nq wavenumbers and nl multipoles (flat space).
 * Replaces the task body of TransferModule::transfer_init (tm.cpp:291-315 -> transfer_compute_for_each_q),
 * including the k-spline of the sources (tm.cpp:604-639) and the flat Bessel table (tm.cpp:246-262).
 *   sources_dev    device [tp_size][ntau][nk] (reference layout) or NULL = use the sources left resident by the
 *                  last cpt_perturb_solve_batch on this handle
 *   k[nk], tau_sampling[ntau], q[nq], l[nl]     host arrays (reference: ppt k_, tau_sampling_, ptr q_, l_)
 *   k_size_cl      number of leading k values used for C_l's (reference k_size_cl_[md]); q beyond k[k_size_cl-1]
 *                  give zero transfer (tm.cpp:1541)
 *   transfer_dev   device, reference layout transfer_[((ic*tt+tt)*l_size+l)*q_size+q] = [tt_size][nl][nq]        */
int cpt_transfer_batch(cpt_handle* h, const double* sources_dev, const double* k, int nk, int k_size_cl,
                       const double* tau_sampling, int ntau, const double* q, int nq, const int* l, int nl,
                       double* transfer_dev);

/* ---- "next" rows of the scope table (SURVEY S8f-2,3): what turns the path's outputs into observables ---- */
/* analytic primordial spectrum P_R(k) = A_s exp((n_s-1) ln(k/k_pivot) + alpha_s/2 ln^2(k/k_pivot))
 * (source/primordial_module.cpp:911-925) and the slots of the C_l types in the output table (reference order
 * tt,ee,te,bb,pp,tp,ep; -1 = absent; source/spectra_module.h:47-53) */
typedef struct cpt_spectra_params {
  double A_s, n_s, alpha_s, k_pivot;
  int ct_size;
  int index_ct_tt, index_ct_ee, index_ct_te, index_ct_bb, index_ct_pp, index_ct_tp, index_ct_ep;
} cpt_spectra_params;

/* C_l = 4 pi int dk/k P_R(k) Delta_l^X(q) Delta_l^Y(q): integrand splined in q and integrated
 * (SpectraModule::spectra_compute_cl, source/spectra_module.cpp:958-1353; flat space, one initial condition).
 *   transfer_dev  device [tt_size][nl][nq] as produced by cpt_transfer_batch;  q[nq] host
 *   cl_dev        device [nl][ct_size] = cl_[md][(l*ic_ic+0)*ct_size+ct]                                        */
int cpt_cl_batch(cpt_handle* h, const cpt_spectra_params* sp, const double* transfer_dev, const double* q, int nq, int nl,
                 double* cl_dev);
/* the same for a PAIR of scalar initial conditions (the ic1 != ic2 terms of the ic x ic loop, spectra_module.cpp:958-1353):
 * C_l^{XY,(12)} = 4 pi int dk/k P_12(k) 1/2 [Delta_l^{X,1} Delta_l^{Y,2} + Delta_l^{Y,1} Delta_l^{X,2}], with sp->A_s, n_s, alpha_s holding the
 * amplitude (may be negative: anticorrelation), tilt and running of the cross spectrum (primordial_module.cpp:770-890).  transfer1_dev and
 * transfer2_dev: device tables [tt_size][nl][nq] of the two initial conditions (two handles of one cosmology; the call may be made
 * on either).  The total is sum_i C^(ii) + 2 sum_{i<j} C^(ij) (spectra_module.cpp cl_output). */
int cpt_cl_cross_batch(cpt_handle* h, const cpt_spectra_params* sp, const double* transfer1_dev, const double* transfer2_dev, const double* q,
                       int nq, int nl, double* cl_dev);
/* linear matter power spectrum today P(k) = 2 pi^2/k^3 delta_m(k,tau0)^2 P_R(k) from the sources resident in the handle
 * (NonlinearModule::nonlinear_pk_linear, source/nonlinear_module.cpp:1886-2040); pk_dev device [nk]               */
int cpt_pk_linear(cpt_handle* h, const cpt_spectra_params* sp, const double* k, int nk, double* pk_dev);
/* sigma(R): rms of the linear density field in spheres of radius R [Mpc] at z = 0, e.g. sigma8 = sigma(8/h)
 * (NonlinearModule::nonlinear_sigmas_at_z / nonlinear_sigmas, source/nonlinear_module.cpp:926-963, 2041-2180;
 * k_per_decade: ppr->sigma_k_per_decade, default 80).  Needs resident sources with delta_m like cpt_pk_linear. */
int cpt_sigma(cpt_handle* h, const cpt_spectra_params* sp, const double* k, int nk, double R, double k_per_decade, double* sigma);
/* sigma(R) of ANY tabulated linear spectrum P(k) > 0 on an increasing k grid (host arrays; no device work, no handle): the rule of
 * cpt_sigma applied to a spectrum assembled by the caller, e.g. the total over several correlated initial conditions. */
int cpt_sigma_of_pk(const double* k, const double* pk, int nk, double R, double k_per_decade, double* sigma);
/* the same two for baryons + cold dark matter only, P_cb(k) and sigma_cb(R), from the delta_cb source: defined when non-cold species
 * are present (NonlinearModule has_pk_cb_ / index_pk_cb_, source/nonlinear_module.cpp:1749-1760; classy pk_cb, sigma8_cb)           */
int cpt_pk_cb_linear(cpt_handle* h, const cpt_spectra_params* sp, const double* k, int nk, double* pk_dev);
int cpt_sigma_cb(cpt_handle* h, const cpt_spectra_params* sp, const double* k, int nk, double R, double k_per_decade, double* sigma);

/* linear P(k, z) and sigma(R, z) at 0 < z <= z_max_pk from the resident sources.  ln P(k, tau_i) at the last `ln_tau_size` sampling times
 * (PerturbationsModule::ln_tau_ / ln_tau_size_, pm.cpp:1554-1592; cpt_host_ln_tau_size in cpt_host.h) is splined in ln tau with estimated end
 * derivatives and evaluated at ln tau_z on the device (NonlinearModule::nonlinear_pk_at_z, nonlinear_module.cpp:81-283, :1136-1190;
 * nonlinear_sigmas_at_z :927-963).  tau_z: conformal time of the redshift (BackgroundModule::background_tau_of_z, cpt_host_background_tau_of_z);
 * cb != 0: baryons + cold dark matter only.  The sources must be those of the handle's last cpt_perturb_solve_batch / cpt_step.  pk_dev device [nk]. */
int cpt_pk_at_tau(cpt_handle* h, const cpt_spectra_params* sp, const double* k, int nk, int ln_tau_size, double tau_z, int cb, double* pk_dev);
int cpt_sigma_at_tau(cpt_handle* h, const cpt_spectra_params* sp, const double* k, int nk, int ln_tau_size, double tau_z, int cb, double R, double k_per_decade,
                     double* sigma);

/* ---- CMB lensing of the C_l's (LensingModule::lensing_init, source/lensing_module.cpp:149-854) ----
 * precision parameters of include/precisions.h:492-495 plus SpectraModule::l_max_tot_ */
typedef struct cpt_lensing_params {
  int l_unlensed_max;        /* last multipole of the unlensed spectra (l_max_scalars + delta_l_max when lensing = yes) */
  int delta_l_max;           /* lensed spectra are returned up to l_unlensed_max - delta_l_max (default 500) */
  int accurate_lensing;      /* 0: Riemann sum of the correlation-function difference on (0, pi/16] (default); 1: Gauss-Legendre */
  int num_mu_minus_lmax;     /* accurate mode: number of nodes - l_max (default 70) */
  double tol_gauss_legendre; /* accurate mode: tolerance on the Legendre roots (<= 0: 1e-14) */
} cpt_lensing_params;
/* number of rows of the lensed table for this l grid (LensingModule::lensing_indices, lensing_module.cpp:983-993) */
int cpt_lensing_l_size(const int* l, int nl, const cpt_lensing_params* lp);
/* cl_dev device [nl][ct_size] (unlensed, from cpt_cl_batch) on the l grid l[nl] (host)
 *   -> cl_lensed_dev device [cpt_lensing_l_size][ct_size] = LensingModule::cl_lens_: TT, TE, EE, BB lensed, the other
 *      types copied.  Replaces the body of LensingModule::lensing_init.                                               */
int cpt_lensing_batch(cpt_handle* h, const cpt_spectra_params* sp, const cpt_lensing_params* lp, const int* l, int nl,
                      const double* cl_dev, double* cl_lensed_dev);

/* ---- one whole pass for one cosmology, fused: cpt_perturb_solve_batch -> cpt_transfer_batch (resident sources) ->
 * cpt_cl_batch -> [cpt_lensing_batch] -> [cpt_pk_linear], enqueued back to back on the handle's stream with ONE
 * synchronisation at the end.  This is what the reference does between the constructors of PerturbationsModule and
 * LensingModule (source/cosmology.cpp:16-86) for the path of this backend; the separate entry points above remain for callers
 * that need the intermediate tables on the host side in between.  Grids that did not change since the last call on the
 * handle are not validated, prepared or uploaded again.  All pointers as in the separate entry points. */
typedef struct cpt_step_io {
  const double* k; int nk; int k_size_cl;          /* host */
  const double* tau_sampling; int ntau;            /* host */
  const double* q; int nq;                         /* host */
  const int* l; int nl;                            /* host */
  const cpt_spectra_params* sp;
  const cpt_lensing_params* lp;                    /* NULL: no lensing */
  double* transfer_dev;                            /* device [tt_size][nl][nq]                         */
  double* cl_dev;                                  /* device [nl][ct_size]                             */
  double* cl_lensed_dev;                           /* device [cpt_lensing_l_size][ct_size] or NULL     */
  double* pk_dev;                                  /* device [nk] or NULL (needs the delta_m source)   */
  cpt_stepstat* stats; int* status;                /* host [nk], may be NULL                           */
} cpt_step_io;
int cpt_step(cpt_handle* h, const cpt_step_io* io);

/* ---- multi-GPU: one process per GPU, RCCL over xGMI on the handle's stream -------------------------------------
 * The path shards in two stages with one exchange after each (classpp_public_amd/csrc/cpt_comm.hip):
 *   rank r integrates k_all[r], k_all[r + W], ...  (cpt_perturb_solve_batch on that subset, sources_dev = NULL)
 *   cpt_allgather_sources        -> every rank holds the full k-major sources, resident
 *   rank r computes the multipoles l_all[r], l_all[r + W], ...  (cpt_transfer_batch with sources_dev = NULL and that l subset)
 *   cpt_gather_transfer          -> rank 0 holds transfer_[tt][nl_all][nq]   (or: cpt_cl_batch on the local rows, then cpt_gather_cl)
 * The reference has no counterpart (one process, a thread pool: pm.cpp:668-718, tm.cpp:287-318 are its two parallel loops).
 * Rendezvous is the caller's business: rank 0 obtains an id, every rank receives its CPT_COMM_ID_BYTES by any means (a file, MPI,
 * torch.distributed's store) and joins.  RCCL is bound at run time: librccl.so of the process (CPT_RCCL_PATH overrides). */
#define CPT_COMM_ID_BYTES 128
int cpt_comm_get_unique_id(void* id /* CPT_COMM_ID_BYTES bytes, host */);
int cpt_comm_init(cpt_handle* h, const void* id, int rank, int world);   /* collective over the `world` ranks */
int cpt_comm_destroy(cpt_handle* h);
/* nk_all: size of the full k grid.  On entry the handle holds the sources of this rank's shard (its last cpt_perturb_solve_batch);
 * on return those of all nk_all modes, in the order of k_all. */
int cpt_allgather_sources(cpt_handle* h, int nk_all, int ntau);
/* transfer_local_dev: device [tt_size][nl_local][nq] of this rank's multipoles; transfer_full_dev: device [tt_size][nl_all][nq],
 * written on rank 0 only (may be NULL elsewhere) */
int cpt_gather_transfer(cpt_handle* h, const double* transfer_local_dev, int nl_all, int nq, double* transfer_full_dev);
/* the same for the spectra: cl_local_dev device [nl_local][ct_size] (cpt_cl_batch on this rank's transfer rows) -> cl_full_dev device
 * [nl_all][ct_size] on rank 0.  The spectra are finished where the transfer functions are; only ct_size numbers per multipole travel. */
int cpt_gather_cl(cpt_handle* h, const double* cl_local_dev, int nl_all, int ct_size, double* cl_full_dev);

/* Device-side copy of the resident sources into the reference layout [tp_size][ntau][nk] (device pointer). */
int cpt_get_sources(cpt_handle* h, double* sources_dev);

/* ---- measurement hooks (used by bench.py; they time with hipEvents on the library's own stream) ---- */
/* milliseconds between two events recorded on the handle's stream during the last call that ran the stage, and the launch count:
 * stage 0 = the perturbation kernel, 1 = the line-of-sight kernel, 2 = the whole transfer stage (uploads, source spline, LOS),
 * 3 = a whole cpt_step, first to last kernel (wall time of the call minus this = host overhead) */
int cpt_last_kernel_ms(const cpt_handle* h, int stage, double* ms, int* launches);
/* work counters of the last transfer call: number of (q,l,type) integrals, of (q,l,type,tau) samples (the
 * reference evaluates each separately) and of fused (q,l,tau) samples (types sharing one Phi_l row) */
int cpt_last_transfer_work(const cpt_handle* h, long long* integrals, long long* type_samples,
                           long long* fused_samples);

/* ---- unit-test hooks: run single device functions so that parity tests can localise a discrepancy ---- */
/* background_at_tau + thermodynamics_at_z at n times -> out[n][7 + 9] (a,H,H',rho_g,rho_b,rho_cdm,rho_ur,
 * then xe,dkappa,tau_d,ddkappa,dddkappa,exp_m_kappa,g,dg,cb2) */
int cpt_dbg_lookup(cpt_handle* h, const double* tau, int n, double* out);
/* one RHS evaluation perturb_derivs(tau, y) for wavenumber k in the regime (tca_on, rsa_on, ufa_on):
 * y[neq] in the reference's index order for that regime (pm.cpp:3302-3481); returns dy[neq] and neq       */
int cpt_dbg_derivs(cpt_handle* h, double k, double tau, int tca_on, int rsa_on, int ufa_on, const double* y,
                   double* dy, int* neq);
/* solve (I - hg J(tau)) x = b for wavenumber k in the regime (tca_on, rsa_on, ufa_on) with the kernel's structured
 * factorisation; b[neq], x[neq] in the reference's index order (arrays of 64 doubles) */
int cpt_dbg_solve(cpt_handle* h, double k, double tau, int tca_on, int rsa_on, int ufa_on, double hg, const double* b,
                  double* x);
/* the packing / un-interleaving kernels of the two exchanges on caller-provided device buffers ([nbatch][n][ninner] blocks) */
int cpt_dbg_pad_rows(cpt_handle* h, const double* in_dev, double* out_dev, int nbatch, int n_local, int n_max, int ninner);
int cpt_dbg_uninterleave(cpt_handle* h, const double* blocks_dev, double* full_dev, int world, int nbatch, int n_max, int n_all, int ninner);
/* flat spherical Bessel table phi[nl][nx], dphi[nl][nx] and chi_at_phimin[nl] as built for (l, xmax) */
int cpt_dbg_bessel(cpt_handle* h, const int* l, int nl, double xmax, int* nx, double* phi, double* dphi,
                   double* chi_at_phimin, int cap_nx);

#ifdef __cplusplus
}
#endif
#endif /* CPT_H */
