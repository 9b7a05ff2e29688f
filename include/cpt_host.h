/* cpt_host.h -- host-side (CPU, C++ inside, C ABI outside) companions of the hot path: the sampling grids that
 * the reference's module constructors build before entering the parallel loops.  They are cheap (< 1 ms), run once
 * per cosmology and must be reproduced EXACTLY, because every downstream spline is defined on them (SURVEY S8a rows
 * A2, A3, B1):
 *   cpt_host_k_list        PerturbationsModule::perturb_get_k_list               pm.cpp:1628-1868 (scalars), :2007-2105 (tensors); flat, open, closed
 *   cpt_host_tau_sampling  PerturbationsModule::perturb_timesampling_for_sources pm.cpp:1247-1533
 *   cpt_host_l_list        TransferModule::transfer_get_l_list                   tm.cpp:694-790
 *   cpt_host_q_list        TransferModule::transfer_get_q_list (+ _k_list)       tm.cpp:884-1096 (flat, open, closed)
 * Built into classpp_public_amd/host/libcpt_host.so (g++, no HIP).  The C++ shim classes that mirror the reference's
 * PerturbationsModule / TransferModule data contract on top of libcpt.so are declared in include/cpt_modules.hpp.
 */
#ifndef CPT_HOST_H
#define CPT_HOST_H
#include "cpt.h"
#ifdef __cplusplus
extern "C" {
#endif

/* precision / physics parameters read by the grid builders (include/precisions.h:162-178, 202, 231, 335-376) */
typedef struct cpt_grid_params {
  /* k grid */
  double k_min_tau0, k_max_tau0_over_l_max, k_step_sub, k_step_super, k_step_transition, k_step_super_reduction,
      k_per_decade_for_pk, k_per_decade_for_bao, k_bao_center, k_bao_width;
  int has_cls, has_pk_matter, l_scalar_max;
  double k_max_for_pk;
  double rs_rec;          /* ThermodynamicsModule::rs_rec_   */
  double tau_ini_thermo;  /* ThermodynamicsModule::tau_ini_  */
  /* tau sampling */
  double start_sources_at_tau_c_over_tau_h, perturb_sampling_stepsize;
  /* l, q grids */
  double l_linstep, l_logstep, q_linstep, q_logstep_spline, q_logstep_open;
  /* appended: tensors (one mode per handle) and closed space */
  int l_tensor_max;
  double q_logstep_trapzd, q_numstep_transition;
  /* appended: P(k, z > 0).  Conformal time of ppt->z_max_pk (BackgroundModule::background_tau_of_z); 0 = z_max_pk is 0: sources kept for z = 0 only */
  double tau_of_z_max_pk;
} cpt_grid_params;

/* Every function returns CPT_OK or CPT_ERR_INVALID (message via cpt_host_error()); *_size are outputs; `cap` is the
 * capacity of the caller's array (CPT_ERR_INVALID if too small, with the needed size stored in *_size). */
int cpt_host_k_list(const cpt_config* cfg, const cpt_grid_params* g, double* k, int cap, int* k_size, int* k_size_cl,
                    int* k_size_cmb);
int cpt_host_tau_sampling(const cpt_config* cfg, const cpt_tables* tabs, const cpt_grid_params* g, double* tau, int cap,
                          int* tau_size);
/* length ln_tau_size_ of the tail of the sampling that covers 0 <= z <= z_max_pk with four more points for the spline (pm.cpp:1554-1592):
 * ln_tau_[i] = log(tau[tau_size - ln_tau_size + i]).  tau_of_z_max_pk <= 0 (z_max_pk = 0): 1. */
int cpt_host_ln_tau_size(const double* tau, int tau_size, double tau_of_z_max_pk, int* ln_tau_size);
int cpt_host_l_list(const cpt_config* cfg, const cpt_grid_params* g, int* l, int cap, int* l_size);
int cpt_host_q_list(const cpt_config* cfg, const cpt_grid_params* g, double k_min, double k_max_cl, double* q, int cap,
                    int* q_size);
const char* cpt_host_error(void);

/* ---- the tables the hot path consumes, computed on the host instead of being handed over ----------------
 * Background (BackgroundModule::background_solve_evolver, source/background_module.cpp:1326-1520 with background_functions
 * :263-610, background_initial_conditions :1521-1690, background_derivs :1934-2064, :2272-2344): flat / curved LambdaCDM with
 * massless neutrinos and non-cold species given their momentum sampling (fluids, scalar fields, decaying species:
 * CPT_ERR_UNSUPPORTED).  The table has the reference's layout for that content (21 + 4 N_ncdm columns; index map returned), so it can be handed to cpt_create unchanged.  */
typedef struct cpt_cosmo_params {
  /* struct background (source/background.h) */
  double H0;                        /* [1/Mpc] */
  double T_cmb, Omega0_g, Omega0_b, Omega0_cdm, Omega0_ur, Omega0_lambda, Omega0_k;
  double K; int sgnK;               /* K = -Omega0_k (a_today H0)^2 */
  double a_today;
  int has_cdm, has_ur, has_lambda, has_ncdm, has_fld, has_scf, has_dcdm, has_dr, has_idr, has_idm_dr;
  /* precision (include/precisions.h:12-38) */
  double a_ini_over_a_today_default, back_integration_stepsize, tol_initial_Omega_r, smallest_allowed_variation;
  /* non-cold species (has_ncdm): what NonColdDarkMatter holds after its own initialisation - the mass in units of the
   * temperature, the normalisation factor and the BACKGROUND momentum sampling q_ncdm_bg_, w_ncdm_bg_ (tools/non_cold_dark_matter.h:
   * 70-79, 120-122; the adaptive quadrature that chooses the nodes stays outside).  Host pointers, read during the call only. */
  int N_ncdm;
  int q_size_ncdm_bg[CPT_MAX_NCDM];
  const double* q_ncdm_bg[CPT_MAX_NCDM];
  const double* w_ncdm_bg[CPT_MAX_NCDM];
  double M_ncdm[CPT_MAX_NCDM], factor_ncdm[CPT_MAX_NCDM];
  double tol_ncdm_initial_w;
} cpt_cosmo_params;

typedef struct cpt_background {
  int bt_size, bg_size;
  double* tau_table;                /* [bt_size]            all arrays owned by this struct: cpt_host_background_free */
  double* z_table;                  /* [bt_size]  */
  double* d2tau_dz2_table;          /* [bt_size]  */
  double* background_table;         /* [bt_size][bg_size] */
  double* d2background_dtau2_table; /* [bt_size][bg_size] */
  int index_bg_a, index_bg_H, index_bg_H_prime, index_bg_rho_g, index_bg_rho_b, index_bg_rho_cdm, index_bg_rho_lambda,
      index_bg_rho_ur, index_bg_rho_tot, index_bg_p_tot, index_bg_p_tot_prime, index_bg_Omega_r, index_bg_rho_crit,
      index_bg_Omega_m, index_bg_conf_distance, index_bg_ang_distance, index_bg_lum_distance, index_bg_time, index_bg_rs,
      index_bg_D, index_bg_f;
  int index_bg_number_ncdm1, index_bg_rho_ncdm1, index_bg_p_ncdm1, index_bg_pseudo_p_ncdm1;   /* -1 without ncdm; species contiguous */
  double conformal_age, age, Neff, Omega0_m, Omega0_r, Omega0_de;
} cpt_background;

/* fills the precision members of `p` with the reference's defaults */
void cpt_host_cosmo_defaults(cpt_cosmo_params* p);
int cpt_host_background(const cpt_cosmo_params* p, cpt_background* out);
void cpt_host_background_free(cpt_background* bg);
/* conformal time at redshift z by spline interpolation in the table (BackgroundModule::background_tau_of_z, :211-255) */
int cpt_host_background_tau_of_z(const cpt_background* bg, double z, double* tau);
/* the same from the z and tau columns of any background table (e.g. the reference's BackgroundModule::z_table_, tau_table_), bt_size rows */
int cpt_host_tau_of_z_from_table(const double* z_table, const double* tau_table, int bt_size, double z, double* tau);

/* Thermodynamics (ThermodynamicsModule::thermodynamics_init, source/thermodynamics_module.cpp:293-1297): RECFAST 1.5 recombination
 * (:3335-3975, adaptive Cash-Karp integration tools/dei_rkck.c), reionization none / CAMB-like tanh with z_reio or tau_reio given
 * (:1893-1950, 2159-2320, 2668-2990), merged table and every derived column the hot path reads (first to third derivative of kappa,
 * exp(-kappa), the visibility g with its first two derivatives, tau_d, T_b, w_b, c_b^2, rate) with their spline second derivatives
 * in z, and the scalars z_rec, tau_rec, r_s(rec), r_a(rec), angular_rescaling, tau_free_streaming, tau_cut.  HyRec, energy
 * injection, the other reionization schemes and interacting dark matter are CPT_ERR_UNSUPPORTED.  The table has the reference's
 * 13-column layout. */
enum { CPT_REIO_NONE = 0, CPT_REIO_CAMB = 1 };
typedef struct cpt_thermo_params {
  double YHe;                        /* primordial helium fraction (the BBN table look-up stays outside) */
  int reio_parametrization;          /* CPT_REIO_NONE / CPT_REIO_CAMB */
  int reio_from_tau;                 /* 0: z_reio given, 1: tau_reio given (bisection on the optical depth, :2222-2318) */
  double z_reio, tau_reio;
  double reionization_exponent, reionization_width, helium_fullreio_redshift, helium_fullreio_width;
  /* precision (include/precisions.h:60-160, 257-300) */
  double recfast_z_initial; int recfast_Nz0; double tol_thermo_integration;
  int recfast_Heswitch; double recfast_fudge_He; int recfast_Hswitch; double recfast_fudge_H, recfast_delta_fudge_H, recfast_AGauss1,
      recfast_AGauss2, recfast_zGauss1, recfast_zGauss2, recfast_wGauss1, recfast_wGauss2, recfast_z_He_1, recfast_delta_z_He_1,
      recfast_z_He_2, recfast_delta_z_He_2, recfast_z_He_3, recfast_delta_z_He_3, recfast_x_He0_trigger, recfast_x_He0_trigger2,
      recfast_x_He0_trigger_delta, recfast_x_H0_trigger, recfast_x_H0_trigger2, recfast_x_H0_trigger_delta, recfast_H_frac;
  double reionization_z_start_max, reionization_sampling, reionization_optical_depth_tol, reionization_start_factor;
  int thermo_rate_smoothing_radius;
  double radiation_streaming_trigger_tau_c_over_tau, neglect_CMB_sources_below_visibility;
} cpt_thermo_params;

typedef struct cpt_thermo {
  int tt_size, th_size;
  double* z_table;                     /* [tt_size] ascending z          arrays owned: cpt_host_thermo_free */
  double* thermodynamics_table;        /* [tt_size][th_size] */
  double* d2thermodynamics_dz2_table;  /* [tt_size][th_size] */
  int index_th_xe, index_th_dkappa, index_th_tau_d, index_th_ddkappa, index_th_dddkappa, index_th_exp_m_kappa, index_th_g,
      index_th_dg, index_th_ddg, index_th_Tb, index_th_wb, index_th_cb2, index_th_rate;
  double tau_ini, YHe, n_e, z_rec, tau_rec, rs_rec, ra_rec, angular_rescaling, tau_free_streaming, tau_cut, z_reionization,
      tau_reionization, z_star, z_d;
} cpt_thermo;

void cpt_host_thermo_defaults(cpt_thermo_params* p);
int cpt_host_thermodynamics(const cpt_cosmo_params* cp, const cpt_thermo_params* tp, const cpt_background* bg, cpt_thermo* out);
void cpt_host_thermo_free(cpt_thermo* th);

/* Non-cold species (massive neutrinos etc.) from their physical parameters: what NonColdDarkMatter holds after its own initialisation
 * (tools/non_cold_dark_matter.cpp:202-790 with the sampling search of tools/quadrature.c:69-360) - the momentum samplings of the
 * perturbations (q_ncdm, w_ncdm, d ln f0 / d ln q) and of the background (q_ncdm_bg, w_ncdm_bg), the mass over the temperature M_ncdm,
 * the normalisation factor_ncdm and the mass <-> density relation.  Fermi-Dirac distribution with chemical potential (the reference's
 * built-in f0); distributions read from files and decaying species: CPT_ERR_UNSUPPORTED.  Feeds cpt_cosmo_params (background) and
 * cpt_tables (perturbations). */
typedef struct cpt_ncdm_params {
  int N_ncdm;
  double T_cmb, h;
  double m_ncdm_in_eV[CPT_MAX_NCDM];   /* 0 = not given: the mass follows from Omega0_ncdm                                  */
  double Omega0_ncdm[CPT_MAX_NCDM];    /* 0 = not given: the density follows from the mass; both given: deg is rescaled     */
  double T_ncdm[CPT_MAX_NCDM];         /* temperature in units of T_cmb (default 0.71611)                                   */
  double ksi_ncdm[CPT_MAX_NCDM];       /* chemical potential over the temperature (default 0)                               */
  double deg_ncdm[CPT_MAX_NCDM];       /* degeneracy (default 1)                                                            */
  double tol_ncdm, tol_ncdm_bg, tol_M_ncdm;   /* include/precisions.h:34-54; tol_ncdm = tol_ncdm_synchronous | _newtonian  */
} cpt_ncdm_params;
typedef struct cpt_ncdm {
  int N_ncdm;
  int q_size_ncdm[CPT_MAX_NCDM], q_size_ncdm_bg[CPT_MAX_NCDM];
  double* q_ncdm[CPT_MAX_NCDM];            /* arrays owned by this struct: cpt_host_ncdm_free */
  double* w_ncdm[CPT_MAX_NCDM];
  double* dlnf0_dlnq_ncdm[CPT_MAX_NCDM];
  double* q_ncdm_bg[CPT_MAX_NCDM];
  double* w_ncdm_bg[CPT_MAX_NCDM];
  double M_ncdm[CPT_MAX_NCDM], factor_ncdm[CPT_MAX_NCDM], Omega0_ncdm[CPT_MAX_NCDM], m_ncdm_in_eV[CPT_MAX_NCDM], deg_ncdm[CPT_MAX_NCDM];
  double Omega0_ncdm_tot;
} cpt_ncdm;
void cpt_host_ncdm_defaults(cpt_ncdm_params* p);
int cpt_host_ncdm(const cpt_ncdm_params* p, cpt_ncdm* out);
void cpt_host_ncdm_free(cpt_ncdm* o);

#ifdef __cplusplus
}
#endif
#endif
