// cpt_adapter.h -- the reference-side binding: fills a cpt::Inputs (include/cpt_modules.hpp) from the reference's own
// InputModule / BackgroundModule / ThermodynamicsModule.  This is the file a maintainer of the reference adds to its source tree
// (INTEGRATION.md); it includes the REFERENCE's headers and is therefore not part of this repository's build.  It is compiled against
// /root/reference/source/*.h by oracle/Makefile (target `adapter`) into oracle/_ref/adapter_check, which tests/test_adapter.py runs:
// the Inputs it produces must equal, field by field, what this repository derives from the fixtures dumped from the reference.
//
// Access to the three private table pointers (BackgroundModule::d2background_dtau2_table_, ThermodynamicsModule::
// d2thermodynamics_dz2_table_, the NonColdDarkMatter grids): the reference has no accessor for them today, so the translation unit that
// includes this header must be able to read them - either `friend cpt::Inputs MakeCptInputs(...)` declarations in the three classes
// (the two-line change INTEGRATION.md proposes), or, in the check driver, the `#define private public` trick the oracle already uses.
#pragma once
#include "cpt_modules.hpp"

#include "background_module.h"
#include "input_module.h"
#include "non_cold_dark_matter.h"
#include "thermodynamics_module.h"

// index maps of perturb_indices_of_perturbs (pm.cpp:1100-1150) and transfer_indices_of_transfers (tm.cpp:418-470) for the types this
// backend produces: common types first (t2, p | t2, e), then the scalar ones in the reference's order
inline void CptFillIndexMaps(const perturbs& pt, const background& ba, bool tensors, cpt_config& c) {
  const bool T = pt.has_cl_cmb_temperature, P = pt.has_cl_cmb_polarization, L = pt.has_cl_cmb_lensing_potential;
  const bool M = pt.has_pk_matter || pt.has_nl_corrections_based_on_delta_m;
  int i = 0;
  c.index_tp_t2 = T ? i++ : -1;
  c.index_tp_p = P ? i++ : -1;
  c.index_tp_t0 = c.index_tp_t1 = c.index_tp_delta_m = c.index_tp_delta_cb = c.index_tp_phi_plus_psi = -1;
  c.has_transfers = 0;
  c.index_tp_delta_ncdm1 = c.index_tp_theta_ncdm1 = -1;
  for (int t = 0; t < CPT_NTK; t++) c.index_tp_transfer[t] = -1;
  if (!tensors) {
    // (the reference's order of definition, pm.cpp:1107-1140: ..., delta_m, delta_cb, delta_tot, delta_g, delta_b, delta_cdm, delta_ur, theta_tot,
    //  theta_g, theta_b, theta_cdm, theta_ur, phi, phi+psi, psi)
    const bool D = pt.has_density_transfers, V = pt.has_velocity_transfers;
    if (T) { c.index_tp_t0 = i++; c.index_tp_t1 = i++; }
    if (M) { c.index_tp_delta_m = i++; if (ba.has_ncdm) c.index_tp_delta_cb = i++; }
    if (D) {
      c.index_tp_transfer[CPT_TK_DELTA_TOT] = i++; c.index_tp_transfer[CPT_TK_DELTA_G] = i++; c.index_tp_transfer[CPT_TK_DELTA_B] = i++;
      if (ba.has_cdm) c.index_tp_transfer[CPT_TK_DELTA_CDM] = i++;
      if (ba.has_ur) c.index_tp_transfer[CPT_TK_DELTA_UR] = i++;
      if (ba.has_ncdm) { c.index_tp_delta_ncdm1 = i; i += ba.N_ncdm; }
    }
    if (V) {
      c.index_tp_transfer[CPT_TK_THETA_TOT] = i++; c.index_tp_transfer[CPT_TK_THETA_G] = i++; c.index_tp_transfer[CPT_TK_THETA_B] = i++;
      if (ba.has_cdm && pt.gauge != synchronous) c.index_tp_transfer[CPT_TK_THETA_CDM] = i++;
      if (ba.has_ur) c.index_tp_transfer[CPT_TK_THETA_UR] = i++;
      if (ba.has_ncdm) { c.index_tp_theta_ncdm1 = i; i += ba.N_ncdm; }
    }
    if (D) c.index_tp_transfer[CPT_TK_PHI] = i++;
    if (L) c.index_tp_phi_plus_psi = i++;
    if (D) c.index_tp_transfer[CPT_TK_PSI] = i++;
    c.has_transfers = (D || V) ? 1 : 0;
  }
  c.tp_size = i;
  i = 0;
  c.index_tt_t2 = T ? i++ : -1;
  c.index_tt_e = P ? i++ : -1;
  c.index_tt_t0 = c.index_tt_t1 = c.index_tt_lcmb = c.index_tt_b = -1;
  if (!tensors) {
    if (T) { c.index_tt_t0 = i++; c.index_tt_t1 = i++; }
    if (L) c.index_tt_lcmb = i++;
  } else if (P) c.index_tt_b = i++;
  c.tt_size = pt.has_cls ? i : 0;
}

inline cpt::Inputs MakeCptInputs(const InputModule& in, const BackgroundModule& bg, const ThermodynamicsModule& th) {
  const precision& pr = in.precision_;
  const background& ba = in.background_;
  const perturbs& pt = in.perturbations_;
  const transfers& tr = in.transfers_;
  cpt::Inputs x{};
  cpt_config& c = x.config;
  // ---- background / thermodynamics scalars ----
  c.H0 = ba.H0; c.K = ba.K; c.sgnK = ba.sgnK; c.T_cmb = ba.T_cmb; c.a_today = ba.a_today;
  c.has_cdm = ba.has_cdm; c.has_ur = ba.has_ur; c.has_ncdm = ba.has_ncdm; c.has_fld = ba.has_fld; c.has_curvature = ba.has_curvature;
  c.YHe = th.YHe_; c.n_e = th.n_e_; c.tau0 = bg.conformal_age_; c.tau_rec = th.tau_rec_;
  c.tau_free_streaming = th.tau_free_streaming_; c.tau_cut = th.tau_cut_; c.angular_rescaling = th.angular_rescaling_;
  // ---- perturbation flags ----
  c.gauge = (int)pt.gauge;
  c.switch_sw = pt.switch_sw; c.switch_eisw = pt.switch_eisw; c.switch_lisw = pt.switch_lisw; c.switch_dop = pt.switch_dop; c.switch_pol = pt.switch_pol;
  c.eisw_lisw_split_z = pt.eisw_lisw_split_z; c.three_ceff2_ur = pt.three_ceff2_ur; c.three_cvis2_ur = pt.three_cvis2_ur;
  // ---- precision ----
  c.start_small_k_at_tau_c_over_tau_h = pr.start_small_k_at_tau_c_over_tau_h; c.start_large_k_at_tau_h_over_tau_k = pr.start_large_k_at_tau_h_over_tau_k;
  c.tight_coupling_trigger_tau_c_over_tau_h = pr.tight_coupling_trigger_tau_c_over_tau_h;
  c.tight_coupling_trigger_tau_c_over_tau_k = pr.tight_coupling_trigger_tau_c_over_tau_k;
  c.tight_coupling_approximation = pr.tight_coupling_approximation;
  c.radiation_streaming_approximation = pr.radiation_streaming_approximation;
  c.radiation_streaming_trigger_tau_over_tau_k = pr.radiation_streaming_trigger_tau_over_tau_k;
  c.ur_fluid_approximation = pr.ur_fluid_approximation; c.ur_fluid_trigger_tau_over_tau_k = pr.ur_fluid_trigger_tau_over_tau_k;
  c.l_max_g = pr.l_max_g; c.l_max_pol_g = pr.l_max_pol_g; c.l_max_ur = pr.l_max_ur;
  c.curvature_ini = pr.curvature_ini; c.tol_perturb_integration = pr.tol_perturb_integration; c.tol_tau_approx = pr.tol_tau_approx;
  c.smallest_allowed_variation = pr.smallest_allowed_variation;
  c.lcmb_rescale = tr.lcmb_rescale; c.lcmb_tilt = tr.lcmb_tilt; c.lcmb_pivot = tr.lcmb_pivot;
  c.hyper_x_min = pr.hyper_x_min; c.hyper_sampling_flat = pr.hyper_sampling_flat; c.hyper_phi_min_abs = pr.hyper_phi_min_abs;
  c.transfer_neglect_delta_k_S_t0 = pr.transfer_neglect_delta_k_S_t0; c.transfer_neglect_delta_k_S_t1 = pr.transfer_neglect_delta_k_S_t1;
  c.transfer_neglect_delta_k_S_t2 = pr.transfer_neglect_delta_k_S_t2; c.transfer_neglect_delta_k_S_e = pr.transfer_neglect_delta_k_S_e;
  c.transfer_neglect_late_source = pr.transfer_neglect_late_source; c.l_switch_limber = pr.l_switch_limber;
  c.entropy_ini = pr.entropy_ini;
  c.l_max_g_ten = pr.l_max_g_ten; c.l_max_pol_g_ten = pr.l_max_pol_g_ten; c.gw_ini = pr.gw_ini;
  c.transfer_neglect_delta_k_T_t2 = pr.transfer_neglect_delta_k_T_t2; c.transfer_neglect_delta_k_T_e = pr.transfer_neglect_delta_k_T_e;
  c.transfer_neglect_delta_k_T_b = pr.transfer_neglect_delta_k_T_b;
  c.hyper_sampling_curved_low_nu = pr.hyper_sampling_curved_low_nu; c.hyper_sampling_curved_high_nu = pr.hyper_sampling_curved_high_nu;
  c.hyper_nu_sampling_step = pr.hyper_nu_sampling_step; c.hyper_flat_approximation_nu = pr.hyper_flat_approximation_nu;
  c.N_ncdm = ba.has_ncdm ? ba.N_ncdm : 0;
  c.l_max_ncdm = pr.l_max_ncdm; c.ncdm_fluid_approximation = pr.ncdm_fluid_approximation;
  c.ncdm_fluid_trigger_tau_over_tau_k = pr.ncdm_fluid_trigger_tau_over_tau_k; c.tol_ncdm_initial_w = pr.tol_ncdm_initial_w;
  c.tensor_method = (int)pt.tensor_method;
  // ---- modes and initial conditions (pm.cpp:590-611, 1153-1170) ----
  const bool tensors_only = !pt.has_scalars && pt.has_tensors;
  c.mode = tensors_only ? CPT_MODE_TENSORS : CPT_MODE_SCALARS;
  // (pm.cpp:590-611) massless neutrinos - or, in the massless approximation, massive ones - source the gravitational waves
  const int evolve_tensor_ur = pt.has_tensors && ((pt.tensor_method == tm_massless_approximation && (ba.has_ur || ba.has_ncdm)) ||
                                                  (pt.tensor_method == tm_exact && ba.has_ur));
  c.evolve_tensor_ur = tensors_only ? evolve_tensor_ur : 0;
  CptFillIndexMaps(pt, ba, tensors_only, c);
  x.n_ic = 0;
  if (pt.has_scalars) {
    if (pt.has_ad) x.ic[x.n_ic++] = CPT_IC_AD;
    if (pt.has_bi) x.ic[x.n_ic++] = CPT_IC_BI;
    if (pt.has_cdi) x.ic[x.n_ic++] = CPT_IC_CDI;
    if (pt.has_nid) x.ic[x.n_ic++] = CPT_IC_NID;
    if (pt.has_niv) x.ic[x.n_ic++] = CPT_IC_NIV;
  }
  if (x.n_ic == 0) x.n_ic = 1;
  c.ic = pt.has_scalars ? x.ic[0] : CPT_IC_AD;
  x.with_tensors = pt.has_scalars && pt.has_tensors;
  if (x.with_tensors) {
    x.config_tensors = c;
    x.config_tensors.mode = CPT_MODE_TENSORS; x.config_tensors.ic = CPT_IC_AD; x.config_tensors.evolve_tensor_ur = evolve_tensor_ur;
    CptFillIndexMaps(pt, ba, true, x.config_tensors);
  }
  // ---- spline tables: pointers into the reference modules' own arrays (only read while the shim modules are constructed) ----
  cpt_tables& t = x.tables;
  t.bt_size = bg.bt_size_; t.bg_size = bg.bg_size_; t.tau_table = bg.tau_table_; t.background_table = bg.background_table_;
  t.d2background_dtau2_table = bg.d2background_dtau2_table_;
  t.index_bg_a = bg.index_bg_a_; t.index_bg_H = bg.index_bg_H_; t.index_bg_H_prime = bg.index_bg_H_prime_; t.index_bg_rho_g = bg.index_bg_rho_g_;
  t.index_bg_rho_b = bg.index_bg_rho_b_; t.index_bg_rho_cdm = ba.has_cdm ? bg.index_bg_rho_cdm_ : -1; t.index_bg_rho_ur = ba.has_ur ? bg.index_bg_rho_ur_ : -1;
  t.tt_size = th.tt_size_; t.th_size = th.th_size_; t.z_table = th.z_table_; t.thermodynamics_table = th.thermodynamics_table_;
  t.d2thermodynamics_dz2_table = th.d2thermodynamics_dz2_table_;
  t.index_th_xe = th.index_th_xe_; t.index_th_dkappa = th.index_th_dkappa_; t.index_th_tau_d = th.index_th_tau_d_;
  t.index_th_ddkappa = th.index_th_ddkappa_; t.index_th_dddkappa = th.index_th_dddkappa_; t.index_th_exp_m_kappa = th.index_th_exp_m_kappa_;
  t.index_th_g = th.index_th_g_; t.index_th_dg = th.index_th_dg_; t.index_th_cb2 = th.index_th_cb2_; t.index_th_rate = th.index_th_rate_;
  t.index_bg_rho_ncdm1 = t.index_bg_p_ncdm1 = t.index_bg_pseudo_p_ncdm1 = -1;
  if (ba.has_ncdm) {   // massive neutrinos: NonColdDarkMatter (tools/non_cold_dark_matter.h:70-79)
    const NonColdDarkMatter& nc = *in.ncdm_;
    t.index_bg_rho_ncdm1 = bg.index_bg_rho_ncdm1_; t.index_bg_p_ncdm1 = bg.index_bg_p_ncdm1_; t.index_bg_pseudo_p_ncdm1 = bg.index_bg_pseudo_p_ncdm1_;
    for (int n = 0; n < nc.N_ncdm_ && n < CPT_MAX_NCDM; n++) {
      t.q_size_ncdm[n] = nc.q_size_ncdm_[n]; t.q_ncdm[n] = nc.q_ncdm_[n]; t.w_ncdm[n] = nc.w_ncdm_[n];
      t.dlnf0_dlnq_ncdm[n] = nc.dlnf0_dlnq_ncdm_[n]; t.M_ncdm[n] = nc.M_ncdm_[n]; t.factor_ncdm[n] = nc.factor_ncdm_[n];
    }
  }
  // ---- sampling grids (include/cpt_host.h) ----
  cpt_grid_params& g = x.grid;
  g.k_min_tau0 = pr.k_min_tau0; g.k_max_tau0_over_l_max = pr.k_max_tau0_over_l_max; g.k_step_sub = pr.k_step_sub; g.k_step_super = pr.k_step_super;
  g.k_step_transition = pr.k_step_transition; g.k_step_super_reduction = pr.k_step_super_reduction; g.k_per_decade_for_pk = pr.k_per_decade_for_pk;
  g.k_per_decade_for_bao = pr.k_per_decade_for_bao; g.k_bao_center = pr.k_bao_center; g.k_bao_width = pr.k_bao_width;
  g.has_cls = pt.has_cls; g.has_pk_matter = pt.has_pk_matter; g.l_scalar_max = pt.l_scalar_max; g.k_max_for_pk = pt.k_max_for_pk;
  g.rs_rec = th.rs_rec_; g.tau_ini_thermo = th.tau_ini_;
  g.start_sources_at_tau_c_over_tau_h = pr.start_sources_at_tau_c_over_tau_h; g.perturb_sampling_stepsize = pr.perturb_sampling_stepsize;
  g.l_linstep = pr.l_linstep; g.l_logstep = pr.l_logstep; g.q_linstep = pr.q_linstep; g.q_logstep_spline = pr.q_logstep_spline;
  g.q_logstep_open = pr.q_logstep_open; g.l_tensor_max = pt.l_tensor_max; g.q_logstep_trapzd = pr.q_logstep_trapzd;
  g.q_numstep_transition = pr.q_numstep_transition;
  g.tau_of_z_max_pk = 0.;   // z_max_pk > 0: the conformal time of that redshift (pm.cpp:1562)
  if (pt.z_max_pk > 0. && bg.background_tau_of_z(pt.z_max_pk, &g.tau_of_z_max_pk) != _SUCCESS_) throw std::runtime_error(bg.error_message_);
  return x;
}
