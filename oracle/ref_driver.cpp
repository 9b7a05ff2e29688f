// ORACLE / TEST INFRASTRUCTURE ONLY -- never linked, imported or executed by the product path.
//
// ref_driver: our own small driver that links against oracle/_ref/libclass_ref.so, i.e. the UNMODIFIED
// reference (AarhusCosmology/CLASSpp_public) compiled from the sources where they lie under /root/reference
// by oracle/Makefile.  It plays the role of main/class.cpp:9-24 (build a Cosmology from a FileContent and
// pull the lazy module DAG, source/cosmology.cpp:16-86) but instead of writing .dat files it
//   dump <ini> <out.bin>          : dumps hot-path INPUTS (background / thermodynamics spline tables, scalars,
//                                   precision & physics parameters, k / tau / q / l grids) and OUTPUTS
//                                   (sources_, transfer_, unlensed C_l, P(k)) as a flat list of named arrays;
//   time <ini> <threads> <reps>   : times the perturbation stage (pm.cpp:668-718) and the transfer stage
//                                   (tm.cpp:287-318) of the reference on this host -> one JSON line
//                                   (this is bench.py's cpu_baseline of kind "reference").
// Private module tables (background_table_, thermodynamics_table_, ...) are reached with the
// `#define private public` trick; nothing of the reference is modified or copied.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <string>
#include <vector>
#include <map>
#include <memory>
#include <tuple>
#include <future>
#include <thread>
#include <chrono>
#include <stdexcept>
#include <functional>
#include <mutex>
#include <condition_variable>
#include <deque>
#include <atomic>
#include <sstream>
#include <iostream>
#include <fstream>
#include <algorithm>
#include <numeric>
#include <set>
#include <list>
#include <array>

#define private public
#define protected public
#include "cosmology.h"
#include "background_module.h"
#include "non_cold_dark_matter.h"
#include "thermodynamics_module.h"
#include "perturbations_module.h"
#include "primordial_module.h"
#include "nonlinear_module.h"
#include "transfer_module.h"
#include "spectra_module.h"
#include "lensing_module.h"
#undef private
#undef protected

static FILE* g_out = nullptr;

static void put_f8(const char* name, const double* p, std::vector<long> shape) {
  long n = 1; for (long s : shape) n *= s;
  fprintf(g_out, "%s f8 %zu", name, shape.size());
  for (long s : shape) fprintf(g_out, " %ld", s);
  fprintf(g_out, "\n");
  fwrite(p, sizeof(double), n, g_out);
}
static void put_i4(const char* name, const int* p, std::vector<long> shape) {
  long n = 1; for (long s : shape) n *= s;
  fprintf(g_out, "%s i4 %zu", name, shape.size());
  for (long s : shape) fprintf(g_out, " %ld", s);
  fprintf(g_out, "\n");
  fwrite(p, sizeof(int), n, g_out);
}
static void put_d(const char* name, double v) { put_f8(name, &v, {1}); }
static void put_i(const char* name, int v) { put_i4(name, &v, {1}); }

static int read_ini(const char* ini, FileContent& fc) {
  ErrorMsg err;
  if (parser_read_file(ini, &fc, err) == _FAILURE_) {
    fprintf(stderr, "parser_read_file failed: %s\n", err);
    return 1;
  }
  return 0;
}

static void set_key(FileContent& fc, const char* key, const char* val) {
  for (int i = 0; i < fc.size; i++) {
    if (strcmp(fc.name[i], key) == 0) { strncpy(fc.value[i], val, _ARGUMENT_LENGTH_MAX_ - 1); return; }
  }
  fprintf(stderr, "key %s must be present in the ini to be overridden\n", key);
  exit(2);
}

#define PD(x) put_d("ppr." #x, ppr->x)
#define PI_(x) put_i("ppr." #x, (int)ppr->x)

static int do_dump(const char* ini, const char* outpath) {
  FileContent fc;
  if (read_ini(ini, fc)) return 1;
  Cosmology cosmo{fc};
  auto inp = cosmo.GetInputModule();
  const precision* ppr = &inp->precision_;
  const background* pba = &inp->background_;
  const thermo* pth = &inp->thermodynamics_;
  const perturbs* ppt = &inp->perturbations_;
  const primordial* ppm = &inp->primordial_;
  const transfers* ptr = &inp->transfers_;

  auto bg = cosmo.GetBackgroundModule();
  auto th = cosmo.GetThermodynamicsModule();
  auto pt = cosmo.GetPerturbationsModule();
  auto prim = cosmo.GetPrimordialModule();
  auto nl = cosmo.GetNonlinearModule();
  auto tr = cosmo.GetTransferModule();
  auto sp = cosmo.GetSpectraModule();

  g_out = fopen(outpath, "wb");
  if (!g_out) { perror("fopen"); return 1; }

  // ---- precision parameters read by the path (include/precisions.h:162-395) ----
  PD(k_min_tau0); PD(k_max_tau0_over_l_max); PD(k_step_sub); PD(k_step_super); PD(k_step_transition);
  PD(k_step_super_reduction); PD(k_per_decade_for_pk); PD(k_per_decade_for_bao); PD(k_bao_center); PD(k_bao_width);
  PD(start_small_k_at_tau_c_over_tau_h); PD(start_large_k_at_tau_h_over_tau_k);
  PD(tight_coupling_trigger_tau_c_over_tau_h); PD(tight_coupling_trigger_tau_c_over_tau_k);
  PD(start_sources_at_tau_c_over_tau_h); PI_(tight_coupling_approximation);
  PI_(l_max_g); PI_(l_max_pol_g); PI_(l_max_ur); PI_(l_max_ncdm); PI_(l_max_g_ten); PI_(l_max_pol_g_ten);
  PD(curvature_ini); PD(gw_ini); PD(perturb_integration_stepsize); PD(perturb_sampling_stepsize);
  PD(tol_perturb_integration); PD(tol_tau_approx); PD(smallest_allowed_variation);
  PI_(radiation_streaming_approximation); PD(radiation_streaming_trigger_tau_over_tau_k);
  PD(radiation_streaming_trigger_tau_c_over_tau);
  PI_(ur_fluid_approximation); PD(ur_fluid_trigger_tau_over_tau_k);
  PI_(ncdm_fluid_approximation); PD(ncdm_fluid_trigger_tau_over_tau_k);
  PD(neglect_CMB_sources_below_visibility); PI_(evolver);
  PD(l_linstep); PD(l_logstep); PD(hyper_x_min); PD(hyper_sampling_flat); PD(hyper_sampling_curved_low_nu);
  PD(hyper_sampling_curved_high_nu); PD(hyper_nu_sampling_step); PD(hyper_phi_min_abs); PD(hyper_x_tol);
  PD(hyper_flat_approximation_nu);
  PD(q_linstep); PD(q_logstep_spline); PD(q_logstep_open); PD(q_logstep_trapzd); PD(q_numstep_transition);
  PD(transfer_neglect_delta_k_S_t0); PD(transfer_neglect_delta_k_S_t1); PD(transfer_neglect_delta_k_S_t2);
  PD(transfer_neglect_delta_k_S_e); PD(transfer_neglect_late_source); PD(l_switch_limber);

  // ---- physics / flags ----
  put_d("pba.H0", pba->H0); put_d("pba.h", pba->h); put_d("pba.K", pba->K); put_i("pba.sgnK", pba->sgnK);
  put_d("pba.a_today", pba->a_today); put_d("pba.T_cmb", pba->T_cmb);
  put_d("pba.Omega0_g", pba->Omega0_g); put_d("pba.Omega0_b", pba->Omega0_b); put_d("pba.Omega0_cdm", pba->Omega0_cdm);
  put_d("pba.Omega0_lambda", pba->Omega0_lambda); put_d("pba.Omega0_ur", pba->Omega0_ur); put_d("pba.Omega0_k", pba->Omega0_k);
  put_i("pba.has_cdm", pba->has_cdm); put_i("pba.has_ur", pba->has_ur); put_i("pba.has_ncdm", pba->has_ncdm);
  put_i("pba.has_lambda", pba->has_lambda); put_i("pba.has_fld", pba->has_fld); put_i("pba.has_curvature", pba->has_curvature);
  put_i("pba.N_ncdm", pba->N_ncdm);
  put_i("pth.reio_parametrization", (int)pth->reio_parametrization);
  put_i("pth.compute_cb2_derivatives", pth->compute_cb2_derivatives);
  put_i("ppt.gauge", (int)ppt->gauge); put_i("ppt.has_scalars", ppt->has_scalars); put_i("ppt.has_tensors", ppt->has_tensors);
  put_i("ppt.has_ad", ppt->has_ad); put_i("ppt.has_bi", ppt->has_bi); put_i("ppt.has_cdi", ppt->has_cdi);
  put_i("ppt.has_nid", ppt->has_nid); put_i("ppt.has_niv", ppt->has_niv);
  put_i("ppt.has_cl_cmb_temperature", ppt->has_cl_cmb_temperature);
  put_i("ppt.has_cl_cmb_polarization", ppt->has_cl_cmb_polarization);
  put_i("ppt.has_cl_cmb_lensing_potential", ppt->has_cl_cmb_lensing_potential);
  put_i("ppt.has_pk_matter", ppt->has_pk_matter);
  put_i("ppt.l_scalar_max", ppt->l_scalar_max); put_i("ppt.l_tensor_max", ppt->l_tensor_max); put_d("ppt.k_max_for_pk", ppt->k_max_for_pk);
  put_d("ppt.z_max_pk", ppt->z_max_pk);
  put_i("ppt.switch_sw", ppt->switch_sw); put_i("ppt.switch_eisw", ppt->switch_eisw); put_i("ppt.switch_lisw", ppt->switch_lisw);
  put_i("ppt.switch_dop", ppt->switch_dop); put_i("ppt.switch_pol", ppt->switch_pol);
  put_d("ppt.eisw_lisw_split_z", ppt->eisw_lisw_split_z);
  put_d("ppt.three_ceff2_ur", ppt->three_ceff2_ur); put_d("ppt.three_cvis2_ur", ppt->three_cvis2_ur);
  put_d("ppt.G_eff_ur", ppt->G_eff_ur);
  put_d("ptr.lcmb_rescale", ptr->lcmb_rescale); put_d("ptr.lcmb_tilt", ptr->lcmb_tilt); put_d("ptr.lcmb_pivot", ptr->lcmb_pivot);
  put_d("ppm.A_s", ppm->A_s); put_d("ppm.n_s", ppm->n_s); put_d("ppm.alpha_s", ppm->alpha_s); put_d("ppm.k_pivot", ppm->k_pivot);
  // effective power law of the first (only) initial condition: amplitude, tilt, running (primordial_module.cpp:684-800)
  put_d("ppm.amplitude0", prim->amplitude_[0][0]); put_d("ppm.tilt0", prim->tilt_[0][0]); put_d("ppm.running0", prim->running_[0][0]);
  put_d("ppr.entropy_ini", ppr->entropy_ini);

  // ---- background tables (source/background_module.h:166-178) ----
  put_i("bg.bt_size", bg->bt_size_); put_i("bg.bg_size", bg->bg_size_);
  put_i("bg.bg_size_short", bg->bg_size_short_); put_i("bg.bg_size_normal", bg->bg_size_normal_);
  put_f8("bg.tau_table", bg->tau_table_, {bg->bt_size_});
  put_f8("bg.z_table", bg->z_table_, {bg->bt_size_});
  put_f8("bg.background_table", bg->background_table_, {bg->bt_size_, bg->bg_size_});
  put_f8("bg.d2background_dtau2_table", bg->d2background_dtau2_table_, {bg->bt_size_, bg->bg_size_});
  put_i("bg.index_bg_a", bg->index_bg_a_); put_i("bg.index_bg_H", bg->index_bg_H_); put_i("bg.index_bg_H_prime", bg->index_bg_H_prime_);
  put_i("bg.index_bg_rho_g", bg->index_bg_rho_g_); put_i("bg.index_bg_rho_b", bg->index_bg_rho_b_);
  put_i("bg.index_bg_rho_cdm", bg->index_bg_rho_cdm_); put_i("bg.index_bg_rho_lambda", bg->index_bg_rho_lambda_);
  put_i("bg.index_bg_rho_ur", bg->index_bg_rho_ur_); put_i("bg.index_bg_rho_tot", bg->index_bg_rho_tot_);
  put_i("bg.index_bg_p_tot", bg->index_bg_p_tot_); put_i("bg.index_bg_p_tot_prime", bg->index_bg_p_tot_prime_);
  put_i("bg.index_bg_Omega_r", bg->index_bg_Omega_r_); put_i("bg.index_bg_rho_crit", bg->index_bg_rho_crit_);
  put_i("bg.index_bg_Omega_m", bg->index_bg_Omega_m_); put_i("bg.index_bg_conf_distance", bg->index_bg_conf_distance_);
  put_i("bg.index_bg_D", bg->index_bg_D_); put_i("bg.index_bg_f", bg->index_bg_f_);
  if (pba->has_ncdm) {   // non-cold species: momentum grids of the perturbation sampling (tools/non_cold_dark_matter.h:72-79)
    const auto& nc = *inp->ncdm_;
    put_i("bg.index_bg_rho_ncdm1", bg->index_bg_rho_ncdm1_); put_i("bg.index_bg_p_ncdm1", bg->index_bg_p_ncdm1_);
    put_i("bg.index_bg_pseudo_p_ncdm1", bg->index_bg_pseudo_p_ncdm1_);
    put_d("ppr.tol_ncdm_initial_w", ppr->tol_ncdm_initial_w);
    std::vector<double> M(nc.N_ncdm_), fac(nc.N_ncdm_);
    for (int n = 0; n < nc.N_ncdm_; n++) {
      M[n] = nc.M_ncdm_[n]; fac[n] = nc.factor_ncdm_[n];
      if (nc.ncdm_types_[n] != NCDMType::standard) { fprintf(stderr, "only standard ncdm species are dumped\n"); exit(3); }
      char nm[64];
      snprintf(nm, sizeof nm, "ncdm.q_%d", n); put_f8(nm, nc.q_ncdm_[n], {nc.q_size_ncdm_[n]});
      snprintf(nm, sizeof nm, "ncdm.w_%d", n); put_f8(nm, nc.w_ncdm_[n], {nc.q_size_ncdm_[n]});
      snprintf(nm, sizeof nm, "ncdm.dlnf0_dlnq_%d", n); put_f8(nm, nc.dlnf0_dlnq_ncdm_[n], {nc.q_size_ncdm_[n]});
      // the background's own (finer) momentum sampling, input of the host background module (tools/non_cold_dark_matter.h:120-122)
      snprintf(nm, sizeof nm, "ncdm.q_bg_%d", n); put_f8(nm, nc.q_ncdm_bg_[n], {nc.q_size_ncdm_bg_[n]});
      snprintf(nm, sizeof nm, "ncdm.w_bg_%d", n); put_f8(nm, nc.w_ncdm_bg_[n], {nc.q_size_ncdm_bg_[n]});
    }
    put_f8("ncdm.M", M.data(), {nc.N_ncdm_}); put_f8("ncdm.factor", fac.data(), {nc.N_ncdm_});
  }
  put_d("bg.conformal_age", bg->conformal_age_); put_d("bg.Omega0_m", bg->Omega0_m_);

  // ---- thermodynamics tables (source/thermodynamics_module.h:120-125) ----
  put_i("th.tt_size", th->tt_size_); put_i("th.th_size", th->th_size_);
  put_f8("th.z_table", th->z_table_, {th->tt_size_});
  put_f8("th.thermodynamics_table", th->thermodynamics_table_, {th->tt_size_, th->th_size_});
  put_f8("th.d2thermodynamics_dz2_table", th->d2thermodynamics_dz2_table_, {th->tt_size_, th->th_size_});
  put_i("th.index_th_xe", th->index_th_xe_); put_i("th.index_th_dkappa", th->index_th_dkappa_);
  put_i("th.index_th_tau_d", th->index_th_tau_d_); put_i("th.index_th_ddkappa", th->index_th_ddkappa_);
  put_i("th.index_th_dddkappa", th->index_th_dddkappa_); put_i("th.index_th_exp_m_kappa", th->index_th_exp_m_kappa_);
  put_i("th.index_th_g", th->index_th_g_); put_i("th.index_th_dg", th->index_th_dg_); put_i("th.index_th_ddg", th->index_th_ddg_);
  put_i("th.index_th_Tb", th->index_th_Tb_); put_i("th.index_th_wb", th->index_th_wb_); put_i("th.index_th_cb2", th->index_th_cb2_);
  put_i("th.index_th_rate", th->index_th_rate_);
  put_d("th.tau_ini", th->tau_ini_); put_d("th.YHe", th->YHe_); put_d("th.z_rec", th->z_rec_); put_d("th.tau_rec", th->tau_rec_);
  put_d("th.rs_rec", th->rs_rec_); put_d("th.ra_rec", th->ra_rec_); put_d("th.angular_rescaling", th->angular_rescaling_);
  put_d("th.tau_free_streaming", th->tau_free_streaming_); put_d("th.tau_cut", th->tau_cut_);
  put_d("th.n_e", th->n_e_); put_d("th.z_reionization", th->z_reionization_);

  // ---- perturbations: grids + sources_ (source/perturbations_module.h:152-178) ----
  // scalars when present, else the tensor mode (a tensors-only run, modes = t)
  const bool tens = !ppt->has_scalars && ppt->has_tensors;
  int md = tens ? pt->index_md_tensors_ : pt->index_md_scalars_;
  put_i("pt.mode_tensors", tens ? 1 : 0);
  put_i("ppt.tensor_method", (int)ppt->tensor_method); put_i("pt.evolve_tensor_ur", tens ? (int)pt->evolve_tensor_ur_ : 0);
  put_d("ppr.transfer_neglect_delta_k_T_t2", ppr->transfer_neglect_delta_k_T_t2); put_d("ppr.transfer_neglect_delta_k_T_e", ppr->transfer_neglect_delta_k_T_e);
  put_d("ppr.transfer_neglect_delta_k_T_b", ppr->transfer_neglect_delta_k_T_b);
  int nk = pt->k_size_[md], ntau = pt->tau_size_, ntp = pt->tp_size_[md];
  put_i("pt.md_size", pt->md_size_); put_i("pt.ic_size", pt->ic_size_[md]); put_i("pt.tp_size", ntp);
  put_i("pt.k_size", nk); put_i("pt.k_size_cl", pt->k_size_cl_[md]); put_i("pt.k_size_cmb", pt->k_size_cmb_[md]);
  put_i("pt.tau_size", ntau); put_i("pt.ln_tau_size", pt->ln_tau_size_);
  put_f8("pt.k", pt->k_[md], {nk});
  put_f8("pt.tau_sampling", pt->tau_sampling_, {ntau});
  put_i("pt.index_tp_t0", (pt->has_source_t_ && !tens) ? pt->index_tp_t0_ : -1);
  put_i("pt.index_tp_t1", (pt->has_source_t_ && !tens) ? pt->index_tp_t1_ : -1);
  put_i("pt.index_tp_t2", pt->has_source_t_ ? pt->index_tp_t2_ : -1);
  put_i("pt.index_tp_p", pt->has_source_p_ ? pt->index_tp_p_ : -1);
  put_i("pt.index_tp_phi_plus_psi", pt->has_source_phi_plus_psi_ ? pt->index_tp_phi_plus_psi_ : -1);
  put_i("pt.index_tp_delta_m", pt->has_source_delta_m_ ? pt->index_tp_delta_m_ : -1);
  put_i("pt.index_tp_delta_cb", pt->has_source_delta_cb_ ? pt->index_tp_delta_cb_ : -1);
  // density / velocity transfer functions (output = mTk, vTk: pm.cpp:1000-1050, 6930-7200)
  put_i("pt.index_tp_delta_tot", (pt->has_source_delta_tot_ && !tens) ? pt->index_tp_delta_tot_ : -1);
  put_i("pt.index_tp_delta_g", (pt->has_source_delta_g_ && !tens) ? pt->index_tp_delta_g_ : -1);
  put_i("pt.index_tp_delta_b", (pt->has_source_delta_b_ && !tens) ? pt->index_tp_delta_b_ : -1);
  put_i("pt.index_tp_delta_cdm", (pt->has_source_delta_cdm_ && !tens) ? pt->index_tp_delta_cdm_ : -1);
  put_i("pt.index_tp_delta_ur", (pt->has_source_delta_ur_ && !tens) ? pt->index_tp_delta_ur_ : -1);
  put_i("pt.index_tp_theta_tot", (pt->has_source_theta_tot_ && !tens) ? pt->index_tp_theta_tot_ : -1);
  put_i("pt.index_tp_theta_g", (pt->has_source_theta_g_ && !tens) ? pt->index_tp_theta_g_ : -1);
  put_i("pt.index_tp_theta_b", (pt->has_source_theta_b_ && !tens) ? pt->index_tp_theta_b_ : -1);
  put_i("pt.index_tp_theta_cdm", (pt->has_source_theta_cdm_ && !tens) ? pt->index_tp_theta_cdm_ : -1);
  put_i("pt.index_tp_theta_ur", (pt->has_source_theta_ur_ && !tens) ? pt->index_tp_theta_ur_ : -1);
  put_i("pt.index_tp_phi", (pt->has_source_phi_ && !tens) ? pt->index_tp_phi_ : -1);
  put_i("pt.index_tp_psi", (pt->has_source_psi_ && !tens) ? pt->index_tp_psi_ : -1);
  put_i("pt.index_tp_delta_ncdm1", (pt->has_source_delta_ncdm_ && !tens) ? pt->index_tp_delta_ncdm1_ : -1);   // (N_ncdm consecutive slots)
  put_i("pt.index_tp_theta_ncdm1", (pt->has_source_theta_ncdm_ && !tens) ? pt->index_tp_theta_ncdm1_ : -1);
  {
    std::vector<double> s((size_t)ntp * ntau * nk);
    for (int tp = 0; tp < ntp; tp++)
      memcpy(&s[(size_t)tp * ntau * nk], pt->sources_[md][tp], sizeof(double) * (size_t)ntau * nk);
    put_f8("pt.sources", s.data(), {ntp, ntau, nk});
  }

  // ---- transfer: grids + transfer_ (source/transfer_module.h:12-57) ----
  if (ppt->has_cls) {
    int nq = tr->q_size_, nl_ = tr->l_size_[md], ntt = tr->tt_size_[md];
    put_i("tr.q_size", nq); put_i("tr.l_size", nl_); put_i("tr.tt_size", ntt);
    put_f8("tr.q", tr->q_, {nq}); put_f8("tr.k", tr->k_[md], {nq});
    put_i4("tr.l", tr->l_, {nl_});
    put_i4("tr.l_size_tt", tr->l_size_tt_[md], {ntt});
    put_i("tr.index_tt_t0", (ppt->has_cl_cmb_temperature && !tens) ? tr->index_tt_t0_ : -1);
    put_i("tr.index_tt_t1", (ppt->has_cl_cmb_temperature && !tens) ? tr->index_tt_t1_ : -1);
    put_i("tr.index_tt_b", (ppt->has_cl_cmb_polarization && tens) ? tr->index_tt_b_ : -1);
    put_i("tr.index_tt_t2", ppt->has_cl_cmb_temperature ? tr->index_tt_t2_ : -1);
    put_i("tr.index_tt_e", ppt->has_cl_cmb_polarization ? tr->index_tt_e_ : -1);
    put_i("tr.index_tt_lcmb", (ppt->has_cl_cmb_lensing_potential && !tens) ? tr->index_tt_lcmb_ : -1);
    put_f8("tr.transfer", tr->transfer_[md], {ntt, nl_, nq});

    // ---- spectra: the C_l table at the l_ grid and at every integer l (source/spectra_module.cpp:146-218) ----
    int lmax = sp->l_max_tot_;
    put_i("sp.l_max_tot", lmax); put_i("sp.ct_size", sp->ct_size_);
    auto cls = sp->cl_output(lmax);
    for (auto& kv : cls) {
      std::string nm = "sp.cl_" + kv.first;
      put_f8(nm.c_str(), kv.second.data(), {(long)kv.second.size()});
    }
    put_f8("sp.cl_table", sp->cl_[md], {sp->l_size_[md], sp->ct_size_});
    put_i("sp.index_ct_tt", sp->has_tt_ ? sp->index_ct_tt_ : -1);
    put_i("sp.index_ct_ee", sp->has_ee_ ? sp->index_ct_ee_ : -1);
    put_i("sp.index_ct_te", sp->has_te_ ? sp->index_ct_te_ : -1);
    put_i("sp.index_ct_pp", sp->has_pp_ ? sp->index_ct_pp_ : -1);
    put_i("sp.index_ct_tp", sp->has_tp_ ? sp->index_ct_tp_ : -1);
    put_i("sp.index_ct_ep", sp->has_ep_ ? sp->index_ct_ep_ : -1);
    put_i("sp.index_ct_bb", sp->has_bb_ ? sp->index_ct_bb_ : -1);

    // ---- lensed C_l table on the lensing module's l grid (source/lensing_module.cpp:149-854) ----
    const lensing* ple = &inp->lensing_;
    if (ple->has_lensed_cls == _TRUE_) {
      auto le = cosmo.GetLensingModule();
      put_i("le.l_unlensed_max", le->l_unlensed_max_); put_i("le.l_lensed_max", le->l_lensed_max_);
      put_i("le.l_size", le->l_size_); put_i("le.lt_size", le->lt_size_);
      put_f8("le.l", le->l_, {le->l_size_});
      put_f8("le.cl_lens", le->cl_lens_, {le->l_size_, le->lt_size_});
      put_i4("le.l_max_lt", le->l_max_lt_, {le->lt_size_});
      put_i("le.accurate_lensing", (int)ppr->accurate_lensing); put_i("le.delta_l_max", (int)ppr->delta_l_max);
      put_i("le.num_mu_minus_lmax", (int)ppr->num_mu_minus_lmax);
      auto lcls = le->cl_output(le->l_lensed_max_);
      for (auto& kv : lcls) {
        std::string nm = "le.cl_" + kv.first;
        put_f8(nm.c_str(), kv.second.data(), {(long)kv.second.size()});
      }
    }
  }

  // ---- linear P(k, z=0) on the module's own k grid (source/nonlinear_module.cpp:1886-2040) ----
  if (ppt->has_pk_matter) {
    int nkk = nl->k_size_;
    std::vector<double> kk(nkk), pk(nkk);
    for (int i = 0; i < nkk; i++) { kk[i] = nl->k_[i]; }
    std::vector<double> lnpk(nkk);
    std::vector<double> lnpk_ic(nkk * nl->ic_ic_size_);
    int st = nl->nonlinear_pk_at_z(logarithmic, pk_linear, 0., nl->index_pk_m_, lnpk.data(), lnpk_ic.data());
    if (st != _SUCCESS_) { fprintf(stderr, "nonlinear_pk_at_z failed: %s\n", nl->error_message_); return 1; }
    for (int i = 0; i < nkk; i++) pk[i] = exp(lnpk[i]);
    put_f8("nl.k", kk.data(), {nkk}); put_f8("nl.pk_lin_z0", pk.data(), {nkk});
    put_d("nl.sigma8", nl->sigma8_[nl->index_pk_m_]);
    if (nl->has_pk_cb_) {   // baryons + cold dark matter only (defined with non-cold species)
      st = nl->nonlinear_pk_at_z(logarithmic, pk_linear, 0., nl->index_pk_cb_, lnpk.data(), lnpk_ic.data());
      if (st != _SUCCESS_) { fprintf(stderr, "nonlinear_pk_at_z (cb) failed: %s\n", nl->error_message_); return 1; }
      for (int i = 0; i < nkk; i++) pk[i] = exp(lnpk[i]);
      put_f8("nl.pk_cb_lin_z0", pk.data(), {nkk});
      put_d("nl.sigma8_cb", nl->sigma8_[nl->index_pk_cb_]);
    }
    // ---- P(k, z) and sigma(8/h, z) at the redshifts of the .ini's z_pk list (z_max_pk > 0: nonlinear_pk_at_z splines ln P in ln tau over the
    //      tail ln_tau_ of the sampling, pm.cpp:1554-1592, nonlinear_module.cpp:81-283, :927-963) ----
    const output* pop = &inp->output_;
    if (ppt->z_max_pk > 0.) {
      const int nz = pop->z_pk_num;
      std::vector<double> zs(nz), pkz((size_t)nz * nkk), s8z(nz), tauz(nz);
      for (int iz = 0; iz < nz; iz++) {
        zs[iz] = pop->z_pk[iz];
        st = nl->nonlinear_pk_at_z(logarithmic, pk_linear, zs[iz], nl->index_pk_m_, lnpk.data(), lnpk_ic.data());
        if (st != _SUCCESS_) { fprintf(stderr, "nonlinear_pk_at_z(z=%g) failed: %s\n", zs[iz], nl->error_message_); return 1; }
        for (int i = 0; i < nkk; i++) pkz[(size_t)iz * nkk + i] = exp(lnpk[i]);
        if (nl->nonlinear_sigmas_at_z(8. / pba->h, zs[iz], nl->index_pk_m_, out_sigma, &s8z[iz]) != _SUCCESS_) { fprintf(stderr, "sigmas_at_z failed\n"); return 1; }
        if (bg->background_tau_of_z(zs[iz], &tauz[iz]) != _SUCCESS_) { fprintf(stderr, "tau_of_z failed\n"); return 1; }
      }
      put_f8("nl.z_pk", zs.data(), {nz}); put_f8("nl.pk_lin_z", pkz.data(), {nz, nkk}); put_f8("nl.sigma8_z", s8z.data(), {nz});
      put_f8("nl.tau_of_z_pk", tauz.data(), {nz});
      put_f8("pt.ln_tau", pt->ln_tau_, {pt->ln_tau_size_});
      put_d("ppt.z_max_pk", ppt->z_max_pk);
      if (ppt->has_density_transfers || ppt->has_velocity_transfers) {
        // every source type at the redshifts of z_pk (perturb_sources_at_tau: what perturb_output_data / classy.get_transfer(z) read)
        const int md0 = pt->index_md_scalars_, ntp0 = pt->tp_size_[md0], nk0 = pt->k_size_[md0];
        std::vector<double> sz((size_t)nz * ntp0 * nk0);
        for (int iz = 0; iz < nz; iz++)
          for (int tp = 0; tp < ntp0; tp++)
            if (pt->perturb_sources_at_tau(md0, 0, tp, tauz[iz], &sz[((size_t)iz * ntp0 + tp) * nk0]) != _SUCCESS_) { fprintf(stderr, "sources_at_tau failed\n"); return 1; }
        put_f8("pt.sources_at_z_pk", sz.data(), {nz, ntp0, nk0});
      }
    }
  }
  // ---- what the reference's Python wrapper reads "at z" (classy.pyx:825-1080: background_tau_of_z + background_at_tau with long_info,
  //      thermodynamics_at_z) on a fixed list of redshifts, and the scalars behind rs_drag(), k_eq(), theta_star_100() ----
  {
    const double zl[] = {0., 0.1, 0.5, 1., 2., 3., 10., 50., 300., 1000., 1100., 2000.};
    const int nz = (int)(sizeof(zl) / sizeof(zl[0]));
    std::vector<double> vb((size_t)nz * bg->bg_size_), vt((size_t)nz * th->th_size_), zs(zl, zl + nz), taus(nz);
    for (int iz = 0; iz < nz; iz++) {
      int last = 0;
      if (bg->background_tau_of_z(zl[iz], &taus[iz]) != _SUCCESS_) { fprintf(stderr, "tau_of_z failed\n"); return 1; }
      if (bg->background_at_tau(taus[iz], pba->long_info, pba->inter_normal, &last, &vb[(size_t)iz * bg->bg_size_]) != _SUCCESS_) { fprintf(stderr, "background_at_tau failed\n"); return 1; }
      if (th->thermodynamics_at_z(zl[iz], th->inter_normal_, &last, &vb[(size_t)iz * bg->bg_size_], &vt[(size_t)iz * th->th_size_]) != _SUCCESS_) { fprintf(stderr, "thermodynamics_at_z failed\n"); return 1; }
    }
    put_f8("atz.z", zs.data(), {nz}); put_f8("atz.tau", taus.data(), {nz});
    put_f8("atz.bg", vb.data(), {nz, bg->bg_size_}); put_f8("atz.th", vt.data(), {nz, th->th_size_});
    put_i("atz.index_bg_ang_distance", bg->index_bg_ang_distance_); put_i("atz.index_bg_lum_distance", bg->index_bg_lum_distance_);
    put_i("atz.index_bg_time", bg->index_bg_time_); put_i("atz.index_bg_rs", bg->index_bg_rs_);
    put_d("atz.a_eq", bg->a_eq_); put_d("atz.H_eq", bg->H_eq_);
    put_d("atz.rs_d", th->rs_d_); put_d("atz.z_d", th->z_d_); put_d("atz.rs_star", th->rs_star_); put_d("atz.ra_star", th->ra_star_);
  }
  fclose(g_out);
  return 0;
}

static int do_time(const char* ini, const char* threads, int reps) {
  double best_pt = 1e30, best_tr = 1e30;
  int nk = 0, nq = 0, nl_ = 0, ntt = 0;
  for (int r = 0; r < reps; r++) {
    FileContent fc;
    if (read_ini(ini, fc)) return 1;
    set_key(fc, "threads", threads);
    Cosmology cosmo{fc};
    cosmo.GetThermodynamicsModule();
    auto t0 = std::chrono::steady_clock::now();
    auto pt = cosmo.GetPerturbationsModule();
    auto t1 = std::chrono::steady_clock::now();
    cosmo.GetNonlinearModule();
    auto t2 = std::chrono::steady_clock::now();
    auto tr = cosmo.GetTransferModule();
    auto t3 = std::chrono::steady_clock::now();
    best_pt = std::min(best_pt, std::chrono::duration<double>(t1 - t0).count());
    best_tr = std::min(best_tr, std::chrono::duration<double>(t3 - t2).count());
    int md = 0;
    nk = pt->k_size_[md]; nq = tr->q_size_; nl_ = tr->l_size_[md]; ntt = tr->tt_size_[md];
  }
  printf("{\"perturb_s\": %.6f, \"transfer_s\": %.6f, \"k_size\": %d, \"q_size\": %d, \"l_size\": %d, \"tt_size\": %d, \"threads\": %s, \"reps\": %d}\n",
         best_pt, best_tr, nk, nq, nl_, ntt, threads, reps);
  return 0;
}

int main(int argc, char** argv) {
  try {
    if (argc >= 4 && strcmp(argv[1], "dump") == 0) return do_dump(argv[2], argv[3]);
    if (argc >= 5 && strcmp(argv[1], "time") == 0) return do_time(argv[2], argv[3], atoi(argv[4]));
  } catch (std::exception& e) {
    fprintf(stderr, "reference raised: %s\n", e.what());
    return 3;
  }
  fprintf(stderr, "usage: ref_driver dump <ini> <out.bin> | time <ini> <threads> <reps>\n");
  return 2;
}
