// ORACLE / TEST INFRASTRUCTURE ONLY -- never linked, imported or executed by the product path.
//
// adapter_check: compiles include/reference_side/cpt_adapter.h - the binding INTEGRATION.md proposes for the reference tree - against
// the UNMODIFIED reference headers, runs the reference's InputModule / BackgroundModule / ThermodynamicsModule on an .ini and prints
// every scalar of the cpt::Inputs the adapter produces as `name value` lines (doubles in %.17g), plus checksums of the tables the
// pointers lead to.  tests/test_adapter.py compares that with what this repository derives from the committed fixtures.
//     adapter_check <ini>
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

// (the reference has no accessor for the spline tables' second derivatives and the ncdm grids: see the header of cpt_adapter.h)
#define private public
#define protected public
#include "cosmology.h"
#include "background_module.h"
#include "non_cold_dark_matter.h"
#include "thermodynamics_module.h"
#undef private
#undef protected

#include "reference_side/cpt_adapter.h"

static double checksum(const double* p, long n) {
  double s = 0.;
  for (long i = 0; i < n; i++) s += p[i] * (1. + (double)(i % 97) / 97.);
  return s;
}

#define D(s, f) printf(#s "." #f " %.17g\n", (double)x.s.f)
#define I(s, f) printf(#s "." #f " %d\n", (int)x.s.f)

static void print_config(const char* tag, const cpt_config& c) {
#define CD(f) printf("%s." #f " %.17g\n", tag, (double)c.f)
#define CI(f) printf("%s." #f " %d\n", tag, (int)c.f)
  CD(H0); CD(K); CI(sgnK); CI(has_cdm); CI(has_ur); CI(has_ncdm); CI(has_fld); CI(has_curvature); CD(T_cmb); CD(a_today);
  CD(YHe); CD(n_e); CD(tau0); CD(tau_rec); CD(tau_free_streaming); CD(tau_cut); CD(angular_rescaling);
  CI(gauge); CI(switch_sw); CI(switch_eisw); CI(switch_lisw); CI(switch_dop); CI(switch_pol); CD(eisw_lisw_split_z);
  CD(three_ceff2_ur); CD(three_cvis2_ur);
  CI(tp_size); CI(index_tp_t0); CI(index_tp_t1); CI(index_tp_t2); CI(index_tp_p); CI(index_tp_delta_m); CI(index_tp_phi_plus_psi);
  CD(start_small_k_at_tau_c_over_tau_h); CD(start_large_k_at_tau_h_over_tau_k); CD(tight_coupling_trigger_tau_c_over_tau_h);
  CD(tight_coupling_trigger_tau_c_over_tau_k); CI(tight_coupling_approximation); CI(radiation_streaming_approximation);
  CD(radiation_streaming_trigger_tau_over_tau_k); CI(ur_fluid_approximation); CD(ur_fluid_trigger_tau_over_tau_k);
  CI(l_max_g); CI(l_max_pol_g); CI(l_max_ur); CD(curvature_ini); CD(tol_perturb_integration); CD(tol_tau_approx); CD(smallest_allowed_variation);
  CI(tt_size); CI(index_tt_t0); CI(index_tt_t1); CI(index_tt_t2); CI(index_tt_e); CI(index_tt_lcmb);
  CD(lcmb_rescale); CD(lcmb_tilt); CD(lcmb_pivot); CD(hyper_x_min); CD(hyper_sampling_flat); CD(hyper_phi_min_abs);
  CD(transfer_neglect_delta_k_S_t0); CD(transfer_neglect_delta_k_S_t1); CD(transfer_neglect_delta_k_S_t2); CD(transfer_neglect_delta_k_S_e);
  CD(transfer_neglect_late_source); CD(l_switch_limber);
  CI(ic); CD(entropy_ini); CI(mode); CI(l_max_g_ten); CI(l_max_pol_g_ten); CD(gw_ini); CI(evolve_tensor_ur); CI(index_tt_b);
  CD(transfer_neglect_delta_k_T_t2); CD(transfer_neglect_delta_k_T_e); CD(transfer_neglect_delta_k_T_b);
  CD(hyper_sampling_curved_low_nu); CD(hyper_sampling_curved_high_nu); CD(hyper_nu_sampling_step); CD(hyper_flat_approximation_nu);
  CI(N_ncdm); CI(l_max_ncdm); CI(ncdm_fluid_approximation); CD(ncdm_fluid_trigger_tau_over_tau_k); CD(tol_ncdm_initial_w);
  CI(index_tp_delta_cb); CI(tensor_method); CI(has_transfers); CI(index_tp_delta_ncdm1); CI(index_tp_theta_ncdm1);
  printf("%s.index_tp_transfer %d\n", tag, (int)CPT_NTK);
  for (int i = 0; i < CPT_NTK; i++) printf("%s.index_tp_transfer.%d %d\n", tag, i, c.index_tp_transfer[i]);
#undef CD
#undef CI
}

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: adapter_check <ini>\n"); return 2; }
  FileContent fc;
  ErrorMsg err;
  if (parser_read_file(argv[1], &fc, err) == _FAILURE_) { fprintf(stderr, "parser_read_file failed: %s\n", err); return 1; }
  Cosmology cosmo{fc};
  auto inp = cosmo.GetInputModule();
  auto bg = cosmo.GetBackgroundModule();
  auto th = cosmo.GetThermodynamicsModule();
  const cpt::Inputs x = MakeCptInputs(*inp, *bg, *th);
  print_config("config", x.config);
  printf("n_ic %d\n", x.n_ic);
  for (int i = 0; i < x.n_ic; i++) printf("ic.%d %d\n", i, x.ic[i]);
  printf("with_tensors %d\n", (int)x.with_tensors);
  if (x.with_tensors) print_config("config_tensors", x.config_tensors);
  I(tables, bt_size); I(tables, bg_size); I(tables, index_bg_a); I(tables, index_bg_H); I(tables, index_bg_H_prime); I(tables, index_bg_rho_g);
  I(tables, index_bg_rho_b); I(tables, index_bg_rho_cdm); I(tables, index_bg_rho_ur); I(tables, tt_size); I(tables, th_size);
  I(tables, index_th_xe); I(tables, index_th_dkappa); I(tables, index_th_tau_d); I(tables, index_th_ddkappa); I(tables, index_th_dddkappa);
  I(tables, index_th_exp_m_kappa); I(tables, index_th_g); I(tables, index_th_dg); I(tables, index_th_cb2); I(tables, index_th_rate);
  I(tables, index_bg_rho_ncdm1); I(tables, index_bg_p_ncdm1); I(tables, index_bg_pseudo_p_ncdm1);
  const cpt_tables& t = x.tables;
  printf("sum.tau_table %.17g\n", checksum(t.tau_table, t.bt_size));
  printf("sum.background_table %.17g\n", checksum(t.background_table, (long)t.bt_size * t.bg_size));
  printf("sum.d2background_dtau2_table %.17g\n", checksum(t.d2background_dtau2_table, (long)t.bt_size * t.bg_size));
  printf("sum.z_table %.17g\n", checksum(t.z_table, t.tt_size));
  printf("sum.thermodynamics_table %.17g\n", checksum(t.thermodynamics_table, (long)t.tt_size * t.th_size));
  printf("sum.d2thermodynamics_dz2_table %.17g\n", checksum(t.d2thermodynamics_dz2_table, (long)t.tt_size * t.th_size));
  for (int n = 0; n < x.config.N_ncdm; n++) {
    printf("ncdm.%d.q_size %d\n", n, t.q_size_ncdm[n]);
    printf("ncdm.%d.M %.17g\nncdm.%d.factor %.17g\n", n, t.M_ncdm[n], n, t.factor_ncdm[n]);
    printf("ncdm.%d.sum_q %.17g\nncdm.%d.sum_w %.17g\nncdm.%d.sum_dlnf0 %.17g\n", n, checksum(t.q_ncdm[n], t.q_size_ncdm[n]), n,
           checksum(t.w_ncdm[n], t.q_size_ncdm[n]), n, checksum(t.dlnf0_dlnq_ncdm[n], t.q_size_ncdm[n]));
  }
  D(grid, k_min_tau0); D(grid, k_max_tau0_over_l_max); D(grid, k_step_sub); D(grid, k_step_super); D(grid, k_step_transition);
  D(grid, k_step_super_reduction); D(grid, k_per_decade_for_pk); D(grid, k_per_decade_for_bao); D(grid, k_bao_center); D(grid, k_bao_width);
  I(grid, has_cls); I(grid, has_pk_matter); I(grid, l_scalar_max); D(grid, k_max_for_pk); D(grid, rs_rec); D(grid, tau_ini_thermo);
  D(grid, start_sources_at_tau_c_over_tau_h); D(grid, perturb_sampling_stepsize); D(grid, l_linstep); D(grid, l_logstep); D(grid, q_linstep);
  D(grid, q_logstep_spline); D(grid, q_logstep_open); I(grid, l_tensor_max); D(grid, q_logstep_trapzd); D(grid, q_numstep_transition); D(grid, tau_of_z_max_pk);
  return 0;
}
