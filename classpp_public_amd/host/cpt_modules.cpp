// C++ shim classes of include/cpt_modules.hpp: host logic only (grids, memory, error mapping); the compute is in
// libcpt.so behind the C ABI.  Link: g++ ... cpt_modules.cpp cpt_grids.cpp -L../csrc -lcpt -lamdhip64
#include "../../include/cpt_modules.hpp"

#include <hip/hip_runtime_api.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace cpt {

namespace {
template <class T>
T* xalloc(size_t n) {
  T* p = (T*)malloc(n * sizeof(T));
  if (!p) throw std::runtime_error("could not allocate memory");  // class_alloc, include/common.h:140-330
  return p;
}
[[noreturn]] void raise(int code, const char* msg) {
  if (code == CPT_ERR_INVALID || code == CPT_ERR_UNSUPPORTED) throw std::invalid_argument(msg);
  throw std::runtime_error(msg);
}
}  // namespace

HostTables::HostTables(const cpt_cosmo_params& cosmo, const cpt_thermo_params& th) {
  int rc = cpt_host_background(&cosmo, &background);
  if (rc) raise(rc, cpt_host_error());
  rc = cpt_host_thermodynamics(&cosmo, &th, &background, &thermo);
  if (rc) {
    const std::string msg = cpt_host_error();
    cpt_host_background_free(&background);
    raise(rc, msg.c_str());
  }
}

HostTables::~HostTables() {
  cpt_host_thermo_free(&thermo);
  cpt_host_background_free(&background);
}

void HostTables::fill(Inputs& in) const {
  cpt_tables& t = in.tables;
  const cpt_background& b = background;
  const cpt_thermo& h = thermo;
  t.bt_size = b.bt_size; t.bg_size = b.bg_size; t.tau_table = b.tau_table; t.background_table = b.background_table;
  t.d2background_dtau2_table = b.d2background_dtau2_table;
  t.index_bg_a = b.index_bg_a; t.index_bg_H = b.index_bg_H; t.index_bg_H_prime = b.index_bg_H_prime; t.index_bg_rho_g = b.index_bg_rho_g;
  t.index_bg_rho_b = b.index_bg_rho_b; t.index_bg_rho_cdm = b.index_bg_rho_cdm; t.index_bg_rho_ur = b.index_bg_rho_ur;
  t.index_bg_rho_ncdm1 = b.index_bg_rho_ncdm1; t.index_bg_p_ncdm1 = b.index_bg_p_ncdm1; t.index_bg_pseudo_p_ncdm1 = b.index_bg_pseudo_p_ncdm1;
  t.tt_size = h.tt_size; t.th_size = h.th_size; t.z_table = h.z_table; t.thermodynamics_table = h.thermodynamics_table;
  t.d2thermodynamics_dz2_table = h.d2thermodynamics_dz2_table;
  t.index_th_xe = h.index_th_xe; t.index_th_dkappa = h.index_th_dkappa; t.index_th_tau_d = h.index_th_tau_d; t.index_th_ddkappa = h.index_th_ddkappa;
  t.index_th_dddkappa = h.index_th_dddkappa; t.index_th_exp_m_kappa = h.index_th_exp_m_kappa; t.index_th_g = h.index_th_g; t.index_th_dg = h.index_th_dg;
  t.index_th_cb2 = h.index_th_cb2; t.index_th_rate = h.index_th_rate;
  cpt_config& c = in.config;
  c.YHe = h.YHe; c.n_e = h.n_e; c.tau0 = b.conformal_age; c.tau_rec = h.tau_rec; c.tau_free_streaming = h.tau_free_streaming;
  c.tau_cut = h.tau_cut; c.angular_rescaling = h.angular_rescaling;
  in.grid.rs_rec = h.rs_rec; in.grid.tau_ini_thermo = h.tau_ini;
}

PerturbationsModule::PerturbationsModule(const Inputs& in) {
  error_message_[0] = '\n';
  const cpt_config& c = in.config;
  // ---- perturb_indices_of_perturbs (pm.cpp:843-1235): the index contract comes in through cpt_config ----
  ic_size_ = xalloc<int>(1); ic_size_[0] = 1;
  tp_size_ = xalloc<int>(1); tp_size_[0] = c.tp_size;
  index_tp_t0_ = c.index_tp_t0; index_tp_t1_ = c.index_tp_t1; index_tp_t2_ = c.index_tp_t2; index_tp_p_ = c.index_tp_p;
  index_tp_delta_m_ = c.index_tp_delta_m; index_tp_phi_plus_psi_ = c.index_tp_phi_plus_psi;
  index_tp_delta_cb_ = c.has_ncdm ? c.index_tp_delta_cb : -1;
  has_source_t_ = c.index_tp_t0 >= 0; has_source_p_ = c.index_tp_p >= 0; has_source_delta_m_ = c.index_tp_delta_m >= 0;
  has_source_phi_plus_psi_ = c.index_tp_phi_plus_psi >= 0;
  // ---- perturb_get_k_list (pm.cpp:1628-2238) ----
  k_size_ = xalloc<int>(1); k_size_cl_ = xalloc<int>(1); k_size_cmb_ = xalloc<int>(1);
  k_ = xalloc<double*>(1);
  k_[0] = nullptr;
  std::vector<double> tmp(1 << 20);
  int rc = cpt_host_k_list(&c, &in.grid, tmp.data(), (int)tmp.size(), &k_size_[0], &k_size_cl_[0], &k_size_cmb_[0]);
  if (rc) { snprintf(error_message_, sizeof(error_message_), "%s", cpt_host_error()); raise(rc, error_message_); }
  k_[0] = xalloc<double>(k_size_[0]);
  memcpy(k_[0], tmp.data(), sizeof(double) * k_size_[0]);
  k_min_ = k_[0][0];
  k_max_ = k_[0][k_size_[0] - 1];
  // ---- perturb_timesampling_for_sources (pm.cpp:1247-1619) ----
  rc = cpt_host_tau_sampling(&c, &in.tables, &in.grid, tmp.data(), (int)tmp.size(), &tau_size_);
  if (rc) { snprintf(error_message_, sizeof(error_message_), "%s", cpt_host_error()); raise(rc, error_message_); }
  tau_sampling_ = xalloc<double>(tau_size_);
  memcpy(tau_sampling_, tmp.data(), sizeof(double) * tau_size_);
  ln_tau_size_ = 1;  // z_max_pk = 0 (pm.cpp:1554-1556)
  // ---- the k loop (pm.cpp:668-718) on the GPU ----
  rc = cpt_create(&c, &in.tables, &h_);
  if (rc) { snprintf(error_message_, sizeof(error_message_), "%s", cpt_create_error()); raise(rc, error_message_); }
  const int nk = k_size_[0], ntp = c.tp_size;
  shard_ = in.shard;
  if (shard_.world < 1 || shard_.rank < 0 || shard_.rank >= shard_.world) raise(CPT_ERR_INVALID, "Shard: rank outside [0, world)");
  stats_ = xalloc<cpt_stepstat>(nk);
  memset(stats_, 0, sizeof(cpt_stepstat) * nk);
  double* d_src = nullptr;
  const size_t nsrc = (size_t)ntp * tau_size_ * nk;
  if (hipMalloc((void**)&d_src, nsrc * sizeof(double)) != hipSuccess) raise(CPT_ERR_RUNTIME, "hipMalloc failed");
  if (shard_.comm_id) {
    // this rank's share of the k loop, then exchange 1: afterwards the handle holds the sources of every mode
    rc = cpt_comm_init(h_, shard_.comm_id, shard_.rank, shard_.world);
    std::vector<double> mine;
    for (int i = shard_.rank; i < nk; i += shard_.world) mine.push_back(k_[0][i]);
    if (!rc) rc = cpt_perturb_solve_batch(h_, mine.data(), (int)mine.size(), tau_sampling_, tau_size_, nullptr, stats_, nullptr);
    if (!rc) rc = cpt_allgather_sources(h_, nk, tau_size_);
    if (!rc) rc = cpt_get_sources(h_, d_src);
  } else
    rc = cpt_perturb_solve_batch(h_, k_[0], nk, tau_sampling_, tau_size_, d_src, stats_, nullptr);
  if (rc) {
    snprintf(error_message_, sizeof(error_message_), "%s", cpt_last_error(h_));
    (void)hipFree(d_src);
    raise(rc, error_message_);
  }
  // sources_[md][ic*tp+tp][tau*k_size+k]: one malloc per type like the reference (pm.cpp:1597-1616)
  sources_ = xalloc<double**>(1);
  sources_[0] = xalloc<double*>(ntp);
  for (int tp = 0; tp < ntp; tp++) {
    sources_[0][tp] = xalloc<double>((size_t)tau_size_ * nk);
    if (hipMemcpy(sources_[0][tp], d_src + (size_t)tp * tau_size_ * nk, sizeof(double) * (size_t)tau_size_ * nk,
                  hipMemcpyDeviceToHost) != hipSuccess) {
      (void)hipFree(d_src);
      raise(CPT_ERR_RUNTIME, "hipMemcpy of the sources failed");
    }
  }
  (void)hipFree(d_src);
}

PerturbationsModule::~PerturbationsModule() {
  if (sources_) {
    for (int tp = 0; tp < tp_size_[0]; tp++) free(sources_[0][tp]);
    free(sources_[0]);
    free(sources_);
  }
  if (k_) { free(k_[0]); free(k_); }
  free(k_size_); free(k_size_cl_); free(k_size_cmb_); free(tau_sampling_); free(ic_size_); free(tp_size_); free(stats_);
  cpt_destroy(h_);
}

double PerturbationsModule::kernel_ms() const {
  double ms = 0; int n = 0;
  cpt_last_kernel_ms(h_, 0, &ms, &n);
  return ms;
}

TransferModule::TransferModule(const Inputs& in, std::shared_ptr<const PerturbationsModule> pt)
    : perturbations_module_(std::move(pt)) {
  error_message_[0] = '\n';
  const cpt_config& c = in.config;
  const PerturbationsModule& P = *perturbations_module_;
  // ---- transfer_indices_of_transfers (tm.cpp:402-540) ----
  tt_size_ = xalloc<int>(1); tt_size_[0] = c.tt_size;
  index_tt_t0_ = c.index_tt_t0; index_tt_t1_ = c.index_tt_t1; index_tt_t2_ = c.index_tt_t2; index_tt_e_ = c.index_tt_e;
  index_tt_lcmb_ = c.index_tt_lcmb; index_tt_b_ = c.index_tt_b;
  std::vector<double> tmp(1 << 22);
  std::vector<int> itmp(1 << 16);
  int nl = 0;
  int rc = cpt_host_l_list(&c, &in.grid, itmp.data(), (int)itmp.size(), &nl);
  if (rc) { snprintf(error_message_, sizeof(error_message_), "%s", cpt_host_error()); raise(rc, error_message_); }
  l_size_max_ = nl;
  l_ = xalloc<int>(nl);
  memcpy(l_, itmp.data(), sizeof(int) * nl);
  l_size_ = xalloc<int>(1); l_size_[0] = nl;
  l_size_tt_ = xalloc<int*>(1);
  l_size_tt_[0] = xalloc<int>(c.tt_size);
  for (int t = 0; t < c.tt_size; t++) l_size_tt_[0][t] = nl;  // every CMB type runs to l_scalar_max (tm.cpp:790-870)
  rc = cpt_host_q_list(&c, &in.grid, P.k_min_, P.k_[0][P.k_size_cl_[0] - 1], tmp.data(), (int)tmp.size(), &q_size_);
  if (rc) { snprintf(error_message_, sizeof(error_message_), "%s", cpt_host_error()); raise(rc, error_message_); }
  q_ = xalloc<double>(q_size_);
  memcpy(q_, tmp.data(), sizeof(double) * q_size_);
  k_ = xalloc<double*>(1);
  k_[0] = xalloc<double>(q_size_);
  // transfer_get_k_list (tm.cpp:1106-1167): k^2 = q^2 - K (1 + m), m = 0 / 2 for scalars / tensors; flat space: k = q
  const double Km = c.K * (c.mode == CPT_MODE_TENSORS ? 3. : 1.);
  for (int i = 0; i < q_size_; i++) k_[0][i] = (c.sgnK == 0) ? q_[i] : sqrt(q_[i] * q_[i] - Km);
  // first wavenumber treated with the flat rescaling approximation (tm.cpp:1078-1090)
  index_q_flat_approximation_ = 0;
  if (c.sgnK != 0) {
    const double q_approximation = c.hyper_flat_approximation_nu * sqrt(c.sgnK * c.K);
    for (index_q_flat_approximation_ = 0; index_q_flat_approximation_ < q_size_ - 1; index_q_flat_approximation_++)
      if (q_[index_q_flat_approximation_] > q_approximation) break;
  }
  // ---- the q loop (tm.cpp:287-318) on the GPU, from the sources left resident in HBM by the perturbation stage ----
  const size_t ntr = (size_t)c.tt_size * nl * q_size_;
  double* d_tr = nullptr;
  if (hipMalloc((void**)&d_tr, ntr * sizeof(double)) != hipSuccess) raise(CPT_ERR_RUNTIME, "hipMalloc failed");
  transfer_ = xalloc<double*>(1);
  transfer_[0] = xalloc<double>(ntr);
  const Shard& sh = P.shard_;
  hipError_t e = hipSuccess;
  if (sh.comm_id) {
    // this rank's share of the multipoles, then exchange 2: the full table lands on rank 0
    std::vector<int> mine;
    for (int i = sh.rank; i < nl; i += sh.world) mine.push_back(l_[i]);
    const int nl_local = (int)mine.size();
    double* d_local = nullptr;
    if (hipMalloc((void**)&d_local, (size_t)c.tt_size * nl_local * q_size_ * sizeof(double)) != hipSuccess) { (void)hipFree(d_tr); raise(CPT_ERR_RUNTIME, "hipMalloc failed"); }
    rc = cpt_transfer_batch(P.handle(), nullptr, P.k_[0], P.k_size_[0], P.k_size_cl_[0], P.tau_sampling_, P.tau_size_, q_, q_size_, mine.data(), nl_local, d_local);
    if (!rc) rc = cpt_gather_transfer(P.handle(), d_local, nl, q_size_, sh.rank == 0 ? d_tr : nullptr);
    if (!rc) {
      if (sh.rank == 0) e = hipMemcpy(transfer_[0], d_tr, ntr * sizeof(double), hipMemcpyDeviceToHost);
      else {   // the other ranks keep their own rows
        memset(transfer_[0], 0, ntr * sizeof(double));
        std::vector<double> local((size_t)c.tt_size * nl_local * q_size_);
        e = hipMemcpy(local.data(), d_local, local.size() * sizeof(double), hipMemcpyDeviceToHost);
        for (int t = 0; t < c.tt_size; t++)
          for (int j = 0; j < nl_local; j++)
            memcpy(transfer_[0] + ((size_t)t * nl + (sh.rank + (size_t)j * sh.world)) * q_size_, local.data() + ((size_t)t * nl_local + j) * q_size_, sizeof(double) * q_size_);
      }
    }
    (void)hipFree(d_local);
  } else {
    rc = cpt_transfer_batch(P.handle(), nullptr, P.k_[0], P.k_size_[0], P.k_size_cl_[0], P.tau_sampling_, P.tau_size_, q_, q_size_,
                            l_, nl, d_tr);
    if (!rc) e = hipMemcpy(transfer_[0], d_tr, ntr * sizeof(double), hipMemcpyDeviceToHost);
  }
  (void)hipFree(d_tr);
  if (rc) {
    snprintf(error_message_, sizeof(error_message_), "%s", cpt_last_error(P.handle()));
    raise(rc, error_message_);
  }
  if (e != hipSuccess) raise(CPT_ERR_RUNTIME, "hipMemcpy of the transfer functions failed");
}

TransferModule::~TransferModule() {
  if (transfer_) { free(transfer_[0]); free(transfer_); }
  if (k_) { free(k_[0]); free(k_); }
  if (l_size_tt_) { free(l_size_tt_[0]); free(l_size_tt_); }
  free(q_); free(l_); free(l_size_); free(tt_size_);
}

double TransferModule::kernel_ms() const {
  double ms = 0; int n = 0;
  cpt_last_kernel_ms(perturbations_module_->handle(), 1, &ms, &n);
  return ms;
}

}  // namespace cpt
