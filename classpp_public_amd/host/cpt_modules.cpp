// C++ shim classes of include/cpt_modules.hpp: host logic only (grids, memory, error mapping); the compute is in
// libcpt.so behind the C ABI.  Link: g++ ... cpt_modules.cpp cpt_grids.cpp -L../csrc -lcpt -lamdhip64
#include "../../include/cpt_modules.hpp"

#include <hip/hip_runtime_api.h>

#include <algorithm>
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
  T* p = (T*)calloc(n ? n : 1, sizeof(T));   // (zeroed: release() may meet a table of pointers whose rows were never allocated)
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

// All work happens in the constructor, and it may throw half way (a device handle or two created, some tables allocated): the destructor of a
// partly constructed object never runs, so the constructor itself releases what it had built before it lets the exception out.
PerturbationsModule::PerturbationsModule(const Inputs& in) {
  try { build(in); } catch (...) { release(); throw; }
}

void PerturbationsModule::build(const Inputs& in) {
  error_message_[0] = '\n';
  const cpt_config& c0 = in.config;
  // ---- perturb_indices_of_perturbs (pm.cpp:843-1235): modes, initial conditions, source types ----
  const bool first_is_tensors = c0.mode == CPT_MODE_TENSORS;
  if (in.with_tensors && first_is_tensors) raise(CPT_ERR_INVALID, "Inputs: with_tensors needs the scalar mode in config and the tensor mode in config_tensors");
  if (in.with_tensors && in.config_tensors.mode != CPT_MODE_TENSORS) raise(CPT_ERR_INVALID, "Inputs: config_tensors.mode must be CPT_MODE_TENSORS");
  if (in.n_ic < 1 || in.n_ic > 5 || (first_is_tensors && in.n_ic != 1)) raise(CPT_ERR_INVALID, "Inputs: n_ic outside 1..5 (tensor modes have one initial condition)");
  for (int i = 1; i < in.n_ic; i++)
    if (in.ic[i] <= in.ic[i - 1] || in.ic[i] > CPT_IC_NIV) raise(CPT_ERR_INVALID, "Inputs: ic[] must list distinct initial conditions in the order ad, bi, cdi, nid, niv");
  has_scalars_ = !first_is_tensors; has_tensors_ = first_is_tensors || in.with_tensors;
  md_size_ = in.with_tensors ? 2 : 1;
  index_md_scalars_ = 0; index_md_tensors_ = in.with_tensors ? 1 : 0;
  shard_ = in.shard;
  if (shard_.world < 1 || shard_.rank < 0 || shard_.rank >= shard_.world) raise(CPT_ERR_INVALID, "Shard: rank outside [0, world)");
  if (shard_.comm_id && (md_size_ > 1 || in.n_ic > 1))
    raise(CPT_ERR_UNSUPPORTED, "Shard: one mode and one initial condition per sharded module (a communicator belongs to one device handle)");
  ic_size_ = xalloc<int>(md_size_); tp_size_ = xalloc<int>(md_size_);
  k_size_ = xalloc<int>(md_size_); k_size_cl_ = xalloc<int>(md_size_); k_size_cmb_ = xalloc<int>(md_size_);
  k_ = xalloc<double*>(md_size_);
  sources_ = xalloc<double**>(md_size_);
  for (int md = 0; md < md_size_; md++) { k_[md] = nullptr; sources_[md] = nullptr; ic_size_[md] = tp_size_[md] = 0; }
  std::vector<int> ics(in.ic, in.ic + in.n_ic);
  if (in.n_ic == 1) ics[0] = c0.ic;
  if (has_scalars_) {
    for (int i = 0; i < (int)ics.size(); i++) {
      int* slot = ics[i] == CPT_IC_AD ? &index_ic_ad_ : ics[i] == CPT_IC_BI ? &index_ic_bi_ : ics[i] == CPT_IC_CDI ? &index_ic_cdi_
                  : ics[i] == CPT_IC_NID ? &index_ic_nid_ : &index_ic_niv_;
      *slot = i;
    }
  }
  index_tp_t0_ = c0.index_tp_t0; index_tp_t1_ = c0.index_tp_t1; index_tp_t2_ = c0.index_tp_t2; index_tp_p_ = c0.index_tp_p;
  index_tp_delta_m_ = c0.index_tp_delta_m; index_tp_phi_plus_psi_ = c0.index_tp_phi_plus_psi;
  index_tp_delta_cb_ = c0.has_ncdm ? c0.index_tp_delta_cb : -1;
  has_source_t_ = c0.index_tp_t2 >= 0; has_source_p_ = c0.index_tp_p >= 0; has_source_delta_m_ = c0.index_tp_delta_m >= 0;
  has_source_phi_plus_psi_ = c0.index_tp_phi_plus_psi >= 0;
  std::vector<double> tmp(1 << 20);
  // ---- perturb_timesampling_for_sources (pm.cpp:1247-1619): one sampling for every mode ----
  int rc = cpt_host_tau_sampling(&c0, &in.tables, &in.grid, tmp.data(), (int)tmp.size(), &tau_size_);
  if (rc) { snprintf(error_message_, sizeof(error_message_), "%s", cpt_host_error()); raise(rc, error_message_); }
  tau_sampling_ = xalloc<double>(tau_size_);
  memcpy(tau_sampling_, tmp.data(), sizeof(double) * tau_size_);
  // the tail of the sampling kept for P(k, z) at 0 <= z <= z_max_pk (pm.cpp:1554-1592)
  rc = cpt_host_ln_tau_size(tau_sampling_, tau_size_, in.grid.tau_of_z_max_pk, &ln_tau_size_);
  if (rc) { snprintf(error_message_, sizeof(error_message_), "%s", cpt_host_error()); raise(rc, error_message_); }
  if (ln_tau_size_ > 1) {
    ln_tau_ = xalloc<double>(ln_tau_size_);
    for (int i = 0; i < ln_tau_size_; i++) ln_tau_[i] = log(tau_sampling_[i - ln_tau_size_ + tau_size_]);
  }
  k_min_ = 1e300; k_max_ = 0.;
  for (int md = 0; md < md_size_; md++) {
    const cpt_config& cm = (md == 0) ? c0 : in.config_tensors;
    const bool tensors = cm.mode == CPT_MODE_TENSORS;
    ic_size_[md] = tensors ? 1 : (int)ics.size();
    tp_size_[md] = cm.tp_size;
    // ---- perturb_get_k_list (pm.cpp:1628-2238) ----
    rc = cpt_host_k_list(&cm, &in.grid, tmp.data(), (int)tmp.size(), &k_size_[md], &k_size_cl_[md], &k_size_cmb_[md]);
    if (rc) { snprintf(error_message_, sizeof(error_message_), "%s", cpt_host_error()); raise(rc, error_message_); }
    k_[md] = xalloc<double>(k_size_[md]);
    memcpy(k_[md], tmp.data(), sizeof(double) * k_size_[md]);
    k_min_ = std::min(k_min_, k_[md][0]);
    k_max_ = std::max(k_max_, k_[md][k_size_[md] - 1]);
    // ---- the k loop (pm.cpp:668-718) on the GPU: one launch per initial condition ----
    const int nk = k_size_[md], ntp = cm.tp_size;
    const size_t nsrc = (size_t)ntp * tau_size_ * nk;
    sources_[md] = xalloc<double*>((size_t)ic_size_[md] * ntp);
    for (int i = 0; i < ic_size_[md] * ntp; i++) sources_[md][i] = nullptr;
    double* d_src = nullptr;
    if (hipMalloc((void**)&d_src, nsrc * sizeof(double)) != hipSuccess) raise(CPT_ERR_RUNTIME, "hipMalloc failed");
    for (int ic = 0; ic < ic_size_[md]; ic++) {
      cpt_config ci = cm;
      if (!tensors) ci.ic = ics[ic];
      cpt_handle*& h = h_[md][ic];
      rc = cpt_create(&ci, &in.tables, &h);
      if (rc) { snprintf(error_message_, sizeof(error_message_), "%s", cpt_create_error()); (void)hipFree(d_src); raise(rc, error_message_); }
      cpt_stepstat* st = nullptr;
      if (md == 0 && ic == 0) {
        stats_ = xalloc<cpt_stepstat>(nk);
        memset(stats_, 0, sizeof(cpt_stepstat) * nk);
        st = stats_;
      }
      if (shard_.comm_id) {
        // this rank's share of the k loop, then exchange 1: afterwards the handle holds the sources of every mode
        rc = cpt_comm_init(h, shard_.comm_id, shard_.rank, shard_.world);
        std::vector<double> mine;
        for (int i = shard_.rank; i < nk; i += shard_.world) mine.push_back(k_[md][i]);
        if (!rc) rc = cpt_perturb_solve_batch(h, mine.data(), (int)mine.size(), tau_sampling_, tau_size_, nullptr, st, nullptr);
        if (!rc) rc = cpt_allgather_sources(h, nk, tau_size_);
        if (!rc) rc = cpt_get_sources(h, d_src);
      } else
        rc = cpt_perturb_solve_batch(h, k_[md], nk, tau_sampling_, tau_size_, d_src, st, nullptr);
      if (rc) {
        snprintf(error_message_, sizeof(error_message_), "%s", cpt_last_error(h));
        (void)hipFree(d_src);
        raise(rc, error_message_);
      }
      // sources_[md][ic*tp_size+tp][tau*k_size+k]: one malloc per type like the reference (pm.cpp:1597-1616)
      for (int tp = 0; tp < ntp; tp++) {
        double*& dst = sources_[md][ic * ntp + tp];
        dst = xalloc<double>((size_t)tau_size_ * nk);
        if (hipMemcpy(dst, d_src + (size_t)tp * tau_size_ * nk, sizeof(double) * (size_t)tau_size_ * nk, hipMemcpyDeviceToHost) != hipSuccess) {
          (void)hipFree(d_src);
          raise(CPT_ERR_RUNTIME, "hipMemcpy of the sources failed");
        }
      }
    }
    (void)hipFree(d_src);
  }
}

PerturbationsModule::~PerturbationsModule() { release(); }

// frees whatever has been built so far (every pointer is null until its allocation succeeded; safe to call twice)
void PerturbationsModule::release() noexcept {
  if (sources_ && ic_size_ && tp_size_) {
    for (int md = 0; md < md_size_; md++) {
      if (!sources_[md]) continue;
      for (int i = 0; i < ic_size_[md] * tp_size_[md]; i++) free(sources_[md][i]);
      free(sources_[md]);
    }
  }
  free(sources_); sources_ = nullptr;
  if (k_) { for (int md = 0; md < md_size_; md++) free(k_[md]); free(k_); k_ = nullptr; }
  free(k_size_); free(k_size_cl_); free(k_size_cmb_); free(tau_sampling_); free(ln_tau_); free(ic_size_); free(tp_size_); free(stats_);
  k_size_ = k_size_cl_ = k_size_cmb_ = ic_size_ = tp_size_ = nullptr; tau_sampling_ = ln_tau_ = nullptr; stats_ = nullptr;
  for (auto& row : h_) for (cpt_handle*& h : row) { if (h) cpt_destroy(h); h = nullptr; }   // (also leaves the RCCL communicator of a sharded module)
}

double PerturbationsModule::kernel_ms() const {
  double tot = 0;
  for (auto& row : h_)
    for (cpt_handle* h : row) {
      if (!h) continue;
      double ms = 0; int n = 0;
      cpt_last_kernel_ms(h, 0, &ms, &n);
      tot += ms;
    }
  return tot;
}

TransferModule::TransferModule(const Inputs& in, std::shared_ptr<const PerturbationsModule> pt)
    : perturbations_module_(std::move(pt)) {
  try { build(in); } catch (...) { release(); throw; }   // (see PerturbationsModule)
}

void TransferModule::build(const Inputs& in) {
  error_message_[0] = '\n';
  const cpt_config& c0 = in.config;
  const PerturbationsModule& P = *perturbations_module_;
  const int nmd = P.md_size_;
  // ---- transfer_indices_of_transfers (tm.cpp:402-540) ----
  tt_size_ = xalloc<int>(nmd); l_size_ = xalloc<int>(nmd);
  l_size_tt_ = xalloc<int*>(nmd); k_ = xalloc<double*>(nmd); transfer_ = xalloc<double*>(nmd);
  for (int md = 0; md < nmd; md++) { l_size_tt_[md] = nullptr; k_[md] = nullptr; transfer_[md] = nullptr; tt_size_[md] = l_size_[md] = 0; }
  const cpt_config& ct = in.with_tensors ? in.config_tensors : c0;
  index_tt_t0_ = c0.index_tt_t0; index_tt_t1_ = c0.index_tt_t1; index_tt_t2_ = c0.index_tt_t2; index_tt_e_ = c0.index_tt_e;
  index_tt_lcmb_ = c0.index_tt_lcmb; index_tt_b_ = ct.index_tt_b;
  std::vector<double> tmp(1 << 22);
  std::vector<int> itmp(1 << 16);
  // ---- transfer_get_l_list (tm.cpp:694-876): one list up to the largest l_max of the modes; a mode with a smaller l_max stops at the
  //      first multipole >= its l_max plus two more (tm.cpp:858-866) ----
  int rc = 0;
  {
    int l_max_all = 0, md_of_max = 0;
    for (int md = 0; md < nmd; md++) {
      const cpt_config& cm = (md == 0) ? c0 : in.config_tensors;
      const int lm = (cm.mode == CPT_MODE_TENSORS) ? in.grid.l_tensor_max : in.grid.l_scalar_max;
      if (lm > l_max_all) { l_max_all = lm; md_of_max = md; }
    }
    const cpt_config& cm = (md_of_max == 0) ? c0 : in.config_tensors;
    int nl = 0;
    rc = cpt_host_l_list(&cm, &in.grid, itmp.data(), (int)itmp.size(), &nl);
    if (rc) { snprintf(error_message_, sizeof(error_message_), "%s", cpt_host_error()); raise(rc, error_message_); }
    l_size_max_ = nl;
    l_ = xalloc<int>(nl);
    memcpy(l_, itmp.data(), sizeof(int) * nl);
  }
  for (int md = 0; md < nmd; md++) {
    const cpt_config& cm = (md == 0) ? c0 : in.config_tensors;
    const int lm = (cm.mode == CPT_MODE_TENSORS) ? in.grid.l_tensor_max : in.grid.l_scalar_max;
    int index_l = 0;
    while (l_[index_l] < lm) index_l++;
    l_size_[md] = std::min(index_l + 3, l_size_max_);
    tt_size_[md] = cm.tt_size;
    l_size_tt_[md] = xalloc<int>(cm.tt_size);
    for (int t = 0; t < cm.tt_size; t++) l_size_tt_[md][t] = l_size_[md];  // every CMB type runs to the mode's l_max
  }
  // ---- transfer_get_q_list (tm.cpp:884-1096): one list for every mode, up to the largest k of the C_l's ----
  double k_max_cl = 0.;
  for (int md = 0; md < nmd; md++) k_max_cl = std::max(k_max_cl, P.k_[md][P.k_size_cl_[md] - 1]);
  {
    cpt_config cq = c0;   // (open space: q_max is limited by the tensor relation k^2 = q^2 - 3 K when tensors are present, tm.cpp:921-925)
    if (P.has_tensors_ && c0.sgnK < 0) cq.mode = CPT_MODE_TENSORS;
    rc = cpt_host_q_list(&cq, &in.grid, P.k_min_, k_max_cl, tmp.data(), (int)tmp.size(), &q_size_);
  }
  if (rc) { snprintf(error_message_, sizeof(error_message_), "%s", cpt_host_error()); raise(rc, error_message_); }
  q_ = xalloc<double>(q_size_);
  memcpy(q_, tmp.data(), sizeof(double) * q_size_);
  // first wavenumber treated with the flat rescaling approximation (tm.cpp:1078-1090)
  index_q_flat_approximation_ = 0;
  if (c0.sgnK != 0) {
    const double q_approximation = c0.hyper_flat_approximation_nu * sqrt(c0.sgnK * c0.K);
    for (index_q_flat_approximation_ = 0; index_q_flat_approximation_ < q_size_ - 1; index_q_flat_approximation_++)
      if (q_[index_q_flat_approximation_] > q_approximation) break;
  }
  const Shard& sh = P.shard_;
  for (int md = 0; md < nmd; md++) {
    const cpt_config& cm = (md == 0) ? c0 : in.config_tensors;
    const int nl = l_size_[md], ntt = cm.tt_size, nic = P.ic_size_[md];
    // transfer_get_k_list (tm.cpp:1106-1167): k^2 = q^2 - K (1 + m), m = 0 / 2 for scalars / tensors; flat space: k = q
    k_[md] = xalloc<double>(q_size_);
    const double Km = cm.K * (cm.mode == CPT_MODE_TENSORS ? 3. : 1.);
    for (int i = 0; i < q_size_; i++) k_[md][i] = (cm.sgnK == 0) ? q_[i] : sqrt(std::max(q_[i] * q_[i] - Km, 0.));
    // ---- the q loop (tm.cpp:287-318) on the GPU, from the sources left resident in HBM by the perturbation stage ----
    const size_t ntr = (size_t)ntt * nl * q_size_;
    double* d_tr = nullptr;
    if (hipMalloc((void**)&d_tr, ntr * sizeof(double)) != hipSuccess) raise(CPT_ERR_RUNTIME, "hipMalloc failed");
    transfer_[md] = xalloc<double>(ntr * nic);
    for (int ic = 0; ic < nic; ic++) {
      cpt_handle* h = P.handle(md, ic);
      double* dst = transfer_[md] + (size_t)ic * ntr;
      hipError_t e = hipSuccess;
      if (sh.comm_id) {
        // this rank's share of the multipoles, then exchange 2: the full table lands on rank 0
        std::vector<int> mine;
        for (int i = sh.rank; i < nl; i += sh.world) mine.push_back(l_[i]);
        const int nl_local = (int)mine.size();
        double* d_local = nullptr;
        if (hipMalloc((void**)&d_local, (size_t)ntt * nl_local * q_size_ * sizeof(double)) != hipSuccess) { (void)hipFree(d_tr); raise(CPT_ERR_RUNTIME, "hipMalloc failed"); }
        rc = cpt_transfer_batch(h, nullptr, P.k_[md], P.k_size_[md], P.k_size_cl_[md], P.tau_sampling_, P.tau_size_, q_, q_size_, mine.data(), nl_local, d_local);
        if (!rc) rc = cpt_gather_transfer(h, d_local, nl, q_size_, sh.rank == 0 ? d_tr : nullptr);
        if (!rc) {
          if (sh.rank == 0) e = hipMemcpy(dst, d_tr, ntr * sizeof(double), hipMemcpyDeviceToHost);
          else {   // the other ranks keep their own rows
            memset(dst, 0, ntr * sizeof(double));
            std::vector<double> local((size_t)ntt * nl_local * q_size_);
            e = hipMemcpy(local.data(), d_local, local.size() * sizeof(double), hipMemcpyDeviceToHost);
            for (int t = 0; t < ntt; t++)
              for (int j = 0; j < nl_local; j++)
                memcpy(dst + ((size_t)t * nl + (sh.rank + (size_t)j * sh.world)) * q_size_, local.data() + ((size_t)t * nl_local + j) * q_size_, sizeof(double) * q_size_);
          }
        }
        (void)hipFree(d_local);
      } else {
        rc = cpt_transfer_batch(h, nullptr, P.k_[md], P.k_size_[md], P.k_size_cl_[md], P.tau_sampling_, P.tau_size_, q_, q_size_, l_, nl, d_tr);
        if (!rc) e = hipMemcpy(dst, d_tr, ntr * sizeof(double), hipMemcpyDeviceToHost);
      }
      if (rc) {
        snprintf(error_message_, sizeof(error_message_), "%s", cpt_last_error(h));
        (void)hipFree(d_tr);
        raise(rc, error_message_);
      }
      if (e != hipSuccess) { (void)hipFree(d_tr); raise(CPT_ERR_RUNTIME, "hipMemcpy of the transfer functions failed"); }
    }
    (void)hipFree(d_tr);
  }
}

TransferModule::~TransferModule() { release(); }

void TransferModule::release() noexcept {
  const int nmd = perturbations_module_ ? perturbations_module_->md_size_ : 0;
  if (transfer_) { for (int md = 0; md < nmd; md++) free(transfer_[md]); free(transfer_); transfer_ = nullptr; }
  if (k_) { for (int md = 0; md < nmd; md++) free(k_[md]); free(k_); k_ = nullptr; }
  if (l_size_tt_) { for (int md = 0; md < nmd; md++) free(l_size_tt_[md]); free(l_size_tt_); l_size_tt_ = nullptr; }
  free(q_); free(l_); free(l_size_); free(tt_size_);
  q_ = nullptr; l_ = nullptr; l_size_ = tt_size_ = nullptr;
}

double TransferModule::kernel_ms() const {
  double tot = 0;
  for (int md = 0; md < perturbations_module_->md_size_; md++)
    for (int ic = 0; ic < perturbations_module_->ic_size_[md]; ic++) {
      double ms = 0; int n = 0;
      cpt_last_kernel_ms(perturbations_module_->handle(md, ic), 1, &ms, &n);
      tot += ms;
    }
  return tot;
}

}  // namespace cpt
