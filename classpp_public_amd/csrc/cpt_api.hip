// libcpt.so -- C ABI (include/cpt.h): handle life cycle, error reporting, table upload, layout transposes.
// The two batched stages live in cpt_perturb.hip (hot path A) and cpt_transfer.hip (hot path B).
#include "cpt_internal.h"

static thread_local std::string g_create_err;

int cpt_fail(cpt_handle* h, int code, const char* fmt, ...) {
  char buf[2048];  // same size as the reference's ErrorMsg (include/common.h)
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (h)
    h->err = buf;
  else
    g_create_err = buf;
  return code;
}

// ---------------------------------------------------------------------------------------------
// A handle is bound to the device that was current at cpt_create: every entry point runs with that device current and
// restores the caller's device on the way out (allocations and launches would otherwise land on whatever device the
// calling thread last selected).
struct DeviceGuard {
  int prev = -1;
  bool switched = false;
  explicit DeviceGuard(const cpt_handle* h) {
    if (h && hipGetDevice(&prev) == hipSuccess && prev != h->device) switched = (hipSetDevice(h->device) == hipSuccess);
  }
  ~DeviceGuard() {
    if (switched) (void)hipSetDevice(prev);
  }
  DeviceGuard(const DeviceGuard&) = delete;
  DeviceGuard& operator=(const DeviceGuard&) = delete;
};

// ---- pinned staging arena ----
void cpt_pin_reset(cpt_handle* h) {
  for (char* p : h->pin_retired) (void)hipHostFree(p);
  h->pin_retired.clear();
  h->pin_off = 0;
}
void* cpt_pin(cpt_handle* h, const void* src, size_t bytes) {
  const size_t need = (bytes + 63) & ~(size_t)63;
  if (h->pin_off + need > h->pin_cap) {
    // outgrown: copies enqueued earlier in this call may still read the old arena, so it is only retired here and freed at the
    // next reset
    size_t cap = h->pin_cap ? h->pin_cap * 2 : ((size_t)1 << 20);
    while (cap < need) cap *= 2;
    char* p = nullptr;
    if (hipHostMalloc((void**)&p, cap, hipHostMallocDefault) != hipSuccess) return nullptr;
    if (h->pin) h->pin_retired.push_back(h->pin);
    h->pin = p; h->pin_cap = cap; h->pin_off = 0;
  }
  char* dst = h->pin + h->pin_off;
  h->pin_off += need;
  if (src) memcpy(dst, src, bytes);
  return dst;
}
int cpt_upload(cpt_handle* h, void* dst_dev, const void* src_host, size_t bytes) {
  if (!bytes) return CPT_OK;
  void* staged = cpt_pin(h, src_host, bytes);
  if (!staged) return cpt_fail(h, CPT_ERR_NO_DEVICE, "hipHostMalloc of the staging arena failed");
  CPT_HIP(h, hipMemcpyAsync(dst_dev, staged, bytes, hipMemcpyHostToDevice, h->stream));
  return CPT_OK;
}
void cpt_timer_start(cpt_handle* h, int which) {
  Timer& t = h->timers[which];
  // (armed only by the matching cpt_timer_stop: a stage that returns early with an error leaves no half-recorded pair for cpt_finish to read)
  t.armed = false;
  t.started = hipEventRecord(t.a, h->stream) == hipSuccess;
}
void cpt_timer_stop(cpt_handle* h, int which) {
  Timer& t = h->timers[which];
  t.armed = t.started && hipEventRecord(t.b, h->stream) == hipSuccess;
  t.started = false;
}
// drain the stream (unless a fused cpt_step is collecting several stages) and read back what the stages left pending
int cpt_finish(cpt_handle* h) {
  if (h->defer) return CPT_OK;
  CPT_HIP(h, hipStreamSynchronize(h->stream));
  for (int i = 0; i < CPT_T_N; i++) {
    Timer& t = h->timers[i];
    if (!t.armed) continue;
    float ms = 0;
    if (hipEventElapsedTime(&ms, t.a, t.b) == hipSuccess) { t.ms = ms; t.launches = 1; }
    t.armed = false;
  }
  if (h->pend_work) {
    const unsigned long long* w = (const unsigned long long*)(h->pin_out);
    h->work_integrals = (long long)w[0]; h->work_samples = (long long)w[1]; h->work_fused = (long long)w[2];
    h->pend_work = false;
  }
  return CPT_OK;
}

// ---------------------------------------------------------------------------------------------
// layout transposes between the reference's [tp][tau][k] and the resident k-major [tp][k][tau]
// (tile through LDS so that both the read and the write are coalesced)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_transpose(const double* __restrict__ in, double* __restrict__ out, int rows,
                                                   int cols) {
  // in: [batch][rows][cols] -> out: [batch][cols][rows]; 32x32 tiles, +1 padding against bank conflicts
  __shared__ double tile[32][33];
  const size_t base = (size_t)blockIdx.z * rows * cols;
  int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int j = ty; j < 32; j += 8) {
    int r = r0 + j, c = c0 + tx;
    if (r < rows && c < cols) tile[j][tx] = in[base + (size_t)r * cols + c];
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    int c = c0 + j, r = r0 + tx;
    if (r < rows && c < cols) out[base + (size_t)c * rows + r] = tile[tx][j];
  }
}

static int launch_transpose(cpt_handle* h, const double* in, double* out, int batch, int rows, int cols) {
  dim3 grid((cols + 31) / 32, (rows + 31) / 32, batch);
  hipLaunchKernelGGL(k_transpose, grid, dim3(256), 0, h->stream, in, out, rows, cols);
  CPT_HIP(h, hipGetLastError());
  return CPT_OK;
}

int cpt_transpose_to_kmajor(cpt_handle* h, const double* src, double* dst, int ntp, int ntau, int nk) {
  return launch_transpose(h, src, dst, ntp, ntau, nk);
}
int cpt_transpose_from_kmajor(cpt_handle* h, const double* src, double* dst, int ntp, int ntau, int nk) {
  return launch_transpose(h, src, dst, ntp, nk, ntau);
}

// ---------------------------------------------------------------------------------------------
static int validate(const cpt_config* c) {
  // physics branches of the reference that this backend does not implement (SURVEY.md S8f "not planned"/later)
  if (c->ic < CPT_IC_AD || c->ic > CPT_IC_NIV) return cpt_fail(nullptr, CPT_ERR_INVALID, "ic=%d is not an initial condition", c->ic);
  if (c->ic == CPT_IC_CDI && !c->has_cdm)
    return cpt_fail(nullptr, CPT_ERR_INVALID, "not consistent to ask for CDI in absence of CDM! (pm.cpp:4961)");
  if ((c->ic == CPT_IC_NID || c->ic == CPT_IC_NIV) && !c->has_ur)
    return cpt_fail(nullptr, CPT_ERR_INVALID, "not consistent to ask for NID/NIV in absence of ur species! (pm.cpp:5024, 5050)");
  if ((c->sgnK == 0) != (c->K == 0.) || (c->sgnK != 0 && (c->sgnK > 0) != (c->K > 0.)))
    return cpt_fail(nullptr, CPT_ERR_INVALID, "inconsistent curvature: K=%g, sgnK=%d", c->K, c->sgnK);
  if (c->has_transfers) {
    if (c->mode == CPT_MODE_TENSORS) return cpt_fail(nullptr, CPT_ERR_INVALID, "density / velocity transfer sources belong to scalar modes (pm.cpp:1000)");
    for (int i = 0; i < CPT_NTK; i++)
      if (c->index_tp_transfer[i] >= c->tp_size) return cpt_fail(nullptr, CPT_ERR_INVALID, "index_tp_transfer[%d] >= tp_size", i);
    if (c->has_ncdm && (c->index_tp_delta_ncdm1 + c->N_ncdm > c->tp_size || c->index_tp_theta_ncdm1 + c->N_ncdm > c->tp_size))
      return cpt_fail(nullptr, CPT_ERR_INVALID, "index_tp_delta_ncdm1 / index_tp_theta_ncdm1 + N_ncdm > tp_size");
    if (c->index_tp_transfer[CPT_TK_THETA_CDM] >= 0 && c->gauge == CPT_GAUGE_SYNCHRONOUS)
      return cpt_fail(nullptr, CPT_ERR_INVALID, "theta_cdm is a source in the Newtonian gauge only (pm.cpp:1036)");
  }
  if (c->has_ncdm) {
    if (c->N_ncdm < 1 || c->N_ncdm > CPT_MAX_NCDM)
      return cpt_fail(nullptr, CPT_ERR_UNSUPPORTED, "N_ncdm=%d: between 1 and %d non-cold species per handle", c->N_ncdm, CPT_MAX_NCDM);
    if (c->mode == CPT_MODE_TENSORS) {
      if (c->tensor_method == CPT_TM_EXACT)
        return cpt_fail(nullptr, CPT_ERR_UNSUPPORTED, "tensor_method = exact with ncdm (tensor ncdm hierarchies, pm.cpp:9158-9201) is not implemented");
    } else {
      if (c->gauge != CPT_GAUGE_SYNCHRONOUS) return cpt_fail(nullptr, CPT_ERR_UNSUPPORTED, "ncdm perturbations are implemented in the synchronous gauge only");
      if (c->l_max_ncdm < 4) return cpt_fail(nullptr, CPT_ERR_INVALID, "ppr->l_max_ncdm=%d should be at least 4 (pm.cpp:3451)", c->l_max_ncdm);
      if (c->l_max_ncdm + 1 > CPT_WAVE / 2) return cpt_fail(nullptr, CPT_ERR_UNSUPPORTED, "l_max_ncdm=%d: a momentum bin must fit half a wavefront", c->l_max_ncdm);
      if (c->ncdm_fluid_approximation < CPT_NCDMFA_MB || c->ncdm_fluid_approximation > CPT_NCDMFA_NONE)
        return cpt_fail(nullptr, CPT_ERR_INVALID, "ncdm_fluid_approximation=%d", c->ncdm_fluid_approximation);
      if (c->index_tp_delta_cb >= c->tp_size) return cpt_fail(nullptr, CPT_ERR_INVALID, "index_tp_delta_cb >= tp_size");
    }
  }
  if (c->has_fld) return cpt_fail(nullptr, CPT_ERR_UNSUPPORTED, "dark-energy fluid perturbations are not implemented");
  if (!c->has_cdm && c->gauge == CPT_GAUGE_SYNCHRONOUS)
    return cpt_fail(nullptr, CPT_ERR_INVALID,
                    "synchronous gauge needs cdm (the reference rejects this too, perturbations_module.cpp:560)");
  if (c->gauge != CPT_GAUGE_SYNCHRONOUS && c->gauge != CPT_GAUGE_NEWTONIAN)
    return cpt_fail(nullptr, CPT_ERR_INVALID, "gauge=%d is neither newtonian (0) nor synchronous (1)", c->gauge);
  if (c->tight_coupling_approximation != CPT_TCA_COMPROMISE_CLASS && c->tight_coupling_approximation != CPT_TCA_FIRST_ORDER_CAMB &&
      c->tight_coupling_approximation != CPT_TCA_FIRST_ORDER_MB)
    return cpt_fail(nullptr, CPT_ERR_UNSUPPORTED, "tight_coupling_approximation=%d is not implemented (first_order_MB, first_order_CAMB and "
                    "compromise_CLASS are; first_order_CLASS and the second-order schemes need the derivatives of c_b^2)", c->tight_coupling_approximation);
  if (c->l_max_g < 4 || c->l_max_pol_g < 4 || (c->has_ur && c->l_max_ur < 4))
    return cpt_fail(nullptr, CPT_ERR_INVALID, "l_max_g, l_max_pol_g, l_max_ur must be at least 4 (pm.cpp:3302-3330)");
  if (c->mode != CPT_MODE_SCALARS && c->mode != CPT_MODE_TENSORS) return cpt_fail(nullptr, CPT_ERR_INVALID, "mode=%d is neither scalars (0) nor tensors (1)", c->mode);
  if (c->mode == CPT_MODE_TENSORS) {
    if (c->l_max_g_ten < 4 || c->l_max_pol_g_ten < 4)
      return cpt_fail(nullptr, CPT_ERR_INVALID, "ppr->l_max_g_ten / l_max_pol_g_ten should be at least 4 (pm.cpp:3521-3527)");
    if (c->evolve_tensor_ur && !c->has_ur && !c->has_ncdm) return cpt_fail(nullptr, CPT_ERR_INVALID, "evolve_tensor_ur without ur species");
    // lane map of the tensor kernel: 17 core lanes + the three l >= 5 tails
    if (17 + (c->l_max_g_ten - 4) + (c->l_max_pol_g_ten - 4) + (c->evolve_tensor_ur ? c->l_max_ur - 4 : 0) > CPT_WAVE)
      return cpt_fail(nullptr, CPT_ERR_UNSUPPORTED, "tensor hierarchy too large: one wavefront (64 lanes) owns one k-mode");
    if (c->index_tp_t0 >= 0 || c->index_tp_t1 >= 0 || c->index_tp_delta_m >= 0 || c->index_tp_phi_plus_psi >= 0)
      return cpt_fail(nullptr, CPT_ERR_INVALID, "tensor modes have the source types t2 and p only (pm.cpp:7243-7280)");
  }
  // lane map of cpt_perturb.hip: 14 core lanes (22 with non-cold species) + the three l >= 3 tails
  // Longer hierarchies of the synchronous scalar system run with each l >= 3 tail as a register set of its own (cpt_perturb_sets.inc):
  // each tail must then fit the 64 lanes of a set.
  const int core_lanes = c->has_ncdm ? 13 + 3 * CPT_MAX_NCDM : 14;
  if (core_lanes + (c->l_max_g - 2) + (c->l_max_pol_g - 2) + (c->has_ur ? c->l_max_ur - 2 : 0) > CPT_WAVE) {
    const bool long_ok = c->mode == CPT_MODE_SCALARS && c->gauge == CPT_GAUGE_SYNCHRONOUS && c->l_max_g - 2 <= CPT_WAVE &&
                         c->l_max_pol_g - 2 <= CPT_WAVE && (!c->has_ur || c->l_max_ur - 2 <= CPT_WAVE);
    if (!long_ok)
      return cpt_fail(nullptr, CPT_ERR_UNSUPPORTED,
                      "hierarchy too large: %d + tails > 64 lanes; hierarchies longer than one wavefront run for synchronous-gauge scalars "
                      "(with or without non-cold species) and l_max_g, l_max_pol_g, l_max_ur <= 66 only", core_lanes);
  }
  if (c->tp_size < 1 || c->tp_size > 8 + CPT_NTK + 2 * CPT_MAX_NCDM) return cpt_fail(nullptr, CPT_ERR_INVALID, "tp_size=%d out of range", c->tp_size);
  const int tps[6] = {c->index_tp_t0, c->index_tp_t1, c->index_tp_t2, c->index_tp_p, c->index_tp_delta_m,
                      c->index_tp_phi_plus_psi};
  for (int i = 0; i < 6; i++)
    if (tps[i] >= c->tp_size) return cpt_fail(nullptr, CPT_ERR_INVALID, "index_tp_* >= tp_size");
  return CPT_OK;
}

extern "C" {

const char* cpt_create_error(void) { return g_create_err.c_str(); }
const char* cpt_last_error(const cpt_handle* h) { return h ? h->err.c_str() : ""; }

int cpt_create(const cpt_config* cfg, const cpt_tables* t, cpt_handle** out) {
  if (!cfg || !t || !out) return cpt_fail(nullptr, CPT_ERR_INVALID, "null argument");
  *out = nullptr;
  int rc = validate(cfg);
  if (rc) return rc;
  if (t->bt_size < 2 || t->tt_size < 2 || !t->tau_table || !t->background_table || !t->d2background_dtau2_table ||
      !t->z_table || !t->thermodynamics_table || !t->d2thermodynamics_dz2_table)
    return cpt_fail(nullptr, CPT_ERR_INVALID, "incomplete spline tables");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
    return cpt_fail(nullptr, CPT_ERR_NO_DEVICE, "no HIP device available (this backend has no CPU fallback)");
  cpt_handle* h = new cpt_handle();
  h->cfg = *cfg;
  auto bail = [&](int code) {
    g_create_err = h->err;
    cpt_destroy(h);
    return code;
  };
  if (hipGetDevice(&h->device) != hipSuccess) return bail(cpt_fail(h, CPT_ERR_NO_DEVICE, "hipGetDevice failed"));
  if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess)
    return bail(cpt_fail(h, CPT_ERR_NO_DEVICE, "hipStreamCreate failed"));

  // pack the columns the path reads, value / second derivative interleaved (see cpt_internal.h)
  const int bgmap[BG_NCOL] = {t->index_bg_a,     t->index_bg_H,       t->index_bg_H_prime, t->index_bg_rho_g,
                              t->index_bg_rho_b, t->index_bg_rho_cdm, t->index_bg_rho_ur};
  const int thmap[TH_NCOL] = {t->index_th_xe,       t->index_th_dkappa,      t->index_th_tau_d,
                              t->index_th_ddkappa,  t->index_th_dddkappa,    t->index_th_exp_m_kappa,
                              t->index_th_g,        t->index_th_dg,          t->index_th_cb2};
  std::vector<double> bg((size_t)t->bt_size * BG_NCOL * 2), th((size_t)t->tt_size * TH_NCOL * 2);
  for (int r = 0; r < t->bt_size; r++)
    for (int c = 0; c < BG_NCOL; c++) {
      bool have = bgmap[c] >= 0 && bgmap[c] < t->bg_size;
      bg[((size_t)r * BG_NCOL + c) * 2 + 0] = have ? t->background_table[(size_t)r * t->bg_size + bgmap[c]] : 0.;
      bg[((size_t)r * BG_NCOL + c) * 2 + 1] = have ? t->d2background_dtau2_table[(size_t)r * t->bg_size + bgmap[c]] : 0.;
    }
  for (int r = 0; r < t->tt_size; r++)
    for (int c = 0; c < TH_NCOL; c++) {
      bool have = thmap[c] >= 0 && thmap[c] < t->th_size;
      th[((size_t)r * TH_NCOL + c) * 2 + 0] = have ? t->thermodynamics_table[(size_t)r * t->th_size + thmap[c]] : 0.;
      th[((size_t)r * TH_NCOL + c) * 2 + 1] = have ? t->d2thermodynamics_dz2_table[(size_t)r * t->th_size + thmap[c]] : 0.;
    }
  // non-cold species: per-species columns {rho, p, pseudo_p} in a table of their own, sharing the background abscissa
  std::vector<double> ncb;
  if (cfg->has_ncdm) {
    if (t->index_bg_rho_ncdm1 < 0 || t->index_bg_p_ncdm1 < 0 || t->index_bg_pseudo_p_ncdm1 < 0 ||
        t->index_bg_pseudo_p_ncdm1 + cfg->N_ncdm > t->bg_size)
      return bail(cpt_fail(h, CPT_ERR_INVALID, "ncdm background columns missing from the tables"));
    for (int n = 0; n < cfg->N_ncdm; n++) {
      if (cfg->mode == CPT_MODE_SCALARS && (t->q_size_ncdm[n] < 1 || t->q_size_ncdm[n] > CPT_MAX_Q_NCDM || !t->q_ncdm[n] || !t->w_ncdm[n] || !t->dlnf0_dlnq_ncdm[n]))
        return bail(cpt_fail(h, CPT_ERR_UNSUPPORTED, "ncdm species %d: between 1 and %d momentum bins", n, CPT_MAX_Q_NCDM));
    }
    ncb.assign((size_t)t->bt_size * NCB_NCOL * 2, 0.);
    for (int r = 0; r < t->bt_size; r++)
      for (int n = 0; n < cfg->N_ncdm; n++) {
        const int cols[3] = {t->index_bg_rho_ncdm1 + n, t->index_bg_p_ncdm1 + n, t->index_bg_pseudo_p_ncdm1 + n};
        for (int j = 0; j < 3; j++) {
          ncb[((size_t)r * NCB_NCOL + 3 * n + j) * 2 + 0] = t->background_table[(size_t)r * t->bg_size + cols[j]];
          ncb[((size_t)r * NCB_NCOL + 3 * n + j) * 2 + 1] = t->d2background_dtau2_table[(size_t)r * t->bg_size + cols[j]];
        }
      }
    if (cfg->mode == CPT_MODE_TENSORS && cfg->tensor_method == CPT_TM_MASSLESS_APPROXIMATION) {
      // pm.cpp:6640-6657: the ur hierarchy of the tensor modes carries rho_relativistic = rho_ur + 3 sum_n p_ncdm_n.  A cubic
      // spline is linear in its ordinates, so the combined column's second derivatives are the combination of the columns'
      for (int r = 0; r < t->bt_size; r++)
        for (int n = 0; n < cfg->N_ncdm; n++)
          for (int j = 0; j < 2; j++) bg[((size_t)r * BG_NCOL + BG_RHO_UR) * 2 + j] += 3. * ncb[((size_t)r * NCB_NCOL + 3 * n + 1) * 2 + j];
    }
  }
  auto up = [&](double** d, const double* src, size_t n) -> bool {
    if (hipMalloc((void**)d, n * sizeof(double)) != hipSuccess) return false;
    return hipMemcpy(*d, src, n * sizeof(double), hipMemcpyHostToDevice) == hipSuccess;
  };
  if (!up(&h->d_tau_table, t->tau_table, t->bt_size) || !up(&h->d_bg, bg.data(), bg.size()) ||
      !up(&h->d_z_table, t->z_table, t->tt_size) || !up(&h->d_th, th.data(), th.size()))
    return bail(cpt_fail(h, CPT_ERR_NO_DEVICE, "table upload failed"));
  const char* force_long = getenv("CPT_LONG_TAILS");   // (diagnostic: run any synchronous scalar configuration on the long-hierarchy kernel)
  if (!cfg->has_ncdm && cfg->mode == CPT_MODE_SCALARS &&
      (14 + (cfg->l_max_g - 2) + (cfg->l_max_pol_g - 2) + (cfg->has_ur ? cfg->l_max_ur - 2 : 0) > CPT_WAVE || (force_long && atoi(force_long) != 0))) {
    // hierarchies longer than one wavefront run on the multi-wavefront kernels of the non-cold species with zero species: those read
    // the (then empty) ncdm columns of the background
    ncb.assign((size_t)t->bt_size * NCB_NCOL * 2, 0.);
    if (!up(&h->d_ncb, ncb.data(), ncb.size())) return bail(cpt_fail(h, CPT_ERR_NO_DEVICE, "table upload failed"));
    h->tabs.ncb = h->d_ncb;
  }
  if (cfg->has_ncdm) {
    if (!up(&h->d_ncb, ncb.data(), ncb.size())) return bail(cpt_fail(h, CPT_ERR_NO_DEVICE, "table upload failed"));
    h->tabs.ncb = h->d_ncb;
    NcdmDev& nd = h->ncdm;
    memset(&nd, 0, sizeof(nd));
    nd.n_species = cfg->N_ncdm; nd.lmax = cfg->l_max_ncdm;
    if (cfg->mode == CPT_MODE_SCALARS)
      for (int n = 0; n < cfg->N_ncdm; n++) {
        nd.M[n] = t->M_ncdm[n]; nd.factor[n] = t->factor_ncdm[n]; nd.first_chain[n] = nd.nchains;
        for (int iq = 0; iq < t->q_size_ncdm[n]; iq++) {
          const int c = nd.nchains++;
          nd.species[c] = n; nd.q[c] = t->q_ncdm[n][iq]; nd.w[c] = t->w_ncdm[n][iq]; nd.dlnf0[c] = t->dlnf0_dlnq_ncdm[n][iq];
        }
      }
    nd.first_chain[cfg->N_ncdm] = nd.nchains;
  }
  h->tabs.bt_size = t->bt_size;
  h->tabs.tt_size = t->tt_size;
  h->tabs.tau_table = h->d_tau_table;
  h->tabs.bg = h->d_bg;
  h->tabs.z_table = h->d_z_table;
  h->tabs.th = h->d_th;
  if (hipMalloc((void**)&h->d_work, 4 * sizeof(unsigned long long)) != hipSuccess)
    return bail(cpt_fail(h, CPT_ERR_NO_DEVICE, "hipMalloc failed"));
  for (Timer& tm : h->timers) {
    if (hipEventCreate(&tm.a) != hipSuccess || hipEventCreate(&tm.b) != hipSuccess)
      return bail(cpt_fail(h, CPT_ERR_NO_DEVICE, "hipEventCreate failed"));
  }
  // landing zone of the device -> host results (grown on demand by the perturbation stage)
  h->pin_out_cap = (size_t)1 << 16;
  if (hipHostMalloc((void**)&h->pin_out, h->pin_out_cap, hipHostMallocDefault) != hipSuccess)
    return bail(cpt_fail(h, CPT_ERR_NO_DEVICE, "hipHostMalloc failed"));
  *out = h;
  return CPT_OK;
}

void cpt_destroy(cpt_handle* h) {
  if (!h) return;
  DeviceGuard guard(h);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  // (d_splc and d_ik are interior pointers into d_k / d_q and must not be freed)
  void* ptrs[] = {h->d_ncb, h->d_tau_table, h->d_bg, h->d_z_table, h->d_th, h->d_src, h->d_dd, h->d_u, h->d_k, h->d_tau, h->d_q,
                  h->d_l, h->d_bes, h->d_chi_min, h->d_work, h->d_pt_scratch, h->d_lens, h->d_lens_w, h->d_lens_l, h->d_his, h->d_his_trig, h->d_his_desc, h->d_kq};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  for (Timer& tm : h->timers) {
    if (tm.a) (void)hipEventDestroy(tm.a);
    if (tm.b) (void)hipEventDestroy(tm.b);
  }
  (void)cpt_comm_destroy(h);
  if (h->d_xsend) (void)hipFree(h->d_xsend);
  if (h->d_xrecv) (void)hipFree(h->d_xrecv);
  if (h->d_clw) (void)hipFree(h->d_clw);
  if (h->d_pk_k) (void)hipFree(h->d_pk_k);
  if (h->d_pkz) (void)hipFree(h->d_pkz);
  cpt_pin_reset(h);
  if (h->pin) (void)hipHostFree(h->pin);
  if (h->pin_out) (void)hipHostFree(h->pin_out);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

// every compute entry point: the handle's device current, error cleared, staging arena recycled (the previous call has drained
// the stream), the stages enqueue, then ONE synchronisation in cpt_finish
#define CPT_ENTER(h)          \
  if (!(h)) return CPT_ERR_INVALID; \
  DeviceGuard guard__(h);     \
  (h)->err.clear();           \
  cpt_pin_reset(h)

static int check_transfer_args(cpt_handle* h, const double* k, int nk, int k_size_cl, const double* tau_sampling, int ntau, const double* q, int nq,
                               const int* l, int nl, const double* transfer_dev) {
  if (!k || !tau_sampling || !q || !l || !transfer_dev || nk < 3 || ntau < 3 || nq < 1 || nl < 1 || k_size_cl < 1 ||
      k_size_cl > nk)
    return cpt_fail(h, CPT_ERR_INVALID, "bad arguments to cpt_transfer_batch");
  if (h->cfg.tt_size < 1 || h->cfg.tt_size > 5) return cpt_fail(h, CPT_ERR_INVALID, "tt_size=%d out of range", h->cfg.tt_size);
  if (h->cfg.K > 0. && sqrt(h->cfg.K) * h->cfg.tau0 >= 1.5707963267948966 - h->cfg.hyper_x_min)
    return cpt_fail(h, CPT_ERR_UNSUPPORTED, "closed space with sqrt(K) tau0 >= pi/2: the folding of chi onto [0, pi/2] (ClosedModY, "
                    "hyperspherical.c:1025-1052) is not implemented");
  return CPT_OK;
}

int cpt_perturb_solve_batch(cpt_handle* h, const double* k, int nk, const double* tau_sampling, int ntau,
                            double* sources_dev, cpt_stepstat* stats, int* status) {
  CPT_ENTER(h);
  if (!k || !tau_sampling || nk < 1 || ntau < 2) return cpt_fail(h, CPT_ERR_INVALID, "bad k / tau_sampling arguments: at least one k-mode and two sampling times (the last one ends the integration)");
  int rc = cpt_perturb_impl(h, k, nk, tau_sampling, ntau, sources_dev, stats, status);
  if (rc) { (void)hipStreamSynchronize(h->stream); h->pend_nk = 0; return rc; }
  if ((rc = cpt_finish(h))) return rc;
  return cpt_perturb_collect(h, k, stats, status);
}

int cpt_transfer_batch(cpt_handle* h, const double* sources_dev, const double* k, int nk, int k_size_cl,
                       const double* tau_sampling, int ntau, const double* q, int nq, const int* l, int nl,
                       double* transfer_dev) {
  CPT_ENTER(h);
  int rc = check_transfer_args(h, k, nk, k_size_cl, tau_sampling, ntau, q, nq, l, nl, transfer_dev);
  if (rc) return rc;
  if ((rc = cpt_transfer_impl(h, sources_dev, k, nk, k_size_cl, tau_sampling, ntau, q, nq, l, nl, transfer_dev))) { (void)hipStreamSynchronize(h->stream); return rc; }
  return cpt_finish(h);
}

int cpt_cl_batch(cpt_handle* h, const cpt_spectra_params* sp, const double* transfer_dev, const double* q, int nq, int nl,
                 double* cl_dev) {
  CPT_ENTER(h);
  if (!sp || !transfer_dev || !q || !cl_dev || nq < 3 || nl < 1) return cpt_fail(h, CPT_ERR_INVALID, "bad arguments to cpt_cl_batch");
  int rc = cpt_cl_impl(h, sp, transfer_dev, q, nq, nl, cl_dev);
  if (rc) { (void)hipStreamSynchronize(h->stream); return rc; }
  return cpt_finish(h);
}

int cpt_cl_cross_batch(cpt_handle* h, const cpt_spectra_params* sp, const double* transfer1_dev, const double* transfer2_dev, const double* q,
                       int nq, int nl, double* cl_dev) {
  CPT_ENTER(h);
  if (!sp || !transfer1_dev || !transfer2_dev || !q || !cl_dev || nq < 3 || nl < 1) return cpt_fail(h, CPT_ERR_INVALID, "bad arguments to cpt_cl_cross_batch");
  int rc = cpt_cl_impl(h, sp, transfer1_dev, q, nq, nl, cl_dev, transfer2_dev);
  if (rc) { (void)hipStreamSynchronize(h->stream); return rc; }
  return cpt_finish(h);
}

int cpt_sigma_of_pk(const double* k, const double* pk, int nk, double R, double k_per_decade, double* sigma) {
  if (!k || !pk || !sigma || nk < 3 || !(R >= 0.) || !(k_per_decade > 0.)) return CPT_ERR_INVALID;
  for (int i = 0; i < nk; i++) if (!(pk[i] > 0.) || (i && !(k[i] > k[i - 1]))) return CPT_ERR_INVALID;
  *sigma = cpt_sigma_of_R(k, pk, nk, R, k_per_decade);
  return CPT_OK;
}

int cpt_sigma(cpt_handle* h, const cpt_spectra_params* sp, const double* k, int nk, double R, double k_per_decade, double* sigma) {
  CPT_ENTER(h);
  if (!sp || !k || !sigma || nk < 3 || !(R >= 0.) || !(k_per_decade > 0.)) return cpt_fail(h, CPT_ERR_INVALID, "bad arguments to cpt_sigma");
  return cpt_sigma_impl(h, sp, k, nk, R, k_per_decade, sigma, 0);
}

int cpt_sigma_cb(cpt_handle* h, const cpt_spectra_params* sp, const double* k, int nk, double R, double k_per_decade, double* sigma) {
  CPT_ENTER(h);
  if (!sp || !k || !sigma || nk < 3 || !(R >= 0.) || !(k_per_decade > 0.)) return cpt_fail(h, CPT_ERR_INVALID, "bad arguments to cpt_sigma_cb");
  return cpt_sigma_impl(h, sp, k, nk, R, k_per_decade, sigma, 1);
}

int cpt_pk_linear(cpt_handle* h, const cpt_spectra_params* sp, const double* k, int nk, double* pk_dev) {
  CPT_ENTER(h);
  if (!sp || !k || !pk_dev || nk < 1) return cpt_fail(h, CPT_ERR_INVALID, "bad arguments to cpt_pk_linear");
  int rc = cpt_pk_impl(h, sp, k, nk, pk_dev, 0);
  if (rc) { (void)hipStreamSynchronize(h->stream); return rc; }
  return cpt_finish(h);
}

int cpt_pk_cb_linear(cpt_handle* h, const cpt_spectra_params* sp, const double* k, int nk, double* pk_dev) {
  CPT_ENTER(h);
  if (!sp || !k || !pk_dev || nk < 1) return cpt_fail(h, CPT_ERR_INVALID, "bad arguments to cpt_pk_cb_linear");
  int rc = cpt_pk_impl(h, sp, k, nk, pk_dev, 1);
  if (rc) { (void)hipStreamSynchronize(h->stream); return rc; }
  return cpt_finish(h);
}

int cpt_pk_at_tau(cpt_handle* h, const cpt_spectra_params* sp, const double* k, int nk, int ln_tau_size, double tau_z, int cb, double* pk_dev) {
  CPT_ENTER(h);
  if (!sp || !k || !pk_dev || nk < 1 || !(tau_z > 0.)) return cpt_fail(h, CPT_ERR_INVALID, "bad arguments to cpt_pk_at_tau");
  int rc = cpt_pk_at_tau_impl(h, sp, k, nk, ln_tau_size, tau_z, cb, pk_dev);
  if (rc) { (void)hipStreamSynchronize(h->stream); return rc; }
  return cpt_finish(h);
}

int cpt_sigma_at_tau(cpt_handle* h, const cpt_spectra_params* sp, const double* k, int nk, int ln_tau_size, double tau_z, int cb, double R, double k_per_decade,
                     double* sigma) {
  CPT_ENTER(h);
  if (!sp || !k || !sigma || nk < 3 || !(R >= 0.) || !(k_per_decade > 0.) || !(tau_z > 0.)) return cpt_fail(h, CPT_ERR_INVALID, "bad arguments to cpt_sigma_at_tau");
  return cpt_sigma_at_tau_impl(h, sp, k, nk, ln_tau_size, tau_z, cb, R, k_per_decade, sigma);
}

int cpt_lensing_l_size(const int* l, int nl, const cpt_lensing_params* lp) {
  if (!l || !lp || nl < 1) return -1;
  return cpt_lensing_l_size_impl(l, nl, lp);
}

int cpt_lensing_batch(cpt_handle* h, const cpt_spectra_params* sp, const cpt_lensing_params* lp, const int* l, int nl,
                      const double* cl_dev, double* cl_lensed_dev) {
  CPT_ENTER(h);
  if (!sp || !lp || !l || !cl_dev || !cl_lensed_dev) return cpt_fail(h, CPT_ERR_INVALID, "bad arguments to cpt_lensing_batch");
  int rc = cpt_lensing_impl(h, sp, lp, l, nl, cl_dev, cl_lensed_dev);
  if (rc) { (void)hipStreamSynchronize(h->stream); return rc; }
  return cpt_finish(h);
}

// One pass for one cosmology - k-modes -> sources -> transfer functions -> C_l (-> lensed C_l) (-> P(k)) - enqueued back to back
// on the handle's stream with a single synchronisation at the end.
int cpt_step(cpt_handle* h, const cpt_step_io* io) {
  CPT_ENTER(h);
  if (!io || !io->k || !io->tau_sampling || !io->q || !io->l || !io->sp || !io->transfer_dev || !io->cl_dev || io->nk < 3 || io->ntau < 3)
    return cpt_fail(h, CPT_ERR_INVALID, "bad arguments to cpt_step");
  if (io->lp && !io->cl_lensed_dev) return cpt_fail(h, CPT_ERR_INVALID, "cpt_step: lensing parameters without an output buffer");
  int rc = check_transfer_args(h, io->k, io->nk, io->k_size_cl, io->tau_sampling, io->ntau, io->q, io->nq, io->l, io->nl, io->transfer_dev);
  if (rc) return rc;
  if (io->nq < 3) return cpt_fail(h, CPT_ERR_INVALID, "bad arguments to cpt_step");
  h->defer = true;
  cpt_timer_start(h, CPT_T_STEP);
  rc = cpt_perturb_impl(h, io->k, io->nk, io->tau_sampling, io->ntau, nullptr, io->stats, io->status);
  if (!rc) rc = cpt_transfer_impl(h, nullptr, io->k, io->nk, io->k_size_cl, io->tau_sampling, io->ntau, io->q, io->nq, io->l, io->nl, io->transfer_dev);
  if (!rc) rc = cpt_cl_impl(h, io->sp, io->transfer_dev, io->q, io->nq, io->nl, io->cl_dev);
  if (!rc && io->lp) rc = cpt_lensing_impl(h, io->sp, io->lp, io->l, io->nl, io->cl_dev, io->cl_lensed_dev);
  if (!rc && io->pk_dev) rc = cpt_pk_impl(h, io->sp, io->k, io->nk, io->pk_dev, 0);
  cpt_timer_stop(h, CPT_T_STEP);
  h->defer = false;
  if (rc) { (void)hipStreamSynchronize(h->stream); h->pend_nk = 0; h->pend_work = false; return rc; }
  if ((rc = cpt_finish(h))) return rc;
  // a k-mode whose integration failed invalidates everything computed from the sources: reported here, after the single sync
  return cpt_perturb_collect(h, io->k, io->stats, io->status);
}

int cpt_get_sources(cpt_handle* h, double* sources_dev) {
  CPT_ENTER(h);
  if (!h->d_src || !h->src_nk) return cpt_fail(h, CPT_ERR_INVALID, "no resident sources: run cpt_perturb_solve_batch first");
  int rc = cpt_transpose_from_kmajor(h, h->d_src, sources_dev, h->cfg.tp_size, h->src_ntau, h->src_nk);
  if (rc) return rc;
  return cpt_finish(h);
}

int cpt_last_kernel_ms(const cpt_handle* h, int stage, double* ms, int* launches) {
  if (!h || !ms || !launches || stage < 0 || stage >= CPT_T_N) return CPT_ERR_INVALID;
  const Timer& t = h->timers[stage];
  *ms = t.ms;
  *launches = t.launches;
  return CPT_OK;
}

int cpt_last_transfer_work(const cpt_handle* h, long long* integrals, long long* type_samples, long long* fused_samples) {
  if (!h || !integrals || !type_samples || !fused_samples) return CPT_ERR_INVALID;
  *integrals = h->work_integrals;
  *type_samples = h->work_samples;
  *fused_samples = h->work_fused;
  return CPT_OK;
}

int cpt_dbg_lookup(cpt_handle* h, const double* tau, int n, double* out) {
  CPT_ENTER(h);
  return cpt_dbg_lookup_impl(h, tau, n, out);
}

int cpt_dbg_derivs(cpt_handle* h, double k, double tau, int tca_on, int rsa_on, int ufa_on, const double* y, double* dy,
                   int* neq) {
  CPT_ENTER(h);
  return cpt_dbg_derivs_impl(h, k, tau, tca_on, rsa_on, ufa_on, y, dy, neq);
}

int cpt_dbg_solve(cpt_handle* h, double k, double tau, int tca_on, int rsa_on, int ufa_on, double hg, const double* b,
                  double* x) {
  CPT_ENTER(h);
  return cpt_dbg_solve_impl(h, k, tau, tca_on, rsa_on, ufa_on, hg, b, x);
}

int cpt_dbg_bessel(cpt_handle* h, const int* l, int nl, double xmax, int* nx, double* phi, double* dphi,
                   double* chi_at_phimin, int cap_nx) {
  CPT_ENTER(h);
  int rc = cpt_bessel_build(h, l, nl, xmax);
  if (rc) return rc;
  CPT_HIP(h, hipStreamSynchronize(h->stream));
  *nx = h->bes_nx;
  if (h->bes_nx > cap_nx) return cpt_fail(h, CPT_ERR_INVALID, "cap_nx=%d too small for nx=%d", cap_nx, h->bes_nx);
  std::vector<double2> tmp((size_t)nl * h->bes_nx);
  CPT_HIP(h, hipMemcpy(tmp.data(), h->d_bes, tmp.size() * sizeof(double2), hipMemcpyDeviceToHost));
  for (size_t i = 0; i < tmp.size(); i++) {
    phi[i] = tmp[i].x;
    dphi[i] = tmp[i].y;
  }
  CPT_HIP(h, hipMemcpy(chi_at_phimin, h->d_chi_min, nl * sizeof(double), hipMemcpyDeviceToHost));
  return CPT_OK;
}

}  // extern "C"
