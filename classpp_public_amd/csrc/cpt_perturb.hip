// Hot path A on MI355X: per-k stiff integration of the scalar Einstein-Boltzmann system.
//
// ONE WAVEFRONT OWNS ONE k-MODE (block = 64 threads = 1 wave).  Lane i owns equation i of the current regime:
// the state y, the backward differences dif[0..6], the Newton iterates, the Jacobian and the factors of
// (I - h*gamma*J) all live in lane registers (the matrix is never stored densely: see "structured linear algebra");
// the adaptive order/step control is scalar control flow that is uniform in the wave, so divergence between modes
// never crosses a wavefront.  The background / thermodynamics spline tables are read through a 64-row window staged
// in LDS whose abscissae sit in lane registers (wave-parallel bracket search by ballot+popcount) plus a row cache, so
// a step that stays inside the current table cell touches no memory at all.
//
// Restates (not translates): perturb_solve pm.cpp:2463-2787, perturb_approximations :5443-5670,
// perturb_vector_init :3271-4688, perturb_initial_conditions :4723-5408, perturb_einstein/total_stress_energy
// :5840-6703, perturb_derivs :7861-9218, perturb_tca_slip_and_shear :9229-9516, perturb_rsa_delta_and_theta
// :9530-9636, perturb_sources :6731-7285, background_at_tau / thermodynamics_at_z, and evolver_ndf15
// ev.cpp:62-705 (+ interp_from_dif :860-905, adjust_stepsize :907-943, new_linearisation :945-998).
// Differences by design: the Jacobian is obtained exactly as J e_j = f(tau, e_j) (the system is linear and
// homogeneous in y) instead of by adaptive finite differences (ev.cpp:1213-1539); the factorisation uses the
// structure of the equations (three tridiagonal hierarchy tails + a <=16x16 dense core) instead of a numerically
// discovered sparsity pattern with AMD ordering (tools/sparse.c:130-599); switch times are located by a 64-ary
// search instead of bisection.
#include "cpt_perturb_device.h"

namespace {


template <int GAUGE, int CURV, int MODE, int ROWS>
__global__ void __launch_bounds__(128) k_perturb(PtParams P) { PT<GAUGE, CURV, MODE, 0, ROWS>::body_perturb(P); }   // integrator wave + sampler wave
template <int GAUGE, int CURV, int MODE, int ROWS>
__global__ void __launch_bounds__(64) k_dbg_lookup(PtParams P, const double* tau, int n, double* out) { PT<GAUGE, CURV, MODE, 0, ROWS>::body_dbg_lookup(P, tau, n, out); }
template <int GAUGE, int CURV, int MODE, int ROWS>
__global__ void __launch_bounds__(64) k_dbg_derivs(PtParams P, double k, double tau, int tca, int rsa, int ufa, const double* y, double* dy, int* neq) {
  PT<GAUGE, CURV, MODE, 0, ROWS>::body_dbg_derivs(P, k, tau, tca, rsa, ufa, y, dy, neq);
}
template <int GAUGE, int CURV, int MODE, int ROWS>
__global__ void __launch_bounds__(64) k_dbg_solve(PtParams P, double k, double tau, int tca, int rsa, int ufa, double hg, const double* b, double* x) {
  PT<GAUGE, CURV, MODE, 0, ROWS>::body_dbg_solve(P, k, tau, tca, rsa, ufa, hg, b, x);
}

// the four instantiations (gauge x curvature) behind one launch expression
#define CPT_PT_DISPATCH(cfg, rows, KERNEL, ...)                                                                   \
  do {                                                                                                            \
    const bool newt__ = (cfg).gauge == CPT_GAUGE_NEWTONIAN, curv__ = (cfg).K != 0., rows__ = (rows) != 0;         \
    if ((cfg).mode == CPT_MODE_TENSORS) { /* the tensor equations are the same in both gauges */                  \
      if (curv__) hipLaunchKernelGGL((KERNEL<CPT_GAUGE_SYNCHRONOUS, 1, 1, 0>), __VA_ARGS__);                      \
      else hipLaunchKernelGGL((KERNEL<CPT_GAUGE_SYNCHRONOUS, 0, 1, 0>), __VA_ARGS__);                             \
    }                                                                                                             \
    else if (rows__) { /* every tail in a 16-lane row of its own: log-depth tail solves */                        \
      if (newt__ && curv__) hipLaunchKernelGGL((KERNEL<CPT_GAUGE_NEWTONIAN, 1, 0, 1>), __VA_ARGS__);              \
      else if (newt__) hipLaunchKernelGGL((KERNEL<CPT_GAUGE_NEWTONIAN, 0, 0, 1>), __VA_ARGS__);                   \
      else if (curv__) hipLaunchKernelGGL((KERNEL<CPT_GAUGE_SYNCHRONOUS, 1, 0, 1>), __VA_ARGS__);                 \
      else hipLaunchKernelGGL((KERNEL<CPT_GAUGE_SYNCHRONOUS, 0, 0, 1>), __VA_ARGS__);                             \
    }                                                                                                             \
    else if (newt__ && curv__) hipLaunchKernelGGL((KERNEL<CPT_GAUGE_NEWTONIAN, 1, 0, 0>), __VA_ARGS__);           \
    else if (newt__) hipLaunchKernelGGL((KERNEL<CPT_GAUGE_NEWTONIAN, 0, 0, 0>), __VA_ARGS__);                     \
    else if (curv__) hipLaunchKernelGGL((KERNEL<CPT_GAUGE_SYNCHRONOUS, 1, 0, 0>), __VA_ARGS__);                   \
    else hipLaunchKernelGGL((KERNEL<CPT_GAUGE_SYNCHRONOUS, 0, 0, 0>), __VA_ARGS__);                               \
  } while (0)

}  // namespace

int cpt_perturb_impl(cpt_handle* h, const double* k, int nk, const double* tau, int ntau, double* sources_dev, cpt_stepstat* stats,
                     int* status) {
  const cpt_config& c = h->cfg;
  for (int i = 0; i < nk; i++)
    if (!(k[i] > 0.)) return cpt_fail(h, CPT_ERR_INVALID, "stop to avoid division by zero: k[%d]=%g (pm.cpp:2524)", i, k[i]);
  for (int i = 1; i < ntau; i++)
    if (!(tau[i] > tau[i - 1])) return cpt_fail(h, CPT_ERR_INVALID, "tau_sampling must be strictly increasing");
  if (!(tau[ntau - 1] <= c.tau0 * (1. + 1e-12))) return cpt_fail(h, CPT_ERR_INVALID, "tau_sampling exceeds the conformal age");
  PtParams P;
  fill_params(h, P);
  const int ntp = c.tp_size;
  const size_t nsrc = (size_t)ntp * nk * ntau;
  int rc;
  if ((rc = cpt_reserve(h, &h->d_src, &h->src_cap, nsrc))) return rc;
  h->src_nk = h->src_ntau = 0;   // no resident sources until this launch has been checked (cpt_perturb_collect)
  // scratch: k[nk] tau[ntau] | stats[nk] | order[nk] status[nk]
  const size_t bytes = (size_t)(nk + ntau) * sizeof(double) + (size_t)2 * nk * sizeof(int) + (size_t)nk * sizeof(cpt_stepstat) + 64;
  if (h->pt_scratch_cap < bytes) {
    if (h->d_pt_scratch) (void)hipFree(h->d_pt_scratch);
    h->d_pt_scratch = nullptr; h->pt_scratch_cap = 0; h->geo_pt_valid = false;
    CPT_HIP(h, hipMalloc(&h->d_pt_scratch, bytes));
    h->pt_scratch_cap = bytes;
  }
  double* d_k = (double*)h->d_pt_scratch;
  double* d_tau = d_k + nk;
  cpt_stepstat* d_stats = (cpt_stepstat*)(d_tau + ntau);
  int* d_order = (int*)(d_stats + nk);
  int* d_status = d_order + nk;
  // the grids travel once per geometry: a call with the k and tau of the previous one finds them (and the launch order) in HBM
  const bool same_grids = h->geo_pt_valid && (int)h->geo_pt_k.size() == nk && (int)h->geo_pt_tau.size() == ntau &&
                          memcmp(h->geo_pt_k.data(), k, nk * sizeof(double)) == 0 && memcmp(h->geo_pt_tau.data(), tau, ntau * sizeof(double)) == 0;
  if (!same_grids) {
    h->geo_pt_valid = false;
    int* order = (int*)cpt_pin(h, nullptr, nk * sizeof(int));
    if (!order) return cpt_fail(h, CPT_ERR_NO_DEVICE, "hipHostMalloc of the staging arena failed");
    for (int i = 0; i < nk; i++) order[i] = nk - 1 - i;  // largest k first: the longest chains start first (pm.cpp:685)
    if ((rc = cpt_upload(h, d_k, k, nk * sizeof(double)))) return rc;
    if ((rc = cpt_upload(h, d_tau, tau, ntau * sizeof(double)))) return rc;
    CPT_HIP(h, hipMemcpyAsync(d_order, order, nk * sizeof(int), hipMemcpyHostToDevice, h->stream));
    h->geo_pt_k.assign(k, k + nk); h->geo_pt_tau.assign(tau, tau + ntau);
    h->geo_pt_valid = true;
  }
  CPT_HIP(h, hipMemsetAsync(h->d_src, 0, nsrc * sizeof(double), h->stream));  // pm.cpp:2767-2771 zero tail
  P.k = d_k; P.tau_s = d_tau; P.order = d_order; P.nk = nk; P.ntau = ntau; P.src = h->d_src; P.stats = d_stats; P.status = d_status;
  cpt_timer_start(h, CPT_T_PERTURB);
  if ((c.has_ncdm || P.long_tails) && c.mode == CPT_MODE_SCALARS) {
    // more than 64 equations per k-mode: the register-set kernels (one wavefront per mode, cpt_perturb_sets.inc; one translation unit
    // per family)
    const int cpw = c.has_ncdm ? 64 / (c.l_max_ncdm + 1) : 1, nsets = c.has_ncdm ? (h->ncdm.nchains + cpw - 1) / cpw : 0;
    if (P.long_tails && nsets == 0) rc = cpt_perturb_sets_launch_0(h, d_k, d_tau, d_order, nk, ntau, h->d_src, d_stats, d_status);
    else if (P.long_tails && nsets <= 3) rc = cpt_perturb_sets_launch_13(h, d_k, d_tau, d_order, nk, ntau, h->d_src, d_stats, d_status);
    else if (P.long_tails) return cpt_fail(h, CPT_ERR_UNSUPPORTED, "hierarchies longer than one wavefront together with %d ncdm momentum bins need %d bin sets (at most 3)", h->ncdm.nchains, nsets);
    else if (nsets <= 2) rc = cpt_perturb_sets_launch_2(h, d_k, d_tau, d_order, nk, ntau, h->d_src, d_stats, d_status);
    else if (nsets <= 5) rc = cpt_perturb_sets_launch_5(h, d_k, d_tau, d_order, nk, ntau, h->d_src, d_stats, d_status);
    else return cpt_fail(h, CPT_ERR_UNSUPPORTED, "%d ncdm momentum bins need %d register sets per k-mode (at most 5)", h->ncdm.nchains, nsets);
    if (rc) return rc;
  } else
  CPT_PT_DISPATCH(c, P.rows, k_perturb, dim3(nk), dim3(128), 0, h->stream, P);
  CPT_HIP(h, hipGetLastError());
  cpt_timer_stop(h, CPT_T_PERTURB);
  if (sources_dev) {
    if ((rc = cpt_transpose_from_kmajor(h, h->d_src, sources_dev, ntp, ntau, nk))) return rc;
  }
  // per-mode status and work counters land in pinned host memory; they are read after the (single) synchronisation of the call
  const size_t need_out = 64 + (size_t)nk * (sizeof(int) + sizeof(cpt_stepstat)) + 64;
  if (h->pin_out_cap < need_out) {
    CPT_HIP(h, hipStreamSynchronize(h->stream));
    if (h->pin_out) (void)hipHostFree(h->pin_out);
    h->pin_out = nullptr; h->pin_out_cap = 0;
    CPT_HIP(h, hipHostMalloc((void**)&h->pin_out, need_out * 2, hipHostMallocDefault));
    h->pin_out_cap = need_out * 2;
  }
  cpt_stepstat* hstats = (cpt_stepstat*)(h->pin_out + 64);
  int* hstatus = (int*)(hstats + nk);
  CPT_HIP(h, hipMemcpyAsync(hstatus, d_status, nk * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  CPT_HIP(h, hipMemcpyAsync(hstats, d_stats, nk * sizeof(cpt_stepstat), hipMemcpyDeviceToHost, h->stream));
  h->pend_nk = nk;
  // (later stages of a fused cpt_step read the resident sources of THIS launch: shape known now, validity checked at the end)
  h->src_nk = nk; h->src_ntau = ntau;
  (void)stats; (void)status;
  return CPT_OK;
}

// after the synchronisation: hand the per-mode results to the caller and turn a failed mode into the call's error
int cpt_perturb_collect(cpt_handle* h, const double* k, cpt_stepstat* stats, int* status) {
  const int nk = h->pend_nk;
  if (!nk) return CPT_OK;
  h->pend_nk = 0;
  const cpt_stepstat* hstats = (const cpt_stepstat*)(h->pin_out + 64);
  const int* hstatus = (const int*)(hstats + nk);
  if (stats) memcpy(stats, hstats, nk * sizeof(cpt_stepstat));
  if (status) memcpy(status, hstatus, nk * sizeof(int));
  for (int i = 0; i < nk; i++) {
    if (hstatus[i]) {
      const char* what = hstatus[i] == 11   ? "Step size too small (ev.cpp:461,492)"
                         : hstatus[i] == 12 ? "singular matrix in LU (ev.cpp:975)"
                         : hstatus[i] == 14 ? "step budget exhausted"
                         : hstatus[i] == 15 ? "helper wavefront unresponsive (internal error)"
                         : hstatus[i] == 20 ? "initial time of the background table is too late for this k (pm.cpp:2562-2573)"
                         : hstatus[i] >= 21 ? "approximation switching times cannot be ordered (pm.cpp:3137-3173)"
                                            : "integration failure";
      h->src_nk = h->src_ntau = 0;   // partially integrated sources must not feed cpt_transfer_batch(NULL) / cpt_pk_linear
      return cpt_fail(h, CPT_ERR_RUNTIME, "perturb_solve failed for k=%e (mode %d): %s [status %d]", k[i], i, what, hstatus[i]);
    }
  }
  return CPT_OK;
}

// cycles spent by the heaviest mode of the last perturb launch: rhs(Newton), lu_solve, factorise, jacobian, sampling,
// adjust_stepsize, schedule, total.  Zeros unless built with -DCPT_PROFILE (diagnostic builds only, tools/prof_run.py).
extern "C" int cpt_dbg_profile(unsigned long long* out) {
#ifdef CPT_COUNT_RESTAGE
  unsigned long long r[4];
  if (hipMemcpyFromSymbol(r, HIP_SYMBOL(g_restage), sizeof(r)) == hipSuccess)
    fprintf(stderr, "[restage] thermo slides %llu  background slides %llu  bsearch fallbacks %llu  lookups %llu (block 0, cumulative)\n", r[0], r[1], r[2], r[3]);
#endif
#ifdef CPT_PROFILE
  // (out[0..15]: the integrator of the heaviest mode; a caller that passes room for 32 also gets its helper wave: CPT_PROFILE_HELPER=1)
  if (getenv("CPT_PROFILE_HELPER") && hipMemcpyFromSymbol(out + 16, HIP_SYMBOL(g_prof_helper), 16 * sizeof(unsigned long long)) != hipSuccess) return 3;
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof), 16 * sizeof(unsigned long long)) == hipSuccess ? 0 : 3;
#else
  for (int i = 0; i < 8; i++) out[i] = 0;
  return 0;
#endif
}

int cpt_dbg_lookup_impl(cpt_handle* h, const double* tau, int n, double* out) {
  PtParams P;
  fill_params(h, P);
  double *d_tau = nullptr, *d_out = nullptr;
  for (int i = 0; i < n; i++)
    if (!(tau[i] >= 0.)) return cpt_fail(h, CPT_ERR_INVALID, "negative tau");
  CPT_HIP(h, hipMalloc((void**)&d_tau, n * sizeof(double)));
  CPT_HIP(h, hipMalloc((void**)&d_out, (size_t)n * 16 * sizeof(double)));
  CPT_HIP(h, hipMemcpy(d_tau, tau, n * sizeof(double), hipMemcpyHostToDevice));
  CPT_PT_DISPATCH(h->cfg, P.rows, k_dbg_lookup, dim3(1), dim3(64), 0, h->stream, P, d_tau, n, d_out);
  CPT_HIP(h, hipGetLastError());
  CPT_HIP(h, hipStreamSynchronize(h->stream));
  CPT_HIP(h, hipMemcpy(out, d_out, (size_t)n * 16 * sizeof(double), hipMemcpyDeviceToHost));
  (void)hipFree(d_tau);
  (void)hipFree(d_out);
  return CPT_OK;
}

int cpt_dbg_derivs_impl(cpt_handle* h, double k, double tau, int tca_on, int rsa_on, int ufa_on, const double* y, double* dy,
                        int* neq) {
  PtParams P;
  fill_params(h, P);
  if (!(k > 0.) || !(tau > 0.)) return cpt_fail(h, CPT_ERR_INVALID, "k and tau must be positive");
  double *d_y = nullptr, *d_dy = nullptr;
  int* d_neq = nullptr;
  CPT_HIP(h, hipMalloc((void**)&d_y, 64 * sizeof(double)));
  CPT_HIP(h, hipMalloc((void**)&d_dy, 64 * sizeof(double)));
  CPT_HIP(h, hipMalloc((void**)&d_neq, sizeof(int)));
  CPT_HIP(h, hipMemset(d_dy, 0, 64 * sizeof(double)));
  CPT_HIP(h, hipMemcpy(d_y, y, 64 * sizeof(double), hipMemcpyHostToDevice));
  CPT_PT_DISPATCH(h->cfg, P.rows, k_dbg_derivs, dim3(1), dim3(64), 0, h->stream, P, k, tau, tca_on ? 1 : 0, rsa_on ? 1 : 0, ufa_on ? 1 : 0, d_y,
                  d_dy, d_neq);
  CPT_HIP(h, hipGetLastError());
  CPT_HIP(h, hipStreamSynchronize(h->stream));
  CPT_HIP(h, hipMemcpy(dy, d_dy, 64 * sizeof(double), hipMemcpyDeviceToHost));
  CPT_HIP(h, hipMemcpy(neq, d_neq, sizeof(int), hipMemcpyDeviceToHost));
  (void)hipFree(d_y);
  (void)hipFree(d_dy);
  (void)hipFree(d_neq);
  return CPT_OK;
}

int cpt_dbg_solve_impl(cpt_handle* h, double k, double tau, int tca_on, int rsa_on, int ufa_on, double hg, const double* b,
                       double* x) {
  PtParams P;
  fill_params(h, P);
  if (!(k > 0.) || !(tau > 0.)) return cpt_fail(h, CPT_ERR_INVALID, "k and tau must be positive");
  if (const char* e = getenv("CPT_DBG_SOLVE_INVERSE")) P.dbg_inverse = atoi(e);   // (tests: the product form of the core solve)
  double *d_b = nullptr, *d_x = nullptr;
  CPT_HIP(h, hipMalloc((void**)&d_b, 64 * sizeof(double)));
  CPT_HIP(h, hipMalloc((void**)&d_x, 64 * sizeof(double)));
  CPT_HIP(h, hipMemset(d_x, 0, 64 * sizeof(double)));
  CPT_HIP(h, hipMemcpy(d_b, b, 64 * sizeof(double), hipMemcpyHostToDevice));
  CPT_PT_DISPATCH(h->cfg, P.rows, k_dbg_solve, dim3(1), dim3(64), 0, h->stream, P, k, tau, tca_on ? 1 : 0, rsa_on ? 1 : 0, ufa_on ? 1 : 0, hg,
                  d_b, d_x);
  CPT_HIP(h, hipGetLastError());
  CPT_HIP(h, hipStreamSynchronize(h->stream));
  CPT_HIP(h, hipMemcpy(x, d_x, 64 * sizeof(double), hipMemcpyDeviceToHost));
  (void)hipFree(d_b);
  (void)hipFree(d_x);
  return CPT_OK;
}
