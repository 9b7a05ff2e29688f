// "Next" rows S8f-2,3: C_l assembly and linear P(k) on the device, so that the path's results can be stated in the
// contract's units without leaving HBM.  Restates SpectraModule::spectra_compute_cl (source/spectra_module.cpp:958-1353,
// flat space, scalar mode, one initial condition) with the integrand spline of tools/arrays.c (array_spline,
// _SPLINE_EST_DERIV_) and array_integrate_all_trapzd_or_spline (:1382-1423), and NonlinearModule::nonlinear_pk_linear
// (source/nonlinear_module.cpp:1886-2040).
#include "cpt_internal.h"
#include <cmath>
#include <vector>

struct ClParams {
  const double* tr;  // [tt][nl][nq]
  const double* tr2; // second initial condition of a cross-correlation spectrum (same layout), or null: tr with itself
  const double* w;   // [nq]: quadrature weight of the integrand spline x primordial spectrum x 4 pi / q
  double* cl;        // [nl][ct]
  int nq, nl, ct_size;
  int tt_t0, tt_t1, tt_t2, tt_e, tt_lcmb, tt_b;
  int tensors;  // tensor mode: temperature = t2 alone, BB = b b, no lensing-potential spectra (spectra_module.cpp:1027-1185)
  int ct_tt, ct_ee, ct_te, ct_bb, ct_pp, ct_tp, ct_ep;
};

// C_l = int dq/q 4 pi P(q) Delta_l^X(q) Delta_l^Y(q), the integral being that of the cubic spline through the integrand
// (spectra_module.cpp:1027-1323).  For a fixed q grid that integral is a LINEAR functional of the integrand samples,
// sum_q W_q y_q: the host folds the whole spline construction (arrays.c array_spline, _SPLINE_EST_DERIV_) and its
// integration (array_integrate_all_trapzd_or_spline, arrays.c:1382-1423) into the weights W_q once per grid (adjoint
// of the two sweeps, spline_integration_weights below), and the kernel is a coalesced dot product: one workgroup per
// l, the 6 spectra share the transfer rows.  The sequential per-(l,ct) sweeps this replaces took 3.5 ms.
__global__ void __launch_bounds__(256) k_cl(ClParams P) {
  const int il = blockIdx.x, tid = threadIdx.x, nq = P.nq;
  const size_t st = (size_t)P.nl * nq, row = (size_t)il * nq;
  double acc[7] = {0., 0., 0., 0., 0., 0., 0.};  // tt, ee, te, pp, tp, ep, bb
  for (int iq = tid; iq < nq; iq += 256) {
    double temp = 0., e = 0., lc = 0., bm = 0.;
    if (P.tensors) {
      if (P.tt_t2 >= 0) temp = P.tr[P.tt_t2 * st + row + iq];
      if (P.tt_b >= 0) bm = P.tr[P.tt_b * st + row + iq];
    } else if (P.tt_t0 >= 0) temp = P.tr[P.tt_t0 * st + row + iq] + P.tr[P.tt_t1 * st + row + iq] + P.tr[P.tt_t2 * st + row + iq];
    if (P.tt_e >= 0) e = P.tr[P.tt_e * st + row + iq];
    if (!P.tensors && P.tt_lcmb >= 0) lc = P.tr[P.tt_lcmb * st + row + iq];
    const double w = P.w[iq];
    if (P.tr2) {
      // two initial conditions (scalars): Delta^X_ic1 Delta^Y_ic2, symmetrised for X != Y (spectra_module.cpp:1137-1185)
      double temp2 = 0., e2 = 0., lc2 = 0.;
      if (P.tt_t0 >= 0) temp2 = P.tr2[P.tt_t0 * st + row + iq] + P.tr2[P.tt_t1 * st + row + iq] + P.tr2[P.tt_t2 * st + row + iq];
      if (P.tt_e >= 0) e2 = P.tr2[P.tt_e * st + row + iq];
      if (P.tt_lcmb >= 0) lc2 = P.tr2[P.tt_lcmb * st + row + iq];
      acc[0] = fma(w, temp * temp2, acc[0]);
      acc[1] = fma(w, e * e2, acc[1]);
      acc[2] = fma(w, 0.5 * (temp * e2 + e * temp2), acc[2]);
      acc[3] = fma(w, lc * lc2, acc[3]);
      acc[4] = fma(w, 0.5 * (temp * lc2 + lc * temp2), acc[4]);
      acc[5] = fma(w, 0.5 * (e * lc2 + lc * e2), acc[5]);
      continue;
    }
    acc[0] = fma(w, temp * temp, acc[0]);
    acc[1] = fma(w, e * e, acc[1]);
    acc[2] = fma(w, temp * e, acc[2]);
    acc[3] = fma(w, lc * lc, acc[3]);
    acc[4] = fma(w, temp * lc, acc[4]);
    acc[5] = fma(w, e * lc, acc[5]);
    acc[6] = fma(w, bm * bm, acc[6]);
  }
  __shared__ double red[4][7];
#pragma unroll
  for (int c = 0; c < 7; c++) {
    double v = acc[c];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((tid & 63) == 0) red[tid >> 6][c] = v;
  }
  __syncthreads();
  if (tid < P.ct_size) {
    const int ct = tid;
    int kind = -1;
    if (ct == P.ct_tt) kind = 0; else if (ct == P.ct_ee) kind = 1; else if (ct == P.ct_te) kind = 2;
    else if (ct == P.ct_pp) kind = 3; else if (ct == P.ct_tp) kind = 4; else if (ct == P.ct_ep) kind = 5;
    if (P.tensors) { if (kind >= 3) kind = -1; if (ct == P.ct_bb) kind = 6; }
    double v = 0.;  // bb vanishes for scalar modes (spectra_module.cpp:1262-1270)
    if (kind >= 0) v = red[0][kind] + red[1][kind] + red[2][kind] + red[3][kind];
    P.cl[(size_t)il * P.ct_size + ct] = v;
  }
}

// W with  sum_i W_i y_i == integral of the _SPLINE_EST_DERIV_ cubic spline through (x_i, y_i)  (arrays.c:1382-1423).
// Forward algorithm (what the reference runs per spectrum):
//   u_0 = 3/h_0 ((y_1-y_0)/h_0 - y'_first);  u_i = (6 d_i/(x_{i+1}-x_{i-1}) - sig_i u_{i-1})/p_i,  d_i = second divided difference
//   dd_{n-1} = (u_n - u_{n-2}/2)/(c_{n-2}/2 + 1);  dd_i = c_i dd_{i+1} + u_i;  I = sum_i (y_i+y_{i+1}) h_i/2 + (dd_i+dd_{i+1}) h_i^3/24
// run here in reverse (adjoint) mode: O(n), exact up to the order of the floating-point additions.
// Cells below istart are integrated with the trapezoidal rule only (array_integrate_all_trapzd_or_spline, arrays.c:1382-1423;
// closed space, spectra_module.cpp:1293-1316), the spline itself is still built on all points.
static void spline_integration_weights(const double* x, int n, int istart, double* W) {
  std::vector<double> c(n), sig(n), p(n), mu(n), nu(n);
  c[0] = -0.5; sig[0] = 0.; p[0] = 1.;
  for (int i = 1; i < n - 1; i++) {
    sig[i] = (x[i] - x[i - 1]) / (x[i + 1] - x[i - 1]);
    p[i] = sig[i] * c[i - 1] + 2.0;
    c[i] = (sig[i] - 1.0) / p[i];
  }
  auto h = [&](int i) { return x[i + 1] - x[i]; };
  for (int i = 0; i < n; i++) W[i] = 0.;
  for (int i = 0; i < n - 1; i++) { W[i] += 0.5 * h(i); W[i + 1] += 0.5 * h(i); }
  // d I / d dd_i, including the chain dd_i -> dd_{i-1} -> ...
  for (int i = 0; i < n; i++) {
    double g = 0.;
    if (i >= 1 && i - 1 >= istart) g += h(i - 1) * h(i - 1) * h(i - 1) / 24.;
    if (i <= n - 2 && i >= istart) g += h(i) * h(i) * h(i) / 24.;
    mu[i] = g + (i >= 1 ? c[i - 1] * mu[i - 1] : 0.);
  }
  const double D = 0.5 * c[n - 2] + 1.0;
  const double a_un = mu[n - 1] / D;
  // d I / d u_i, including the chain u_i -> u_{i+1} -> ...
  nu[n - 2] = mu[n - 2] - 0.5 * mu[n - 1] / D;
  for (int i = n - 3; i >= 0; i--) nu[i] = mu[i] - sig[i + 1] / p[i + 1] * nu[i + 1];
  for (int i = 1; i <= n - 2; i++) {
    const double kap = 6. * nu[i] / ((x[i + 1] - x[i - 1]) * p[i]);
    W[i + 1] += kap / h(i);
    W[i] -= kap * (1. / h(i) + 1. / h(i - 1));
    W[i - 1] += kap / h(i - 1);
  }
  {  // u_0 with the estimated first derivative at x_0
    const double den = (x[2] - x[0]) * (x[1] - x[0]) * (x[2] - x[1]);
    const double A = (x[2] - x[0]) * (x[2] - x[0]) / den, B = (x[1] - x[0]) * (x[1] - x[0]) / den;
    const double c0 = nu[0] * 3. / h(0);
    W[1] += c0 * (1. / h(0) - A);
    W[0] += c0 * (-1. / h(0) + A - B);
    W[2] += c0 * B;
  }
  {  // u_n with the estimated first derivative at x_{n-1}
    const double den = (x[n - 3] - x[n - 1]) * (x[n - 2] - x[n - 1]) * (x[n - 3] - x[n - 2]);
    const double A = (x[n - 3] - x[n - 1]) * (x[n - 3] - x[n - 1]) / den, B = (x[n - 2] - x[n - 1]) * (x[n - 2] - x[n - 1]) / den;
    const double cn = a_un * 3. / h(n - 2);
    W[n - 2] += cn * (A + 1. / h(n - 2));
    W[n - 1] += cn * (-A + B - 1. / h(n - 2));
    W[n - 3] -= cn * B;
  }
}

__global__ void k_pk(const double* __restrict__ src, const double* __restrict__ k, double* __restrict__ pk, int nk, int ntau,
                     int tp_dm, double A_s, double n_s, double alpha_s, double k_pivot) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nk) return;
  const double dm = src[((size_t)tp_dm * nk + i) * ntau + (ntau - 1)];  // resident sources are k-major: [tp][k][tau]
  const double lk = log(k[i] / k_pivot);
  const double pr = A_s * exp((n_s - 1.) * lk + 0.5 * alpha_s * lk * lk);
  const double PI = 3.1415926535897932384626433832795e0;
  pk[i] = 2. * PI * PI / (k[i] * k[i] * k[i]) * dm * dm * pr;  // nonlinear_module.cpp:1952-1991
}

// linear P(k, z) at 0 < z <= z_max_pk: one thread per k.  ln P(k, tau_j) at the last n sampling times (the tail ln_tau_ of the sampling,
// pm.cpp:1554-1592; P from the delta_m source as in k_pk, nonlinear_module.cpp:1952-1991) is splined in ln tau with estimated end
// derivatives (array_spline_table_lines / _SPLINE_EST_DERIV_, tools/arrays.c:261-353) and evaluated at ln tau(z)
// (NonlinearModule::nonlinear_pk_at_z, nonlinear_module.cpp:81-283).  u, dd: scratch [n][nk] (thread i walks column i: coalesced).
__global__ void k_pk_z(const double* __restrict__ src, const double* __restrict__ k, const double* __restrict__ tau, double* __restrict__ pk, double* __restrict__ u,
                       double* __restrict__ dd, int nk, int ntau, int n, int tp_dm, double ln_tau_z, double A_s, double n_s, double alpha_s, double k_pivot) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nk) return;
  const double PI = 3.1415926535897932384626433832795e0;
  const double lk = log(k[i] / k_pivot);
  const double ln_amp = log(2. * PI * PI / (k[i] * k[i] * k[i]) * A_s) + (n_s - 1.) * lk + 0.5 * alpha_s * lk * lk;
  const double* col = src + ((size_t)tp_dm * nk + i) * ntau + (ntau - n);   // resident sources are k-major: [tp][k][tau]
  const double* t = tau + (ntau - n);
  auto X = [&](int j) { return log(t[j]); };
  auto Y = [&](int j) { const double dm = col[j]; return ln_amp + log(dm * dm); };
  if (n < 3) { pk[i] = exp(Y(n - 1)); return; }
  {
    const double x0 = X(0), x1 = X(1), x2 = X(2), y0 = Y(0), y1 = Y(1), y2 = Y(2);
    const double dy_first = ((x2 - x0) * (x2 - x0) * (y1 - y0) - (x1 - x0) * (x1 - x0) * (y2 - y0)) / ((x2 - x0) * (x1 - x0) * (x2 - x1));
    dd[i] = -0.5;
    u[i] = (3. / (x1 - x0)) * ((y1 - y0) / (x1 - x0) - dy_first);
  }
  for (int j = 1; j < n - 1; j++) {
    const double xm = X(j - 1), x = X(j), xp = X(j + 1), ym = Y(j - 1), y = Y(j), yp = Y(j + 1);
    const double sig = (x - xm) / (xp - xm);
    const double p = sig * dd[(size_t)(j - 1) * nk + i] + 2.0;
    dd[(size_t)j * nk + i] = (sig - 1.0) / p;
    double uj = (yp - y) / (xp - x) - (y - ym) / (x - xm);
    uj = (6.0 * uj / (xp - xm) - sig * u[(size_t)(j - 1) * nk + i]) / p;
    u[(size_t)j * nk + i] = uj;
  }
  {
    const double xa = X(n - 3), xb = X(n - 2), xc = X(n - 1), ya = Y(n - 3), yb = Y(n - 2), yc = Y(n - 1);
    const double dy_last = ((xa - xc) * (xa - xc) * (yb - yc) - (xb - xc) * (xb - xc) * (ya - yc)) / ((xa - xc) * (xb - xc) * (xa - xb));
    const double qn = 0.5, un = (3. / (xc - xb)) * (dy_last - (yc - yb) / (xc - xb));
    dd[(size_t)(n - 1) * nk + i] = (un - qn * u[(size_t)(n - 2) * nk + i]) / (qn * dd[(size_t)(n - 2) * nk + i] + 1.0);
  }
  for (int j = n - 2; j >= 0; j--) dd[(size_t)j * nk + i] = dd[(size_t)j * nk + i] * dd[(size_t)(j + 1) * nk + i] + u[(size_t)j * nk + i];
  // array_interpolate_spline at ln tau(z) (wave-uniform interval)
  int inf = 0, sup = n - 1;
  while (sup - inf > 1) { const int mid = (int)(0.5 * (inf + sup)); if (ln_tau_z < X(mid)) sup = mid; else inf = mid; }
  const double h = X(sup) - X(inf), b = (ln_tau_z - X(inf)) / h, a = 1. - b;
  pk[i] = exp(a * Y(inf) + b * Y(sup) + ((a * a * a - a) * dd[(size_t)inf * nk + i] + (b * b * b - b) * dd[(size_t)sup * nk + i]) * h * h / 6.);
}

int cpt_cl_impl(cpt_handle* h, const cpt_spectra_params* sp, const double* transfer_dev, const double* q, int nq, int nl,
                double* cl_dev, const double* transfer2_dev) {
  const cpt_config& c = h->cfg;
  if (transfer2_dev && c.mode == CPT_MODE_TENSORS) return cpt_fail(h, CPT_ERR_INVALID, "tensor modes have one initial condition: no cross-correlation spectra");
  if (sp->ct_size < 1 || sp->ct_size > 8) return cpt_fail(h, CPT_ERR_INVALID, "ct_size=%d out of range", sp->ct_size);
  const int cts[7] = {sp->index_ct_tt, sp->index_ct_ee, sp->index_ct_te, sp->index_ct_bb, sp->index_ct_pp, sp->index_ct_tp, sp->index_ct_ep};
  for (int i = 0; i < 7; i++)
    if (cts[i] >= sp->ct_size) return cpt_fail(h, CPT_ERR_INVALID, "index_ct_* >= ct_size");
  const bool tens = c.mode == CPT_MODE_TENSORS;
  if (tens && (sp->index_ct_pp >= 0 || sp->index_ct_tp >= 0 || sp->index_ct_ep >= 0))
    return cpt_fail(h, CPT_ERR_INVALID, "tensor modes have no lensing-potential spectra");
  if (tens && sp->index_ct_bb >= 0 && c.index_tt_b < 0) return cpt_fail(h, CPT_ERR_INVALID, "BB requested without B transfer functions");
  if ((sp->index_ct_tt >= 0 || sp->index_ct_te >= 0 || sp->index_ct_tp >= 0) && (tens ? c.index_tt_t2 < 0 : (c.index_tt_t0 < 0 || c.index_tt_t1 < 0 || c.index_tt_t2 < 0)))
    return cpt_fail(h, CPT_ERR_INVALID, "temperature C_l requested without temperature transfer functions");
  if ((sp->index_ct_ee >= 0 || sp->index_ct_te >= 0 || sp->index_ct_ep >= 0) && c.index_tt_e < 0)
    return cpt_fail(h, CPT_ERR_INVALID, "polarisation C_l requested without E transfer functions");
  if ((sp->index_ct_pp >= 0 || sp->index_ct_tp >= 0 || sp->index_ct_ep >= 0) && c.index_tt_lcmb < 0)
    return cpt_fail(h, CPT_ERR_INVALID, "lensing C_l requested without lensing transfer functions");
  for (int i = 1; i < nq; i++)
    if (!(q[i] > q[i - 1])) return cpt_fail(h, CPT_ERR_INVALID, "q grid must be strictly increasing");
  if (nq < 4) return cpt_fail(h, CPT_ERR_INVALID, "need at least 4 q values for the integrand spline");
  int rc;
  const double* w0 = h->d_clw;
  if ((rc = cpt_reserve(h, &h->d_clw, &h->clw_cap, (size_t)nq))) return rc;
  if (h->d_clw != w0) h->geo_cl_valid = false;
  // the quadrature weights depend on the q grid and the primordial spectrum only: computed and uploaded once per (q, A_s, n_s, ...)
  const double spk[4] = {sp->A_s, sp->n_s, sp->alpha_s, sp->k_pivot};
  const bool hit = h->geo_cl_valid && (int)h->geo_cl_q.size() == nq && memcmp(h->geo_cl_q.data(), q, nq * sizeof(double)) == 0 &&
                   memcmp(h->geo_cl_sp, spk, sizeof(spk)) == 0;
  if (!hit) {
    h->geo_cl_valid = false;
    std::vector<double> W(nq), kk(nq);
    // integration variable k(q) = sqrt(q^2 - K(1+m)) (spectra_module.cpp:990-994); flat: k = q
    for (int i = 0; i < nq; i++) kk[i] = (c.K == 0.) ? q[i] : sqrt(q[i] * q[i] - c.K * (tens ? 3. : 1.));
    int index_q_spline = 0;
    if (c.K > 0.) {  // closed: trapezoidal rule where nu is integer (below the flat-approximation index), spectra_module.cpp:1293-1316
      const double q_approximation = c.hyper_flat_approximation_nu * sqrt(c.K);
      for (index_q_spline = 0; index_q_spline < nq - 1; index_q_spline++)
        if (q[index_q_spline] > q_approximation) break;
    }
    spline_integration_weights(kk.data(), nq, index_q_spline, W.data());
    if (c.K > 0.) W[0] += q[0] / kk[0] * sqrt(c.K) / 2.;   // discrete sum over nu: weight of the first point, spectra_module.cpp:1319-1321
    const double PI = 3.1415926535897932384626433832795e0;
    for (int i = 0; i < nq; i++) {  // primordial_module.cpp:911-925 (analytic spectrum) and the 4 pi / k of the measure
      const double lk = log(kk[i] / sp->k_pivot);
      W[i] *= sp->A_s * exp((sp->n_s - 1.) * lk + 0.5 * sp->alpha_s * lk * lk) * (4. * PI / kk[i]);
    }
    if ((rc = cpt_upload(h, h->d_clw, W.data(), (size_t)nq * sizeof(double)))) return rc;
    h->geo_cl_q.assign(q, q + nq); memcpy(h->geo_cl_sp, spk, sizeof(spk));
    h->geo_cl_valid = true;
  }
  ClParams P;
  P.tr = transfer_dev; P.tr2 = transfer2_dev; P.w = h->d_clw; P.cl = cl_dev; P.nq = nq; P.nl = nl; P.ct_size = sp->ct_size;
  P.tt_t0 = c.index_tt_t0; P.tt_t1 = c.index_tt_t1; P.tt_t2 = c.index_tt_t2; P.tt_e = c.index_tt_e; P.tt_lcmb = c.index_tt_lcmb;
  P.tt_b = c.index_tt_b; P.tensors = (c.mode == CPT_MODE_TENSORS) ? 1 : 0;
  P.ct_tt = sp->index_ct_tt; P.ct_ee = sp->index_ct_ee; P.ct_te = sp->index_ct_te; P.ct_bb = sp->index_ct_bb;
  P.ct_pp = sp->index_ct_pp; P.ct_tp = sp->index_ct_tp; P.ct_ep = sp->index_ct_ep;
  hipLaunchKernelGGL(k_cl, dim3(nl), dim3(256), 0, h->stream, P);
  CPT_HIP(h, hipGetLastError());
  return CPT_OK;
}

// sigma(R) = sqrt( 1/(2 pi^2) int dk k^2 P(k) W^2(kR) ): NonlinearModule::nonlinear_sigmas_at_z + nonlinear_sigmas
// (source/nonlinear_module.cpp:926-963, 2041-2180): ln P splined in ln k (estimated end derivatives), integrand sampled
// at k_per_decade points per decade, integrated over t = 1/(1+k) with the spline rule.
static void spline_est_deriv(const double* x, const double* y, int n, double* dd) {   // tools/arrays.c:967-1092 / 261-353, EST_DERIV, one column
  std::vector<double> u(n - 1);
  const double dy_first = ((x[2] - x[0]) * (x[2] - x[0]) * (y[1] - y[0]) - (x[1] - x[0]) * (x[1] - x[0]) * (y[2] - y[0])) / ((x[2] - x[0]) * (x[1] - x[0]) * (x[2] - x[1]));
  dd[0] = -0.5;
  u[0] = (3. / (x[1] - x[0])) * ((y[1] - y[0]) / (x[1] - x[0]) - dy_first);
  for (int i = 1; i < n - 1; i++) {
    const double sig = (x[i] - x[i - 1]) / (x[i + 1] - x[i - 1]);
    const double p = sig * dd[i - 1] + 2.0;
    dd[i] = (sig - 1.0) / p;
    u[i] = (y[i + 1] - y[i]) / (x[i + 1] - x[i]) - (y[i] - y[i - 1]) / (x[i] - x[i - 1]);
    u[i] = (6.0 * u[i] / (x[i + 1] - x[i - 1]) - sig * u[i - 1]) / p;
  }
  const double dy_last = ((x[n - 3] - x[n - 1]) * (x[n - 3] - x[n - 1]) * (y[n - 2] - y[n - 1]) - (x[n - 2] - x[n - 1]) * (x[n - 2] - x[n - 1]) * (y[n - 3] - y[n - 1])) /
                         ((x[n - 3] - x[n - 1]) * (x[n - 2] - x[n - 1]) * (x[n - 3] - x[n - 2]));
  const double qn = 0.5, un = (3. / (x[n - 1] - x[n - 2])) * (dy_last - (y[n - 1] - y[n - 2]) / (x[n - 1] - x[n - 2]));
  dd[n - 1] = (un - qn * u[n - 2]) / (qn * dd[n - 2] + 1.0);
  for (int k = n - 2; k >= 0; k--) dd[k] = dd[k] * dd[k + 1] + u[k];
}
double cpt_sigma_of_R(const double* kk, const double* pk, int nk, double R, double k_per_decade) {
  const double PI = 3.1415926535897932384626433832795e0;
  std::vector<double> lnk(nk), lnpk(nk), dd(nk);
  for (int i = 0; i < nk; i++) { lnk[i] = log(kk[i]); lnpk[i] = log(pk[i]); }
  spline_est_deriv(lnk.data(), lnpk.data(), nk, dd.data());
  const int n = (int)(log(kk[nk - 1] / kk[0]) / log(10.) * k_per_decade) + 1;
  std::vector<double> xs(n), ys(n), d2(n);
  int last = 0;
  for (int i = 0; i < n; i++) {
    double k = kk[0] * pow(10., i / k_per_decade), p;
    if (i == 0) p = exp(lnpk[0]);
    else {   // array_interpolate_spline at ln k
      const double v = log(k);
      int inf = 0, sup = nk - 1;
      while (sup - inf > 1) { const int mid = (int)(0.5 * (inf + sup)); if (v < lnk[mid]) sup = mid; else inf = mid; }
      last = inf;
      const double h = lnk[sup] - lnk[inf], b = (v - lnk[inf]) / h, a = 1 - b;
      p = exp(a * lnpk[inf] + b * lnpk[sup] + ((a * a * a - a) * dd[inf] + (b * b * b - b) * dd[sup]) * h * h / 6.);
    }
    const double t = 1. / (1. + k);
    if (i == (n - 1)) k *= 0.9999999;
    const double x = k * R;
    const double W = (x < 0.01) ? 1. - x * x / 10. : 3. / x / x / x * (sin(x) - x * cos(x));
    xs[n - 1 - i] = t;
    ys[n - 1 - i] = k * k * k * p * W * W / (t * (1. - t));
  }
  (void)last;
  spline_est_deriv(xs.data(), ys.data(), n, d2.data());
  double res = 0.;
  for (int i = 0; i < n - 1; i++) {
    const double h = xs[i + 1] - xs[i];
    res += (ys[i] + ys[i + 1]) * h / 2. + (d2[i] + d2[i + 1]) * h * h * h / 24.;
  }
  return sqrt(res / (2. * PI * PI));
}

// host post-processing of the linear P(k): the device result is copied back (nk doubles) and integrated on the host
int cpt_sigma_impl(cpt_handle* h, const cpt_spectra_params* sp, const double* k, int nk, double R, double k_per_decade, double* sigma, int cb) {
  double* d_pk = nullptr;
  CPT_HIP(h, hipMalloc((void**)&d_pk, nk * sizeof(double)));
  int rc = cpt_pk_impl(h, sp, k, nk, d_pk, cb);
  if (!rc && hipStreamSynchronize(h->stream) != hipSuccess) rc = cpt_fail(h, CPT_ERR_NO_DEVICE, "hipStreamSynchronize failed");
  std::vector<double> pk(nk);
  if (!rc && hipMemcpy(pk.data(), d_pk, nk * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) rc = cpt_fail(h, CPT_ERR_NO_DEVICE, "hipMemcpy of P(k) failed");
  (void)hipFree(d_pk);
  if (rc) return rc;
  for (int i = 0; i < nk; i++) if (!(pk[i] > 0.)) return cpt_fail(h, CPT_ERR_RUNTIME, "P(k) is not positive at k[%d]", i);
  *sigma = cpt_sigma_of_R(k, pk.data(), nk, R, k_per_decade);
  return CPT_OK;
}

// cb: the baryon + cold dark matter spectrum P_cb (delta_cb source, only defined with non-cold species; nonlinear_module.cpp:1749-1760, 1952-1991)
int cpt_pk_impl(cpt_handle* h, const cpt_spectra_params* sp, const double* k, int nk, double* pk_dev, int cb) {
  const cpt_config& c = h->cfg;
  const int tp = cb ? c.index_tp_delta_cb : c.index_tp_delta_m;
  if (cb && !c.has_ncdm) return cpt_fail(h, CPT_ERR_INVALID, "P_cb(k) is only defined with non-cold species (has_pk_cb, nonlinear_module.cpp:1749)");
  if (tp < 0) return cpt_fail(h, CPT_ERR_INVALID, "P(k) requested but %s was not among the source types", cb ? "delta_cb" : "delta_m");
  if (!h->d_src || h->src_nk != nk) return cpt_fail(h, CPT_ERR_INVALID, "no resident sources for %d k-modes: run cpt_perturb_solve_batch first", nk);
  int rc;
  if ((rc = cpt_reserve(h, &h->d_pk_k, &h->pk_k_cap, (size_t)nk))) return rc;
  if ((rc = cpt_upload(h, h->d_pk_k, k, nk * sizeof(double)))) return rc;
  hipLaunchKernelGGL(k_pk, dim3((nk + 63) / 64), dim3(64), 0, h->stream, h->d_src, h->d_pk_k, pk_dev, nk, h->src_ntau,
                     tp, sp->A_s, sp->n_s, sp->alpha_s, sp->k_pivot);
  CPT_HIP(h, hipGetLastError());
  return CPT_OK;
}

// P(k, z) from the resident sources: see k_pk_z.  tau_z: conformal time of the redshift; ln_tau_size: length of the tail of the sampling that
// covers 0 <= z <= z_max_pk (cpt_host_ln_tau_size / PerturbationsModule::ln_tau_size_)
int cpt_pk_at_tau_impl(cpt_handle* h, const cpt_spectra_params* sp, const double* k, int nk, int ln_tau_size, double tau_z, int cb, double* pk_dev) {
  const cpt_config& c = h->cfg;
  const int tp = cb ? c.index_tp_delta_cb : c.index_tp_delta_m;
  if (cb && !c.has_ncdm) return cpt_fail(h, CPT_ERR_INVALID, "P_cb(k) is only defined with non-cold species (has_pk_cb, nonlinear_module.cpp:1749)");
  if (tp < 0) return cpt_fail(h, CPT_ERR_INVALID, "P(k) requested but %s was not among the source types", cb ? "delta_cb" : "delta_m");
  if (!h->d_src || h->src_nk != nk) return cpt_fail(h, CPT_ERR_INVALID, "no resident sources for %d k-modes: run cpt_perturb_solve_batch first", nk);
  const int ntau = h->src_ntau;
  if (!h->geo_pt_valid || (int)h->geo_pt_tau.size() != ntau || (int)h->geo_pt_k.size() != nk)
    return cpt_fail(h, CPT_ERR_INVALID, "P(k, z) needs the sampling times of the resident sources: they were not left by cpt_perturb_solve_batch / cpt_step on this handle");
  if (ln_tau_size < 2 || ln_tau_size > ntau)
    return cpt_fail(h, CPT_ERR_INVALID, "You are asking for the matter power spectrum at z > 0 but the sources were only stored for z = 0 (ln_tau_size = %d). "
                                        "You probably forgot to pass the input parameter z_max_pk (nonlinear_module.cpp:129-131)", ln_tau_size);
  const double* ts = h->geo_pt_tau.data() + (ntau - ln_tau_size);
  double ln_tau = log(tau_z);
  const double lo = log(ts[0]), hi = log(ts[ln_tau_size - 1]), EPS = 1e-10;
  if (ln_tau < lo - EPS)
    return cpt_fail(h, CPT_ERR_INVALID, "requested z was not inside of tau tabulation range (Requested ln(tau_=%.10e, Min %.10e). Solution might be to increase input "
                                        "parameter z_max_pk (nonlinear_module.cpp:144-146)", ln_tau, lo);
  if (ln_tau > hi + EPS) return cpt_fail(h, CPT_ERR_INVALID, "requested z was not inside of tau tabulation range (Requested ln(tau_=%.10e, Max %.10e)", ln_tau, hi);
  ln_tau = fmin(fmax(ln_tau, lo), hi);
  int rc;
  if ((rc = cpt_reserve(h, &h->d_pk_k, &h->pk_k_cap, (size_t)nk))) return rc;
  if ((rc = cpt_upload(h, h->d_pk_k, k, nk * sizeof(double)))) return rc;
  if ((rc = cpt_reserve(h, &h->d_pkz, &h->pkz_cap, (size_t)2 * ln_tau_size * nk))) return rc;
  const double* d_tau = (const double*)h->d_pt_scratch + nk;   // (cpt_perturb_impl: k[nk] tau[ntau] | ...)
  hipLaunchKernelGGL(k_pk_z, dim3((nk + 63) / 64), dim3(64), 0, h->stream, h->d_src, h->d_pk_k, d_tau, pk_dev, h->d_pkz, h->d_pkz + (size_t)ln_tau_size * nk, nk, ntau,
                     ln_tau_size, tp, ln_tau, sp->A_s, sp->n_s, sp->alpha_s, sp->k_pivot);
  CPT_HIP(h, hipGetLastError());
  return CPT_OK;
}

int cpt_sigma_at_tau_impl(cpt_handle* h, const cpt_spectra_params* sp, const double* k, int nk, int ln_tau_size, double tau_z, int cb, double R, double k_per_decade,
                          double* sigma) {
  double* d_pk = nullptr;
  CPT_HIP(h, hipMalloc((void**)&d_pk, nk * sizeof(double)));
  int rc = cpt_pk_at_tau_impl(h, sp, k, nk, ln_tau_size, tau_z, cb, d_pk);
  if (!rc && hipStreamSynchronize(h->stream) != hipSuccess) rc = cpt_fail(h, CPT_ERR_NO_DEVICE, "hipStreamSynchronize failed");
  std::vector<double> pk(nk);
  if (!rc && hipMemcpy(pk.data(), d_pk, nk * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) rc = cpt_fail(h, CPT_ERR_NO_DEVICE, "hipMemcpy of P(k) failed");
  (void)hipFree(d_pk);
  if (rc) return rc;
  for (int i = 0; i < nk; i++) if (!(pk[i] > 0.)) return cpt_fail(h, CPT_ERR_RUNTIME, "P(k) is not positive at k[%d]", i);
  *sigma = cpt_sigma_of_R(k, pk.data(), nk, R, k_per_decade);
  return CPT_OK;
}
