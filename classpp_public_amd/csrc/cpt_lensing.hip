// "Next" row S8f-4: CMB lensing of the C_l's on the device (LensingModule::lensing_init, source/lensing_module.cpp:149-854).
//
//   unlensed C_l table on the l grid  --spline in l-->  C_l at every integer l             (k_cl_spline, k_cl_full)
//   Wigner d^l_{mn}(mu) for the 12 (m,n) pairs, all l, all angles                          (k_lens_fac, k_lens_d)
//   Cgl(mu), Cgl2(mu) -> sigma2(mu); lensed correlation functions ksi, ksiX, ksi+, ksi-    (k_lens_cgl, k_lens_ksi)
//   back to harmonic space at the l values of the lensed table                             (k_lens_out)
//
// The reference computes the twelve d-function tables with twelve copies of one three-term recurrence
// (lensing_module.cpp:1261-1935); here the recurrence is written once for general (m, n) and one thread owns one
// (m, n, mu) ladder.  The table (12 x num_mu x (l_max+1) doubles: 75 MB in fast mode for l_max = 3000) lives in HBM
// like the reference's buf_dxx lives in RAM; the reductions over l (per angle) and over angles (per l) are one
// workgroup each.  Both quadratures of the reference are supported: the default Riemann sum of the correlation
// function DIFFERENCE on theta in (0, pi/16] with the unlensed spectrum added back, and Gauss-Legendre nodes.
#include <cmath>
#include <vector>

#include "cpt_internal.h"

namespace {
constexpr int NM = 12;
enum { D00, D11, D1M1, D2M2, D20, D3M1, D4M2, D22, D31, D3M3, D40, D4M4 };
__constant__ int c_m[NM] = {0, 1, 1, 2, 2, 3, 4, 2, 3, 3, 4, 4};
__constant__ int c_n[NM] = {0, 1, -1, -2, 0, -1, -2, 2, 1, -3, 0, -4};

// spline in l of every spectrum (array_spline_table_lines, _SPLINE_EST_DERIV_; spectra_module.cpp:926-934): one thread
// per spectrum type, the table has ~100 rows
__global__ void k_cl_spline(const double* __restrict__ cl, const int* __restrict__ l, int nl, int ct_size, double* __restrict__ dd,
                            double* __restrict__ u) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ct_size) return;
  auto X = [&](int i) { return (double)l[i]; };
  auto Y = [&](int i) { return cl[(size_t)i * ct_size + c]; };
  auto DD = [&](int i) -> double& { return dd[(size_t)i * ct_size + c]; };
  auto U = [&](int i) -> double& { return u[(size_t)i * ct_size + c]; };
  const int n = nl;
  const double dy_first = ((X(2) - X(0)) * (X(2) - X(0)) * (Y(1) - Y(0)) - (X(1) - X(0)) * (X(1) - X(0)) * (Y(2) - Y(0))) /
                          ((X(2) - X(0)) * (X(1) - X(0)) * (X(2) - X(1)));
  DD(0) = -0.5;
  U(0) = (3. / (X(1) - X(0))) * ((Y(1) - Y(0)) / (X(1) - X(0)) - dy_first);
  for (int i = 1; i < n - 1; i++) {
    const double sig = (X(i) - X(i - 1)) / (X(i + 1) - X(i - 1));
    const double p = sig * DD(i - 1) + 2.0;
    DD(i) = (sig - 1.0) / p;
    const double ui = (Y(i + 1) - Y(i)) / (X(i + 1) - X(i)) - (Y(i) - Y(i - 1)) / (X(i) - X(i - 1));
    U(i) = (6.0 * ui / (X(i + 1) - X(i - 1)) - sig * U(i - 1)) / p;
  }
  const double dy_last = ((X(n - 3) - X(n - 1)) * (X(n - 3) - X(n - 1)) * (Y(n - 2) - Y(n - 1)) -
                          (X(n - 2) - X(n - 1)) * (X(n - 2) - X(n - 1)) * (Y(n - 3) - Y(n - 1))) /
                         ((X(n - 3) - X(n - 1)) * (X(n - 2) - X(n - 1)) * (X(n - 3) - X(n - 2)));
  const double un = (3. / (X(n - 1) - X(n - 2))) * (dy_last - (Y(n - 1) - Y(n - 2)) / (X(n - 1) - X(n - 2)));
  DD(n - 1) = (un - 0.5 * U(n - 2)) / (0.5 * DD(n - 2) + 1.0);
  for (int i = n - 2; i >= 0; i--) DD(i) = DD(i) * DD(i + 1) + U(i);
}

// C_l at every integer l <= lmax (SpectraModule::spectra_cl_at_l, spectra_module.cpp:146-218): full [ct][lmax+1]
__global__ void k_cl_full(const double* __restrict__ cl, const double* __restrict__ dd, const int* __restrict__ l, int nl, int ct_size,
                          int lmax, double* __restrict__ full) {
  const int L = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y;
  if (L > lmax) return;
  double v = 0.;
  if (L >= 2 && L <= l[nl - 1]) {
    int inf = 0, sup = nl - 1;
    while (sup - inf > 1) { const int mid = (inf + sup) >> 1; if (L < l[mid]) sup = mid; else inf = mid; }
    if (inf > nl - 2) inf = nl - 2;
    const double x0 = l[inf], x1 = l[inf + 1], h = x1 - x0, b = (L - x0) / h, a = 1. - b;
    v = a * cl[(size_t)inf * ct_size + c] + b * cl[(size_t)(inf + 1) * ct_size + c] +
        ((a * a * a - a) * dd[(size_t)inf * ct_size + c] + (b * b * b - b) * dd[(size_t)(inf + 1) * ct_size + c]) * h * h / 6.;
  }
  full[(size_t)c * (lmax + 1) + L] = v;
}

// coefficients of  D_{l+1} = f1 (mu - f2) D_l - f3 D_{l-1},  d_{l+1} = f4 D_{l+1},  D_l = sqrt((2l+1)/2) d^l_{mn}
// (general form of lensing_module.cpp:1272-1280 and its eleven siblings): fac [NM][4][lmax+1]
__global__ void k_lens_fac(int lmax, double* __restrict__ fac) {
  const int l = blockIdx.x * blockDim.x + threadIdx.x, w = blockIdx.y;
  if (l > lmax) return;
  const int m = c_m[w], n = c_n[w];
  double f1 = 0., f2 = 0., f3 = 0., f4 = 0.;
  if (l >= 1 && l >= m) {
    const double ll = l, a = (ll + 1) * (ll + 1);
    const double den = sqrt((a - m * m) * (a - n * n));
    f1 = sqrt((2 * ll + 3) / (2 * ll + 1)) * (ll + 1) * (2 * ll + 1) / den;
    f2 = (double)(m * n) / (ll * (ll + 1));
    f3 = sqrt((2 * ll + 3) / (2 * ll - 1)) * sqrt((ll * ll - m * m) * (ll * ll - n * n)) / den * (ll + 1) / ll;
    f4 = sqrt(2. / (2 * ll + 3));
  }
  double* f = fac + (size_t)w * 4 * (lmax + 1);
  f[l] = f1; f[(lmax + 1) + l] = f2; f[2 * (lmax + 1) + l] = f3; f[3 * (lmax + 1) + l] = f4;
}

// one thread per (m,n) ladder and angle: d [NM][num_mu][lmax+1]
__global__ void k_lens_d(const double* __restrict__ mu, int num_mu, int lmax, const double* __restrict__ fac, double* __restrict__ d) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, w = blockIdx.y;
  if (i >= num_mu) return;
  const int m = c_m[w], n = c_n[w];
  const double x = mu[i];
  double* out = d + ((size_t)w * num_mu + i) * (lmax + 1);
  for (int l = 0; l < m && l <= lmax; l++) out[l] = 0.;
  // d^m_{mn} = sqrt((2m)!/((m+n)!(m-n)!)) cos(b/2)^(m+n) sin(b/2)^(m-n)
  double fm2 = 1., fpn = 1., fmn = 1.;
  for (int j = 2; j <= 2 * m; j++) fm2 *= j;
  for (int j = 2; j <= m + n; j++) fpn *= j;
  for (int j = 2; j <= m - n; j++) fmn *= j;
  double val = sqrt(fm2 / (fpn * fmn));
  int pc = m + n, ps = m - n;
  while (pc >= 2) { val *= (1. + x) / 2.; pc -= 2; }
  while (ps >= 2) { val *= (1. - x) / 2.; ps -= 2; }
  if (pc == 1 && ps == 1) val *= sqrt(1. - x * x) / 2.;
  double Dm1 = 0., D = val * sqrt((2. * m + 1.) / 2.);
  if (m <= lmax) out[m] = val;
  int l = m;
  if (m == 0) {
    if (lmax >= 1) { Dm1 = D; D = x * sqrt(1.5); out[1] = x; }
    l = 1;
  }
  const double* f1 = fac + (size_t)w * 4 * (lmax + 1);
  const double* f2 = f1 + (lmax + 1);
  const double* f3 = f2 + (lmax + 1);
  const double* f4 = f3 + (lmax + 1);
  for (; l < lmax; l++) {
    const double Dp1 = f1[l] * (x - f2[l]) * D - f3[l] * Dm1;
    out[l + 1] = Dp1 * f4[l];
    Dm1 = D; D = Dp1;
  }
}

template <int N>
__device__ void block_sum(double (&v)[N], double* red /* [4][N] */) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int c = 0; c < N; c++) {
    double x = v[c];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
    if ((tid & 63) == 0) red[(tid >> 6) * N + c] = x;
  }
  __syncthreads();
#pragma unroll
  for (int c = 0; c < N; c++) v[c] = red[c] + red[N + c] + red[2 * N + c] + red[3 * N + c];
}

// Cgl(mu), Cgl2(mu) (lensing_module.cpp:561-579): one workgroup of 256 per angle
__global__ void __launch_bounds__(256) k_lens_cgl(const double* __restrict__ d, const double* __restrict__ cl_pp, int num_mu, int lmax,
                                                 double* __restrict__ Cgl, double* __restrict__ Cgl2) {
  __shared__ double red[4 * 2];
  const int i = blockIdx.x;
  const double* d11 = d + ((size_t)D11 * num_mu + i) * (lmax + 1);
  const double* d1m1 = d + ((size_t)D1M1 * num_mu + i) * (lmax + 1);
  double acc[2] = {0., 0.};
  for (int L = 2 + threadIdx.x; L <= lmax; L += 256) {
    const double wgt = (2. * L + 1.) * L * (L + 1.) * cl_pp[L];
    acc[0] = fma(wgt, d11[L], acc[0]);
    acc[1] = fma(wgt, d1m1[L], acc[1]);
  }
  block_sum<2>(acc, red);
  if (threadIdx.x == 0) {
    const double PI = 3.1415926535897932384626433832795e0;
    Cgl[i] = acc[0] / (4. * PI); Cgl2[i] = acc[1] / (4. * PI);
  }
}

struct KsiParams {
  const double* d; const double* full;  // [ct][lmax+1]
  const double* Cgl; const double* Cgl2;
  double* ksi;  // [4][num_mu-1]: ksi, ksiX, ksip, ksim
  int num_mu, lmax, ct_tt, ct_te, ct_ee, ct_bb, accurate;
};

// lensed correlation functions at one angle (lensing_module.cpp:641-749): one workgroup of 256 per angle
__global__ void __launch_bounds__(256) k_lens_ksi(KsiParams P) {
  __shared__ double red[4 * 4];
  const int i = blockIdx.x, num_mu = P.num_mu, lmax = P.lmax;
  const double PI = 3.1415926535897932384626433832795e0;
  const double s2 = P.Cgl[num_mu - 1] - P.Cgl[i], c2 = P.Cgl2[i];   // sigma2 = Cgl(1) - Cgl(mu)
  const bool has_tt = P.ct_tt >= 0, has_te = P.ct_te >= 0, has_pol = P.ct_ee >= 0 || P.ct_bb >= 0;
  auto Dp = [&](int w) { return P.d + ((size_t)w * num_mu + i) * (lmax + 1); };
  const double *d00 = Dp(D00), *d11 = Dp(D11), *d1m1 = Dp(D1M1), *d2m2 = Dp(D2M2), *d20 = Dp(D20), *d3m1 = Dp(D3M1), *d4m2 = Dp(D4M2),
               *d22 = Dp(D22), *d31 = Dp(D31), *d3m3 = Dp(D3M3), *d40 = Dp(D40), *d4m4 = Dp(D4M4);
  const double* cl_tt = has_tt ? P.full + (size_t)P.ct_tt * (lmax + 1) : nullptr;
  const double* cl_te = has_te ? P.full + (size_t)P.ct_te * (lmax + 1) : nullptr;
  const double* cl_ee = P.ct_ee >= 0 ? P.full + (size_t)P.ct_ee * (lmax + 1) : nullptr;
  const double* cl_bb = P.ct_bb >= 0 ? P.full + (size_t)P.ct_bb * (lmax + 1) : nullptr;
  double acc[4] = {0., 0., 0., 0.};
  for (int L = 2 + threadIdx.x; L <= lmax; L += 256) {
    const double ll = L, fac = ll * (ll + 1) / 4., fac1 = (2 * ll + 1) / (4. * PI);
    const double sqrt1 = sqrt((ll + 2) * (ll + 1) * ll * (ll - 1)), sqrt2 = sqrt((ll + 2) * (ll - 1)), sqrt3 = sqrt((ll + 3) * (ll - 2)),
                 sqrt4 = sqrt((ll + 4) * (ll + 3) * (ll - 2.) * (ll - 3)), sqrt5 = sqrt(ll * (ll + 1));
    const double X_000 = exp(-fac * s2), X_p000 = -fac * X_000, X_220 = 0.25 * sqrt1 * X_000;
    double X_022 = 0., X_p022 = 0., X_242 = 0., X_121 = 0., X_132 = 0.;
    if (has_te || has_pol) {
      X_022 = X_000 * (1 + s2 * (1 + 0.5 * s2));
      X_p022 = -(fac - 1.) * X_022;
      X_242 = 0.25 * sqrt4 * X_000;
      if (has_pol) { X_121 = -0.5 * sqrt2 * X_000 * (1 + 2. / 3. * s2); X_132 = -0.5 * sqrt3 * X_000 * (1 + 5. / 3. * s2); }
    }
    if (has_tt) {
      double lens = X_000 * X_000 * d00[L] + X_p000 * X_p000 * d1m1[L] * c2 * 8. / (ll * (ll + 1)) +
                    (X_p000 * X_p000 * d00[L] + X_220 * X_220 * d2m2[L]) * c2 * c2;
      if (!P.accurate) lens -= d00[L];
      acc[0] = fma(fac1 * cl_tt[L], lens, acc[0]);
    }
    if (has_te) {
      double lens = X_022 * X_000 * d20[L] + c2 * 2. * X_p000 / sqrt5 * (X_121 * d11[L] + X_132 * d3m1[L]) +
                    0.5 * c2 * c2 * ((2. * X_p022 * X_p000 + X_220 * X_220) * d20[L] + X_220 * X_242 * d4m2[L]);
      if (!P.accurate) lens -= d20[L];
      acc[1] = fma(fac1 * cl_te[L], lens, acc[1]);
    }
    if (has_pol) {
      double lensp = X_022 * X_022 * d22[L] + 2. * c2 * X_132 * X_121 * d31[L] + c2 * c2 * (X_p022 * X_p022 * d22[L] + X_242 * X_220 * d40[L]);
      double lensm = X_022 * X_022 * d2m2[L] + c2 * (X_121 * X_121 * d1m1[L] + X_132 * X_132 * d3m3[L]) +
                     0.5 * c2 * c2 * (2. * X_p022 * X_p022 * d2m2[L] + X_220 * X_220 * d00[L] + X_242 * X_242 * d4m4[L]);
      if (!P.accurate) { lensp -= d22[L]; lensm -= d2m2[L]; }
      const double ee = cl_ee ? cl_ee[L] : 0., bb = cl_bb ? cl_bb[L] : 0.;
      acc[2] = fma(fac1 * (ee + bb), lensp, acc[2]);
      acc[3] = fma(fac1 * (ee - bb), lensm, acc[3]);
    }
  }
  block_sum<4>(acc, red);
  if (threadIdx.x < 4) P.ksi[(size_t)threadIdx.x * (num_mu - 1) + i] = acc[threadIdx.x];
}

struct OutParams {
  const double* d; const double* ksi; const double* w8; const double* full; const double* cl;  // cl: unlensed table [nl][ct]
  const int* l;
  double* out;  // [l_size][ct]
  int num_mu, lmax, ct_size, ct_tt, ct_te, ct_ee, ct_bb, accurate;
};

// lensed C_l at one l of the output table (lensing_module.cpp:1094-1246): one wavefront per l
__global__ void __launch_bounds__(64) k_lens_out(OutParams P) {
  const int il = blockIdx.x, L = P.l[il], nmu = P.num_mu - 1, lane = threadIdx.x;
  const size_t row = (size_t)(P.lmax + 1);
  double a = 0., b = 0., cp = 0., cm = 0.;
  for (int i = lane; i < nmu; i += 64) {
    const double w = P.w8[i];
    a = fma(P.ksi[i] * w, P.d[((size_t)D00 * P.num_mu + i) * row + L], a);
    b = fma(P.ksi[nmu + i] * w, P.d[((size_t)D20 * P.num_mu + i) * row + L], b);
    cp = fma(P.ksi[2 * (size_t)nmu + i] * w, P.d[((size_t)D22 * P.num_mu + i) * row + L], cp);
    cm = fma(P.ksi[3 * (size_t)nmu + i] * w, P.d[((size_t)D2M2 * P.num_mu + i) * row + L], cm);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    a += __shfl_down(a, off, 64); b += __shfl_down(b, off, 64); cp += __shfl_down(cp, off, 64); cm += __shfl_down(cm, off, 64);
  }
  // every type starts as the unlensed value (lensing_module.cpp:1038-1044), TT/TE/EE/BB are then replaced
  for (int c = lane; c < P.ct_size; c += 64) P.out[(size_t)il * P.ct_size + c] = P.cl[(size_t)il * P.ct_size + c];
  __syncthreads();
  if (lane == 0) {
    const double PI = 3.1415926535897932384626433832795e0;
    auto unl = [&](int ct) { return P.accurate ? 0. : P.full[(size_t)ct * row + L]; };
    if (P.ct_tt >= 0) P.out[(size_t)il * P.ct_size + P.ct_tt] = a * 2.0 * PI + unl(P.ct_tt);
    if (P.ct_te >= 0) P.out[(size_t)il * P.ct_size + P.ct_te] = b * 2.0 * PI + unl(P.ct_te);
    if (P.ct_ee >= 0) P.out[(size_t)il * P.ct_size + P.ct_ee] = (cp + cm) * PI + unl(P.ct_ee);
    if (P.ct_bb >= 0) P.out[(size_t)il * P.ct_size + P.ct_bb] = (cp - cm) * PI + unl(P.ct_bb);
  }
}

// Gauss-Legendre nodes and weights on [-1, 1] (what tools/quadrature.c quadrature_gauss_legendre provides): Newton
// iteration on P_n from the asymptotic root estimate, symmetric fill
void gauss_legendre_nodes(int n, double tol, double* mu, double* w) {
  const double PI = 3.1415926535897932384626433832795e0;
  for (int i = 0; i < (n + 1) / 2; i++) {
    double z = std::cos(PI * (i + 0.75) / (n + 0.5)), dz, dp;
    int it = 0;
    do {
      double pa = 1., pb = 0.;
      for (int j = 1; j <= n; j++) { const double pc = pb; pb = pa; pa = ((2. * j - 1.) * z * pb - (j - 1.) * pc) / j; }
      dp = n * (z * pa - pb) / (z * z - 1.);
      dz = pa / dp;
      z -= dz;
    } while (std::fabs(dz) > tol && ++it < 10000);
    mu[i] = -z; mu[n - 1 - i] = z;
    w[i] = w[n - 1 - i] = 2. / ((1. - z * z) * dp * dp);
  }
}
}  // namespace

int cpt_lensing_l_size_impl(const int* l, int nl, const cpt_lensing_params* lp) {
  const int l_lensed_max = lp->l_unlensed_max - lp->delta_l_max;  // lensing_indices, lensing_module.cpp:983-993
  int i;
  for (i = 0; (i < nl) && (l[i] <= l_lensed_max); i++) {}
  if (i < nl) i++;
  return (i + 1 <= nl) ? i + 1 : nl;
}

int cpt_lensing_impl(cpt_handle* h, const cpt_spectra_params* sp, const cpt_lensing_params* lp, const int* l, int nl,
                     const double* cl_dev, double* cl_lensed_dev) {
  const int lmax = lp->l_unlensed_max, ct = sp->ct_size;
  // (the kernels index the C_l table with these: an inconsistent cpt_spectra_params must not reach them)
  if (ct < 1 || ct > 8) return cpt_fail(h, CPT_ERR_INVALID, "ct_size=%d out of range", ct);
  {
    const int cts[7] = {sp->index_ct_tt, sp->index_ct_ee, sp->index_ct_te, sp->index_ct_bb, sp->index_ct_pp, sp->index_ct_tp, sp->index_ct_ep};
    for (int i = 0; i < 7; i++)
      if (cts[i] >= ct) return cpt_fail(h, CPT_ERR_INVALID, "index_ct_* >= ct_size");
  }
  if (sp->index_ct_pp < 0) return cpt_fail(h, CPT_ERR_INVALID, "lensing needs the lensing potential spectrum C_l^phiphi (lCl)");
  if (nl < 4) return cpt_fail(h, CPT_ERR_INVALID, "need at least 4 l values");
  for (int i = 1; i < nl; i++)
    if (l[i] <= l[i - 1]) return cpt_fail(h, CPT_ERR_INVALID, "l grid must be strictly increasing");
  if (lmax < 4 || lmax > l[nl - 1]) return cpt_fail(h, CPT_ERR_INVALID, "l_unlensed_max=%d outside the l grid (last l = %d)", lmax, l[nl - 1]);
  if (lp->delta_l_max < 0 || lp->delta_l_max >= lmax)
    return cpt_fail(h, CPT_ERR_INVALID, "you asked for lensed Cls with delta_l_max=%d >= l_max=%d", lp->delta_l_max, lmax);
  const int l_size = cpt_lensing_l_size_impl(l, nl, lp);
  int num_mu;
  if (lp->accurate_lensing) { num_mu = lmax + lp->num_mu_minus_lmax; num_mu += num_mu % 2; }
  else num_mu = (lmax * 2) / 16;
  if (num_mu < 3) return cpt_fail(h, CPT_ERR_INVALID, "l_max too small for the lensing quadrature");
  // angles and weights (lensing_module.cpp:251-292), cached per (mode, num_mu)
  if (h->lens_num_mu != num_mu || h->lens_accurate != lp->accurate_lensing || h->lens_lmax != lmax) {
    std::vector<double> mu(num_mu), w8(num_mu);
    mu[num_mu - 1] = 1.0; w8[num_mu - 1] = 0.;
    const double PI = 3.1415926535897932384626433832795e0;
    if (lp->accurate_lensing) gauss_legendre_nodes(num_mu - 1, lp->tol_gauss_legendre > 0 ? lp->tol_gauss_legendre : 1e-14, mu.data(), w8.data());
    else {
      const double dth = PI / 16. / (double)(num_mu - 1);
      for (int i = 0; i < num_mu - 1; i++) { const double th = (i + 1) * dth; mu[i] = std::cos(th); w8[i] = std::sin(th) * dth; }
    }
    int rc;
    const size_t need = (size_t)NM * num_mu * (lmax + 1) + (size_t)NM * 4 * (lmax + 1) + (size_t)8 * num_mu + (size_t)(3 * nl + lmax + 1) * 8 + 64;
    if ((rc = cpt_reserve(h, &h->d_lens, &h->lens_cap, need))) return rc;
    double* p = h->d_lens;
    h->lens_d = p; p += (size_t)NM * num_mu * (lmax + 1);
    h->lens_fac = p; p += (size_t)NM * 4 * (lmax + 1);
    h->lens_mu = p; p += num_mu;
    h->lens_w8 = p; p += num_mu;
    h->lens_cgl = p; p += 2 * num_mu;
    h->lens_ksi = p; p += 4 * num_mu;
    h->lens_work = p;
    if ((rc = cpt_upload(h, h->lens_mu, mu.data(), num_mu * sizeof(double)))) return rc;
    if ((rc = cpt_upload(h, h->lens_w8, w8.data(), num_mu * sizeof(double)))) return rc;
    hipLaunchKernelGGL(k_lens_fac, dim3((lmax + 256) / 256, NM), dim3(256), 0, h->stream, lmax, h->lens_fac);
    hipLaunchKernelGGL(k_lens_d, dim3((num_mu + 63) / 64, NM), dim3(64), 0, h->stream, h->lens_mu, num_mu, lmax, h->lens_fac, h->lens_d);
    CPT_HIP(h, hipGetLastError());
    h->lens_num_mu = num_mu; h->lens_accurate = lp->accurate_lensing; h->lens_lmax = lmax;
  }
  // work area: l grid (as int), ddcl, u, full spectra
  int rc;
  if ((rc = cpt_reserve(h, &h->d_lens_l, &h->lens_l_cap, (size_t)nl))) return rc;
  if ((rc = cpt_upload(h, h->d_lens_l, l, nl * sizeof(int)))) return rc;
  const size_t wneed = (size_t)2 * nl * ct + (size_t)ct * (lmax + 1);
  if ((rc = cpt_reserve(h, &h->d_lens_w, &h->lens_w_cap, wneed))) return rc;
  double* dd = h->d_lens_w;
  double* u = dd + (size_t)nl * ct;
  double* full = u + (size_t)nl * ct;
  hipLaunchKernelGGL(k_cl_spline, dim3(1), dim3(64), 0, h->stream, cl_dev, h->d_lens_l, nl, ct, dd, u);
  hipLaunchKernelGGL(k_cl_full, dim3((lmax + 256) / 256, ct), dim3(256), 0, h->stream, cl_dev, dd, h->d_lens_l, nl, ct, lmax, full);
  hipLaunchKernelGGL(k_lens_cgl, dim3(num_mu), dim3(256), 0, h->stream, h->lens_d, full + (size_t)sp->index_ct_pp * (lmax + 1), num_mu, lmax,
                     h->lens_cgl, h->lens_cgl + num_mu);
  KsiParams K;
  K.d = h->lens_d; K.full = full; K.Cgl = h->lens_cgl; K.Cgl2 = h->lens_cgl + num_mu; K.ksi = h->lens_ksi; K.num_mu = num_mu; K.lmax = lmax;
  K.ct_tt = sp->index_ct_tt; K.ct_te = sp->index_ct_te; K.ct_ee = sp->index_ct_ee; K.ct_bb = sp->index_ct_bb; K.accurate = lp->accurate_lensing;
  hipLaunchKernelGGL(k_lens_ksi, dim3(num_mu - 1), dim3(256), 0, h->stream, K);
  OutParams O;
  O.d = h->lens_d; O.ksi = h->lens_ksi; O.w8 = h->lens_w8; O.full = full; O.cl = cl_dev; O.l = h->d_lens_l; O.out = cl_lensed_dev;
  O.num_mu = num_mu; O.lmax = lmax; O.ct_size = ct; O.ct_tt = sp->index_ct_tt; O.ct_te = sp->index_ct_te; O.ct_ee = sp->index_ct_ee;
  O.ct_bb = sp->index_ct_bb; O.accurate = lp->accurate_lensing;
  hipLaunchKernelGGL(k_lens_out, dim3(l_size), dim3(64), 0, h->stream, O);
  CPT_HIP(h, hipGetLastError());
  return CPT_OK;
}
