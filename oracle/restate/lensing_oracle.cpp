// ORACLE / TEST INFRASTRUCTURE ONLY -- CPU restatement of the CMB lensing of the C_l's (LensingModule::lensing_init,
// source/lensing_module.cpp:149-854, with lensing_indices :886-1081, the quadratures :1094-1246 and the Wigner
// d-function recurrences :1261-1935), fast mode (accurate_lensing = no: Riemann sum of the correlation-function
// DIFFERENCE on theta in (0, pi/16], unlensed spectrum added back) and accurate mode (Gauss-Legendre nodes on [-1,1],
// tools/quadrature.c quadrature_gauss_legendre).  Pinned by tests/test_oracle_lensing.py against the reference's own
// cl_lens_ table (tests/golden/explanatory.npz, keys le.*).
//
// The twelve reference routines lensing_d00 ... lensing_d4m4 are instances of one three-term recurrence for
// D^l_{mn} = sqrt((2l+1)/2) d^l_{mn}(mu) (Kostelec & Rockmore 2003); here it is written once, for general (m, n).
#include <cmath>
#include <vector>

#include "../../include/cpt.h"

namespace {
const double PI = 3.1415926535897932384626433832795e0;

void spline_est_deriv(const double* x, int n, const double* y, double* ddy, double* u) {  // arrays.c _SPLINE_EST_DERIV_
  double dy_first = ((x[2] - x[0]) * (x[2] - x[0]) * (y[1] - y[0]) - (x[1] - x[0]) * (x[1] - x[0]) * (y[2] - y[0])) /
                    ((x[2] - x[0]) * (x[1] - x[0]) * (x[2] - x[1]));
  ddy[0] = -0.5;
  u[0] = (3. / (x[1] - x[0])) * ((y[1] - y[0]) / (x[1] - x[0]) - dy_first);
  for (int i = 1; i < n - 1; i++) {
    double sig = (x[i] - x[i - 1]) / (x[i + 1] - x[i - 1]);
    double p = sig * ddy[i - 1] + 2.0;
    ddy[i] = (sig - 1.0) / p;
    double ui = (y[i + 1] - y[i]) / (x[i + 1] - x[i]) - (y[i] - y[i - 1]) / (x[i] - x[i - 1]);
    u[i] = (6.0 * ui / (x[i + 1] - x[i - 1]) - sig * u[i - 1]) / p;
  }
  double dy_last = ((x[n - 3] - x[n - 1]) * (x[n - 3] - x[n - 1]) * (y[n - 2] - y[n - 1]) -
                    (x[n - 2] - x[n - 1]) * (x[n - 2] - x[n - 1]) * (y[n - 3] - y[n - 1])) /
                   ((x[n - 3] - x[n - 1]) * (x[n - 2] - x[n - 1]) * (x[n - 3] - x[n - 2]));
  double un = (3. / (x[n - 1] - x[n - 2])) * (dy_last - (y[n - 1] - y[n - 2]) / (x[n - 1] - x[n - 2]));
  ddy[n - 1] = (un - 0.5 * u[n - 2]) / (0.5 * ddy[n - 2] + 1.0);
  for (int i = n - 2; i >= 0; i--) ddy[i] = ddy[i] * ddy[i + 1] + u[i];
}

double factorial(int n) { double f = 1.; for (int i = 2; i <= n; i++) f *= i; return f; }

// d^l_{mn}(mu) for l = 0..lmax, m >= |n| >= 0 (lensing_module.cpp:1261-1935)
void wigner_d(int m, int n, double mu, int lmax, double* d) {
  const int l0 = m;
  for (int l = 0; l < l0 && l <= lmax; l++) d[l] = 0.;
  // d^m_{mn} = sqrt((2m)!/((m+n)!(m-n)!)) cos(b/2)^(m+n) sin(b/2)^(m-n), with cos^2 = (1+mu)/2, sin^2 = (1-mu)/2
  double start = std::sqrt(factorial(2 * m) / (factorial(m + n) * factorial(m - n)));
  // (m+n) and (m-n) have the same parity; odd powers only occur in pairs cos*sin = sqrt(1-mu^2)/2
  int pc = m + n, ps = m - n;
  double val = start;
  while (pc >= 2) { val *= (1. + mu) / 2.; pc -= 2; }
  while (ps >= 2) { val *= (1. - mu) / 2.; ps -= 2; }
  if (pc == 1 && ps == 1) val *= std::sqrt(1. - mu * mu) / 2.;
  double Dm1 = 0., D = val * std::sqrt((2. * l0 + 1.) / 2.);
  if (l0 <= lmax) d[l0] = val;
  int l = l0;
  if (l0 == 0) {  // Legendre start: D_1 = mu sqrt(3/2)
    if (lmax >= 1) { Dm1 = D; D = mu * std::sqrt(1.5); d[1] = mu; }
    l = 1;
  }
  for (; l < lmax; l++) {
    const double ll = l, a = (ll + 1) * (ll + 1);
    const double den = std::sqrt((a - m * m) * (a - n * n));
    const double f1 = std::sqrt((2 * ll + 3) / (2 * ll + 1)) * (ll + 1) * (2 * ll + 1) / den;
    const double f2 = (double)(m * n) / (ll * (ll + 1));
    const double f3 = std::sqrt((2 * ll + 3) / (2 * ll - 1)) * std::sqrt((ll * ll - m * m) * (ll * ll - n * n)) / den * (ll + 1) / ll;
    const double Dp1 = f1 * (mu - f2) * D - f3 * Dm1;
    d[l + 1] = Dp1 * std::sqrt(2. / (2 * ll + 3));
    Dm1 = D; D = Dp1;
  }
}

// roots of P_n and Gauss-Legendre weights (tools/quadrature.c quadrature_gauss_legendre: Newton iteration from the
// Chebyshev-like first guess, symmetric fill)
void gauss_legendre(int n, double tol, double* mu, double* w) {
  const int m = (n + 1) / 2;
  for (int i = 1; i <= m; i++) {
    double z = std::cos(PI * (i - 0.25) / (n + 0.5)), z1, pp;
    do {
      double p1 = 1., p2 = 0.;
      for (int j = 1; j <= n; j++) { double p3 = p2; p2 = p1; p1 = ((2. * j - 1.) * z * p2 - (j - 1.) * p3) / j; }
      pp = n * (z * p1 - p2) / (z * z - 1.);
      z1 = z; z = z1 - p1 / pp;
    } while (std::fabs(z - z1) > tol);
    mu[i - 1] = -z; mu[n - i] = z;
    w[i - 1] = 2. / ((1. - z * z) * pp * pp); w[n - i] = w[i - 1];
  }
}
}  // namespace

extern "C" {
// number of l values of the lensed table (lensing_indices, lensing_module.cpp:983-993)
int orc_lensing_l_size(const int* l, int nl, int l_unlensed_max, int delta_l_max) {
  const int l_lensed_max = l_unlensed_max - delta_l_max;
  int i;
  for (i = 0; (i < nl) && (l[i] <= l_lensed_max); i++) {}
  if (i < nl) i++;
  return (i + 1 <= nl) ? i + 1 : nl;
}

// cl [nl][ct] on the l grid (unlensed, as cpt_cl_batch / orc_cl return it)  ->  cl_lens [l_size][ct]
int orc_lensing(const cpt_spectra_params* s, const int* l, int nl, const double* cl, int l_unlensed_max, int delta_l_max,
                int accurate, int num_mu_minus_lmax, double tol_gl, double* cl_lens) {
  const int ct = s->ct_size, lmax = l_unlensed_max;
  const int l_size = orc_lensing_l_size(l, nl, l_unlensed_max, delta_l_max);
  const bool has_tt = s->index_ct_tt >= 0, has_te = s->index_ct_te >= 0, has_ee = s->index_ct_ee >= 0, has_bb = s->index_ct_bb >= 0;
  const bool has_pol = has_ee || has_bb;
  if (s->index_ct_pp < 0) return 1;
  // unlensed spectra at every integer l (spectra_cl_at_l: spline in l, spectra_module.cpp:146-218)
  std::vector<double> x(nl), y(nl), dd(nl), u(nl);
  for (int i = 0; i < nl; i++) x[i] = l[i];
  std::vector<std::vector<double>> full(ct, std::vector<double>(lmax + 1, 0.));
  for (int c = 0; c < ct; c++) {
    for (int i = 0; i < nl; i++) y[i] = cl[(size_t)i * ct + c];
    spline_est_deriv(x.data(), nl, y.data(), dd.data(), u.data());
    int inf = 0;
    for (int L = 2; L <= lmax; L++) {
      while (inf < nl - 2 && x[inf + 1] < L) inf++;
      double h = x[inf + 1] - x[inf], b = (L - x[inf]) / h, a = 1 - b;
      full[c][L] = a * y[inf] + b * y[inf + 1] + ((a * a * a - a) * dd[inf] + (b * b * b - b) * dd[inf + 1]) * h * h / 6.;
    }
  }
  const std::vector<double> zero(lmax + 1, 0.);
  const double* cl_tt = has_tt ? full[s->index_ct_tt].data() : zero.data();
  const double* cl_te = has_te ? full[s->index_ct_te].data() : zero.data();
  const double* cl_ee = has_ee ? full[s->index_ct_ee].data() : zero.data();
  const double* cl_bb = has_bb ? full[s->index_ct_bb].data() : zero.data();
  const double* cl_pp = full[s->index_ct_pp].data();
  // the table starts as a copy of the unlensed one (lensing_module.cpp:1038-1044)
  for (int i = 0; i < l_size; i++)
    for (int c = 0; c < ct; c++) cl_lens[(size_t)i * ct + c] = cl[(size_t)i * ct + c];
  // angles and weights (lensing_module.cpp:251-292)
  int num_mu;
  if (accurate) { num_mu = lmax + num_mu_minus_lmax; num_mu += num_mu % 2; }
  else num_mu = (lmax * 2) / 16;
  std::vector<double> mu(num_mu), w8(num_mu - 1);
  mu[num_mu - 1] = 1.0;
  if (accurate) gauss_legendre(num_mu - 1, tol_gl, mu.data(), w8.data());
  else {
    const double dth = PI / 16. / (double)(num_mu - 1);
    for (int i = 0; i < num_mu - 1; i++) { double th = (i + 1) * dth; mu[i] = std::cos(th); w8[i] = std::sin(th) * dth; }
  }
  // d functions
  const int NM = 12;
  const int mm[NM] = {0, 1, 1, 2, 2, 3, 4, 2, 3, 3, 4, 4}, nn[NM] = {0, 1, -1, -2, 0, -1, -2, 2, 1, -3, 0, -4};
  enum { D00, D11, D1M1, D2M2, D20, D3M1, D4M2, D22, D31, D3M3, D40, D4M4 };
  std::vector<std::vector<double>> d(NM, std::vector<double>((size_t)num_mu * (lmax + 1)));
  for (int k = 0; k < NM; k++)
    for (int i = 0; i < num_mu; i++) wigner_d(mm[k], nn[k], mu[i], lmax, &d[k][(size_t)i * (lmax + 1)]);
  auto D = [&](int k, int i, int L) { return d[k][(size_t)i * (lmax + 1) + L]; };
  // Cgl, Cgl2, sigma2 (lensing_module.cpp:561-584)
  std::vector<double> Cgl(num_mu), Cgl2(num_mu), sigma2(num_mu - 1);
  for (int i = 0; i < num_mu; i++) {
    double a = 0., b = 0.;
    for (int L = 2; L <= lmax; L++) {
      a += (2. * L + 1.) * L * (L + 1.) * cl_pp[L] * D(D11, i, L);
      b += (2. * L + 1.) * L * (L + 1.) * cl_pp[L] * D(D1M1, i, L);
    }
    Cgl[i] = a / (4. * PI); Cgl2[i] = b / (4. * PI);
  }
  for (int i = 0; i < num_mu - 1; i++) sigma2[i] = Cgl[num_mu - 1] - Cgl[i];
  // lensed correlation functions (lensing_module.cpp:624-749)
  std::vector<double> ksi(num_mu - 1, 0.), ksiX(num_mu - 1, 0.), ksip(num_mu - 1, 0.), ksim(num_mu - 1, 0.);
  for (int i = 0; i < num_mu - 1; i++) {
    for (int L = 2; L <= lmax; L++) {
      const double ll = L, fac = ll * (ll + 1) / 4., fac1 = (2 * ll + 1) / (4. * PI);
      const double sqrt1 = std::sqrt((ll + 2) * (ll + 1) * ll * (ll - 1)), sqrt2 = std::sqrt((ll + 2) * (ll - 1)),
                   sqrt3 = std::sqrt((ll + 3) * (ll - 2)), sqrt4 = std::sqrt((ll + 4) * (ll + 3) * (ll - 2.) * (ll - 3)),
                   sqrt5 = std::sqrt(ll * (ll + 1));
      const double s2 = sigma2[i], c2 = Cgl2[i];
      const double X_000 = std::exp(-fac * s2), X_p000 = -fac * X_000, X_220 = 0.25 * sqrt1 * X_000;
      double X_022 = 0., X_p022 = 0., X_242 = 0., X_121 = 0., X_132 = 0.;
      if (has_te || has_pol) {
        X_022 = X_000 * (1 + s2 * (1 + 0.5 * s2));
        X_p022 = -(fac - 1.) * X_022;
        X_242 = 0.25 * sqrt4 * X_000;
        if (has_pol) { X_121 = -0.5 * sqrt2 * X_000 * (1 + 2. / 3. * s2); X_132 = -0.5 * sqrt3 * X_000 * (1 + 5. / 3. * s2); }
      }
      if (has_tt) {
        double lens = X_000 * X_000 * D(D00, i, L) + X_p000 * X_p000 * D(D1M1, i, L) * c2 * 8. / (ll * (ll + 1)) +
                      (X_p000 * X_p000 * D(D00, i, L) + X_220 * X_220 * D(D2M2, i, L)) * c2 * c2;
        if (!accurate) lens -= D(D00, i, L);
        ksi[i] += fac1 * cl_tt[L] * lens;
      }
      if (has_te) {
        double lens = X_022 * X_000 * D(D20, i, L) + c2 * 2. * X_p000 / sqrt5 * (X_121 * D(D11, i, L) + X_132 * D(D3M1, i, L)) +
                      0.5 * c2 * c2 * ((2. * X_p022 * X_p000 + X_220 * X_220) * D(D20, i, L) + X_220 * X_242 * D(D4M2, i, L));
        if (!accurate) lens -= D(D20, i, L);
        ksiX[i] += fac1 * cl_te[L] * lens;
      }
      if (has_pol) {
        double lensp = X_022 * X_022 * D(D22, i, L) + 2. * c2 * X_132 * X_121 * D(D31, i, L) +
                       c2 * c2 * (X_p022 * X_p022 * D(D22, i, L) + X_242 * X_220 * D(D40, i, L));
        double lensm = X_022 * X_022 * D(D2M2, i, L) + c2 * (X_121 * X_121 * D(D1M1, i, L) + X_132 * X_132 * D(D3M3, i, L)) +
                       0.5 * c2 * c2 * (2. * X_p022 * X_p022 * D(D2M2, i, L) + X_220 * X_220 * D(D00, i, L) + X_242 * X_242 * D(D4M4, i, L));
        if (!accurate) { lensp -= D(D22, i, L); lensm -= D(D2M2, i, L); }
        ksip[i] += fac1 * (cl_ee[L] + cl_bb[L]) * lensp;
        ksim[i] += fac1 * (cl_ee[L] - cl_bb[L]) * lensm;
      }
    }
  }
  // back to harmonic space (lensing_module.cpp:1094-1246)
  for (int il = 0; il < l_size; il++) {
    const int L = l[il];
    double a = 0., b = 0., cp = 0., cm = 0.;
    for (int i = 0; i < num_mu - 1; i++) {
      a += ksi[i] * D(D00, i, L) * w8[i];
      b += ksiX[i] * D(D20, i, L) * w8[i];
      cp += ksip[i] * D(D22, i, L) * w8[i];
      cm += ksim[i] * D(D2M2, i, L) * w8[i];
    }
    if (has_tt) cl_lens[(size_t)il * ct + s->index_ct_tt] = a * 2.0 * PI + (accurate ? 0. : cl_tt[L]);
    if (has_te) cl_lens[(size_t)il * ct + s->index_ct_te] = b * 2.0 * PI + (accurate ? 0. : cl_te[L]);
    if (has_pol) {
      if (has_ee) cl_lens[(size_t)il * ct + s->index_ct_ee] = (cp + cm) * PI + (accurate ? 0. : cl_ee[L]);
      if (has_bb) cl_lens[(size_t)il * ct + s->index_ct_bb] = (cp - cm) * PI + (accurate ? 0. : cl_bb[L]);
    }
  }
  return 0;
}
}
