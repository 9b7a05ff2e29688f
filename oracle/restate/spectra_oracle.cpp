// ORACLE / TEST INFRASTRUCTURE ONLY -- CPU restatement of the C_l assembly (SpectraModule::spectra_compute_cl,
// source/spectra_module.cpp:958-1353, flat, scalars, one ic), of cl_output's spline in l (:146-218, 926-934) and of
// the linear P(k) (source/nonlinear_module.cpp:1886-2040), with the analytic primordial spectrum
// (source/primordial_module.cpp:911-925).  Pinned by tests/test_oracle_spectra.py against the reference's own
// cl_ table / cl_output / P(k).
#include <cmath>
#include <vector>

#include "../../include/cpt.h"

namespace {
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
double primordial(const cpt_spectra_params& s, double k) {
  double lk = std::log(k / s.k_pivot);
  return s.A_s * std::exp((s.n_s - 1.) * lk + 0.5 * s.alpha_s * lk * lk);
}
}  // namespace

extern "C" {
// transfer [tt][nl][nq] -> cl [nl][ct]
int orc_cl(const cpt_config* c, const cpt_spectra_params* s, const double* tr, const double* q, int nq, int nl, double* cl) {
  const double PI = 3.1415926535897932384626433832795e0;
  std::vector<double> y(nq), dd(nq), u(nq), kk(nq);
  const size_t st = (size_t)nl * nq;
  // integration variable: k(q) = sqrt(q^2 - K(1+m)) (tm.cpp:1106-1167; spectra_module.cpp:990-994); flat: k = q
  for (int iq = 0; iq < nq; iq++) kk[iq] = (c->K == 0.) ? q[iq] : std::sqrt(q[iq] * q[iq] - c->K * (c->mode == CPT_MODE_TENSORS ? 3. : 1.));
  // closed space: trapezoidal rule below the flat-approximation index (integer nu => uneven dq), spectra_module.cpp:1293-1323
  int index_q_spline = 0;
  if (c->sgnK == 1) {
    const double q_approximation = c->hyper_flat_approximation_nu * std::sqrt(c->K);
    for (index_q_spline = 0; index_q_spline < nq - 1; index_q_spline++)
      if (q[index_q_spline] > q_approximation) break;
  }
  for (int il = 0; il < nl; il++) {
    for (int ct = 0; ct < s->ct_size; ct++) {
      int kind = -1;
      if (ct == s->index_ct_tt) kind = 0; else if (ct == s->index_ct_ee) kind = 1; else if (ct == s->index_ct_te) kind = 2;
      else if (ct == s->index_ct_pp) kind = 4; else if (ct == s->index_ct_tp) kind = 5; else if (ct == s->index_ct_ep) kind = 6;
      const bool tens = c->mode == CPT_MODE_TENSORS;   // tensors: TT, EE, TE and BB only (spectra_module.cpp:1027-1185)
      if (tens) { if (kind >= 4) kind = -1; if (ct == s->index_ct_bb) kind = 3; }
      if (kind < 0) { cl[(size_t)il * s->ct_size + ct] = 0.; continue; }
      for (int iq = 0; iq < nq; iq++) {
        double k = kk[iq], temp = 0., e = 0., lc = 0., bm = 0.;
        size_t o = (size_t)il * nq + iq;
        if (tens) { if (c->index_tt_t2 >= 0) temp = tr[c->index_tt_t2 * st + o]; if (c->index_tt_b >= 0) bm = tr[c->index_tt_b * st + o]; }
        else if (c->index_tt_t0 >= 0) temp = tr[c->index_tt_t0 * st + o] + tr[c->index_tt_t1 * st + o] + tr[c->index_tt_t2 * st + o];
        if (c->index_tt_e >= 0) e = tr[c->index_tt_e * st + o];
        if (!tens && c->index_tt_lcmb >= 0) lc = tr[c->index_tt_lcmb * st + o];
        double prod = kind == 3 ? bm * bm : kind == 0 ? temp * temp : kind == 1 ? e * e : kind == 2 ? 0.5 * (temp * e + e * temp)
                    : kind == 4 ? lc * lc : kind == 5 ? 0.5 * (temp * lc + lc * temp) : 0.5 * (e * lc + lc * e);
        y[iq] = primordial(*s, k) * prod * (4. * PI / k);
      }
      spline_est_deriv(kk.data(), nq, y.data(), dd.data(), u.data());
      double sum = 0.;
      for (int i = 0; i < index_q_spline; i++) sum += (y[i] + y[i + 1]) * (kk[i + 1] - kk[i]) / 2.;   // arrays.c:1402-1409
      for (int i = index_q_spline; i < nq - 1; i++) {  // arrays.c:1413-1421
        double h = kk[i + 1] - kk[i];
        sum += (y[i] + y[i + 1]) * h / 2. + (dd[i] + dd[i + 1]) * h * h * h / 24.;
      }
      if (c->sgnK == 1) sum += y[0] * q[0] / kk[0] * std::sqrt(c->K) / 2.;   // discrete sum: weight of the first point, :1319-1321
      cl[(size_t)il * s->ct_size + ct] = sum;
    }
  }
  return 0;
}

// cl table [nl][ct] on the l grid -> every integer l in [2, lmax]: out [ct][lmax+1] (spectra_module.cpp:146-218)
int orc_cl_at_integer_l(const int* l, int nl, int ct_size, const double* cl, int lmax, double* out) {
  std::vector<double> x(nl), y(nl), dd(nl), u(nl);
  for (int i = 0; i < nl; i++) x[i] = l[i];
  for (int ct = 0; ct < ct_size; ct++) {
    for (int i = 0; i < nl; i++) y[i] = cl[(size_t)i * ct_size + ct];
    spline_est_deriv(x.data(), nl, y.data(), dd.data(), u.data());
    out[(size_t)ct * (lmax + 1) + 0] = out[(size_t)ct * (lmax + 1) + 1] = 0.;
    int inf = 0;
    for (int L = 2; L <= lmax; L++) {
      while (inf < nl - 2 && x[inf + 1] < L) inf++;
      double h = x[inf + 1] - x[inf], b = (L - x[inf]) / h, a = 1 - b;
      out[(size_t)ct * (lmax + 1) + L] = a * y[inf] + b * y[inf + 1] + ((a * a * a - a) * dd[inf] + (b * b * b - b) * dd[inf + 1]) * h * h / 6.;
    }
  }
  return 0;
}

// sigma(R) = sqrt( 1/(2 pi^2) int dk k^2 P(k) W^2(kR) ): NonlinearModule::nonlinear_sigmas_at_z + nonlinear_sigmas
// (source/nonlinear_module.cpp:926-963, 2041-2180): ln P splined in ln k (estimated end derivatives), integrand sampled
// at k_per_decade points per decade, integrated over t = 1/(1+k) with the spline rule.
static void sigma_spline(const double* x, const double* y, int n, double* dd) {   // tools/arrays.c:967-1092 / 261-353, EST_DERIV, one column
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
static double sigma_func(const double* kk, const double* pk, int nk, double R, double k_per_decade) {
  const double PI = 3.1415926535897932384626433832795e0;
  std::vector<double> lnk(nk), lnpk(nk), dd(nk);
  for (int i = 0; i < nk; i++) { lnk[i] = log(kk[i]); lnpk[i] = log(pk[i]); }
  sigma_spline(lnk.data(), lnpk.data(), nk, dd.data());
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
  sigma_spline(xs.data(), ys.data(), n, d2.data());
  double res = 0.;
  for (int i = 0; i < n - 1; i++) {
    const double h = xs[i + 1] - xs[i];
    res += (ys[i] + ys[i + 1]) * h / 2. + (d2[i] + d2[i + 1]) * h * h * h / 24.;
  }
  return sqrt(res / (2. * PI * PI));
}
int orc_sigma(const double* k, const double* pk, int nk, double R, double k_per_decade, double* sigma) {
  *sigma = sigma_func(k, pk, nk, R, k_per_decade);
  return 0;
}

int orc_pk(const cpt_spectra_params* s, const double* k, int nk, const double* delta_m_today, double* pk) {
  const double PI = 3.1415926535897932384626433832795e0;
  for (int i = 0; i < nk; i++) pk[i] = 2. * PI * PI / (k[i] * k[i] * k[i]) * delta_m_today[i] * delta_m_today[i] * primordial(*s, k[i]);
  return 0;
}
}
