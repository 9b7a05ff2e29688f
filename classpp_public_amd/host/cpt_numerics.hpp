// Numerical building blocks of the host library (libcpt_host.so): this project's own designs, not restatements of the
// reference's tools/ directory.
//   * Tridiag / ClampedSpline : cubic splines in moment form.  The tridiagonal system of the second derivatives is factorised ONCE
//     per abscissa grid (Thomas algorithm) and then back-substituted for any number of tabulated columns; the end conditions are
//     prescribed first derivatives, estimated from the parabola through the three outermost nodes (the rule the reference's tables
//     are built with, so that both sides interpolate the same function between nodes).
//   * Dopri5<N> : embedded Runge-Kutta 5(4) pair of Dormand & Prince (J. Comp. Appl. Math. 6, 1980) with first-same-as-last
//     stage reuse and a PI step-size controller; the caller walks it from output node to output node.  Used at tolerances far
//     below what the tables need (1e-10 / 1e-9), so the integration error is not a parameter of the result.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <vector>

namespace cpt_num {

// ---------------------------------------------------------------------------------------------------------------------
// Cubic spline through (x_i, y_i), i = 0..n-1, with prescribed end slopes, in moment form: M_i = s''(x_i) solve
//   h_{i-1} M_{i-1} + 2 (h_{i-1} + h_i) M_i + h_i M_{i+1} = 6 (d_i - d_{i-1}),   d_i = (y_{i+1} - y_i) / h_i,    0 < i < n-1
//   2 h_0 M_0 + h_0 M_1 = 6 (d_0 - s'_0),        h_{n-2} M_{n-2} + 2 h_{n-2} M_{n-1} = 6 (s'_{n-1} - d_{n-2}).
// ---------------------------------------------------------------------------------------------------------------------
class ClampedSpline {
 public:
  // x strictly monotonic (either direction), n >= 3
  ClampedSpline(const double* x, int n) : n_(n), h_(n - 1), low_(n), piv_(n), x_(x) {
    for (int i = 0; i < n - 1; i++) h_[i] = x[i + 1] - x[i];
    // LU of the tridiagonal matrix (diagonal dominant: no pivoting): low_[i] = sub_i / piv_{i-1}
    piv_[0] = 2. * h_[0];
    low_[0] = 0.;
    for (int i = 1; i < n; i++) {
      const double sub = h_[i - 1], diag = (i < n - 1) ? 2. * (h_[i - 1] + h_[i]) : 2. * h_[n - 2], sup_prev = h_[i - 1];
      low_[i] = sub / piv_[i - 1];
      piv_[i] = diag - low_[i] * sup_prev;
    }
  }
  // slope at an end of the table of the parabola through the three outermost nodes (Newton form: f[a,b] + f[a,b,c] (a - b))
  static double end_slope(double xa, double xb, double xc, double ya, double yb, double yc) {
    const double fab = (yb - ya) / (xb - xa), fbc = (yc - yb) / (xc - xb);
    return fab + (fbc - fab) / (xc - xa) * (xa - xb);
  }
  // second derivatives of `ncol` columns tabulated row-major: y[i * stride + c] -> m[i * stride + c].  The two sweeps walk the
  // table row by row with the columns innermost: memory is read in order and the columns' recurrences (each a chain of dependent
  // multiply-adds and one division per node) overlap in the pipeline
  void moments(const double* y, int ncol, int stride, double* m) const {
    const int n = n_;
    std::vector<double> work(2 * (size_t)ncol);
    double* d_prev = work.data();
    double* s_last = d_prev + ncol;
    auto Y = [&](int i) { return y + (size_t)i * stride; };
    auto M = [&](int i) { return m + (size_t)i * stride; };
    for (int c = 0; c < ncol; c++) {
      const double s0 = end_slope(x_[0], x_[1], x_[2], Y(0)[c], Y(1)[c], Y(2)[c]);
      s_last[c] = end_slope(x_[n - 1], x_[n - 2], x_[n - 3], Y(n - 1)[c], Y(n - 2)[c], Y(n - 3)[c]);
      d_prev[c] = (Y(1)[c] - Y(0)[c]) / h_[0];
      M(0)[c] = 6. * (d_prev[c] - s0);
    }
    // forward elimination on the right-hand side
    for (int i = 1; i < n - 1; i++) {
      const double *y0 = Y(i), *y1 = Y(i + 1), *mp = M(i - 1);
      double* mi = M(i);
      const double h = h_[i], low = low_[i];
      for (int c = 0; c < ncol; c++) {
        const double d = (y1[c] - y0[c]) / h;
        mi[c] = 6. * (d - d_prev[c]) - low * mp[c];
        d_prev[c] = d;
      }
    }
    for (int c = 0; c < ncol; c++) M(n - 1)[c] = (6. * (s_last[c] - d_prev[c]) - low_[n - 1] * M(n - 2)[c]) / piv_[n - 1];
    // back substitution (super-diagonal entry of row i is h_i)
    for (int i = n - 2; i >= 0; i--) {
      const double* mn = M(i + 1);
      double* mi = M(i);
      const double h = h_[i], piv = piv_[i];
      for (int c = 0; c < ncol; c++) mi[c] = (mi[c] - h * mn[c]) / piv;
    }
  }

 private:
  int n_;
  std::vector<double> h_, low_, piv_;
  const double* x_;
};

// the interval [lo, lo + 1] of a monotonic (either direction) table that holds v: x[lo] <= v < x[lo + 1] going up, x[lo] > v >= x[lo + 1]
// going down, the first / last interval at the ends.  `hint` (an interval found earlier, or null) is tried first together with its
// two neighbours, which is what a caller walking along the table needs; the answer does not depend on the hint.
inline int spline_bracket(const double* x, int n, double v, bool up, const int* hint) {
  auto holds = [&](int i) {
    if (i < 0 || i > n - 2) return false;
    if (up) return (x[i] <= v || i == 0) && (v < x[i + 1] || (i == n - 2 && v == x[n - 1]));
    return (x[i] > v || i == 0) && v >= x[i + 1];
  };
  if (hint) {
    const int h = *hint;
    if (holds(h)) return h;
    if (holds(h + 1)) return h + 1;
    if (holds(h - 1)) return h - 1;
  }
  int lo = 0, hi = n - 1;   // bisection
  while (hi - lo > 1) {
    const int mid = (lo + hi) / 2;
    if ((v < x[mid]) == up) hi = mid; else lo = mid;
  }
  return lo;
}
// value of the spline (y, m) at v, x monotonic in either direction; returns false outside the table.  ncol columns starting at
// y / m are evaluated (rows `stride` apart); *hint, when given, is read as the caller's guess of the interval and updated
inline bool spline_eval(const double* x, int n, const double* y, const double* m, int ncol, int stride, double v, double* out, int* hint = nullptr) {
  const bool up = x[0] < x[n - 1];
  if (up ? (v < x[0] || v > x[n - 1]) : (v > x[0] || v < x[n - 1])) return false;
  const int lo = spline_bracket(x, n, v, up, hint), hi = lo + 1;
  if (hint) *hint = lo;
  const double h = x[hi] - x[lo], t = (v - x[lo]) / h, u = 1. - t;
  const double cu = (u * u * u - u) * h * h / 6., ct = (t * t * t - t) * h * h / 6.;
  for (int c = 0; c < ncol; c++)
    out[c] = u * y[(size_t)lo * stride + c] + t * y[(size_t)hi * stride + c] + cu * m[(size_t)lo * stride + c] + ct * m[(size_t)hi * stride + c];
  return true;
}
// running integral from x_0: I_0 = 0, I_{i+1} = I_i + h (y_i + y_{i+1}) / 2 + curvature_sign h^3 (m_i + m_{i+1}) / 24.
// The exact integral of the spline has curvature_sign = -1.  The reference's thermodynamics tables (optical depth, drag depth,
// reionization depth) are built with +1 (tools/arrays.c:255-256); those tables are the contract of this library, so its callers
// pass +1 and say so - a 1e-5 effect on the visibility function.
inline void spline_cumulative_integral(const double* x, int n, const double* y, const double* m, int stride, double* integral, double curvature_sign) {
  integral[0] = 0.;
  for (int i = 0; i + 1 < n; i++) {
    const double h = x[i + 1] - x[i];
    integral[(size_t)(i + 1) * stride] = integral[(size_t)i * stride] + 0.5 * h * (y[(size_t)i * stride] + y[(size_t)(i + 1) * stride]) +
                                         curvature_sign * h * h * h / 24. * (m[(size_t)i * stride] + m[(size_t)(i + 1) * stride]);
  }
}
// first derivative of the spline at its nodes (right-sided formula on every interval, left-sided at the last node)
inline void spline_node_derivative(const double* x, int n, const double* y, const double* m, int stride, double* dy) {
  for (int i = 0; i + 1 < n; i++) {
    const double h = x[i + 1] - x[i];
    dy[(size_t)i * stride] = (y[(size_t)(i + 1) * stride] - y[(size_t)i * stride]) / h - h / 6. * (2. * m[(size_t)i * stride] + m[(size_t)(i + 1) * stride]);
  }
  const double h = x[n - 1] - x[n - 2];
  dy[(size_t)(n - 1) * stride] = (y[(size_t)(n - 1) * stride] - y[(size_t)(n - 2) * stride]) / h + h / 6. * (m[(size_t)(n - 2) * stride] + 2. * m[(size_t)(n - 1) * stride]);
}

// ---------------------------------------------------------------------------------------------------------------------
// Dormand-Prince 5(4), N equations.  advance(f, x_to): integrates from the current x to x_to (either direction) with adaptive
// steps; error norm = max_i |err_i| / (atol_i + rtol max(|y_i|, |y_i^new|)).  Returns false on step-size underflow.
// ---------------------------------------------------------------------------------------------------------------------
template <int N>
class Dopri5 {
 public:
  double x = 0., y[N];
  double rtol = 1e-10, atol[N];
  long steps = 0, rejected = 0;
  Dopri5() { for (int i = 0; i < N; i++) { y[i] = 0.; atol[i] = 0.; } }

  template <class F>
  bool advance(F&& f, double x_to) {
    if (x_to == x) return true;
    const double dir = (x_to > x) ? 1. : -1.;
    if (!have_k1_) { f(x, y, k1_); have_k1_ = true; }
    double h = (h_next_ != 0. && h_next_ * dir > 0.) ? h_next_ : (x_to - x);
    for (int guard = 0; guard < 10000000; guard++) {
      bool last = false;
      if ((x + h - x_to) * dir >= 0.) { h = x_to - x; last = true; }
      if (x + h == x) return false;   // step-size underflow
      double ynew[N], err = 0.;
      stages(f, h, ynew);
      for (int i = 0; i < N; i++) {
        const double e = h * (e1 * k1_[i] + e3 * k3_[i] + e4 * k4_[i] + e5 * k5_[i] + e6 * k6_[i] + e7 * k7_[i]);
        const double sc = atol[i] + rtol * std::max(std::fabs(y[i]), std::fabs(ynew[i]));
        err = std::max(err, sc > 0. ? std::fabs(e) / sc : (e != 0. ? 1e300 : 0.));
      }
      if (err <= 1.) {
        steps++;
        x = last ? x_to : x + h;
        for (int i = 0; i < N; i++) { y[i] = ynew[i]; k1_[i] = k7_[i]; }   // first same as last
        // PI controller (Gustafsson): exponents 0.7/5 and 0.4/5, growth limited to 5, never after a rejection
        const double fac = 0.9 * std::pow(std::max(err, 1e-10), -0.14) * std::pow(std::max(err_prev_, 1e-10), 0.08);
        const double hn = h * std::min(5., std::max(0.2, just_rejected_ ? std::min(1., fac) : fac));
        err_prev_ = std::max(err, 1e-4);
        just_rejected_ = false;
        if (last) { h_next_ = last_full_h_ != 0. ? std::max(std::fabs(hn), std::fabs(last_full_h_)) * dir : hn; return true; }
        last_full_h_ = h;
        h = hn;
      } else {
        rejected++;
        just_rejected_ = true;
        h *= std::max(0.1, 0.9 * std::pow(err, -0.2));
      }
    }
    return false;
  }
  // f(x, y) as left by the last accepted step (its last stage), or null when the caller has to evaluate it (before the first
  // step, after restart())
  const double* slope() const { return have_k1_ ? k1_ : nullptr; }
  void forget_step_size() { h_next_ = 0.; last_full_h_ = 0.; err_prev_ = 1e-4; just_rejected_ = false; }   // the next advance() tries its whole interval first
  void restart() { have_k1_ = false; forget_step_size(); }   // after y was changed by the caller

 private:
  // Butcher tableau
  static constexpr double c2 = 1. / 5, c3 = 3. / 10, c4 = 4. / 5, c5 = 8. / 9;
  static constexpr double a21 = 1. / 5, a31 = 3. / 40, a32 = 9. / 40, a41 = 44. / 45, a42 = -56. / 15, a43 = 32. / 9;
  static constexpr double a51 = 19372. / 6561, a52 = -25360. / 2187, a53 = 64448. / 6561, a54 = -212. / 729;
  static constexpr double a61 = 9017. / 3168, a62 = -355. / 33, a63 = 46732. / 5247, a64 = 49. / 176, a65 = -5103. / 18656;
  static constexpr double b1 = 35. / 384, b3 = 500. / 1113, b4 = 125. / 192, b5 = -2187. / 6784, b6 = 11. / 84;
  // b - b^ (fifth minus embedded fourth order weights)
  static constexpr double e1 = 71. / 57600, e3 = -71. / 16695, e4 = 71. / 1920, e5 = -17253. / 339200, e6 = 22. / 525, e7 = -1. / 40;
  double k1_[N], k2_[N], k3_[N], k4_[N], k5_[N], k6_[N], k7_[N];
  bool have_k1_ = false, just_rejected_ = false;
  double h_next_ = 0., last_full_h_ = 0., err_prev_ = 1e-4;

  template <class F>
  void stages(F&& f, double h, double* ynew) {
    double w[N];
    for (int i = 0; i < N; i++) w[i] = y[i] + h * a21 * k1_[i];
    f(x + c2 * h, w, k2_);
    for (int i = 0; i < N; i++) w[i] = y[i] + h * (a31 * k1_[i] + a32 * k2_[i]);
    f(x + c3 * h, w, k3_);
    for (int i = 0; i < N; i++) w[i] = y[i] + h * (a41 * k1_[i] + a42 * k2_[i] + a43 * k3_[i]);
    f(x + c4 * h, w, k4_);
    for (int i = 0; i < N; i++) w[i] = y[i] + h * (a51 * k1_[i] + a52 * k2_[i] + a53 * k3_[i] + a54 * k4_[i]);
    f(x + c5 * h, w, k5_);
    for (int i = 0; i < N; i++) w[i] = y[i] + h * (a61 * k1_[i] + a62 * k2_[i] + a63 * k3_[i] + a64 * k4_[i] + a65 * k5_[i]);
    f(x + h, w, k6_);
    for (int i = 0; i < N; i++) ynew[i] = y[i] + h * (b1 * k1_[i] + b3 * k3_[i] + b4 * k4_[i] + b5 * k5_[i] + b6 * k6_[i]);
    f(x + h, ynew, k7_);
  }
};

}  // namespace cpt_num
