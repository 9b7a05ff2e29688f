// Unit checks of the host library's numerical building blocks (classpp_public_amd/host/cpt_numerics.hpp), compiled and run by
// tests/test_host_numerics.py.  Exit code 0 = every check passed; a message names the first failure.
//   1. spline_bracket / spline_eval with a hint: the same interval and the same bits as without, for ascending and descending
//      tables, at nodes, at both ends, for good, stale and out-of-range hints.
//   2. ClampedSpline::moments (row-by-row sweeps over all columns) against a plain column-by-column Thomas solve of the same system.
//   3. Dopri5: slope() is f at the end of the last step; forget_step_size() makes the next advance try its whole interval
//      (one step per node on a smooth problem), restart() discards the kept stage.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../classpp_public_amd/host/cpt_numerics.hpp"

static int fail(const char* what) { std::fprintf(stderr, "FAILED: %s\n", what); return 1; }

int main() {
  std::mt19937_64 rng(12345);
  std::uniform_real_distribution<double> U(0., 1.);
  // ---- 1. hinted look-ups
  for (int dir = 0; dir < 2; dir++) {
    for (int n : {3, 4, 17, 200}) {
      std::vector<double> x(n), y((size_t)n * 3), m((size_t)n * 3);
      double at = dir ? 50. : -3.;
      for (int i = 0; i < n; i++) { at += (dir ? -1. : 1.) * (0.01 + U(rng)); x[i] = at; }
      for (auto& v : y) v = U(rng) - 0.5;
      cpt_num::ClampedSpline(x.data(), n).moments(y.data(), 3, 3, m.data());
      std::vector<double> probes;
      for (int i = 0; i < n; i++) probes.push_back(x[i]);                                   // every node, both ends included
      for (int i = 0; i + 1 < n; i++) probes.push_back(x[i] + (x[i + 1] - x[i]) * U(rng));
      for (int i = 0; i + 1 < n; i++) probes.push_back(std::nextafter(x[i + 1], x[i]));        // a hair before a node
      for (double v : probes) {
        double ref[3], got[3];
        if (!cpt_num::spline_eval(x.data(), n, y.data(), m.data(), 3, 3, v, ref)) return fail("a probe inside the table was refused");
        const bool up = x[0] < x[n - 1];
        const int lo_ref = cpt_num::spline_bracket(x.data(), n, v, up, nullptr);
        for (int hint : {-5, 0, lo_ref - 2, lo_ref - 1, lo_ref, lo_ref + 1, lo_ref + 2, n - 2, n + 7}) {
          int h = hint;
          if (cpt_num::spline_bracket(x.data(), n, v, up, &h) != lo_ref) return fail("the interval depends on the hint");
          if (!cpt_num::spline_eval(x.data(), n, y.data(), m.data(), 3, 3, v, got, &h)) return fail("hinted evaluation refused");
          if (h != lo_ref) return fail("the hint was not updated to the interval found");
          if (std::memcmp(ref, got, sizeof ref)) return fail("hinted evaluation differs from the plain one");
        }
      }
      double out[3];
      const double lo = std::min(x[0], x[n - 1]), hi = std::max(x[0], x[n - 1]);
      if (cpt_num::spline_eval(x.data(), n, y.data(), m.data(), 3, 3, lo - 1e-9, out) || cpt_num::spline_eval(x.data(), n, y.data(), m.data(), 3, 3, hi + 1e-9, out))
        return fail("a point outside the table was accepted");
    }
  }
  // ---- 2. spline moments: all columns at once against one column at a time
  {
    const int n = 300, nc = 7;
    std::vector<double> x(n), y((size_t)n * nc), m((size_t)n * nc);
    double at = 0.;
    for (int i = 0; i < n; i++) { at += 0.05 + U(rng); x[i] = at; }
    for (int i = 0; i < n; i++) for (int c = 0; c < nc; c++) y[(size_t)i * nc + c] = std::sin(0.3 * x[i] * (c + 1)) + 0.1 * U(rng);
    cpt_num::ClampedSpline(x.data(), n).moments(y.data(), nc, nc, m.data());
    for (int c = 0; c < nc; c++) {
      // dense Thomas solve of  h_{i-1} M_{i-1} + 2 (h_{i-1} + h_i) M_i + h_i M_{i+1} = 6 (d_i - d_{i-1})  with the clamped end rows
      std::vector<double> sub(n), dia(n), sup(n), rhs(n), h(n - 1), d(n - 1);
      auto Y = [&](int i) { return y[(size_t)i * nc + c]; };
      for (int i = 0; i + 1 < n; i++) { h[i] = x[i + 1] - x[i]; d[i] = (Y(i + 1) - Y(i)) / h[i]; }
      const double s0 = cpt_num::ClampedSpline::end_slope(x[0], x[1], x[2], Y(0), Y(1), Y(2));
      const double s1 = cpt_num::ClampedSpline::end_slope(x[n - 1], x[n - 2], x[n - 3], Y(n - 1), Y(n - 2), Y(n - 3));
      dia[0] = 2. * h[0]; sup[0] = h[0]; rhs[0] = 6. * (d[0] - s0);
      for (int i = 1; i + 1 < n; i++) { sub[i] = h[i - 1]; dia[i] = 2. * (h[i - 1] + h[i]); sup[i] = h[i]; rhs[i] = 6. * (d[i] - d[i - 1]); }
      sub[n - 1] = h[n - 2]; dia[n - 1] = 2. * h[n - 2]; rhs[n - 1] = 6. * (s1 - d[n - 2]);
      for (int i = 1; i < n; i++) { const double w = sub[i] / dia[i - 1]; dia[i] -= w * sup[i - 1]; rhs[i] -= w * rhs[i - 1]; }
      std::vector<double> M(n);
      M[n - 1] = rhs[n - 1] / dia[n - 1];
      for (int i = n - 2; i >= 0; i--) M[i] = (rhs[i] - sup[i] * M[i + 1]) / dia[i];
      double scale = 0.;
      for (int i = 0; i < n; i++) scale = std::max(scale, std::fabs(M[i]));
      for (int i = 0; i < n; i++)
        if (std::fabs(M[i] - m[(size_t)i * nc + c]) > 1e-12 * scale) return fail("spline moments differ from the column-by-column solve");
    }
  }
  // ---- 3. the integrator between table nodes
  {
    long calls = 0;
    auto f = [&](double t, const double* y, double* dy) { calls++; dy[0] = -y[0] + std::sin(t); dy[1] = y[0]; };
    cpt_num::Dopri5<2> ode;
    ode.rtol = 1e-9;
    ode.x = 0.; ode.y[0] = 1.; ode.y[1] = 0.;
    if (ode.slope()) return fail("a stage before the first step");
    const int nodes = 200;
    for (int i = 1; i <= nodes; i++) {
      ode.forget_step_size();
      if (!ode.advance(f, 0.01 * i)) return fail("advance failed");
      const double* s = ode.slope();
      double dy[2];
      const long before = calls;
      f(ode.x, ode.y, dy);
      calls = before;
      if (!s || std::fabs(s[0] - dy[0]) > 1e-13 || std::fabs(s[1] - dy[1]) > 1e-13) return fail("slope() is not f at the end of the step");
    }
    if (ode.steps != nodes || ode.rejected != 0) return fail("more than one step per node on a smooth problem");
    if (calls != 1 + 6L * nodes) return fail("the last stage of a step was not handed on as the first of the next");
    const double exact = 1.5 * std::exp(-2.) + 0.5 * (std::sin(2.) - std::cos(2.));   // y' = -y + sin t, y(0) = 1
    if (std::fabs(ode.y[0] - exact) > 1e-9) return fail("integration error above the tolerance");
    ode.restart();
    if (ode.slope()) return fail("restart() kept the stage");
  }
  std::puts("numerics ok");
  return 0;
}
