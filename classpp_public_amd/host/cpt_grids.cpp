// Host-side sampling grids (include/cpt_host.h): exact restatements of the grid builders of the reference's
// PerturbationsModule / TransferModule constructors: scalars or tensors (one mode per handle), flat / open / closed space.
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/cpt_host.h"

namespace {
thread_local std::string g_err;
int fail(const char* fmt, ...) {
  char buf[2048];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return CPT_ERR_INVALID;
}
const double PI = 3.1415926535897932384626433832795e0;
}  // namespace
namespace cpt_host {
// error reporting shared with the other host translation units (cpt_cosmo.cpp)
int fail_msg(int code, const char* fmt, ...) {
  char buf[2048];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}
}  // namespace cpt_host
namespace {

// spline row lookup on the ORIGINAL tables (tools/arrays.c:1565-1628 bisection; the closeby walk of :2173-2225 finds
// the same bracket)
struct HostTables {
  const cpt_tables& t;
  int bracket(const double* x, int n, double v) const {
    int inf = 0, sup = n - 1;
    while (sup - inf > 1) {
      int mid = (int)(0.5 * (inf + sup));
      if (v < x[mid]) sup = mid; else inf = mid;
    }
    return inf;
  }
  double spl(const double* tab, const double* dd, int nc, int inf, int col, double a, double b, double h) const {
    return a * tab[(size_t)inf * nc + col] + b * tab[(size_t)(inf + 1) * nc + col] +
           ((a * a * a - a) * dd[(size_t)inf * nc + col] + (b * b * b - b) * dd[(size_t)(inf + 1) * nc + col]) * h * h / 6.;
  }
  // background_at_tau, short_info: a, H, H'
  void bg(double tau, double* a_, double* H_, double* Hp_) const {
    int inf = bracket(t.tau_table, t.bt_size, tau);
    double h = t.tau_table[inf + 1] - t.tau_table[inf], b = (tau - t.tau_table[inf]) / h, a = 1 - b;
    *a_ = spl(t.background_table, t.d2background_dtau2_table, t.bg_size, inf, t.index_bg_a, a, b, h);
    *H_ = spl(t.background_table, t.d2background_dtau2_table, t.bg_size, inf, t.index_bg_H, a, b, h);
    *Hp_ = spl(t.background_table, t.d2background_dtau2_table, t.bg_size, inf, t.index_bg_H_prime, a, b, h);
  }
  // thermodynamics_at_z: dkappa and rate (th.cpp:114-285; above the table dkappa = rate = (1+z)^2 n_e x_e sigma_T)
  void th(const cpt_config& c, double z, double* dkappa, double* rate) const {
    const int n = t.tt_size, nc = t.th_size;
    if (z >= t.z_table[n - 1]) {
      double x0 = t.thermodynamics_table[(size_t)(n - 1) * nc + t.index_th_xe];
      *dkappa = (1. + z) * (1. + z) * c.n_e * x0 * 6.6524616e-29 * 3.085677581282e22;
      *rate = *dkappa;
      return;
    }
    int inf = bracket(t.z_table, n, z);
    double h = t.z_table[inf + 1] - t.z_table[inf], b = (z - t.z_table[inf]) / h, a = 1 - b;
    *dkappa = spl(t.thermodynamics_table, t.d2thermodynamics_dz2_table, nc, inf, t.index_th_dkappa, a, b, h);
    *rate = spl(t.thermodynamics_table, t.d2thermodynamics_dz2_table, nc, inf, t.index_th_rate, a, b, h);
  }
};
}  // namespace

extern "C" {

const char* cpt_host_error(void) { return g_err.c_str(); }

// pm.cpp:1628-1868 (scalars), :2007-2105 (tensors: the same linear grid up to k_max_cmb, no P(k) extension)
int cpt_host_k_list(const cpt_config* c, const cpt_grid_params* g, double* k_out, int cap, int* k_size, int* k_size_cl,
                    int* k_size_cmb) {
  if (g->k_step_transition == 0.) return fail("stop to avoid division by zero (k_step_transition)");
  if (g->rs_rec == 0.) return fail("stop to avoid division by zero (rs_rec)");
  const bool tens = c->mode == CPT_MODE_TENSORS;
  // first value (pm.cpp:1677-1691, 2011-2025): K > 0 starts from q = sqrt(k^2 + (1+m) K) = 3 sqrt(K), m = 0 / 2 for scalars / tensors
  double k_min;
  if (c->sgnK == 0) k_min = g->k_min_tau0 / c->tau0;
  else if (c->sgnK == -1) k_min = sqrt(-c->K + pow(g->k_min_tau0 / c->tau0 / c->angular_rescaling, 2));
  else k_min = sqrt(((tens ? 6. : 8.) - 1.e-4) * c->K);
  const double k_rec = 2. * PI / g->rs_rec;
  double k_max_cmb = k_min, k_max_cl = k_min, k_max = k_min;
  if (g->has_cls) {
    k_max_cmb = g->k_max_tau0_over_l_max * (tens ? g->l_tensor_max : g->l_scalar_max) / c->tau0 / c->angular_rescaling;
    k_max_cl = k_max_cmb;
    k_max = k_max_cmb;
  }
  if (g->has_pk_matter && !tens) k_max = std::max(k_max, g->k_max_for_pk);
  if (k_max < k_min) return fail("buggy definition of k_min and/or k_max");
  std::vector<double> ks;
  double k = k_min;
  ks.push_back(k);
  while (k < k_max_cmb) {
    double step = (g->k_step_super + 0.5 * (tanh((k - k_rec) / k_rec / g->k_step_transition) + 1.) * (g->k_step_sub - g->k_step_super)) * k_rec;
    double scale2 = pow(c->a_today * c->H0, 2) + fabs(c->K);
    step *= (k * k / scale2 + 1.) / (k * k / scale2 + 1. / g->k_step_super_reduction);
    if (step / k < c->smallest_allowed_variation) return fail("k step =%e < machine precision", step * k_rec);
    k += step;
    if (k <= ks.back()) return fail("consecutive values of k should differ and should be in growing order");
    ks.push_back(k);
  }
  *k_size_cmb = (int)ks.size();
  if (tens) {   // pm.cpp:2096-2098
    *k_size_cl = *k_size = *k_size_cmb;
    if (*k_size > cap) return fail("k array too small: need %d", *k_size);
    std::copy(ks.begin(), ks.end(), k_out);
    return CPT_OK;
  }
  auto logstep = [&](double kk) {
    return kk * pow(10., 1. / (g->k_per_decade_for_pk + (g->k_per_decade_for_bao - g->k_per_decade_for_pk) *
                                                              (1. - tanh(pow((log(kk) - log(g->k_bao_center * k_rec)) / log(g->k_bao_width), 4)))));
  };
  while (k < k_max_cl) { k = logstep(k); ks.push_back(k); }
  *k_size_cl = (int)ks.size();
  while (k < k_max) { k = logstep(k); ks.push_back(k); }
  *k_size = (int)ks.size();
  if (*k_size > cap) return fail("k array too small: need %d", *k_size);
  std::copy(ks.begin(), ks.end(), k_out);
  return CPT_OK;
}

// pm.cpp:1554-1592
int cpt_host_ln_tau_size(const double* tau, int tau_size, double tau_lower, int* ln_tau_size) {
  if (!tau || !ln_tau_size || tau_size < 1) return fail("bad arguments to cpt_host_ln_tau_size");
  if (!(tau_lower > 0.)) { *ln_tau_size = 1; return CPT_OK; }
  if (tau_lower <= tau[0])
    return fail("you asked for z_max_pk with taumin=%e, smaller than or equal to the first possible value =%e; it should be strictly bigger for a successfull "
                "interpolation", tau_lower, tau[0]);
  int index_tau = 0;
  while (index_tau < tau_size && tau[index_tau] < tau_lower) index_tau++;
  index_tau--;
  for (int extra = 0; extra < 4; extra++) if (index_tau > 0) index_tau--;   // a few more values against boundary effects of the interpolation
  *ln_tau_size = tau_size - index_tau;
  return CPT_OK;
}

// pm.cpp:1247-1533 (has_cmb branch)
int cpt_host_tau_sampling(const cpt_config* c, const cpt_tables* t, const cpt_grid_params* g, double* tau_out, int cap,
                          int* tau_size) {
  HostTables T{*t};
  double a, H, Hp, dk, rate;
  double tau_lower = g->tau_ini_thermo;
  T.bg(tau_lower, &a, &H, &Hp);
  T.th(*c, 1. / a - 1., &dk, &rate);
  if (a * H / dk > g->start_sources_at_tau_c_over_tau_h)
    return fail("your choice of initial time for computing sources is inappropriate: it corresponds to an earlier time than "
                "the one at which the integration of thermodynamical variables started (tau=%g)", tau_lower);
  double tau_upper = c->tau_rec;
  T.bg(tau_upper, &a, &H, &Hp);
  T.th(*c, 1. / a - 1., &dk, &rate);
  if (a * H / dk < g->start_sources_at_tau_c_over_tau_h)
    return fail("your choice of initial time for computing sources is inappropriate: it corresponds to a time after recombination");
  double tau_mid = 0.5 * (tau_lower + tau_upper);
  while (tau_upper - tau_lower > c->tol_tau_approx) {
    T.bg(tau_mid, &a, &H, &Hp);
    T.th(*c, 1. / a - 1., &dk, &rate);
    if (a * H / dk > g->start_sources_at_tau_c_over_tau_h) tau_upper = tau_mid; else tau_lower = tau_mid;
    tau_mid = 0.5 * (tau_lower + tau_upper);
  }
  const double tau_ini = tau_mid;
  std::vector<double> ts;
  ts.push_back(tau_ini);
  double tau = tau_ini;
  while (tau < c->tau0) {
    T.bg(tau, &a, &H, &Hp);
    T.th(*c, 1. / a - 1., &dk, &rate);
    double aH = H * a;
    double app = Hp * a + 2. * aH * aH;
    double rate_isw_squared = fabs(2. * app - aH * aH);
    double timescale = sqrt(rate * rate + rate_isw_squared);
    if (timescale == 0.) return fail("null evolution rate, integration is diverging");
    timescale = 1. / timescale;
    if (fabs(g->perturb_sampling_stepsize * timescale / tau) < c->smallest_allowed_variation)
      return fail("integration step =%e < machine precision", g->perturb_sampling_stepsize * timescale);
    tau = tau + g->perturb_sampling_stepsize * timescale;
    ts.push_back(tau);
  }
  ts.back() = c->tau0;
  *tau_size = (int)ts.size();
  if (*tau_size > cap) return fail("tau array too small: need %d", *tau_size);
  std::copy(ts.begin(), ts.end(), tau_out);
  return CPT_OK;
}

// tm.cpp:694-790
int cpt_host_l_list(const cpt_config* c, const cpt_grid_params* g, int* l_out, int cap, int* l_size) {
  const int l_max = (c->mode == CPT_MODE_TENSORS) ? g->l_tensor_max : g->l_scalar_max;   // one mode per handle (tm.cpp:712-736)
  const double ar = c->angular_rescaling;
  std::vector<int> l;
  l.push_back(2);
  int increment = std::max((int)(l.back() * (pow(g->l_logstep, ar) - 1.)), 1);
  while (((l.back() + increment) < l_max) && (increment < g->l_linstep * ar)) {
    l.push_back(l.back() + increment);
    increment = std::max((int)(l.back() * (pow(g->l_logstep, ar) - 1.)), 1);
  }
  increment = (int)(g->l_linstep * ar);
  while ((l.back() + increment) <= l_max) l.push_back(l.back() + increment);
  if (l.back() != l_max) l.push_back(l_max);
  *l_size = (int)l.size();
  if (*l_size > cap) return fail("l array too small: need %d", *l_size);
  std::copy(l.begin(), l.end(), l_out);
  return CPT_OK;
}

// tm.cpp:884-1096: flat / open (one formula), closed (integer nu = q / sqrt(K) below hyper_flat_approximation_nu, then a
// gradual return to the flat step); k_min / k_max_cl: first and last k of the mode's C_l range
int cpt_host_q_list(const cpt_config* c, const cpt_grid_params* g, double k_min, double k_max_cl, double* q_out, int cap,
                    int* q_size) {
  const double q_period = 2. * PI / (c->tau0 - c->tau_rec) * c->angular_rescaling;  // tm.cpp:189
  const double K = c->K, m1 = (c->mode == CPT_MODE_TENSORS) ? 3. : 1.;   // 1 + m
  double q_min, q_max;
  if (c->sgnK == 0) { q_min = k_min; q_max = k_max_cl; }
  else if (c->sgnK == -1) { q_min = sqrt(k_min * k_min + K); q_max = std::min(sqrt(k_max_cl * k_max_cl + K), sqrt(k_max_cl * k_max_cl + m1 * K)); }
  else { q_min = 3. * sqrt(K); q_max = k_max_cl; }
  const double q_logstep_spline = g->q_logstep_spline / pow(c->angular_rescaling, g->q_logstep_open);
  const double q_logstep_trapzd = g->q_logstep_trapzd;
  std::vector<double> q;
  q.push_back(q_min);
  int nu = 3, last_index = 0;
  double last_step = 0.;
  while (q.back() < q_max) {
    const double last = q.back();
    const int index_q = (int)q.size();
    double qn;
    if (c->sgnK <= 0) qn = last + q_period * g->q_linstep * last / (last + g->q_linstep / q_logstep_spline);
    else if (nu < (int)c->hyper_flat_approximation_nu) {
      qn = last + q_period * g->q_linstep * last / (last + g->q_linstep / q_logstep_trapzd);
      const int nu_proposed = (int)(qn / sqrt(K));
      nu = (nu_proposed <= nu + 1) ? nu + 1 : nu_proposed;
      qn = nu * sqrt(K);
      last_step = qn - last;
      last_index = index_q + 1;
    } else {
      const double q_step = q_period * g->q_linstep * last / (last + g->q_linstep / q_logstep_spline);
      if (index_q - last_index < (int)g->q_numstep_transition)
        qn = last + (1 - (double)(index_q - last_index) / g->q_numstep_transition) * last_step + (double)(index_q - last_index) / g->q_numstep_transition * q_step;
      else qn = last + q_step;
    }
    q.push_back(qn);
    if (q.size() > 10000000) return fail("buggy q-list definition");
  }
  if (q.back() > q_max) q.pop_back();
  if (q.size() < 2) return fail("buggy q-list definition");
  *q_size = (int)q.size();
  if (*q_size > cap) return fail("q array too small: need %d", *q_size);
  std::copy(q.begin(), q.end(), q_out);
  return CPT_OK;
}
}
