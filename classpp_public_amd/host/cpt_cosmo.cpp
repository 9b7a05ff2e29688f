// Host-side background cosmology (include/cpt_host.h): the reference's BackgroundModule for flat / curved LambdaCDM
// with massless neutrinos, restated (not translated): same integration variable (ln a), same integrator (ndf15 at rtol 1e-6 with
// dense output on a uniform ln a grid), same derived columns and spline second derivatives, so that the table agrees with the
// reference's to integrator round-off and can be handed to cpt_create unchanged.
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/cpt_host.h"
#include "cpt_ndf15.hpp"

namespace cpt_host {
int fail_msg(int code, const char* fmt, ...);   // cpt_grids.cpp

// array_spline_table_lines (tools/arrays.c:514-690), _SPLINE_EST_DERIV_: second derivatives of ny columns tabulated row-major
void spline_table_lines(const double* x, int n, const double* y, int ny, double* ddy) {
  std::vector<double> u((size_t)(n - 1) * ny), p(ny), qn(ny), un(ny);
  const bool natural = (n == 2);
  for (int c = 0; c < ny; c++) {
    if (natural) { ddy[c] = u[c] = 0.; continue; }
    const double dy_first = ((x[2] - x[0]) * (x[2] - x[0]) * (y[1 * ny + c] - y[0 * ny + c]) - (x[1] - x[0]) * (x[1] - x[0]) * (y[2 * ny + c] - y[0 * ny + c])) /
                            ((x[2] - x[0]) * (x[1] - x[0]) * (x[2] - x[1]));
    ddy[c] = -0.5;
    u[c] = (3. / (x[1] - x[0])) * ((y[1 * ny + c] - y[0 * ny + c]) / (x[1] - x[0]) - dy_first);
  }
  for (int i = 1; i < n - 1; i++) {
    const double sig = (x[i] - x[i - 1]) / (x[i + 1] - x[i - 1]);
    for (int c = 0; c < ny; c++) {
      p[c] = sig * ddy[(size_t)(i - 1) * ny + c] + 2.0;
      ddy[(size_t)i * ny + c] = (sig - 1.0) / p[c];
      double v = (y[(size_t)(i + 1) * ny + c] - y[(size_t)i * ny + c]) / (x[i + 1] - x[i]) - (y[(size_t)i * ny + c] - y[(size_t)(i - 1) * ny + c]) / (x[i] - x[i - 1]);
      u[(size_t)i * ny + c] = (6.0 * v / (x[i + 1] - x[i - 1]) - sig * u[(size_t)(i - 1) * ny + c]) / p[c];
    }
  }
  for (int c = 0; c < ny; c++) {
    if (natural) { qn[c] = un[c] = 0.; continue; }
    const double dy_last = ((x[n - 3] - x[n - 1]) * (x[n - 3] - x[n - 1]) * (y[(size_t)(n - 2) * ny + c] - y[(size_t)(n - 1) * ny + c]) -
                            (x[n - 2] - x[n - 1]) * (x[n - 2] - x[n - 1]) * (y[(size_t)(n - 3) * ny + c] - y[(size_t)(n - 1) * ny + c])) /
                           ((x[n - 3] - x[n - 1]) * (x[n - 2] - x[n - 1]) * (x[n - 3] - x[n - 2]));
    qn[c] = 0.5;
    un[c] = (3. / (x[n - 1] - x[n - 2])) * (dy_last - (y[(size_t)(n - 1) * ny + c] - y[(size_t)(n - 2) * ny + c]) / (x[n - 1] - x[n - 2]));
  }
  for (int c = 0; c < ny; c++)
    ddy[(size_t)(n - 1) * ny + c] = (un[c] - qn[c] * u[(size_t)(n - 2) * ny + c]) / (qn[c] * ddy[(size_t)(n - 2) * ny + c] + 1.0);
  for (int i = n - 2; i >= 0; i--)
    for (int c = 0; c < ny; c++) ddy[(size_t)i * ny + c] = ddy[(size_t)i * ny + c] * ddy[(size_t)(i + 1) * ny + c] + u[(size_t)i * ny + c];
}

// array_interpolate_spline (tools/arrays.c:1565-1628): one row at abscissa v, x ascending or descending
int interpolate_spline(const double* x, int n, const double* y, const double* ddy, int ny, double v, double* out) {
  int inf = 0, sup = n - 1;
  if (x[inf] < x[sup]) {
    if (v < x[inf] || v > x[sup]) return 1;
    while (sup - inf > 1) { const int mid = (int)(0.5 * (inf + sup)); if (v < x[mid]) sup = mid; else inf = mid; }
  } else {
    if (v < x[sup] || v > x[inf]) return 1;
    while (sup - inf > 1) { const int mid = (int)(0.5 * (inf + sup)); if (v > x[mid]) sup = mid; else inf = mid; }
  }
  const double h = x[sup] - x[inf], b = (v - x[inf]) / h, a = 1 - b;
  for (int c = 0; c < ny; c++)
    out[c] = a * y[(size_t)inf * ny + c] + b * y[(size_t)sup * ny + c] + ((a * a * a - a) * ddy[(size_t)inf * ny + c] + (b * b * b - b) * ddy[(size_t)sup * ny + c]) * h * h / 6.;
  return 0;
}

namespace {
const double GYR_OVER_MPC = 3.06601394e2;
enum { BG_a = 0, BG_H, BG_H_prime, BG_rho_g, BG_rho_b, BG_rho_cdm, BG_rho_lambda, BG_rho_ur };   // (cdm / lambda / ur columns exist when present)

struct BgLayout {
  int a, H, Hp, rho_g, rho_b, rho_cdm, rho_lambda, rho_ur, rho_tot, p_tot, p_tot_prime, Omega_r, rho_crit, Omega_m, conf_distance,
      ang_distance, lum_distance, time, rs, D, f, size;
};
BgLayout make_bg_layout(const cpt_cosmo_params& p) {   // background_indices, :832-1025 (the species this restatement knows)
  BgLayout L;
  int i = 0;
  L.a = i++; L.H = i++; L.Hp = i++; L.rho_g = i++; L.rho_b = i++;
  L.rho_cdm = p.has_cdm ? i++ : -1; L.rho_lambda = p.has_lambda ? i++ : -1; L.rho_ur = p.has_ur ? i++ : -1;
  L.rho_tot = i++; L.p_tot = i++; L.p_tot_prime = i++; L.Omega_r = i++;
  L.rho_crit = i++; L.Omega_m = i++; L.conf_distance = i++; L.ang_distance = i++; L.lum_distance = i++; L.time = i++; L.rs = i++;
  L.D = i++; L.f = i++;
  L.size = i;
  return L;
}

// background_functions, :263-610: everything that depends on a alone
int bg_functions(const cpt_cosmo_params& p, const BgLayout& L, double a, bool long_info, double* v) {
  const double a_rel = a / p.a_today, H02 = p.H0 * p.H0;
  if (a_rel <= 0.) return fail_msg(CPT_ERR_INVALID, "a = %e instead of strictly positive", a_rel);
  double rho_tot = 0., p_tot = 0., dp_dloga = 0., rho_r = 0., rho_m = 0.;
  v[L.a] = a;
  v[L.rho_g] = p.Omega0_g * H02 / pow(a_rel, 4);
  rho_tot += v[L.rho_g]; p_tot += 1. / 3. * v[L.rho_g]; dp_dloga += -4. / 3. * v[L.rho_g]; rho_r += v[L.rho_g];
  v[L.rho_b] = p.Omega0_b * H02 / pow(a_rel, 3);
  rho_tot += v[L.rho_b]; rho_m += v[L.rho_b];
  if (p.has_cdm) { v[L.rho_cdm] = p.Omega0_cdm * H02 / pow(a_rel, 3); rho_tot += v[L.rho_cdm]; rho_m += v[L.rho_cdm]; }
  if (p.has_lambda) { v[L.rho_lambda] = p.Omega0_lambda * H02; rho_tot += v[L.rho_lambda]; p_tot -= v[L.rho_lambda]; }
  if (p.has_ur) {
    v[L.rho_ur] = p.Omega0_ur * H02 / pow(a_rel, 4);
    rho_tot += v[L.rho_ur]; p_tot += 1. / 3. * v[L.rho_ur]; dp_dloga += -4. / 3. * v[L.rho_ur]; rho_r += v[L.rho_ur];
  }
  v[L.H] = sqrt(rho_tot - p.K / a / a);
  v[L.Hp] = -3. / 2. * (rho_tot + p_tot) * a + p.K / a;
  v[L.rho_tot] = rho_tot; v[L.p_tot] = p_tot; v[L.p_tot_prime] = a * v[L.H] * dp_dloga;
  const double rho_crit = rho_tot - p.K / a / a;
  if (rho_crit <= 0.) return fail_msg(CPT_ERR_INVALID, "rho_crit = %e instead of strictly positive", rho_crit);
  v[L.Omega_r] = rho_r / rho_crit;
  if (long_info) { v[L.rho_crit] = rho_crit; v[L.Omega_m] = rho_m / rho_crit; }
  return CPT_OK;
}
}  // namespace
}  // namespace cpt_host

using namespace cpt_host;

extern "C" {

void cpt_host_cosmo_defaults(cpt_cosmo_params* p) {
  p->a_ini_over_a_today_default = 1.e-14; p->back_integration_stepsize = 7.e-3; p->tol_initial_Omega_r = 1.e-4;
  p->smallest_allowed_variation = 2.220446049250313e-16;   // DBL_EPSILON (source/input_module.cpp:3481)
}

void cpt_host_background_free(cpt_background* bg) {
  if (!bg) return;
  free(bg->tau_table); free(bg->z_table); free(bg->d2tau_dz2_table); free(bg->background_table); free(bg->d2background_dtau2_table);
  memset(bg, 0, sizeof(*bg));
}

int cpt_host_background(const cpt_cosmo_params* pp, cpt_background* out) {
  if (!pp || !out) return fail_msg(CPT_ERR_INVALID, "null argument");
  const cpt_cosmo_params& p = *pp;
  memset(out, 0, sizeof(*out));
  if (p.has_ncdm || p.has_fld || p.has_scf || p.has_dcdm || p.has_dr || p.has_idr || p.has_idm_dr)
    return fail_msg(CPT_ERR_UNSUPPORTED, "host background: only photons, baryons, cdm, massless neutrinos, Lambda and curvature");
  if (p.a_today <= 0) return fail_msg(CPT_ERR_INVALID, "input a_today = %e instead of strictly positive", p.a_today);
  const BgLayout L = make_bg_layout(p);
  std::vector<double> v(L.size, 0.);
  // ---- background_initial_conditions, :1521-1690 ----
  const double a_ini = p.a_ini_over_a_today_default * p.a_today;
  int rc = bg_functions(p, L, a_ini, false, v.data());
  if (rc) return rc;
  if (fabs(v[L.Omega_r] - 1.) > p.tol_initial_Omega_r)
    return fail_msg(CPT_ERR_INVALID, "Omega_r = %e, not close enough to 1. Decrease a_ini_over_a_today_default in order to start from radiation domination.", v[L.Omega_r]);
  if (v[L.H] <= 0.) return fail_msg(CPT_ERR_INVALID, "H = %e instead of strictly positive", v[L.H]);
  // integrated vector in the reference's order with tau in the slot of a (:1363): tau, proper time, sound horizon, D, D'
  double y[5];
  y[0] = 1. / (a_ini * v[L.H]);
  y[1] = 1. / (2. * v[L.H]);
  y[2] = y[0] / sqrt(3.);
  y[3] = a_ini;
  y[4] = 2 * y[3] * v[L.H];
  // ---- output grid, :1351-1361 ----
  const double loga_ini = log(a_ini), loga_final = log(p.a_today);
  const int n = (int)((loga_final - loga_ini) / p.back_integration_stepsize);
  if (n < 3) return fail_msg(CPT_ERR_INVALID, "background table too short");
  std::vector<double> loga(n);
  for (int i = 0; i < n; i++) loga[i] = loga_ini + i * (loga_final - loga_ini) / (n - 1);
  out->bt_size = n; out->bg_size = L.size;
  out->tau_table = (double*)malloc(sizeof(double) * n); out->z_table = (double*)malloc(sizeof(double) * n);
  out->d2tau_dz2_table = (double*)malloc(sizeof(double) * n);
  out->background_table = (double*)calloc((size_t)n * L.size, sizeof(double));
  out->d2background_dtau2_table = (double*)calloc((size_t)n * L.size, sizeof(double));
  if (!out->tau_table || !out->z_table || !out->d2tau_dz2_table || !out->background_table || !out->d2background_dtau2_table) {
    cpt_host_background_free(out);
    return fail_msg(CPT_ERR_RUNTIME, "could not allocate the background table");
  }
  int err = 0;
  // background_derivs_loga (:2272-2310) on top of background_derivs (:1934-2064)
  auto rhs = [&](double lg, const double* yy, double* dy) {
    const double a = exp(lg);
    if (bg_functions(p, L, a, false, v.data())) { err = 1; }
    const double H = v[L.H];
    double rho_M = v[L.rho_b];
    if (p.has_cdm) rho_M += v[L.rho_cdm];
    dy[0] = 1.0;                                                    // (then scaled like the others: dtau/dlna = 1/(aH))
    dy[1] = a;
    dy[2] = 1. / sqrt(3. * (1. + 3. * v[L.rho_b] / 4. / v[L.rho_g])) * sqrt(1. - p.K * yy[2] * yy[2]);
    dy[3] = yy[4];
    dy[4] = -a * H * yy[4] + 1.5 * a * a * rho_M * yy[3];
    for (int i = 0; i < 5; i++) dy[i] *= 1. / (a * H);
  };
  // background_add_line_to_bg_table (:2312-2344)
  auto add_line = [&](double lg, const double* yy, const double* /*dy*/, int i) {
    const double a = exp(lg);
    out->z_table[i] = std::max(0., p.a_today / exp(lg) - 1.);
    out->tau_table[i] = yy[0];
    double* row = out->background_table + (size_t)i * L.size;
    if (bg_functions(p, L, a, true, row)) err = 1;
    row[L.time] = yy[1]; row[L.rs] = yy[2]; row[L.D] = yy[3];
    row[L.f] = yy[4] / (yy[3] * a * row[L.H]);
  };
  std::vector<int> used(5, 1);
  Ndf S;
  rc = ndf15(rhs, add_line, loga_ini, loga_final, y, used.data(), 5, 1e-6, p.smallest_allowed_variation, loga.data(), n, S);
  if (rc || err) { cpt_host_background_free(out); return fail_msg(CPT_ERR_RUNTIME, "background integration failed (evolver status %d)", rc); }
  out->age = y[1] / GYR_OVER_MPC;
  out->conformal_age = y[0];
  const double D_today = y[3];
  for (int i = 0; i < n; i++) {
    double* row = out->background_table + (size_t)i * L.size;
    const double conformal_distance = out->conformal_age - out->tau_table[i];
    row[L.conf_distance] = conformal_distance;
    double comoving_radius = conformal_distance;
    if (p.sgnK > 0) comoving_radius = sin(sqrt(p.K) * conformal_distance) / sqrt(p.K);
    else if (p.sgnK < 0) comoving_radius = sinh(sqrt(-p.K) * conformal_distance) / sqrt(-p.K);
    row[L.ang_distance] = p.a_today * comoving_radius / (1. + out->z_table[i]);
    row[L.lum_distance] = p.a_today * comoving_radius * (1. + out->z_table[i]);
    row[L.D] /= D_today;
  }
  spline_table_lines(out->z_table, n, out->tau_table, 1, out->d2tau_dz2_table);
  spline_table_lines(out->tau_table, n, out->background_table, L.size, out->d2background_dtau2_table);
  const double* r0 = out->background_table;
  out->Neff = (r0[L.Omega_r] * r0[L.rho_crit] - r0[L.rho_g]) / (7. / 8. * pow(4. / 11., 4. / 3.) * r0[L.rho_g]);
  const double* rl = out->background_table + (size_t)(n - 1) * L.size;
  out->Omega0_m = rl[L.Omega_m]; out->Omega0_r = rl[L.Omega_r]; out->Omega0_de = 1. - (out->Omega0_m + out->Omega0_r + p.Omega0_k);
  out->index_bg_a = L.a; out->index_bg_H = L.H; out->index_bg_H_prime = L.Hp; out->index_bg_rho_g = L.rho_g; out->index_bg_rho_b = L.rho_b;
  out->index_bg_rho_cdm = L.rho_cdm; out->index_bg_rho_lambda = L.rho_lambda; out->index_bg_rho_ur = L.rho_ur; out->index_bg_rho_tot = L.rho_tot;
  out->index_bg_p_tot = L.p_tot; out->index_bg_p_tot_prime = L.p_tot_prime; out->index_bg_Omega_r = L.Omega_r; out->index_bg_rho_crit = L.rho_crit;
  out->index_bg_Omega_m = L.Omega_m; out->index_bg_conf_distance = L.conf_distance; out->index_bg_ang_distance = L.ang_distance;
  out->index_bg_lum_distance = L.lum_distance; out->index_bg_time = L.time; out->index_bg_rs = L.rs; out->index_bg_D = L.D; out->index_bg_f = L.f;
  return CPT_OK;
}

int cpt_host_background_tau_of_z(const cpt_background* bg, double z, double* tau) {
  if (z < bg->z_table[bg->bt_size - 1] || z > bg->z_table[0]) return fail_msg(CPT_ERR_INVALID, "out of range: z=%e outside [%e, %e]", z, bg->z_table[bg->bt_size - 1], bg->z_table[0]);
  if (interpolate_spline(bg->z_table, bg->bt_size, bg->tau_table, bg->d2tau_dz2_table, 1, z, tau)) return fail_msg(CPT_ERR_INVALID, "tau(z): interpolation failed");
  return CPT_OK;
}
}
