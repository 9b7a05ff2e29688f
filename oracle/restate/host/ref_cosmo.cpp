// ORACLE / TEST INFRASTRUCTURE ONLY (built into oracle/libcpt_oracle.so; the product's host library is classpp_public_amd/host/).
// A close restatement of the reference's background and thermodynamics modules that reproduces its tables BIT FOR BIT (same
// integration variable, same integrators - oracle/restate/host/ref_ndf15.hpp, Cash-Karp - same spline recurrences, same order of
// operations): pinned by tests/test_oracle_host.py against the tables dumped from the unmodified reference, and used as the
// checker of the product's own, independently designed host numerics (tests/test_host_cosmo.py).
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

#include "../../../include/cpt_host.h"
#include "ref_ndf15.hpp"

namespace orc_host {
static thread_local std::string g_err;
int fail_msg(int code, const char* fmt, ...) {
  char buf[2048];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

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
  int number_ncdm1, rho_ncdm1, p_ncdm1, pseudo_p_ncdm1;
  int a, H, Hp, rho_g, rho_b, rho_cdm, rho_lambda, rho_ur, rho_tot, p_tot, p_tot_prime, Omega_r, rho_crit, Omega_m, conf_distance,
      ang_distance, lum_distance, time, rs, D, f, size;
};
BgLayout make_bg_layout(const cpt_cosmo_params& p) {   // background_indices, :832-1025 (the species this restatement knows)
  BgLayout L;
  int i = 0;
  L.a = i++; L.H = i++; L.Hp = i++; L.rho_g = i++; L.rho_b = i++;
  L.rho_cdm = p.has_cdm ? i++ : -1;
  L.number_ncdm1 = L.rho_ncdm1 = L.p_ncdm1 = L.pseudo_p_ncdm1 = -1;
  if (p.has_ncdm) { L.number_ncdm1 = i; i += p.N_ncdm; L.rho_ncdm1 = i; i += p.N_ncdm; L.p_ncdm1 = i; i += p.N_ncdm; L.pseudo_p_ncdm1 = i; i += p.N_ncdm; }
  L.rho_lambda = p.has_lambda ? i++ : -1; L.rho_ur = p.has_ur ? i++ : -1;
  L.rho_tot = i++; L.p_tot = i++; L.p_tot_prime = i++; L.Omega_r = i++;
  L.rho_crit = i++; L.Omega_m = i++; L.conf_distance = i++; L.ang_distance = i++; L.lum_distance = i++; L.time = i++; L.rs = i++;
  L.D = i++; L.f = i++;
  L.size = i;
  return L;
}

// NonColdDarkMatter::background_ncdm_momenta_mass (tools/non_cold_dark_matter.cpp:805-846): number, density, pressure and
// pseudo-pressure of species n at redshift z from the background momentum sampling
void ncdm_momenta(const cpt_cosmo_params& p, int n, double z, double* num, double* rho, double* pr, double* pseudo_p) {
  const double factor2 = p.factor_ncdm[n] * pow(1 + z, 4), M = p.M_ncdm[n];
  double sn = 0., srho = 0., sp = 0., spp = 0.;
  for (int iq = 0; iq < p.q_size_ncdm_bg[n]; iq++) {
    const double q2 = p.q_ncdm_bg[n][iq] * p.q_ncdm_bg[n][iq], w = p.w_ncdm_bg[n][iq];
    const double epsilon = sqrt(q2 + M * M / (1. + z) / (1. + z));
    sn += q2 * w;
    srho += q2 * epsilon * w;
    sp += q2 * q2 / 3. / epsilon * w;
    spp += pow(q2 / epsilon, 3) / 3.0 * w;
  }
  *num = sn * (factor2 / (1. + z)); *rho = srho * factor2; *pr = sp * factor2; *pseudo_p = spp * factor2;
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
  if (p.has_ncdm)   // :389-420
    for (int n = 0; n < p.N_ncdm; n++) {
      double num, rho, pr, pp;
      ncdm_momenta(p, n, 1. / a_rel - 1., &num, &rho, &pr, &pp);
      v[L.number_ncdm1 + n] = num; v[L.rho_ncdm1 + n] = rho; v[L.p_ncdm1 + n] = pr; v[L.pseudo_p_ncdm1 + n] = pp;
      rho_tot += rho; p_tot += pr;
      dp_dloga += (pp - 5 * pr);
      rho_r += 3. * pr;
      rho_m += rho - 3. * pr;
    }
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
}  // namespace orc_host

using namespace orc_host;

extern "C" {

const char* orc_host_error(void) { return orc_host::g_err.c_str(); }
// relative tolerance of the background integration: 1e-6 is the reference's (bit-exact tables); a tighter value shows what the
// reference's tables converge to (tests/test_host_cosmo.py compares the product's own integrator with that limit)
static double g_bg_rtol = 1e-6;
void orc_host_set_background_rtol(double rtol) { g_bg_rtol = rtol; }

void orc_host_cosmo_defaults(cpt_cosmo_params* p) {
  p->a_ini_over_a_today_default = 1.e-14; p->back_integration_stepsize = 7.e-3; p->tol_initial_Omega_r = 1.e-4;
  p->smallest_allowed_variation = 2.220446049250313e-16;   // DBL_EPSILON (source/input_module.cpp:3481)
  p->tol_ncdm_initial_w = 1.e-3;
}

void orc_host_background_free(cpt_background* bg) {
  if (!bg) return;
  free(bg->tau_table); free(bg->z_table); free(bg->d2tau_dz2_table); free(bg->background_table); free(bg->d2background_dtau2_table);
  memset(bg, 0, sizeof(*bg));
}

int orc_host_background(const cpt_cosmo_params* pp, cpt_background* out) {
  if (!pp || !out) return fail_msg(CPT_ERR_INVALID, "null argument");
  const cpt_cosmo_params& p = *pp;
  memset(out, 0, sizeof(*out));
  if (p.has_fld || p.has_scf || p.has_dcdm || p.has_dr || p.has_idr || p.has_idm_dr)
    return fail_msg(CPT_ERR_UNSUPPORTED, "host background: only photons, baryons, cdm, massless and massive neutrinos, Lambda and curvature");
  if (p.has_ncdm) {
    if (p.N_ncdm < 1 || p.N_ncdm > CPT_MAX_NCDM) return fail_msg(CPT_ERR_UNSUPPORTED, "host background: between 1 and %d non-cold species", CPT_MAX_NCDM);
    for (int n = 0; n < p.N_ncdm; n++)
      if (p.q_size_ncdm_bg[n] < 1 || !p.q_ncdm_bg[n] || !p.w_ncdm_bg[n])
        return fail_msg(CPT_ERR_INVALID, "host background: the momentum sampling of ncdm species %d is missing", n);
  }
  if (p.a_today <= 0) return fail_msg(CPT_ERR_INVALID, "input a_today = %e instead of strictly positive", p.a_today);
  const BgLayout L = make_bg_layout(p);
  std::vector<double> v(L.size, 0.);
  // ---- background_initial_conditions, :1521-1690 ----
  double a_ini = p.a_ini_over_a_today_default * p.a_today;
  if (p.has_ncdm) {   // NonColdDarkMatter::GetIni (tools/non_cold_dark_matter.cpp:1080-1106): start early enough for every species to be relativistic
    int counter;
    for (counter = 0; counter < 10000; counter++) {
      bool early = true;
      for (int n = 0; n < p.N_ncdm; n++) {
        double num, rho, pr, pp;
        ncdm_momenta(p, n, p.a_today / a_ini - 1.0, &num, &rho, &pr, &pp);
        if (fabs(pr / rho - 1. / 3.) > p.tol_ncdm_initial_w) early = false;
      }
      if (early) break;
      a_ini *= 0.1;
    }
    if (counter == 10000) return fail_msg(CPT_ERR_RUNTIME, "Search for initial scale factor a such that all ncdm species are relativistic failed.");
  }
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
    orc_host_background_free(out);
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
  rc = ndf15(rhs, add_line, loga_ini, loga_final, y, used.data(), 5, g_bg_rtol, p.smallest_allowed_variation, loga.data(), n, S);
  if (rc || err) { orc_host_background_free(out); return fail_msg(CPT_ERR_RUNTIME, "background integration failed (evolver status %d)", rc); }
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
  out->index_bg_number_ncdm1 = L.number_ncdm1; out->index_bg_rho_ncdm1 = L.rho_ncdm1; out->index_bg_p_ncdm1 = L.p_ncdm1; out->index_bg_pseudo_p_ncdm1 = L.pseudo_p_ncdm1;
  out->index_bg_lum_distance = L.lum_distance; out->index_bg_time = L.time; out->index_bg_rs = L.rs; out->index_bg_D = L.D; out->index_bg_f = L.f;
  return CPT_OK;
}

int orc_host_background_tau_of_z(const cpt_background* bg, double z, double* tau) {
  if (z < bg->z_table[bg->bt_size - 1] || z > bg->z_table[0]) return fail_msg(CPT_ERR_INVALID, "out of range: z=%e outside [%e, %e]", z, bg->z_table[bg->bt_size - 1], bg->z_table[0]);
  if (interpolate_spline(bg->z_table, bg->bt_size, bg->tau_table, bg->d2tau_dz2_table, 1, z, tau)) return fail_msg(CPT_ERR_INVALID, "tau(z): interpolation failed");
  return CPT_OK;
}
}

// =====================================================================================================================
// Thermodynamics (include/cpt_host.h): RECFAST + CAMB-like reionization + derived columns, th.cpp = source/thermodynamics_module.cpp
// =====================================================================================================================
namespace orc_host {
namespace {
const double C_LIGHT = 2.99792458e8, G_NEWTON = 6.67428e-11, K_B = 1.3806504e-23, H_P = 6.62606896e-34, MPC_OVER_M = 3.085677581282e22,
             M_ELECTRON = 9.10938215e-31, M_HYDROGEN = 1.673575e-27, NOT4 = 3.9715, SIGMA_T = 6.6524616e-29, PI_ = 3.1415926535897932384626433832795e0,
             E_ = 2.7182818284590452353602874713526624977572470936999595749669676277;
// RECFAST atomic data (source/thermodynamics.h:386-419)
const double LAMBDA_H = 8.2245809, LAMBDA_HE = 51.3, L_H_ION = 1.096787737e7, L_H_ALPHA = 8.225916453e6, L_HE1_ION = 1.98310772e7,
             L_HE2_ION = 4.389088863e7, L_HE_2S = 1.66277434e7, L_HE_2P = 1.71134891e7, A2P_S = 1.798287e9, A2P_T = 177.58e0,
             L_HE_2PT = 1.690871466e7, L_HE_2ST = 1.5985597526e7, L_HE2ST_ION = 3.8454693845e6, SIGMA_HE_2PS = 1.436289e-22,
             SIGMA_HE_2PT = 1.484872e-22, A_PPB = 4.309, B_PPB = -0.6166, C_PPB = 0.6703, D_PPB = 0.5300, B_VF = 0.711, B_TRIP = 0.761;
const double Z_REC_MAX = 2000., Z_REC_MIN = 500., YHE_BIG = 0.5, YHE_SMALL = 0.01;
inline double f1(double x) { return (-0.75 * x * (x * x / 3. - 1.) + 0.5); }   // thermodynamics.h:46-47
inline double f2(double x) { return (x * x * (0.5 - x / 3.) * 6.); }

// ---- one-column spline helpers of tools/arrays.c on a row-major table: array[i*nc + col] ----
void spline_col(const double* x, int n, double* arr, int nc, int iy, int idd) {   // array_spline_table_line_to_line :422-512, EST_DERIV
  std::vector<double> u(n - 1);
  auto Y = [&](int i) -> double& { return arr[(size_t)i * nc + iy]; };
  auto D = [&](int i) -> double& { return arr[(size_t)i * nc + idd]; };
  const double dy_first = ((x[2] - x[0]) * (x[2] - x[0]) * (Y(1) - Y(0)) - (x[1] - x[0]) * (x[1] - x[0]) * (Y(2) - Y(0))) / ((x[2] - x[0]) * (x[1] - x[0]) * (x[2] - x[1]));
  D(0) = -0.5;
  u[0] = (3. / (x[1] - x[0])) * ((Y(1) - Y(0)) / (x[1] - x[0]) - dy_first);
  for (int i = 1; i < n - 1; i++) {
    const double sig = (x[i] - x[i - 1]) / (x[i + 1] - x[i - 1]);
    const double p = sig * D(i - 1) + 2.0;
    D(i) = (sig - 1.0) / p;
    u[i] = (Y(i + 1) - Y(i)) / (x[i + 1] - x[i]) - (Y(i) - Y(i - 1)) / (x[i] - x[i - 1]);
    u[i] = (6.0 * u[i] / (x[i + 1] - x[i - 1]) - sig * u[i - 1]) / p;
  }
  const double dy_last = ((x[n - 3] - x[n - 1]) * (x[n - 3] - x[n - 1]) * (Y(n - 2) - Y(n - 1)) - (x[n - 2] - x[n - 1]) * (x[n - 2] - x[n - 1]) * (Y(n - 3) - Y(n - 1))) /
                         ((x[n - 3] - x[n - 1]) * (x[n - 2] - x[n - 1]) * (x[n - 3] - x[n - 2]));
  const double qn = 0.5, un = (3. / (x[n - 1] - x[n - 2])) * (dy_last - (Y(n - 1) - Y(n - 2)) / (x[n - 1] - x[n - 2]));
  D(n - 1) = (un - qn * u[n - 2]) / (qn * D(n - 2) + 1.0);
  for (int k = n - 2; k >= 0; k--) D(k) = D(k) * D(k + 1) + u[k];
}
void integrate_spline_col(const double* x, int n, double* arr, int nc, int iy, int idd, int iint) {   // :235-258
  arr[iint] = 0.;
  for (int i = 0; i < n - 1; i++) {
    const double h = x[i + 1] - x[i];
    arr[(size_t)(i + 1) * nc + iint] = arr[(size_t)i * nc + iint] + (arr[(size_t)i * nc + iy] + arr[(size_t)(i + 1) * nc + iy]) * h / 2. +
                                        (arr[(size_t)i * nc + idd] + arr[(size_t)(i + 1) * nc + idd]) * h * h * h / 24.;
  }
}
void derive_spline_col(const double* x, int n, double* arr, int nc, int iy, int idd, int idy) {   // :100-145
  for (int i = 0; i < n - 1; i++) {
    const double h = x[i + 1] - x[i];
    arr[(size_t)i * nc + idy] = (arr[(size_t)(i + 1) * nc + iy] - arr[(size_t)i * nc + iy]) / h - h / 6. * (arr[(size_t)(i + 1) * nc + idd] + 2. * arr[(size_t)i * nc + idd]);
  }
  const double h = x[n - 1] - x[n - 2];
  arr[(size_t)(n - 1) * nc + idy] = (arr[(size_t)(n - 1) * nc + iy] - arr[(size_t)(n - 2) * nc + iy]) / h + h / 6. * (2. * arr[(size_t)(n - 1) * nc + idd] + arr[(size_t)(n - 2) * nc + idd]);
}
void smooth_col(double* arr, int nc, int n, int col, int radius) {   // array_smooth :2762-2795
  std::vector<double> sm(n);
  for (int i = 0; i < n; i++) {
    double s = 0., w = 0.;
    const int jmin = std::max(i - radius, 0), jmax = std::min(i + radius, n - 1);
    for (int j = jmin; j <= jmax; j++) { s += arr[(size_t)j * nc + col]; w += 1.; }
    sm[i] = s / w;
  }
  for (int i = 0; i < n; i++) arr[(size_t)i * nc + col] = sm[i];
}

struct BgAccess {   // background_tau_of_z + background_at_tau on the host table
  const cpt_background& bg;
  std::vector<double> row;
  explicit BgAccess(const cpt_background& b) : bg(b), row(b.bg_size) {}
  int tau_of_z(double z, double* tau) const { return orc_host_background_tau_of_z(&bg, z, tau); }
  int at_tau(double tau) {
    if (interpolate_spline(bg.tau_table, bg.bt_size, bg.background_table, bg.d2background_dtau2_table, bg.bg_size, tau, row.data()))
      return fail_msg(CPT_ERR_INVALID, "background_at_tau: tau=%e out of range", tau);
    return CPT_OK;
  }
  double H() const { return row[bg.index_bg_H]; }
  double Hp() const { return row[bg.index_bg_H_prime]; }
  double rho_g() const { return row[bg.index_bg_rho_g]; }
  double rho_b() const { return row[bg.index_bg_rho_b]; }
};

// recombination table columns (struct recombination, source/thermodynamics.h) and RECFAST workspace
enum { RE_Z = 0, RE_XE, RE_TB, RE_WB, RE_CB2, RE_DKAPPADTAU, RE_DKAPPADZ, RE_D3KAPPADZ3, RE_SIZE };
struct Reco {
  double H0, YHe, Tnow, H_frac, fu, fHe, Nnow, CDB, CDB_He, CB1, CB1_He1, CB1_He2, CR, CK, CK_He, CL, CL_He, CT, Bfact;
};

struct Recfast {
  const cpt_cosmo_params& cp; const cpt_thermo_params& tp; Reco re; BgAccess& B;
  int err = 0;
  // thermodynamics_derivs_with_recfast, th.cpp:3727-3975 (no energy injection)
  void derivs(double z, const double* y, double* dy) {
    const double x_H = y[0], x_He = y[1], x = x_H + re.fHe * x_He, Tmat = y[2];
    const double n = re.Nnow * (1. + z) * (1. + z) * (1. + z), n_He = re.fHe * n, Trad = re.Tnow * (1. + z);
    double tau;
    if (B.tau_of_z(z, &tau) || B.at_tau(tau)) { err = 1; dy[0] = dy[1] = dy[2] = 0.; return; }
    const double Hz = B.H() * C_LIGHT / MPC_OVER_M;
    const double Rdown = 1.e-19 * A_PPB * pow((Tmat / 1.e4), B_PPB) / (1. + C_PPB * pow((Tmat / 1.e4), D_PPB));
    const double Rup = Rdown * pow((re.CR * Tmat), 1.5) * exp(-re.CDB / Tmat);
    const double T_0 = pow(10., 0.477121), T_1 = pow(10., 5.114), a_VF = pow(10., -16.744), a_trip = pow(10., -16.306);
    const double sq_0 = sqrt(Tmat / T_0), sq_1 = sqrt(Tmat / T_1);
    const double Rdown_He = a_VF / (sq_0 * pow((1. + sq_0), (1. - B_VF)) * pow((1. + sq_1), (1. + B_VF)));
    const double Rup_He = 4. * Rdown_He * pow((re.CR * Tmat), 1.5) * exp(-re.CDB_He / Tmat);
    double K = re.CK / Hz;
    if (tp.recfast_Hswitch)
      K *= 1. + tp.recfast_AGauss1 * exp(-pow((log(1. + z) - tp.recfast_zGauss1) / tp.recfast_wGauss1, 2)) +
           tp.recfast_AGauss2 * exp(-pow((log(1. + z) - tp.recfast_zGauss2) / tp.recfast_wGauss2, 2));
    const double Rdown_trip = a_trip / (sq_0 * pow((1. + sq_0), (1. - B_TRIP)) * pow((1. + sq_1), (1. + B_TRIP)));
    const double Rup_trip = Rdown_trip * exp(-H_P * C_LIGHT * L_HE2ST_ION / (K_B * Tmat)) * pow(re.CR * Tmat, 1.5) * 4. / 3.;
    int Heflag;
    if ((x_He < 5.e-9) || (x_He > tp.recfast_x_He0_trigger2)) Heflag = 0; else Heflag = tp.recfast_Heswitch;
    double K_He, CfHe_t = 0.;
    if (Heflag == 0) K_He = re.CK_He / Hz;
    else {
      const double tauHe_s = A2P_S * re.CK_He * 3. * n_He * (1. - x_He) / Hz;
      const double pHe_s = (1. - exp(-tauHe_s)) / tauHe_s;
      K_He = 1. / (A2P_S * pHe_s * 3. * n_He * (1. - x_He));
      if (((Heflag == 2) || (Heflag >= 5)) && (x_H < 0.9999999)) {
        double Doppler = 2. * K_B * Tmat / (M_HYDROGEN * NOT4 * C_LIGHT * C_LIGHT);
        Doppler = C_LIGHT * L_HE_2P * sqrt(Doppler);
        const double gamma_2Ps = 3. * A2P_S * re.fHe * (1. - x_He) * C_LIGHT * C_LIGHT / (sqrt(PI_) * SIGMA_HE_2PS * 8. * PI_ * Doppler * (1. - x_H)) / pow(C_LIGHT * L_HE_2P, 2);
        const double pb = 0.36, qb = tp.recfast_fudge_He;
        const double AHcon = A2P_S / (1. + pb * pow(gamma_2Ps, qb));
        K_He = 1. / ((A2P_S * pHe_s + AHcon) * 3. * n_He * (1. - x_He));
      }
      if (Heflag >= 3) {
        const double tauHe_t = A2P_T * n_He * (1. - x_He) * 3. / (8. * PI_ * Hz * pow(L_HE_2PT, 3));
        const double pHe_t = (1. - exp(-tauHe_t)) / tauHe_t;
        const double CL_PSt = H_P * C_LIGHT * (L_HE_2PT - L_HE_2ST) / K_B;
        if ((Heflag == 3) || (Heflag == 5) || (x_H >= 0.99999)) {
          CfHe_t = A2P_T * pHe_t * exp(-CL_PSt / Tmat);
          CfHe_t = CfHe_t / (Rup_trip + CfHe_t);
        } else {
          double Doppler = 2. * K_B * Tmat / (M_HYDROGEN * NOT4 * C_LIGHT * C_LIGHT);
          Doppler = C_LIGHT * L_HE_2PT * sqrt(Doppler);
          const double gamma_2Pt = 3. * A2P_T * re.fHe * (1. - x_He) * C_LIGHT * C_LIGHT / (sqrt(PI_) * SIGMA_HE_2PT * 8. * PI_ * Doppler * (1. - x_H)) / pow(C_LIGHT * L_HE_2PT, 2);
          const double pb = 0.66, qb = 0.9;
          const double AHcon = A2P_T / (1. + pb * pow(gamma_2Pt, qb)) / 3.;
          CfHe_t = (A2P_T * pHe_t + AHcon) * exp(-CL_PSt / Tmat);
          CfHe_t = CfHe_t / (Rup_trip + CfHe_t);
        }
      }
    }
    const double timeTh = (1. / (re.CT * pow(Trad, 4))) * (1. + x + re.fHe) / x;
    const double timeH = 2. / (3. * re.H0 * pow(1. + z, 1.5));
    if (x_H > tp.recfast_x_H0_trigger) dy[0] = 0.;
    else {
      double C;
      if (x_H < tp.recfast_x_H0_trigger2) C = (1. + K * LAMBDA_H * n * (1. - x_H)) / (1. / re.fu + K * LAMBDA_H * n * (1. - x_H) / re.fu + K * Rup * n * (1. - x_H));
      else C = 1.;
      dy[0] = (x * x_H * n * Rdown - Rup * (1. - x_H) * exp(-re.CL / Tmat)) * C / (Hz * (1. + z));
    }
    if (x_He < 1.e-15) dy[1] = 0.;
    else {
      const double He_Boltz = (re.Bfact / Tmat < 680.) ? exp(re.Bfact / Tmat) : exp(680.);
      dy[1] = ((x * x_He * n * Rdown_He - Rup_He * (1. - x_He) * exp(-re.CL_He / Tmat)) * (1. + K_He * LAMBDA_HE * n_He * (1. - x_He) * He_Boltz)) /
              (Hz * (1 + z) * (1. + K_He * (LAMBDA_HE + Rup_He) * n_He * (1. - x_He) * He_Boltz));
      if (Heflag >= 3)
        dy[1] = dy[1] + (x * x_He * n * Rdown_trip - (1. - x_He) * 3. * Rup_trip * exp(-H_P * C_LIGHT * L_HE_2ST / (K_B * Tmat))) * CfHe_t / (Hz * (1. + z));
    }
    if (timeTh < re.H_frac * timeH) {
      const double dHdz = -B.Hp() / B.H() / cp.a_today * C_LIGHT / MPC_OVER_M;
      const double epsilon = Hz * (1. + x + re.fHe) / (re.CT * pow(Trad, 3) * x);
      dy[2] = re.Tnow + epsilon * ((1. + re.fHe) / (1. + re.fHe + x)) * ((dy[0] + re.fHe * dy[1]) / x) - epsilon * dHdz / Hz + 3. * epsilon / (1. + z);
    } else
      dy[2] = re.CT * pow(Trad, 4) * x / (1. + x + re.fHe) * (Tmat - Trad) / (Hz * (1. + z)) + 2. * Tmat / (1. + z);
  }
};

// generic_integrator / rkqs / rkck: tools/dei_rkck.c (Cash-Karp with step-doubling control), 3 equations
struct Rkck {
  double y[3], dydx[3], yscal[3], yerr[3], ytemp[3], ak2[3], ak3[3], ak4[3], ak5[3], ak6[3];
  template <class F>
  void rkck(F& f, double x, double h) {
    for (int i = 0; i < 3; i++) ytemp[i] = y[i] + 0.2 * h * dydx[i];
    f(x + 0.2 * h, ytemp, ak2);
    for (int i = 0; i < 3; i++) ytemp[i] = y[i] + h * (3.0 / 40.0 * dydx[i] + 9.0 / 40.0 * ak2[i]);
    f(x + 0.3 * h, ytemp, ak3);
    for (int i = 0; i < 3; i++) ytemp[i] = y[i] + h * (0.3 * dydx[i] + -0.9 * ak2[i] + 1.2 * ak3[i]);
    f(x + 0.6 * h, ytemp, ak4);
    for (int i = 0; i < 3; i++) ytemp[i] = y[i] + h * (-11.0 / 54.0 * dydx[i] + 2.5 * ak2[i] + -70.0 / 27.0 * ak3[i] + 35.0 / 27.0 * ak4[i]);
    f(x + 1.0 * h, ytemp, ak5);
    for (int i = 0; i < 3; i++)
      ytemp[i] = y[i] + h * (1631.0 / 55296.0 * dydx[i] + 175.0 / 512.0 * ak2[i] + 575.0 / 13824.0 * ak3[i] + 44275.0 / 110592.0 * ak4[i] + 253.0 / 4096.0 * ak5[i]);
    f(x + 0.875 * h, ytemp, ak6);
    for (int i = 0; i < 3; i++) ytemp[i] = y[i] + h * (37.0 / 378.0 * dydx[i] + 250.0 / 621.0 * ak3[i] + 125.0 / 594.0 * ak4[i] + 512.0 / 1771.0 * ak6[i]);
    for (int i = 0; i < 3; i++)
      yerr[i] = h * ((37.0 / 378.0 - 2825.0 / 27648.) * dydx[i] + (250.0 / 621.0 - 18575.0 / 48384.0) * ak3[i] + (125.0 / 594.0 - 13525.0 / 55296.0) * ak4[i] +
                     -277.00 / 14336.0 * ak5[i] + (512.0 / 1771.0 - 0.25) * ak6[i]);
  }
  template <class F>
  int integrate(F& f, double x1, double x2, double* ystart, double eps, double hmin) {
    const double h1 = x2 - x1;
    double x = x1, h = ((x2 - x1) > 0. ? h1 : -h1), hnext = 0.;
    for (int i = 0; i < 3; i++) y[i] = ystart[i];
    for (int nstp = 1; nstp <= 100000; nstp++) {
      f(x, y, dydx);
      for (int i = 0; i < 3; i++) yscal[i] = fabs(y[i]) + fabs(dydx[i] * h) + 1.0e-30;
      if ((x + h - x2) * (x + h - x1) > 0.0) h = x2 - x;
      {  // rkqs
        double errmax, hh = h;
        for (;;) {
          rkck(f, x, hh);
          errmax = 0.0;
          for (int i = 0; i < 3; i++) errmax = std::max(errmax, fabs(yerr[i] / yscal[i]));
          errmax /= eps;
          if (errmax <= 1.0) break;
          const double htemp = 0.9 * hh * pow(errmax, -0.25);
          hh = (hh >= 0.0 ? std::max(htemp, 0.1 * hh) : std::min(htemp, 0.1 * hh));
          if (x + hh == x) return fail_msg(CPT_ERR_RUNTIME, "stepsize underflow at x=%e", x);
        }
        if (errmax > 1.89e-4) hnext = 0.9 * hh * pow(errmax, -0.2); else hnext = 5.0 * hh;
        x += hh;
        for (int i = 0; i < 3; i++) y[i] = ytemp[i];
      }
      if ((x - x2) * (x2 - x1) >= 0.0) { for (int i = 0; i < 3; i++) ystart[i] = y[i]; return CPT_OK; }
      if (fabs(hnext / x1) <= hmin) return fail_msg(CPT_ERR_RUNTIME, "Step size too small: step:%g, minimum:%g, in interval: [%g:%g]", fabs(hnext / x1), hmin, x1, x2);
      h = hnext;
    }
    return fail_msg(CPT_ERR_RUNTIME, "Too many integration steps needed within interval [%g : %g]", x1, x2);
  }
};

// thermodynamics_recombination_with_recfast, th.cpp:3335-3697: table [Nz][RE_SIZE] in growing z
int recombination(const cpt_cosmo_params& cp, const cpt_thermo_params& tp, BgAccess& B, Reco& re, std::vector<double>& tab) {
  const int Nz = tp.recfast_Nz0;
  tab.assign((size_t)Nz * RE_SIZE, 0.);
  re.H0 = cp.H0 * C_LIGHT / MPC_OVER_M;
  re.YHe = tp.YHe; re.Tnow = cp.T_cmb; re.H_frac = tp.recfast_H_frac;
  re.fu = tp.recfast_fudge_H;
  if (tp.recfast_Hswitch) re.fu += tp.recfast_delta_fudge_H;
  if (tp.recfast_Heswitch < 0 || tp.recfast_Heswitch > 6) return fail_msg(CPT_ERR_INVALID, "RECFAST error: unknown He fudging scheme");
  const double zinitial = tp.recfast_z_initial;
  const double mu_H = 1. / (1. - re.YHe);
  re.fHe = re.YHe / (NOT4 * (1. - re.YHe));
  re.Nnow = 3. * re.H0 * re.H0 * cp.Omega0_b / (8. * PI_ * G_NEWTON * mu_H * M_HYDROGEN);
  const double Lalpha = 1. / L_H_ALPHA, Lalpha_He = 1. / L_HE_2P;
  const double DeltaB = H_P * C_LIGHT * (L_H_ION - L_H_ALPHA);
  re.CDB = DeltaB / K_B;
  const double DeltaB_He = H_P * C_LIGHT * (L_HE1_ION - L_HE_2S);
  re.CDB_He = DeltaB_He / K_B;
  re.CB1 = H_P * C_LIGHT * L_H_ION / K_B;
  re.CB1_He1 = H_P * C_LIGHT * L_HE1_ION / K_B;
  re.CB1_He2 = H_P * C_LIGHT * L_HE2_ION / K_B;
  re.CR = 2. * PI_ * (M_ELECTRON / H_P) * (K_B / H_P);
  re.CK = pow(Lalpha, 3) / (8. * PI_);
  re.CK_He = pow(Lalpha_He, 3) / (8. * PI_);
  re.CL = C_LIGHT * H_P / (K_B * Lalpha);
  re.CL_He = C_LIGHT * H_P / (K_B / L_HE_2S);
  re.CT = (8. / 3.) * (SIGMA_T / (M_ELECTRON * C_LIGHT)) * (8. * pow(PI_, 5) * pow(K_B, 4) / 15. / pow(H_P, 3) / pow(C_LIGHT, 3));
  re.Bfact = H_P * C_LIGHT * (L_HE_2P - L_HE_2S) / K_B;
  if (zinitial < tp.recfast_z_He_3) return fail_msg(CPT_ERR_INVALID, "increase zinitial, otherwise should get initial conditions from recfast's get_init routine");
  Recfast R{cp, tp, re, B};
  auto f = [&](double z, const double* y, double* dy) { R.derivs(z, y, dy); };
  Rkck gi;
  double y[3], dy[3];
  double z = zinitial, x0 = 1. + 2. * re.fHe, x_H0 = 0., x_He0;
  y[0] = 1.; y[1] = 1.; y[2] = re.Tnow * (1. + z);
  const double smallest = cp.smallest_allowed_variation;
  for (int i = 0; i < Nz; i++) {
    const double zstart = zinitial * (double)(Nz - i) / (double)Nz;
    const double zend = zinitial * (double)(Nz - i - 1) / (double)Nz;
    z = zend;
    if (z > tp.recfast_z_He_1 + tp.recfast_delta_z_He_1) {
      x_H0 = 1.; x_He0 = 1.; x0 = 1. + 2. * re.fHe;
      y[0] = x_H0; y[1] = x_He0; y[2] = re.Tnow * (1. + z);
    } else if (z > tp.recfast_z_He_2 + tp.recfast_delta_z_He_2) {
      x_H0 = 1.; x_He0 = 1.;
      const double rhs = exp(1.5 * log(re.CR * re.Tnow / (1. + z)) - re.CB1_He2 / (re.Tnow * (1. + z))) / re.Nnow;
      if (z > tp.recfast_z_He_1 - tp.recfast_delta_z_He_1) {
        const double x0_previous = 1. + 2. * re.fHe;
        const double x0_new = 0.5 * (sqrt(pow((rhs - 1. - re.fHe), 2) + 4. * (1. + 2. * re.fHe) * rhs) - (rhs - 1. - re.fHe));
        const double s = (tp.recfast_z_He_1 - z) / tp.recfast_delta_z_He_1, weight = f1(s);
        x0 = weight * x0_new + (1. - weight) * x0_previous;
      } else x0 = 0.5 * (sqrt(pow((rhs - 1. - re.fHe), 2) + 4. * (1. + 2. * re.fHe) * rhs) - (rhs - 1. - re.fHe));
      y[0] = x_H0; y[1] = x_He0; y[2] = re.Tnow * (1. + z);
    } else if (z > tp.recfast_z_He_3 + tp.recfast_delta_z_He_3) {
      x_H0 = 1.; x_He0 = 1.;
      if (z > tp.recfast_z_He_2 - tp.recfast_delta_z_He_2) {
        const double rhs = exp(1.5 * log(re.CR * re.Tnow / (1. + z)) - re.CB1_He2 / (re.Tnow * (1. + z))) / re.Nnow;
        const double x0_previous = 0.5 * (sqrt(pow((rhs - 1. - re.fHe), 2) + 4. * (1. + 2. * re.fHe) * rhs) - (rhs - 1. - re.fHe));
        const double x0_new = 1. + re.fHe;
        const double s = (tp.recfast_z_He_2 - z) / tp.recfast_delta_z_He_2, weight = f1(s);
        x0 = weight * x0_new + (1. - weight) * x0_previous;
      } else x0 = 1. + re.fHe;
      y[0] = x_H0; y[1] = x_He0; y[2] = re.Tnow * (1. + z);
    } else if (y[1] > tp.recfast_x_He0_trigger) {
      x_H0 = 1.;
      const double rhs = 4. * exp(1.5 * log(re.CR * re.Tnow / (1. + z)) - re.CB1_He1 / (re.Tnow * (1. + z))) / re.Nnow;
      x_He0 = 0.5 * (sqrt(pow((rhs - 1.), 2) + 4. * (1. + re.fHe) * rhs) - (rhs - 1.));
      if (z > tp.recfast_z_He_3 - tp.recfast_delta_z_He_3) {
        const double x0_previous = 1. + re.fHe, x0_new = x_He0;
        const double s = (tp.recfast_z_He_3 - z) / tp.recfast_delta_z_He_3, weight = f1(s);
        x0 = weight * x0_new + (1. - weight) * x0_previous;
      } else x0 = x_He0;
      x_He0 = (x0 - 1.) / re.fHe;
      y[0] = x_H0; y[1] = x_He0; y[2] = re.Tnow * (1. + z);
    } else if (y[0] > tp.recfast_x_H0_trigger) {
      double rhs = exp(1.5 * log(re.CR * re.Tnow / (1. + z)) - re.CB1 / (re.Tnow * (1. + z))) / re.Nnow;
      x_H0 = 0.5 * (sqrt(pow(rhs, 2) + 4. * rhs) - rhs);
      int rc = gi.integrate(f, zstart, zend, y, tp.tol_thermo_integration, smallest);
      if (rc) return rc;
      y[0] = x_H0;
      if (tp.recfast_x_He0_trigger - y[1] < tp.recfast_x_He0_trigger_delta) {
        rhs = 4. * exp(1.5 * log(re.CR * re.Tnow / (1. + z)) - re.CB1_He1 / (re.Tnow * (1. + z))) / re.Nnow;
        const double x0_previous = 0.5 * (sqrt(pow((rhs - 1.), 2) + 4. * (1. + re.fHe) * rhs) - (rhs - 1.));
        const double x0_new = y[0] + re.fHe * y[1];
        const double s = (tp.recfast_x_He0_trigger - y[1]) / tp.recfast_x_He0_trigger_delta, weight = f2(s);
        x0 = weight * x0_new + (1. - weight) * x0_previous;
      } else x0 = y[0] + re.fHe * y[1];
    } else {
      if (tp.recfast_x_H0_trigger - y[0] < tp.recfast_x_H0_trigger_delta) {
        const double rhs = exp(1.5 * log(re.CR * re.Tnow / (1. + z)) - re.CB1 / (re.Tnow * (1. + z))) / re.Nnow;
        x_H0 = 0.5 * (sqrt(pow(rhs, 2) + 4. * rhs) - rhs);
      }
      int rc = gi.integrate(f, zstart, zend, y, tp.tol_thermo_integration, smallest);
      if (rc) return rc;
      if (tp.recfast_x_H0_trigger - y[0] < tp.recfast_x_H0_trigger_delta) {
        const double s = (tp.recfast_x_H0_trigger - y[0]) / tp.recfast_x_H0_trigger_delta, weight = f2(s);
        x0 = weight * y[0] + (1. - weight) * x_H0 + re.fHe * y[1];
      } else x0 = y[0] + re.fHe * y[1];
    }
    if (R.err) return fail_msg(CPT_ERR_RUNTIME, "recfast: background look-up failed at z=%e", z);
    double* row = &tab[(size_t)(Nz - i - 1) * RE_SIZE];
    row[RE_Z] = zend; row[RE_XE] = x0; row[RE_TB] = y[2];
    R.derivs(zend, y, dy);
    row[RE_WB] = K_B / (C_LIGHT * C_LIGHT * M_HYDROGEN) * (1. + (1. / NOT4 - 1.) * re.YHe + x0 * (1. - re.YHe)) * y[2];
    row[RE_CB2] = row[RE_WB] * (1. + (1. + zend) * dy[2] / y[2] / 3.);
    row[RE_DKAPPADTAU] = (1. + zend) * (1. + zend) * re.Nnow * x0 * SIGMA_T * MPC_OVER_M;
  }
  return CPT_OK;
}

// reionization: th.cpp:1893-1950 (CAMB-like tanh), 2668-2990 (adaptive sampling, T_b, optical depth)
struct Reio { double xe_before, xe_after, z_reio, z_start, exponent, width, he_frac, he_z, he_width; };
double reio_xe(const Reio& r, double z) {
  if (z > r.z_start) return r.xe_before;
  double argument = (pow((1. + r.z_reio), r.exponent) - pow((1. + z), r.exponent)) / (r.exponent * pow((1. + r.z_reio), (r.exponent - 1.))) / r.width;
  double xe = (r.xe_after - r.xe_before) * (tanh(argument) + 1.) / 2. + r.xe_before;
  argument = (r.he_z - z) / r.he_width;
  xe += r.he_frac * (tanh(argument) + 1.) / 2.;
  return xe;
}
int xe_before_reio(const std::vector<double>& reco, int Nz, double z, double* xe) {   // array_interpolate_one_growing_closeby from index 0
  int inf = 0;
  while (z < reco[(size_t)inf * RE_SIZE + RE_Z]) { inf--; if (inf < 0) return fail_msg(CPT_ERR_INVALID, "x=%e < x_min", z); }
  int sup = inf + 1;
  while (z > reco[(size_t)sup * RE_SIZE + RE_Z]) { sup++; if (sup > Nz - 1) return fail_msg(CPT_ERR_INVALID, "x=%e > x_max", z); }
  inf = sup - 1;
  const double weight = (z - reco[(size_t)inf * RE_SIZE + RE_Z]) / (reco[(size_t)sup * RE_SIZE + RE_Z] - reco[(size_t)inf * RE_SIZE + RE_Z]);
  *xe = reco[(size_t)inf * RE_SIZE + RE_XE] * (1. - weight) + reco[(size_t)sup * RE_SIZE + RE_XE] * weight;
  return CPT_OK;
}
int reio_sample(const cpt_cosmo_params& cp, const cpt_thermo_params& tp, BgAccess& B, const Reco& re, const std::vector<double>& reco, const Reio& r,
                std::vector<double>& tab, int* rt_size, int* index_reco_when_reio_start, double* optical_depth) {
  const int Nz = tp.recfast_Nz0;
  const double Yp = tp.YHe, n_e = re.Nnow;
  std::vector<double> grow;   // rows in decreasing z
  double vec[RE_SIZE] = {0};
  int i = 0;
  while (reco[(size_t)i * RE_SIZE + RE_Z] < r.z_start) {
    i++;
    if (i == Nz) return fail_msg(CPT_ERR_INVALID, "reionization_z_start_max = %e > largest redshift in thermodynamics table", tp.reionization_z_start_max);
  }
  double z = reco[(size_t)i * RE_SIZE + RE_Z];
  vec[RE_Z] = z;
  *index_reco_when_reio_start = i;
  double xe = reio_xe(r, z);
  vec[RE_XE] = xe;
  double tau;
  int rc;
  if ((rc = B.tau_of_z(z, &tau)) || (rc = B.at_tau(tau))) return rc;
  vec[RE_DKAPPADTAU] = (1. + z) * (1. + z) * n_e * xe * SIGMA_T * MPC_OVER_M;
  if (B.H() == 0.) return fail_msg(CPT_ERR_INVALID, "stop to avoid division by zero");
  vec[RE_DKAPPADZ] = vec[RE_DKAPPADTAU] / B.H();
  double dkappadz = vec[RE_DKAPPADZ], dkappadtau = vec[RE_DKAPPADTAU];
  const double Tb = reco[(size_t)i * RE_SIZE + RE_TB];
  vec[RE_TB] = Tb;
  vec[RE_WB] = K_B / (C_LIGHT * C_LIGHT * M_HYDROGEN) * (1. + (1. / NOT4 - 1.) * Yp + xe * (1. - Yp)) * Tb;
  vec[RE_CB2] = 5. / 3. * vec[RE_WB];
  grow.insert(grow.end(), vec, vec + RE_SIZE);
  int number_of_redshifts = 1;
  const double dz_max = reco[(size_t)i * RE_SIZE + RE_Z] - reco[(size_t)(i - 1) * RE_SIZE + RE_Z];
  double dz = dz_max;
  while (z > 0.) {
    if (dz < cp.smallest_allowed_variation) return fail_msg(CPT_ERR_RUNTIME, "stuck in the loop for reionization sampling, as if you were trying to impose a discontinuous evolution for xe(z)");
    double z_next = z - dz;
    if (z_next < 0.) z_next = 0.;
    const double xe_next = reio_xe(r, z_next);
    if ((rc = B.tau_of_z(z_next, &tau)) || (rc = B.at_tau(tau))) return rc;
    if (B.H() == 0.) return fail_msg(CPT_ERR_INVALID, "stop to avoid division by zero");
    const double dkappadz_next = (1. + z_next) * (1. + z_next) * n_e * xe_next * SIGMA_T * MPC_OVER_M / B.H();
    const double dkappadtau_next = (1. + z_next) * (1. + z_next) * n_e * xe_next * SIGMA_T * MPC_OVER_M;
    if ((dkappadz == 0.) || (dkappadtau == 0.)) return fail_msg(CPT_ERR_INVALID, "stop to avoid division by zero");
    const double relative_variation = fabs((dkappadz_next - dkappadz) / dkappadz) + fabs((dkappadtau_next - dkappadtau) / dkappadtau);
    if (relative_variation < tp.reionization_sampling) {
      z = z_next; xe = xe_next; dkappadz = dkappadz_next; dkappadtau = dkappadtau_next;
      if ((dkappadz == 0.) || (dkappadtau == 0.)) return fail_msg(CPT_ERR_INVALID, "dkappadz=%e, dkappadtau=%e, stop to avoid division by zero", dkappadz, dkappadtau);
      vec[RE_Z] = z; vec[RE_XE] = xe; vec[RE_DKAPPADZ] = dkappadz; vec[RE_DKAPPADTAU] = dkappadz * B.H();
      grow.insert(grow.end(), vec, vec + RE_SIZE);
      number_of_redshifts++;
      dz = std::min(0.9 * (tp.reionization_sampling / relative_variation), 5.) * dz;
      dz = std::min(dz, dz_max);
    } else dz = 0.9 * (tp.reionization_sampling / relative_variation) * dz;
  }
  const int n = number_of_redshifts;
  tab.assign((size_t)n * RE_SIZE, 0.);
  for (int j = 0; j < n; j++) memcpy(&tab[(size_t)j * RE_SIZE], &grow[(size_t)(n - j - 1) * RE_SIZE], RE_SIZE * sizeof(double));
  *rt_size = n;
  // baryon temperature by forward Euler in decreasing z (th.cpp:2871-2957)
  for (int j = n - 1; j > 0; j--) {
    const double zz = tab[(size_t)j * RE_SIZE + RE_Z];
    if ((rc = B.tau_of_z(zz, &tau)) || (rc = B.at_tau(tau))) return rc;
    const double dzz = tab[(size_t)j * RE_SIZE + RE_Z] - tab[(size_t)(j - 1) * RE_SIZE + RE_Z];
    const double opacity = (1. + zz) * (1. + zz) * n_e * tab[(size_t)j * RE_SIZE + RE_XE] * SIGMA_T * MPC_OVER_M;
    const double mu = M_HYDROGEN / (1. + (1. / NOT4 - 1.) * tp.YHe + tab[(size_t)j * RE_SIZE + RE_XE] * (1. - tp.YHe));
    const double dTdz = 2. / (1 + zz) * tab[(size_t)j * RE_SIZE + RE_TB] -
                        2. * mu / M_ELECTRON * 4. * B.rho_g() / 3. / B.rho_b() * opacity * (cp.T_cmb * (1. + zz) - tab[(size_t)j * RE_SIZE + RE_TB]) / B.H();
    tab[(size_t)(j - 1) * RE_SIZE + RE_TB] = tab[(size_t)j * RE_SIZE + RE_TB] - dTdz * dzz;
    tab[(size_t)(j - 1) * RE_SIZE + RE_WB] = K_B / (C_LIGHT * C_LIGHT * mu) * tab[(size_t)(j - 1) * RE_SIZE + RE_TB];
    tab[(size_t)(j - 1) * RE_SIZE + RE_CB2] = tab[(size_t)(j - 1) * RE_SIZE + RE_WB] * (1. + (1 + zz) / 3. * dTdz / tab[(size_t)(j - 1) * RE_SIZE + RE_TB]);
  }
  // optical depth: spline of dkappa/dz in z, integrated (array_spline + array_integrate_all_spline)
  std::vector<double> zz(n);
  for (int j = 0; j < n; j++) zz[j] = tab[(size_t)j * RE_SIZE + RE_Z];
  spline_col(zz.data(), n, tab.data(), RE_SIZE, RE_DKAPPADZ, RE_D3KAPPADZ3);
  double res = 0.;
  for (int j = 0; j < n - 1; j++) {
    const double h = zz[j + 1] - zz[j];
    res += (tab[(size_t)j * RE_SIZE + RE_DKAPPADZ] + tab[(size_t)(j + 1) * RE_SIZE + RE_DKAPPADZ]) * h / 2. +
           (tab[(size_t)j * RE_SIZE + RE_D3KAPPADZ3] + tab[(size_t)(j + 1) * RE_SIZE + RE_D3KAPPADZ3]) * h * h * h / 24.;
  }
  *optical_depth = res;
  return CPT_OK;
}
}  // namespace
}  // namespace orc_host

extern "C" {

void orc_host_thermo_defaults(cpt_thermo_params* p) {
  p->reionization_exponent = 1.5; p->reionization_width = 0.5; p->helium_fullreio_redshift = 3.5; p->helium_fullreio_width = 0.5;
  p->recfast_z_initial = 1.0e4; p->recfast_Nz0 = 20000; p->tol_thermo_integration = 1.0e-2;
  p->recfast_Heswitch = 6; p->recfast_fudge_He = 0.86; p->recfast_Hswitch = 1; p->recfast_fudge_H = 1.14; p->recfast_delta_fudge_H = -0.015;
  p->recfast_AGauss1 = -0.14; p->recfast_AGauss2 = 0.079; p->recfast_zGauss1 = 7.28; p->recfast_zGauss2 = 6.73; p->recfast_wGauss1 = 0.18;
  p->recfast_wGauss2 = 0.33; p->recfast_z_He_1 = 8000.0; p->recfast_delta_z_He_1 = 50.0; p->recfast_z_He_2 = 5000.0; p->recfast_delta_z_He_2 = 100.0;
  p->recfast_z_He_3 = 3500.0; p->recfast_delta_z_He_3 = 50.0; p->recfast_x_He0_trigger = 0.995; p->recfast_x_He0_trigger2 = 0.995;
  p->recfast_x_He0_trigger_delta = 0.05; p->recfast_x_H0_trigger = 0.995; p->recfast_x_H0_trigger2 = 0.995; p->recfast_x_H0_trigger_delta = 0.05;
  p->recfast_H_frac = 1.0e-3;
  p->reionization_z_start_max = 50.0; p->reionization_sampling = 5.0e-2; p->reionization_optical_depth_tol = 1.0e-4; p->reionization_start_factor = 8.0;
  p->thermo_rate_smoothing_radius = 50;
  p->radiation_streaming_trigger_tau_c_over_tau = 5.0; p->neglect_CMB_sources_below_visibility = 1.e-3;
}

void orc_host_thermo_free(cpt_thermo* th) {
  if (!th) return;
  free(th->z_table); free(th->thermodynamics_table); free(th->d2thermodynamics_dz2_table);
  memset(th, 0, sizeof(*th));
}

int orc_host_thermodynamics(const cpt_cosmo_params* cpp, const cpt_thermo_params* tpp, const cpt_background* bg, cpt_thermo* out) {
  if (!cpp || !tpp || !bg || !out) return fail_msg(CPT_ERR_INVALID, "null argument");
  const cpt_cosmo_params& cp = *cpp;
  const cpt_thermo_params& tp = *tpp;
  memset(out, 0, sizeof(*out));
  if (tp.reio_parametrization != CPT_REIO_NONE && tp.reio_parametrization != CPT_REIO_CAMB)
    return fail_msg(CPT_ERR_UNSUPPORTED, "host thermodynamics: reionization schemes none and camb only");
  if ((tp.YHe < YHE_SMALL) || (tp.YHe > YHE_BIG)) return fail_msg(CPT_ERR_INVALID, "Y_He=%g out of bounds (%g<Y_He<%g)", tp.YHe, YHE_SMALL, YHE_BIG);
  BgAccess B(*bg);
  Reco re;
  std::vector<double> reco, reio;
  int rc = recombination(cp, tp, B, re, reco);
  if (rc) return rc;
  const int Nz = tp.recfast_Nz0;
  int rt_size = 0, index_reco_when_reio_start = -1;
  double z_reionization = tp.z_reio, tau_reionization = tp.tau_reio;
  if (tp.reio_parametrization == CPT_REIO_CAMB) {   // thermodynamics_reionization, th.cpp:2159-2320
    Reio r;
    r.xe_after = 1. + tp.YHe / (NOT4 * (1. - tp.YHe));
    r.exponent = tp.reionization_exponent; r.width = tp.reionization_width;
    r.he_frac = tp.YHe / (NOT4 * (1. - tp.YHe)); r.he_z = tp.helium_fullreio_redshift; r.he_width = tp.helium_fullreio_width;
    if (r.exponent == 0 || r.width == 0 || r.he_width == 0) return fail_msg(CPT_ERR_INVALID, "stop to avoid division by zero");
    auto start_of = [&](double zr) {
      double zs = zr + tp.reionization_start_factor * tp.reionization_width;
      if (zs < tp.helium_fullreio_redshift + tp.reionization_start_factor * tp.helium_fullreio_width)
        zs = tp.helium_fullreio_redshift + tp.reionization_start_factor * tp.helium_fullreio_width;
      return zs;
    };
    double depth = 0.;
    if (!tp.reio_from_tau) {
      r.z_reio = z_reionization; r.z_start = start_of(r.z_reio);
      if (r.z_start > tp.reionization_z_start_max) return fail_msg(CPT_ERR_INVALID, "starting redshift for reionization > reionization_z_start_max = %e", tp.reionization_z_start_max);
      if ((rc = xe_before_reio(reco, Nz, r.z_start, &r.xe_before))) return rc;
      if ((rc = reio_sample(cp, tp, B, re, reco, r, reio, &rt_size, &index_reco_when_reio_start, &depth))) return rc;
      tau_reionization = depth;
    } else {
      double z_sup = tp.reionization_z_start_max - tp.reionization_start_factor * tp.reionization_width;
      if (z_sup < 0.) return fail_msg(CPT_ERR_INVALID, "parameters are such that reionization cannot take place before today while starting after z_start_max; need to increase z_start_max");
      r.z_reio = z_sup; r.z_start = tp.reionization_z_start_max;
      if ((rc = xe_before_reio(reco, Nz, r.z_start, &r.xe_before))) return rc;
      if ((rc = reio_sample(cp, tp, B, re, reco, r, reio, &rt_size, &index_reco_when_reio_start, &depth))) return rc;
      double tau_sup = depth;
      if (tau_sup < tau_reionization) return fail_msg(CPT_ERR_INVALID, "parameters are such that reionization cannot start after z_start_max");
      double z_inf = 0., tau_inf = 0.;
      int counter = 0;
      while ((tau_sup - tau_inf) > tau_reionization * tp.reionization_optical_depth_tol) {
        const double z_mid = 0.5 * (z_sup + z_inf);
        r.z_reio = z_mid; r.z_start = start_of(z_mid);
        if (r.z_start > tp.reionization_z_start_max) return fail_msg(CPT_ERR_INVALID, "starting redshift for reionization > reionization_z_start_max = %e", tp.reionization_z_start_max);
        if ((rc = xe_before_reio(reco, Nz, r.z_start, &r.xe_before))) return rc;
        if ((rc = reio_sample(cp, tp, B, re, reco, r, reio, &rt_size, &index_reco_when_reio_start, &depth))) return rc;
        const double tau_mid = depth;
        if (tau_mid > tau_reionization) { z_sup = z_mid; tau_sup = tau_mid; } else { z_inf = z_mid; tau_inf = tau_mid; }
        if (++counter > 10000) return fail_msg(CPT_ERR_RUNTIME, "while searching for reionization_optical_depth, maximum number of iterations exceeded");
      }
      z_reionization = r.z_reio;
    }
  }
  // ---- thermodynamics_merge_reco_and_reio, th.cpp:3977-4085 ----
  enum { TH_xe = 0, TH_dkappa, TH_tau_d, TH_ddkappa, TH_dddkappa, TH_exp_m_kappa, TH_g, TH_dg, TH_ddg, TH_Tb, TH_wb, TH_cb2, TH_rate, TH_SIZE };
  if (rt_size > 0 && reco[(size_t)index_reco_when_reio_start * RE_SIZE + RE_Z] != reio[(size_t)(rt_size - 1) * RE_SIZE + RE_Z])
    return fail_msg(CPT_ERR_RUNTIME, "mismatch which should never happen");
  const int nt = Nz + rt_size - index_reco_when_reio_start - 1;
  const int nc = TH_SIZE;
  out->tt_size = nt; out->th_size = nc;
  out->z_table = (double*)malloc(sizeof(double) * nt);
  out->thermodynamics_table = (double*)calloc((size_t)nt * nc, sizeof(double));
  out->d2thermodynamics_dz2_table = (double*)calloc((size_t)nt * nc, sizeof(double));
  if (!out->z_table || !out->thermodynamics_table || !out->d2thermodynamics_dz2_table) { orc_host_thermo_free(out); return fail_msg(CPT_ERR_RUNTIME, "could not allocate the thermodynamics table"); }
  double* T = out->thermodynamics_table;
  double* zt = out->z_table;
  for (int i = 0; i < rt_size; i++) {
    zt[i] = reio[(size_t)i * RE_SIZE + RE_Z];
    T[(size_t)i * nc + TH_xe] = reio[(size_t)i * RE_SIZE + RE_XE]; T[(size_t)i * nc + TH_dkappa] = reio[(size_t)i * RE_SIZE + RE_DKAPPADTAU];
    T[(size_t)i * nc + TH_Tb] = reio[(size_t)i * RE_SIZE + RE_TB]; T[(size_t)i * nc + TH_wb] = reio[(size_t)i * RE_SIZE + RE_WB];
    T[(size_t)i * nc + TH_cb2] = reio[(size_t)i * RE_SIZE + RE_CB2];
  }
  for (int i = 0; i < Nz - index_reco_when_reio_start - 1; i++) {
    const int ith = i + rt_size, ire = i + index_reco_when_reio_start + 1;
    zt[ith] = reco[(size_t)ire * RE_SIZE + RE_Z];
    T[(size_t)ith * nc + TH_xe] = reco[(size_t)ire * RE_SIZE + RE_XE]; T[(size_t)ith * nc + TH_dkappa] = reco[(size_t)ire * RE_SIZE + RE_DKAPPADTAU];
    T[(size_t)ith * nc + TH_Tb] = reco[(size_t)ire * RE_SIZE + RE_TB]; T[(size_t)ith * nc + TH_wb] = reco[(size_t)ire * RE_SIZE + RE_WB];
    T[(size_t)ith * nc + TH_cb2] = reco[(size_t)ire * RE_SIZE + RE_CB2];
  }
  auto bail = [&](int code) { orc_host_thermo_free(out); return code; };
  // ---- derived columns, th.cpp:456-790 ----
  std::vector<double> tau_table(nt);
  for (int i = 0; i < nt; i++) if ((rc = B.tau_of_z(zt[i], &tau_table[i]))) return bail(rc);
  out->tau_ini = tau_table[nt - 1];
  for (int i = 0; i < nt; i++) {   // minus the baryon drag rate -[1/R kappa'], temporarily in column ddkappa
    if ((rc = B.at_tau(tau_table[i]))) return bail(rc);
    const double R = 3. / 4. * B.rho_b() / B.rho_g();
    T[(size_t)i * nc + TH_ddkappa] = -1. / R * T[(size_t)i * nc + TH_dkappa];
  }
  spline_col(tau_table.data(), nt, T, nc, TH_ddkappa, TH_dddkappa);
  integrate_spline_col(tau_table.data(), nt, T, nc, TH_ddkappa, TH_dddkappa, TH_tau_d);
  spline_col(tau_table.data(), nt, T, nc, TH_dkappa, TH_dddkappa);           // kappa''' (as the spline's second derivative of kappa')
  derive_spline_col(tau_table.data(), nt, T, nc, TH_dkappa, TH_dddkappa, TH_ddkappa);   // kappa''
  integrate_spline_col(tau_table.data(), nt, T, nc, TH_dkappa, TH_dddkappa, TH_g);       // -kappa, temporarily in column g
  for (int i = nt - 1; i >= 0; i--) {   // visibility and its derivatives, th.cpp:745-790
    double* r = T + (size_t)i * nc;
    const double g = r[TH_dkappa] * exp(r[TH_g]);
    r[TH_exp_m_kappa] = exp(r[TH_g]);
    r[TH_dg] = (r[TH_ddkappa] + r[TH_dkappa] * r[TH_dkappa]) * exp(r[TH_g]);
    r[TH_ddg] = (r[TH_dddkappa] + r[TH_dkappa] * r[TH_ddkappa] * 3. + r[TH_dkappa] * r[TH_dkappa] * r[TH_dkappa]) * exp(r[TH_g]);
    r[TH_g] = g;
    if (r[TH_dkappa] == 0.) return bail(fail_msg(CPT_ERR_RUNTIME, "variation rate diverges"));
    r[TH_rate] = sqrt(pow(r[TH_dkappa], 2) + pow(r[TH_ddkappa] / r[TH_dkappa], 2) + fabs(r[TH_dddkappa] / r[TH_dkappa]));
  }
  smooth_col(T, nc, nt, TH_rate, tp.thermo_rate_smoothing_radius);
  spline_table_lines(zt, nt, T, nc, out->d2thermodynamics_dz2_table);
  // ---- recombination time and the scalars derived from it, th.cpp:1000-1060 ----
  int it = nt - 1;
  while (zt[it] > Z_REC_MAX) it--;
  if (T[(size_t)(it + 1) * nc + TH_g] > T[(size_t)it * nc + TH_g])
    return bail(fail_msg(CPT_ERR_RUNTIME, "found a recombination redshift greater or equal to the maximum value imposed in thermodynamics.h, z_rec_max=%g", Z_REC_MAX));
  while (T[(size_t)(it + 1) * nc + TH_g] < T[(size_t)it * nc + TH_g]) it--;
  const double g_max = T[(size_t)it * nc + TH_g];
  const int index_tau_max = it;
  out->z_rec = zt[it + 1] + 0.5 * (zt[it + 1] - zt[it]) * (T[(size_t)it * nc + TH_g] - 1. * T[(size_t)(it + 2) * nc + TH_g]) /
                                (T[(size_t)it * nc + TH_g] - 2. * T[(size_t)(it + 1) * nc + TH_g] + T[(size_t)(it + 2) * nc + TH_g]);
  if (out->z_rec + cp.smallest_allowed_variation >= Z_REC_MAX || out->z_rec - cp.smallest_allowed_variation <= Z_REC_MIN)
    return bail(fail_msg(CPT_ERR_RUNTIME, "recombination redshift %g outside [%g, %g]", out->z_rec, Z_REC_MIN, Z_REC_MAX));
  if ((rc = B.tau_of_z(out->z_rec, &out->tau_rec)) || (rc = B.at_tau(out->tau_rec))) return bail(rc);
  out->rs_rec = B.row[bg->index_bg_rs];
  const double da_rec = B.row[bg->index_bg_ang_distance];
  out->ra_rec = da_rec * (1. + out->z_rec) / cp.a_today;
  out->angular_rescaling = out->ra_rec / (bg->conformal_age - out->tau_rec);
  // free streaming time, th.cpp:1065-1078
  double tau;
  if ((rc = B.tau_of_z(zt[it], &tau))) return bail(rc);
  while ((1. / T[(size_t)it * nc + TH_dkappa] / tau < tp.radiation_streaming_trigger_tau_c_over_tau) && (it > 0)) {
    it--;
    if ((rc = B.tau_of_z(zt[it], &tau))) return bail(rc);
  }
  out->tau_free_streaming = tau;
  // z_star, z_d (th.cpp:1130-1175)
  it = 0;
  while ((T[(size_t)it * nc + TH_exp_m_kappa] > 1. / E_) && (it < nt)) it++;
  out->z_star = zt[it - 1] + (1. / E_ - T[(size_t)(it - 1) * nc + TH_exp_m_kappa]) / (T[(size_t)it * nc + TH_exp_m_kappa] - T[(size_t)(it - 1) * nc + TH_exp_m_kappa]) * (zt[it] - zt[it - 1]);
  it = 0;
  while ((T[(size_t)it * nc + TH_tau_d] < 1.) && (it < nt)) it++;
  out->z_d = zt[it - 1] + (1. - T[(size_t)(it - 1) * nc + TH_tau_d]) / (T[(size_t)it * nc + TH_tau_d] - T[(size_t)(it - 1) * nc + TH_tau_d]) * (zt[it] - zt[it - 1]);
  // visibility cut, th.cpp:1215-1222
  it = index_tau_max;
  while ((T[(size_t)it * nc + TH_g] > g_max * tp.neglect_CMB_sources_below_visibility) && (it > 0)) it--;
  if ((rc = B.tau_of_z(zt[it], &out->tau_cut))) return bail(rc);
  out->YHe = tp.YHe; out->n_e = re.Nnow; out->z_reionization = z_reionization; out->tau_reionization = tau_reionization;
  out->index_th_xe = TH_xe; out->index_th_dkappa = TH_dkappa; out->index_th_tau_d = TH_tau_d; out->index_th_ddkappa = TH_ddkappa;
  out->index_th_dddkappa = TH_dddkappa; out->index_th_exp_m_kappa = TH_exp_m_kappa; out->index_th_g = TH_g; out->index_th_dg = TH_dg;
  out->index_th_ddg = TH_ddg; out->index_th_Tb = TH_Tb; out->index_th_wb = TH_wb; out->index_th_cb2 = TH_cb2; out->index_th_rate = TH_rate;
  return CPT_OK;
}
}
