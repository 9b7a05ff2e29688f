// Host-side background and thermodynamics tables (include/cpt_host.h): what the hot path consumes, computed
// from the cosmological parameters instead of being handed over by the reference's BackgroundModule / ThermodynamicsModule.
//
// The MODEL is the reference's (it has to be: the tables are its data contract) -
//   background:      Friedmann equation for photons, baryons, cdm, massless and massive neutrinos, Lambda, curvature; conformal time,
//                    proper time, sound horizon and growth factor on a grid uniform in ln a (source/background_module.cpp:263-610,
//                    1326-1520, 1934-2064), the 21 (+ 4 per massive species) columns and their order;
//   thermodynamics:  RECFAST 1.5 (Seager, Sasselov & Scott 1999; Wong, Moss & Scott 2008) with the smoothed Saha <-> ODE hand-overs,
//                    the CAMB-like tanh reionization, its adaptive redshift sampling, the derived opacity / visibility columns
//                    (source/thermodynamics_module.cpp:293-1297, 2159-2320, 2668-2990, 3335-3975) -
// the NUMERICS are this project's own (cpt_numerics.hpp): every ODE goes through one embedded Dormand-Prince 5(4) integrator at
// tolerances (1e-10, 1e-9) far below the reference's (its variable-order NDF evolver and Cash-Karp stepper at 1e-2 are not
// restated here), every spline through one factorise-once tridiagonal solver.  The tables therefore agree with the reference's to
// its own integration error (3e-6 relative on tau, tests/test_host_cosmo.py), not bit for bit; the test infrastructure holds a
// bit-exact restatement of the reference's modules as the checker.
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "../../include/cpt_host.h"
#include "cpt_numerics.hpp"

namespace cpt_host {
int fail_msg(int code, const char* fmt, ...);   // cpt_grids.cpp

// second derivatives of `ny` row-major columns (used by the grid builders as well)
void spline_table_lines(const double* x, int n, const double* y, int ny, double* ddy) {
  if (n < 3) { for (size_t i = 0; i < (size_t)n * ny; i++) ddy[i] = 0.; return; }
  cpt_num::ClampedSpline(x, n).moments(y, ny, ny, ddy);
}
int interpolate_spline(const double* x, int n, const double* y, const double* ddy, int ny, double v, double* out) {
  return cpt_num::spline_eval(x, n, y, ddy, ny, ny, v, out) ? 0 : 1;
}

namespace {
constexpr double kGyrOverMpc = 3.06601394e2;

// ---- column layout of the background table: the reference's order for the species this library knows (background_indices) ----
struct BgCols {
  int a, H, Hp, rho_g, rho_b, rho_cdm, n_ncdm, rho_ncdm, p_ncdm, pp_ncdm, rho_lambda, rho_ur, rho_tot, p_tot, p_tot_prime, Omega_r, rho_crit,
      Omega_m, conf_distance, ang_distance, lum_distance, time, rs, D, f, size;
  explicit BgCols(const cpt_cosmo_params& c) {
    int next = 0;
    auto take = [&](bool present, int count = 1) { const int at = present ? next : -1; if (present) next += count; return at; };
    a = take(true); H = take(true); Hp = take(true); rho_g = take(true); rho_b = take(true);
    rho_cdm = take(c.has_cdm);
    n_ncdm = take(c.has_ncdm, c.N_ncdm); rho_ncdm = take(c.has_ncdm, c.N_ncdm); p_ncdm = take(c.has_ncdm, c.N_ncdm); pp_ncdm = take(c.has_ncdm, c.N_ncdm);
    rho_lambda = take(c.has_lambda); rho_ur = take(c.has_ur);
    rho_tot = take(true); p_tot = take(true); p_tot_prime = take(true); Omega_r = take(true); rho_crit = take(true); Omega_m = take(true);
    conf_distance = take(true); ang_distance = take(true); lum_distance = take(true); time = take(true); rs = take(true); D = take(true); f = take(true);
    size = next;
  }
};

// momentum integrals of one non-cold species at scale factor a / a_today = 1 / (1 + z): number density, energy density, pressure and
// the "pseudo-pressure" integral of q^6 / eps^3 (the reference's quadrature: nodes q, weights w, eps = sqrt(q^2 + (M a)^2))
struct NcdmMoments { double number, rho, p, pseudo_p; };
NcdmMoments ncdm_moments(const cpt_cosmo_params& c, int species, double one_plus_z) {
  const double Ma = c.M_ncdm[species] / one_plus_z, norm = c.factor_ncdm[species] * one_plus_z * one_plus_z * one_plus_z * one_plus_z;
  NcdmMoments m{0., 0., 0., 0.};
  for (int i = 0; i < c.q_size_ncdm_bg[species]; i++) {
    const double q = c.q_ncdm_bg[species][i], q2 = q * q, wq2 = c.w_ncdm_bg[species][i] * q2, eps = std::sqrt(q2 + Ma * Ma), r = q2 / eps;
    m.number += wq2;
    m.rho += wq2 * eps;
    m.p += wq2 * r / 3.;
    m.pseudo_p += c.w_ncdm_bg[species][i] * r * r * r / 3.;
  }
  m.number *= norm / one_plus_z; m.rho *= norm; m.p *= norm; m.pseudo_p *= norm;
  return m;
}

// Everything that is a function of the scale factor alone.  The energy budget is accumulated species by species:
// rho, p, dp/dln a, and the split of rho into a relativistic and a non-relativistic part.
struct Budget {
  double rho = 0., p = 0., dp_dlna = 0., rho_rel = 0., rho_nr = 0.;
  void radiation(double r) { rho += r; p += r / 3.; dp_dlna -= 4. / 3. * r; rho_rel += r; }
  void dust(double r) { rho += r; rho_nr += r; }
};
int fill_background_row(const cpt_cosmo_params& c, const BgCols& col, double a, bool with_fractions, double* row) {
  if (!(a > 0.)) return fail_msg(CPT_ERR_INVALID, "a = %e instead of strictly positive", a / c.a_today);
  const double x = a / c.a_today, inv3 = 1. / (x * x * x), H0sq = c.H0 * c.H0;
  Budget all;
  row[col.a] = a;
  row[col.rho_g] = c.Omega0_g * H0sq * inv3 / x; all.radiation(row[col.rho_g]);
  row[col.rho_b] = c.Omega0_b * H0sq * inv3; all.dust(row[col.rho_b]);
  if (c.has_cdm) { row[col.rho_cdm] = c.Omega0_cdm * H0sq * inv3; all.dust(row[col.rho_cdm]); }
  if (c.has_ncdm)
    for (int s = 0; s < c.N_ncdm; s++) {
      const NcdmMoments m = ncdm_moments(c, s, 1. / x);
      row[col.n_ncdm + s] = m.number; row[col.rho_ncdm + s] = m.rho; row[col.p_ncdm + s] = m.p; row[col.pp_ncdm + s] = m.pseudo_p;
      all.rho += m.rho; all.p += m.p; all.dp_dlna += m.pseudo_p - 5. * m.p;
      all.rho_rel += 3. * m.p; all.rho_nr += m.rho - 3. * m.p;          // the reference's split of a semi-relativistic species
    }
  if (c.has_lambda) { row[col.rho_lambda] = c.Omega0_lambda * H0sq; all.rho += row[col.rho_lambda]; all.p -= row[col.rho_lambda]; }
  if (c.has_ur) { row[col.rho_ur] = c.Omega0_ur * H0sq * inv3 / x; all.radiation(row[col.rho_ur]); }
  const double curvature = c.K / (a * a), rho_crit = all.rho - curvature;
  if (!(rho_crit > 0.)) return fail_msg(CPT_ERR_INVALID, "rho_crit = %e instead of strictly positive", rho_crit);
  row[col.H] = std::sqrt(rho_crit);
  row[col.Hp] = -1.5 * (all.rho + all.p) * a + c.K / a;
  row[col.rho_tot] = all.rho; row[col.p_tot] = all.p; row[col.p_tot_prime] = a * row[col.H] * all.dp_dlna;
  row[col.Omega_r] = all.rho_rel / rho_crit;
  if (with_fractions) { row[col.rho_crit] = rho_crit; row[col.Omega_m] = all.rho_nr / rho_crit; }
  return CPT_OK;
}
}  // namespace
}  // namespace cpt_host

using namespace cpt_host;

extern "C" {

void cpt_host_cosmo_defaults(cpt_cosmo_params* p) {
  p->a_ini_over_a_today_default = 1.e-14; p->back_integration_stepsize = 7.e-3; p->tol_initial_Omega_r = 1.e-4;
  p->smallest_allowed_variation = 2.220446049250313e-16;   // DBL_EPSILON
  p->tol_ncdm_initial_w = 1.e-3;
}

void cpt_host_background_free(cpt_background* bg) {
  if (!bg) return;
  free(bg->tau_table); free(bg->z_table); free(bg->d2tau_dz2_table); free(bg->background_table); free(bg->d2background_dtau2_table);
  memset(bg, 0, sizeof(*bg));
}

int cpt_host_background(const cpt_cosmo_params* pp, cpt_background* out) {
  if (!pp || !out) return fail_msg(CPT_ERR_INVALID, "null argument");
  const cpt_cosmo_params& c = *pp;
  memset(out, 0, sizeof(*out));
  if (c.has_fld || c.has_scf || c.has_dcdm || c.has_dr || c.has_idr || c.has_idm_dr)
    return fail_msg(CPT_ERR_UNSUPPORTED, "host background: only photons, baryons, cdm, massless and massive neutrinos, Lambda and curvature");
  if (c.has_ncdm) {
    if (c.N_ncdm < 1 || c.N_ncdm > CPT_MAX_NCDM) return fail_msg(CPT_ERR_UNSUPPORTED, "host background: between 1 and %d non-cold species", CPT_MAX_NCDM);
    for (int s = 0; s < c.N_ncdm; s++)
      if (c.q_size_ncdm_bg[s] < 1 || !c.q_ncdm_bg[s] || !c.w_ncdm_bg[s])
        return fail_msg(CPT_ERR_INVALID, "host background: the momentum sampling of ncdm species %d is missing", s);
  }
  if (c.a_today <= 0) return fail_msg(CPT_ERR_INVALID, "input a_today = %e instead of strictly positive", c.a_today);
  const BgCols col(c);
  std::vector<double> work(col.size, 0.);

  // ---- where to start: deep in radiation domination, and early enough for every non-cold species to be ultra-relativistic ----
  double a_start = c.a_ini_over_a_today_default * c.a_today;
  if (c.has_ncdm) {
    auto all_relativistic = [&](double a) {
      for (int s = 0; s < c.N_ncdm; s++) {
        const NcdmMoments m = ncdm_moments(c, s, c.a_today / a);
        if (std::fabs(m.p / m.rho - 1. / 3.) > c.tol_ncdm_initial_w) return false;
      }
      return true;
    };
    int tries = 0;
    while (!all_relativistic(a_start)) {
      a_start *= 0.1;
      if (++tries == 10000) return fail_msg(CPT_ERR_RUNTIME, "Search for initial scale factor a such that all ncdm species are relativistic failed.");
    }
  }
  int rc = fill_background_row(c, col, a_start, false, work.data());
  if (rc) return rc;
  if (std::fabs(work[col.Omega_r] - 1.) > c.tol_initial_Omega_r)
    return fail_msg(CPT_ERR_INVALID, "Omega_r = %e, not close enough to 1. Decrease a_ini_over_a_today_default in order to start from radiation domination.", work[col.Omega_r]);
  if (!(work[col.H] > 0.)) return fail_msg(CPT_ERR_INVALID, "H = %e instead of strictly positive", work[col.H]);

  // ---- grid uniform in ln a ----
  const double lna0 = std::log(a_start), lna1 = std::log(c.a_today);
  const int n = (int)((lna1 - lna0) / c.back_integration_stepsize);
  if (n < 3) return fail_msg(CPT_ERR_INVALID, "background table too short");
  out->bt_size = n; out->bg_size = col.size;
  out->tau_table = (double*)malloc(sizeof(double) * n); out->z_table = (double*)malloc(sizeof(double) * n);
  out->d2tau_dz2_table = (double*)malloc(sizeof(double) * n);
  out->background_table = (double*)calloc((size_t)n * col.size, sizeof(double));
  out->d2background_dtau2_table = (double*)calloc((size_t)n * col.size, sizeof(double));
  if (!out->tau_table || !out->z_table || !out->d2tau_dz2_table || !out->background_table || !out->d2background_dtau2_table) {
    cpt_host_background_free(out);
    return fail_msg(CPT_ERR_RUNTIME, "could not allocate the background table");
  }

  // ---- the five integrated quantities as functions of ln a:  d/dln a = (1 / aH) d/dtau
  //      tau' = 1,  t' = a,  r_s' = c_s sqrt(1 - K r_s^2),  D' = D_tau,  D_tau' = -aH D_tau + 3/2 a^2 rho_M D     (primes: d/dtau)
  enum { kTau, kTime, kRs, kD, kDtau, kNeq };
  bool model_failed = false;
  auto derivatives = [&](double lna, const double* y, double* dy) {
    const double a = std::exp(lna);
    if (fill_background_row(c, col, a, false, work.data())) { model_failed = true; for (int i = 0; i < kNeq; i++) dy[i] = 0.; return; }
    const double aH = a * work[col.H], rho_matter = work[col.rho_b] + (c.has_cdm ? work[col.rho_cdm] : 0.);
    const double sound_speed = 1. / std::sqrt(3. * (1. + 0.75 * work[col.rho_b] / work[col.rho_g]));
    dy[kTau] = 1. / aH;
    dy[kTime] = a / aH;
    dy[kRs] = sound_speed * std::sqrt(std::max(0., 1. - c.K * y[kRs] * y[kRs])) / aH;
    dy[kD] = y[kDtau] / aH;
    dy[kDtau] = -y[kDtau] + 1.5 * a * a * rho_matter * y[kD] / aH;
  };
  cpt_num::Dopri5<kNeq> ode;
  ode.rtol = 1e-10;
  ode.x = lna0;
  // radiation-dominated start: tau = 1 / aH, t = 1 / 2H, r_s = tau / sqrt 3, growing mode D = a
  ode.y[kTau] = 1. / (a_start * work[col.H]);
  ode.y[kTime] = 0.5 / work[col.H];
  ode.y[kRs] = ode.y[kTau] / std::sqrt(3.);
  ode.y[kD] = a_start;
  ode.y[kDtau] = 2. * a_start * work[col.H];
  for (int i = 0; i < n; i++) {
    const double lna = lna0 + i * (lna1 - lna0) / (n - 1);
    ode.forget_step_size();   // one step per node: the nodes are closer than the tolerance needs, and a step just short of one costs a second
    if (!ode.advance(derivatives, lna) || model_failed) { cpt_host_background_free(out); return fail_msg(CPT_ERR_RUNTIME, "background integration failed at ln a = %g", lna); }
    const double a = std::exp(lna);
    double* row = out->background_table + (size_t)i * col.size;
    if ((rc = fill_background_row(c, col, a, true, row))) { cpt_host_background_free(out); return rc; }
    out->z_table[i] = std::max(0., c.a_today / a - 1.);
    out->tau_table[i] = ode.y[kTau];
    row[col.time] = ode.y[kTime]; row[col.rs] = ode.y[kRs]; row[col.D] = ode.y[kD];
    row[col.f] = ode.y[kDtau] / (ode.y[kD] * a * row[col.H]);
  }
  out->age = ode.y[kTime] / kGyrOverMpc;
  out->conformal_age = ode.y[kTau];
  // distances from the conformal distance to today, growth factor normalised today
  const double growth_today = ode.y[kD], sqrtK = std::sqrt(std::fabs(c.K));
  for (int i = 0; i < n; i++) {
    double* row = out->background_table + (size_t)i * col.size;
    const double chi = out->conformal_age - out->tau_table[i];
    const double radius = (c.sgnK > 0) ? std::sin(sqrtK * chi) / sqrtK : (c.sgnK < 0) ? std::sinh(sqrtK * chi) / sqrtK : chi;
    row[col.conf_distance] = chi;
    row[col.ang_distance] = c.a_today * radius / (1. + out->z_table[i]);
    row[col.lum_distance] = c.a_today * radius * (1. + out->z_table[i]);
    row[col.D] /= growth_today;
  }
  spline_table_lines(out->z_table, n, out->tau_table, 1, out->d2tau_dz2_table);
  spline_table_lines(out->tau_table, n, out->background_table, col.size, out->d2background_dtau2_table);
  const double* first = out->background_table;
  const double* last = out->background_table + (size_t)(n - 1) * col.size;
  out->Neff = (first[col.Omega_r] * first[col.rho_crit] - first[col.rho_g]) / (7. / 8. * std::pow(4. / 11., 4. / 3.) * first[col.rho_g]);
  out->Omega0_m = last[col.Omega_m]; out->Omega0_r = last[col.Omega_r]; out->Omega0_de = 1. - (out->Omega0_m + out->Omega0_r + c.Omega0_k);
  out->index_bg_a = col.a; out->index_bg_H = col.H; out->index_bg_H_prime = col.Hp; out->index_bg_rho_g = col.rho_g; out->index_bg_rho_b = col.rho_b;
  out->index_bg_rho_cdm = col.rho_cdm; out->index_bg_rho_lambda = col.rho_lambda; out->index_bg_rho_ur = col.rho_ur; out->index_bg_rho_tot = col.rho_tot;
  out->index_bg_p_tot = col.p_tot; out->index_bg_p_tot_prime = col.p_tot_prime; out->index_bg_Omega_r = col.Omega_r; out->index_bg_rho_crit = col.rho_crit;
  out->index_bg_Omega_m = col.Omega_m; out->index_bg_conf_distance = col.conf_distance; out->index_bg_ang_distance = col.ang_distance;
  out->index_bg_number_ncdm1 = col.n_ncdm; out->index_bg_rho_ncdm1 = col.rho_ncdm; out->index_bg_p_ncdm1 = col.p_ncdm; out->index_bg_pseudo_p_ncdm1 = col.pp_ncdm;
  out->index_bg_lum_distance = col.lum_distance; out->index_bg_time = col.time; out->index_bg_rs = col.rs; out->index_bg_D = col.D; out->index_bg_f = col.f;
  return CPT_OK;
}

// the same from the two columns alone (a table handed over without its second derivatives: they are rebuilt as background_solve does,
// array_spline_table_lines with estimated end derivatives, source/background_module.cpp:1478-1487)
int cpt_host_tau_of_z_from_table(const double* z_table, const double* tau_table, int n, double z, double* tau) {
  if (!z_table || !tau_table || !tau || n < 4) return fail_msg(CPT_ERR_INVALID, "bad arguments to cpt_host_tau_of_z_from_table");
  if (z < z_table[n - 1] || z > z_table[0]) return fail_msg(CPT_ERR_INVALID, "out of range: z=%e outside [%e, %e]", z, z_table[n - 1], z_table[0]);
  std::vector<double> d2(n);
  spline_table_lines(z_table, n, tau_table, 1, d2.data());
  if (interpolate_spline(z_table, n, tau_table, d2.data(), 1, z, tau)) return fail_msg(CPT_ERR_INVALID, "tau(z): interpolation failed");
  return CPT_OK;
}

int cpt_host_background_tau_of_z(const cpt_background* bg, double z, double* tau) {
  if (z < bg->z_table[bg->bt_size - 1] || z > bg->z_table[0]) return fail_msg(CPT_ERR_INVALID, "out of range: z=%e outside [%e, %e]", z, bg->z_table[bg->bt_size - 1], bg->z_table[0]);
  if (interpolate_spline(bg->z_table, bg->bt_size, bg->tau_table, bg->d2tau_dz2_table, 1, z, tau)) return fail_msg(CPT_ERR_INVALID, "tau(z): interpolation failed");
  return CPT_OK;
}
}

// =====================================================================================================================
// Thermodynamics
// =====================================================================================================================
namespace cpt_host {
namespace {
// physical constants (SI), the values the reference's tables are built with
constexpr double kC = 2.99792458e8, kG = 6.67428e-11, kBoltz = 1.3806504e-23, kPlanck = 6.62606896e-34, kMpc = 3.085677581282e22,
                 kMe = 9.10938215e-31, kMH = 1.673575e-27, kHe4OverH = 3.9715, kSigmaT = 6.6524616e-29, kPi = 3.1415926535897932384626433832795,
                 kEuler = 2.7182818284590452353602874713526624977572470936999595749669676277;
constexpr double kZRecMax = 2000., kZRecMin = 500., kYHeMax = 0.5, kYHeMin = 0.01;
constexpr double kReferenceIntegralRule = +1.;   // see cpt_num::spline_cumulative_integral

// the host background table seen as functions of redshift
struct Background {
  const cpt_background& bg;
  std::vector<double> row;
  explicit Background(const cpt_background& b) : bg(b), row(b.bg_size) {}
  int at_z_bracket = 0, at_tau_bracket = 0;   // where the last look-ups fell: the callers walk along the tables
  int tau_of_z(double z, double* tau) {
    if (!cpt_num::spline_eval(bg.z_table, bg.bt_size, bg.tau_table, bg.d2tau_dz2_table, 1, 1, z, tau, &at_z_bracket))
      return fail_msg(CPT_ERR_INVALID, "out of range: z=%e outside [%e, %e]", z, bg.z_table[bg.bt_size - 1], bg.z_table[0]);
    return CPT_OK;
  }
  int at_tau(double tau) {
    if (!cpt_num::spline_eval(bg.tau_table, bg.bt_size, bg.background_table, bg.d2background_dtau2_table, bg.bg_size, bg.bg_size, tau, row.data(), &at_tau_bracket))
      return fail_msg(CPT_ERR_INVALID, "background_at_tau: tau=%e out of range", tau);
    return CPT_OK;
  }
  int at_z(double z) { double tau; const int rc = tau_of_z(z, &tau); return rc ? rc : at_tau(tau); }
  // only the columns between two named ones (the recombination equations read H and H', the drag depth rho_g and rho_b: a handful
  // of neighbouring columns out of two dozen), into the same row
  int columns_at_tau(double tau, int col_a, int col_b) {
    const int first = std::min(col_a, col_b), span = std::abs(col_a - col_b) + 1;
    if (!cpt_num::spline_eval(bg.tau_table, bg.bt_size, bg.background_table + first, bg.d2background_dtau2_table + first, span, bg.bg_size, tau, row.data() + first, &at_tau_bracket))
      return fail_msg(CPT_ERR_INVALID, "background_at_tau: tau=%e out of range", tau);
    return CPT_OK;
  }
  int expansion_at_z(double z) { double tau; const int rc = tau_of_z(z, &tau); return rc ? rc : columns_at_tau(tau, bg.index_bg_H, bg.index_bg_H_prime); }
  double H() const { return row[bg.index_bg_H]; }
  double Hp() const { return row[bg.index_bg_H_prime]; }
  double rho_g() const { return row[bg.index_bg_rho_g]; }
  double rho_b() const { return row[bg.index_bg_rho_b]; }
};

// ---- RECFAST: the three-level-atom rate equations for x_H = n_p / n_H, x_He = n_HeII / n_He and the matter temperature ----
// Spectroscopic data (wavenumbers in 1/m, Einstein coefficients in 1/s, cross sections in m^2)
namespace atom {
constexpr double H_ion = 1.096787737e7, H_lya = 8.225916453e6, H_2s_rate = 8.2245809;
constexpr double He1_ion = 1.98310772e7, He2_ion = 4.389088863e7, He_2s = 1.66277434e7, He_2p = 1.71134891e7, He_2s_rate = 51.3;
constexpr double He_2p_A = 1.798287e9, He_2pt_A = 177.58, He_2pt = 1.690871466e7, He_2st = 1.5985597526e7, He_2st_ion = 3.8454693845e6;
constexpr double He_2ps_sigma = 1.436289e-22, He_2pt_sigma = 1.484872e-22;
// case-B hydrogen recombination fit of Pequignot et al.; Verner & Ferland fits for singlet and triplet helium
constexpr double ppb_a = 4.309, ppb_b = -0.6166, ppb_c = 0.6703, ppb_d = 0.5300, vf_b = 0.711, trip_b = 0.761;
}  // namespace atom

inline double smooth_step_cubic(double s) { return 0.5 - 0.75 * s * (s * s / 3. - 1.); }     // 0 -> 1 over s in [-1, 1]
inline double smooth_step_quadratic(double s) { return 6. * s * s * (0.5 - s / 3.); }       // 0 -> 1 over s in [0, 1]

class Recfast {
 public:
  Recfast(const cpt_cosmo_params& cp, const cpt_thermo_params& tp, Background& B) : cp_(cp), tp_(tp), B_(B) {
    H0_ = cp.H0 * kC / kMpc;
    T0_ = cp.T_cmb;
    fudge_H_ = tp.recfast_fudge_H + (tp.recfast_Hswitch ? tp.recfast_delta_fudge_H : 0.);
    fHe_ = tp.YHe / (kHe4OverH * (1. - tp.YHe));                      // n_He / n_H
    nH0_ = 3. * H0_ * H0_ * cp.Omega0_b / (8. * kPi * kG * kMH / (1. - tp.YHe));   // hydrogen number density today
    const double hc_over_k = kPlanck * kC / kBoltz;
    T_ion_H_n2_ = hc_over_k * (atom::H_ion - atom::H_lya);            // binding energy of the n = 2 level / k
    T_ion_He_n2_ = hc_over_k * (atom::He1_ion - atom::He_2s);
    T_ion_H_ = hc_over_k * atom::H_ion; T_ion_He1_ = hc_over_k * atom::He1_ion; T_ion_He2_ = hc_over_k * atom::He2_ion;
    T_lya_ = hc_over_k * atom::H_lya; T_He_2s_ = hc_over_k * atom::He_2s;
    T_He_2p2s_ = hc_over_k * (atom::He_2p - atom::He_2s);
    saha_prefactor_ = 2. * kPi * (kMe / kPlanck) * (kBoltz / kPlanck);           // (2 pi m_e k / h^2)
    lya_escape_ = 1. / (8. * kPi * atom::H_lya * atom::H_lya * atom::H_lya);     // lambda^3 / 8 pi
    He_escape_ = 1. / (8. * kPi * atom::He_2p * atom::He_2p * atom::He_2p);
    compton_ = (8. / 3.) * (kSigmaT / (kMe * kC)) * (8. * std::pow(kPi, 5) * std::pow(kBoltz, 4) / 15. / std::pow(kPlanck, 3) / std::pow(kC, 3));
  }
  double fHe() const { return fHe_; }
  double nH0() const { return nH0_; }
  double T0() const { return T0_; }
  bool failed() const { return failed_; }

  // Saha equilibria: x_e for doubly -> singly ionised helium, and the ionised fractions of the last electron of He and of H
  double saha_ratio(double z, double T_ion, double weight) const {
    const double T = T0_ * (1. + z);
    return weight * std::exp(1.5 * std::log(saha_prefactor_ * T0_ / (1. + z)) - T_ion / T) / nH0_;
  }
  double xe_saha_He2(double z) const { const double r = saha_ratio(z, T_ion_He2_, 1.), b = r - 1. - fHe_; return 0.5 * (std::sqrt(b * b + 4. * (1. + 2. * fHe_) * r) - b); }
  double xe_saha_He1(double z) const { const double r = saha_ratio(z, T_ion_He1_, 4.), b = r - 1.; return 0.5 * (std::sqrt(b * b + 4. * (1. + fHe_) * r) - b); }
  double xH_saha(double z) const { const double r = saha_ratio(z, T_ion_H_, 1.); return 0.5 * (std::sqrt(r * r + 4. * r) - r); }

  // dy/dz for y = (x_H, x_He, T_matter)
  void operator()(double z, const double* y, double* dy) {
    const double xH = y[0], xHe = y[1], Tm = y[2], xe = xH + fHe_ * xHe;
    const double opz = 1. + z, nH = nH0_ * opz * opz * opz, nHe = fHe_ * nH, Tr = T0_ * opz;
    if (B_.expansion_at_z(z)) { failed_ = true; dy[0] = dy[1] = dy[2] = 0.; return; }
    const double Hz = B_.H() * kC / kMpc, dt_dz_inv = Hz * opz;   // |dz/dt| = H (1 + z)
    // (powers as x sqrt x and exp(b ln x) with the logarithms shared: the equations are evaluated some 5e4 times per history and
    // general pow() calls were most of their cost)
    const double sT = saha_prefactor_ * Tm, thermal = sT * std::sqrt(sT);
    // --- hydrogen: case-B recombination, photoionisation from n = 2, Peebles factor with the Lyman-alpha escape correction; not
    //     evaluated while hydrogen is held at its Saha value (two thirds of the nodes: only dT/dz is wanted there)
    const bool hydrogen_evolves = !(xH > tp_.recfast_x_H0_trigger);
    double alpha_H = 0., beta_H = 0., K_H = 0.;
    if (hydrogen_evolves) {
      const double ln_t4 = std::log(Tm / 1.e4);
      alpha_H = 1.e-19 * atom::ppb_a * std::exp(atom::ppb_b * ln_t4) / (1. + atom::ppb_c * std::exp(atom::ppb_d * ln_t4));
      beta_H = alpha_H * thermal * std::exp(-T_ion_H_n2_ / Tm);
      K_H = lya_escape_ / Hz;
      if (tp_.recfast_Hswitch) {
        const double lz = std::log(opz), g1 = (lz - tp_.recfast_zGauss1) / tp_.recfast_wGauss1, g2 = (lz - tp_.recfast_zGauss2) / tp_.recfast_wGauss2;
        K_H *= 1. + tp_.recfast_AGauss1 * std::exp(-g1 * g1) + tp_.recfast_AGauss2 * std::exp(-g2 * g2);
      }
    }
    // --- helium singlets and triplets (Verner-Ferland fits)
    const double s0 = std::sqrt(Tm / std::pow(10., 0.477121)), s1 = std::sqrt(Tm / std::pow(10., 5.114));
    const double ln_1s0 = std::log1p(s0), ln_1s1 = std::log1p(s1);
    auto vf = [&](double amp, double b) { return amp / (s0 * std::exp((1. - b) * ln_1s0 + (1. + b) * ln_1s1)); };
    const double alpha_He = vf(std::pow(10., -16.744), atom::vf_b), beta_He = 4. * alpha_He * thermal * std::exp(-T_ion_He_n2_ / Tm);
    const double alpha_He_t = vf(std::pow(10., -16.306), atom::trip_b);
    const double beta_He_t = alpha_He_t * std::exp(-kPlanck * kC * atom::He_2st_ion / (kBoltz * Tm)) * thermal * 4. / 3.;
    const int he_mode = (xHe < 5.e-9 || xHe > tp_.recfast_x_He0_trigger2) ? 0 : tp_.recfast_Heswitch;
    double K_He = He_escape_ / Hz, triplet_branch = 0.;
    if (he_mode != 0) {
      const double neutral_He = nHe * (1. - xHe);
      const double doppler_width = std::sqrt(2. * kBoltz * Tm / (kMH * kHe4OverH * kC * kC));
      // Sobolev escape probability of a line with optical depth tau
      auto escape = [](double tau) { return (1. - std::exp(-tau)) / tau; };
      // continuum opacity of neutral hydrogen inside a helium line (Kholupenko et al.): extra escape channel A / (1 + p gamma^q)
      auto hydrogen_channel = [&](double A, double wavenumber, double sigma, double p, double q, double norm) {
        const double width = kC * wavenumber * doppler_width;
        const double gamma = norm * A * fHe_ * (1. - xHe) * kC * kC / (std::sqrt(kPi) * sigma * 8. * kPi * width * (1. - xH)) / std::pow(kC * wavenumber, 2);
        return A / (1. + p * std::pow(gamma, q));
      };
      const double tau_s = atom::He_2p_A * He_escape_ * 3. * neutral_He / Hz, p_s = escape(tau_s);
      double A_eff = atom::He_2p_A * p_s;
      if ((he_mode == 2 || he_mode >= 5) && xH < 0.9999999) A_eff += hydrogen_channel(atom::He_2p_A, atom::He_2p, atom::He_2ps_sigma, 0.36, tp_.recfast_fudge_He, 3.);
      K_He = 1. / (A_eff * 3. * neutral_He);
      if (he_mode >= 3) {
        const double tau_t = atom::He_2pt_A * neutral_He * 3. / (8. * kPi * Hz * std::pow(atom::He_2pt, 3)), p_t = escape(tau_t);
        const double T_2p2s_t = kPlanck * kC * (atom::He_2pt - atom::He_2st) / kBoltz;
        double A_t = atom::He_2pt_A * p_t;
        if (!(he_mode == 3 || he_mode == 5 || xH >= 0.99999)) A_t += hydrogen_channel(atom::He_2pt_A, atom::He_2pt, atom::He_2pt_sigma, 0.66, 0.9, 3.) / 3.;
        const double out_rate = A_t * std::exp(-T_2p2s_t / Tm);
        triplet_branch = out_rate / (beta_He_t + out_rate);
      }
    }
    // --- the three equations
    if (!hydrogen_evolves) dy[0] = 0.;
    else {
      const double n1s = nH * (1. - xH);
      const double peebles = (xH < tp_.recfast_x_H0_trigger2)
                                 ? (1. + K_H * atom::H_2s_rate * n1s) / (1. / fudge_H_ + K_H * atom::H_2s_rate * n1s / fudge_H_ + K_H * beta_H * n1s)
                                 : 1.;
      dy[0] = (xe * xH * nH * alpha_H - beta_H * (1. - xH) * std::exp(-T_lya_ / Tm)) * peebles / dt_dz_inv;
    }
    if (xHe < 1.e-15) dy[1] = 0.;
    else {
      const double n1s = nHe * (1. - xHe), boltz = std::exp(std::min(T_He_2p2s_ / Tm, 680.));
      dy[1] = (xe * xHe * nH * alpha_He - beta_He * (1. - xHe) * std::exp(-T_He_2s_ / Tm)) * (1. + K_He * atom::He_2s_rate * n1s * boltz) /
              (dt_dz_inv * (1. + K_He * (atom::He_2s_rate + beta_He) * n1s * boltz));
      if (he_mode >= 3)
        dy[1] += (xe * xHe * nH * alpha_He_t - (1. - xHe) * 3. * beta_He_t * std::exp(-kPlanck * kC * atom::He_2st / (kBoltz * Tm))) * triplet_branch / dt_dz_inv;
    }
    // matter temperature: tightly coupled to the radiation while Compton scattering is fast (first-order expansion in the
    // coupling time), the full Compton + adiabatic equation afterwards
    const double t_compton = (1. + xe + fHe_) / (compton_ * Tr * Tr * Tr * Tr * xe), t_hubble = 2. / (3. * H0_ * opz * std::sqrt(opz));
    if (t_compton < tp_.recfast_H_frac * t_hubble) {
      const double dlnH_dz = -B_.Hp() / B_.H() / cp_.a_today * kC / kMpc / Hz;
      const double lag = Hz * (1. + xe + fHe_) / (compton_ * Tr * Tr * Tr * xe);
      dy[2] = T0_ + lag * ((1. + fHe_) / (1. + fHe_ + xe)) * ((dy[0] + fHe_ * dy[1]) / xe) - lag * dlnH_dz + 3. * lag / opz;
    } else
      dy[2] = (Tm - Tr) / (t_compton * dt_dz_inv) + 2. * Tm / opz;
  }

 private:
  const cpt_cosmo_params& cp_;
  const cpt_thermo_params& tp_;
  Background& B_;
  bool failed_ = false;
  double H0_, T0_, fudge_H_, fHe_, nH0_, T_ion_H_n2_, T_ion_He_n2_, T_ion_H_, T_ion_He1_, T_ion_He2_, T_lya_, T_He_2s_, T_He_2p2s_, saha_prefactor_,
      lya_escape_, He_escape_, compton_;
};

// recombination history on the uniform redshift grid z_i = z_initial i / Nz, i = 0..Nz-1 (ascending z), columns below
enum { RE_Z = 0, RE_XE, RE_TB, RE_WB, RE_CB2, RE_DKAPPADTAU, RE_DKAPPADZ, RE_D3KAPPADZ3, RE_SIZE };

// Going down in redshift the ionisation state is taken from Saha equilibria for as long as they hold (He III -> He II -> He I, then
// hydrogen), each hand-over - between two equilibria, and from an equilibrium to the rate equations - blended over a finite
// window so that x_e(z) has no kinks; the matter temperature follows the radiation until the rate equations take over.
int recombination_history(const cpt_cosmo_params& cp, const cpt_thermo_params& tp, Background& B, Recfast& model, std::vector<double>& tab) {
  const int Nz = tp.recfast_Nz0;
  tab.assign((size_t)Nz * RE_SIZE, 0.);
  if (tp.recfast_Heswitch < 0 || tp.recfast_Heswitch > 6) return fail_msg(CPT_ERR_INVALID, "RECFAST error: unknown He fudging scheme");
  const double z_top = tp.recfast_z_initial, fHe = model.fHe();
  if (z_top < tp.recfast_z_He_3) return fail_msg(CPT_ERR_INVALID, "increase zinitial, otherwise should get initial conditions from recfast's get_init routine");
  cpt_num::Dopri5<3> ode;
  ode.rtol = 1e-9;
  ode.atol[0] = ode.atol[1] = 1e-14;
  double y[3] = {1., 1., model.T0() * (1. + z_top)};
  // One step of the integrator per node.  While the state is handed on untouched from node to node (everything by the rate
  // equations) the integrator keeps its last stage - the derivative at the node, which the table needs as well - as the first
  // stage of the next step; a caller that overwrites y (a Saha value) makes it start afresh.
  auto integrate_to = [&](double z_from, double z_to) -> int {
    const bool handed_on = ode.slope() && ode.x == z_from && ode.y[0] == y[0] && ode.y[1] == y[1] && ode.y[2] == y[2];
    if (!handed_on) {
      ode.x = z_from;
      for (int i = 0; i < 3; i++) ode.y[i] = y[i];
      ode.restart();
    } else ode.forget_step_size();
    if (!ode.advance(model, z_to) || model.failed()) return fail_msg(CPT_ERR_RUNTIME, "recfast: integration failed in [%g : %g]", z_from, z_to);
    for (int i = 0; i < 3; i++) y[i] = ode.y[i];
    return CPT_OK;
  };
  auto blend = [](double w, double fresh, double old) { return w * fresh + (1. - w) * old; };
  double xe = 1. + 2. * fHe, xH_saha = 0.;
  for (int i = 0; i < Nz; i++) {
    const double z_hi = z_top * (double)(Nz - i) / (double)Nz, z = z_top * (double)(Nz - i - 1) / (double)Nz;
    const double Tr = model.T0() * (1. + z);
    if (z > tp.recfast_z_He_1 + tp.recfast_delta_z_He_1) {                       // everything ionised
      xe = 1. + 2. * fHe;
      y[0] = 1.; y[1] = 1.; y[2] = Tr;
    } else if (z > tp.recfast_z_He_2 + tp.recfast_delta_z_He_2) {                // He III -> He II in equilibrium
      xe = model.xe_saha_He2(z);
      if (z > tp.recfast_z_He_1 - tp.recfast_delta_z_He_1)
        xe = blend(smooth_step_cubic((tp.recfast_z_He_1 - z) / tp.recfast_delta_z_He_1), xe, 1. + 2. * fHe);
      y[0] = 1.; y[1] = 1.; y[2] = Tr;
    } else if (z > tp.recfast_z_He_3 + tp.recfast_delta_z_He_3) {                // helium singly ionised
      xe = 1. + fHe;
      if (z > tp.recfast_z_He_2 - tp.recfast_delta_z_He_2)
        xe = blend(smooth_step_cubic((tp.recfast_z_He_2 - z) / tp.recfast_delta_z_He_2), xe, model.xe_saha_He2(z));
      y[0] = 1.; y[1] = 1.; y[2] = Tr;
    } else if (y[1] > tp.recfast_x_He0_trigger) {                                // He II -> He I in equilibrium
      const double x_saha = model.xe_saha_He1(z);
      xe = x_saha;
      if (z > tp.recfast_z_He_3 - tp.recfast_delta_z_He_3)
        xe = blend(smooth_step_cubic((tp.recfast_z_He_3 - z) / tp.recfast_delta_z_He_3), x_saha, 1. + fHe);
      y[0] = 1.; y[1] = (xe - 1.) / fHe; y[2] = Tr;
    } else if (y[0] > tp.recfast_x_H0_trigger) {                                 // helium by the rate equations, hydrogen still in equilibrium
      xH_saha = model.xH_saha(z);
      int rc = integrate_to(z_hi, z);
      if (rc) return rc;
      y[0] = xH_saha;
      xe = y[0] + fHe * y[1];
      if (tp.recfast_x_He0_trigger - y[1] < tp.recfast_x_He0_trigger_delta)
        xe = blend(smooth_step_quadratic((tp.recfast_x_He0_trigger - y[1]) / tp.recfast_x_He0_trigger_delta), xe, model.xe_saha_He1(z));
    } else {                                                                      // everything by the rate equations
      const bool near_switch_before = tp.recfast_x_H0_trigger - y[0] < tp.recfast_x_H0_trigger_delta;
      if (near_switch_before) xH_saha = model.xH_saha(z);
      int rc = integrate_to(z_hi, z);
      if (rc) return rc;
      if (tp.recfast_x_H0_trigger - y[0] < tp.recfast_x_H0_trigger_delta)
        xe = blend(smooth_step_quadratic((tp.recfast_x_H0_trigger - y[0]) / tp.recfast_x_H0_trigger_delta), y[0], xH_saha) + fHe * y[1];
      else xe = y[0] + fHe * y[1];
    }
    double* row = &tab[(size_t)(Nz - i - 1) * RE_SIZE];
    double dy[3];
    const double* kept = (ode.slope() && ode.x == z && ode.y[0] == y[0] && ode.y[1] == y[1] && ode.y[2] == y[2]) ? ode.slope() : nullptr;
    if (kept) for (int j = 0; j < 3; j++) dy[j] = kept[j];
    else model(z, y, dy);
    if (model.failed()) return fail_msg(CPT_ERR_RUNTIME, "recfast: background look-up failed at z=%e", z);
    row[RE_Z] = z; row[RE_XE] = xe; row[RE_TB] = y[2];
    row[RE_WB] = kBoltz / (kC * kC * kMH) * (1. + (1. / kHe4OverH - 1.) * tp.YHe + xe * (1. - tp.YHe)) * y[2];
    row[RE_CB2] = row[RE_WB] * (1. + (1. + z) * dy[2] / y[2] / 3.);
    row[RE_DKAPPADTAU] = (1. + z) * (1. + z) * model.nH0() * xe * kSigmaT * kMpc;
  }
  return CPT_OK;
}

// ---- reionization (CAMB-like): x_e(z) = tanh step in (1+z)^exponent for hydrogen + first helium electron, a second tanh for the
//      second helium electron ----
struct ReioModel {
  double xe_before, xe_after, z_reio, z_start, exponent, width, he_frac, he_z, he_width;
  double xe(double z) const {
    if (z > z_start) return xe_before;
    const double u = (std::pow(1. + z_reio, exponent) - std::pow(1. + z, exponent)) / (exponent * std::pow(1. + z_reio, exponent - 1.)) / width;
    return (xe_after - xe_before) * 0.5 * (std::tanh(u) + 1.) + xe_before + he_frac * 0.5 * (std::tanh((he_z - z) / he_width) + 1.);
  }
};
// x_e of the recombination table at z by linear interpolation between its nodes
int xe_from_recombination(const std::vector<double>& reco, int Nz, double z, double* xe) {
  if (z < reco[RE_Z] || z > reco[(size_t)(Nz - 1) * RE_SIZE + RE_Z]) return fail_msg(CPT_ERR_INVALID, "z=%e outside the recombination table", z);
  int hi = 1;
  while (z > reco[(size_t)hi * RE_SIZE + RE_Z]) hi++;
  const double z0 = reco[(size_t)(hi - 1) * RE_SIZE + RE_Z], z1 = reco[(size_t)hi * RE_SIZE + RE_Z], w = (z - z0) / (z1 - z0);
  *xe = (1. - w) * reco[(size_t)(hi - 1) * RE_SIZE + RE_XE] + w * reco[(size_t)hi * RE_SIZE + RE_XE];
  return CPT_OK;
}

// The reionization part of the table: redshifts chosen adaptively from z_start down to 0 so that the opacity changes by less than
// `reionization_sampling` (relative, in kappa'(z) and kappa'(tau) together) from one node to the next - this rule fixes the nodes
// of the final table, so it is the reference's; baryon temperature by explicit Euler steps on those nodes; optical depth from the
// spline integral of dkappa/dz.  Output rows in ascending z.
int reionization_history(const cpt_cosmo_params& cp, const cpt_thermo_params& tp, Background& B, double nH0, const std::vector<double>& reco, const ReioModel& R,
                         std::vector<double>& tab, int* n_rows, int* first_reco_row_kept, double* optical_depth) {
  const int Nz = tp.recfast_Nz0;
  int i0 = 0;
  while (reco[(size_t)i0 * RE_SIZE + RE_Z] < R.z_start)
    if (++i0 == Nz) return fail_msg(CPT_ERR_INVALID, "reionization_z_start_max = %e > largest redshift in thermodynamics table", tp.reionization_z_start_max);
  *first_reco_row_kept = i0;
  auto opacity = [&](double z, double xe) { return (1. + z) * (1. + z) * nH0 * xe * kSigmaT * kMpc; };   // kappa' = dkappa/dtau
  struct Node { double z, xe, dk_dtau, dk_dz; };
  std::vector<Node> nodes;   // descending z
  int rc;
  double z = reco[(size_t)i0 * RE_SIZE + RE_Z];
  if ((rc = B.at_z(z))) return rc;
  if (B.H() == 0.) return fail_msg(CPT_ERR_INVALID, "stop to avoid division by zero");
  nodes.push_back({z, R.xe(z), opacity(z, R.xe(z)), opacity(z, R.xe(z)) / B.H()});
  const double dz_max = reco[(size_t)i0 * RE_SIZE + RE_Z] - reco[(size_t)(i0 - 1) * RE_SIZE + RE_Z];
  double dz = dz_max;
  while (z > 0.) {
    if (dz < cp.smallest_allowed_variation) return fail_msg(CPT_ERR_RUNTIME, "stuck in the loop for reionization sampling, as if you were trying to impose a discontinuous evolution for xe(z)");
    const double z_try = std::max(0., z - dz), xe_try = R.xe(z_try);
    if ((rc = B.at_z(z_try))) return rc;
    if (B.H() == 0.) return fail_msg(CPT_ERR_INVALID, "stop to avoid division by zero");
    const Node& prev = nodes.back();
    if (prev.dk_dz == 0. || prev.dk_dtau == 0.) return fail_msg(CPT_ERR_INVALID, "stop to avoid division by zero");
    const double dk_dtau = opacity(z_try, xe_try), dk_dz = dk_dtau / B.H();
    const double change = std::fabs((dk_dz - prev.dk_dz) / prev.dk_dz) + std::fabs((dk_dtau - prev.dk_dtau) / prev.dk_dtau);
    if (change < tp.reionization_sampling) {
      z = z_try;
      nodes.push_back({z, xe_try, dk_dz * B.H(), dk_dz});
      dz = std::min(std::min(0.9 * (tp.reionization_sampling / change), 5.) * dz, dz_max);
    } else dz *= 0.9 * (tp.reionization_sampling / change);
  }
  const int n = (int)nodes.size();
  *n_rows = n;
  tab.assign((size_t)n * RE_SIZE, 0.);
  for (int j = 0; j < n; j++) {
    const Node& nd = nodes[n - 1 - j];
    double* row = &tab[(size_t)j * RE_SIZE];
    row[RE_Z] = nd.z; row[RE_XE] = nd.xe; row[RE_DKAPPADTAU] = nd.dk_dtau; row[RE_DKAPPADZ] = nd.dk_dz;
  }
  // baryon temperature: starts from the recombination value at z_start, explicit Euler towards z = 0 with Compton heating by the CMB
  {
    double* top = &tab[(size_t)(n - 1) * RE_SIZE];
    top[RE_TB] = reco[(size_t)i0 * RE_SIZE + RE_TB];
    top[RE_WB] = kBoltz / (kC * kC * kMH) * (1. + (1. / kHe4OverH - 1.) * tp.YHe + top[RE_XE] * (1. - tp.YHe)) * top[RE_TB];
    top[RE_CB2] = 5. / 3. * top[RE_WB];
  }
  for (int j = n - 1; j > 0; j--) {
    double* here = &tab[(size_t)j * RE_SIZE];
    double* below = &tab[(size_t)(j - 1) * RE_SIZE];
    const double zz = here[RE_Z];
    if ((rc = B.at_z(zz))) return rc;
    const double mean_mass = kMH / (1. + (1. / kHe4OverH - 1.) * tp.YHe + here[RE_XE] * (1. - tp.YHe));
    const double dT_dz = 2. / (1. + zz) * here[RE_TB] -
                         2. * mean_mass / kMe * 4. * B.rho_g() / 3. / B.rho_b() * opacity(zz, here[RE_XE]) * (cp.T_cmb * (1. + zz) - here[RE_TB]) / B.H();
    below[RE_TB] = here[RE_TB] - dT_dz * (here[RE_Z] - below[RE_Z]);
    below[RE_WB] = kBoltz / (kC * kC * mean_mass) * below[RE_TB];
    below[RE_CB2] = below[RE_WB] * (1. + (1. + zz) / 3. * dT_dz / below[RE_TB]);
  }
  // optical depth = integral of dkappa/dz over the reionization nodes (spline rule)
  std::vector<double> zz(n), f(n), m(n), cum(n);
  for (int j = 0; j < n; j++) { zz[j] = tab[(size_t)j * RE_SIZE + RE_Z]; f[j] = tab[(size_t)j * RE_SIZE + RE_DKAPPADZ]; }
  spline_table_lines(zz.data(), n, f.data(), 1, m.data());
  for (int j = 0; j < n; j++) tab[(size_t)j * RE_SIZE + RE_D3KAPPADZ3] = m[j];
  cpt_num::spline_cumulative_integral(zz.data(), n, f.data(), m.data(), 1, cum.data(), kReferenceIntegralRule);
  *optical_depth = cum[n - 1];
  return CPT_OK;
}

// running average over [i - radius, i + radius] clipped to the table
void box_smooth(double* table, int ncol, int n, int col, int radius) {
  std::vector<double> prefix(n + 1, 0.);
  for (int i = 0; i < n; i++) prefix[i + 1] = prefix[i] + table[(size_t)i * ncol + col];
  for (int i = 0; i < n; i++) {
    const int lo = std::max(i - radius, 0), hi = std::min(i + radius, n - 1);
    table[(size_t)i * ncol + col] = (prefix[hi + 1] - prefix[lo]) / (double)(hi - lo + 1);
  }
}
}  // namespace
}  // namespace cpt_host

extern "C" {

void cpt_host_thermo_defaults(cpt_thermo_params* p) {
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

void cpt_host_thermo_free(cpt_thermo* th) {
  if (!th) return;
  free(th->z_table); free(th->thermodynamics_table); free(th->d2thermodynamics_dz2_table);
  memset(th, 0, sizeof(*th));
}

int cpt_host_thermodynamics(const cpt_cosmo_params* cpp, const cpt_thermo_params* tpp, const cpt_background* bg, cpt_thermo* out) {
  if (!cpp || !tpp || !bg || !out) return fail_msg(CPT_ERR_INVALID, "null argument");
  const cpt_cosmo_params& cp = *cpp;
  const cpt_thermo_params& tp = *tpp;
  memset(out, 0, sizeof(*out));
  if (tp.reio_parametrization != CPT_REIO_NONE && tp.reio_parametrization != CPT_REIO_CAMB)
    return fail_msg(CPT_ERR_UNSUPPORTED, "host thermodynamics: reionization schemes none and camb only");
  if ((tp.YHe < kYHeMin) || (tp.YHe > kYHeMax)) return fail_msg(CPT_ERR_INVALID, "Y_He=%g out of bounds (%g<Y_He<%g)", tp.YHe, kYHeMin, kYHeMax);
  Background B(*bg);
  Recfast model(cp, tp, B);
  std::vector<double> reco, reio;
  int rc = recombination_history(cp, tp, B, model, reco);
  if (rc) return rc;
  const int Nz = tp.recfast_Nz0;
  int n_reio = 0, first_reco_kept = -1;
  double z_reionization = tp.z_reio, tau_reionization = tp.tau_reio;
  if (tp.reio_parametrization == CPT_REIO_CAMB) {
    ReioModel R;
    R.he_frac = tp.YHe / (kHe4OverH * (1. - tp.YHe));
    R.xe_after = 1. + R.he_frac;
    R.exponent = tp.reionization_exponent; R.width = tp.reionization_width; R.he_z = tp.helium_fullreio_redshift; R.he_width = tp.helium_fullreio_width;
    if (R.exponent == 0 || R.width == 0 || R.he_width == 0) return fail_msg(CPT_ERR_INVALID, "stop to avoid division by zero");
    const double helium_start = tp.helium_fullreio_redshift + tp.reionization_start_factor * tp.helium_fullreio_width;
    // history for a given reionization redshift -> optical depth
    auto history = [&](double z_reio, double z_start, double* depth) -> int {
      R.z_reio = z_reio; R.z_start = z_start;
      if (R.z_start > tp.reionization_z_start_max) return fail_msg(CPT_ERR_INVALID, "starting redshift for reionization > reionization_z_start_max = %e", tp.reionization_z_start_max);
      int r = xe_from_recombination(reco, Nz, R.z_start, &R.xe_before);
      if (r) return r;
      return reionization_history(cp, tp, B, model.nH0(), reco, R, reio, &n_reio, &first_reco_kept, depth);
    };
    auto start_of = [&](double z_reio) { return std::max(z_reio + tp.reionization_start_factor * tp.reionization_width, helium_start); };
    if (!tp.reio_from_tau) {
      if ((rc = history(z_reionization, start_of(z_reionization), &tau_reionization))) return rc;
    } else {   // bisection on z_reio for the requested optical depth
      double z_hi = tp.reionization_z_start_max - tp.reionization_start_factor * tp.reionization_width, z_lo = 0., tau_hi, tau_lo = 0.;
      if (z_hi < 0.) return fail_msg(CPT_ERR_INVALID, "parameters are such that reionization cannot take place before today while starting after z_start_max; need to increase z_start_max");
      if ((rc = history(z_hi, tp.reionization_z_start_max, &tau_hi))) return rc;
      if (tau_hi < tau_reionization) return fail_msg(CPT_ERR_INVALID, "parameters are such that reionization cannot start after z_start_max");
      for (int it = 0; (tau_hi - tau_lo) > tau_reionization * tp.reionization_optical_depth_tol; it++) {
        if (it > 10000) return fail_msg(CPT_ERR_RUNTIME, "while searching for reionization_optical_depth, maximum number of iterations exceeded");
        const double z_mid = 0.5 * (z_hi + z_lo);
        double tau_mid;
        if ((rc = history(z_mid, start_of(z_mid), &tau_mid))) return rc;
        if (tau_mid > tau_reionization) { z_hi = z_mid; tau_hi = tau_mid; } else { z_lo = z_mid; tau_lo = tau_mid; }
      }
      z_reionization = R.z_reio;
    }
  }
  // ---- merged table: reionization nodes, then the recombination nodes above them; the reference's 13 columns ----
  enum { TH_xe = 0, TH_dkappa, TH_tau_d, TH_ddkappa, TH_dddkappa, TH_exp_m_kappa, TH_g, TH_dg, TH_ddg, TH_Tb, TH_wb, TH_cb2, TH_rate, TH_SIZE };
  if (n_reio > 0 && reco[(size_t)first_reco_kept * RE_SIZE + RE_Z] != reio[(size_t)(n_reio - 1) * RE_SIZE + RE_Z])
    return fail_msg(CPT_ERR_RUNTIME, "mismatch which should never happen");
  const int n_above = Nz - first_reco_kept - 1, nt = n_reio + n_above, nc = TH_SIZE;
  out->tt_size = nt; out->th_size = nc;
  out->z_table = (double*)malloc(sizeof(double) * nt);
  out->thermodynamics_table = (double*)calloc((size_t)nt * nc, sizeof(double));
  out->d2thermodynamics_dz2_table = (double*)calloc((size_t)nt * nc, sizeof(double));
  if (!out->z_table || !out->thermodynamics_table || !out->d2thermodynamics_dz2_table) { cpt_host_thermo_free(out); return fail_msg(CPT_ERR_RUNTIME, "could not allocate the thermodynamics table"); }
  double* T = out->thermodynamics_table;
  double* zt = out->z_table;
  for (int i = 0; i < nt; i++) {
    const double* src = (i < n_reio) ? &reio[(size_t)i * RE_SIZE] : &reco[(size_t)(i - n_reio + first_reco_kept + 1) * RE_SIZE];
    double* row = T + (size_t)i * nc;
    zt[i] = src[RE_Z];
    row[TH_xe] = src[RE_XE]; row[TH_dkappa] = src[RE_DKAPPADTAU]; row[TH_Tb] = src[RE_TB]; row[TH_wb] = src[RE_WB]; row[TH_cb2] = src[RE_CB2];
  }
  auto bail = [&](int code) { cpt_host_thermo_free(out); return code; };
  // ---- opacity integrals and derivatives along conformal time (columns are strided views of the table) ----
  std::vector<double> tau(nt);
  for (int i = 0; i < nt; i++) if ((rc = B.tau_of_z(zt[i], &tau[i]))) return bail(rc);
  out->tau_ini = tau[nt - 1];
  std::vector<double> f(nt), m(nt), cum(nt), d1(nt);
  const cpt_num::ClampedSpline in_tau(tau.data(), nt);
  // baryon drag depth tau_d = integral of kappa' / R, R = 3 rho_b / 4 rho_g, from today backwards
  for (int i = 0; i < nt; i++) {
    if ((rc = B.columns_at_tau(tau[i], bg->index_bg_rho_g, bg->index_bg_rho_b))) return bail(rc);
    f[i] = -T[(size_t)i * nc + TH_dkappa] * (4. / 3.) * B.rho_g() / B.rho_b();
  }
  in_tau.moments(f.data(), 1, 1, m.data());
  cpt_num::spline_cumulative_integral(tau.data(), nt, f.data(), m.data(), 1, cum.data(), kReferenceIntegralRule);
  for (int i = 0; i < nt; i++) T[(size_t)i * nc + TH_tau_d] = cum[i];
  // kappa''' (the spline's second derivative of kappa'), kappa'' (its first derivative at the nodes), -kappa (its integral)
  for (int i = 0; i < nt; i++) f[i] = T[(size_t)i * nc + TH_dkappa];
  in_tau.moments(f.data(), 1, 1, m.data());
  cpt_num::spline_node_derivative(tau.data(), nt, f.data(), m.data(), 1, d1.data());
  cpt_num::spline_cumulative_integral(tau.data(), nt, f.data(), m.data(), 1, cum.data(), kReferenceIntegralRule);
  for (int i = 0; i < nt; i++) {
    double* r = T + (size_t)i * nc;
    const double k1 = r[TH_dkappa], k2 = d1[i], k3 = m[i], damp = std::exp(cum[i]);   // cum = -kappa (tau decreases with the index)
    if (k1 == 0.) return bail(fail_msg(CPT_ERR_RUNTIME, "variation rate diverges"));
    r[TH_ddkappa] = k2; r[TH_dddkappa] = k3;
    r[TH_exp_m_kappa] = damp;
    r[TH_g] = k1 * damp;                                     // visibility and its first two derivatives
    r[TH_dg] = (k2 + k1 * k1) * damp;
    r[TH_ddg] = (k3 + 3. * k1 * k2 + k1 * k1 * k1) * damp;
    r[TH_rate] = std::sqrt(k1 * k1 + (k2 / k1) * (k2 / k1) + std::fabs(k3 / k1));
  }
  box_smooth(T, nc, nt, TH_rate, tp.thermo_rate_smoothing_radius);
  spline_table_lines(zt, nt, T, nc, out->d2thermodynamics_dz2_table);
  // ---- recombination = maximum of the visibility function below z = 2000, located by the parabola through the three nodes around it ----
  auto G = [&](int i) { return T[(size_t)i * nc + TH_g]; };
  int it = nt - 1;
  while (zt[it] > kZRecMax) it--;
  if (G(it + 1) > G(it)) return bail(fail_msg(CPT_ERR_RUNTIME, "found a recombination redshift greater or equal to the maximum value imposed in thermodynamics.h, z_rec_max=%g", kZRecMax));
  while (G(it + 1) < G(it)) it--;
  const int i_peak = it;
  const double g_peak = G(it);
  out->z_rec = zt[it + 1] + 0.5 * (zt[it + 1] - zt[it]) * (G(it) - G(it + 2)) / (G(it) - 2. * G(it + 1) + G(it + 2));
  if (out->z_rec + cp.smallest_allowed_variation >= kZRecMax || out->z_rec - cp.smallest_allowed_variation <= kZRecMin)
    return bail(fail_msg(CPT_ERR_RUNTIME, "recombination redshift %g outside [%g, %g]", out->z_rec, kZRecMin, kZRecMax));
  if ((rc = B.tau_of_z(out->z_rec, &out->tau_rec)) || (rc = B.at_tau(out->tau_rec))) return bail(rc);
  out->rs_rec = B.row[bg->index_bg_rs];
  out->ra_rec = B.row[bg->index_bg_ang_distance] * (1. + out->z_rec) / cp.a_today;
  out->angular_rescaling = out->ra_rec / (bg->conformal_age - out->tau_rec);
  // time after which photons free-stream: 1 / kappa' exceeds the trigger fraction of tau (searched from the visibility peak down in z)
  {
    double tau_here;
    if ((rc = B.tau_of_z(zt[it], &tau_here))) return bail(rc);
    while (it > 0 && 1. / T[(size_t)it * nc + TH_dkappa] / tau_here < tp.radiation_streaming_trigger_tau_c_over_tau) {
      it--;
      if ((rc = B.tau_of_z(zt[it], &tau_here))) return bail(rc);
    }
    out->tau_free_streaming = tau_here;
  }
  // redshifts of unit optical depth (z_star) and unit drag depth (z_d), by linear interpolation between nodes
  auto crossing = [&](int col, double level, bool falling) {
    int i = 0;
    while (i < nt && (falling ? T[(size_t)i * nc + col] > level : T[(size_t)i * nc + col] < level)) i++;
    const double v0 = T[(size_t)(i - 1) * nc + col], v1 = T[(size_t)i * nc + col];
    return zt[i - 1] + (level - v0) / (v1 - v0) * (zt[i] - zt[i - 1]);
  };
  out->z_star = crossing(TH_exp_m_kappa, 1. / kEuler, true);
  out->z_d = crossing(TH_tau_d, 1., false);
  // time before which the CMB sources are negligible: visibility below a fraction of its peak value
  it = i_peak;
  while (it > 0 && G(it) > g_peak * tp.neglect_CMB_sources_below_visibility) it--;
  if ((rc = B.tau_of_z(zt[it], &out->tau_cut))) return bail(rc);
  out->YHe = tp.YHe; out->n_e = model.nH0(); out->z_reionization = z_reionization; out->tau_reionization = tau_reionization;
  out->index_th_xe = TH_xe; out->index_th_dkappa = TH_dkappa; out->index_th_tau_d = TH_tau_d; out->index_th_ddkappa = TH_ddkappa;
  out->index_th_dddkappa = TH_dddkappa; out->index_th_exp_m_kappa = TH_exp_m_kappa; out->index_th_g = TH_g; out->index_th_dg = TH_dg;
  out->index_th_ddg = TH_ddg; out->index_th_Tb = TH_Tb; out->index_th_wb = TH_wb; out->index_th_cb2 = TH_cb2; out->index_th_rate = TH_rate;
  return CPT_OK;
}
}
