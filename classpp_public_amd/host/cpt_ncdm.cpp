// Non-cold species on the host: what the reference's NonColdDarkMatter object holds after its own initialisation
// (tools/non_cold_dark_matter.cpp:202-790, tools/quadrature.c:69-360) - the momentum samplings of the perturbations and of the
// background, d ln f0 / d ln q at the nodes, the mass in units of the temperature, the normalisation factor, and the mass <-> density
// relation - computed from (m | Omega, T, deg, ksi) by this library's own routines:
//   * nodes and weights of Gauss-Laguerre rules from the three-term recurrence, roots by Newton iteration from the standard
//     asymptotic starting points, polished to round-off;
//   * the rule is the smallest Gauss-Laguerre rule that integrates the test function of ncdm.cpp:179-200 against f0 to the requested
//     tolerance, found by the reference's search order (coarse scan in steps of ten, then bisection, tools/quadrature.c:240-285) so
//     that both arrive at the same number of nodes; the converged value of the integral comes from a composite Gauss-Legendre rule;
//   * d ln f0 / d ln q analytically (the reference differentiates f0 numerically, ncdm.cpp:682-722);
//   * M from Omega by Newton iteration on rho(M) with the analytic derivative (ncdm.cpp:893-926).
// Scope: the Fermi-Dirac distribution with chemical potential (the reference's built-in f0, ncdm.cpp:105); distributions read from
// files and decaying species are CPT_ERR_UNSUPPORTED.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/cpt_host.h"

namespace cpt_host {
int fail_msg(int code, const char* fmt, ...);   // cpt_grids.cpp
}
#define cpt_host_set_error cpt_host::fail_msg

namespace {
constexpr double PI = 3.1415926535897932384626433832795;
// physical constants (include/background.h of the reference: the values define the unit conversions of the input parameters)
constexpr double kB = 1.3806504e-23, hP = 6.62606896e-34, cc = 2.99792458e8, GN = 6.67428e-11, eV = 1.602176487e-19, Mpc_over_m = 3.085677581282e22;
constexpr double zeta3 = 1.2020569031595942853997381615114499907649862923404988817922, zeta5 = 1.0369277551433699263313654864570341680570809195019128119741;

// Fermi-Dirac with chemical potential, mass eigenstate = flavour eigenstate (ncdm.cpp:105)
double f0_fd(double q, double ksi) { return (1. / (exp(q - ksi) + 1.) + 1. / (exp(q + ksi) + 1.)) / pow(2. * PI, 3); }
double dlnf0_dlnq_fd(double q, double ksi) {
  // f' / f with f = s(q - ksi) + s(q + ksi), s(x) = 1 / (e^x + 1), s' = -s (1 - s)
  const double a = 1. / (exp(q - ksi) + 1.), b = 1. / (exp(q + ksi) + 1.);
  if (a + b == 0.) return -q;   // extreme tail: f0 ~ exp(-q)
  return -q * (a * (1. - a) + b * (1. - b)) / (a + b);
}
// test function of the sampling search (ncdm.cpp:179-200): a cubic + quartic mix normalised so that its Fermi-Dirac integral is O(1)
double test_fn(double q) {
  const double c = 2. / (3. * zeta3), d = 120. / (7. * pow(PI, 4)), e = 2. / (45. * zeta5);
  return pow(2. * PI, 3) / 6. * (c * q * q - d * q * q * q - e * q * q * q * q);
}

// L_n(x) and L_n'(x) by the recurrence (k + 1) L_{k+1} = (2k + 1 - x) L_k - k L_{k-1}
void laguerre(int n, double x, double* L, double* dL, double* Lm1) {
  double p0 = 1., p1 = 1. - x;
  for (int k = 1; k < n; k++) {
    const double p2 = ((2. * k + 1. - x) * p1 - k * p0) / (k + 1.);
    p0 = p1; p1 = p2;
  }
  *L = p1; *Lm1 = p0;
  *dL = n * (p1 - p0) / x;   // x L_n' = n (L_n - L_{n-1})
}
// nodes (ascending) and weights of the n-point Gauss-Laguerre rule: int_0^inf e^-x g(x) dx ~ sum w_i g(x_i)
void gauss_laguerre(int n, std::vector<double>& x, std::vector<double>& w) {
  x.assign(n, 0.); w.assign(n, 0.);
  double z = 0.;
  for (int i = 0; i < n; i++) {
    // starting points (Stroud & Secrest): the first two from the small-root asymptotics, the rest by extrapolating the spacing
    if (i == 0) z = 3. / (1. + 2.4 * n);
    else if (i == 1) z += 15. / (1. + 2.5 * n);
    else { const double ai = i - 1.; z += (1. + 2.55 * ai) / (1.9 * ai) * (z - x[i - 2]); }
    double L, dL, Lm1;
    for (int it = 0; it < 100; it++) {
      laguerre(n, z, &L, &dL, &Lm1);
      // Newton on L_n with the roots already found divided out (keeps the iteration from falling back onto an earlier root)
      double s = 0.;
      for (int j = 0; j < i; j++) s += 1. / (z - x[j]);
      const double dz = L / (dL - L * s);
      z -= dz;
      if (fabs(dz) <= 4e-16 * fabs(z)) break;
    }
    for (int it = 0; it < 2; it++) {   // polish on the undeflated polynomial
      laguerre(n, z, &L, &dL, &Lm1);
      z -= L / dL;
    }
    laguerre(n, z, &L, &dL, &Lm1);
    x[i] = z;
    w[i] = -1. / (n * dL * Lm1);        // w_i = 1 / (x_i L_n'(x_i)^2) = -1 / (n L_n'(x_i) L_{n-1}(x_i)) at a root of L_n
  }
}

// converged value of int_0^inf f0(q) test(q) dq: 8-point Gauss-Legendre on panels of width 1/2 out to q = 120 + |ksi| (f0 < 1e-52 there)
double reference_integral(double ksi) {
  static const double gx[4] = {0.1834346424956498049394761, 0.5255324099163289858177390, 0.7966664774136267395915539, 0.9602898564975362316835609};
  static const double gw[4] = {0.3626837833783619829651504, 0.3137066458778872873379622, 0.2223810344533744705443560, 0.1012285362903762591525314};
  double s = 0.;
  const double qmax = 120. + fabs(ksi), hpanel = 0.5;
  for (double a = 0.; a < qmax; a += hpanel) {
    const double m = a + 0.5 * hpanel, r = 0.5 * hpanel;
    for (int i = 0; i < 4; i++)
      s += gw[i] * r * (f0_fd(m - r * gx[i], ksi) * test_fn(m - r * gx[i]) + f0_fd(m + r * gx[i], ksi) * test_fn(m + r * gx[i]));
  }
  return s;
}

struct Rule { std::vector<double> q, w; };
// the n-point rule for int f0(q) g(q) dq, and its estimate of the test integral
double laguerre_rule(int n, double ksi, Rule& R) {
  std::vector<double> x, w;
  gauss_laguerre(n, x, w);
  R.q = x; R.w.resize(n);
  double I = 0.;
  for (int i = 0; i < n; i++) {
    R.w[i] = w[i] * exp(x[i]) * f0_fd(x[i], ksi);
    I += R.w[i] * test_fn(x[i]);
  }
  return I;
}
// the sampling for tolerance rtol with at most n_max nodes: coarse scan n = 2, 12, 22, ... then bisection between the last failing and
// the first passing order; the rule returned is the last one evaluated (tools/quadrature.c:240-285, 318-321), trailing zero weights cut
bool choose_sampling(double ksi, double rtol, int n_max, Rule& R) {
  const double I = reference_integral(ksi);
  const int n_lim = n_max < 80 ? n_max : 80;
  int n = 2, n_old = 2;
  bool ok = false;
  for (;; ) {
    const double In = laguerre_rule(n, ksi, R);
    if (fabs((I - In) / I) < rtol) { ok = true; break; }
    n_old = n;
    if (n == n_lim) break;
    n = (n + 10 < n_lim) ? n + 10 : n_lim;
  }
  if (!ok) return false;
  int lo = n_old, hi = n;
  while (hi - lo > 1) {
    n = (lo + hi) / 2;
    const double In = laguerre_rule(n, ksi, R);
    if (fabs((I - In) / I) < rtol) hi = n; else lo = n;
  }
  while (!R.w.empty() && R.w.back() == 0.) { R.w.pop_back(); R.q.pop_back(); }
  return true;
}

// n, rho, p, d rho / d M, pseudo-pressure today from the background sampling (ncdm.cpp:805-851 at z = 0)
void momenta(const Rule& R, double factor, double M, double* nn, double* rho, double* drho_dM) {
  double sn = 0., sr = 0., sd = 0.;
  for (size_t i = 0; i < R.q.size(); i++) {
    const double q2 = R.q[i] * R.q[i], eps = sqrt(q2 + M * M);
    sn += q2 * R.w[i]; sr += q2 * eps * R.w[i]; sd += q2 * M / eps * R.w[i];
  }
  if (nn) *nn = sn * factor;
  if (rho) *rho = sr * factor;
  if (drho_dM) *drho_dM = sd * factor;
}

double* dup(const std::vector<double>& v) {
  double* p = (double*)malloc(sizeof(double) * (v.empty() ? 1 : v.size()));
  if (p && !v.empty()) memcpy(p, v.data(), sizeof(double) * v.size());
  return p;
}
}  // namespace

extern "C" {

void cpt_host_ncdm_defaults(cpt_ncdm_params* p) {
  memset(p, 0, sizeof(*p));
  p->T_cmb = 2.7255; p->h = 0.67556;
  for (int n = 0; n < CPT_MAX_NCDM; n++) { p->T_ncdm[n] = 0.71611; p->deg_ncdm[n] = 1.; }   // input_module.cpp:1040-1075
  p->tol_ncdm = 1e-3; p->tol_ncdm_bg = 1e-5; p->tol_M_ncdm = 1e-7;                           // include/precisions.h:34-54
}

void cpt_host_ncdm_free(cpt_ncdm* o) {
  if (!o) return;
  for (int n = 0; n < CPT_MAX_NCDM; n++) {
    free(o->q_ncdm[n]); free(o->w_ncdm[n]); free(o->dlnf0_dlnq_ncdm[n]); free(o->q_ncdm_bg[n]); free(o->w_ncdm_bg[n]);
  }
  memset(o, 0, sizeof(*o));
}

int cpt_host_ncdm(const cpt_ncdm_params* p, cpt_ncdm* out) {
  if (!p || !out) return cpt_host_set_error(CPT_ERR_INVALID, "null argument");
  memset(out, 0, sizeof(*out));
  if (p->N_ncdm < 1 || p->N_ncdm > CPT_MAX_NCDM) return cpt_host_set_error(CPT_ERR_UNSUPPORTED, "N_ncdm = %d: between 1 and %d species", p->N_ncdm, CPT_MAX_NCDM);
  if (!(p->T_cmb > 0.) || !(p->h > 0.)) return cpt_host_set_error(CPT_ERR_INVALID, "T_cmb and h must be positive");
  const double H0 = p->h * 1.e5 / cc;
  out->N_ncdm = p->N_ncdm;
  for (int n = 0; n < p->N_ncdm; n++) {
    const double ksi = p->ksi_ncdm[n], T = p->T_ncdm[n];
    if (!(T > 0.) || !(p->deg_ncdm[n] > 0.)) { cpt_host_ncdm_free(out); return cpt_host_set_error(CPT_ERR_INVALID, "T_ncdm and deg_ncdm must be positive (species %d)", n); }
    if (p->m_ncdm_in_eV[n] < 0. || p->Omega0_ncdm[n] < 0. || (p->m_ncdm_in_eV[n] == 0. && p->Omega0_ncdm[n] == 0.)) {
      cpt_host_ncdm_free(out);
      return cpt_host_set_error(CPT_ERR_INVALID, "species %d: give a positive m_ncdm or Omega_ncdm / omega_ncdm", n);
    }
    Rule pert, bg;
    if (!choose_sampling(ksi, p->tol_ncdm, 250, pert) || !choose_sampling(ksi, p->tol_ncdm_bg, 800, bg)) {
      cpt_host_ncdm_free(out);
      return cpt_host_set_error(CPT_ERR_UNSUPPORTED, "no Gauss-Laguerre rule of at most 80 nodes reaches tol_ncdm = %g / tol_ncdm_bg = %g (species %d)", p->tol_ncdm,
                                p->tol_ncdm_bg, n);
    }
    std::vector<double> dl(pert.q.size());
    for (size_t i = 0; i < dl.size(); i++) dl[i] = dlnf0_dlnq_fd(pert.q[i], ksi);
    // normalisation: rho = factor (1 + z)^4 int q^2 eps f0 dq in units where H^2 = sum rho (ncdm.cpp:724-725)
    double factor = p->deg_ncdm[n] * 4. * PI * pow(p->T_cmb * T * kB, 4) * 8. * PI * GN / 3. / pow(hP / 2. / PI, 3) / pow(cc, 7) * Mpc_over_m * Mpc_over_m;
    double deg = p->deg_ncdm[n], M, Omega0 = p->Omega0_ncdm[n], m_eV = p->m_ncdm_in_eV[n];
    if (m_eV != 0.) {
      M = m_eV / kB * eV / T / p->T_cmb;
      double rho;
      momenta(bg, factor, M, nullptr, &rho, nullptr);
      if (Omega0 == 0.) Omega0 = rho / H0 / H0;
      else {   // mass and density both given: the degeneracy absorbs the difference (ncdm.cpp:774-779)
        const double f = H0 * H0 * Omega0 / rho;
        factor *= f; deg *= f;
      }
    } else {
      // M from Omega: Newton on rho(M) from the non-relativistic guess M = rho0 / n (ncdm.cpp:893-926)
      const double rho0 = H0 * H0 * Omega0;
      double nn, rho, drho;
      momenta(bg, factor, 0., &nn, &rho, nullptr);
      if (rho0 < rho) {
        cpt_host_ncdm_free(out);
        return cpt_host_set_error(CPT_ERR_INVALID, "The value of Omega for species %d, %g, is less than for a massless species: it should be at least %g", n, Omega0, Omega0 * rho / rho0);
      }
      M = rho0 / nn;
      bool conv = false;
      for (int it = 0; it < 50 && !conv; it++) {
        momenta(bg, factor, M, nullptr, &rho, &drho);
        double dM = (rho0 - rho) / drho;
        if (M + dM < 0.) dM = -M / 2.;
        M += dM;
        conv = fabs(dM / M) < p->tol_M_ncdm;
      }
      if (!conv) { cpt_host_ncdm_free(out); return cpt_host_set_error(CPT_ERR_RUNTIME, "Newton iteration for the mass of species %d did not converge", n); }
      m_eV = kB / eV * T * M * p->T_cmb;
    }
    out->q_size_ncdm[n] = (int)pert.q.size(); out->q_size_ncdm_bg[n] = (int)bg.q.size();
    out->q_ncdm[n] = dup(pert.q); out->w_ncdm[n] = dup(pert.w); out->dlnf0_dlnq_ncdm[n] = dup(dl);
    out->q_ncdm_bg[n] = dup(bg.q); out->w_ncdm_bg[n] = dup(bg.w);
    out->M_ncdm[n] = M; out->factor_ncdm[n] = factor; out->Omega0_ncdm[n] = Omega0; out->m_ncdm_in_eV[n] = m_eV; out->deg_ncdm[n] = deg;
    out->Omega0_ncdm_tot += Omega0;
  }
  return CPT_OK;
}

}  // extern "C"
