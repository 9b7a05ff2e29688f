// Host-side stiff integrator: evolver_ndf15 of the reference (tools/evolver_ndf15.cpp:62-705) in its dense branch (the reference
// switches to sparse LU above 15 equations; the host-side systems - background: 5 equations - are below that), with numjac
// (:1213-1539), the dense LU (:1001-1064), adjust_stepsize (:907-943) and interp_from_dif (:860-905).  Used by the background
// integration (classpp_public_amd/host/cpt_cosmo.cpp).  Header-only, templated on the RHS / output functors.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <vector>

namespace orc_host {

// ---- ndf15: ev.cpp:62-705 (+ numjac :1213-1539 in its dense mode, dense LU :1001-1064) ----
struct Ndf {
  int neq;
  std::vector<double> J, LU, fac;  // dense Jacobian (row-major), LU of I - h*gamma*J, numjac increments
  std::vector<int> piv;
  long stat[6] = {0, 0, 0, 0, 0, 0};
};

inline bool ludcmp(std::vector<double>& A, int n, std::vector<int>& indx) {  // ev.cpp:1021-1064, 0-based
  std::vector<double> vv(n);
  for (int i = 0; i < n; i++) {
    double big = 0.;
    for (int j = 0; j < n; j++) big = std::max(big, std::fabs(A[i * n + j]));
    if (big == 0.) return false;
    vv[i] = 1.0 / big;
  }
  for (int j = 0; j < n; j++) {
    for (int i = 0; i < j; i++) {
      double sum = A[i * n + j];
      for (int k = 0; k < i; k++) sum -= A[i * n + k] * A[k * n + j];
      A[i * n + j] = sum;
    }
    double big = 0.; int imax = j;
    for (int i = j; i < n; i++) {
      double sum = A[i * n + j];
      for (int k = 0; k < j; k++) sum -= A[i * n + k] * A[k * n + j];
      A[i * n + j] = sum;
      double dum = vv[i] * std::fabs(sum);
      if (dum >= big) { big = dum; imax = i; }
    }
    if (j != imax) {
      for (int k = 0; k < n; k++) std::swap(A[imax * n + k], A[j * n + k]);
      vv[imax] = vv[j];
    }
    indx[j] = imax;
    if (A[j * n + j] == 0.0) A[j * n + j] = 1e-50;
    if (j != n - 1) {
      double dum = 1.0 / A[j * n + j];
      for (int i = j + 1; i < n; i++) A[i * n + j] *= dum;
    }
  }
  return true;
}
inline void lubksb(const std::vector<double>& A, int n, const std::vector<int>& indx, double* b) {  // ev.cpp:1001-1019
  int ii = -1;
  for (int i = 0; i < n; i++) {
    int ip = indx[i];
    double sum = b[ip];
    b[ip] = b[i];
    if (ii >= 0) for (int j = ii; j <= i - 1; j++) sum -= A[i * n + j] * b[j];
    else if (sum) ii = i;
    b[i] = sum;
  }
  for (int i = n - 1; i >= 0; i--) {
    double sum = b[i];
    for (int j = i + 1; j < n; j++) sum -= A[i * n + j] * b[j];
    b[i] = sum / A[i * n + i];
  }
}

template <class F>
void numjac(F&& f, double t, const double* y, const double* fval, Ndf& S, int* nfe) {  // ev.cpp:1213-1539, dense branch
  const int n = S.neq;
  const double eps = 1e-16, br = std::pow(eps, 0.875), bl = std::pow(eps, 0.75), bu = std::pow(eps, 0.25);
  const double facmin = std::pow(eps, 0.78), facmax = 0.1, thresh = 1e-15, TINY = 1e-50;
  std::vector<double> yscale(n), del(n), ydel(n), ffdel(n), Fdel((size_t)n * n), Difmax(n), absFdelRm(n), absFvalue(n), absFvalueRm(n), Fscale(n), tmp(n);
  std::vector<int> Rowmax(n, 0), logj(n);
  std::vector<double>& fac = S.fac;
  for (int j = 0; j < n; j++) {
    yscale[j] = std::max(std::fabs(y[j]), thresh);
    del[j] = (y[j] + fac[j] * yscale[j]) - y[j];
  }
  for (int j = 0; j < n; j++) {
    if (del[j] == 0.0) {
      for (;;) {
        if (fac[j] < facmax) {
          fac[j] = std::min(100 * fac[j], facmax);
          del[j] = (y[j] + fac[j] * yscale[j]) - y[j];
          if (del[j] == 0.0) break;
        } else { del[j] = thresh; break; }
      }
    }
  }
  for (int j = 0; j < n; j++) del[j] = (fval[j] >= 0.0) ? std::fabs(del[j]) : -std::fabs(del[j]);
  for (int j = 0; j < n; j++) {
    for (int i = 0; i < n; i++) ydel[i] = y[i];
    ydel[j] += del[j];
    f(t, ydel.data(), ffdel.data());
    (*nfe)++;
    for (int i = 0; i < n; i++) Fdel[(size_t)i * n + j] = ffdel[i];
  }
  for (int j = 0; j < n; j++) {
    double Fdiff_new = 0.0, Fdiff_absrm = 0.0;
    for (int i = 0; i < n; i++) {
      Fdiff_absrm = std::max(std::fabs(Fdiff_new), Fdiff_absrm);
      Fdiff_new = Fdel[(size_t)i * n + j] - fval[i];
      S.J[(size_t)i * n + j] = Fdiff_new / del[j];
      if (std::fabs(Fdiff_new) >= Fdiff_absrm) { Rowmax[j] = i; Difmax[j] = std::fabs(Fdiff_new); }
    }
    absFdelRm[j] = std::fabs(Fdel[(size_t)Rowmax[j] * n + j]);
  }
  for (int i = 0; i < n; i++) absFvalue[i] = std::fabs(fval[i]);
  for (int j = 0; j < n; j++) absFvalueRm[j] = absFvalue[Rowmax[j]];
  int logjpos = 0;
  for (int j = 0; j < n; j++) {
    if (((absFdelRm[j] < TINY) && (absFvalueRm[j] < TINY)) || (std::fabs(Difmax[j]) < TINY)) { logj[j] = 1; logjpos = 1; }
    else logj[j] = 0;
  }
  if (logjpos == 1) {
    for (int i = 0; i < n; i++) { ydel[i] = y[i]; Fscale[i] = std::max(absFdelRm[i], absFvalueRm[i]); }
    for (int j = 0; j < n; j++) {
      if ((logj[j] == 1) && (Difmax[j] <= (br * Fscale[j]))) {
        double tmpfac = std::min(std::sqrt(fac[j]), facmax);
        double del2 = (y[j] + tmpfac * yscale[j]) - y[j];
        if ((tmpfac != fac[j]) && (del2 != 0.0)) {
          del2 = (fval[j] >= 0.0) ? std::fabs(del2) : -std::fabs(del2);
          ydel[j] = y[j] + del2;
          f(t, ydel.data(), ffdel.data());
          (*nfe)++;
          ydel[j] = y[j];
          int rowmax2 = 0; double difmax2 = 0., Fdiff_new = 0., Fdiff_absrm = 0.;
          for (int i = 0; i < n; i++) {
            Fdiff_absrm = std::max(Fdiff_absrm, std::fabs(Fdiff_new));
            Fdiff_new = ffdel[i] - fval[i];
            tmp[i] = Fdiff_new / del2;
            if (std::fabs(Fdiff_new) >= Fdiff_absrm) { rowmax2 = i; difmax2 = std::fabs(Fdiff_new); }
          }
          double maxval1 = difmax2 * std::fabs(del2) * tmpfac, maxval2 = Difmax[j] * std::fabs(del[j]);
          if (maxval1 >= maxval2) {
            for (int i = 0; i < n; i++) S.J[(size_t)i * n + j] = tmp[i];
            double ffscale = std::max(std::fabs(ffdel[rowmax2]), absFvalue[rowmax2]);
            if (difmax2 <= bl * ffscale) fac[j] = std::min(10 * tmpfac, facmax);
            else if (difmax2 > bu * ffscale) fac[j] = std::max(0.1 * tmpfac, facmin);
            else fac[j] = tmpfac;
          }
        }
      }
    }
  }
}

inline void adjust_stepsize(std::vector<double>& dif, int neq, double r, int k) {  // ev.cpp:907-943; dif[i*7 + j], j = 0..6
  const double U[5][5] = {{-1, -2, -3, -4, -5}, {0, 1, 3, 6, 10}, {0, 0, -1, -4, -10}, {0, 0, 0, 1, 5}, {0, 0, 0, 0, -1}};
  double RU[5][5], tv[5];
  for (int ii = 1; ii <= 5; ii++) RU[0][ii - 1] = -ii * r;
  for (int jj = 2; jj <= 5; jj++)
    for (int ii = 1; ii <= 5; ii++) RU[jj - 1][ii - 1] = RU[jj - 2][ii - 1] * (1.0 - (1.0 + ii * r) / jj);
  for (int ii = 0; ii < 5; ii++) {
    for (int kk = 0; kk < 5; kk++) tv[kk] = RU[ii][kk];
    for (int jj = 0; jj < 5; jj++) {
      RU[ii][jj] = 0.0;
      for (int kk = 0; kk < 5; kk++) RU[ii][jj] += tv[kk] * U[kk][jj];
    }
  }
  for (int ii = 0; ii < neq; ii++) {
    for (int kk = 0; kk < k; kk++) tv[kk] = dif[(size_t)ii * 7 + kk];
    for (int jj = 0; jj < k; jj++) {
      double s = 0.0;
      for (int kk = 0; kk < k; kk++) s += tv[kk] * RU[kk][jj];
      dif[(size_t)ii * 7 + jj] = s;
    }
  }
}

inline bool new_linearisation(Ndf& S, double hinvGak) {  // ev.cpp:945-998, dense branch
  const int n = S.neq;
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) S.LU[(size_t)i * n + j] = -hinvGak * S.J[(size_t)i * n + j] + (i == j ? 1.0 : 0.0);
  return ludcmp(S.LU, n, S.piv);
}

// returns 0 ok, 1 "step size too small", 2 singular matrix
template <class F, class O>
int ndf15(F&& f, O&& output, double t0, double tfinal, double* y_inout, const int* used_in_output, int neq, double rtol,
          double minimum_variation, const double* t_vec, int tres, Ndf& S) {
  const double G[5] = {1.0, 3.0 / 2.0, 11.0 / 6.0, 25.0 / 12.0, 137.0 / 60.0};
  const double alpha[5] = {-37.0 / 200, -1.0 / 9.0, -8.23e-2, -4.15e-2, 0};
  double invGa[5], erconst[5];
  const double abstol = 1e-15, eps = 1e-16, threshold = abstol;
  const int maxit = 4, maxk = 5;
  for (int i = 0; i < 5; i++) { invGa[i] = 1.0 / (G[i] * (1.0 - alpha[i])); erconst[i] = alpha[i] * G[i] + 1.0 / (2.0 + i); }
  S.neq = neq;
  S.J.assign((size_t)neq * neq, 0.); S.LU.assign((size_t)neq * neq, 0.); S.piv.assign(neq, 0);
  S.fac.assign(neq, 1.490116119384765597872e-8);
  std::vector<double> f0(neq), wt(neq), ddfddt(neq), pred(neq), y(neq), invwt(neq), rhs(neq), psi(neq), difkp1(neq), del(neq),
      yinterp(neq), ypinterp(neq), tempvec1(neq), dif((size_t)neq * 7, 0.);
  double* ynew = y_inout;
  for (int i = 0; i < neq; i++) y[i] = y_inout[i];
  int next = 0;
  while (t_vec[next] < t0) next++;
  double htspan = std::fabs(tfinal - t0);
  f(t0, y.data(), f0.data()); S.stat[2]++;
  int tdir = (tfinal - t0) < 0.0 ? -1 : 1;
  double hmax = (tfinal - t0) / 10.0;
  double t = t0;
  int nfenj = 0;
  numjac(f, t, y.data(), f0.data(), S, &nfenj);
  S.stat[3]++; S.stat[2] += nfenj;
  bool Jcurrent = true;
  double hmin = 16.0 * eps * std::fabs(t);
  double rh = 0.0;
  for (int j = 0; j < neq; j++) { wt[j] = std::max(std::fabs(y[j]), threshold); rh = std::max(rh, 1.25 / std::sqrt(rtol) * std::fabs(f0[j] / wt[j])); }
  double absh = std::min(hmax, htspan);
  if (absh * rh > 1.0) absh = 1.0 / rh;
  absh = std::max(absh, hmin);
  double h = tdir * absh;
  double tdel = (t + tdir * std::min(std::sqrt(eps) * std::max(std::fabs(t), std::fabs(t + h)), absh)) - t;
  f(t + tdel, y.data(), tempvec1.data()); S.stat[2]++;
  for (int i = 0; i < neq; i++) { ddfddt[i] = 0.0; for (int j = 0; j < neq; j++) ddfddt[i] += S.J[(size_t)i * neq + j] * f0[j]; }
  rh = 0.0;
  for (int i = 0; i < neq; i++) { ddfddt[i] += (tempvec1[i] - f0[i]) / tdel; rh = std::max(rh, 1.25 * std::sqrt(0.5 * std::fabs(ddfddt[i] / wt[i]) / rtol)); }
  absh = std::min(hmax, htspan);
  if (absh * rh > 1.0) absh = 1.0 / rh;
  absh = std::max(absh, hmin);
  h = tdir * absh;
  int k = 1, klast = k;
  double abshlast = absh;
  for (int i = 0; i < neq; i++) dif[(size_t)i * 7 + 0] = h * f0[i];
  double hinvGak = h * invGa[k - 1];
  int nconhk = 0;
  if (!new_linearisation(S, hinvGak)) return 2;
  S.stat[4]++;
  bool havrate = false, done = false, at_hmin = false;
  double rate = 0., oldnrm = 0., tnew = t, err = 0.;
  while (!done) {
    hmin = minimum_variation;
    absh = std::min(hmax, std::max(hmin, absh));
    if (std::fabs(absh - hmin) < 100 * eps) { if (at_hmin) absh = abshlast; at_hmin = true; } else at_hmin = false;
    h = tdir * absh;
    if (1.1 * absh >= std::fabs(tfinal - t)) { h = tfinal - t; absh = std::fabs(h); done = true; }
    if (((std::fabs(absh - abshlast) / absh) > 1e-6) || (k != klast)) {
      adjust_stepsize(dif, neq, absh / abshlast, k);
      hinvGak = h * invGa[k - 1];
      nconhk = 0;
      if (!new_linearisation(S, hinvGak)) return 2;
      S.stat[4]++;
      havrate = false;
    }
    bool nofailed = true;
    for (;;) {
      bool gotynew = false;
      while (!gotynew) {
        for (int i = 0; i < neq; i++) { psi[i] = 0.0; for (int j = 1; j <= k; j++) psi[i] += dif[(size_t)i * 7 + j - 1] * G[j - 1] * invGa[k - 1]; }
        tnew = t + h;
        if (done) tnew = tfinal;
        h = tnew - t;
        for (int i = 0; i < neq; i++) { pred[i] = y[i]; for (int j = 1; j <= k; j++) pred[i] += dif[(size_t)i * 7 + j - 1]; }
        for (int i = 0; i < neq; i++) ynew[i] = pred[i];
        double minnrm = 0.0;
        for (int j = 0; j < neq; j++) {
          difkp1[j] = 0.0;
          invwt[j] = 1.0 / std::max(std::max(std::fabs(ynew[j]), std::fabs(y[j])), threshold);
          minnrm = std::max(minnrm, 100 * eps * std::fabs(ynew[j] * invwt[j]));
        }
        bool tooslow = false;
        for (int iter = 1; iter <= maxit; iter++) {
          for (int i = 0; i < neq; i++) tempvec1[i] = psi[i] + difkp1[i];
          f(tnew, ynew, f0.data()); S.stat[2]++;
          for (int j = 0; j < neq; j++) rhs[j] = hinvGak * f0[j] - tempvec1[j];
          for (int j = 0; j < neq; j++) del[j] = rhs[j];
          lubksb(S.LU, neq, S.piv, del.data());
          S.stat[5]++;
          double newnrm = 0.0;
          for (int j = 0; j < neq; j++) newnrm = std::max(newnrm, std::fabs(del[j] * invwt[j]));
          for (int j = 0; j < neq; j++) { difkp1[j] += del[j]; ynew[j] = pred[j] + difkp1[j]; }
          if (newnrm <= minnrm) { gotynew = true; break; }
          else if (iter == 1) {
            if (havrate) { double errit = newnrm * rate / (1.0 - rate); if (errit <= 0.05 * rtol) { gotynew = true; break; } }
            else rate = 0.0;
          } else if (newnrm > 0.9 * oldnrm) { tooslow = true; break; }
          else {
            rate = std::max(0.9 * rate, newnrm / oldnrm);
            havrate = true;
            double errit = newnrm * rate / (1.0 - rate);
            if (errit <= 0.5 * rtol) { gotynew = true; break; }
            else if (iter == maxit) { tooslow = true; break; }
            else if (0.5 * rtol < errit * std::pow(rate, (maxit - iter))) { tooslow = true; break; }
          }
          oldnrm = newnrm;
        }
        if (tooslow) {
          S.stat[1]++;
          if (!Jcurrent) {
            f(t, y.data(), f0.data());
            nfenj = 0;
            numjac(f, t, y.data(), f0.data(), S, &nfenj);
            S.stat[3]++; S.stat[2] += nfenj + 1;
            Jcurrent = true;
          } else if (absh <= hmin) return 1;
          else {
            abshlast = absh;
            absh = std::max(0.3 * absh, hmin);
            h = tdir * absh;
            done = false;
            adjust_stepsize(dif, neq, absh / abshlast, k);
            hinvGak = h * invGa[k - 1];
            nconhk = 0;
          }
          if (!new_linearisation(S, hinvGak)) return 2;
          S.stat[4]++;
          havrate = false;
        }
      }
      err = 0.0;
      for (int j = 0; j < neq; j++) err = std::max(err, std::fabs(difkp1[j] * invwt[j]));
      err = err * erconst[k - 1];
      if (err > rtol) {
        S.stat[1]++;
        if (absh <= hmin) return 1;
        abshlast = absh;
        if (nofailed) {
          nofailed = false;
          double hopt = absh * std::max(0.1, 0.833 * std::pow((rtol / err), (1.0 / (k + 1))));
          if (k > 1) {
            double errkm1 = 0.0;
            for (int j = 0; j < neq; j++) errkm1 = std::max(errkm1, std::fabs((dif[(size_t)j * 7 + k - 1] + difkp1[j]) * invwt[j]));
            errkm1 = errkm1 * erconst[k - 2];
            double hkm1 = absh * std::max(0.1, 0.769 * std::pow((rtol / errkm1), (1.0 / k)));
            if (hkm1 > hopt) { hopt = std::min(absh, hkm1); k = k - 1; }
          }
          absh = std::max(hmin, hopt);
        } else absh = std::max(hmin, 0.5 * absh);
        h = tdir * absh;
        if (absh < abshlast) done = false;
        adjust_stepsize(dif, neq, absh / abshlast, k);
        hinvGak = h * invGa[k - 1];
        nconhk = 0;
        if (!new_linearisation(S, hinvGak)) return 2;
        S.stat[4]++;
        havrate = false;
      } else break;
    }
    S.stat[0]++;
    for (int j = 0; j < neq; j++) { dif[(size_t)j * 7 + k + 1] = difkp1[j] - dif[(size_t)j * 7 + k]; dif[(size_t)j * 7 + k] = difkp1[j]; }
    for (int j = k; j >= 1; j--) for (int i = 0; i < neq; i++) dif[(size_t)i * 7 + j - 1] += dif[(size_t)i * 7 + j];
    while ((next < tres) && (tdir * (tnew - t_vec[next]) >= 0.0)) {
      if (tnew == t_vec[next]) output(t_vec[next], ynew, f0.data(), next);
      else {
        // interp_from_dif ev.cpp:860-905
        double s = (t_vec[next] - tnew) / h, prod = 1.0, sumfrac = 0., fact = 1.0, vecy[5], vecdy[5];
        for (int j = 0; j < k; j++) { prod *= (s + j); fact *= (j + 1); sumfrac += 1.0 / (s + j); vecy[j] = prod / fact; vecdy[j] = prod * sumfrac / (h * fact); }
        for (int i = 0; i < neq; i++) {
          if (used_in_output[i]) {
            double s1 = 0, s2 = 0;
            for (int j = 0; j < k; j++) { s1 += vecy[j] * dif[(size_t)i * 7 + j]; s2 += vecdy[j] * dif[(size_t)i * 7 + j]; }
            yinterp[i] = ynew[i] + s1; ypinterp[i] = s2;
          }
        }
        output(t_vec[next], yinterp.data(), ypinterp.data(), next);
      }
      next++;
    }
    if (done) break;
    klast = k;
    abshlast = absh;
    nconhk = std::min(nconhk + 1, maxk + 2);
    if (nconhk >= k + 2) {
      double temp = 1.2 * std::pow((err / rtol), (1.0 / (k + 1.0)));
      double hopt = temp > 0.1 ? absh / temp : 10 * absh;
      int kopt = k;
      if (k > 1) {
        double errkm1 = 0.0;
        for (int j = 0; j < neq; j++) errkm1 = std::max(errkm1, std::fabs(dif[(size_t)j * 7 + k - 1] * invwt[j]));
        errkm1 = errkm1 * erconst[k - 2];
        temp = 1.3 * std::pow((errkm1 / rtol), (1.0 / k));
        double hkm1 = temp > 0.1 ? absh / temp : 10 * absh;
        if (hkm1 > hopt) { hopt = hkm1; kopt = k - 1; }
      }
      if (k < maxk) {
        double errkp1 = 0.0;
        for (int j = 0; j < neq; j++) errkp1 = std::max(errkp1, std::fabs(dif[(size_t)j * 7 + k + 1] * invwt[j]));
        errkp1 = errkp1 * erconst[k];
        temp = 1.4 * std::pow((errkp1 / rtol), (1.0 / (k + 2.0)));
        double hkp1 = temp > 0.1 ? absh / temp : 10 * absh;
        if (hkp1 > hopt) { hopt = hkp1; kopt = k + 1; }
      }
      if (hopt > absh) { absh = hopt; if (k != kopt) k = kopt; }
    }
    t = tnew;
    for (int i = 0; i < neq; i++) y[i] = ynew[i];
    Jcurrent = false;
  }
  f(tnew, ynew, f0.data());  // ev.cpp:653-662: leaves the workspace consistent for the next regime
  return 0;
}

}  // namespace orc_host
