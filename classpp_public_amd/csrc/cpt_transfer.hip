// Hot path B on MI355X: line-of-sight projection of the source functions onto flat spherical Bessel functions.
//
// Restates (not translates) TransferModule::transfer_init / transfer_compute_for_each_q of the reference
// (source/transfer_module.cpp, "tm.cpp") for flat space and the CMB scalar types t0,t1,t2,e + lcmb:
//   k_bessel_*      tools/hyperspherical.c:11-246   flat j_l(x), j_l'(x) table on a uniform x grid (one thread per x)
//   k_source_spline tm.cpp:604-639 -> tools/arrays.c:967-1092  d2S/dk2 for every (type, tau) row
//   k_los           tm.cpp:1488-1715, 1767-1820, 1845-2109, 2586-2892, 3187-3272, 3274-3445 + the Hermite-4
//                   interpolation of tools/hermite4_interpolation_csource.h:55-160, fused over the types that share
//                   the same Phi_l row; one workgroup per q, one wavefront per (q,l), lanes stride tau, DPP/shuffle
//                   reduction.  HBM/L2-bound (no dense contraction => no MFMA), see DESIGN.md.
#include "cpt_internal.h"

// ---------------------------------------------------------------------------------------------
// Bessel table
// ---------------------------------------------------------------------------------------------
// Lentz continued fraction for j_l'(x)/j_l(x) (hyperspherical.c:677-716 get_CF1, K=0, beta=1)
__device__ static inline void cf1_flat(int l, double cotK, double* CF, int* isign) {
  const double tiny = 1e-100, reltol = 2.220446049250313e-16;
  double bj = l * cotK, fj = bj, Cj = bj, Dj = 0.0;
  int sgn = 1;
  for (int j = 1; j <= 1000000; j++) {
    bj = (double)(2 * (l + j) + 1) * cotK;
    Dj = bj - Dj;
    if (Dj == 0.0) Dj = tiny;
    Cj = bj - 1.0 / Cj;
    if (Cj == 0.0) Cj = tiny;
    Dj = 1.0 / Dj;
    double Delj = Cj * Dj;
    fj = fj * Delj;
    if (Dj < 0) sgn = -sgn;
    if (fabs(Delj - 1.0) < reltol) break;
  }
  *CF = fj;
  *isign = sgn;
}

// One thread per abscissa x_j. Recurrences run over every l up to lmax+1, only the listed l are stored.
// Backward branch: pass 1 counts the 1e-200 overflow rescalings and gets the normalisation, pass 2 repeats
// the recurrence and writes final values (equivalent to the reference's in-place rescaling of PhiL[]).
__global__ void __launch_bounds__(64) k_bessel(double2* __restrict__ bes, const int* __restrict__ lvec, int nl, int nx,
                                               double xmin, double deltax, int xfwdidx) {
  int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= nx) return;
  const int lmax = lvec[nl - 1];
  const int L = lmax + 1;
  const double x = xmin + j * deltax;
  const double cotK = 1.0 / x;
  if (j >= xfwdidx) {
    // forward recurrence, hyperspherical.c:490-514
    double p0 = sin(x) / x;
    double p1 = p0 * (cotK - 1.0 / tan(x));
    int kk = 0;
    // l = 0,1 are never in the list (l >= 2) but handle them anyway
    while (kk < nl && lvec[kk] == 0) { bes[(size_t)kk * nx + j] = make_double2(p0, -p1); kk++; }
    double pm = p0, pc = p1;  // P[l-1], P[l]
    for (int l = 1; l <= lmax; l++) {
      double pn = (double)(2 * l + 1) * cotK * pc - pm;  // P[l+1]
      if (kk < nl && lvec[kk] == l) {
        bes[(size_t)kk * nx + j] = make_double2(pc, l * cotK * pc - pn);
        kk++;
      }
      pm = pc;
      pc = pn;
    }
    return;
  }
  // backward recurrence, hyperspherical.c:517-603
  const double phi0 = sin(x) / x;
  double phipr1;
  int isign;
  cf1_flat(L, cotK, &phipr1, &isign);
  const double phi1 = (double)isign;
  phipr1 *= phi1;
  const int l_align = L - L % 8;
  int n_rescale_total = 0;
  double scaling = 0.;
  for (int pass = 0; pass < 2; pass++) {
    double phi = phi1, phi_plus = L * cotK * phi1 - phipr1;  // P[l], P[l+1]
    int kk = nl - 1;
    int n_done = 0;
    // P[L] itself is never stored (L = lmax+1 is not in the list)
    int l = L;
    for (; l > l_align; l--) {
      double pmv = (double)(2 * l + 1) * cotK * phi - phi_plus;  // P[l-1]
      phi_plus = phi;
      phi = pmv;
      // now phi = P[l-1], phi_plus = P[l]
      if (pass == 1 && kk >= 0 && lvec[kk] == l - 1) {
        double v = phi, vp = phi_plus;
        int later = n_rescale_total - n_done;
        for (int r = 0; r < later && r < 8; r++) { v *= 1e-200; vp *= 1e-200; }
        if (later >= 8) { v = 0.; vp = 0.; }
        v *= scaling; vp *= scaling;
        bes[(size_t)kk * nx + j] = make_double2(v, (l - 1) * cotK * v - vp);
        kk--;
      }
    }
    for (int l_ini = l_align; l_ini > 0; l_ini -= 8) {
      for (l = l_ini; l > l_ini - 8; l--) {
        double pmv = (double)(2 * l + 1) * cotK * phi - phi_plus;
        phi_plus = phi;
        phi = pmv;
        if (pass == 1 && kk >= 0 && lvec[kk] == l - 1) {
          // the reference stores the raw value now and multiplies it by 1e-200 at every LATER rescaling,
          // including the one that may follow this very block
          double v = phi, vp = phi_plus;
          int later = n_rescale_total - n_done;
          for (int r = 0; r < later && r < 8; r++) { v *= 1e-200; vp *= 1e-200; }
          if (later >= 8) { v = 0.; vp = 0.; }
          v *= scaling; vp *= scaling;
          bes[(size_t)kk * nx + j] = make_double2(v, (l - 1) * cotK * v - vp);
          kk--;
        }
      }
      if (fabs(phi) > 1e200) {
        phi *= 1e-200;
        phi_plus *= 1e-200;
        n_done++;
      }
    }
    if (pass == 0) {
      n_rescale_total = n_done;
      scaling = phi0 / phi;
    }
  }
}

// chi_at_phimin[l] = hyperspherical_get_xmin_from_approx (hyperspherical.c:1419-1450), K=0, nu=1
__global__ void k_chi_at_phimin(double* __restrict__ out, const int* __restrict__ lvec, int nl, double phiminabs) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nl) return;
  double lph = lvec[i] + 0.5;
  double lhs = 1.0 / lph * log(2 * phiminabs * lph);
  double alpha = -2.0 * lhs / 5.0 * (1.0 + 2.0 * cosh(1.0 / 3.0 * acosh(1.0 + 375.0 / (16.0 * lhs * lhs))));
  out[i] = lph / cosh(alpha);
}

int cpt_bessel_build(cpt_handle* h, const int* l, int nl, double xmax) {
  const cpt_config& c = h->cfg;
  if (nl < 1) return cpt_fail(h, CPT_ERR_INVALID, "empty l list");
  for (int i = 1; i < nl; i++)
    if (l[i] <= l[i - 1]) return cpt_fail(h, CPT_ERR_INVALID, "l list must be strictly increasing");
  if (l[0] < 0) return cpt_fail(h, CPT_ERR_INVALID, "negative l");
  bool hit = (h->bes_xmax == xmax) && ((int)h->bes_l.size() == nl) && (memcmp(h->bes_l.data(), l, nl * sizeof(int)) == 0);
  if (hit) return CPT_OK;
  h->geo_tr_valid = false;   // the cached transfer geometry refers to the table (and l list) replaced here
  int rc;
  const double PI = 3.1415926535897932384626433832795;
  const double xmin = c.hyper_x_min;
  int nx = (int)((xmax - xmin) * c.hyper_sampling_flat / (2 * PI));
  if (nx < 2) nx = 2;
  const double dx = (xmax - xmin) / (nx - 1.0);
  const int lmax = l[nl - 1];
  const double xfwd = sqrt(lmax * (lmax + 1.0));
  const int xfwdidx = (int)((xfwd - xmin) / dx);
  if ((rc = cpt_reserve(h, &h->d_bes, &h->bes_cap, (size_t)nl * nx))) return rc;
  if ((rc = cpt_reserve(h, &h->d_l, &h->grid_cap_l, (size_t)nl))) return rc;
  if (h->d_chi_min) { (void)hipFree(h->d_chi_min); h->d_chi_min = nullptr; }
  CPT_HIP(h, hipMalloc((void**)&h->d_chi_min, nl * sizeof(double)));
  if ((rc = cpt_upload(h, h->d_l, l, nl * sizeof(int)))) return rc;
  hipLaunchKernelGGL(k_bessel, dim3((nx + 63) / 64), dim3(64), 0, h->stream, h->d_bes, h->d_l, nl, nx, xmin, dx, xfwdidx);
  CPT_HIP(h, hipGetLastError());
  hipLaunchKernelGGL(k_chi_at_phimin, dim3((nl + 63) / 64), dim3(64), 0, h->stream, h->d_chi_min, h->d_l, nl,
                     c.hyper_phi_min_abs);
  CPT_HIP(h, hipGetLastError());
  h->bes_l.assign(l, l + nl);
  h->bes_xmax = xmax;
  h->bes_nx = nx;
  h->bes_dx = dx;
  return CPT_OK;
}

// ---------------------------------------------------------------------------------------------
// Spline of the sources along k (sources k-major: S[tp][k][tau]); one thread per (tp, tau) row.
// splc = { c[nk] (forward elimination factor, identical for every row), sig[nk], p[nk] } computed on the host
// from the k grid alone.  Forward sweep stores u[i] in dd, backward sweep finishes in place.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) k_source_spline(const double* __restrict__ S, double* __restrict__ dd,
                                                      const double* __restrict__ x, const double* __restrict__ splc,
                                                      int ntp, int nk, int ntau) {
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= ntp * ntau) return;
  const int tp = r / ntau, it = r - tp * ntau;
  const double* y = S + (size_t)tp * nk * ntau + it;  // y[i] at y[i*ntau]
  double* d = dd + (size_t)tp * nk * ntau + it;
  const double* cc = splc;
  const double* sg = splc + nk;
  const double* pp = splc + 2 * nk;
  const size_t st = ntau;
  const int n = nk;
  double y0 = y[0], y1 = y[st], y2 = y[2 * st];
  double x0 = x[0], x1 = x[1], x2 = x[2];
  double dy_first = ((x2 - x0) * (x2 - x0) * (y1 - y0) - (x1 - x0) * (x1 - x0) * (y2 - y0)) / ((x2 - x0) * (x1 - x0) * (x2 - x1));
  double u = (3. / (x1 - x0)) * ((y1 - y0) / (x1 - x0) - dy_first);
  d[0] = u;
  double ym = y0, yc = y1;
  double xm = x0, xc = x1;
  // The recurrence in u is sequential, the loads are not: fetch a block of rows first (one HBM round trip per block instead
  // of one per k), then run the arithmetic on registers.  46 waves cannot hide the latency by occupancy.
  constexpr int BLK = 16;
  int i = 1;
  for (; i + BLK <= n - 1; i += BLK) {
    double yb[BLK], xb[BLK], sb[BLK], pb[BLK];
#pragma unroll
    for (int j = 0; j < BLK; j++) { yb[j] = y[(size_t)(i + 1 + j) * st]; xb[j] = x[i + 1 + j]; sb[j] = sg[i + j]; pb[j] = pp[i + j]; }
#pragma unroll
    for (int j = 0; j < BLK; j++) {
      const double yn = yb[j], xn = xb[j];
      const double ui = (yn - yc) / (xn - xc) - (yc - ym) / (xc - xm);
      u = (6.0 * ui / (xn - xm) - sb[j] * u) / pb[j];
      d[(size_t)(i + j) * st] = u;
      ym = yc; yc = yn; xm = xc; xc = xn;
    }
  }
  for (; i < n - 1; i++) {
    double yn = y[(size_t)(i + 1) * st];
    double xn = x[i + 1];
    double ui = (yn - yc) / (xn - xc) - (yc - ym) / (xc - xm);
    u = (6.0 * ui / (xn - xm) - sg[i] * u) / pp[i];
    d[(size_t)i * st] = u;
    ym = yc; yc = yn; xm = xc; xc = xn;
  }
  // here ym = y[n-2], yc = y[n-1], xm = x[n-2], xc = x[n-1]
  double y3 = y[(size_t)(n - 3) * st], x3 = x[n - 3];
  double dy_last = ((x3 - xc) * (x3 - xc) * (ym - yc) - (xm - xc) * (xm - xc) * (y3 - yc)) / ((x3 - xc) * (xm - xc) * (x3 - xm));
  double un = (3. / (xc - xm)) * (dy_last - (yc - ym) / (xc - xm));
  double ddn = (un - 0.5 * u) / (0.5 * cc[n - 2] + 1.0);
  d[(size_t)(n - 1) * st] = ddn;
  i = n - 2;
  for (; i - BLK + 1 >= 0; i -= BLK) {
    double ub[BLK], cb[BLK];
#pragma unroll
    for (int j = 0; j < BLK; j++) { ub[j] = d[(size_t)(i - j) * st]; cb[j] = cc[i - j]; }
#pragma unroll
    for (int j = 0; j < BLK; j++) {
      ddn = cb[j] * ddn + ub[j];
      d[(size_t)(i - j) * st] = ddn;
    }
  }
  for (; i >= 0; i--) {
    ddn = cc[i] * ddn + d[(size_t)i * st];
    d[(size_t)i * st] = ddn;
  }
}

// ---------------------------------------------------------------------------------------------
// LOS kernel
// ---------------------------------------------------------------------------------------------
struct LosParams {
  const double* src;  // [tp][nk][ntau]
  const double* dd;
  const double* k;
  const double* tau;
  const double* q;
  const int* l;
  const int* ik;  // bracketing k index per q (-1: q beyond k_size_cl -> zeros)
  const double2* bes;
  const double* chi_min;
  double* out;  // [tt][nl][nq]
  unsigned long long* work;
  int nk, ntau, nq, nl, nx;
  double bes_xmin, bes_dx, bes_xmax;
  int tts[5], tps[5];   // slots 0..3: scalars t0,t1,t2,e | tensors t2,e,b,-;  slot 4: lensing potential (scalars)
  int tensors;          // radial functions of the tensor types (tm.cpp:3494-3529) in slots 0..2
  double dk[4];
  double tau0, tau_rec, ra_rec, t0mt_cut, late_l, l_switch_limber;
  double lcmb_fac_rescale, lcmb_tilt, lcmb_pivot;
  int imin_lcmb;  // first tau index kept by the lensing source (tau > tau_rec), tm.cpp:1366-1371
  int i_cut;      // last index with tau0-tau >= tau0-tau_cut (-1 if none), tm.cpp:2837-2845
};

__device__ static inline double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// last index i in [lo, hi] with a[i] >= val, for a[] decreasing; lo-1 if none
__device__ static inline int last_ge(const double* a, int lo, int hi, double val) {
  int L = lo - 1, H = hi + 1;  // invariant: a[L] >= val (or L = lo-1), a[H] < val (or H = hi+1)
  while (H - L > 1) {
    int m = (L + H) >> 1;
    if (a[m] >= val) L = m; else H = m;
  }
  return L;
}

__device__ static inline double parabola(double x1, double x2, double x3, double x, double y1, double y2, double y3) {
  double b = ((y1 - y2) * (x3 - x2) * (x3 + x2) - (y3 - y2) * (x1 - x2) * (x1 + x2)) / (x1 - x2) / (x3 - x2) / (x3 - x1);
  double a = (y1 - y2 - b * (x1 - x2)) / (x1 - x2) / (x1 + x2);
  double c = y2 - b * x2 - a * x2 * x2;
  return a * x * x + b * x + c;
}

__global__ void __launch_bounds__(256) k_los(LosParams P) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int ntau = P.ntau, nl = P.nl, nq = P.nq;
  double* t0mt = lds;             // [ntau] tau0 - tau
  double* wt = lds + ntau;        // [ntau] trapezoidal weights of the full grid, arrays.c:2856-2879
  double* S = lds + 2 * ntau;     // [5][ntau] transfer sources (t0,t1,t2,e,lcmb); lcmb kept at its original index
  int& next_l = *(int*)(lds + 7 * ntau);  // work queue over l, shared by the waves of the block

  const int iq = nq - 1 - blockIdx.x;  // heaviest wavenumbers first
  const double q = P.q[iq], k = q;     // flat space: k = q (tm.cpp:1106-1167)
  const int ik = P.ik[iq];
  const int tid = threadIdx.x;

  if (ik < 0) {  // beyond the largest k used for C_l's: tm.cpp:1541, 1696-1710
    for (int e = tid; e < 5 * nl; e += blockDim.x) {
      int t = e / nl, il = e - t * nl;
      if (P.tts[t] >= 0) P.out[((size_t)P.tts[t] * nl + il) * nq + iq] = 0.;
    }
    return;
  }

  // ---- stage the sources of this q in LDS: tm.cpp:1767-1820 + 1845-2109 ----
  const double k_lo = P.k[ik], k_hi = P.k[ik + 1];
  const double h = k_hi - k_lo;
  const double b = (k - k_lo) / h, a = 1. - b;
  const double ca = (a * a * a - a), cb = (b * b * b - b), h2 = h * h / 6.0;
  const double lfac = (P.tts[4] >= 0) ? P.lcmb_fac_rescale * pow(k / P.lcmb_pivot, P.lcmb_tilt) : 0.;
  if (tid == 0) next_l = 0;
  for (int i = tid; i < ntau; i += blockDim.x) {
    const double tau = P.tau[i];
    const double tm = P.tau0 - tau;
    t0mt[i] = tm;
    double w;
    if (i == 0) w = 0.5 * (tm - (P.tau0 - P.tau[1]));
    else if (i == ntau - 1) w = 0.5 * ((P.tau0 - P.tau[ntau - 2]) - tm);
    else w = 0.5 * ((P.tau0 - P.tau[i - 1]) - (P.tau0 - P.tau[i + 1]));
    wt[i] = w;
#pragma unroll
    for (int t = 0; t < 5; t++) {
      double v = 0.;
      if (P.tts[t] >= 0) {
        const size_t o = ((size_t)P.tps[t] * P.nk + ik) * ntau + i;
        v = a * P.src[o] + b * P.src[o + ntau] + (ca * P.dd[o] + cb * P.dd[o + ntau]) * h2;
        if (t == 4) {
          double resc = (i == ntau - 1 || i < P.imin_lcmb) ? 0. : (P.tau_rec - tau) / (P.tau0 - tau) / (P.tau0 - P.tau_rec);
          v = v * resc * lfac;
        }
      }
      S[t * ntau + i] = v;
    }
  }
  __syncthreads();

  const int lane = tid & 63;
  unsigned long long n_int = 0, n_samp = 0, n_fused = 0;
  const double* S_l = S + 4 * ntau;
  const int imin = P.imin_lcmb;
  const int tsz_l = ntau - imin;  // tau_size of the lensing source

  for (;;) {
    int il = 0;
    if (lane == 0) il = atomicAdd(&next_l, 1);
    il = __shfl(il, 0, 64);
    if (il >= nl) break;
    const double l = (double)P.l[il];

    // ---- per-type decisions (uniform across the wave) ----
    const double tmin_bessel = P.chi_min[il] / k;  // tm.cpp:2773
    const bool late = l > P.late_l;                // tm.cpp:3229-3245 (applies to t1,t2,e)
    int imax_t[5];
    bool trunc_t[5];
    int imax_all = -1;
    // CMB types share the full time grid
    int imax_b = (tmin_bessel >= t0mt[0]) ? -1 : last_ge(t0mt, 0, ntau - 1, tmin_bessel);
#pragma unroll
    for (int t = 0; t < 4; t++) {
      imax_t[t] = -1;
      trunc_t[t] = false;
      if (P.tts[t] < 0) continue;
      if (l < (k - P.dk[t]) * P.ra_rec) continue;  // tm.cpp:3187-3196 -> zero
      if (imax_b < 0) continue;                    // tm.cpp:2793-2796 no overlap -> zero
      int im = imax_b;
      const double* St = S + t * ntau;
      while (im >= 0 && St[im] == 0.) im--;        // tm.cpp:2826-2832
      if (im >= 0 && late && t != 0) im = (P.i_cut < im) ? P.i_cut : im;  // tm.cpp:2834-2845
      if (im < 0) continue;
      imax_t[t] = im;
      trunc_t[t] = (im != ntau - 1) && (im == imax_b);  // tm.cpp:2883
      imax_all = im > imax_all ? im : imax_all;
    }
    // lensing potential: Limber above l_switch_limber (tm.cpp:2661-2675), integral on the truncated grid otherwise
    imax_t[4] = -1;
    trunc_t[4] = false;
    bool lcmb_limber = false;
    if (P.tts[4] >= 0) {
      if (l > P.l_switch_limber) lcmb_limber = true;
      else if (tmin_bessel < t0mt[imin]) {
        int imb = last_ge(t0mt, imin, ntau - 1, tmin_bessel);
        int im = imb;
        while (im >= imin && S_l[im] == 0.) im--;
        if (im >= imin) {
          imax_t[4] = im;
          trunc_t[4] = (im != ntau - 1) && (im == imb);
          imax_all = im > imax_all ? im : imax_all;
        }
      }
    }

    // ---- fused quadrature over tau: lanes stride the samples ----
    double acc[5] = {0., 0., 0., 0., 0.};
    if (imax_all >= 0) {
      const double lxlp1 = l * (l + 1.0);
      const double fac_e = sqrt(3.0 / 8.0 * (l + 2.0) * (l + 1.0) * l * (l - 1.0));
      const double2* bl = P.bes + (size_t)il * P.nx;
      const double dx = P.bes_dx;
      for (int i = lane; i <= imax_all; i += 64) {
        const double tm = t0mt[i];
        const double x = k * tm;  // chi = k (tau0 - tau), tm.cpp:1735
        double Phi = 0., dPhi = 0., d2Phi = 0.;
        if (x >= P.bes_xmin && x <= P.bes_xmax) {
          // Hermite-4 on the uniform grid (hermite4_interpolation_csource.h:80-160), K = 0, beta = 1
          int idx = (int)((x - P.bes_xmin) / dx) + 1;
          idx = idx < 1 ? 1 : idx;
          idx = idx > P.nx - 1 ? P.nx - 1 : idx;
          const double2 vm = bl[idx - 1], vp = bl[idx];
          const double xm = P.bes_xmin + (idx - 1) * dx, xp = P.bes_xmin + idx * dx;
          const double ym = vm.x, dym = vm.y, yp = vp.x, dyp = vp.y;
          const double cotm = 1.0 / xm, cotp = 1.0 / xp;
          const double ism2 = cotm * cotm, isp2 = cotp * cotp;
          const double d2ym = -2 * dym * cotm + ym * (lxlp1 * ism2 - 1.0);
          const double d2yp = -2 * dyp * cotp + yp * (lxlp1 * isp2 - 1.0);
          const double d3ym = -2 * cotm * d2ym - 2 * ym * lxlp1 * cotm * ism2 + dym * (-1.0 + (2 + lxlp1) * ism2);
          const double d3yp = -2 * cotp * d2yp - 2 * yp * lxlp1 * cotp * isp2 + dyp * (-1.0 + (2 + lxlp1) * isp2);
          const double a0 = dym * dx, a1 = -2 * dym * dx - dyp * dx - 3 * ym + 3 * yp, a2 = dym * dx + dyp * dx + 2 * ym - 2 * yp;
          const double b0 = d2ym * dx, b1 = -2 * d2ym * dx - d2yp * dx - 3 * dym + 3 * dyp, b2 = d2ym * dx + d2yp * dx + 2 * dym - 2 * dyp;
          const double c0 = d3ym * dx, c1 = -2 * d3ym * dx - d3yp * dx - 3 * d2ym + 3 * d2yp, c2 = d3ym * dx + d3yp * dx + 2 * d2ym - 2 * d2yp;
          const double z = (x - xm) / dx, z2 = z * z, z3 = z2 * z;
          Phi = ym + a0 * z + a1 * z2 + a2 * z3;
          dPhi = dym + b0 * z + b1 * z2 + b2 * z3;
          d2Phi = d2ym + c0 * z + c1 * z2 + c2 * z3;
        }
        const double w = wt[i];
        const double ix = 1.0 / x;
        // radial functions, tm.cpp:3413-3445 with sqrt_absK_over_k = 1, s2 = 1
        double R[4] = {Phi, dPhi, 0.5 * (3. * d2Phi + Phi), fac_e * ix * ix * Phi};
        if (P.tensors) {  // tm.cpp:3494-3529 with cscK = cotK = 1/chi, K = 0: tensor temperature, E and B polarisation
          R[0] = fac_e * ix * ix * Phi;
          R[1] = 0.25 * (d2Phi + 4.0 * ix * dPhi - (1.0 - 2.0 * ix * ix) * Phi);
          R[2] = 0.5 * (dPhi + 2.0 * ix * Phi);
          R[3] = 0.;
        }
#pragma unroll
        for (int t = 0; t < 4; t++) {
          if (i <= imax_t[t]) {
            const double s = S[t * ntau + i];
            double term = s * R[t] * w;
            // Bessel-truncation triangle, tm.cpp:2883-2887
            if (trunc_t[t] && i == imax_t[t]) term -= 0.5 * (t0mt[i + 1] - tmin_bessel) * R[t] * s;
            acc[t] += term;
          }
        }
        if (i >= imin && i <= imax_t[4]) {
          const double s = S_l[i];
          const double wl = (i == imin) ? 0.5 * (tm - t0mt[i + 1]) : w;
          double term = s * Phi * wl;
          if (trunc_t[4] && i == imax_t[4]) term -= 0.5 * (t0mt[i + 1] - tmin_bessel) * Phi * s;
          acc[4] += term;
        }
      }
#pragma unroll
      for (int t = 0; t < 5; t++) acc[t] = wave_sum(acc[t]);
    }

    if (lane == 0) {
      if (lcmb_limber) {
        // tm.cpp:2912-2969 (flat, radial type T0) + transfer_limber_interpolate :3054-3109 on the truncated grid
        double res = 0.;
        const double tl = (l + 0.5) / q;
        if (!(tl > t0mt[imin] || tl < t0mt[ntau - 1])) {
          int it = 1;  // index in the truncated arrays
          // first it in [1, tsz-2] with t0mt_l[it] <= tl  (t0mt decreasing)
          int j = last_ge(t0mt, imin, ntau - 1, tl) - imin;  // last index with t0mt >= tl
          // reference: it=1; while (t0mt[it] > tl && it < tsz-2) it++
          it = j + 1;                          // first index with t0mt < tl ...
          if (j >= 0 && t0mt[imin + j] == tl) it = j;  // ... or == tl (the while stops on equality)
          if (it < 1) it = 1;
          if (it > tsz_l - 2) it = tsz_l - 2;
          const int o = imin + it;
          double y3 = (it < tsz_l - 2) ? S_l[o + 1] * t0mt[o + 1] : S_l[o] * t0mt[o];
          double Sv = parabola(t0mt[o - 1], t0mt[o], t0mt[o + 1], tl, S_l[o - 1] * t0mt[o - 1], S_l[o] * t0mt[o], y3);
          double IPhiFlat = sqrt(3.1415926535897932384626433832795 / (2. * l)) * (1. - 0.25 / l + 1. / 32. / (l * l));
          res = IPhiFlat * Sv / (l + 0.5);
        }
        acc[4] = res;
      }
#pragma unroll
      for (int t = 0; t < 5; t++)
        if (P.tts[t] >= 0) P.out[((size_t)P.tts[t] * nl + il) * nq + iq] = acc[t];
      for (int t = 0; t < 5; t++)
        if (imax_t[t] >= 0) { n_int++; n_samp += (t == 4) ? (imax_t[t] - imin + 1) : (imax_t[t] + 1); }
      if (imax_all >= 0) n_fused += imax_all + 1;
    }
  }
  if (lane == 0 && P.work) {
    atomicAdd(&P.work[0], n_int);
    atomicAdd(&P.work[1], n_samp);
    atomicAdd(&P.work[2], n_fused);
  }
}

// =============================================================================================
// Closed space (K > 0): every q below the flat-approximation threshold has its own table of hyperspherical Bessel
// functions Phi_l^nu(chi), nu = q/sqrt(K) integer (transfer_update_HIS, tm.cpp:3777-3887 -> hyperspherical_HIS_create,
// hyperspherical.c:11-246).  The reference builds and frees one table per q task; here ALL tables are built by one
// launch (one thread per (q, x_j), recurrence over l in registers, only the listed l stored) into one HBM buffer
// - sum_q nl(q) nx(q) 16 B, ~2 GB for the default-precision config - and the LOS kernel of that q reads its slice.
// =============================================================================================
struct HisDesc {
  double nu, dx;
  int nl, nx, xfwdidx, L, special;  // special: nu == lmax+1, Phi_{lmax+1} := 0 (hyperspherical.c:146-155)
  unsigned long long off, trig_off; // offsets (in double2) of the table [nl][nx] {Phi, Phi'} and of {sinK, cotK}[nx]
};

// Lentz continued fraction for Phi_l'/Phi_l, K = +-1 (hyperspherical.c:677-716); false if not converged
__device__ static inline bool cf1_curved(int K, int l, double beta, double cotK, double* CF, int* isign) {
  const double tiny = 1e-100, reltol = 2.220446049250313e-16, beta2 = beta * beta;
  const int maxiter = (K == 1) ? (int)(beta - l - 10) : 1000000;
  double bj = l * cotK, fj = bj, Cj = bj, Dj = 0.0;
  int sgn = 1;
  for (int j = 1; j <= maxiter; j++) {
    const double sqrttmp = sqrt(beta2 - K * (l + j + 1.) * (l + j + 1.));
    double aj = -sqrt(beta2 - K * (double)(l + j) * (l + j)) / sqrttmp;
    if (j == 1) aj = sqrt(beta2 - K * (l + 1.) * (l + 1.)) * aj;
    bj = (2 * (l + j) + 1) / sqrttmp * cotK;
    Dj = bj + aj * Dj;
    if (Dj == 0.0) Dj = tiny;
    Cj = bj + aj / Cj;
    if (Cj == 0.0) Cj = tiny;
    Dj = 1.0 / Dj;
    const double Delj = Cj * Dj;
    fj = fj * Delj;
    if (Dj < 0) sgn = -sgn;
    if (fabs(Delj - 1.0) < reltol) { *CF = fj; *isign = sgn; return true; }
  }
  *isign = sgn;
  return false;
}

// Phi_l'/Phi_l from the Gegenbauer polynomial C_n^{l+1}(cos chi), n = nu - l - 1 (hyperspherical.c:718-780)
__device__ static inline void cf1_gegenbauer(int l, int beta, double sinK, double cotK, double* CF) {
  const int n = beta - l - 1;
  const double alpha = l + 1, x = sinK * cotK;
  double G, dG;
  if (n <= 0) { G = 1; dG = 0; }
  else if (n == 1) { G = 2 * alpha * x; dG = 2 * alpha; }
  else if (n == 2) { G = -alpha + 2 * alpha * (1 + alpha) * x * x; dG = 4 * x * alpha * (1 + alpha); }
  else if (n == 3) {
    G = -2 * alpha * (1 + alpha) * x + 4.0 / 3.0 * alpha * (1 + alpha) * (2 + alpha) * x * x * x;
    dG = 2 * alpha * (1 + alpha) * (2 * (2 + alpha) * x * x - 1);
  } else {
    G = 0.0;
    double Gkm2 = -alpha + 2 * alpha * (1 + alpha) * x * x;
    double Gkm1 = -2 * alpha * (1 + alpha) * x + 4.0 / 3.0 * alpha * (1 + alpha) * (2 + alpha) * x * x * x;
    for (int k = 4; k <= n; k++) {
      G = (2 * (k + alpha - 1) * x * Gkm1 - (k + 2 * alpha - 2) * Gkm2) / k;
      if (fabs(G) > 1e200) { Gkm2 = Gkm1 / 1e200; G = G / 1e200; Gkm1 = G; }
      else { Gkm2 = Gkm1; Gkm1 = G; }
    }
    dG = (-n * x * G + (n + 2 * alpha - 1) * Gkm2) / (1.0 - x * x);
  }
  *CF = l * cotK - sinK * dG / G;
}

// one thread per (own-table q, abscissa x_j): blockIdx.y = q index, blockIdx.x * 64 + lane = j
__global__ void __launch_bounds__(64) k_his_curved(const HisDesc* __restrict__ desc, const int* __restrict__ lvec, double xmin, int sgnK,
                                                   double2* __restrict__ tab, double2* __restrict__ trig) {
  const HisDesc D = desc[blockIdx.y];
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= D.nx || D.nl <= 0) return;
  const int nl = D.nl, nx = D.nx, L = D.L;
  const double beta = D.nu, beta2 = beta * beta;
  const double x = xmin + j * D.dx;
  const double sinK = (sgnK == 1) ? sin(x) : sinh(x), cotK = (sgnK == 1) ? 1.0 / tan(x) : 1.0 / tanh(x);
  trig[D.trig_off + j] = make_double2(sinK, cotK);
  double2* out = tab + D.off;
  auto sk = [&](int l) { return sqrt(beta2 - sgnK * (double)l * l); };   // sqrtK[l], hyperspherical.c:104-121
  if (j >= D.xfwdidx) {
    // forward recurrence, hyperspherical.c:472-514
    double pm = sin(beta * x) / (beta * sinK);                                 // P[0]
    double pc = pm * (cotK - beta / tan(beta * x)) / sk(1);                    // P[1]
    int kk = 0;
    while (kk < nl && lvec[kk] == 0) { out[(size_t)kk * nx + j] = make_double2(pm, -sk(1) * pc); kk++; }
    for (int l = 1; l <= L; l++) {
      // P[l+1] exists up to l+1 = L; beyond (the special case l = L = lmax) it is 0 by definition
      const double s1 = (l < L) ? sk(l + 1) : 0.;
      const double pn = (l < L) ? ((2 * l + 1) * cotK * pc - pm * sk(l)) / s1 : 0.;
      if (kk < nl && lvec[kk] == l) { out[(size_t)kk * nx + j] = make_double2(pc, l * cotK * pc - s1 * pn); kk++; }
      pm = pc; pc = pn;
    }
    return;
  }
  // backward recurrence, hyperspherical.c:517-603 (closed: phi1 = 1, no sign bookkeeping)
  const double phi0 = sin(beta * x) / (beta * sinK);
  double phipr1 = 0., phi1 = 1.0;
  int isign = 1;
  if (sgnK == 1) {
    bool ok = false;
    if (beta > 1.5 * L) ok = cf1_curved(1, L, beta, cotK, &phipr1, &isign);
    if (!ok) cf1_gegenbauer(L, (int)(beta + 0.2), sinK, cotK, &phipr1);
  } else {
    cf1_curved(-1, L, beta, cotK, &phipr1, &isign);
    phi1 = (double)isign;
    phipr1 *= phi1;
  }
  const int l_align = L - L % 8;
  int n_rescale_total = 0;
  double scaling = 0.;
  for (int pass = 0; pass < 2; pass++) {
    double phi = phi1, ppts = L * cotK * phi1 - phipr1;  // P[l], sqrtK[l+1] P[l+1]
    int kk = nl - 1, n_done = 0;
    auto emit = [&](int l) {  // P[l] = phi, sqrtK[l+1] P[l+1] = ppts, as they stand now
      if (pass == 1 && kk >= 0 && lvec[kk] == l) {
        double v = phi, vp = ppts;
        const int later = n_rescale_total - n_done;
        for (int r = 0; r < later && r < 8; r++) { v *= 1e-200; vp *= 1e-200; }
        if (later >= 8) { v = 0.; vp = 0.; }
        v *= scaling; vp *= scaling;
        out[(size_t)kk * nx + j] = make_double2(v, l * cotK * v - vp);
        kk--;
      }
    };
    if (D.special) { const double keep = ppts; ppts = 0.; emit(L); ppts = keep; }   // l = lmax = L is in the list, Phi_{L+1} := 0
    int l = L;
    for (; l > l_align; l--) {
      const double pmv = ((2 * l + 1) * cotK * phi - ppts) / sk(l);
      ppts = phi * sk(l);
      phi = pmv;
      emit(l - 1);
    }
    for (int l_ini = l_align; l_ini > 0; l_ini -= 8) {
      for (l = l_ini; l > l_ini - 8; l--) {
        const double pmv = ((2 * l + 1) * cotK * phi - ppts) / sk(l);
        ppts = phi * sk(l);
        phi = pmv;
        emit(l - 1);
      }
      if (fabs(phi) > 1e200) { phi *= 1e-200; ppts *= 1e-200; n_done++; }
    }
    if (pass == 0) { n_rescale_total = n_done; scaling = phi0 / phi; }
  }
}

struct LosClosedParams {
  LosParams b;               // everything of the flat kernel (the flat table serves q >= index_q_flat)
  const double* kq;          // k(q) = sqrt(q^2 - K), tm.cpp:1106-1167
  const HisDesc* desc;       // [index_q_flat]
  const double2* his;        // per-q tables
  const double2* trig;       // per-q {sinK, cotK} at the table nodes
  double K, sqrtK, his_xmin, phiminabs;   // sqrtK = sqrt|K|
  int index_q_flat, sgnK;
};

// Hermite interpolation of order 6 on a per-q table (hermite6_interpolation_csource.h), stateless per sample
__device__ static inline void hermite6_curved(const double2* __restrict__ tl, const double2* __restrict__ tg, int nx, double xmin, double dx,
                                              double beta, int sgnK, double lxlp1, double x, double* Phi, double* dPhi, double* d2Phi) {
  const double xmax = xmin + (nx - 1) * dx;
  if (x < xmin || x > xmax) { *Phi = *dPhi = *d2Phi = 0.; return; }
  int idx = (int)((x - xmin) / dx) + 1;
  idx = idx < 1 ? 1 : idx;
  idx = idx > nx - 1 ? nx - 1 : idx;
  const double KmB2 = (double)sgnK - beta * beta, dx2 = dx * dx;
  double y[2], dy[2], d2y[2], d3y[2], d4y[2];
#pragma unroll
  for (int s = 0; s < 2; s++) {
    const double2 v = tl[idx - 1 + s], g = tg[idx - 1 + s];
    const double cot = g.y, is2 = 1.0 / (g.x * g.x);
    y[s] = v.x; dy[s] = v.y;
    d2y[s] = -2 * dy[s] * cot + y[s] * (lxlp1 * is2 + KmB2);
    d3y[s] = -2 * cot * d2y[s] - 2 * y[s] * lxlp1 * cot * is2 + dy[s] * (KmB2 + (2 + lxlp1) * is2);
    d4y[s] = -2 * cot * d3y[s] + d2y[s] * (KmB2 + (4 + lxlp1) * is2) + dy[s] * (-4 * (1 + lxlp1) * cot * is2) +
             y[s] * (2 * lxlp1 * is2 * (2 * cot * cot + is2));
  }
  const double ym = y[0], yp = y[1], dym = dy[0], dyp = dy[1], d2ym = d2y[0], d2yp = d2y[1], d3ym = d3y[0], d3yp = d3y[1],
               d4ym = d4y[0], d4yp = d4y[1];
  const double a1 = dym * dx, a2 = 0.5 * d2ym * dx2, a3 = (-1.5 * d2ym + 0.5 * d2yp) * dx2 - (6 * dym + 4 * dyp) * dx - 10 * (ym - yp),
               a4 = (1.5 * d2ym - d2yp) * dx2 + (8 * dym + 7 * dyp) * dx + 15 * (ym - yp),
               a5 = (-0.5 * d2ym + 0.5 * d2yp) * dx2 - 3 * (dym + dyp) * dx - 6 * (ym - yp);
  const double b1 = d2ym * dx, b2 = 0.5 * d3ym * dx2, b3 = (-1.5 * d3ym + 0.5 * d3yp) * dx2 - (6 * d2ym + 4 * d2yp) * dx - 10 * (dym - dyp),
               b4 = (1.5 * d3ym - d3yp) * dx2 + (8 * d2ym + 7 * d2yp) * dx + 15 * (dym - dyp),
               b5 = (-0.5 * d3ym + 0.5 * d3yp) * dx2 - 3 * (d2ym + d2yp) * dx - 6 * (dym - dyp);
  const double c1 = d3ym * dx, c2 = 0.5 * d4ym * dx2, c3 = (-1.5 * d4ym + 0.5 * d4yp) * dx2 - (6 * d3ym + 4 * d3yp) * dx - 10 * (d2ym - d2yp),
               c4 = (1.5 * d4ym - d4yp) * dx2 + (8 * d3ym + 7 * d3yp) * dx + 15 * (d2ym - d2yp),
               c5 = (-0.5 * d4ym + 0.5 * d4yp) * dx2 - 3 * (d3ym + d3yp) * dx - 6 * (d2ym - d2yp);
  const double z = (x - (xmin + (idx - 1) * dx)) / dx, z2 = z * z, z3 = z2 * z, z4 = z2 * z2, z5 = z2 * z3;
  *Phi = ym + a1 * z + a2 * z2 + a3 * z3 + a4 * z4 + a5 * z5;
  *dPhi = dym + b1 * z + b2 * z2 + b3 * z3 + b4 * z4 + b5 * z5;
  *d2Phi = d2ym + c1 * z + c2 * z2 + c3 * z3 + c4 * z4 + c5 * z5;
}

// Hermite-4 on the flat table (same arithmetic as in k_los)
__device__ static inline void hermite4_flat(const double2* __restrict__ bl, int nx, double xmin, double dx, double xmax, double lxlp1, double x,
                                            double* Phi, double* dPhi, double* d2Phi) {
  *Phi = *dPhi = *d2Phi = 0.;
  if (!(x >= xmin && x <= xmax)) return;
  int idx = (int)((x - xmin) / dx) + 1;
  idx = idx < 1 ? 1 : idx;
  idx = idx > nx - 1 ? nx - 1 : idx;
  const double2 vm = bl[idx - 1], vp = bl[idx];
  const double xm = xmin + (idx - 1) * dx, xp = xmin + idx * dx;
  const double ym = vm.x, dym = vm.y, yp = vp.x, dyp = vp.y;
  const double cotm = 1.0 / xm, cotp = 1.0 / xp, ism2 = cotm * cotm, isp2 = cotp * cotp;
  const double d2ym = -2 * dym * cotm + ym * (lxlp1 * ism2 - 1.0), d2yp = -2 * dyp * cotp + yp * (lxlp1 * isp2 - 1.0);
  const double d3ym = -2 * cotm * d2ym - 2 * ym * lxlp1 * cotm * ism2 + dym * (-1.0 + (2 + lxlp1) * ism2);
  const double d3yp = -2 * cotp * d2yp - 2 * yp * lxlp1 * cotp * isp2 + dyp * (-1.0 + (2 + lxlp1) * isp2);
  const double a0 = dym * dx, a1 = -2 * dym * dx - dyp * dx - 3 * ym + 3 * yp, a2 = dym * dx + dyp * dx + 2 * ym - 2 * yp;
  const double b0 = d2ym * dx, b1 = -2 * d2ym * dx - d2yp * dx - 3 * dym + 3 * dyp, b2 = d2ym * dx + d2yp * dx + 2 * dym - 2 * dyp;
  const double c0 = d3ym * dx, c1 = -2 * d3ym * dx - d3yp * dx - 3 * d2ym + 3 * d2yp, c2 = d3ym * dx + d3yp * dx + 2 * d2ym - 2 * d2yp;
  const double z = (x - xm) / dx, z2 = z * z, z3 = z2 * z;
  *Phi = ym + a0 * z + a1 * z2 + a2 * z3;
  *dPhi = dym + b0 * z + b1 * z2 + b2 * z3;
  *d2Phi = d2ym + c0 * z + c1 * z2 + c2 * z3;
}

// hyperspherical_get_xmin_from_approx, K = +-1 (hyperspherical.c:1419-1450)
__device__ static inline double xmin_from_approx_curved(int sgnK, double l, double nu, double phiminabs) {
  const double lph = l + 0.5, lhs = 1.0 / lph * log(2 * phiminabs * lph);
  const double alpha = -2.0 * lhs / 5.0 * (1.0 + 2.0 * cosh(1.0 / 3.0 * acosh(1.0 + 375.0 / (16.0 * lhs * lhs))));
  double x = lph / cosh(alpha) / nu;
  if (sgnK == 1) x *= asin(l / nu) / (l / nu);
  else { x *= asinh(l / nu) / (l / nu); x *= ((nu + 0.4567) / (nu + 1.24) - 2.209e-3); }
  return x;
}

// LOS kernel for closed space, scalar types (transfer_compute_for_each_q tm.cpp:1488-1715 with the sgnK = 1 branches of
// transfer_radial_coordinates :1717-1749, transfer_radial_function :3274-3445, transfer_sources :1905-1964,
// transfer_integrate :2762-2792, transfer_limber :2930-2968).  Same organisation as k_los: one workgroup per q, sources
// in LDS, waves pull multipoles from an LDS queue, the CMB types share one interpolation of Phi_l.
__global__ void __launch_bounds__(256) k_los_curved(LosClosedParams C) {
  const LosParams& P = C.b;
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int ntau = P.ntau, nl = P.nl, nq = P.nq;
  double* t0mt = lds;
  double* wt = lds + ntau;
  double* S = lds + 2 * ntau;        // [5][ntau]
  double* csc2 = lds + 7 * ntau;     // [ntau] cscK_gen^2 = |K| / k^2 / sin_K^2(chi)
  double* cotg = lds + 8 * ntau;     // [ntau] cotK_gen = cscK_gen cos_K(chi)  (tensor E and B types)
  int& next_l = *(int*)(lds + 9 * ntau);

  const int iq = nq - 1 - blockIdx.x;
  const double q = P.q[iq], k = C.kq[iq], sqrtK = C.sqrtK, K = C.K;
  const int sgnK = C.sgnK;
  auto sinKf = [&](double x) { return sgnK == 1 ? sin(x) : sinh(x); };      // sin_K
  auto asinKf = [&](double x) { return sgnK == 1 ? asin(x) : asinh(x); };
  const int ik = P.ik[iq];
  const int tid = threadIdx.x;
  if (ik < 0) {
    for (int e = tid; e < 5 * nl; e += blockDim.x) {
      int t = e / nl, il = e - t * nl;
      if (P.tts[t] >= 0) P.out[((size_t)P.tts[t] * nl + il) * nq + iq] = 0.;
    }
    return;
  }
  const bool own = iq < C.index_q_flat;
  const double nu = (own && sgnK == 1) ? (double)(int)(q / sqrtK + 0.2) : q / sqrtK;   // tm.cpp:3796-3804 / :3331
  const int nu_int = (sgnK == 1) ? (int)(q / sqrtK + 0.2) : 2147483647;                 // closed (tm.cpp:1636): l < nu only
  HisDesc D;
  D.nl = 0; D.nx = 0; D.dx = 1.; D.off = 0; D.trig_off = 0;
  if (own) D = C.desc[iq];

  const double k_lo = P.k[ik], k_hi = P.k[ik + 1], h = k_hi - k_lo;
  const double b = (k - k_lo) / h, a = 1. - b;
  const double ca = (a * a * a - a), cb = (b * b * b - b), h2 = h * h / 6.0;
  const double lfac = (P.tts[4] >= 0) ? P.lcmb_fac_rescale * pow(k / P.lcmb_pivot, P.lcmb_tilt) : 0.;
  if (tid == 0) next_l = 0;
  for (int i = tid; i < ntau; i += blockDim.x) {
    const double tau = P.tau[i], tm = P.tau0 - tau;
    t0mt[i] = tm;
    double w;
    if (i == 0) w = 0.5 * (tm - (P.tau0 - P.tau[1]));
    else if (i == ntau - 1) w = 0.5 * ((P.tau0 - P.tau[ntau - 2]) - tm);
    else w = 0.5 * ((P.tau0 - P.tau[i - 1]) - (P.tau0 - P.tau[i + 1]));
    wt[i] = w;
    const double sc = sqrtK / k / sinKf(sqrtK * tm);
    csc2[i] = sc * sc;
    cotg[i] = sc * (sgnK == 1 ? cos(sqrtK * tm) : cosh(sqrtK * tm));   // tm.cpp:1726, 1742
#pragma unroll
    for (int t = 0; t < 5; t++) {
      double v = 0.;
      if (P.tts[t] >= 0) {
        const size_t o = ((size_t)P.tps[t] * P.nk + ik) * ntau + i;
        v = a * P.src[o] + b * P.src[o + ntau] + (ca * P.dd[o] + cb * P.dd[o + ntau]) * h2;
        if (t == 4) {
          double resc = 0.;   // tm.cpp:1926-1932
          if (!(i == ntau - 1 || i < P.imin_lcmb))
            resc = sqrtK * sinKf((P.tau_rec - tau) * sqrtK) / sinKf((P.tau0 - tau) * sqrtK) / sinKf((P.tau0 - P.tau_rec) * sqrtK);
          v = v * resc * lfac;
        }
      }
      S[t * ntau + i] = v;
    }
  }
  __syncthreads();

  const int lane = tid & 63;
  unsigned long long n_int = 0, n_samp = 0, n_fused = 0;
  const double* S_l = S + 4 * ntau;
  const int imin = P.imin_lcmb;
  const int tsz_l = ntau - imin;
  const double sqrt_absK_over_k = sqrtK / k, absK_over_k2 = sqrt_absK_over_k * sqrt_absK_over_k;
  const double s2 = sqrt(1.0 - 3.0 * K / (k * k));

  for (;;) {
    int il = 0;
    if (lane == 0) il = atomicAdd(&next_l, 1);
    il = __shfl(il, 0, 64);
    if (il >= nl) break;
    const int li = P.l[il];
    const double l = (double)li;
    double acc[5] = {0., 0., 0., 0., 0.};
    int imax_t[5] = {-1, -1, -1, -1, -1};
    bool trunc_t[5] = {false, false, false, false, false};
    int imax_all = -1;
    bool lcmb_limber = false;
    // closed space: l < nu, and (own table) l within the list the table was built for (tm.cpp:1636-1643)
    const bool exists = (li < nu_int) && (!own || il < D.nl);
    double tmin_bessel = 0.;
    // flat rescaling approximation of this (q, l): tm.cpp:3329-3342
    double rescale_argument = 1., rescale_amplitude = 1., chi_tp = 0., at = 0.;
    if (exists) {
      if (own) tmin_bessel = xmin_from_approx_curved(sgnK, l, nu, C.phiminabs) / sqrtK;
      else {
        chi_tp = asinKf(sqrt(l * (l + 1.)) / nu);
        tmin_bessel = P.chi_min[il] / sqrtK * chi_tp / sqrt(l * (l + 1.));   // tm.cpp:2781-2786 (asin(sqrt(l(l+1)) sqrtK / q))
        rescale_argument = sqrt(l * (l + 1.)) / chi_tp;
        rescale_amplitude = pow(1. - K * l * (l + 1.) / q / q, -1. / 12.);
        at = atan(l / nu);
      }
      const bool late = l > P.late_l;
      const int imax_b = (tmin_bessel >= t0mt[0]) ? -1 : last_ge(t0mt, 0, ntau - 1, tmin_bessel);
#pragma unroll
      for (int t = 0; t < 4; t++) {
        if (P.tts[t] < 0) continue;
        if (l < (q - P.dk[t]) * P.ra_rec) continue;     // tm.cpp:3187-3196, called with q (tm.cpp:1626)
        if (imax_b < 0) continue;
        int im = imax_b;
        const double* St = S + t * ntau;
        while (im >= 0 && St[im] == 0.) im--;
        if (im >= 0 && late && t != 0) im = (P.i_cut < im) ? P.i_cut : im;
        if (im < 0) continue;
        imax_t[t] = im;
        trunc_t[t] = (im != ntau - 1) && (im == imax_b);
        imax_all = im > imax_all ? im : imax_all;
      }
      if (P.tts[4] >= 0) {
        if (l > P.l_switch_limber) lcmb_limber = true;
        else if (tmin_bessel < t0mt[imin]) {
          const int imb = last_ge(t0mt, imin, ntau - 1, tmin_bessel);
          int im = imb;
          while (im >= imin && S_l[im] == 0.) im--;
          if (im >= imin) {
            imax_t[4] = im;
            trunc_t[4] = (im != ntau - 1) && (im == imb);
            imax_all = im > imax_all ? im : imax_all;
          }
        }
      }
    }
    if (imax_all >= 0) {
      const double lxlp1 = l * (l + 1.0);
      const double fac_e = sqrt(3.0 / 8.0 * (l + 2.0) * (l + 1.0) * l * (l - 1.0)) / s2;
      const double2* tl = own ? C.his + D.off + (size_t)il * D.nx : P.bes + (size_t)il * P.nx;
      const double2* tg = C.trig + D.trig_off;
      for (int i = lane; i <= imax_all; i += 64) {
        const double tm = t0mt[i];
        const double chi0 = sqrtK * tm;            // tm.cpp:1724
        double Phi, dPhi, d2Phi, rf = 1.;
        if (own) hermite6_curved(tl, tg, D.nx, C.his_xmin, D.dx, nu, sgnK, lxlp1, chi0, &Phi, &dPhi, &d2Phi);
        else {
          hermite4_flat(tl, P.nx, P.bes_xmin, P.bes_dx, P.bes_xmax, lxlp1, chi0 * rescale_argument, &Phi, &dPhi, &d2Phi);
          const double dxx = at * (chi0 - chi_tp);  // tm.cpp:3375-3383
          rf = (sgnK == 1) ? fmin(rescale_amplitude * (1 + 0.34 * dxx + 2.00 * dxx * dxx), chi0 / sin(chi0))
                           : fmax(rescale_amplitude * (1 - 0.38 * dxx + 0.40 * dxx * dxx), chi0 / sinh(chi0));
        }
        const double w = wt[i];
        double R[4] = {Phi * rf, sqrt_absK_over_k * dPhi * rescale_argument * rf,
                       1.0 / (2.0 * s2) * (3 * absK_over_k2 * d2Phi * rescale_argument * rescale_argument + Phi) * rf,
                       fac_e * csc2[i] * Phi * rf};
        if (P.tensors) {  // tm.cpp:3494-3529
          const double Kk2 = K / (k * k), ssqrt2 = sqrt(1.0 - Kk2), si = sqrt(1.0 + 2.0 * Kk2), ssqrt2i = sqrt(1.0 + 3.0 * Kk2), cg = cotg[i];
          R[0] = fac_e * s2 / si / ssqrt2 * csc2[i] * Phi * rf;   // (fac_e carries the scalar 1/s2: undo it)
          R[1] = 0.25 / si / ssqrt2 * (absK_over_k2 * d2Phi * rescale_argument * rescale_argument + 4.0 * cg * sqrt_absK_over_k * dPhi * rescale_argument -
                                       (1.0 + 4 * Kk2 - 2.0 * cg * cg) * Phi) * rf;
          R[2] = 0.5 * ssqrt2i / ssqrt2 / si * (sqrt_absK_over_k * dPhi * rescale_argument + 2.0 * cg * Phi) * rf;
          R[3] = 0.;
        }
#pragma unroll
        for (int t = 0; t < 4; t++) {
          if (i <= imax_t[t]) {
            const double s = S[t * ntau + i];
            double term = s * R[t] * w;
            if (trunc_t[t] && i == imax_t[t]) term -= 0.5 * (t0mt[i + 1] - tmin_bessel) * R[t] * s;
            acc[t] += term;
          }
        }
        if (i >= imin && i <= imax_t[4]) {
          const double s = S_l[i];
          const double wl = (i == imin) ? 0.5 * (tm - t0mt[i + 1]) : w;
          double term = s * R[0] * wl;
          if (trunc_t[4] && i == imax_t[4]) term -= 0.5 * (t0mt[i + 1] - tmin_bessel) * R[0] * s;
          acc[4] += term;
        }
      }
#pragma unroll
      for (int t = 0; t < 5; t++) acc[t] = wave_sum(acc[t]);
    }
    if (lane == 0) {
      if (lcmb_limber) {  // tm.cpp:2930-2968, closed
        double res = 0.;
        const double tl = (sgnK == 1) ? asin(sqrt(l * (l + 1.)) / q * sqrtK) / sqrtK : asinh((l + 0.5) / q * sqrtK) / sqrtK;
        if (!(tl > t0mt[imin] || tl < t0mt[ntau - 1])) {
          int j = last_ge(t0mt, imin, ntau - 1, tl) - imin;
          int it = j + 1;
          if (j >= 0 && t0mt[imin + j] == tl) it = j;
          if (it < 1) it = 1;
          if (it > tsz_l - 2) it = tsz_l - 2;
          const int o = imin + it;
          const double y3 = (it < tsz_l - 2) ? S_l[o + 1] * t0mt[o + 1] : S_l[o] * t0mt[o];
          const double Sv = parabola(t0mt[o - 1], t0mt[o], t0mt[o + 1], tl, S_l[o - 1] * t0mt[o - 1], S_l[o] * t0mt[o], y3);
          const double IPhiFlat = sqrt(3.1415926535897932384626433832795 / (2. * l)) * (1. - 0.25 / l + 1. / 32. / (l * l));
          res = IPhiFlat * Sv * pow(1. - K * l * l / q / q, -1. / 4.) / (tl * q);
        }
        acc[4] = res;
      }
#pragma unroll
      for (int t = 0; t < 5; t++)
        if (P.tts[t] >= 0) P.out[((size_t)P.tts[t] * nl + il) * nq + iq] = acc[t];
      for (int t = 0; t < 5; t++)
        if (imax_t[t] >= 0) { n_int++; n_samp += (t == 4) ? (imax_t[t] - imin + 1) : (imax_t[t] + 1); }
      if (imax_all >= 0) n_fused += imax_all + 1;
    }
  }
  if (lane == 0 && P.work) {
    atomicAdd(&P.work[0], n_int);
    atomicAdd(&P.work[1], n_samp);
    atomicAdd(&P.work[2], n_fused);
  }
}

// ---------------------------------------------------------------------------------------------
int cpt_transfer_impl(cpt_handle* h, const double* sources_dev, const double* k, int nk, int k_size_cl,
                      const double* tau, int ntau, const double* q, int nq, const int* l, int nl, double* transfer_dev) {
  const cpt_config& c = h->cfg;
  const int ntp = c.tp_size;
  int rc;
  const bool tens = c.mode == CPT_MODE_TENSORS;
  const int tts_s[5] = {c.index_tt_t0, c.index_tt_t1, c.index_tt_t2, c.index_tt_e, c.index_tt_lcmb};
  const int tps_s[5] = {c.index_tp_t0, c.index_tp_t1, c.index_tp_t2, c.index_tp_p, c.index_tp_phi_plus_psi};
  const int tts_t[5] = {c.index_tt_t2, c.index_tt_e, c.index_tt_b, -1, -1};   // tm.cpp:455-470: tensor types t2, e, b
  const int tps_t[5] = {c.index_tp_t2, c.index_tp_p, c.index_tp_p, -1, -1};   // E and B both project the polarisation source
  const int* tts = tens ? tts_t : tts_s;
  const int* tps = tens ? tps_t : tps_s;
  for (int t = 0; t < 5; t++) {
    if (tts[t] >= c.tt_size) return cpt_fail(h, CPT_ERR_INVALID, "index_tt_* >= tt_size");
    if (tts[t] >= 0 && (tps[t] < 0 || tps[t] >= ntp))
      return cpt_fail(h, CPT_ERR_INVALID, "transfer type %d requested but its source type is absent", t);
  }
  if (!sources_dev && (!h->d_src || h->src_nk != nk || h->src_ntau != ntau))
    return cpt_fail(h, CPT_ERR_INVALID, "sources_dev is NULL and the handle holds no resident sources of shape [%d][%d][%d]",
                    ntp, nk, ntau);
  size_t lds_bytes = (size_t)(c.K != 0. ? 9 : 7) * ntau * sizeof(double) + 16;
  if (lds_bytes > 160 * 1024 - 256) return cpt_fail(h, CPT_ERR_UNSUPPORTED, "ntau=%d too large for the LDS staging (160 KB/CU)", ntau);
  const bool closed = c.K != 0.;   // (any curved space: closed or open)
  const int sgnK = (c.K > 0.) ? 1 : (c.K < 0. ? -1 : 0);

  // ---- geometry cache: everything below that depends on the grids only (validation, spline elimination factors, bracketing
  //      indices, Bessel / hyperspherical tables, their uploads) is done once per (k, tau, q, l) and found in HBM afterwards ----
  const double* kbuf0 = h->d_k; const double* qbuf0 = h->d_q; const double* taubuf0 = h->d_tau;
  const size_t nsrc = (size_t)ntp * nk * ntau;
  if ((rc = cpt_reserve(h, &h->d_dd, &h->dd_cap, nsrc))) return rc;
  if ((rc = cpt_reserve(h, &h->d_k, &h->grid_cap_k, (size_t)4 * nk))) return rc;  // k + splc[3][nk]
  if ((rc = cpt_reserve(h, &h->d_tau, &h->grid_cap_tau, (size_t)ntau))) return rc;
  if ((rc = cpt_reserve(h, &h->d_q, &h->grid_cap_q, (size_t)2 * nq))) return rc;  // q + ik (as int)
  h->d_splc = h->d_k + nk;
  h->d_ik = (int*)(h->d_q + nq);
  if (h->d_k != kbuf0 || h->d_q != qbuf0 || h->d_tau != taubuf0) h->geo_tr_valid = false;
  const bool hit = h->geo_tr_valid && h->geo_tr_k_size_cl == k_size_cl && (int)h->geo_tr_k.size() == nk && (int)h->geo_tr_tau.size() == ntau &&
                   (int)h->geo_tr_q.size() == nq && (int)h->geo_tr_l.size() == nl && memcmp(h->geo_tr_k.data(), k, nk * sizeof(double)) == 0 &&
                   memcmp(h->geo_tr_tau.data(), tau, ntau * sizeof(double)) == 0 && memcmp(h->geo_tr_q.data(), q, nq * sizeof(double)) == 0 &&
                   memcmp(h->geo_tr_l.data(), l, nl * sizeof(int)) == 0;
  cpt_timer_start(h, CPT_T_TRANSFER);
  if (!hit) {
    h->geo_tr_valid = false;
    // ---- host-side validation of everything the kernels index with (no out-of-bounds launches) ----
    for (int i = 1; i < nk; i++)
      if (!(k[i] > k[i - 1])) return cpt_fail(h, CPT_ERR_INVALID, "k grid must be strictly increasing");
    for (int i = 1; i < ntau; i++)
      if (!(tau[i] > tau[i - 1])) return cpt_fail(h, CPT_ERR_INVALID, "tau_sampling must be strictly increasing");
    for (int i = 1; i < nq; i++)
      if (!(q[i] > q[i - 1])) return cpt_fail(h, CPT_ERR_INVALID, "q grid must be strictly increasing");
    if (!(q[0] > 0.) || !(k[0] > 0.)) return cpt_fail(h, CPT_ERR_INVALID, "wavenumbers must be positive");
    if (!(tau[ntau - 1] <= c.tau0)) return cpt_fail(h, CPT_ERR_INVALID, "tau_sampling exceeds conformal age");

    // ---- Bessel table (cached on (l list, xmax)); tm.cpp:246-262 ----
    double xmax = q[nq - 1] * c.tau0;
    if (c.K < 0.) xmax *= (l[nl - 1] / c.hyper_flat_approximation_nu) / asinh(l[nl - 1] / c.hyper_flat_approximation_nu) * 1.01;   // tm.cpp:247-249
    if ((rc = cpt_bessel_build(h, l, nl, xmax))) return rc;
    const double bes_xmax = c.hyper_x_min + (h->bes_nx - 1) * h->bes_dx;
    if (c.K == 0. && q[nq - 1] > bes_xmax / (c.tau0 - tau[0]))
      return cpt_fail(h, CPT_ERR_RUNTIME, "q_max exceeds q_max_bessel (tm.cpp:1660): Limber fallback for CMB types not implemented");

    // ---- host prep: spline elimination factors of the k grid, bracketing indices ----
    std::vector<double> hk((size_t)4 * nk);
    memcpy(hk.data(), k, nk * sizeof(double));
    double* cc = hk.data() + nk;
    double* sg = cc + nk;
    double* pp = sg + nk;
    cc[0] = -0.5; sg[0] = 0.; pp[0] = 1.;
    for (int i = 1; i < nk - 1; i++) {
      sg[i] = (k[i] - k[i - 1]) / (k[i + 1] - k[i - 1]);
      pp[i] = sg[i] * cc[i - 1] + 2.0;
      cc[i] = (sg[i] - 1.0) / pp[i];
    }
    cc[nk - 1] = 0.; sg[nk - 1] = 0.; pp[nk - 1] = 1.;
    std::vector<double> kq(nq);  // k(q) = sqrt(q^2 - K(1+m)), tm.cpp:1106-1167 (scalars: m = 0); flat: k = q
    for (int i = 0; i < nq; i++) kq[i] = closed ? sqrt(q[i] * q[i] - c.K * (tens ? 3. : 1.)) : q[i];   // m = 2 for tensors
    std::vector<int> ik(nq);
    {
      int j = 0;  // tm.cpp:1794-1802 (q ascending -> resume the scan)
      for (int i = 0; i < nq; i++) {
        if (!(kq[i] <= k[k_size_cl - 1])) { ik[i] = -1; continue; }
        while ((j + 1) < nk && k[j + 1] < kq[i]) j++;
        ik[i] = (j + 1 < nk) ? j : nk - 2;
      }
    }
    // ---- closed space: one hyperspherical table per q below the flat-approximation threshold (tm.cpp:3777-3887, 1081-1088) ----
    int index_q_flat = 0, his_max_nx = 0;
    std::vector<HisDesc> desc;
    size_t his_total = 0, trig_total = 0;
    if (closed) {
      const double sqrtK = sqrt(fabs(c.K)), PI = 3.1415926535897932384626433832795;
      const double q_approximation = c.hyper_flat_approximation_nu * sqrtK;
      for (index_q_flat = 0; index_q_flat < nq - 1; index_q_flat++)
        if (q[index_q_flat] > q_approximation) break;
      desc.resize(index_q_flat);
      const double xmin = c.hyper_x_min, xmaxK = (sgnK == 1) ? std::min(sqrtK * c.tau0, PI / 2.0 - xmin) : sqrtK * c.tau0;
      for (int i = 0; i < index_q_flat; i++) {
        HisDesc& D = desc[i];
        double nu = q[i] / sqrtK;
        int nlq = nl;
        if (sgnK == 1) {
          nu = (double)(int)(q[i] / sqrtK + 0.2);
          if (q[i] / sqrtK - nu > 1.e-6)
            return cpt_fail(h, CPT_ERR_INVALID, "problem in q list definition in closed case for index_q=%d, nu=%e (tm.cpp:3800-3803)", i, q[i] / sqrtK);
          while (nlq > 0 && (double)l[nlq - 1] >= nu) nlq--;
        }
        // open space: every l of the list (the reference's WKB/Airy l_max cut, tm.cpp:3823-3856, only drops functions that
        // stay below hyper_phi_min_abs on the whole range)
        const double sampling = (nu > c.hyper_nu_sampling_step) ? c.hyper_sampling_curved_high_nu : c.hyper_sampling_curved_low_nu;
        int nx = (int)((xmaxK - xmin) * sampling / (2 * PI / nu));
        if (nx < 2) nx = 2;
        D.nu = nu; D.nl = nlq; D.nx = nx; D.dx = (xmaxK - xmin) / (nx - 1.0);
        D.off = his_total; D.trig_off = trig_total;
        D.special = 0; D.L = 0; D.xfwdidx = 0;
        if (nlq > 0) {
          const int lmax = l[nlq - 1];
          D.special = (sgnK == 1 && (int)(nu + 0.2) == lmax + 1) ? 1 : 0;
          D.L = D.special ? lmax : lmax + 1;
          const double xfwd = (sgnK == 1) ? asin(sqrt(lmax * (lmax + 1.0)) / nu) : asinh(sqrt(lmax * (lmax + 1.0)) / nu);
          D.xfwdidx = (int)((xfwd - xmin) / D.dx);
          his_total += (size_t)nlq * nx;
        }
        trig_total += (size_t)nx;
        his_max_nx = std::max(his_max_nx, nx);
      }
    }
    int imin_lcmb = 0;
    while (imin_lcmb < ntau && tau[imin_lcmb] <= c.tau_rec) imin_lcmb++;
    if (tts[4] >= 0 && ntau - imin_lcmb < 3)
      return cpt_fail(h, CPT_ERR_INVALID, "fewer than 3 sampling times after recombination for the lensing source");
    int i_cut = -1;
    for (int i = 0; i < ntau; i++)
      if (c.tau0 - tau[i] >= c.tau0 - c.tau_cut) i_cut = i;

    if ((rc = cpt_upload(h, h->d_k, hk.data(), hk.size() * sizeof(double)))) return rc;
    if ((rc = cpt_upload(h, h->d_tau, tau, ntau * sizeof(double)))) return rc;
    if ((rc = cpt_upload(h, h->d_q, q, nq * sizeof(double)))) return rc;
    if ((rc = cpt_upload(h, h->d_ik, ik.data(), nq * sizeof(int)))) return rc;
    if (closed) {
      if ((rc = cpt_reserve(h, &h->d_his, &h->his_cap, his_total + 1))) return rc;
      if ((rc = cpt_reserve(h, &h->d_his_trig, &h->his_trig_cap, trig_total + 1))) return rc;
      if ((rc = cpt_reserve(h, &h->d_kq, &h->kq_cap, (size_t)nq))) return rc;
      {
        char* pdesc = (char*)h->d_his_desc;
        size_t cap = h->his_desc_cap;
        if ((rc = cpt_reserve(h, &pdesc, &cap, (desc.size() + 1) * sizeof(HisDesc)))) return rc;
        h->d_his_desc = pdesc; h->his_desc_cap = cap;
      }
      if ((rc = cpt_upload(h, h->d_kq, kq.data(), nq * sizeof(double)))) return rc;
      if (!desc.empty()) {
        if ((rc = cpt_upload(h, h->d_his_desc, desc.data(), desc.size() * sizeof(HisDesc)))) return rc;
        // the per-q hyperspherical tables depend on (q, l, K) only: built here, once per geometry
        hipLaunchKernelGGL(k_his_curved, dim3((his_max_nx + 63) / 64, (unsigned)desc.size()), dim3(64), 0, h->stream, (const HisDesc*)h->d_his_desc,
                           h->d_l, c.hyper_x_min, sgnK, h->d_his, h->d_his_trig);
        CPT_HIP(h, hipGetLastError());
      }
    }
    h->geo_imin_lcmb = imin_lcmb; h->geo_i_cut = i_cut; h->geo_index_q_flat = index_q_flat; h->geo_his_max_nx = his_max_nx; h->geo_n_desc = desc.size();
    h->geo_tr_k.assign(k, k + nk); h->geo_tr_tau.assign(tau, tau + ntau); h->geo_tr_q.assign(q, q + nq); h->geo_tr_l.assign(l, l + nl);
    h->geo_tr_k_size_cl = k_size_cl;
    h->geo_tr_valid = true;
  }
  const double bes_xmax = c.hyper_x_min + (h->bes_nx - 1) * h->bes_dx;
  CPT_HIP(h, hipMemsetAsync(h->d_work, 0, 3 * sizeof(unsigned long long), h->stream));

  if (sources_dev) {
    if ((rc = cpt_reserve(h, &h->d_src, &h->src_cap, nsrc))) return rc;
    if ((rc = cpt_transpose_to_kmajor(h, sources_dev, h->d_src, ntp, ntau, nk))) return rc;
    h->src_nk = nk;
    h->src_ntau = ntau;
  }
  hipLaunchKernelGGL(k_source_spline, dim3((ntp * ntau + 63) / 64), dim3(64), 0, h->stream, h->d_src, h->d_dd, h->d_k,
                     h->d_splc, ntp, nk, ntau);
  CPT_HIP(h, hipGetLastError());

  LosParams P;
  P.src = h->d_src; P.dd = h->d_dd; P.k = h->d_k; P.tau = h->d_tau; P.q = h->d_q; P.l = h->d_l; P.ik = h->d_ik;
  P.bes = h->d_bes; P.chi_min = h->d_chi_min; P.out = transfer_dev; P.work = h->d_work;
  P.nk = nk; P.ntau = ntau; P.nq = nq; P.nl = nl; P.nx = h->bes_nx;
  P.bes_xmin = c.hyper_x_min; P.bes_dx = h->bes_dx; P.bes_xmax = bes_xmax;
  for (int t = 0; t < 5; t++) { P.tts[t] = tts[t]; P.tps[t] = tps[t] < 0 ? 0 : tps[t]; }
  P.tensors = tens ? 1 : 0;
  P.dk[0] = c.transfer_neglect_delta_k_S_t0; P.dk[1] = c.transfer_neglect_delta_k_S_t1;
  P.dk[2] = c.transfer_neglect_delta_k_S_t2; P.dk[3] = c.transfer_neglect_delta_k_S_e;
  if (tens) { P.dk[0] = c.transfer_neglect_delta_k_T_t2; P.dk[1] = c.transfer_neglect_delta_k_T_e; P.dk[2] = c.transfer_neglect_delta_k_T_b; P.dk[3] = 0.; }
  P.tau0 = c.tau0; P.tau_rec = c.tau_rec; P.ra_rec = (c.tau0 - c.tau_rec) * c.angular_rescaling;
  P.t0mt_cut = c.tau0 - c.tau_cut; P.late_l = c.transfer_neglect_late_source * c.angular_rescaling;
  P.l_switch_limber = c.l_switch_limber;
  P.lcmb_fac_rescale = c.lcmb_rescale; P.lcmb_tilt = c.lcmb_tilt; P.lcmb_pivot = c.lcmb_pivot;
  P.imin_lcmb = h->geo_imin_lcmb; P.i_cut = h->geo_i_cut;

  cpt_timer_start(h, CPT_T_LOS);
  if (closed) {
    LosClosedParams CP;
    CP.b = P; CP.kq = h->d_kq; CP.desc = (const HisDesc*)h->d_his_desc; CP.his = h->d_his; CP.trig = h->d_his_trig;
    CP.K = c.K; CP.sqrtK = sqrt(fabs(c.K)); CP.sgnK = sgnK; CP.his_xmin = c.hyper_x_min; CP.phiminabs = c.hyper_phi_min_abs; CP.index_q_flat = h->geo_index_q_flat;
    hipLaunchKernelGGL(k_los_curved, dim3(nq), dim3(256), lds_bytes, h->stream, CP);
  } else
    hipLaunchKernelGGL(k_los, dim3(nq), dim3(256), lds_bytes, h->stream, P);
  CPT_HIP(h, hipGetLastError());
  cpt_timer_stop(h, CPT_T_LOS);
  cpt_timer_stop(h, CPT_T_TRANSFER);
  // work counters -> pinned landing zone, read by cpt_finish after the synchronisation of the call
  CPT_HIP(h, hipMemcpyAsync(h->pin_out, h->d_work, 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
  h->pend_work = true;
  return CPT_OK;
}
