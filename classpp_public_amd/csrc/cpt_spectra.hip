// "Next" rows S8f-2,3: C_l assembly and linear P(k) on the device, so that the path's results can be stated in the
// contract's units without leaving HBM.  Restates SpectraModule::spectra_compute_cl (source/spectra_module.cpp:958-1353,
// flat space, scalar mode, one initial condition) with the integrand spline of tools/arrays.c (array_spline,
// _SPLINE_EST_DERIV_) and array_integrate_all_trapzd_or_spline (:1382-1423), and NonlinearModule::nonlinear_pk_linear
// (source/nonlinear_module.cpp:1886-2040).
#include "cpt_internal.h"

struct ClParams {
  const double* tr;  // [tt][nl][nq]
  const double* q;
  double* cl;        // [nl][ct]
  double* scratch;   // [nl*ct][2][nq]: integrand y and spline work array
  const double* splc; // [3][nq]: c, sig, p of the q grid
  int nq, nl, ct_size;
  int tt_t0, tt_t1, tt_t2, tt_e, tt_lcmb;
  int ct_tt, ct_ee, ct_te, ct_bb, ct_pp, ct_tp, ct_ep;
  double A_s, n_s, alpha_s, k_pivot;
};

__device__ static inline double primordial(const ClParams& P, double k) {  // primordial_module.cpp:911-925
  const double lk = log(k / P.k_pivot);
  return P.A_s * exp((P.n_s - 1.) * lk + 0.5 * P.alpha_s * lk * lk);
}

// one thread per (l, ct): build the integrand over q, spline it (natural order, sequential sweeps), integrate
__global__ void __launch_bounds__(64) k_cl(ClParams P) {
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= P.nl * P.ct_size) return;
  const int il = id / P.ct_size, ct = id - il * P.ct_size;
  const int nq = P.nq;
  double* y = P.scratch + (size_t)id * 2 * nq;
  double* dd = y + nq;
  const double* x = P.q;
  // which product of transfer functions is this spectrum? (spectra_module.cpp:1027-1185)
  int kind = -1;
  if (ct == P.ct_tt) kind = 0; else if (ct == P.ct_ee) kind = 1; else if (ct == P.ct_te) kind = 2;
  else if (ct == P.ct_pp) kind = 4; else if (ct == P.ct_tp) kind = 5; else if (ct == P.ct_ep) kind = 6;
  if (kind < 0) { P.cl[id] = 0.; return; }  // bb vanishes for scalar modes (spectra_module.cpp:1262-1270)
  const size_t st = (size_t)P.nl * nq, row = (size_t)il * nq;
  const double PI = 3.1415926535897932384626433832795e0;
  for (int iq = 0; iq < nq; iq++) {
    const double k = x[iq];
    double temp = 0., e = 0., lc = 0.;
    if (P.tt_t0 >= 0) temp = P.tr[P.tt_t0 * st + row + iq] + P.tr[P.tt_t1 * st + row + iq] + P.tr[P.tt_t2 * st + row + iq];
    if (P.tt_e >= 0) e = P.tr[P.tt_e * st + row + iq];
    if (P.tt_lcmb >= 0) lc = P.tr[P.tt_lcmb * st + row + iq];
    double prod;
    switch (kind) {
      case 0: prod = temp * temp; break;
      case 1: prod = e * e; break;
      case 2: prod = 0.5 * (temp * e + e * temp); break;
      case 4: prod = lc * lc; break;
      case 5: prod = 0.5 * (temp * lc + lc * temp); break;
      default: prod = 0.5 * (e * lc + lc * e); break;
    }
    y[iq] = primordial(P, k) * prod * (4. * PI / k);
  }
  // spline, _SPLINE_EST_DERIV_: the elimination factors c, sig, p depend on the q grid only (host, P.splc);
  // forward sweep stores u_i in dd, the backward sweep finishes dd and integrates on the fly (arrays.c:1413-1421)
  const int n = nq;
  const double* cc = P.splc;
  const double* sg = P.splc + n;
  const double* pp = P.splc + 2 * n;
  const double dy_first = ((x[2] - x[0]) * (x[2] - x[0]) * (y[1] - y[0]) - (x[1] - x[0]) * (x[1] - x[0]) * (y[2] - y[0])) /
                          ((x[2] - x[0]) * (x[1] - x[0]) * (x[2] - x[1]));
  double u = (3. / (x[1] - x[0])) * ((y[1] - y[0]) / (x[1] - x[0]) - dy_first);
  dd[0] = u;
  for (int i = 1; i < n - 1; i++) {
    const double ui = (y[i + 1] - y[i]) / (x[i + 1] - x[i]) - (y[i] - y[i - 1]) / (x[i] - x[i - 1]);
    u = (6.0 * ui / (x[i + 1] - x[i - 1]) - sg[i] * u) / pp[i];
    dd[i] = u;
  }
  const double dy_last = ((x[n - 3] - x[n - 1]) * (x[n - 3] - x[n - 1]) * (y[n - 2] - y[n - 1]) -
                          (x[n - 2] - x[n - 1]) * (x[n - 2] - x[n - 1]) * (y[n - 3] - y[n - 1])) /
                         ((x[n - 3] - x[n - 1]) * (x[n - 2] - x[n - 1]) * (x[n - 3] - x[n - 2]));
  const double un = (3. / (x[n - 1] - x[n - 2])) * (dy_last - (y[n - 1] - y[n - 2]) / (x[n - 1] - x[n - 2]));
  double dd_next = (un - 0.5 * u) / (0.5 * cc[n - 2] + 1.0);  // dd[n-1]
  double sum = 0.;
  for (int i = n - 2; i >= 0; i--) {
    const double ddi = cc[i] * dd_next + dd[i];
    const double h = x[i + 1] - x[i];
    sum += (y[i] + y[i + 1]) * h / 2. + (ddi + dd_next) * h * h * h / 24.;
    dd_next = ddi;
  }
  P.cl[id] = sum;
}

__global__ void k_pk(const double* __restrict__ src, const double* __restrict__ k, double* __restrict__ pk, int nk, int ntau,
                     int tp_dm, double A_s, double n_s, double alpha_s, double k_pivot) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nk) return;
  const double dm = src[((size_t)tp_dm * nk + i) * ntau + (ntau - 1)];  // resident sources are k-major: [tp][k][tau]
  const double lk = log(k[i] / k_pivot);
  const double pr = A_s * exp((n_s - 1.) * lk + 0.5 * alpha_s * lk * lk);
  const double PI = 3.1415926535897932384626433832795e0;
  pk[i] = 2. * PI * PI / (k[i] * k[i] * k[i]) * dm * dm * pr;  // nonlinear_module.cpp:1952-1991
}

int cpt_cl_impl(cpt_handle* h, const cpt_spectra_params* sp, const double* transfer_dev, const double* q, int nq, int nl,
                double* cl_dev) {
  const cpt_config& c = h->cfg;
  if (sp->ct_size < 1 || sp->ct_size > 8) return cpt_fail(h, CPT_ERR_INVALID, "ct_size=%d out of range", sp->ct_size);
  const int cts[7] = {sp->index_ct_tt, sp->index_ct_ee, sp->index_ct_te, sp->index_ct_bb, sp->index_ct_pp, sp->index_ct_tp, sp->index_ct_ep};
  for (int i = 0; i < 7; i++)
    if (cts[i] >= sp->ct_size) return cpt_fail(h, CPT_ERR_INVALID, "index_ct_* >= ct_size");
  if ((sp->index_ct_tt >= 0 || sp->index_ct_te >= 0 || sp->index_ct_tp >= 0) && (c.index_tt_t0 < 0 || c.index_tt_t1 < 0 || c.index_tt_t2 < 0))
    return cpt_fail(h, CPT_ERR_INVALID, "temperature C_l requested without temperature transfer functions");
  if ((sp->index_ct_ee >= 0 || sp->index_ct_te >= 0 || sp->index_ct_ep >= 0) && c.index_tt_e < 0)
    return cpt_fail(h, CPT_ERR_INVALID, "polarisation C_l requested without E transfer functions");
  if ((sp->index_ct_pp >= 0 || sp->index_ct_tp >= 0 || sp->index_ct_ep >= 0) && c.index_tt_lcmb < 0)
    return cpt_fail(h, CPT_ERR_INVALID, "lensing C_l requested without lensing transfer functions");
  for (int i = 1; i < nq; i++)
    if (!(q[i] > q[i - 1])) return cpt_fail(h, CPT_ERR_INVALID, "q grid must be strictly increasing");
  int rc;
  if ((rc = cpt_reserve(h, &h->d_q, &h->grid_cap_q, (size_t)4 * nq))) return rc;
  std::vector<double> hq((size_t)4 * nq);
  memcpy(hq.data(), q, nq * sizeof(double));
  {
    double* cc = hq.data() + nq; double* sg = cc + nq; double* pp = sg + nq;
    cc[0] = -0.5; sg[0] = 0.; pp[0] = 1.;
    for (int i = 1; i < nq - 1; i++) {
      sg[i] = (q[i] - q[i - 1]) / (q[i + 1] - q[i - 1]);
      pp[i] = sg[i] * cc[i - 1] + 2.0;
      cc[i] = (sg[i] - 1.0) / pp[i];
    }
    cc[nq - 1] = 0.; sg[nq - 1] = 0.; pp[nq - 1] = 1.;
  }
  const size_t need = (size_t)nl * sp->ct_size * 2 * nq;
  if ((rc = cpt_reserve(h, &h->d_u, &h->u_cap, need))) return rc;
  CPT_HIP(h, hipMemcpy(h->d_q, hq.data(), hq.size() * sizeof(double), hipMemcpyHostToDevice));
  ClParams P;
  P.tr = transfer_dev; P.q = h->d_q; P.cl = cl_dev; P.scratch = h->d_u; P.splc = h->d_q + nq; P.nq = nq; P.nl = nl; P.ct_size = sp->ct_size;
  P.tt_t0 = c.index_tt_t0; P.tt_t1 = c.index_tt_t1; P.tt_t2 = c.index_tt_t2; P.tt_e = c.index_tt_e; P.tt_lcmb = c.index_tt_lcmb;
  P.ct_tt = sp->index_ct_tt; P.ct_ee = sp->index_ct_ee; P.ct_te = sp->index_ct_te; P.ct_bb = sp->index_ct_bb;
  P.ct_pp = sp->index_ct_pp; P.ct_tp = sp->index_ct_tp; P.ct_ep = sp->index_ct_ep;
  P.A_s = sp->A_s; P.n_s = sp->n_s; P.alpha_s = sp->alpha_s; P.k_pivot = sp->k_pivot;
  const int nthreads = nl * sp->ct_size;
  hipLaunchKernelGGL(k_cl, dim3((nthreads + 63) / 64), dim3(64), 0, h->stream, P);
  CPT_HIP(h, hipGetLastError());
  CPT_HIP(h, hipStreamSynchronize(h->stream));
  return CPT_OK;
}

int cpt_pk_impl(cpt_handle* h, const cpt_spectra_params* sp, const double* k, int nk, double* pk_dev) {
  const cpt_config& c = h->cfg;
  if (c.index_tp_delta_m < 0) return cpt_fail(h, CPT_ERR_INVALID, "P(k) requested but delta_m was not among the source types");
  if (!h->d_src || h->src_nk != nk) return cpt_fail(h, CPT_ERR_INVALID, "no resident sources for %d k-modes: run cpt_perturb_solve_batch first", nk);
  int rc;
  if ((rc = cpt_reserve(h, &h->d_k, &h->grid_cap_k, (size_t)4 * nk))) return rc;
  CPT_HIP(h, hipMemcpyAsync(h->d_k, k, nk * sizeof(double), hipMemcpyHostToDevice, h->stream));
  hipLaunchKernelGGL(k_pk, dim3((nk + 63) / 64), dim3(64), 0, h->stream, h->d_src, h->d_k, pk_dev, nk, h->src_ntau,
                     c.index_tp_delta_m, sp->A_s, sp->n_s, sp->alpha_s, sp->k_pivot);
  CPT_HIP(h, hipGetLastError());
  CPT_HIP(h, hipStreamSynchronize(h->stream));
  return CPT_OK;
}
