// Internal definitions shared by the HIP translation units of libcpt.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/cpt.h"

#define CPT_WAVE 64

// Compact spline tables in HBM: only the columns the RHS / sampler read, one row contiguous,
// value and second derivative interleaved: row r = { y[0], y''[0], y[1], y''[1], ... }.
// background columns (reference: background_at_tau normal_info, source/background_module.cpp:125-199)
enum { BG_A = 0, BG_H, BG_HP, BG_RHO_G, BG_RHO_B, BG_RHO_CDM, BG_RHO_UR, BG_NCOL };
// thermodynamics columns (reference: thermodynamics_at_z, source/thermodynamics_module.cpp:114-285)
enum { TH_XE = 0, TH_DKAPPA, TH_TAU_D, TH_DDKAPPA, TH_DDDKAPPA, TH_EXPMK, TH_G, TH_DG, TH_CB2, TH_NCOL };

// non-cold species: {rho, p, pseudo_p} of species n in columns 3n..3n+2 of a table that shares the background abscissa
enum { NCB_NCOL = 3 * CPT_MAX_NCDM };
// momentum bins ("chains": one Boltzmann hierarchy l = 0..lmax per (species, q)) of the ncdm species, kernel argument
struct NcdmDev {
  int n_species, nchains, lmax;
  int species[CPT_MAX_NCDM * CPT_MAX_Q_NCDM];
  int first_chain[CPT_MAX_NCDM + 1];
  double q[CPT_MAX_NCDM * CPT_MAX_Q_NCDM], w[CPT_MAX_NCDM * CPT_MAX_Q_NCDM], dlnf0[CPT_MAX_NCDM * CPT_MAX_Q_NCDM];
  double M[CPT_MAX_NCDM], factor[CPT_MAX_NCDM];
};

struct DevTables {
  int bt_size, tt_size;
  const double* tau_table;  // [bt_size]
  const double* bg;         // [bt_size][BG_NCOL][2]
  const double* z_table;    // [tt_size] ascending z
  const double* th;         // [tt_size][TH_NCOL][2]
  const double* ncb;        // [bt_size][NCB_NCOL][2] or null
};

struct Timer {
  hipEvent_t a = nullptr, b = nullptr;   // created once with the handle, recorded around the stage on the handle's stream
  double ms = 0;
  int launches = 0;
  bool started = false;                  // first event recorded, waiting for cpt_timer_stop
  bool armed = false;                    // both events recorded since the last read-out
};
// timers of the last call(s): perturbation kernel, line-of-sight kernel, whole transfer stage, whole cpt_step (first to last kernel)
enum { CPT_T_PERTURB = 0, CPT_T_LOS, CPT_T_TRANSFER, CPT_T_STEP, CPT_T_N };

struct cpt_handle {
  cpt_config cfg;
  int device = 0;
  hipStream_t stream = nullptr;
  std::string err;
  // tables
  DevTables tabs{};
  double *d_tau_table = nullptr, *d_bg = nullptr, *d_z_table = nullptr, *d_th = nullptr, *d_ncb = nullptr;
  NcdmDev ncdm{};
  // resident sources, k-major [tp][nk][ntau], left by the last perturb call or a transposed upload
  double* d_src = nullptr;
  size_t src_cap = 0;
  int src_nk = 0, src_ntau = 0;
  // transfer scratch
  double *d_dd = nullptr, *d_u = nullptr;  // spline second derivative + scratch, same shape as d_src
  size_t dd_cap = 0, u_cap = 0;
  double *d_k = nullptr, *d_tau = nullptr, *d_q = nullptr, *d_splc = nullptr;
  int *d_l = nullptr, *d_ik = nullptr;
  size_t grid_cap_k = 0, grid_cap_tau = 0, grid_cap_q = 0, grid_cap_l = 0;
  // Bessel table cache
  double2* d_bes = nullptr;  // [nl][nx] {phi, dphi}
  double* d_chi_min = nullptr;
  size_t bes_cap = 0;
  std::vector<int> bes_l;
  double bes_xmax = -1;
  int bes_nx = 0;
  double bes_dx = 0;
  unsigned long long* d_work = nullptr;  // [3] integrals, per-type samples, fused samples
  long long work_integrals = 0, work_samples = 0, work_fused = 0;
  // closed space: per-q hyperspherical tables {Phi, Phi'}, their {sinK, cotK} nodes, descriptors, k(q) (cpt_transfer.hip)
  double2 *d_his = nullptr, *d_his_trig = nullptr;
  size_t his_cap = 0, his_trig_cap = 0;
  void* d_his_desc = nullptr;
  size_t his_desc_cap = 0;  // bytes
  double* d_kq = nullptr;
  size_t kq_cap = 0;
  // lensing: Wigner-d table cache [12][num_mu][lmax+1] + coefficient / angle / work arrays (cpt_lensing.hip)
  double* d_lens = nullptr;
  size_t lens_cap = 0;
  double *lens_d = nullptr, *lens_fac = nullptr, *lens_mu = nullptr, *lens_w8 = nullptr, *lens_cgl = nullptr, *lens_ksi = nullptr,
         *lens_work = nullptr;
  int lens_num_mu = -1, lens_accurate = -1, lens_lmax = -1;
  double* d_lens_w = nullptr;
  size_t lens_w_cap = 0;
  int* d_lens_l = nullptr;
  size_t lens_l_cap = 0;
  // perturb scratch
  void* d_pt_scratch = nullptr;
  size_t pt_scratch_cap = 0;
  Timer timers[CPT_T_N];
  // ---- host staging and geometry caches: nothing on the per-step path allocates, copies from pageable memory or syncs twice ----
  // pinned arena: every host -> device copy of a call is staged here (the caller's arrays and the library's own prepared
  // arrays are pageable), bump-allocated, reset when a call begins (the previous call has drained the stream)
  char* pin = nullptr;
  size_t pin_cap = 0, pin_off = 0;
  std::vector<char*> pin_retired;   // arenas outgrown during the current call (still referenced by copies in flight)
  // pinned landing zone of the per-mode status / statistics and of the transfer work counters (device -> host, read after the sync)
  char* pin_out = nullptr;
  size_t pin_out_cap = 0;
  int pend_nk = 0;                  // perturbation results waiting in pin_out for cpt_perturb_collect (0 = none)
  bool pend_work = false;           // transfer work counters waiting in pin_out
  bool defer = false;               // inside cpt_step: stages enqueue only, one synchronisation at the end
  // geometry of the last perturbation / transfer call (content of k, tau, q, l): identical grids are not prepared or uploaded again
  std::vector<double> geo_pt_k, geo_pt_tau, geo_tr_k, geo_tr_tau, geo_tr_q;
  std::vector<int> geo_tr_l;
  int geo_tr_k_size_cl = -1;
  bool geo_tr_valid = false, geo_pt_valid = false;
  int geo_imin_lcmb = 0, geo_i_cut = -1, geo_index_q_flat = 0, geo_his_max_nx = 0;
  size_t geo_n_desc = 0;
  double* d_clw = nullptr;          // C_l quadrature weights (own buffer: d_q keeps the q grid of the cached geometry)
  size_t clw_cap = 0;
  std::vector<double> geo_cl_q; double geo_cl_sp[4] = {0, 0, 0, 0}; bool geo_cl_valid = false;
  double* d_pk_k = nullptr;
  size_t pk_k_cap = 0;
  double* d_pkz = nullptr;          // scratch of the P(k, z) spline in ln tau, [2][ln_tau_size][nk]
  size_t pkz_cap = 0;
  // multi-GPU (cpt_comm.hip): RCCL communicator of this rank (null: single GPU) and the padded exchange buffers
  void* comm = nullptr;
  int comm_rank = 0, comm_world = 1;
  double *d_xsend = nullptr, *d_xrecv = nullptr;
  size_t xsend_cap = 0, xrecv_cap = 0;
};

// stage `bytes` of host memory in the pinned arena and return the staged copy (valid until the next call on the handle begins)
void* cpt_pin(cpt_handle* h, const void* src, size_t bytes);
void cpt_pin_reset(cpt_handle* h);
// enqueue an asynchronous host -> device copy of pageable host memory through the arena
int cpt_upload(cpt_handle* h, void* dst_dev, const void* src_host, size_t bytes);
// end of an entry point: unless inside cpt_step, drain the stream and read timers / counters / statuses back
int cpt_finish(cpt_handle* h);
int cpt_perturb_collect(cpt_handle* h, const double* k, cpt_stepstat* stats, int* status);
void cpt_timer_start(cpt_handle* h, int which);
void cpt_timer_stop(cpt_handle* h, int which);

int cpt_fail(cpt_handle* h, int code, const char* fmt, ...);

#define CPT_HIP(h, call)                                                                         \
  do {                                                                                           \
    hipError_t e__ = (call);                                                                     \
    if (e__ != hipSuccess)                                                                       \
      return cpt_fail(h, CPT_ERR_NO_DEVICE, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), __FILE__, \
                      __LINE__);                                                                 \
  } while (0)

template <class T>
int cpt_reserve(cpt_handle* h, T** p, size_t* cap, size_t n) {
  if (*cap >= n && *p) return CPT_OK;
  if (*p) (void)hipFree(*p);
  *p = nullptr;
  *cap = 0;
  CPT_HIP(h, hipMalloc((void**)p, n * sizeof(T)));
  *cap = n;
  return CPT_OK;
}

// stage entry points implemented in the other translation units
int cpt_transfer_impl(cpt_handle* h, const double* sources_dev, const double* k, int nk, int k_size_cl,
                      const double* tau_sampling, int ntau, const double* q, int nq, const int* l, int nl,
                      double* transfer_dev);
int cpt_lensing_l_size_impl(const int* l, int nl, const cpt_lensing_params* lp);
int cpt_lensing_impl(cpt_handle* h, const cpt_spectra_params* sp, const cpt_lensing_params* lp, const int* l, int nl,
                     const double* cl_dev, double* cl_lensed_dev);
int cpt_bessel_build(cpt_handle* h, const int* l, int nl, double xmax);
int cpt_perturb_impl(cpt_handle* h, const double* k, int nk, const double* tau_sampling, int ntau, double* sources_dev,
                     cpt_stepstat* stats, int* status);
// the register-set kernels (cpt_perturb_sets.inc, one translation unit per family): scalars with non-cold species / hierarchies longer than one wavefront
#define CPT_SETS_LAUNCH_ARGS cpt_handle* h, const double* d_k, const double* d_tau, const int* d_order, int nk, int ntau, double* d_src, cpt_stepstat* d_stats, int* d_status
int cpt_perturb_sets_launch_0(CPT_SETS_LAUNCH_ARGS);   // tails (long hierarchies)
int cpt_perturb_sets_launch_2(CPT_SETS_LAUNCH_ARGS);   // <= 2 momentum-bin sets
int cpt_perturb_sets_launch_5(CPT_SETS_LAUNCH_ARGS);   // <= 5 momentum-bin sets
int cpt_perturb_sets_launch_13(CPT_SETS_LAUNCH_ARGS);  // tails + <= 3 momentum-bin sets
double cpt_sigma_of_R(const double* k, const double* pk, int nk, double R, double k_per_decade);   // host: sigma(R) of a tabulated P(k)
int cpt_cl_impl(cpt_handle* h, const cpt_spectra_params* sp, const double* transfer_dev, const double* q, int nq, int nl,
                double* cl_dev, const double* transfer2_dev = nullptr);
int cpt_pk_impl(cpt_handle* h, const cpt_spectra_params* sp, const double* k, int nk, double* pk_dev, int cb);
int cpt_pk_at_tau_impl(cpt_handle* h, const cpt_spectra_params* sp, const double* k, int nk, int ln_tau_size, double tau_z, int cb, double* pk_dev);
int cpt_sigma_at_tau_impl(cpt_handle* h, const cpt_spectra_params* sp, const double* k, int nk, int ln_tau_size, double tau_z, int cb, double R,
                          double k_per_decade, double* sigma);
int cpt_sigma_impl(cpt_handle* h, const cpt_spectra_params* sp, const double* k, int nk, double R, double k_per_decade, double* sigma, int cb);
int cpt_dbg_lookup_impl(cpt_handle* h, const double* tau, int n, double* out);
int cpt_dbg_derivs_impl(cpt_handle* h, double k, double tau, int tca_on, int rsa_on, int ufa_on, const double* y,
                        double* dy, int* neq);
int cpt_dbg_solve_impl(cpt_handle* h, double k, double tau, int tca_on, int rsa_on, int ufa_on, double hg, const double* b,
                       double* x);
int cpt_transpose_to_kmajor(cpt_handle* h, const double* src_ref_layout, double* dst, int ntp, int ntau, int nk);
int cpt_transpose_from_kmajor(cpt_handle* h, const double* src_kmajor, double* dst_ref_layout, int ntp, int ntau, int nk);
