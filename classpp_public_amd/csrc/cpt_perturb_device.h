// Device code of hot path A, shared by the translation units that instantiate its kernels (cpt_perturb.hip: the two-wave kernels of the
// massless configurations and the unit-test kernels; cpt_perturb_sets_*.hip: the kernels whose k-mode needs more than 64 equations).
// Everything lives in an anonymous namespace: each translation unit compiles its own copy.
#pragma once
// Hot path A on MI355X: per-k stiff integration of the scalar Einstein-Boltzmann system.
//
// ONE WAVEFRONT OWNS ONE k-MODE (a second wave of the block helps it: table look-ups and source samples).  Lane i owns equation i of the current regime:
// the state y, the backward differences dif[0..6], the Newton iterates, the Jacobian and the factors of
// (I - h*gamma*J) all live in lane registers (the matrix is never stored densely: see "structured linear algebra");
// the adaptive order/step control is scalar control flow that is uniform in the wave, so divergence between modes
// never crosses a wavefront.  The background / thermodynamics spline tables are read through a 64-row window staged
// in LDS whose abscissae sit in lane registers (wave-parallel bracket search by ballot+popcount) plus a row cache, so
// a step that stays inside the current table cell touches no memory at all.
//
// Restates (not translates): perturb_solve pm.cpp:2463-2787, perturb_approximations :5443-5670,
// perturb_vector_init :3271-4688, perturb_initial_conditions :4723-5408, perturb_einstein/total_stress_energy
// :5840-6703, perturb_derivs :7861-9218, perturb_tca_slip_and_shear :9229-9516, perturb_rsa_delta_and_theta
// :9530-9636, perturb_sources :6731-7285, background_at_tau / thermodynamics_at_z, and evolver_ndf15
// ev.cpp:62-705 (+ interp_from_dif :860-905, adjust_stepsize :907-943, new_linearisation :945-998).
// Differences by design: the Jacobian is obtained exactly as J e_j = f(tau, e_j) (the system is linear and
// homogeneous in y) instead of by adaptive finite differences (ev.cpp:1213-1539); the factorisation uses the
// structure of the equations (three tridiagonal hierarchy tails + a <=16x16 dense core) instead of a numerically
// discovered sparsity pattern with AMD ordering (tools/sparse.c:130-599); switch times are located by a 64-ary
// search instead of bisection.
#include "cpt_internal.h"

namespace {

constexpr double SIGMA_T = 6.6524616e-29, MPC_OVER_M = 3.085677581282e22, K_B = 1.3806504e-23, C_LIGHT = 2.99792458e8,
                 M_H = 1.673575e-27, NOT4 = 3.9715;

// (diagnostic) comment lines in the ISA listing (hipcc -S -DCPT_ISA_MARKS): where a section of the integrator begins and ends
#ifdef CPT_ISA_MARKS
#define ISA_MARK(name) asm volatile("; ==== " name)
#else
#define ISA_MARK(name)
#endif

struct PtParams {
  DevTables tabs;
  // config scalars
  int has_cdm, has_ur, tca_method, rsa_method, ufa_method, l_max_g, l_max_pol_g, l_max_ur;
  int rows;                    // scalars: every tail fits a 16-lane row of its own (lanes 16.., 32.., 48..) => log-depth tail solves
  double T_cmb, a_today, YHe, n_e, tau_free_streaming;
  double K;  // spatial curvature (pba->K); 0 in flat space
  int gauge;                   // CPT_GAUGE_NEWTONIAN / CPT_GAUGE_SYNCHRONOUS
  int l_max_g_ten, l_max_pol_g_ten, evolve_tensor_ur; double gw_ini;  // tensor modes
  int ic; double entropy_ini;  // initial condition of the mode (CPT_IC_*), isocurvature normalisation
  int has_ncdm, nfa_method, tp_dcb; double nfa_trig, tol_ncdm_w;  // non-cold species (massive neutrinos)
  int long_tails;              // hierarchies longer than one wavefront: the three l >= 3 tails live on chain waves of their own (see "long tails")
  int long_len;                // ... and the longest of them
  NcdmDev nc;
  int switch_sw, switch_eisw, switch_lisw, switch_dop, switch_pol;
  double eisw_lisw_split_z, three_ceff2_ur, three_cvis2_ur;
  int tp_size, tp_t0, tp_t1, tp_t2, tp_p, tp_dm, tp_pp;
  int tp_tk[CPT_NTK];   // density / velocity transfer sources (cpt_config::index_tp_transfer; all -1 unless has_transfers)
  int tp_dn, tp_tn;     // ... of the non-cold species: first of n_species consecutive slots each (index_tp_delta_ncdm1, index_tp_theta_ncdm1)
  double start_small_k, start_large_k, tca_trig_h, tca_trig_k, rsa_trig, ufa_trig, curvature_ini, rtol, tol_tau_approx, min_var;
  // batch
  const double* k;
  const double* tau_s;
  const int* order;  // block -> mode index (heaviest first)
  int nk, ntau;
  double* src;  // [tp][nk][ntau]
  cpt_stepstat* stats;
  int* status;
  int max_steps;
  int dbg_inverse;   // (unit-test kernel k_dbg_solve) solve the core block in the product form of helper_inverse
};

// ---- wave helpers -------------------------------------------------------------------------------
__device__ inline double bcast(double v, int lane) {  // lane is wave-uniform
  int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
__device__ inline double first(double v) {
  int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
  int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double(hi, lo);
}
__device__ inline int ufirst(int v) { return __builtin_amdgcn_readfirstlane(v); }
// DPP row_shr:n on a double (lanes without a source keep their own value)
template <int CTRL>
__device__ inline double dpp_keep(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
// DPP row_shr:n within each row of 16 lanes; lanes without a source get 0 (bound_ctrl: the hardware writes the zero, no
// register has to be cleared first)
template <int N>
__device__ inline double row_shr0(double v) {
  int l2 = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x110 + N, 0xf, 0xf, true);
  int h2 = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x110 + N, 0xf, 0xf, true);
  asm volatile("" : "+v"(l2), "+v"(h2));
  return __hiloint2double(h2, l2);
}
// DPP row_shl:n: value of lane + n of the same row, 0 beyond the row
template <int N>
__device__ inline double row_shl0(double v) {
  int l2 = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x100 + N, 0xf, 0xf, true);
  int h2 = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x100 + N, 0xf, 0xf, true);
  asm volatile("" : "+v"(l2), "+v"(h2));
  return __hiloint2double(h2, l2);
}
// v >= 0 as a float that is >= v: round-to-nearest conversion + one ulp up (integer increment of a non-negative float, saturated
// at +inf) - 3 instructions, where the directed-rounding conversion __double2float_ru is a 12-instruction software sequence
__device__ inline float f32_up(double v) { return __int_as_float(min(__float_as_int((float)v) + 1, 0x7f800000)); }
__device__ inline double wave_max(double v) {
  int x = __float_as_int(f32_up(v)), t;
  asm volatile(
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
      "s_nop 1\n\t"
      "v_readlane_b32 %1, %0, 63"
      : "+v"(x), "=s"(t));
  return (double)__int_as_float(t);
}
// A condition that is the same in every lane (all control flow of the integrator is), made PROVABLY uniform: the compiler then
// branches on the scalar unit instead of masking EXEC and merging every variable of the two arms with v_cndmask.
__device__ inline bool uni(bool c) { return __builtin_amdgcn_ballot_w64(c) != 0ull; }
// max over the wave of v (>= 0) <= thr, without forming the maximum: one compare + one scalar test instead of a
// conversion, six DPP steps with their wait states and a readlane.  A NaN lane counts as "not below".
__device__ inline bool wave_all_le(double v, double thr) { return __builtin_amdgcn_ballot_w64(!(v <= thr)) == 0ull; }
// Cross-lane reads must execute with EVERY lane active: a DPP / bpermute source lane that is masked off by EXEC
// yields 0.  Never call these inside a lane-dependent branch or the lazy arm of a ?: - hoist the call into its own
// statement.  The volatile asm additionally stops the compiler from sinking the (side-effect free) instruction into
// a divergent arm of a later select.
__device__ inline double pin(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  asm volatile("" : "+v"(lo), "+v"(hi));
  return __hiloint2double(hi, lo);
}
// neighbours in the wave (hierarchy couplings): wave_shr:1 / wave_shl:1, lanes without a source get 0
__device__ inline double lane_below(double v) {  // value of lane-1
  int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x138, 0xf, 0xf, false);
  int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x138, 0xf, 0xf, false);
  return pin(__hiloint2double(hi, lo));
}
__device__ inline double lane_above(double v) {  // value of lane+1
  int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x130, 0xf, 0xf, false);
  int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x130, 0xf, 0xf, false);
  return pin(__hiloint2double(hi, lo));
}
// ds_bpermute with a per-lane source index, pinned for the same reason
__device__ inline double shfl_all(double v, int src) { return pin(__shfl(v, src, 64)); }
// x^p for the step-size heuristics (ev.cpp:497-505, 580-625): single precision is ample (the result only steers h)
// (the bare v_log_f32 / v_exp_f32 / v_rcp_f32: the library forms add subnormal scaling and an IEEE division, ~25 instructions)
__device__ inline double fast_root(double x, int n) { return (double)__builtin_amdgcn_exp2f(__builtin_amdgcn_logf((float)x) * __builtin_amdgcn_rcpf((float)n)); }  // x^(1/n)
__device__ inline double fast_powi(double x, int n) { return (double)__builtin_amdgcn_exp2f(__builtin_amdgcn_logf((float)x) * (float)n); }                // x^n
// 1/x: hardware v_rcp_f64 seed + two Newton-Raphson refinements (full double accuracy to ~1 ulp, no IEEE division
// expansion with its denormal/scale handling on the critical path)
// Hide a lane-dependent integer from loop-invariant code motion: without this, hipcc hoists the dozens of
// `lane > j` / `role == X` masks of the hot inline functions out of the step loop, keeps them in SGPR pairs, runs
// out of SGPRs and spills them into VGPR lanes (two v_readlane + exec juggling per use).
__device__ inline int opaque(int v) {
  asm volatile("" : "+v"(v));
  return v;
}
// 1 / x: v_rcp_f64 is good to 4.6e-8, one Newton step makes it 2.1e-15 (19 ulp), a second one 1.1e-16 (tools/rcp/rcp_test.hip on gfx950).
// Both steps stay: with one, the look-ups lose their 1e-15 agreement with the reference's tables and the integrator takes 1 - 2 % MORE
// steps (more error-test failures: 2 578 -> 2 625 on explanatory_mpk) - the 18 cycles saved per chain do not pay for that.
__device__ inline double fast_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(r, fma(-x, r, 1.0), r);
  r = fma(r, fma(-x, r, 1.0), r);
  return r;
}
// sqrt(x), x > 0: hardware v_rsq_f64 seed + two Goldschmidt refinements (no IEEE sqrt expansion on a critical path)
__device__ inline double fast_sqrt(double x) {
  const double r0 = __builtin_amdgcn_rsq(x);
  double g = x * r0, h = 0.5 * r0;
  double rr = fma(-h, g, 0.5);
  g = fma(g, rr, g); h = fma(h, rr, h);
  rr = fma(-h, g, 0.5);
  g = fma(g, rr, g);
  return g;
}
// NDF constants (ev.cpp:87-88, 171-174) as immediates: no private arrays, no scratch
__device__ inline double ndf_G(int i) { return i == 0 ? 1.0 : i == 1 ? 1.5 : i == 2 ? 11.0 / 6.0 : i == 3 ? 25.0 / 12.0 : 137.0 / 60.0; }
__device__ inline double ndf_alpha(int i) { return i == 0 ? -37.0 / 200 : i == 1 ? -1.0 / 9.0 : i == 2 ? -8.23e-2 : i == 3 ? -4.15e-2 : 0.; }
// (1 / (G (1 - alpha)) and alpha G + 1 / (k + 2): folded at compile time with the same IEEE arithmetic - a run-time division on every step otherwise)
constexpr double ndf_G_c(int i) { return i == 0 ? 1.0 : i == 1 ? 1.5 : i == 2 ? 11.0 / 6.0 : i == 3 ? 25.0 / 12.0 : 137.0 / 60.0; }
constexpr double ndf_alpha_c(int i) { return i == 0 ? -37.0 / 200 : i == 1 ? -1.0 / 9.0 : i == 2 ? -8.23e-2 : i == 3 ? -4.15e-2 : 0.; }
constexpr double ndf_invGa_c(int i) { return 1.0 / (ndf_G_c(i) * (1.0 - ndf_alpha_c(i))); }
constexpr double ndf_erconst_c(int i) { return ndf_alpha_c(i) * ndf_G_c(i) + 1.0 / (2.0 + i); }
__device__ inline double ndf_invGa(int i) {
  constexpr double c0 = ndf_invGa_c(0), c1 = ndf_invGa_c(1), c2 = ndf_invGa_c(2), c3 = ndf_invGa_c(3), c4 = ndf_invGa_c(4);
  return i == 0 ? c0 : i == 1 ? c1 : i == 2 ? c2 : i == 3 ? c3 : c4;
}
__device__ inline double ndf_erconst(int i) {
  constexpr double c0 = ndf_erconst_c(0), c1 = ndf_erconst_c(1), c2 = ndf_erconst_c(2), c3 = ndf_erconst_c(3), c4 = ndf_erconst_c(4), c5 = ndf_erconst_c(5);
  return i == 0 ? c0 : i == 1 ? c1 : i == 2 ? c2 : i == 3 ? c3 : i == 4 ? c4 : c5;
}

enum Role : int {
  R_NONE = 0, R_DELTA_G, R_THETA_G, R_SHEAR_G, R_LG /* l>=3 photon temperature */, R_POL /* l>=0 polarisation */,
  R_DELTA_B, R_THETA_B, R_DELTA_CDM, R_DELTA_UR, R_THETA_UR, R_SHEAR_UR, R_LUR /* l>=3 ur */, R_ETA, R_THETA_CDM, R_GW, R_GWDOT,
  R_FLUID /* delta, theta or sigma of a non-cold species in the fluid approximation, held by the core wave */,
  R_NCD, R_NCT /* auxiliary unknowns of the Newton system: delta rho and (rho+p) theta summed over the ncdm species */
};

#ifdef CPT_COUNT_RESTAGE
__device__ unsigned long long g_restage[4];   // diagnostic: window slides (thermo, background), bsearch fallbacks, lookups
#endif
#ifdef CPT_PROFILE
__device__ unsigned long long g_prof[16];
__device__ unsigned long long g_prof_helper[16];   // the helper wave of the heaviest mode: cycles and counts of look-ups, inversions, samples; idle turns
#define PROF_DECL unsigned long long pf_t0 = 0
#define PROF_START() pf_t0 = clock64()
#define PROF_STOP(slot) prof[slot] += clock64() - pf_t0
#else
#define PROF_DECL
#define PROF_START()
#define PROF_STOP(slot)
#endif

// Everything from the lane map to the kernel bodies is a class template on the gauge: the Newtonian gauge has one more
// core variable (theta_cdm: NC = 14 instead of 13) and different metric terms; compiled into the synchronous kernel as
// run-time branches it pushed the step loop over its register budget (672 B/lane of scratch, 21 -> 34 ms).  Two
// instantiations cost code size only.  The same holds for non-flat space (CURV): the s_l factors, k cotK(tau) and the
// separate 1/tau coefficient cost the flat kernel 35 % when they were run-time values; in the flat instantiation they fold
// to 1, 1/tau and nothing.
// NCDM = 1: scalars with non-cold species (massive neutrinos); NCDM = 2: hierarchies longer than one wavefront.  Both lane maps have
// nine more core lanes (auxiliary unknowns of the bordered Newton system, or the ncdm fluids) and run in the register-set kernels of
// cpt_perturb_sets.inc: the momentum-bin hierarchies / the long tails are further register sets of the ONE wave that owns the k-mode.

// (NCDM = 0) The block holds TWO wavefronts per k-mode: wave 0 integrates, wave 1 is its HELPER.  Two jobs are taken off the
// integrator's dependency chain - the one thing that sets the run time of the launch - and done concurrently on the second SIMD:
//  * the table look-ups.  background_at_tau + thermodynamics_at_z is a chain of ~150 dependent instructions (two bracket
//    searches, two row fetches, two splines, a dozen reciprocals) that depends on tau alone.  The integrator therefore ASKS for the
//    row of the time it will need next - as soon as that time is known, i.e. one whole step ahead in the common case that the step
//    size stays - and finds the answer (22 wave-uniform doubles) in LDS when it gets there.  The integrator itself owns no table
//    windows at all.
//  * the source samples.  Evaluating a sample costs a look-up at the sample time, one RHS evaluation and the source algebra, none
//    of which feeds back into the integration: wave 0 only interpolates (y, y') from its backward differences and posts them in a
//    ring of NSLOT slots, so bursts of samples inside one step do not stall it.
// No barriers: both directions are single-producer / single-consumer counters in LDS (release / acquire at workgroup scope, LDS
// executes a wave's operations in order).  The helper never waits for the integrator except by polling, and leaves when `done` is
// set and the ring is drained; the integrator only waits for work the helper is certain to finish: no cycle, every wave exits.
constexpr int MB_NSLOT = 4;      // sample ring
constexpr int MB_NANS = 32;      // doubles of a look-up answer (22 + {rho, p, pseudo_p} of up to three non-cold species)
struct Mailbox {
  double yi[MB_NSLOT][64], ypi[MB_NSLOT][64];   // dense output at the sample time, one entry per lane
  double tca_keep[MB_NSLOT];                    // tight-coupling shear left by the evolver's last RHS call (pm.cpp:6810)
  int it[MB_NSLOT], flags[MB_NSLOT];            // sample index; approximation scheme (tca | rsa<<1 | ufa<<2)
  // what the helper polls, in one 16-byte read (mb_poll): look-ups requested, factorisations posted (see helper_inverse), samples posted,
  // the mode is finished - all written by the integrator
  alignas(16) int req_seq; int fact_seq, head, done;
  int tail, ans_seq, inv_seq;                   // samples consumed, look-ups answered, factorisations inverted (helper)
  double req_tau[2];                            // time of request n in req_tau[n & 1]: up to two requests may be waiting (this step's row after a
                                                // change of step size and the next step's), the helper answers them in order
  double rp[32];                                // reciprocal pivots of the posted factorisation (the rest of it is the integrator's LuReg::fw)
  alignas(16) double ans[2][MB_NANS];           // the answer to request n goes to ans[n & 1]: the integrator reads the row of the step it is
                                                // working on from LDS (it is never copied to registers) while the helper fills the other half
                                                // with the row of the step after it
};
__device__ inline int mb_load(const int* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
// (no wait is attached to this load: the value is looked at later, behind other work - see mb_take)
__device__ inline int mb_peek(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
// the four counters the helper polls, one LDS round trip; what they announce is read after an acquire fence
struct MbPoll { int req_seq, fact_seq, head, done; };
__device__ inline MbPoll mb_poll(const Mailbox* mb) {
  MbPoll c;
  c.req_seq = mb_peek(&mb->req_seq); c.fact_seq = mb_peek(&mb->fact_seq); c.head = mb_peek(&mb->head); c.done = mb_peek(&mb->done);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  return c;
}
__device__ inline void mb_store(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }

template <int GAUGE, int CURV, int MODE, int NCDM = 0, int ROWS = 0>
struct PT {
static constexpr bool SAMPLER = (NCDM == 0);
// NCDM = 2: the multi-wavefront machinery of the non-cold species with ZERO species and the chain waves carrying the three l >= 3 tails
// instead ("long tails": hierarchies longer than one wavefront).  An instantiation of its own, so that the ncdm kernels proper pay
// nothing for it (as run-time branches it cost them 6 - 10 %).
static constexpr bool LONG = (NCDM & 2) != 0;       // NCDM = 2: long tails alone; NCDM = 3: long tails AND momentum bins (permille-class hierarchies with massive neutrinos)
static constexpr bool HAS_BINS = (NCDM & 1) != 0;
// Lane map.  One lane per equation of pm.cpp:3302-3481, at a FIXED lane whatever the approximation scheme: the (at most
// 13) densely coupled variables - densities, velocities, shears, polarisation l<=2, metric - are the CORE in lanes
// 0..12, followed by the three free-streaming hierarchy tails (photon temperature l>=3, polarisation l>=3, ur l>=3)
// in ascending l.  A variable that the current scheme does not evolve keeps its lane with y = 0 and dy = 0 (an identity
// row of the Newton matrix).  Fixed lanes make every broadcast of a named component a v_readlane with an immediate
// lane number - no index arithmetic, no SGPRs holding a layout - and the hand-over between schemes the identity.
// A tail is a tridiagonal chain that touches the core only through its l=3 element (l3 <-> shear / pol2): that
// structure, fixed per regime and shared by all modes, is what the linear algebra below exploits.
// LN_ETA holds eta (synchronous gauge) or phi (Newtonian gauge, pm.cpp:3470-3478); LN_TC = theta_cdm exists in the Newtonian gauge only
enum Lane : int { LN_DG = 0, LN_TG, LN_SG, LN_P0, LN_P1, LN_P2, LN_DB, LN_TB, LN_DC, LN_DUR, LN_TUR, LN_SUR, LN_ETA, LN_TC };
// (NCDM: nine more core lanes, 13..21.  While the momentum hierarchies are integrated the first two hold the auxiliary unknowns of the
//  bordered Newton system; once the ncdm fluid approximation is on and the core wave integrates alone they hold (delta, theta, sigma)
//  of up to three species - see "the core wave alone" below.  An idle core lane costs nothing: Layout::pmask.)
static constexpr int NC = MODE ? 17 : ((GAUGE == CPT_GAUGE_NEWTONIAN) ? 14 : 13) + (NCDM ? 3 * CPT_MAX_NCDM : 0) + (NCDM == 3 ? 3 : 0);
static constexpr int LN_ND = 13, LN_NT = 14;   // (momentum bins) auxiliary unknowns: ncdm density / momentum sums
static constexpr int LN_T3 = (NCDM == 3) ? 22 : 13;   // (long tails) auxiliary unknowns: the l = 3 elements of the three tails, lanes LN_T3 .. LN_T3 + 2
static constexpr int LN_F0 = 13;               // (NCDM only, fluids in the core) delta of species 0; species n, moment j at LN_F0 + 3 n + j
static_assert(!NCDM || CPT_MAX_NCDM == 3, "lane map of the ncdm kernels");
// Tensor modes (MODE = 1; pm.cpp:3519-3586): the same three ladders plus the gravitational wave (gw, gw').  The photon
// source P^(2) reads the l = 4 multipoles of temperature and polarisation and the gravitational-wave source reads the
// l = 4 multipoles of photons and ur, so the core holds every ladder up to l = 4 and the tails start at l = 5.
enum TLane : int { TL_DG = 0, TL_TG, TL_SG, TL_G3, TL_G4, TL_P0, TL_P1, TL_P2, TL_P3, TL_P4, TL_DUR, TL_TUR, TL_SUR, TL_U3, TL_U4, TL_GW, TL_GWD };
static constexpr int LFIRST = MODE ? 5 : 3;   // multipole of the first element of a tail
// ROWS: scalars without non-cold species whose three tails each fit a 16-lane row (the host decides, PtParams::rows): tail of the
// photon temperature on lanes 16.., polarisation 32.., ur 48.., and the tail solves become four-level cyclic reductions on row
// DPP instead of 2 x maxlen dependent sweeps.  (Not for the ncdm kernels: their 256-register build cannot afford the eight
// extra doubles per lane - three species 142 -> 165 ms - and the one-species kernel gains nothing, its critical path is elsewhere.)
static constexpr bool PCR = (ROWS != 0) && (MODE == 0) && (NCDM == 0);

struct Layout {
  int tca, rsa, ufa, nfa;
  int fic;                     // (NCDM) the ncdm fluids live in core lanes LN_F0.. of the core wave (which then integrates alone)
  int lng;                     // (NCDM kernels, long tails) the tails live on chain waves; core lanes 13..15 hold their l = 3 elements as auxiliary unknowns
  int g3, gN, q3, qN, u3, uN;  // tails: lane of l=3 and length (lengths are 0 when the scheme drops the tail)
  int lmg, lmp, lmu;
  int maxlen;                  // longest tail present
  int ur_special;              // the rarely used ur variants of the RHS are on: three_ceff2_ur != 1 (pm.cpp:8630-8641) or ufa_hu (pm.cpp:8711-8716)
};

static __device__ __forceinline__ Layout make_layout(const PtParams& P, int tca, int rsa, int ufa, int nfa = 0, int fic = 0) {
  Layout L;
  L.tca = tca; L.rsa = rsa; L.ufa = ufa; L.nfa = nfa; L.fic = fic; L.lng = 0;
  L.ur_special = (P.three_ceff2_ur != 1. || (ufa && P.ufa_method == CPT_UFA_HU)) ? 1 : 0;
  if (MODE) {  // tensors: photons are evolved when neither approximation is on; ur always (pm.cpp:3529-3560)
    L.ufa = 0;
    L.lmg = P.l_max_g_ten; L.lmp = P.l_max_pol_g_ten; L.lmu = P.l_max_ur;
    const bool hi = !rsa && !tca;
    L.g3 = NC; L.q3 = L.g3 + (L.lmg - 4); L.u3 = L.q3 + (L.lmp - 4);
    L.gN = hi ? L.lmg - 4 : 0;
    L.qN = hi ? L.lmp - 4 : 0;
    L.uN = P.evolve_tensor_ur ? L.lmu - 4 : 0;
    L.maxlen = max(L.gN, max(L.qN, L.uN));
    return L;
  }
  L.lmg = P.l_max_g; L.lmp = P.l_max_pol_g; L.lmu = P.l_max_ur;
  L.g3 = NC; L.q3 = L.g3 + (P.l_max_g - 2); L.u3 = L.q3 + (P.l_max_pol_g - 2);
  if (PCR) { L.g3 = 16; L.q3 = 32; L.u3 = 48; }   // one tail per 16-lane row: the tail solves are cyclic reductions on row DPP
  const bool hi = !rsa && !tca;
  L.gN = hi ? P.l_max_g - 2 : 0;
  L.qN = hi ? P.l_max_pol_g - 2 : 0;
  L.uN = (P.has_ur && !rsa && !ufa) ? P.l_max_ur - 2 : 0;
  L.maxlen = max(L.gN, max(L.qN, L.uN));
  L.lng = LONG ? 1 : 0;
  if (L.lng) { L.g3 = L.q3 = L.u3 = 64; L.maxlen = 0; }   // no tail lane in the core wave (gN, qN, uN still say which tails exist)
  return L;
}

// is core variable `i` evolved in this scheme?  (i wave-uniform)
static __device__ __forceinline__ bool core_present(const PtParams& P, const Layout& L, int i) {
  if (MODE) return (i <= TL_P4) ? (!L.rsa && !L.tca) : (i <= TL_U4) ? (P.evolve_tensor_ur != 0) : true;
  if (LONG && i >= LN_T3 && i < LN_T3 + 3) return (i == LN_T3) ? L.gN > 0 : (i == LN_T3 + 1) ? L.qN > 0 : L.uN > 0;
  if (NCDM && i >= LN_ND) return !HAS_BINS ? false : L.fic ? (i - LN_F0 < 3 * P.nc.n_species) : (i <= LN_NT);
  switch (i) {
    case LN_DG: case LN_TG: return !L.rsa;
    case LN_SG: case LN_P0: case LN_P1: case LN_P2: return !L.rsa && !L.tca;
    case LN_DC: return P.has_cdm != 0;
    case LN_TC: return GAUGE == CPT_GAUGE_NEWTONIAN && P.has_cdm != 0;   // (lane 13 is the first tail lane in the synchronous kernel)
    case LN_DUR: case LN_TUR: case LN_SUR: return P.has_ur && !L.rsa;
    default: return true;  // delta_b, theta_b, eta
  }
}

static __device__ __forceinline__ unsigned present_mask(const PtParams& P, const Layout& L) {
  unsigned m = 0;
#pragma unroll
  for (int i = 0; i < NC; i++) if (core_present(P, L, i)) m |= 1u << i;
  return (unsigned)__builtin_amdgcn_readfirstlane((int)m);
}

// (role, multipole) of lane i in the current scheme; R_NONE = not evolved
static __device__ __forceinline__ void role_of(const PtParams& P, const Layout& L, int i, int* role, int* ell) {
  *role = R_NONE; *ell = 0;
  if (MODE) {
    const bool hi = !L.rsa && !L.tca, ur = P.evolve_tensor_ur != 0;
    if (i < 0) return;
    if (i <= TL_G4) { if (hi) { *role = (i == TL_DG) ? R_DELTA_G : (i == TL_TG) ? R_THETA_G : (i == TL_SG) ? R_SHEAR_G : R_LG; *ell = i; } return; }
    if (i <= TL_P4) { if (hi) { *role = R_POL; *ell = i - TL_P0; } return; }
    if (i <= TL_U4) { if (ur) { const int l = i - TL_DUR; *role = (l == 0) ? R_DELTA_UR : (l == 1) ? R_THETA_UR : (l == 2) ? R_SHEAR_UR : R_LUR; *ell = l; } return; }
    if (i == TL_GW) { *role = R_GW; return; }
    if (i == TL_GWD) { *role = R_GWDOT; return; }
    if (i >= L.g3 && i < L.g3 + L.gN) { *role = R_LG; *ell = 5 + (i - L.g3); return; }
    if (i >= L.q3 && i < L.q3 + L.qN) { *role = R_POL; *ell = 5 + (i - L.q3); return; }
    if (i >= L.u3 && i < L.u3 + L.uN) { *role = R_LUR; *ell = 5 + (i - L.u3); return; }
    return;
  }
  const bool g = !L.rsa, hi = !L.rsa && !L.tca, ur = P.has_ur && !L.rsa;
  if (LONG && i >= LN_T3 && i < LN_T3 + 3) { if (core_present(P, L, i)) *role = R_NCD; return; }
  if (NCDM && L.fic && i >= LN_F0 && i < NC) { if (i - LN_F0 < 3 * P.nc.n_species) *role = R_FLUID; return; }
  if (HAS_BINS && i == LN_ND) { *role = R_NCD; return; }
  if (HAS_BINS && i == LN_NT) { *role = R_NCT; return; }
  if (NCDM && i >= LN_ND && i < NC) return;   // (core lanes this lane map leaves idle)
  if (i == LN_DG) { if (g) *role = R_DELTA_G; return; }
  if (i == LN_TG) { if (g) { *role = R_THETA_G; *ell = 1; } return; }
  if (i == LN_SG) { if (hi) { *role = R_SHEAR_G; *ell = 2; } return; }
  if (i == LN_P0) { if (hi) { *role = R_POL; *ell = 0; } return; }
  if (i == LN_P1) { if (hi) { *role = R_POL; *ell = 1; } return; }
  if (i == LN_P2) { if (hi) { *role = R_POL; *ell = 2; } return; }
  if (i == LN_DB) { *role = R_DELTA_B; return; }
  if (i == LN_TB) { *role = R_THETA_B; return; }
  if (i == LN_DC) { if (P.has_cdm) *role = R_DELTA_CDM; return; }
  if (i == LN_DUR) { if (ur) *role = R_DELTA_UR; return; }
  if (i == LN_TUR) { if (ur) { *role = R_THETA_UR; *ell = 1; } return; }
  if (i == LN_SUR) { if (ur) { *role = R_SHEAR_UR; *ell = 2; } return; }
  if (i == LN_ETA) { *role = R_ETA; return; }
  if (GAUGE == CPT_GAUGE_NEWTONIAN && i == LN_TC) { if (P.has_cdm) *role = R_THETA_CDM; return; }
  if (i >= L.g3 && i < L.g3 + L.gN) { *role = R_LG; *ell = 3 + (i - L.g3); return; }
  if (i >= L.q3 && i < L.q3 + L.qN) { *role = R_POL; *ell = 3 + (i - L.q3); return; }
  if (i >= L.u3 && i < L.u3 + L.uN) { *role = R_LUR; *ell = 3 + (i - L.u3); return; }
}
// index of (role, ell) in the REFERENCE's ordering of the same regime (pm.cpp:3302-3481): only the unit-test hooks
// cpt_dbg_derivs / cpt_dbg_solve need it, to exchange vectors with the oracle in the reference's order
static __device__ __forceinline__ int ref_index_of(const PtParams& P, int tca, int rsa, int ufa, int role, int ell, int* neq) {
  if (MODE) {  // pm.cpp:3519-3586
    int i = 0, dg = -1, pol0 = -1, dur = -1, gw;
    if (!rsa && !tca) { dg = i; i += P.l_max_g_ten + 1; pol0 = i; i += P.l_max_pol_g_ten + 1; }
    if (P.evolve_tensor_ur) { dur = i; i += P.l_max_ur + 1; }
    gw = i; i += 2;
    *neq = i;
    switch (role) {
      case R_DELTA_G: case R_THETA_G: case R_SHEAR_G: case R_LG: return dg >= 0 ? dg + ell : -1;
      case R_POL: return pol0 >= 0 ? pol0 + ell : -1;
      case R_DELTA_UR: case R_THETA_UR: case R_SHEAR_UR: case R_LUR: return dur >= 0 ? dur + ell : -1;
      case R_GW: return gw;
      case R_GWDOT: return gw + 1;
      default: return -1;
    }
  }
  int i = 0, dg = -1, tg = -1, sg = -1, l3g = -1, pol0 = -1, db, tb, dc = -1, tc = -1, dur = -1, tur = -1, sur = -1, l3ur = -1, eta;
  if (!rsa) {
    dg = i++; tg = i++;
    if (!tca) { sg = i++; l3g = i; i += P.l_max_g - 2; pol0 = i; i += P.l_max_pol_g + 1; }
  }
  db = i++; tb = i++;
  if (P.has_cdm) { dc = i++; if (GAUGE == CPT_GAUGE_NEWTONIAN) tc = i++; }
  if (P.has_ur && !rsa) { dur = i++; tur = i++; sur = i++; if (!ufa) { l3ur = i; i += P.l_max_ur - 2; } }
  eta = i++;
  *neq = i;
  switch (role) {
    case R_DELTA_G: return dg;
    case R_THETA_G: return tg;
    case R_SHEAR_G: return sg;
    case R_LG: return l3g >= 0 ? l3g + ell - 3 : -1;
    case R_POL: return pol0 >= 0 ? pol0 + ell : -1;
    case R_DELTA_B: return db;
    case R_THETA_B: return tb;
    case R_DELTA_CDM: return dc;
    case R_THETA_CDM: return tc;
    case R_DELTA_UR: return dur;
    case R_THETA_UR: return tur;
    case R_SHEAR_UR: return sur;
    case R_LUR: return l3ur >= 0 ? l3ur + ell - 3 : -1;
    case R_ETA: return eta;
    default: return -1;
  }
}


// ---- spline tables ------------------------------------------------------------------------------
struct BgV { double a, H, Hp, rg, rb, rc, ru; };
struct ThV { double xe, dkappa, tau_d, ddkappa, dddkappa, expmk, g, dg, cb2; };

// per-thread (scalar) lookup with binary search: used by the schedule search, where every lane probes its own tau
static __device__ __forceinline__ int bsearch_up(const double* __restrict__ x, int n, double v) {  // arrays.c:1586-1594
  int inf = 0, sup = n - 1;
  while (sup - inf > 1) {
    int mid = (inf + sup) >> 1;
    if (v < x[mid]) sup = mid; else inf = mid;
  }
  return inf;
}
static __device__ __forceinline__ double spl2(const double2 lo, const double2 hi, double a, double b, double h2) {
  return a * lo.x + b * hi.x + ((a * a * a - a) * lo.y + (b * b * b - b) * hi.y) * h2;
}
// a, H and dkappa at tau (what perturb_approximations and the start-time search need)
struct AHK { double a, H, dk, wdev; };   // wdev: max over the ncdm species of |p/rho - 1/3| (0 without ncdm)
static __device__ __noinline__ AHK lookup_aHk(DevTables T, double n_e, double tau) {
  int inf = bsearch_up(T.tau_table, T.bt_size, tau);
  double h = T.tau_table[inf + 1] - T.tau_table[inf], b = (tau - T.tau_table[inf]) / h, a = 1. - b, h2 = h * h / 6.;
  const double2* r0 = (const double2*)T.bg + (size_t)inf * BG_NCOL;
  const double2* r1 = r0 + BG_NCOL;
  double av = spl2(r0[BG_A], r1[BG_A], a, b, h2), Hv = spl2(r0[BG_H], r1[BG_H], a, b, h2);
  double z = 1. / av - 1.;
  double dk;
  if (z >= T.z_table[T.tt_size - 1]) {
    double x0 = ((const double2*)T.th)[(size_t)(T.tt_size - 1) * TH_NCOL + TH_XE].x;
    dk = (1. + z) * (1. + z) * n_e * x0 * SIGMA_T * MPC_OVER_M;
  } else {
    int iz = bsearch_up(T.z_table, T.tt_size, z);
    double hz = T.z_table[iz + 1] - T.z_table[iz], bz = (z - T.z_table[iz]) / hz, az = 1. - bz;
    const double2* t0 = (const double2*)T.th + (size_t)iz * TH_NCOL;
    dk = spl2(t0[TH_DKAPPA], t0[TH_NCOL + TH_DKAPPA], az, bz, hz * hz / 6.);
  }
  AHK r;
  r.a = av; r.H = Hv; r.dk = dk; r.wdev = 0.;
  if (T.ncb) {   // pm.cpp:2574-2603: every non-cold species must still be ultra-relativistic at the initial time
    const double2* n0 = (const double2*)T.ncb + (size_t)inf * NCB_NCOL;
    for (int n = 0; n < CPT_MAX_NCDM; n++) {
      const double rho = spl2(n0[3 * n], n0[NCB_NCOL + 3 * n], a, b, h2), pr = spl2(n0[3 * n + 1], n0[NCB_NCOL + 3 * n + 1], a, b, h2);
      if (rho > 0.) r.wdev = fmax(r.wdev, fabs(pr / rho - 1. / 3.));
    }
  }
  return r;
}

// Rows of a table window.  64 = one row per lane; the register-set kernels, whose resident k-modes per CU are limited by LDS, stage
// half-windows (lanes >= 32 hold a +huge abscissa and are never selected): twice as many slides, half the LDS.
static constexpr int WR = NCDM ? 32 : 64;
static constexpr int wsize(int ncol) { return ((WR * ncol + 63) / 64) * 64; }   // double2 entries of a window of `ncol` columns in LDS (whole 64-entry DMA blocks)
// Wave-cooperative cached lookup used by the RHS / sampler (all arguments and results wave-uniform).
// A window of 64 consecutive table rows is staged in LDS (one coalesced copy when the wave walks out of it), its 64
// abscissae sit in lane registers: the bracket is one ballot+popcount, the two rows come from LDS, and a step that
// stays in the same table cell re-uses the rows already in registers.  Everything that depends on tau only
// (reciprocals, tight-coupling coefficients) is derived here once and shared by the Newton iterations and by the
// columns of the Jacobian, which all evaluate the RHS at the same tau.
struct Lookup {
  double bgx, thx;                 // lane l: abscissa of window row l (+huge past the table end)
  int bg_base, th_base, bg_inf, th_inf;
  double2 bg_lo, bg_hi, th_lo, th_hi;  // lane c: column c of rows inf / inf+1
  double2* bgw;                    // LDS [64][BG_NCOL]
  double2* thw;                    // LDS [64][TH_NCOL]
  double tau_cached;
  // lane c: column c of the background / thermodynamics row at tau_cached.  Values only the sampler needs (a, H',
  // e^-kappa, g, g') are extracted from these on demand instead of occupying registers through the step loop.
  double vbg, vth;
  double vnc;                      // (NCDM) lane c: column c of the ncdm background row {rho, p, pseudo_p} x species
  double2 nc_lo, nc_hi; double2* ncw;
  double rg, rb, rc, ru, kap, ddkappa, cb2;  // what every RHS evaluation needs (wave-uniform)
  // derived, tau-only
  double a2, aH, two_over_aH, R, inv_1pR, inv_R, tau_c, dtau_c, F, Fp, app, inv_tau, rg43, ru43;
  double zmax, xe_last, taud_last;  // last row of the thermodynamics table (analytic continuation beyond it)
  double k2s2, inv_k2s2, s2, s2sq, kcot;  // per-mode curvature factors (set_mode) and k cotK_gen(tau_cached); flat: k^2, 1/k^2, 1, 1, 1/tau
  // (integrator wave of the two-wave kernels) the rows come from the helper wave: mailbox, number of requests posted, time of the last one
  Mailbox* mb; int my_req; double rq_tau0, rq_tau1;   // requests posted so far; time of the last even / odd one
  int ans_seen;                      // last value read of the helper's ans_seq
  const double* row;                 // the row in work, in LDS where the helper left it (mb_take, mb_fetch_row) or this wave put it (row_store)
  double row_tau;                    // ... and its time (mb_fetch_row, row_store; tau_cached is the time of the row in the registers below)
  double ncv[NCDM ? NCB_NCOL : 1];   // (rows from the helper, ncdm kernels) {rho, p, pseudo_p} of every species, wave-uniform
#ifdef CPT_PROFILE
  unsigned long long* prof;
#endif
};

// The window travels HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4: 64 lanes x 16 B = one 1 KiB column block per
// instruction, no VGPR staging): all NCOL blocks and the abscissa load are in flight together and retire behind ONE wait.
// (Staged through registers the compiler re-used one 4-VGPR temporary under the kernel's register pressure and waited for every
// load before issuing the next: NCOL dependent L2 round trips per restage.)  Rows past the end of the table are clamped to the
// last row - their abscissa is +huge, so they are never selected.
template <int NCOL>
static __device__ __forceinline__ void window_stage(const double* __restrict__ x, const double2* __restrict__ rows, int n, int base, int lane,
                                             double* xw, double2* w) {
  const int i = base + lane;
  const double xv = x[min(i, n - 1)];
  const size_t g0 = (size_t)base * NCOL, glast = (size_t)n * NCOL - 1;
  // (the wave's own earlier LDS reads of this window have returned: they fed registers that were consumed before this call)
#pragma unroll
  for (int c = 0; c < (WR * NCOL + 63) / 64; c++)
    __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(rows + min(g0 + (size_t)(lane + 64 * c), glast)),
                                     (void __attribute__((address_space(3)))*)(w + 64 * c), 16, 0, 0);
  *xw = (i < n && (WR == 64 || lane < WR)) ? xv : 1e300;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the DMA writes have landed (and xv has arrived)
}

// returns inf with x[inf] <= v <= x[inf+1] (x ascending), re-staging the 64-row window when v leaves it.
// The integration walks through the tables monotonically, so the window that v left is almost always adjacent to the
// one it entered: slide by one window (keeping `bias` rows on the side the wave comes from) and only fall back to the
// binary search - 15 dependent global loads, microseconds - after a jump (first lookup, hand-over to a new mode).
template <int NCOL>
static __device__ __forceinline__ int window_find(const double* __restrict__ x, const double2* __restrict__ rows, int n, double v,
                                           int lane, double* xw, double2* w, int* base, int bias) {
  double lo = bcast(*xw, 0), hi = bcast(*xw, WR - 1);
  if (!(v >= lo && v < hi)) {
    int nb = (v >= hi) ? *base + (WR - 1) - bias : *base - (WR - 1) + ((WR - 1) - bias);   // slide up / down
    if (nb > n - WR) nb = n - WR;
    if (nb < 0) nb = 0;
    *base = nb;
    window_stage<NCOL>(x, rows, n, nb, lane, xw, w);
#ifdef CPT_COUNT_RESTAGE
    if (blockIdx.x == 0 && lane == 0) g_restage[NCOL == TH_NCOL ? 0 : 1]++;
#endif
    lo = bcast(*xw, 0); hi = bcast(*xw, WR - 1);
    if (!(v >= lo && v < hi) && !(nb == 0 && v < lo) && !(nb == n - WR && v >= hi)) {
      const int inf = bsearch_up(x, n, v);  // uniform
      nb = inf - bias;
      if (nb > n - WR) nb = n - WR;
      if (nb < 0) nb = 0;
      *base = nb;
      window_stage<NCOL>(x, rows, n, nb, lane, xw, w);
#ifdef CPT_COUNT_RESTAGE
      if (blockIdx.x == 0 && lane == 0) g_restage[2]++;
#endif
    }
  }
  const unsigned long long m = __ballot(*xw <= v);
  int inf = *base + __popcll(m) - 1;
  if (inf > n - 2) inf = n - 2;
  if (inf < 0) inf = 0;
  return inf;
}

static __device__ __forceinline__ void lookup_init(const PtParams& P, Lookup& Q, double2* bgw, double2* thw, int lane, double2* ncw = nullptr) {
  Q.bgw = bgw; Q.thw = thw; Q.ncw = ncw; Q.vnc = 0.; Q.nc_lo = Q.nc_hi = make_double2(0., 0.);
  Q.mb = nullptr; Q.my_req = 0; Q.rq_tau0 = Q.rq_tau1 = -1.; Q.ans_seen = 0; Q.row = nullptr; Q.row_tau = -1.;
  if (NCDM) { double dummy; window_stage<NCB_NCOL>(P.tabs.tau_table, (const double2*)P.tabs.ncb, P.tabs.bt_size, 0, lane, &dummy, ncw); }
  Q.bg_base = 0; Q.th_base = 0; Q.bg_inf = -1; Q.th_inf = -1; Q.tau_cached = -1.;
  window_stage<BG_NCOL>(P.tabs.tau_table, (const double2*)P.tabs.bg, P.tabs.bt_size, 0, lane, &Q.bgx, bgw);
  window_stage<TH_NCOL>(P.tabs.z_table, (const double2*)P.tabs.th, P.tabs.tt_size, 0, lane, &Q.thx, thw);
  Q.bg_lo = Q.bg_hi = Q.th_lo = Q.th_hi = make_double2(0., 0.);
  const double2* last = (const double2*)P.tabs.th + (size_t)(P.tabs.tt_size - 1) * TH_NCOL;
  Q.zmax = P.tabs.z_table[P.tabs.tt_size - 1];
  Q.xe_last = last[TH_XE].x;
  Q.taud_last = last[TH_TAU_D].x;
}

// per-mode constants of the curved-space equations (pm.cpp:2530-2533, 5856): s_2, s_2^2 = 1 - 3K/k^2
static __device__ __forceinline__ void lookup_set_mode(const PtParams& P, Lookup& Q, double k) {
  const double k2 = k * k;
  Q.s2sq = CURV ? 1. - 3. * P.K / k2 : 1.;
  Q.s2 = CURV ? sqrt(fmax(Q.s2sq, 0.)) : 1.;
  Q.k2s2 = k2 * Q.s2sq;
  Q.inv_k2s2 = 1. / Q.k2s2;   // (flat: k^2 and 1/k^2, the same values as the kernel's k2 / inv_k2)
}

// background_at_tau (normal_info, source/background_module.cpp:125-199) + thermodynamics_at_z (th.cpp:114-285)
static __device__ __forceinline__ void lookup(const PtParams& P, Lookup& Q, double tau, int lane) {
  if (tau == Q.tau_cached) return;
  Q.tau_cached = tau;
#ifdef CPT_COUNT_RESTAGE
  if (blockIdx.x == 0 && lane == 0) g_restage[3]++;
#endif
  const DevTables& T = P.tabs;
#ifdef CPT_PROFILE_LOOKUP
  unsigned long long tl0 = clock64();
#define LK_MARK(slot) { const unsigned long long tl1 = clock64(); Q.prof[slot] += tl1 - tl0; tl0 = tl1; }
#else
#define LK_MARK(slot)
#endif
  const int base_was = Q.bg_base;
  const int inf = window_find<BG_NCOL>(T.tau_table, (const double2*)T.bg, T.bt_size, tau, lane, &Q.bgx, Q.bgw, &Q.bg_base, 8);
  if (NCDM && Q.bg_base != base_was) {   // the ncdm columns ride in a window of their own on the same rows
    double dummy;
    window_stage<NCB_NCOL>(T.tau_table, (const double2*)T.ncb, T.bt_size, Q.bg_base, lane, &dummy, Q.ncw);
  }
  if (inf != Q.bg_inf) {
    Q.bg_inf = inf;
    if (lane < BG_NCOL) {
      const double2* r = Q.bgw + (inf - Q.bg_base) * BG_NCOL + lane;
      Q.bg_lo = r[0];
      Q.bg_hi = r[BG_NCOL];
    }
    if (NCDM && lane < NCB_NCOL) {
      const double2* r = Q.ncw + (inf - Q.bg_base) * NCB_NCOL + lane;
      Q.nc_lo = r[0];
      Q.nc_hi = r[NCB_NCOL];
    }
  }
  {
    const double x0 = bcast(Q.bgx, inf - Q.bg_base), x1 = bcast(Q.bgx, inf - Q.bg_base + 1);
    const double h = x1 - x0, b = (tau - x0) * fast_rcp(h), a = 1. - b;
    Q.vbg = spl2(Q.bg_lo, Q.bg_hi, a, b, h * h * (1.0 / 6.0));
    if (NCDM) Q.vnc = spl2(Q.nc_lo, Q.nc_hi, a, b, h * h * (1.0 / 6.0));
  }
  LK_MARK(12)   // background: window, rows, spline
  const double bg_a = bcast(Q.vbg, BG_A), bg_H = bcast(Q.vbg, BG_H), bg_Hp = bcast(Q.vbg, BG_HP);
  Q.rg = bcast(Q.vbg, BG_RHO_G); Q.rb = bcast(Q.vbg, BG_RHO_B); Q.rc = bcast(Q.vbg, BG_RHO_CDM); Q.ru = bcast(Q.vbg, BG_RHO_UR);
  const double inv_a = fast_rcp(bg_a);
  const double z = inv_a - 1.;
  const double zmax = Q.zmax;
  LK_MARK(13)   // background broadcasts
  if (z >= zmax) {  // analytic extrapolation, th.cpp:128-219
    const double x0 = Q.xe_last, inv_1pz = fast_rcp(1. + z);
    const double dk = (1. + z) * (1. + z) * P.n_e * x0 * SIGMA_T * MPC_OVER_M;
    const double r = (1. + z) / (1. + zmax);
    const double ddk = -bg_H * 2. * inv_1pz * dk;
    const double dddk = (bg_H * bg_H * inv_1pz - bg_Hp) * 2. * inv_1pz * dk;
    const double wb = K_B / (C_LIGHT * C_LIGHT * M_H) * (1. + (1. / NOT4 - 1.) * P.YHe + x0 * (1. - P.YHe)) * P.T_cmb * (1. + z);
    Q.kap = dk; Q.ddkappa = ddk; Q.cb2 = wb * 4. / 3.;
    const int c = opaque(lane);
    Q.vth = (c == TH_XE) ? x0 : (c == TH_DKAPPA) ? dk : (c == TH_TAU_D) ? Q.taud_last * r * r : (c == TH_DDKAPPA) ? ddk :
            (c == TH_DDDKAPPA) ? dddk : (c == TH_CB2) ? Q.cb2 : 0.;   // e^-kappa = g = g' = 0
    Q.th_inf = -1;
  } else {
    const int iz = window_find<TH_NCOL>(T.z_table, (const double2*)T.th, T.tt_size, z, lane, &Q.thx, Q.thw, &Q.th_base, WR - 10);
    if (iz != Q.th_inf) {
      Q.th_inf = iz;
      if (lane < TH_NCOL) {
        const double2* r = Q.thw + (iz - Q.th_base) * TH_NCOL + lane;
        Q.th_lo = r[0];
        Q.th_hi = r[TH_NCOL];
      }
    }
    const double x0 = bcast(Q.thx, iz - Q.th_base), x1 = bcast(Q.thx, iz - Q.th_base + 1);
    const double h = x1 - x0, b = (z - x0) * fast_rcp(h), a = 1. - b;
    Q.vth = spl2(Q.th_lo, Q.th_hi, a, b, h * h * (1.0 / 6.0));
    Q.kap = bcast(Q.vth, TH_DKAPPA); Q.ddkappa = bcast(Q.vth, TH_DDKAPPA); Q.cb2 = bcast(Q.vth, TH_CB2);
  }
  LK_MARK(14)   // thermodynamics: window, rows, spline, broadcasts
  // tau-only derived quantities (shared by every RHS evaluation at this tau); reciprocals by v_rcp_f64 + Newton
  Q.a2 = bg_a * bg_a;
  Q.aH = bg_a * bg_H;
  Q.two_over_aH = 2.0 * fast_rcp(Q.aH);
  Q.rg43 = 4. / 3. * Q.rg;
  Q.ru43 = 4. / 3. * Q.ru;
  Q.R = Q.rg43 * fast_rcp(Q.rb);
  Q.inv_1pR = fast_rcp(1.0 + Q.R);
  Q.inv_R = fast_rcp(Q.R);
  Q.tau_c = fast_rcp(Q.kap);                     // pm.cpp:9290-9297
  Q.dtau_c = -Q.ddkappa * Q.tau_c * Q.tau_c;
  Q.F = Q.tau_c * Q.inv_1pR;
  Q.Fp = Q.dtau_c * Q.inv_1pR + Q.tau_c * Q.aH * Q.R * Q.inv_1pR * Q.inv_1pR;
  Q.app = bg_Hp * bg_a + 2. * Q.aH * Q.aH;       // a''/a
  Q.inv_tau = fast_rcp(tau);
  if (!CURV) Q.kcot = Q.inv_tau;                       // k cotK_gen = 1/tau (pm.cpp:7969)
  else {                                                    // pm.cpp:7972-7977
    const double sq = sqrt(fabs(P.K));
    Q.kcot = (P.K < 0.) ? sq / tanh(sq * tau) : sq / tan(sq * tau);
  }
  LK_MARK(15)   // derived quantities
}

// ---- look-ups through the helper wave (see Mailbox) ----------------------------------------------------------------
// ask for the row at tau unless it is one of the last two that were asked for (the answers may still be on their way: mb_wait).
// Request n travels in slot n & 1 of the mailbox and so does its answer, which therefore stays where it is until request n + 2 is
// posted: the integrator posts the row of the NEXT step at the start of every step, when the previous step's row is dead.
static __device__ __forceinline__ void mb_request(Lookup& Q, double tau, int lane) {
  if (uni(tau == Q.rq_tau0) || uni(tau == Q.rq_tau1)) return;
  Q.my_req++;
  const int p = Q.my_req & 1;
  if (p) Q.rq_tau1 = tau; else Q.rq_tau0 = tau;
  if (lane == 0) Q.mb->req_tau[p] = tau;
  mb_store(&Q.mb->req_seq, Q.my_req);
}
// the slot (0 / 1) that holds the row at tau once the helper has answered - after asking for it if need be and waiting for the answer;
// -1 if the helper never answered (cannot happen unless the kernel is broken: the caller turns it into an error status instead of spinning for ever)
static __device__ __forceinline__ int mb_wait(Lookup& Q, double tau, int lane) {
  mb_request(Q, tau, lane);
  const int p = uni(tau == Q.rq_tau1) ? 1 : 0;
  const int want = Q.my_req - (((Q.my_req & 1) == p) ? 0 : 1);
  if (Q.ans_seen - want >= 0) return p;
  int spins = 0;
  while ((Q.ans_seen = mb_load(&Q.mb->ans_seq)) - want < 0) {
    __builtin_amdgcn_s_sleep(1);
    if (++spins > (1 << 24)) return -1;
  }
  return p;
}
static __device__ __forceinline__ bool mb_fetch(Lookup& Q, double tau, int lane) {
  if (uni(tau == Q.tau_cached)) return true;
  const int slot = mb_wait(Q, tau, lane);
  if (slot < 0) return false;
  const double* a = Q.mb->ans[slot];
  Q.rg = a[0]; Q.rb = a[1]; Q.rc = a[2]; Q.ru = a[3]; Q.kap = a[4]; Q.ddkappa = a[5]; Q.cb2 = a[6]; Q.a2 = a[7];
  Q.aH = a[8]; Q.two_over_aH = a[9]; Q.R = a[10]; Q.inv_1pR = a[11]; Q.inv_R = a[12]; Q.tau_c = a[13]; Q.dtau_c = a[14]; Q.F = a[15];
  Q.Fp = a[16]; Q.app = a[17]; Q.inv_tau = a[18]; Q.rg43 = a[19]; Q.ru43 = a[20]; Q.kcot = a[21];
  if (NCDM) {
#pragma unroll
    for (int i = 0; i < NCB_NCOL; i++) Q.ncv[i] = a[22 + i];
  }
  Q.tau_cached = tau;
  return true;
}
// The scalar integrator of the two-wave kernels does not copy its rows at all, and it does not find them by their time either: it
// knows which request is which.  mb_post asks and returns the request's number, mb_take waits for that number and points Q.row at the
// answer, which every RHS evaluation then reads from LDS (uniform ds_read_b128: eleven instructions behind one wait).
//  * 44 VGPRs less through the step loop: with them the loop needs more than the 256 architectural registers and the compiler parked
//    and restored ~150 dwords in AGPRs per step around the factorisation and the sampling block;
//  * no comparisons of times: a v_cmp_f64 feeding a scalar branch costs a lone wavefront ~45 cycles (tools/ubench.hip), and finding a
//    row by its time took four to six of them per step.
// Request n and its answer live in slot n & 1, so an answer stays intact until request n + 2 is posted.  The integrator posts the row
// of the NEXT step at the start of every step (when the previous step's row is dead) and everything else only after the rows it
// replaces are dead: at most two requests wait at any time, the helper answers them in order.
enum RowIdx : int { ROW_rg = 0, ROW_rb, ROW_rc, ROW_ru, ROW_kap, ROW_ddkappa, ROW_cb2, ROW_a2, ROW_aH, ROW_two_over_aH, ROW_R, ROW_inv_1pR, ROW_inv_R,
                    ROW_tau_c, ROW_dtau_c, ROW_F, ROW_Fp, ROW_app, ROW_inv_tau, ROW_rg43, ROW_ru43, ROW_kcot };
static __device__ __forceinline__ int mb_post(Lookup& Q, double tau, int lane) {
  // (an integrator uses either this pair or mb_request / mb_fetch, never both on one mailbox: the latter's record of what was asked
  //  for is not kept here)
  Q.my_req++;
  if (lane == 0) Q.mb->req_tau[Q.my_req & 1] = tau;
  mb_store(&Q.mb->req_seq, Q.my_req);
  return Q.my_req;
}
static __device__ __forceinline__ bool mb_take(Lookup& Q, int seq) {
  if (Q.ans_seen - seq < 0) {
    int spins = 0;
    while ((Q.ans_seen = mb_load(&Q.mb->ans_seq)) - seq < 0) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1 << 24)) return false;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");   // (the answer may have been seen by a relaxed mb_peek)
  Q.row = Q.mb->ans[seq & 1];
  return true;
}
// The register-set kernels (cpt_perturb_sets.inc) read their rows from LDS as well - they spill as it is, 62 VGPRs of row were what
// pushed the per-lane coefficients of the RHS into scratch memory (seven serialised scratch reloads per evaluation) - but still find
// them by time, with the helper (mb_fetch_row: valid while at most one request has been posted since the row was asked for, which is
// how ndf15v asks) or without (row_store: this wave's own look-up, written out once per new time).
static __device__ __forceinline__ bool mb_fetch_row(Lookup& Q, double tau, int lane) {
  if (uni(tau == Q.row_tau) && uni(tau == ((Q.row == Q.mb->ans[1]) ? Q.rq_tau1 : Q.rq_tau0))) return true;
  const int slot = mb_wait(Q, tau, lane);
  if (slot < 0) return false;
  Q.row = Q.mb->ans[slot];
  Q.row_tau = tau;
  return true;
}
static __device__ __forceinline__ void row_store(Lookup& Q, double* a, int lane) {
  if (lane == 0) {
    a[0] = Q.rg; a[1] = Q.rb; a[2] = Q.rc; a[3] = Q.ru; a[4] = Q.kap; a[5] = Q.ddkappa; a[6] = Q.cb2; a[7] = Q.a2;
    a[8] = Q.aH; a[9] = Q.two_over_aH; a[10] = Q.R; a[11] = Q.inv_1pR; a[12] = Q.inv_R; a[13] = Q.tau_c; a[14] = Q.dtau_c; a[15] = Q.F;
    a[16] = Q.Fp; a[17] = Q.app; a[18] = Q.inv_tau; a[19] = Q.rg43; a[20] = Q.ru43; a[21] = Q.kcot;
  }
  if (NCDM) { if (lane < NCB_NCOL) a[22 + lane] = Q.vnc; }   // lane c holds column c of the ncdm row
  Q.row = a;
  Q.row_tau = Q.tau_cached;
}
// (hand-over between schemes, Jacobian diagonal: the few places outside the RHS that read a row's fields)
static __device__ __forceinline__ void mb_row_to_regs(Lookup& Q) {
  const double* a = Q.row;
  if (NCDM) {
#pragma unroll
    for (int i = 0; i < NCB_NCOL; i++) Q.ncv[i] = a[22 + i];
  }
  Q.rg = a[0]; Q.rb = a[1]; Q.rc = a[2]; Q.ru = a[3]; Q.kap = a[4]; Q.ddkappa = a[5]; Q.cb2 = a[6]; Q.a2 = a[7];
  Q.aH = a[8]; Q.two_over_aH = a[9]; Q.R = a[10]; Q.inv_1pR = a[11]; Q.inv_R = a[12]; Q.tau_c = a[13]; Q.dtau_c = a[14]; Q.F = a[15];
  Q.Fp = a[16]; Q.app = a[17]; Q.inv_tau = a[18]; Q.rg43 = a[19]; Q.ru43 = a[20]; Q.kcot = a[21];
  if (Q.row_tau >= 0.) Q.tau_cached = Q.row_tau;   // (mb_take does not keep times: the scalar integrator never looks a row up by its time)
}

// ---- physics ------------------------------------------------------------------------------------
// Per-lane description of the current regime.  EVERY equation of the scalar system has the shape
//   dy = A y[dn] - B y[up] - (D kappa' + G k cotK(tau) + Gt/tau) y + Xmc h'/2 + Xms k^2 alpha + XP kappa' Pi/8... + X4 S4 + Xeta eta' + Xtb theta_b'
// with per-lane constants (A, B, D, G, X*, dn, up) fixed by the regime and a handful of wave-uniform scalars (the
// metric perturbations, the polarisation source, the baryon-photon coupling) that depend on (tau, y):
//   * the streaming terms A y[dn] - B y[up] couple neighbours of one multipole ladder (delta, theta, shear, l=3, ...);
//     dn / up are lane addresses, so a ladder may jump from its core part (lanes < nc) to its tail;
//   * the RHS is then two ds_bpermute, a dozen v_readlane, the wave-uniform Einstein / tight-coupling algebra and
//     nine fused multiply-adds: no lane-dependent branch at all.
// chain: 0 = core, 1 = photon temperature tail, 2 = polarisation tail, 3 = ur tail (l >= 3 elements).
struct LaneEq {
  int role, ell, chain;
  bool first, last;    // l == 3 / l == l_max of a tail
  int dn, up;          // byte address (lane * 4) of the lanes holding y_{l-1} / y_{l+1}
  int first_addr;      // core parents of a tail (shear_g, pol2, shear_ur): byte address of the tail's l=3 lane; else own lane
  int parent_addr;     // tail lanes: byte address of the core parent; else own lane
  double Bpar;         // core parents of a present tail: B (their coupling to the tail's l=3 element); else 0
  double A, B, D, G, Gt;   // G multiplies k cotK_gen(tau) (hierarchy truncation), Gt multiplies 1/tau (ur fluid): equal in flat space
  double Xmc, Xms, XP, X4, Xeta, Xtb, Xeu;   // Xeu multiplies metric_euler = k^2 psi (Newtonian gauge; 0 in synchronous)
  int rem;             // (long tails) 1, 2, 3: this lane is the parent of the photon / polarisation / ur tail, whose l = 3 element lives on a chain wave
  unsigned pmask;      // (wave-uniform) bit i: core variable i is evolved in this scheme.  An idle core lane is an identity row AND column
                       // of the Newton matrix: the factorisation and the substitutions skip its pivot altogether.
  const double2* cw;   // LDS [6][64] or null: the twelve coefficients again (lane_eq_store), for an RHS that fetches them per evaluation
                       // (rhs<LK, true>) instead of holding 24 VGPRs through the step loop
};
static constexpr int CW_PAIRS = 6;
static __device__ __forceinline__ void lane_eq_store(LaneEq& e, double2* cw, int lane) {
  cw[0 * 64 + lane] = make_double2(e.A, e.B); cw[1 * 64 + lane] = make_double2(e.D, e.G); cw[2 * 64 + lane] = make_double2(e.Xmc, e.Xms);
  cw[3 * 64 + lane] = make_double2(e.XP, e.X4); cw[4 * 64 + lane] = make_double2(e.Xeta, e.Xtb); cw[5 * 64 + lane] = make_double2(e.Gt, e.Xeu);
  e.cw = cw;
}

static __device__ __forceinline__ LaneEq make_lane_eq(const PtParams& P, const Layout& L, int lane, double k) {
  LaneEq e;
  e.cw = nullptr;
  role_of(P, L, lane, &e.role, &e.ell);
  e.A = e.B = e.D = e.G = e.Gt = 0.;
  e.Xmc = e.Xms = e.XP = e.X4 = e.Xeta = e.Xtb = e.Xeu = 0.;
  e.chain = 0; e.first = false; e.last = false;
  int dn = lane, up = lane;
  const int l = e.ell;
  const double k2 = k * k, c3 = P.three_ceff2_ur, v3 = P.three_cvis2_ur;
  // curvature factors s_l = sqrt(1 - K (l^2-1)/k^2) of the multipole ladders (pm.cpp:2530-2533); all 1 in flat space
  auto S = [&](int ll) { return CURV ? sqrt(fmax(1.0 - P.K * (ll * ll - 1.0) / k2, 0.)) : 1.0; };
  const double s2 = S(2), s3 = S(3), s2sq = CURV ? 1. - 3. * P.K / k2 : 1.;
  int lm = 0, parent = lane;
  if (MODE) {  // ---- tensor modes, pm.cpp:9045-9215: one generic ladder element per lane ----
    const int role = e.role;
    const bool photon = (role == R_DELTA_G || role == R_THETA_G || role == R_SHEAR_G || role == R_LG), pol = (role == R_POL),
               ur = (role == R_DELTA_UR || role == R_THETA_UR || role == R_SHEAR_UR || role == R_LUR);
    if (photon || pol || ur) {
      lm = photon ? L.lmg : pol ? L.lmp : L.lmu;
      const int base = photon ? TL_DG : pol ? TL_P0 : TL_DUR, tail0 = photon ? L.g3 : pol ? L.q3 : L.u3;
      auto lane_of = [&](int ll) { return ll <= 4 ? base + ll : tail0 + (ll - 5); };
      e.D = ur ? 0. : 1.;
      if (l >= 1) dn = lane_of(l - 1);
      if (l < lm) up = lane_of(l + 1);
      if (l >= 5) { e.chain = photon ? 1 : pol ? 2 : 3; e.first = (l == 5); e.last = (l == lm); parent = lane_of(4); }
      if (l == lm) { e.A = k * S(l); e.G = 1. + l; }
      else if (l == 0) { e.B = pol ? k : 4. / 3.; }
      else if (pol) { e.A = k * l * S(l) / (2. * l + 1.); e.B = k * (l + 1.) * S(l + 1) / (2. * l + 1.); }
      else if (l == 1) { e.A = 0.25 * k2; e.B = k2 * (ur ? s2sq : s2); }                    // theta = (3k/4) F_1
      else if (l == 2) { e.A = 4. / 15. * (ur ? 1. : s2); e.B = 0.3 * k * (ur ? s3 / s2 : s3); }   // shear = F_2/2
      else if (l == 3) { e.A = 6. * k * s3 * (ur ? s2 : 1.) / 7.; e.B = 4. * k * S(4) / 7.; }
      else { e.A = k * l * S(l) / (2. * l + 1.); e.B = k * (l + 1.) * S(l + 1) / (2. * l + 1.); }
      if (l == 0) { e.Xmc = pol ? 0. : 1.; e.XP = photon ? -1. : pol ? 1. : 0.; }           // + sqrt6 gw' ;  -/+ kappa' sqrt6 P2
    } else if (role == R_GW) { e.A = 1.; dn = TL_GWD; }
    else if (role == R_GWDOT) e.Xtb = 1.;
    e.rem = 0;
    e.dn = dn * 4; e.up = up * 4; e.parent_addr = parent * 4;
    const bool is_parent = (lane < NC) && (up >= NC);
    e.first_addr = (is_parent ? up : lane) * 4;
    e.Bpar = is_parent ? e.B : 0.;
    e.pmask = present_mask(P, L);
    return e;
  }
  if (e.role == R_LG) { e.chain = 1; lm = L.lmg; e.D = 1.; parent = LN_SG; }
  else if (e.role == R_POL && l >= 3) { e.chain = 2; lm = L.lmp; e.D = 1.; parent = LN_P2; }
  else if (e.role == R_LUR) { e.chain = 3; lm = L.lmu; parent = LN_SUR; }
  if (e.chain) {
    e.first = (l == LFIRST); e.last = (l == lm);
    dn = e.first ? parent : lane - 1;
    up = e.last ? lane : lane + 1;
    if (l == 3 && e.chain != 2) { e.A = 6. * k * s3 * s2 / 7.; e.B = 4. * k * S(4) / 7.; }   // pm.cpp:8158-8161: F_2 = 2 s_2 shear
    else if (l < lm) { e.A = k * l * S(l) / (2. * l + 1.); e.B = k * (l + 1.) * S(l + 1) / (2. * l + 1.); }
    else { e.A = k * S(l); e.G = 1. + l; }                                     // pm.cpp:8171-8176: -(l+1) k cotKgen y_l
  } else {
    switch (e.role) {
      case R_DELTA_G: e.B = 4. / 3.; up = LN_TG; e.Xmc = -4. / 3.; break;                                 // pm.cpp:8095
      case R_THETA_G:
        if (!L.tca) { e.A = 0.25 * k2; dn = LN_DG; e.B = k2 * s2sq; up = LN_SG; e.D = 1.; e.Xeu = 1.; } // pm.cpp:8145-8148
        e.X4 = 1.;                           // S4 = kappa' theta_b, or the whole tight-coupling expression (pm.cpp:8214-8217)
        break;
      case R_SHEAR_G:                                                                                    // pm.cpp:8151-8155
        e.A = 4. / 15.; dn = LN_TG; e.D = 1.; e.Xms = 4. / 15.; e.XP = 0.4 / s2;
        if (L.gN > 0) { e.B = 0.3 * k * s3 / s2; up = L.g3; }
        break;
      case R_POL:
        if (l == 0) { e.B = k; up = LN_P1; e.D = 1.; e.XP = 4.; }                                        // pm.cpp:8179-8181
        else if (l == 1) { e.A = k / 3.; dn = LN_P0; e.B = 2. * k * s2 / 3.; up = LN_P2; e.D = 1.; }     // pm.cpp:8184-8186
        else { e.A = 2. * k * s2 / 5.; dn = LN_P1; e.D = 1.; e.XP = 0.8; if (L.qN > 0) { e.B = 3. * k * s3 / 5.; up = L.q3; } }  // :8189-8191
        break;
      case R_DELTA_B: e.B = 1.; up = LN_TB; e.Xmc = -1.; break;                                          // pm.cpp:8101
      case R_THETA_B: e.Xtb = 1.; break;
      case R_DELTA_CDM: e.Xmc = -1.; if (GAUGE == CPT_GAUGE_NEWTONIAN) { e.B = 1.; up = LN_TC; } break;  // pm.cpp:8232-8240
      case R_THETA_CDM: e.Xeu = 1.; break;                                     // pm.cpp:8235 (-a'/a theta_cdm added in rhs)
      case R_DELTA_UR: e.B = 4. / 3.; up = LN_TUR; e.Xmc = -4. / 3.; break;                              // pm.cpp:8630-8634
      case R_THETA_UR: e.A = 0.25 * c3 * k2; dn = LN_DUR; e.B = k2 * s2sq; up = LN_SUR; e.Xeu = 1.; break;  // pm.cpp:8637-8641
      case R_SHEAR_UR:
        dn = LN_TUR;
        if (!L.ufa) { e.A = 4. / 15. * v3; e.Xms = 4. / 15. * v3; if (L.uN > 0) { e.B = 0.3 * k * s3 / s2; up = L.u3; } }  // pm.cpp:8645-8651
        else {                                                                                           // pm.cpp:8704-8708
          e.A = 2. / 3.;
          // ufa_class source: h'/2 = metric_continuity (synchronous), -6 phi' = 2 metric_continuity (Newtonian), pm.cpp:8060-8073
          if (P.ufa_method == CPT_UFA_CLASS) { (CURV ? e.Gt : e.G) = 3.; e.Xmc = (GAUGE == CPT_GAUGE_NEWTONIAN) ? 4. / 3. : 2. / 3.; }
          else if (P.ufa_method == CPT_UFA_MB) { (CURV ? e.Gt : e.G) = 3.; e.Xms = 2. / 3.; }
          else e.Xms = 2. / 3.;                                                 // ufa_hu: -3 a'/a shear added in rhs
        }
        break;
      case R_ETA: e.Xeta = 1.; break;                                                                    // pm.cpp:8896
      default: break;
    }
  }
  e.rem = 0;
  if (NCDM && L.lng && up >= 64) { e.rem = (lane == LN_SG) ? 1 : (lane == LN_P2) ? 2 : 3; up = lane; }
  e.dn = dn * 4;
  e.up = up * 4;
  e.parent_addr = parent * 4;
  const bool is_parent = (lane < NC) && (up >= NC);   // its ladder continues in a tail
  e.first_addr = (is_parent ? up : lane) * 4;
  e.Bpar = is_parent ? e.B : 0.;
  e.pmask = present_mask(P, L);
  return e;
}

// metric + fluid summary left behind by the last RHS evaluation (struct perturb_workspace of the reference)
struct Metric {
  double hp, etap, alpha, alphap;
  double psi, phip;   // Newtonian gauge
  double rsa_dg, rsa_tg;
  double tca_shear_g;
};

// (NCDM) what the non-cold species contribute to the Einstein equations in this RHS evaluation: delta rho, (rho+p) theta,
// (rho+p) sigma summed over species (pm.cpp:6317-6432), formed by the caller from the momentum-bin sets or from the fluids.
struct NcIn {
  double D, T, S; double y3[3];   // y3: (long tails) the l = 3 elements of the photon / polarisation / ur tails
  double Dn[CPT_MAX_NCDM], Tn[CPT_MAX_NCDM];   // delta rho and (rho + p) theta of every species on its own: only formed where a sample of the
                                               // density / velocity transfer sources is about to be stored (store_sources), dead in the step loop
};

// y of another lane (per-lane byte address): two ds_bpermute_b32, executed by every lane
static __device__ __forceinline__ double gather(double v, int addr) {
  const int lo = __builtin_amdgcn_ds_bpermute(addr, __double2loint(v));
  const int hi = __builtin_amdgcn_ds_bpermute(addr, __double2hiint(v));
  return pin(__hiloint2double(hi, lo));
}

// tensor modes: gw_source (pm.cpp:6616-6660), the Einstein equation for gw'' (:6036-6040) and perturb_derivs :9045-9215
// (LK as in rhs below: 0 own look-up, 1 from the helper into Q, 3 the row lies in LDS at Q.row)
template <int LK>
static __device__ __forceinline__ double rhs_tensor(const PtParams& P, const Layout& L, const LaneEq& e, Lookup& Q, Metric& M, double k,
                                                    double tau, double y, int lane) {
  if (LK == 1) { if (!mb_fetch(Q, tau, lane)) return nan(""); }
  else if (LK == 0) lookup(P, Q, tau, lane);
#define QT(f) ((LK == 3) ? Q.row[ROW_##f] : Q.f)
  const double q_a2 = QT(a2), q_rg = QT(rg), q_ru = QT(ru), q_aH = QT(aH), q_kap = QT(kap), q_kcot = QT(kcot);
#undef QT
  const double ym = gather(y, e.dn), yp = gather(y, e.up);
  const double SQRT6 = 2.449489742783178;
  // (lanes the scheme does not evolve read as 0)
  const double dg = bcast(y, TL_DG), sg = bcast(y, TL_SG), g4 = bcast(y, TL_G4), p0 = bcast(y, TL_P0), p2 = bcast(y, TL_P2), p4 = bcast(y, TL_P4);
  const double dur = bcast(y, TL_DUR), sur = bcast(y, TL_SUR), u4 = bcast(y, TL_U4), gw = bcast(y, TL_GW), gwd = bcast(y, TL_GWD);
  const double P2 = -1.0 / SQRT6 * (0.1 * dg + 2. / 7. * sg + 3. / 70. * g4 - 0.6 * p0 + 6. / 7. * p2 - 3. / 70. * p4);
  double gw_source = -SQRT6 * 4. * q_a2 * q_rg * (1. / 15. * dg + 4. / 21. * sg + 1. / 35. * g4);
  if (P.evolve_tensor_ur) gw_source += -SQRT6 * 4. * q_a2 * q_ru * (1. / 15. * dur + 4. / 21. * sur + 1. / 35. * u4);
  const double gwpp = -2. * q_aH * gwd - (k * k + (CURV ? 2. * P.K : 0.)) * gw + gw_source;
  double dy = e.A * ym - e.B * yp - (e.D * q_kap + e.G * q_kcot) * y;
  dy = fma(e.XP, q_kap * SQRT6 * P2, dy);
  dy = fma(e.Xmc, SQRT6 * gwd, dy);
  dy = fma(e.Xtb, gwpp, dy);
  return dy;
}

// perturb_sources for tensor modes (pm.cpp:7243-7280)
static __device__ __forceinline__ void store_sources_tensor(const PtParams& P, const Layout& L, const Lookup& Q, double y, int it, int ik, int lane) {
  const double g = bcast(Q.vth, TH_G), expmk = bcast(Q.vth, TH_EXPMK), gwd = bcast(y, TL_GWD);
  double Pi = 0.;
  if (!L.rsa) {
    if (!L.tca)
      Pi = -(0.1 * bcast(y, TL_DG) + 2. / 7. * bcast(y, TL_SG) + 3. / 70. * bcast(y, TL_G4) - 0.6 * bcast(y, TL_P0) + 6. / 7. * bcast(y, TL_P2) -
             3. / 70. * bcast(y, TL_P4)) / 2.449489742783178;
    else Pi = 0.4 * 2.449489742783178 * gwd / Q.kap;
  }
  if (lane == 0) {
    const size_t base = (size_t)ik * P.ntau + it, tstride = (size_t)P.nk * P.ntau;
    if (P.tp_t2 >= 0) P.src[P.tp_t2 * tstride + base] = -gwd * expmk + g * Pi;
    if (P.tp_p >= 0) P.src[P.tp_p * tstride + base] = 2.449489742783178 * g * Pi;
  }
}

// perturb_derivs (pm.cpp:7861-9218) with perturb_total_stress_energy + perturb_einstein (pm.cpp:6047-6703, 5840-6045),
// perturb_rsa_delta_and_theta (pm.cpp:9530-9636) and perturb_tca_slip_and_shear (pm.cpp:9229-9516) folded in;
// synchronous gauge, K = 0.  y: this lane's component (named components are broadcast with v_readlane).
// Returns dy of this lane and leaves M describing the state (tau, y).
// LK: where the row of the tables comes from - 0 this wave's own look-up, 1 the helper wave (copied to Q), 2 it is in Q already,
//     3 it lies in LDS where the helper wave left it (Q.row, see mb_take)
// ECO: the per-lane coefficients are fetched from LDS (LaneEq::cw) instead of being read from e's registers: the register-set kernels,
//      which spill, otherwise reload them one by one from scratch memory inside every evaluation
template <int LK = 0, bool ECO = false>
static __device__ __forceinline__ double rhs(const PtParams& P, const Layout& L, const LaneEq& e, Lookup& Q, Metric& M, double k,
                                      double inv_k2, double tau, double y, int lane, NcIn* Np = nullptr) {
  if (MODE) return rhs_tensor<LK>(P, L, e, Q, M, k, tau, y, lane);
#ifdef CPT_PROFILE
  unsigned long long* prof = Q.prof;
  PROF_DECL;
  PROF_START();
#endif
  // (a helper that never answers - impossible unless the kernel is broken - poisons the result: the step then fails its norm
  //  tests, shrinks to the minimal step and the mode ends with "step size too small" instead of spinning for ever)
  if (LK == 1) { if (!mb_fetch(Q, tau, lane)) return nan(""); }
  else if (LK == 0) lookup(P, Q, tau, lane);
  // a field of the row at tau: from LDS where the helper left it (LK = 3, see mb_take: eleven uniform b128 reads issued together
  // here, behind one wait) or from this wave's registers
  double rw[22];
  if (LK == 3) {
    const double2* r2 = reinterpret_cast<const double2*>(Q.row);
#pragma unroll
    for (int i = 0; i < 11; i++) { const double2 v = r2[i]; rw[2 * i] = v.x; rw[2 * i + 1] = v.y; }
  }
#define QV(f) ((LK == 3) ? rw[ROW_##f] : Q.f)
#ifdef CPT_PROFILE
  PROF_STOP(8); PROF_START();
#endif
  // neighbours on the multipole ladders: issued first, their LDS-crossbar latency hides behind the scalar algebra
  const double ym = gather(y, e.dn);
  const double yp = gather(y, e.up);
  const double a2 = QV(a2), aH = QV(aH), k2 = k * k, R = QV(R), kap = QV(kap);
  // ---- named components ----
  // (a variable the scheme does not evolve reads as 0 from its idle lane)
  double dg = bcast(y, LN_DG), tg = bcast(y, LN_TG);
  const double sg = bcast(y, LN_SG), p0 = bcast(y, LN_P0), p2 = bcast(y, LN_P2);
  const double dur = bcast(y, LN_DUR), tur = bcast(y, LN_TUR), sur = bcast(y, LN_SUR);
  const double db = bcast(y, LN_DB), tb = bcast(y, LN_TB), eta = bcast(y, LN_ETA), dc = bcast(y, LN_DC);
  const double cb2 = QV(cb2);
#ifdef CPT_PROFILE
  PROF_STOP(9); PROF_START();
#endif
  // ---- stress-energy sums ----
  double delta_rho = QV(rg) * dg + QV(rb) * db;
  double rpt = QV(rg43) * tg + QV(rb) * tb;
  double rps = QV(rg43) * sg;
  // (a species that is absent has zero density in the tables and an idle lane: no test - a scalar branch costs a lone wavefront five
  //  multiply-adds)
  delta_rho += QV(rc) * dc;
  delta_rho += QV(ru) * dur; rpt += QV(ru43) * tur; rps += QV(ru43) * sur;
  if (NCDM) {
    const NcIn& N = *Np;
    delta_rho += N.D; rpt += N.T; rps += N.S;
  }
  // ---- Einstein equations -> the metric terms of the matter equations (pm.cpp:8049-8074):
  //      mc = metric_continuity, me = metric_euler, ms = metric_shear, msp = its derivative, mdot = eta' or phi'
  double mc, me, ms, msp, mdot;
  if (GAUGE == CPT_GAUGE_SYNCHRONOUS) {
    const double hp = (Q.k2s2 * eta + 1.5 * a2 * delta_rho) * QV(two_over_aH);          // pm.cpp:5913-5914, k2s2 = k^2 (1 - 3K/k^2)
    if (L.rsa) {
      double rdg = 0., rtg = 0., rdur = 0., rtur = 0.;
      if (P.rsa_method != CPT_RSA_NULL) { rdg = 4. * inv_k2 * (aH * hp - k2 * eta); rtg = -0.5 * hp; }
      if (P.rsa_method == CPT_RSA_MD_WITH_REIO) {
        rdg += -4. * inv_k2 * kap * (tb + 0.5 * hp);
        rtg += 3. * inv_k2 * (QV(ddkappa) * (tb + 0.5 * hp) + kap * (-aH * tb + cb2 * k2 * db - aH * hp + k2 * eta));
      }
      if (P.rsa_method != CPT_RSA_NULL) { rdur = 4. * inv_k2 * (aH * hp - k2 * eta); rtur = -0.5 * hp; }
      delta_rho += QV(rg) * rdg;
      rpt += QV(rg43) * rtg;
      delta_rho += QV(ru) * rdur; rpt += QV(ru43) * rtur;
      M.rsa_dg = rdg; M.rsa_tg = rtg;
      dg = rdg; tg = rtg;  // pm.cpp:8085-8088: the equations below use the streaming values
    }
    const double etap = (1.5 * a2 * rpt + (CURV ? 0.5 * P.K * hp : 0.)) * Q.inv_k2s2;                 // pm.cpp:5938
    const double alpha = (hp + 6. * etap) * 0.5 * inv_k2;
    if (L.tca) rps += QV(rg43) * (16. / 45. * QV(tau_c) * (tg + k2 * alpha));
    const double alphap = -2. * aH * alpha + eta - 4.5 * (a2 * inv_k2) * rps;
    M.hp = hp; M.etap = etap; M.alpha = alpha; M.alphap = alphap;
    mc = 0.5 * hp; me = 0.; ms = k2 * alpha; msp = k2 * alphap; mdot = etap;
  } else {
    // Newtonian gauge (pm.cpp:5869-5897): the LN_ETA lane holds phi; cdm has a velocity
    const double tc = bcast(y, LN_TC);
    rpt += QV(rc) * tc;
    if (L.tca) rps += QV(rg43) * (16. / 45. * QV(tau_c) * tg);                            // pm.cpp:6134-6136
    const double psi = eta - 4.5 * (a2 * inv_k2) * rps;
    const double phip = -aH * psi + 1.5 * (a2 * inv_k2) * rpt;
    if (L.rsa) {                                                                      // pm.cpp:9549-9592
      double rdg = 0., rtg = 0.;
      if (P.rsa_method != CPT_RSA_NULL) { rdg = -4. * eta; rtg = 6. * phip; }
      if (P.rsa_method == CPT_RSA_MD_WITH_REIO) {
        rdg += -4. * inv_k2 * kap * tb;
        rtg += 3. * inv_k2 * (QV(ddkappa) * tb + kap * (-aH * tb + cb2 * k2 * db + k2 * eta));
      }
      M.rsa_dg = rdg; M.rsa_tg = rtg;
      dg = rdg; tg = rtg;
    }
    M.psi = psi; M.phip = phip;
    mc = -3. * phip; me = k2 * psi; ms = 0.; msp = 0.; mdot = phip;
  }
  // ---- baryon velocity / tight coupling ----
  double dtb, S4;
  if (!L.tca) {
    dtb = -aH * tb + me + k2 * cb2 * db + R * kap * (tg - tb);  // pm.cpp:8108-8113
    S4 = kap * tb;
  } else {
    const double tau_c = QV(tau_c), dtau_c = QV(dtau_c), F = QV(F);
    // first order: dkappa ~ a^-2 assumed (Ma & Bertschinger, pm.cpp:9351-9361) or not (CAMB form, :9364-9373)
    const double slip_c = (P.tca_method == CPT_TCA_FIRST_ORDER_MB) ? 2. * R * QV(inv_1pR) * aH : dtau_c * kap - 2. * aH * QV(inv_1pR);
    double slip = slip_c * (tb - tg) +
                  F * (-QV(app) * tb + k2 * (-aH * dg * 0.5 + cb2 * (-tb - mc) - (-tg - mc) * (1. / 3.)) - aH * me);
    double shear = 16. / 45. * tau_c * (tg + ms);
    const double theta_prime = (-aH * tb + k2 * (cb2 * db + R * 0.25 * dg)) * QV(inv_1pR) + me;
    const double shear_prime = 16. / 45. * (tau_c * (theta_prime + msp) + dtau_c * (tg + ms));
    if (P.tca_method == CPT_TCA_COMPROMISE_CLASS) {
      slip = (1. - 2. * aH * F) * slip + F * k2 * (Q.s2sq * (2. * aH * shear + shear_prime) - (1. / 3. - cb2) * (F * theta_prime + 2. * QV(Fp) * tb));  // pm.cpp:9501
      shear = (1. - 11. / 6. * dtau_c) * shear - (11. / 6. * 16. / 45.) * tau_c * tau_c * (theta_prime + msp);
    }
    M.tca_shear_g = shear;
    dtb = (-aH * tb + k2 * (cb2 * db + R * (dg * 0.25 - Q.s2sq * shear)) + R * slip) * QV(inv_1pR) + me;  // pm.cpp:8123-8129
    S4 = -(dtb + aH * tb - k2 * cb2 * db) * QV(inv_R) + k2 * (0.25 * dg - Q.s2sq * shear) + (1. + R) * QV(inv_R) * me;  // pm.cpp:8214-8222
  }
  const double SP = kap * (p0 + p2 + 2. * Q.s2 * sg) * 0.125;  // kappa' Pi,  Pi = (G_gamma0 + G_gamma2 + F_gamma2)/8 (pm.cpp:8142)
#ifdef CPT_PROFILE
  PROF_STOP(10); PROF_START();
#endif
  // ---- every equation: streaming + damping + sources ----
  double yup = yp;
  if (NCDM) { if (LONG) { const int rem = opaque(e.rem); yup = (rem == 1) ? Np->y3[0] : (rem == 2) ? Np->y3[1] : (rem == 3) ? Np->y3[2] : yp; } }
  double eA = e.A, eB = e.B, eD = e.D, eG = e.G, eGt = e.Gt, eXmc = e.Xmc, eXms = e.Xms, eXP = e.XP, eX4 = e.X4, eXeta = e.Xeta, eXtb = e.Xtb, eXeu = e.Xeu;
  if (ECO) {
    const double2 c0 = e.cw[0 * 64 + lane], c1 = e.cw[1 * 64 + lane], c2 = e.cw[2 * 64 + lane], c3 = e.cw[3 * 64 + lane], c4 = e.cw[4 * 64 + lane];
    eA = c0.x; eB = c0.y; eD = c1.x; eG = c1.y; eXmc = c2.x; eXms = c2.y; eXP = c3.x; eX4 = c3.y; eXeta = c4.x; eXtb = c4.y;
    if (CURV || GAUGE == CPT_GAUGE_NEWTONIAN) { const double2 c5 = e.cw[5 * 64 + lane]; eGt = c5.x; eXeu = c5.y; }
  }
  double dy = eA * ym - eB * yup - (eD * kap + eG * QV(kcot) + (CURV ? eGt * QV(inv_tau) : 0.)) * y;
  dy = fma(eXmc, mc, dy);
  dy = fma(eXms, ms, dy);
  dy = fma(eXP, SP, dy);
  dy = fma(eX4, S4, dy);
  dy = fma(eXeta, mdot, dy);
  dy = fma(eXtb, dtb, dy);
  if (GAUGE == CPT_GAUGE_NEWTONIAN) {
    dy = fma(eXeu, me, dy);
    if (opaque(e.role) == R_THETA_CDM) dy -= aH * y;                                  // pm.cpp:8235
  }
  // rarely used variants: non-standard ur sound speed (pm.cpp:8630-8641), ufa_hu (pm.cpp:8711-8716)
  if (L.ur_special) {   // (Layout: three_ceff2_ur != 1 or ufa_hu)
    const double c3 = P.three_ceff2_ur;
    const int role = opaque(e.role);
    if (role == R_DELTA_UR) dy += (1. - c3) * aH * (dur + 4. * aH * tur * inv_k2);
    if (role == R_THETA_UR) dy -= (1. - c3) * aH * tur;
    if (role == R_SHEAR_UR && L.ufa && P.ufa_method == CPT_UFA_HU) dy -= 3. * aH * sur;
  }
#ifdef CPT_PROFILE
  PROF_STOP(11);
#endif
#undef QV
  return dy;
}

// perturb_sources (pm.cpp:6731-7285): the RHS has just been evaluated at (tau, y) => Q and M describe the sample.
// dy is the dense-output derivative (only theta_b' is used, pm.cpp:6883). Lane 0 stores the tp_size values.
static __device__ __forceinline__ void store_sources(const PtParams& P, const Layout& L, const Lookup& Q, const Metric& M, double k,
                                              double inv_k2, double y, double dy, double tca_shear_prev, int it, int ik, int lane,
                                              const NcIn& N = NcIn{0., 0., 0., {0., 0., 0.}, {0., 0., 0.}, {0., 0., 0.}}) {
  if (MODE) { store_sources_tensor(P, L, Q, y, it, ik, lane); return; }
  struct { double g, dg, expmk; } th;
  th.g = bcast(Q.vth, TH_G); th.dg = bcast(Q.vth, TH_DG); th.expmk = bcast(Q.vth, TH_EXPMK);
  const double bg_a = bcast(Q.vbg, BG_A);
  const double z = P.a_today / bg_a - 1.;
  const double aH = Q.aH, aHp = bcast(Q.vbg, BG_HP) * bg_a + aH * aH;   // (a'/a)'
  double delta_g, Pi;
  if (L.rsa) { delta_g = M.rsa_dg; Pi = 0.; }
  else {
    delta_g = bcast(y, LN_DG);
    if (L.tca) Pi = 5. * Q.s2 * tca_shear_prev / 8.;  // left over from the last derivs call of the evolver (pm.cpp:6810)
    else Pi = (bcast(y, LN_P0) + bcast(y, LN_P2) + 2. * Q.s2 * bcast(y, LN_SG)) / 8.;
  }
  const double eta = bcast(y, LN_ETA), tb = bcast(y, LN_TB), dtb = bcast(dy, LN_TB);
  double delta_m = 0., delta_cb = 0.;
  if (P.tp_dm >= 0) {  // gauge-invariant matter density contrast, pm.cpp:6573, 5979-5981
    double drm = Q.rb * bcast(y, LN_DB), rho_m = Q.rb;
    double rptm = Q.rb * tb;                       // [(rho+p) theta]_matter: cdm contributes in the Newtonian gauge (pm.cpp:6241-6243)
    if (P.has_cdm) {
      drm += Q.rc * bcast(y, LN_DC); rho_m += Q.rc;
      if (GAUGE == CPT_GAUGE_NEWTONIAN) rptm += Q.rc * bcast(y, LN_TC);
    }
    delta_m = (drm + 3. * aH * rptm * inv_k2) / rho_m;
    if (NCDM) {   // pm.cpp:6309-6315, 6414-6426, 5979-5993: cdm+baryons alone, then with the non-cold species
      delta_cb = delta_m;
      double rho_nc = 0., p_nc = 0.;
      for (int n = 0; n < P.nc.n_species; n++) { rho_nc += bcast(Q.vnc, 3 * n); p_nc += bcast(Q.vnc, 3 * n + 1); }
      delta_m = (drm + N.D) / (rho_m + rho_nc) + 3. * aH * inv_k2 * (rptm + N.T) / (rho_m + rho_nc + p_nc);
    }
  }
  int switch_isw = 1;
  if ((P.switch_eisw == 0) && (z >= P.eisw_lisw_split_z)) switch_isw = 0;
  if ((P.switch_lisw == 0) && (z < P.eisw_lisw_split_z)) switch_isw = 0;
  if (lane == 0 && GAUGE == CPT_GAUGE_NEWTONIAN) {  // pm.cpp:6849-6860, 6955-6957 (eta = phi here)
    const size_t base = (size_t)ik * P.ntau + it, tstride = (size_t)P.nk * P.ntau;
    if (P.tp_t0 >= 0)
      P.src[P.tp_t0 * tstride + base] = P.switch_sw * th.g * (delta_g / 4. + M.psi) +
                                        switch_isw * (th.g * (eta - M.psi) + th.expmk * 2. * M.phip) +
                                        P.switch_dop * inv_k2 * (th.g * dtb + th.dg * tb);
    if (P.tp_t1 >= 0) P.src[P.tp_t1 * tstride + base] = switch_isw * th.expmk * k * (M.psi - eta);
    if (P.tp_t2 >= 0) P.src[P.tp_t2 * tstride + base] = P.switch_pol * th.g * Pi;
    if (P.tp_p >= 0) P.src[P.tp_p * tstride + base] = sqrt(6.) * th.g * Pi;
    if (P.tp_pp >= 0) P.src[P.tp_pp * tstride + base] = eta + M.psi;
    if (P.tp_dm >= 0) P.src[P.tp_dm * tstride + base] = delta_m;
  } else if (lane == 0) {
    const size_t base = (size_t)ik * P.ntau + it, tstride = (size_t)P.nk * P.ntau;
    if (P.tp_t0 >= 0)
      P.src[P.tp_t0 * tstride + base] =
          P.switch_sw * th.g * (delta_g / 4. + M.alphap) +
          switch_isw * (th.g * (eta - M.alphap - 2 * aH * M.alpha) + th.expmk * 2. * (M.etap - aHp * M.alpha - aH * M.alphap)) +
          P.switch_dop * (th.g * (dtb * inv_k2 + M.alphap) + th.dg * (tb * inv_k2 + M.alpha));
    if (P.tp_t1 >= 0) P.src[P.tp_t1 * tstride + base] = switch_isw * th.expmk * k * (M.alphap + 2. * aH * M.alpha - eta);
    if (P.tp_t2 >= 0) P.src[P.tp_t2 * tstride + base] = P.switch_pol * th.g * Pi;
    if (P.tp_p >= 0) P.src[P.tp_p * tstride + base] = sqrt(6.) * th.g * Pi;
    if (P.tp_pp >= 0) P.src[P.tp_pp * tstride + base] = eta + M.alphap;
    if (P.tp_dm >= 0) P.src[P.tp_dm * tstride + base] = delta_m;
    if (NCDM && P.tp_dcb >= 0) P.src[P.tp_dcb * tstride + base] = delta_cb;   // pm.cpp:7001-7003
  }
  // ---- density and velocity transfer functions (output = mTk, vTk; pm.cpp:6930-6975, 7017-7200; no N-body gauge shifts) ----
  if constexpr (GAUGE == CPT_GAUGE_SYNCHRONOUS || NCDM == 0) {
    bool any = (P.tp_dn >= 0) || (P.tp_tn >= 0);
#pragma unroll
    for (int i = 0; i < CPT_NTK; i++) any = any || (P.tp_tk[i] >= 0);
    if (__builtin_expect(any, 0)) {   // (cold unless output = mTk / vTk: kept out of the way of the lone-wave kernels, which sample in place)
      const double k2 = k * k;
      double dgam = delta_g, tgam, dur, tur;
      if (L.rsa) {      // the streaming values (pm.cpp:9530-9636), as the RHS formed them
        tgam = M.rsa_tg;
        if (GAUGE == CPT_GAUGE_SYNCHRONOUS) { dur = (P.rsa_method != CPT_RSA_NULL) ? 4. * inv_k2 * (aH * M.hp - k2 * eta) : 0.; tur = (P.rsa_method != CPT_RSA_NULL) ? -0.5 * M.hp : 0.; }
        else { dur = (P.rsa_method != CPT_RSA_NULL) ? -4. * eta : 0.; tur = (P.rsa_method != CPT_RSA_NULL) ? 6. * M.phip : 0.; }
      } else { tgam = bcast(y, LN_TG); dur = bcast(y, LN_DUR); tur = bcast(y, LN_TUR); }
      const double db = bcast(y, LN_DB), dc = bcast(y, LN_DC), tc = (GAUGE == CPT_GAUGE_NEWTONIAN) ? bcast(y, LN_TC) : 0.;
      // totals: every species but the cosmological constant (pm.cpp:7019-7030), (rho + p) theta over rho + p (pm.cpp:7142-7145)
      double rho_tot = Q.rg + Q.rb + Q.rc + Q.ru, rho_p_tot = Q.rg43 + Q.rb + Q.rc + Q.ru43;
      double delta_rho = Q.rg * dgam + Q.rb * db + Q.rc * dc + Q.ru * dur;
      double rpt = Q.rg43 * tgam + Q.rb * tb + Q.rc * tc + Q.ru43 * tur;
      double dn[CPT_MAX_NCDM] = {0., 0., 0.}, tn[CPT_MAX_NCDM] = {0., 0., 0.};
      if (NCDM) {   // the non-cold species: in the totals, and each on its own (pm.cpp:6341-6410, 7112-7118)
        delta_rho += N.D; rpt += N.T;
#pragma unroll
        for (int n = 0; n < CPT_MAX_NCDM; n++) {
          if (n >= P.nc.n_species) break;
          const double rho_n = bcast(Q.vnc, 3 * n), p_n = bcast(Q.vnc, 3 * n + 1);
          rho_tot += rho_n; rho_p_tot += rho_n + p_n;
          // (after the fluid switch delta and theta of a species are its first two fluid variables; before it, the momentum-bin integrals
          //  that sets_species_integrals of cpt_perturb_sets.inc formed for this sample)
          if (L.fic) { dn[n] = bcast(y, LN_F0 + 3 * n); tn[n] = bcast(y, LN_F0 + 3 * n + 1); }
          else { dn[n] = N.Dn[n] / rho_n; tn[n] = N.Tn[n] / (rho_n + p_n); }
        }
      }
      const double phi = (GAUGE == CPT_GAUGE_NEWTONIAN) ? eta : eta - aH * M.alpha;
      const double psi = (GAUGE == CPT_GAUGE_NEWTONIAN) ? M.psi : aH * M.alpha + M.alphap;
      if (lane == 0) {
        const size_t base = (size_t)ik * P.ntau + it, tstride = (size_t)P.nk * P.ntau;
        const double v[CPT_NTK] = {delta_rho / rho_tot, dgam, db, dc, dur, rpt / rho_p_tot, tgam, tb, tc, tur, phi, psi};
#pragma unroll
        for (int i = 0; i < CPT_NTK; i++) if (P.tp_tk[i] >= 0) P.src[P.tp_tk[i] * tstride + base] = v[i];
        if (NCDM) {
#pragma unroll
          for (int n = 0; n < CPT_MAX_NCDM; n++) {
            if (n >= P.nc.n_species) break;
            if (P.tp_dn >= 0) P.src[(P.tp_dn + n) * tstride + base] = dn[n];
            if (P.tp_tn >= 0) P.src[(P.tp_tn + n) * tstride + base] = tn[n];
          }
        }
      }
    }
  }
}

// perturb_approximations (pm.cpp:5443-5670) evaluated independently by every lane at its own tau
static __device__ __forceinline__ void approx_flags(const PtParams& P, double k, double tau, int* tca, int* rsa, int* ufa, int* nfa) {
  const AHK q = lookup_aHk(P.tabs, P.n_e, tau);
  const double a = q.a, H = q.H, dk = q.dk;
  const double tau_h = 1. / (H * a);
  if (dk == 0.) *tca = 0;
  else {
    const double tau_c = 1. / dk;
    *tca = ((tau_c / tau_h < P.tca_trig_h) && (tau_c * k < P.tca_trig_k)) ? 1 : 0;
  }
  *rsa = ((tau * k > P.rsa_trig) && (tau > P.tau_free_streaming) && (P.rsa_method != CPT_RSA_NONE)) ? 1 : 0;
  *ufa = (!MODE && P.has_ur && (tau * k > P.ufa_trig) && (P.ufa_method != CPT_UFA_NONE)) ? 1 : 0;   // no ur fluid for tensors
  *nfa = (HAS_BINS && (tau * k > P.nfa_trig) && (P.nfa_method != CPT_NCDMFA_NONE)) ? 1 : 0;             // pm.cpp:5606-5614
}

// 64-ary search for the time at which a monotone predicate flips between lo (false) and hi (true):
// kind 0: "no longer early enough to start" (pm.cpp:2590-2635), kind 1..4: approximation ap-1 differs from `ref`
static __device__ __forceinline__ double search_flip(const PtParams& P, double k, double lo, double hi, double tol_abs, double tol_rel,
                                           int kind, int ref, int lane) {
  for (int round = 0; round < 64; round++) {
    const double width = hi - lo;
    if (kind == 0 ? (width / lo <= tol_rel) : (width <= tol_abs)) break;
    const double t = lo + width * (double)(lane + 1) / 65.;
    bool pred;
    if (kind == 0) {
      const AHK q = lookup_aHk(P.tabs, P.n_e, t);
      pred = (q.a * q.H / q.dk > P.start_small_k) || (k / q.a / q.H > P.start_large_k) || (q.wdev > P.tol_ncdm_w);
    } else {
      int f0, f1, f2, f3;
      approx_flags(P, k, t, &f0, &f1, &f2, &f3);
      pred = ((kind == 1) ? f0 : (kind == 2) ? f1 : (kind == 3) ? f2 : f3) != ref;
    }
    const unsigned long long m = __ballot(pred);
    const int j = m ? (__ffsll((long long)m) - 1) : 64;  // first lane whose sample is past the flip
    const double nlo = (j == 0) ? lo : lo + width * (double)j / 65.;
    const double nhi = (j == 64) ? hi : lo + width * (double)(j + 1) / 65.;
    lo = nlo; hi = nhi;
  }
  return 0.5 * (lo + hi);
}

// ---- what the integrator wave carries beside the equations ------------------------------------------------------------------------
struct Ctx {
  Mailbox* mb; int posted, tail_seen;          // mailbox of the helper wave, samples posted so far, last value read of the helper's tail
  int fact_posted;   // factorisations handed to the helper wave for inversion (helper_inverse)
  double* rowbuf;    // (register-set kernels without a helper) LDS [MB_NANS]: the row of this wave's own look-up (row_store)
  double2* cwbuf;    // (register-set kernels with room in LDS) [CW_PAIRS][64]: the per-lane coefficients of the core RHS (lane_eq_store)
};

// ---- wave-wide cyclic reduction (the long tails of cpt_perturb_sets.inc: up to 64 multipoles of one ladder along the lanes) ----------
// value held by lane - S (DOWN) or lane + S of the wave, 0 beyond its ends: the LDS crossbar, executed by every lane
template <int S, bool DOWN>
static __device__ __forceinline__ double lane_far(double v) {
  const int lane = (int)(threadIdx.x & 63u), src = DOWN ? lane - S : lane + S;
  const double g = gather(v, (src & 63) * 4);
  return (src >= 0 && src < 64) ? g : 0.;
}
// one level of parallel cyclic reduction over the whole wavefront (a tail of up to 64 multipoles is one chain per wave): row l loses
// its couplings to l -+ S and gains couplings to l -+ 2S.  Six levels instead of 2 x (length - 1) dependent sweeps per solve.
template <int S>
static __device__ __forceinline__ void wave_pcr_level(double& a, double& c, double& d, double& al, double& ga) {
  const double dm = lane_far<S, true>(d), dq = lane_far<S, false>(d);
  const double am = lane_far<S, true>(a), cm = lane_far<S, true>(c), aq = lane_far<S, false>(a), cq = lane_far<S, false>(c);
  al = (a != 0.) ? a * fast_rcp(dm) : 0.;   // (a = 0 where the neighbour does not exist: never 0 * inf)
  ga = (c != 0.) ? c * fast_rcp(dq) : 0.;
  d = fma(-al, cm, fma(-ga, aq, d));
  a = -al * am;
  c = -ga * cq;
}
template <int S>
static __device__ __forceinline__ double wave_pcr_apply(double b, double al, double ga) {
  return fma(-al, lane_far<S, true>(b), fma(-ga, lane_far<S, false>(b), b));
}
// ---- structured linear algebra: (I - hg J) x = b -------------------------------------------------------------------
// J = [ J_cc  J_ct ]   core (nc <= 16 lanes, dense)        J_ct: only (parent of a tail, its l=3 element)
//     [ J_tc  J_tt ]   tails (three tridiagonal chains)     J_tc: only (l=3 element, parent)
// The reference discovers this sparsity numerically, orders it with AMD and runs a sparse LU (tools/sparse.c:130-599);
// here it is a property of the equations, known per regime:
//   * tails: eliminated from l_max downwards with the continued fraction d'_l = d_l - c_l a_{l+1}/d'_{l+1}.  Each lane
//     holds (a_l, d_l, c_l) of its row; the recurrence is run SYSTOLICALLY: every lane recomputes its d' from the
//     value its upper neighbour holds (one DPP wave_shl per sweep), after n sweeps the n lanes furthest from l_max
//     are final.  Three chains advance at once; no LDS, no global memory.
//   * the Schur complement on the core changes 3 diagonal entries only; the core matrix (<= 16 x 16) lives in
//     registers, one row per lane, and is factorised with threshold-diagonal pivoting (tools/sparse.c:171) by
//     fully unrolled readlane/fma code.
// f(j) for the core variables j = lo..NC-1 in ascending order (for_core) or NC-1..0 in descending order (for_core_down).  The ncdm
// kernels, whose core has 22 lanes of which a scheme uses 7 to 16, skip the lane groups that a scheme leaves idle as a whole - photons
// 0..5, ur 9..11, the three ncdm fluids 13.., 16.., 19.. - by ONE scalar branch per group (Layout: an idle core variable is an identity row
// and column of the Newton matrix).  A test per lane would break the unrolled substitution chains into 22 basic blocks: measured
// on the 13-lane core of the two-wave kernels, 11.1 -> 13.9 ms.
template <int A, int B, class F>
static __device__ __forceinline__ void core_run(int lo, F&& f) {
#pragma unroll
  for (int j = A; j <= B; j++) if (j >= lo) f(j);
}
template <int A, int B, class F>
static __device__ __forceinline__ void core_run_down(F&& f) {
#pragma unroll
  for (int j = B; j >= A; j--) f(j);
}
// (SYS = 1: the last compact interval of the register-set kernels - radiation streaming and both fluid approximations on - where the
//  scheme itself says which groups idle: photons, ur and the tails' auxiliary unknowns.  Five of the eight group tests go, and a
//  test is a scalar branch, ~20 cycles of a lone wavefront, in every one of the 7 - 13 nested passes of a factorisation.)
template <int SYS = 0, class F>
static __device__ __forceinline__ void for_core(unsigned pm, int lo, F&& f) {
  if (!NCDM) { core_run<0, NC - 1>(lo, f); return; }
  if constexpr (SYS == 1) {
    if (lo <= 8) core_run<6, 8>(lo, f);
    if (lo <= 12) core_run<12, 12>(lo, f);
    if (lo <= 15) core_run<13, (NCDM ? 15 : 0)>(lo, f);
    if (lo <= 18 && (pm & (7u << 16))) core_run<16, (NCDM ? 18 : 0)>(lo, f);
    if (lo <= 21 && (pm & (7u << 19))) core_run<19, (NCDM ? 21 : 0)>(lo, f);
    return;
  }
  if constexpr (SYS == 2) {   // (the other compact intervals: whatever idles among the first sixteen is an identity pivot, cheaper than a test)
    if (lo <= 15) core_run<0, (NCDM ? 15 : 0)>(lo, f);
    if (lo <= 18 && (pm & (7u << 16))) core_run<16, (NCDM ? 18 : 0)>(lo, f);
    if (lo <= 21 && (pm & (7u << 19))) core_run<19, (NCDM ? 21 : 0)>(lo, f);
    if constexpr (NCDM == 3) { if (pm & (7u << 22)) core_run<22, 24>(lo, f); }
    return;
  }
  if (lo <= 5 && (pm & 0x3Fu)) core_run<0, 5>(lo, f);
  if (lo <= 8) core_run<6, 8>(lo, f);
  if (lo <= 11 && (pm & 0xE00u)) core_run<9, 11>(lo, f);
  if (lo <= 12) core_run<12, 12>(lo, f);
  if (lo <= 15 && (pm & (7u << 13))) core_run<13, (NCDM ? 15 : 0)>(lo, f);
  if (lo <= 18 && (pm & (7u << 16))) core_run<16, (NCDM ? 18 : 0)>(lo, f);
  if (lo <= 21 && (pm & (7u << 19))) core_run<19, (NCDM ? 21 : 0)>(lo, f);
  if constexpr (NCDM == 3) { if (pm & (7u << 22)) core_run<22, 24>(lo, f); }
}
template <int SYS = 0, class F>
static __device__ __forceinline__ void for_core_down(unsigned pm, F&& f) {
  if (!NCDM) { core_run_down<0, NC - 1>(f); return; }
  if constexpr (SYS == 1) {
    if (pm & (7u << 19)) core_run_down<19, (NCDM ? 21 : 0)>(f);
    if (pm & (7u << 16)) core_run_down<16, (NCDM ? 18 : 0)>(f);
    core_run_down<13, (NCDM ? 15 : 0)>(f);
    core_run_down<12, 12>(f);
    core_run_down<6, 8>(f);
    return;
  }
  if constexpr (SYS == 2) {
    if constexpr (NCDM == 3) { if (pm & (7u << 22)) core_run_down<22, 24>(f); }
    if (pm & (7u << 19)) core_run_down<19, (NCDM ? 21 : 0)>(f);
    if (pm & (7u << 16)) core_run_down<16, (NCDM ? 18 : 0)>(f);
    core_run_down<0, (NCDM ? 15 : 0)>(f);
    return;
  }
  if constexpr (NCDM == 3) { if (pm & (7u << 22)) core_run_down<22, 24>(f); }
  if (pm & (7u << 19)) core_run_down<19, (NCDM ? 21 : 0)>(f);
  if (pm & (7u << 16)) core_run_down<16, (NCDM ? 18 : 0)>(f);
  if (pm & (7u << 13)) core_run_down<13, (NCDM ? 15 : 0)>(f);
  core_run_down<12, 12>(f);
  if (pm & 0xE00u) core_run_down<9, 11>(f);
  core_run_down<6, 8>(f);
  if (pm & 0x3Fu) core_run_down<0, 5>(f);
}

struct Jac {
  double* Jc;      // LDS [NC][64]: Jc[j * 64 + i] = J_cc(i, j) for lane i < NC, 0 on the other lanes.  Only the (rare)
                   // Jacobian refresh writes it and only the factorisation reads it: no reason to pin 26 VGPRs
  double jdiag;    // tail lanes: J_ll = -(D kappa' + G/tau) frozen at the time of the Jacobian (ev.cpp keeps J fixed)
};
// The bulky part of the factors lives in LDS, not in registers: row i of the core factors (L below / unit-diagonal U above the
// diagonal) of lane i < NC, and the cyclic-reduction multipliers of the tails.  Kept in registers they push the integrator past
// 256 VGPRs, and every use then costs a v_accvgpr_read per dword; from LDS a ds_read_b128 brings two doubles per instruction,
// issued ahead of their use.  Layout: pair p of lane l at fw[p * 64 + l] (conflict-free b128 accesses).
// Lane stride of the two LDS arrays below.  Only the core lanes (< NC) hold anything in them; the register-set kernels (NC = 22), whose
// resident k-modes per CU are limited by LDS, store 32 lanes per row: lanes >= 32 neither write nor use what they read.
static constexpr int JS = NCDM ? 32 : 64;
static __device__ __forceinline__ void jc_store(double* Jc, int j, int lane, double v) { if (JS == 64 || lane < JS) Jc[j * JS + lane] = v; }
static __device__ __forceinline__ double jc_load(const double* Jc, int j, int lane) {
  const double v = Jc[j * JS + (lane & (JS - 1))];
  return (JS == 64 || lane < JS) ? v : 0.;
}
static constexpr int FW_ACP = (NC + 1) / 2;      // pairs holding Ac[0..NC-1]
static constexpr int FW_PAIRS = FW_ACP + (PCR ? 4 : 0);   // + (al, ga) of the four reduction levels (row layout only: LDS is what limits
                                                         // the resident k-modes of the ncdm kernels)
struct LuReg {
  double2* fw;     // LDS [FW_PAIRS][64]
  const double2* inv;  // LDS [FW_ACP][64]: rows of the inverse of the core block, written by the helper wave (helper_inverse); same layout as fw
  double rpivc;    // lane j < nc: reciprocal of the j-th core pivot
  int rowperm;     // lane i < nc: original row now at position i (identity on tail lanes)
  int permuted;    // (wave-uniform) some rows were exchanged: rowperm is not the identity
  double rinv;     // tail lanes: 1 / d'_l   (0 on core lanes)
  double g;        // tail lanes: c_l / d'_{l+1}, the downward-sweep multiplier (0 on the l_max element and on core lanes)
  double r;        // tail lanes: a_l / d'_l, the upward-sweep multiplier
  double cpar;     // core parents of a tail: coupling to the tail's l=3 element (0 elsewhere)
};

template <int N>
static __device__ __forceinline__ double reg_get(const double (&a)[N], int i) {
  double v = 0.;
#pragma unroll
  for (int j = 0; j < N; j++) if (j == i) v = a[j];
  return v;
}

// (NCDM) the rows of the two auxiliary unknowns u_D, u_T (lanes LN_ND, LN_NT) read, after the chains are eliminated,
//   u_D - alpha_D1 dmc - alpha_D2 dms = sum_chains w_D [T^-1 r]_0     with dmc = sum_j gmc_j x_j, dms = sum_j gms_j x_j
// where gmc_j / gms_j are the responses of (metric_continuity, metric_shear) to unit core variable j (al[] = the four alphas)
// b <- the right-hand side after the four reduction levels (u = b * rinv solves T u = b on every tail at once)
// (all four levels, whatever the tails' lengths: a level the factorisation did not need has al = ga = 0 and leaves b alone,
//  which costs two multiply-adds where a test of maxlen costs a scalar reload, a compare and a branch per level)
static __device__ __forceinline__ double pcr_apply(const LuReg& F, double b, int lane) {
  const double2 m0 = F.fw[(FW_ACP + 0) * 64 + lane], m1 = F.fw[(FW_ACP + 1) * 64 + lane], m2 = F.fw[(FW_ACP + 2) * 64 + lane],
                m3 = F.fw[(FW_ACP + 3) * 64 + lane];   // {al, ga} of the four levels
  b = fma(-m0.x, row_shr0<1>(b), fma(-m0.y, row_shl0<1>(b), b));
  b = fma(-m1.x, row_shr0<2>(b), fma(-m1.y, row_shl0<2>(b), b));
  b = fma(-m2.x, row_shr0<4>(b), fma(-m2.y, row_shl0<4>(b), b));
  b = fma(-m3.x, row_shr0<8>(b), fma(-m3.y, row_shl0<8>(b), b));
  return b;
}
template <int S>
static __device__ __forceinline__ void pcr_level(double& a, double& c, double& d, double& al, double& ga) {
  const double dm = row_shr0<S>(d), dq = row_shl0<S>(d);
  const double am = row_shr0<S>(a), cm = row_shr0<S>(c), aq = row_shl0<S>(a), cq = row_shl0<S>(c);
  al = (a != 0.) ? a * fast_rcp(dm) : 0.;   // (a = 0 where the neighbour does not exist: never 0 * inf)
  ga = (c != 0.) ? c * fast_rcp(dq) : 0.;
  d = fma(-al, cm, fma(-ga, aq, d));
  a = -al * am;
  c = -ga * cq;
}

template <int SYS = 0>
static __device__ __forceinline__ bool factorise(const LaneEq& e, const Jac& J, double hg, int maxlen, int lane, LuReg& F,
                                                 const double* al = nullptr, double gmc = 0., double gms = 0., int aux = 0) {
  ISA_MARK("FACT_BEGIN");
  lane = opaque(lane);
  const int chain = opaque(e.chain);
  // ---- tails ----
  const double a = chain ? -hg * e.A : 0.;             // coefficient of x_{l-1} in row l
  const double c = (chain && !e.last) ? hg * e.B : 0.; // coefficient of x_{l+1}
  const double d = 1.0 - hg * J.jdiag;
  double r;
  if (PCR) {
    // T (tridiagonal inside each tail; the first element's a couples to the core parent and stays outside)
    double ta = (chain && !e.first) ? a : 0., tc = c, td = chain ? d : 1.;
    double al[4] = {0., 0., 0., 0.}, ga[4] = {0., 0., 0., 0.};
    if (maxlen > 1) pcr_level<1>(ta, tc, td, al[0], ga[0]);
    if (maxlen > 2) pcr_level<2>(ta, tc, td, al[1], ga[1]);
    if (maxlen > 4) pcr_level<4>(ta, tc, td, al[2], ga[2]);
    if (maxlen > 8) pcr_level<8>(ta, tc, td, al[3], ga[3]);
#pragma unroll
    for (int i = 0; i < 4; i++) F.fw[(FW_ACP + i) * 64 + lane] = make_double2(al[i], ga[i]);
    const double rinv = fast_rcp(td);
    F.rinv = chain ? rinv : 0.;
    // v = T^-1 (a_first e_first): what a unit core parent sends into its tail
    r = pcr_apply(F, (chain && e.first) ? a : 0., lane) * F.rinv;
    F.r = r; F.g = 0.;
  } else {
  double dp = d;
  r = 0.;
  for (int s = 0; s < maxlen; s++) {
    r = a * fast_rcp(dp);
    const double r_up = lane_above(r);
    dp = fma(-c, r_up, d);
  }
  const double rinv = fast_rcp(dp);
  r = a * rinv;
  const double rinv_up = lane_above(rinv);
  F.rinv = chain ? rinv : 0.; F.r = chain ? r : 0.;
  F.g = c * rinv_up;
  }
  ISA_MARK("FACT_CORE");
  // ---- core: A_cc = I - hg J_cc, Schur-corrected on the diagonal of the parents of the tails ----
  const double cpar = hg * e.Bpar;               // row parent, column l3:  -hg * (-B);  0 on every other lane
  F.cpar = cpar;
  const double r3 = gather(r, e.first_addr);
  const double schur = cpar * r3;
  double A[NC];
#pragma unroll
  for (int j = 0; j < NC; j++) A[j] = ((j == lane) ? 1.0 - schur : 0.0) - hg * jc_load(J.Jc, j, lane);   // J.Jc = 0 outside the core
  if (NCDM) {
    // aux: which auxiliary rows of the bordered system exist (register-set kernels) - bit 0: the tails' l = 3 elements, al[4..6];
    // bit 1: the ncdm density / momentum sums, al[0..3]
    if (LONG && (aux & 1)) {
      // auxiliary unknown u_t = l = 3 element of tail t, in lane LN_T3 + t: u_t - alpha_t x_parent(t) = [T_t^-1 r_t]_first,
      // alpha_t = hg [T_t^-1 (a_first e_first)]_first from the tail's set
      A[LN_SG] -= (lane == LN_T3) ? al[4] : 0.;
      A[LN_P2] -= (lane == LN_T3 + 1) ? al[5] : 0.;
      A[LN_SUR] -= (lane == LN_T3 + 2) ? al[6] : 0.;
    }
    if (HAS_BINS && (aux & 2)) {
    const double c1 = (lane == LN_ND) ? al[0] : (lane == LN_NT) ? al[2] : 0., c2 = (lane == LN_ND) ? al[1] : (lane == LN_NT) ? al[3] : 0.;
#pragma unroll
    for (int j = 0; j < NC; j++) A[j] -= c1 * bcast(gmc, j) + c2 * bcast(gms, j);
    }
  }
  int rowperm = lane, permuted = 0;
  double rpivc = 1.;
  bool ok = true;
  const unsigned pm = e.pmask;
  for_core<SYS>(pm, 0, [&](const int j) {
    const double mag = (lane >= j) ? fabs(A[j]) : 0.;      // rows >= NC hold zeros in the core columns
    const double diag = bcast(mag, j);
    if (__ballot(mag > 1e3 * diag) != 0ull || diag == 0.) {  // rare: the diagonal is not an acceptable pivot
      const double big = wave_max(mag);
      if (big == 0.) ok = false;
      const int p = __ffsll((long long)__ballot((double)f32_up(mag) == big && big > 0.)) - 1;  // wave_max rounds up to float
      if (p > j) {
        permuted = 1;
        // exchange rows p and j (register rows of two lanes) and the row bookkeeping
#pragma unroll
        for (int cidx = 0; cidx < NC; cidx++) {
          const double vp = bcast(A[cidx], p), vj = bcast(A[cidx], j);
          if (lane == p) A[cidx] = vj;
          if (lane == j) A[cidx] = vp;
        }
        const int rp_p = __builtin_amdgcn_readlane(rowperm, p), rp_j = __builtin_amdgcn_readlane(rowperm, j);
        if (lane == p) rowperm = rp_j;
        if (lane == j) rowperm = rp_p;
      }
    }
    const double rp = fast_rcp(bcast(A[j], j));
    if (lane == j) rpivc = rp;
    const double m = (lane > j) ? A[j] * rp : 0.;
    if (lane > j) A[j] = m;
    for_core<SYS>(pm, j + 1, [&](const int cidx) {
      const double pj = bcast(A[cidx], j);
      A[cidx] = fma(-m, pj, A[cidx]);   // m = 0 on rows <= j
    });
  });
  ISA_MARK("FACT_SCALE");
  // unit-diagonal U: scale the upper part of every row by its reciprocal pivot
#pragma unroll
  for (int j = 0; j < NC; j++) A[j] = (j > lane) ? A[j] * rpivc : A[j];
  if (JS == 64 || lane < JS) {
#pragma unroll
    for (int q = 0; q < FW_ACP; q++) F.fw[q * JS + lane] = make_double2(A[2 * q], (2 * q + 1 < NC) ? A[2 * q + 1] : 0.);
  }
  F.rpivc = rpivc;
  F.rowperm = rowperm;
  F.permuted = __builtin_amdgcn_readfirstlane(permuted);
  ISA_MARK("FACT_END");
  return ok;
}

// solve (I - hg J) x = b; lane i holds b_i on entry and x_i on return
template <bool USE_INV = false, int SYS = 0>
static __device__ __forceinline__ double lu_solve(const LaneEq& e, const LuReg& F, int maxlen, double b, int lane) {
  lane = opaque(lane);
  const int chain = opaque(e.chain);
  // 1. tails, downward sweep: b'_l = b_l - (c_l / d'_{l+1}) b'_{l+1}; the l_max element is final at once
  double u;
  // the rows of the core factors: requested now, they arrive behind the tail reduction
  double Ac[2 * FW_ACP];
#pragma unroll
  for (int q = 0; q < FW_ACP; q++) { const double2 v = (USE_INV ? F.inv : F.fw)[q * JS + (lane & (JS - 1))]; Ac[2 * q] = v.x; Ac[2 * q + 1] = v.y; }   // (JS = 32: lanes >= 32 read rows that are not theirs, see the end)
  if (PCR) u = pcr_apply(F, chain ? b : 0., lane) * F.rinv;   // T^-1 b on every tail lane, 0 on core lanes
  else {
    double bp = b;
    for (int s = 1; s < maxlen; s++) bp = fma(-F.g, lane_above(bp), b);
    // 2. core right-hand side: parents of the tails see b'_3 / d'_3
    u = bp * F.rinv;                       // 0 on core lanes
  }
  double t3;
  if (PCR) {
    // the three parents (shear_g, pol2, shear_ur) take u from the first lane of their tail's row: three broadcasts and two selects
    // are a shorter dependency chain than a trip through the LDS crossbar (ds_bpermute + wait), and cpar is 0 on every other lane
    const double u1 = bcast(u, 16), u2 = bcast(u, 32), u3 = bcast(u, 48);
    t3 = (lane == LN_SG) ? u1 : (lane == LN_P2) ? u2 : u3;
  } else t3 = gather(u, e.first_addr);          // executed by every lane
  const double bc = fma(-F.cpar, t3, b);
  // 3. core solve with the register-resident factors (idle rows / lanes >= NC are identity rows: x = b there)
  // (rows are only exchanged when a diagonal entry was not an acceptable pivot: almost never, and then the gather is skipped)
  double x;
  const unsigned pm = e.pmask;
  if constexpr (USE_INV) {
    // x = A_cc^-1 bc with row i of the inverse in lane i (zero rows on the tail lanes): NC independent broadcasts and two chains of
    // multiply-adds instead of 2 NC dependent (broadcast, multiply-add) pairs - 210 against 670 cycles on a lone wavefront
    static_assert(NCDM == 0, "inverse of the core: two-wave kernels only");
    double x0 = 0., x1 = 0.;
#pragma unroll
    for (int j = 0; j < NC; j++) {
      const double bj = bcast(bc, j);
      if (j & 1) x1 = fma(Ac[j], bj, x1); else x0 = fma(Ac[j], bj, x0);
    }
    x = x0 + x1;
  } else {
  x = chain ? 0. : bc;
  if (F.permuted) x = gather(x, F.rowperm * 4);
  // Row i keeps its L entries (columns j < i) and its U entries (j > i) in ONE register array, so each substitution step must
  // switch the entry off on the rows it does not concern.  Clearing the HIGH word alone does that in one v_cndmask instead of
  // two: what is left is a subnormal (|m| < 2^-1022), and m * xj then vanishes against x unless |xj / x| > 2^970.
  for_core<SYS>(pm, 0, [&](const int j) {   // forward, unit lower
    const double xj = bcast(x, j);
    const double m = __hiloint2double((lane > j) ? __double2hiint(Ac[j]) : 0, __double2loint(Ac[j]));
    x = fma(-m, xj, x);
  });
  x *= F.rpivc;   // 1 outside the core
  for_core_down<SYS>(pm, [&](const int j) {  // backward, unit upper
    const double xj = bcast(x, j);
    const double uj = __hiloint2double((lane < j) ? __double2hiint(Ac[j]) : 0, __double2loint(Ac[j]));
    x = fma(-uj, xj, x);
  });
  }
  // 4. tails, upward sweep: x_l = b'_l / d'_l - (a_l / d'_l) x_{l-1}; the l=3 element takes x_{l-1} from its core parent
  if (maxlen > 0) {
    double xpar;
    if (PCR) {
      const double x1 = bcast(x, LN_SG), x2 = bcast(x, LN_P2), x3 = bcast(x, LN_SUR);
      xpar = (lane < 32) ? x1 : (lane < 48) ? x2 : x3;   // (core lanes of row 0 take x1: unused there, `chain` selects below)
    } else xpar = gather(x, e.parent_addr);
    double xt;
    if (PCR) xt = fma(-F.r, xpar, u);              // x_t = T^-1 b_t - x_parent T^-1 (a_first e_first)
    else {
      const double u0 = e.first ? fma(-F.r, xpar, u) : u;
      const double rr = e.first ? 0. : F.r;
      xt = u0;
      for (int s = 1; s < maxlen; s++) xt = fma(-rr, lane_below(xt), u0);
    }
    if (chain) x = xt;
  }
  if (JS < 64) x = (lane >= JS && !chain) ? b : x;   // (identity rows; what these lanes computed from rows of other lanes is dropped)
  return x;
}


// ---- the inverse of the core block, by the helper wave -------------------------------------------------------------------------
// The two triangular sweeps of a solve are 2 NC dependent (broadcast, multiply-add) pairs: 670 cycles of a lone wavefront, 11 % of the
// kernel.  With row i of A_cc^-1 in lane i the same product is NC independent broadcasts and two short chains: 210 cycles.  Forming
// the inverse costs more than the factorisation, and the integrator would lose what it wins - but the helper wave idles three quarters
// of the time.  So: the integrator factorises as before, posts (fact_seq) and goes on with the LU solve for the FIRST Newton iteration;
// the helper reads the factors from LDS, computes U^-1 D^-1 L^-1 with one COLUMN per lane (the factors' entries are wave-uniform
// broadcast reads, the arithmetic NC (NC - 1) multiply-adds per lane with no cross-lane traffic), stores it row-wise and posts
// (inv_seq); every later solve with these factors - eight of nine - takes the product form.  Which solve takes which form is fixed
// by the program, not by timing: results are reproducible bit for bit.  A factorisation that had to exchange rows (almost never) is
// not posted and keeps the LU solve.  The Newton iteration does not care that the two forms round differently.
// Safe without further handshakes: the integrator accepts an inverse only if inv_seq equals the factorisation it posted last; the helper
// tags its result with the fact_seq it read BEFORE reading the factors, so an inverse computed from factors that were being overwritten
// carries a stale tag and is never used.
static constexpr bool INV = (NCDM == 0);
static __device__ __forceinline__ void helper_inverse(const double2* fw, const double* rp, double2* inv, int lane) {
  ISA_MARK("INV_BEGIN");
  double X[NC];
#pragma unroll
  for (int i = 0; i < NC; i++) X[i] = (i == lane) ? 1.0 : 0.0;
  // L (unit lower triangle of the rows), then D^-1
#pragma unroll
  for (int i = 1; i < NC; i++) {
    double acc = X[i];
#pragma unroll
    for (int q = 0; 2 * q < i; q++) {
      const double2 v = fw[q * JS + i];          // (row i, columns 2q and 2q + 1: the same address on every lane)
      acc = fma(-v.x, X[2 * q], acc);
      if (2 * q + 1 < i) acc = fma(-v.y, X[2 * q + 1], acc);
    }
    X[i] = acc;
  }
#pragma unroll
  for (int i = 0; i < NC; i++) X[i] *= rp[i];
  // U (unit upper triangle)
#pragma unroll
  for (int i = NC - 2; i >= 0; i--) {
    double acc = X[i];
#pragma unroll
    for (int q = (i + 1) / 2; q < FW_ACP; q++) {
      const double2 v = fw[q * JS + i];
      if (2 * q > i) acc = fma(-v.x, X[2 * q], acc);
      if (2 * q + 1 < NC) acc = fma(-v.y, X[2 * q + 1], acc);
    }
    X[i] = acc;
  }
  // lane c holds column c: entry (i, c) goes to pair c / 2 of row i
  if (lane < NC) {
    double* o = reinterpret_cast<double*>(inv) + ((lane >> 1) * JS) * 2 + (lane & 1);
#pragma unroll
    for (int i = 0; i < NC; i++) o[2 * i] = X[i];
  }
  ISA_MARK("INV_END");
}

// adjust_stepsize (ev.cpp:907-943): dif(1:k) <- dif(1:k) R(1:k,1:k) U(1:k,1:k) with R[m][p] = prod_{i<=m} (i - (p+1) r)/(i+1) and the
// constant upper-triangular U.  Evaluated right to left, w[p] = sum_m dif[m] R[m][p] first: ~130 instructions instead of the
// ~1 500 of forming R U (20 divisions, two 5x5 products) - this runs on every change of step size.  Static indices => registers only.
static __device__ __forceinline__ void adjust_stepsize(double* dif, double r, int k) {
  double tv[5], w[5];
#pragma unroll
  for (int m = 0; m < 5; m++) tv[m] = (m < k) ? dif[m] : 0.;   // rows m >= k drop out
#pragma unroll
  for (int p = 0; p < 5; p++) {
    const double c = (p + 1) * r;
    double R = -c, acc = tv[0] * R;
#pragma unroll
    for (int m = 1; m < 5; m++) {
      const double inv = (m == 1) ? 0.5 : (m == 2) ? 1.0 / 3.0 : (m == 3) ? 0.25 : 0.2;
      R *= (m - c) * inv;
      acc = fma(tv[m], R, acc);
    }
    w[p] = acc;
  }
  // columns of U = {{-1,-2,-3,-4,-5},{0,1,3,6,10},{0,0,-1,-4,-10},{0,0,0,1,5},{0,0,0,0,-1}}: column jj < k only meets p <= jj < k
  const double d0 = -w[0];
  const double d1 = fma(-2., w[0], w[1]);
  const double d2 = fma(-3., w[0], fma(3., w[1], -w[2]));
  const double d3 = fma(-4., w[0], fma(6., w[1], fma(-4., w[2], w[3])));
  const double d4 = fma(-5., w[0], fma(10., w[1], fma(-10., w[2], fma(5., w[3], -w[4]))));
  if (0 < k) dif[0] = d0;
  if (1 < k) dif[1] = d1;
  if (2 < k) dif[2] = d2;
  if (3 < k) dif[3] = d3;
  if (4 < k) dif[4] = d4;
}

struct Stat { int steps, failed, fevals, jacs, lus, solves; };


// element `i` of the backward-difference array; i is wave-uniform (held in an SGPR), so this is a scalar jump to one register
// move instead of seven compare + select pairs (static indices only => no scratch)
static __device__ __forceinline__ double dif_get(const double* dif, int i) {
  switch (__builtin_amdgcn_readfirstlane(i)) {
    case 0: return dif[0];
    case 1: return dif[1];
    case 2: return dif[2];
    case 3: return dif[3];
    case 4: return dif[4];
    case 5: return dif[5];
    case 6: return dif[6];
    default: return 0.;
  }
}

// ---- non-cold species in the fluid approximation -------------------------------------------------------------------------------
// Once the ncdm fluid approximation is on every species is three variables (delta, theta, sigma) - yet these intervals hold 90 % of the
// steps of the heaviest mode (the fluids oscillate until today).  At that switch the register-set kernels (cpt_perturb_sets.inc)
// integrate the momentum hierarchies into the fluid variables, which become ordinary members of the dense core in lanes LN_F0.., and
// go on with the structured integrator ndf15s below - no sets, no auxiliary unknowns:
//   SYS = 2: any scheme of the other species (photon / ur hierarchies still running): the general RHS + the fluid equations
//   SYS = 1: radiation streaming and the ur fluid on as well (the last interval): baryons, cdm, eta and the fluids are all that is left,
//            4 + 3 N variables, and the RHS is written out for exactly that
static __device__ __forceinline__ int fluid_lane(int species, int j) { return LN_F0 + 3 * species + j; }
// perturb_derivs (pm.cpp:7861-9218 with perturb_einstein, perturb_total_stress_energy, perturb_rsa_delta_and_theta and the fluid
// equations of pm.cpp:8737-8823 folded in); synchronous gauge.  Leaves M and N describing (tau, y) for the sources.
// (FETCH = false: the row of the tables is there already; LK then says where: 1 in Q with the ncdm columns in Q.ncv (from the helper),
//  0 in Q with the ncdm columns in Q.vnc (own look-up), 3 in LDS at Q.row - SYS = 2 only)
template <int SYS, int LK, bool FETCH = true, bool ECO = false>
static __device__ __forceinline__ double rhs_fluid(const PtParams& P, const Layout& L, const LaneEq& e, Lookup& Q, Metric& M, NcIn& N, double k, double inv_k2,
                                                   double tau, double y, int lane) {
  if (FETCH) {
    if (LK == 1) { if (!mb_fetch(Q, tau, lane)) return nan(""); }
    else lookup(P, Q, tau, lane);
  }
  static_assert(LK != 3 || !FETCH, "rows in LDS: the caller has put the row there");
  // (LK = 3: the whole row - 22 doubles and the nine ncdm columns - by sixteen uniform b128 reads issued together, behind one wait)
  double rw[32];
  if (LK == 3) {
    const double2* r2 = reinterpret_cast<const double2*>(Q.row);
#pragma unroll
    for (int i = 0; i < 16; i++) { const double2 v = r2[i]; rw[2 * i] = v.x; rw[2 * i + 1] = v.y; }
  }
#define QF(f) ((LK == 3) ? rw[ROW_##f] : Q.f)
  const double aH = QF(aH), k2 = k * k;
  const int ln = opaque(lane);
  // the non-cold fluids: integrals for the Einstein equations, and this lane's species (wave-uniform per species, selected per lane)
  double D = 0., T = 0., S = 0., rho_l = 1., p_l = 1., pp_l = 1.;
#pragma unroll
  for (int n = 0; n < CPT_MAX_NCDM; n++) {
    if (n >= P.nc.n_species) break;
    double rho, pr, pp;
    if (LK == 3) { rho = rw[22 + 3 * n]; pr = rw[23 + 3 * n]; pp = rw[24 + 3 * n]; }
    else if (LK == 1) { rho = reg_get(Q.ncv, 3 * n); pr = reg_get(Q.ncv, 3 * n + 1); pp = reg_get(Q.ncv, 3 * n + 2); }
    else { rho = bcast(Q.vnc, 3 * n); pr = bcast(Q.vnc, 3 * n + 1); pp = bcast(Q.vnc, 3 * n + 2); }
    const int l0 = fluid_lane(n, 0);
    D = fma(rho, bcast(y, l0), D); T = fma(rho + pr, bcast(y, l0 + 1), T); S = fma(rho + pr, bcast(y, l0 + 2), S);
    const bool mine = (ln >= l0) && (ln <= l0 + 2);
    rho_l = mine ? rho : rho_l; p_l = mine ? pr : p_l; pp_l = mine ? pp : pp_l;
  }
  N.D = D; N.T = T; N.S = S;
  double dy, mc, ms;
  if (SYS == 2) {
    dy = rhs<(LK == 3) ? 3 : 2, ECO>(P, L, e, Q, M, k, inv_k2, tau, y, lane, &N);   // (0 on the fluid lanes: their LaneEq is empty)
    mc = 0.5 * M.hp; ms = k2 * M.alpha;
  } else {
    const double a2 = QF(a2), kap = QF(kap), cb2 = QF(cb2);
    const double db = bcast(y, LN_DB), tb = bcast(y, LN_TB), dc = bcast(y, LN_DC), eta = bcast(y, LN_ETA);
    double delta_rho = QF(rb) * db + D, rpt = QF(rb) * tb + T;
    const double rps = S;
    delta_rho += QF(rc) * dc;   // (an absent species has zero density in the tables)
    const double hp = (Q.k2s2 * eta + 1.5 * a2 * delta_rho) * QF(two_over_aH);                       // pm.cpp:5913-5914
    // radiation streaming: photons and ur follow the metric (pm.cpp:9530-9636)
    double rdg = 0., rtg = 0., rdur = 0., rtur = 0.;
    if (P.rsa_method != CPT_RSA_NULL) { rdg = 4. * inv_k2 * (aH * hp - k2 * eta); rtg = -0.5 * hp; }
    if (P.rsa_method == CPT_RSA_MD_WITH_REIO) {
      rdg += -4. * inv_k2 * kap * (tb + 0.5 * hp);
      rtg += 3. * inv_k2 * (QF(ddkappa) * (tb + 0.5 * hp) + kap * (-aH * tb + cb2 * k2 * db - aH * hp + k2 * eta));
    }
    if (P.rsa_method != CPT_RSA_NULL) { rdur = 4. * inv_k2 * (aH * hp - k2 * eta); rtur = -0.5 * hp; }
    delta_rho += QF(rg) * rdg;
    rpt += QF(rg43) * rtg;
    delta_rho += QF(ru) * rdur; rpt += QF(ru43) * rtur;
    const double etap = (1.5 * a2 * rpt + (CURV ? 0.5 * P.K * hp : 0.)) * Q.inv_k2s2;             // pm.cpp:5938
    const double alpha = (hp + 6. * etap) * 0.5 * inv_k2;
    const double alphap = -2. * aH * alpha + eta - 4.5 * (a2 * inv_k2) * rps;
    M.hp = hp; M.etap = etap; M.alpha = alpha; M.alphap = alphap; M.rsa_dg = rdg; M.rsa_tg = rtg;
    mc = 0.5 * hp; ms = k2 * alpha;
    const double dtb = -aH * tb + k2 * cb2 * db + QF(R) * kap * (rtg - tb);                          // pm.cpp:8108-8113 with the streaming theta_g
    dy = 0.;
    dy = (ln == LN_DB) ? -(tb + mc) : dy;
    dy = (ln == LN_TB) ? dtb : dy;
    dy = (ln == LN_DC && P.has_cdm) ? -mc : dy;
    dy = (ln == LN_ETA) ? etap : dy;
  }
  {  // fluid lanes (pm.cpp:8737-8823): j = 0 delta, 1 theta, 2 sigma; ym / yp: the species' neighbouring variable
    const int f = ln - LN_F0, sp = (f >= 6) ? 2 : (f >= 3) ? 1 : 0, j = f - 3 * sp;
    const bool fluid = (f >= 0) && (f < 3 * CPT_MAX_NCDM) && (sp < P.nc.n_species);
    const double ym = lane_below(y), yp = lane_above(y);
    const double w = p_l * fast_rcp(rho_l), inv_1pw = fast_rcp(1. + w), pp_over_p = pp_l * fast_rcp(p_l);
    const double ca2 = w / 3. * inv_1pw * (5. - pp_over_p), ceff2 = ca2;
    const double cvis2 = (P.nfa_method == CPT_NCDMFA_HU) ? w : 3. * w * ca2;
    const double s2 = CURV ? sqrt(fmax(1.0 - 3. * P.K / k2, 0.)) : 1.;
    double fv;
    if (j == 0) fv = -(1. + w) * (yp + mc) - 3. * aH * (ceff2 - w) * y;
    else if (j == 1) fv = ceff2 * inv_1pw * k2 * ym - k2 * yp - aH * (1. - 3. * ca2) * y;
    else {
      const double src = 8. / 3. * cvis2 * inv_1pw * s2;
      if (P.nfa_method == CPT_NCDMFA_HU) fv = src * (ym + ms) - 3. * aH * ca2 * fast_rcp(w) * y;
      else fv = src * (ym + ((P.nfa_method == CPT_NCDMFA_MB) ? ms : mc)) - 3. * (aH * (2. / 3. - ca2 - pp_over_p / 3.) + QF(inv_tau)) * y;
    }
    dy = fluid ? fv : dy;
  }
#undef QF
  return dy;
}

// evolver_ndf15 (ev.cpp:62-705) for one interval of constant approximation scheme, as structured code (initial step, step loop, Newton
// loop, order selection): with the table look-ups on the helper wave an inlined RHS is ~300 instructions.  ONE wave runs it on ONE copy
// of the control state; every condition is wave-uniform.  Returns 0 / error code (1 step too small, 2 singular, 4 budget, 5 helper
// unresponsive).
// SYS = 0: the integrator wave of the two-wave kernels (NCDM = 0).  SYS = 1, 2: the compact intervals of the register-set kernels
// (cpt_perturb_sets.inc) once the ncdm fluid approximation is on - every species three core variables (rhs_fluid above).
// HELPED: with a helper wave (look-ups one step ahead, source samples: launches that are resident at once, latency is all that counts) -
// or without, own table look-ups and samples evaluated in place (grids larger than the chip: a lone wave leaves room for more k-modes).
template <int SYS, bool HELPED = true>
static __device__ __forceinline__ int ndf15s(const PtParams& P, const Layout& L, const LaneEq& e, Ctx& C, Lookup& Q, Metric& M, double k, double inv_k2,
                                             double t0, double tfinal, double& y_io, Stat& st, int lane, int& budget, double* jac_lds,
                                             double2* fw_lds, unsigned long long* prof, int ik = 0) {
  PROF_DECL;
  const double eps = 1e-16, threshold = 1e-15, rtol = P.rtol, inv_rtol = 1.0 / rtol;
  const int maxit = 4, maxk = 5;
  const double* ts = P.tau_s;
  const int tres = P.ntau;
  const double htspan = fabs(tfinal - t0), hmax = (tfinal - t0) / 10.0;
  const int maxlen = L.maxlen;
  const double minnrm = 100 * eps;   // (see ndf15: upper bound of 100 eps max |ynew / wt|)

  Jac J;
  J.Jc = jac_lds;
  for (int j = 0; j < NC; j++) jc_store(J.Jc, j, lane, 0.);
  J.jdiag = 0.;
  LuReg F;
  F.fw = fw_lds;
  F.inv = fw_lds + FW_PAIRS * JS;
  constexpr bool USE_INV = INV && HELPED && SYS == 0;   // the helper inverts the core block behind the integrator's back (helper_inverse)
  int inv_want = -1; bool inv_ok = false, lu_first = true;
  F.rpivc = 1.; F.rowperm = lane; F.permuted = 0; F.rinv = F.r = F.g = F.cpar = 0.;
  double y = y_io;
  double dif[7] = {0., 0., 0., 0., 0., 0., 0.};
  // sample times: the next one and the one after it sit in registers (the second is loaded one sample ahead, so its trip to
  // memory is never on the path of a step)
  int next = 0;
  while (next < tres && ts[next] < t0) next++;
  double tnext = (next < tres) ? ts[next] : 1e300, tnext2 = (next + 1 < tres) ? ts[next + 1] : 1e300;

  NcIn N = {0., 0., 0., {0., 0., 0.}, {0., 0., 0.}, {0., 0., 0.}};
  // ROWQ: the scalar integrator of the two-wave kernels handles its table rows by request number (mb_post / mb_take): it says which row
  // an evaluation uses before it evaluates, and the RHS reads it from LDS
  constexpr bool ROWQ = HELPED;
  int seq_this = 0, seq_next = 0;
  bool have_next = false, post_this = true, post_next = true;
  if constexpr (ROWQ) { Q.rq_tau0 = Q.rq_tau1 = -1.; Q.row_tau = -1.; }   // (what an integrator that finds its rows by time knew is void: mb_post does not keep it)
  auto eval = [&](double tq, double yq) {
    st.fevals++;
#ifdef CPT_CHECK_ROWS
    if constexpr (ROWQ) { if (Q.row[31] != tq && lane == 0) printf("row mismatch: block %d SYS %d row %.17g eval %.17g (step %d)\n", (int)blockIdx.x, SYS, Q.row[31], tq, st.steps); }
#endif
    if constexpr (SYS != 0 && ROWQ) return rhs_fluid<SYS, 3, false>(P, L, e, Q, M, N, k, inv_k2, tq, yq, lane);
    else if constexpr (SYS != 0) return rhs_fluid<SYS, HELPED ? 1 : 0>(P, L, e, Q, M, N, k, inv_k2, tq, yq, lane);
    else if constexpr (ROWQ) return rhs<3>(P, L, e, Q, M, k, inv_k2, tq, yq, lane);
    else return rhs<1>(P, L, e, Q, M, k, inv_k2, tq, yq, lane);
  };
  auto request = [&](double tq) { if constexpr (HELPED && !ROWQ) mb_request(Q, tq, lane); };
  const double no_alpha[4] = {0., 0., 0., 0.};
  // J e_r = f(t, e_r): exact, the system is linear and homogeneous; idle variables have no column; tails analytic
  auto jacobian = [&](double tq) {
    const double keep = M.tca_shear_g;
    for (int r = 0; r < NC; r++) {
      if (!((e.pmask >> r) & 1u)) continue;
      const double col = eval(tq, (lane == r) ? 1.0 : 0.0);
      jc_store(J.Jc, r, lane, (lane < NC) ? col : 0.);
    }
    if constexpr (ROWQ) J.jdiag = -(e.D * Q.row[ROW_kap] + e.G * Q.row[ROW_kcot] + (CURV ? e.Gt * Q.row[ROW_inv_tau] : 0.));
    else J.jdiag = -(e.D * Q.kap + e.G * Q.kcot + (CURV ? e.Gt * Q.inv_tau : 0.));   // frozen at the time of this Jacobian (ev.cpp keeps J fixed)
    M.tca_shear_g = keep;
    st.jacs++;
  };

  // ------------------------------------------------------------------ first guess of h (ev.cpp:225-283)
  double t = t0, absh, h;
  const double hmin0 = 16.0 * eps * fabs(t0);
  PROF_START();
  if constexpr (ROWQ) { if (!mb_take(Q, mb_post(Q, t, lane))) return 5; }
  jacobian(t);
  {
    const double f0 = eval(t, y);
    const double wt = fmax(fabs(y), threshold);
    double rh = wave_max(1.25 / sqrt(rtol) * fabs(f0 / wt));
    absh = fmin(hmax, htspan);
    if (uni(absh * rh > 1.0)) absh = 1.0 / rh;
    absh = fmax(absh, hmin0);
    const double tdel = (t + fmin(sqrt(eps) * fmax(fabs(t), fabs(t + absh)), absh)) - t;
    double f1, Jf0;
    if constexpr (ROWQ) {                                // (the evaluations at t first: one row at a time)
      Jf0 = eval(t, f0);
      if (!mb_take(Q, mb_post(Q, t + tdel, lane))) return 5;
      f1 = eval(t + tdel, y);
    } else {
      f1 = eval(t + tdel, y);
      Jf0 = eval(t, f0);                                 // J f0 = f(t, f0)
    }
    const double acc = Jf0 + (f1 - f0) / tdel;           // ddfddt (ev.cpp:261-283)
    rh = wave_max(1.25 * sqrt(0.5 * fabs(acc / wt) / rtol));
    absh = fmin(hmax, htspan);
    if (uni(absh * rh > 1.0)) absh = 1.0 / rh;
    absh = fmax(absh, hmin0);
    h = absh;
    dif[0] = h * f0;
  }
  PROF_STOP(3);
  int kk = 1, klast = 1, nconhk = 0;
  double iga = ndf_invGa(0), erc = ndf_erconst(0), errthr = rtol * fast_rcp(ndf_erconst(0));
  auto set_order = [&]() { iga = ndf_invGa(kk - 1); erc = ndf_erconst(kk - 1); errthr = rtol * fast_rcp(erc); };
  double abshlast = absh, hinvGak = h * iga, hmin = hmin0;
  double rate = 0., thr1 = minnrm, err = 0.;
  bool Jcurrent = true, havrate = false, done = false, at_hmin = false, need_fact = true, first_step = true;
  double tnew = t0, ynew = y, fnewton = 0., difkp1 = 0., invwt = 0.;

  for (;;) {   // ------------------------------------------------ one turn = one accepted step
    // ---------------------------------------------------------------- start of a step (ev.cpp:299-334)
    PROF_START();
    ISA_MARK("STEP_START");
    // (nothing below can change absh while it is what the previous step used and that step was not at the minimal step: the clamp and
    //  the test of the minimal step size are then skipped - every test here is a v_cmp_f64 feeding a scalar branch, ~45 cycles)
    hmin = P.min_var;
    bool moved = first_step || at_hmin || uni(absh != abshlast);
    first_step = false;
    if (moved) {
      absh = fmin(hmax, fmax(hmin, absh));
      if (uni(fabs(absh - hmin) < 100 * eps)) { if (at_hmin) absh = abshlast; at_hmin = true; } else at_hmin = false;
    }
    h = absh;
    if (uni(1.1 * absh >= fabs(tfinal - t))) { h = tfinal - t; absh = fabs(h); done = true; moved = true; }
    if (moved || (kk != klast)) {
      if (uni(fabs(absh - abshlast) > 1e-6 * absh) || (kk != klast)) {   // (ev.cpp:318: |dh| / h > 1e-6, without the division)
        adjust_stepsize(dif, absh * fast_rcp(abshlast), kk);
        hinvGak = h * iga;
        nconhk = 0;
        need_fact = true;
      }
    }
    // the time of this step is known: ask the helper for its table row now (a no-op when the step size did not change - the row
    // was requested a whole step ago - and otherwise early enough to arrive behind the factorisation)
    if constexpr (ROWQ) {
      if (have_next && !moved) { seq_this = seq_next; post_this = false; } else post_this = true;
      post_next = true;               // ... and the row of the step after this one, on the guess that the step size stays
    } else {
      request(done ? tfinal : t + h);
      if (!done) request((t + h) + absh);
    }
    bool nofailed = true;
    PROF_STOP(13);
    for (;;) {   // -------------------------------------------- attempts at this step
      if (--budget < 0) return 4;
      if constexpr (ROWQ) {
        if (post_this) seq_this = mb_post(Q, done ? tfinal : t + h, lane);
        if (post_next) { have_next = !done; if (!done) seq_next = mb_post(Q, (t + h) + absh, lane); }
        post_this = post_next = false;
      }
      if (need_fact) {
        need_fact = false;
        PROF_START();
        if (!factorise<SYS>(e, J, hinvGak, maxlen, lane, F, nullptr, 0., 0.)) return 2;
        if constexpr (USE_INV) {
          inv_want = -1; inv_ok = false; lu_first = true;
          if (!F.permuted) {
            if (lane < NC) C.mb->rp[lane] = F.rpivc;
            inv_want = ++C.fact_posted;
            mb_store(&C.mb->fact_seq, inv_want);
          }
        }
        PROF_STOP(2);
        st.lus++;
        havrate = false;
        thr1 = minnrm;
      }
      // -------------------------------------------------------------- predictor + simplified Newton (ev.cpp:342-445)
#ifdef CPT_PROFILE
      const unsigned long long t_newton0 = clock64();
      unsigned long long t_inner = 0;
#endif
      if constexpr (ROWQ) Q.ans_seen = mb_peek(&Q.mb->ans_seq);   // (looked at behind the predictor: mb_take)
      double psi, pred;
      ISA_MARK("NEWT_PRED");
      switch (__builtin_amdgcn_readfirstlane(kk)) {
        case 1: psi = dif[0]; pred = y + dif[0]; break;
        case 2: psi = fma(1.5, dif[1], dif[0]); pred = y + (dif[0] + dif[1]); break;
        case 3: psi = fma(11.0 / 6.0, dif[2], fma(1.5, dif[1], dif[0])); pred = y + ((dif[0] + dif[1]) + dif[2]); break;
        case 4: psi = fma(25.0 / 12.0, dif[3], fma(11.0 / 6.0, dif[2], fma(1.5, dif[1], dif[0]))); pred = y + ((dif[0] + dif[1]) + (dif[2] + dif[3])); break;
        default: psi = fma(137.0 / 60.0, dif[4], fma(25.0 / 12.0, dif[3], fma(11.0 / 6.0, dif[2], fma(1.5, dif[1], dif[0]))));
                 pred = y + (((dif[0] + dif[1]) + (dif[2] + dif[3])) + dif[4]); break;
      }
      psi *= iga;
      tnew = t + h;
      if (done) tnew = tfinal;
      h = tnew - t;
      ynew = pred;
      difkp1 = 0.;
      {  // weights of the norms (ev.cpp:367-374): the seed + one Newton step (2e-15) is ample for a weight
        const double w = fmax(fmax(fabs(ynew), fabs(y)), threshold);
        const double r = __builtin_amdgcn_rcp(w);
        invwt = fma(r, fma(-w, r, 1.0), r);
      }
      bool tooslow = false;
      double newnrm = 0., oldnrm = 0.;
#ifdef CPT_PROFILE
      const unsigned long long t_take0 = clock64();
#endif
      if constexpr (ROWQ) { if (!mb_take(Q, seq_this)) return 5; }
#ifdef CPT_PROFILE
      prof[8] += clock64() - t_take0;
      t_inner += clock64() - t_take0;
#endif
      for (int iter = 1; iter <= maxit; iter++) {
        PROF_START();
        ISA_MARK("NEWT_EVAL");
        fnewton = eval(tnew, ynew);
        PROF_STOP(0);
#ifdef CPT_PROFILE
        t_inner += clock64() - pf_t0;
#endif
        const double rhsv = hinvGak * fnewton - (psi + difkp1);
        PROF_START();
        ISA_MARK("NEWT_SOLVE");
        double del;
        if constexpr (USE_INV) {
          if (!lu_first && inv_want >= 0) {
            if (!inv_ok) {   // (once per factorisation; the helper has had a whole RHS evaluation and a solve to get there)
#ifdef CPT_PROFILE
              const unsigned long long t_wait0 = clock64();
#endif
              int spins = 0;
              while (mb_load(&C.mb->inv_seq) != inv_want) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1 << 24)) return 5;
              }
              inv_ok = true;
#ifdef CPT_PROFILE
              prof[15] += clock64() - t_wait0;
#endif
            }
            del = lu_solve<true>(e, F, maxlen, rhsv, lane);
          } else {
            del = lu_solve<false>(e, F, maxlen, rhsv, lane);
            lu_first = false;
          }
        } else del = lu_solve<false, SYS>(e, F, maxlen, rhsv, lane);
        ISA_MARK("NEWT_CTRL");
        PROF_STOP(1);
#ifdef CPT_PROFILE
        t_inner += clock64() - pf_t0;
#endif
        st.solves++;
        const double dn = fabs(del * invwt);
        difkp1 += del;
        ynew = pred + difkp1;
        if (iter == 1) {   // converged <=> max |del / wt| <= thr1 (see ndf15): one compare + one scalar test, no reduction
          if (wave_all_le(dn, thr1)) break;
          newnrm = wave_max(dn);
          if (!havrate) rate = 0.0;
        } else {
          // (ev.cpp:401-437.  The tests are evaluated together and decided by ONE scalar branch in the usual case - converged at this
          //  iteration; each `if (uni(...))` of its own would be a v_cmp_f64 feeding a branch, ~45 cycles of a lone wavefront)
          newnrm = wave_max(dn);
          const double rate_new = fmax(0.9 * rate, newnrm * fast_rcp(oldnrm));
          const double q = rate_new * fast_rcp(1.0 - rate_new), errit = newnrm * q;
          const bool c_min = newnrm <= minnrm, c_slow = newnrm > 0.9 * oldnrm, c_ok = errit <= 0.5 * rtol;
          const bool upd = !(c_min || c_slow);          // the rate estimate is only renewed past the first two tests
          rate = upd ? rate_new : rate;
          thr1 = upd ? fmax(minnrm, 0.05 * rtol * fast_rcp(q)) : thr1;
          if (uni(c_min || (!c_slow && c_ok))) { havrate = havrate || !uni(c_min); break; }
          if (uni(c_slow)) { tooslow = true; break; }
          havrate = true;
          if (iter == maxit) { tooslow = true; break; }
          else if (uni(0.5 * rtol < errit * fast_powi(rate, maxit - iter))) { tooslow = true; break; }
        }
        oldnrm = newnrm;
        ISA_MARK("NEWT_ITER_END");
      }
      ISA_MARK("NEWT_END");
#ifdef CPT_PROFILE
      prof[5] += clock64() - t_newton0 - t_inner;  // predictor + Newton control without rhs / solve
#endif
      if (tooslow) {  // ev.cpp:446-479
        st.failed++;
        if (!Jcurrent) {
          PROF_START();
          if constexpr (ROWQ) {   // (the rows of this attempt are dead; both are asked for again at the next attempt)
            if (!mb_take(Q, mb_post(Q, t, lane))) return 5;
            post_this = post_next = true;
          }
          jacobian(t);
          st.fevals++;  // the reference also re-evaluates f(t,y) here (ev.cpp:451)
          Jcurrent = true;
          need_fact = true;
          PROF_STOP(3);
          continue;
        }
        if (uni(absh <= hmin)) return 1;
        abshlast = absh;
        absh = fmax(0.3 * absh, hmin);
        h = absh;
        done = false;
        if constexpr (ROWQ) post_this = post_next = true;
        else { request(t + h); request((t + h) + absh); }
        adjust_stepsize(dif, absh * fast_rcp(abshlast), kk);
        hinvGak = h * iga;
        nconhk = 0;
        need_fact = true;
        continue;
      }
      // -------------------------------------------------------------- error test (ev.cpp:483-532)
      // err = max |difkp1 / wt| * erconst > rtol  <=>  some lane has |difkp1 / wt| > rtol / erconst: a compare, not a reduction
      PROF_START();
      const double dn = fabs(difkp1 * invwt);
      if (wave_all_le(dn, errthr)) { PROF_STOP(14); break; }   // accepted
      err = wave_max(dn) * erc;
      st.failed++;
      if (uni(absh <= hmin)) return 1;
      abshlast = absh;
      if (nofailed) {
        nofailed = false;
        double hopt = absh * fmax(0.1, 0.833 * fast_root(rtol * fast_rcp(err), kk + 1));
        if (kk > 1) {
          const double errkm1 = wave_max(fabs((dif_get(dif, kk - 1) + difkp1) * invwt)) * ndf_erconst(kk - 2);
          const double hkm1 = absh * fmax(0.1, 0.769 * fast_root(rtol * fast_rcp(errkm1), kk));
          if (uni(hkm1 > hopt)) { hopt = fmin(absh, hkm1); kk = kk - 1; set_order(); }
        }
        absh = fmax(hmin, hopt);
      } else absh = fmax(hmin, 0.5 * absh);
      h = absh;
      if (uni(absh < abshlast)) done = false;
      if constexpr (ROWQ) post_this = post_next = true;
      else { request(done ? tfinal : t + h); if (!done) request((t + h) + absh); }
      adjust_stepsize(dif, absh * fast_rcp(abshlast), kk);
      hinvGak = h * iga;
      nconhk = 0;
      need_fact = true;
      PROF_STOP(14);
    }
    // ------------------------------------------------------------------ accepted: update differences (ev.cpp:537-545)
    st.steps++;
    ISA_MARK("ACCEPTED");
    switch (__builtin_amdgcn_readfirstlane(kk)) {
      case 1: dif[2] = difkp1 - dif[1]; dif[1] = difkp1; dif[0] += dif[1]; break;
      case 2: dif[3] = difkp1 - dif[2]; dif[2] = difkp1; dif[1] += dif[2]; dif[0] += dif[1]; break;
      case 3: dif[4] = difkp1 - dif[3]; dif[3] = difkp1; dif[2] += dif[3]; dif[1] += dif[2]; dif[0] += dif[1]; break;
      case 4: dif[5] = difkp1 - dif[4]; dif[4] = difkp1; dif[3] += dif[4]; dif[2] += dif[3]; dif[1] += dif[2]; dif[0] += dif[1]; break;
      default: dif[6] = difkp1 - dif[5]; dif[5] = difkp1; dif[4] += dif[5]; dif[3] += dif[4]; dif[2] += dif[3]; dif[1] += dif[2]; dif[0] += dif[1]; break;
    }
    // ------------------------------------------------------------------ dense output at the sample times this step has passed
    // (ev.cpp:547-571, interp_from_dif :860-905), handed to the helper wave
    if (uni(tnew - tnext >= 0.0)) {
      PROF_START();
      const double tca_keep = M.tca_shear_g;
      const int flags = L.tca | (L.rsa << 1) | (L.ufa << 2) | (L.nfa << 3) | (L.fic << 4);   // (bits 3, 4: read by the helper of the register-set kernels)
      const double inv_h = fast_rcp(h);
      do {
        double yi, ypi;
        if (uni(tnew == tnext)) { yi = ynew; ypi = fnewton; }
        else {
          // y(tn) = ynew + sum_j p_j / j! dif_j, p_j = prod_{i <= j} (s + i), s = (tn - tnew) / h; the derivative takes dp_j / ds by the
          // product rule dp_j = dp_{j-1} (s + j) + p_{j-1} (= p_j sum_i 1 / (s + i), the form of ev.cpp:885-896, without divisions)
          const double sx = (tnext - tnew) * inv_h;
          double pj = sx, dpj = 1.0;
          yi = fma(pj, dif[0], ynew); ypi = dif[0];
#pragma unroll
          for (int j = 1; j < 5; j++) {
            if (j < kk) {
              const double inv_fact = (j == 1) ? 0.5 : (j == 2) ? 1.0 / 6.0 : (j == 3) ? 1.0 / 24.0 : 1.0 / 120.0;
              dpj = fma(dpj, sx + j, pj);
              pj *= (sx + j);
              yi = fma(pj * inv_fact, dif[j], yi);
              ypi = fma(dpj * inv_fact, dif[j], ypi);
            }
          }
          ypi *= inv_h;
        }
        if constexpr (!HELPED) {                    // no helper: perturb_sources right here
          (void)eval(tnext, yi);
          store_sources(P, L, Q, M, k, inv_k2, yi, ypi, tca_keep, next, ik, lane, N);
          M.tca_shear_g = tca_keep;
        } else {
        if (C.posted - C.tail_seen >= MB_NSLOT) {   // ring full: wait for the helper (it is certain to consume)
          int spins = 0;
          while (C.posted - (C.tail_seen = mb_load(&C.mb->tail)) >= MB_NSLOT) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1 << 24)) return 5;
          }
        }
        const int slot = C.posted & (MB_NSLOT - 1);
        C.mb->yi[slot][lane] = yi; C.mb->ypi[slot][lane] = ypi;
        if (lane == 0) { C.mb->tca_keep[slot] = tca_keep; C.mb->it[slot] = next; C.mb->flags[slot] = flags; }
        C.posted++;
        mb_store(&C.mb->head, C.posted);
        st.fevals++;                             // (the evaluation is counted where the reference makes it)
        }
        next++;
        tnext = tnext2;
        tnext2 = (next + 1 < tres) ? ts[next + 1] : 1e300;
      } while (uni(tnew - tnext >= 0.0));
      PROF_STOP(4);
    }
    if (done) break;
    // ------------------------------------------------------------------ after an accepted step (ev.cpp:573-635)
    PROF_START();
    ISA_MARK("POST_STEP");
    klast = kk;
    abshlast = absh;
    nconhk = min(nconhk + 1, maxk + 2);
    if (nconhk >= kk + 2) {
      // (the norm of the accepted correction is only formed here, where its value steers the step size: one step in ~five)
      err = wave_max(fabs(difkp1 * invwt)) * erc;
      double temp = 1.2 * fast_root(err * inv_rtol, kk + 1);
      // (selects, not branches: the operands are wave-uniform, a v_cmp_f64 feeding a scalar branch costs ~45 cycles)
      double hopt = (temp > 0.1) ? absh * fast_rcp(temp) : 10 * absh;
      int kopt = kk;
      if (kk > 1) {
        const double errkm1 = wave_max(fabs(dif_get(dif, kk - 1) * invwt)) * ndf_erconst(kk - 2);
        temp = 1.3 * fast_root(errkm1 * inv_rtol, kk);
        const double hkm1 = (temp > 0.1) ? absh * fast_rcp(temp) : 10 * absh;
        const bool better = uni(hkm1 > hopt);
        hopt = fmax(hopt, hkm1); kopt = better ? kk - 1 : kopt;
      }
      if (kk < maxk) {
        const double errkp1 = wave_max(fabs(dif_get(dif, kk + 1) * invwt)) * ndf_erconst(kk);
        temp = 1.4 * fast_root(errkp1 * inv_rtol, kk + 2);
        const double hkp1 = (temp > 0.1) ? absh * fast_rcp(temp) : 10 * absh;
        const bool better = uni(hkp1 > hopt);
        hopt = fmax(hopt, hkp1); kopt = better ? kk + 1 : kopt;
      }
      if (uni(hopt > absh)) { absh = hopt; if (kopt != kk) { kk = kopt; set_order(); } }
    }
    t = tnew;
    y = ynew;
    Jcurrent = false;
    PROF_STOP(12);
  }
  // ev.cpp:653-662: one last evaluation leaves M and Q describing (tfinal, y) for the hand-over to the next scheme
  (void)eval(tnew, ynew);
  if constexpr (ROWQ) mb_row_to_regs(Q);   // (the hand-over reads fields of the row)
  y_io = ynew;
  return 0;
}

// perturb_initial_conditions (pm.cpp:4723-5408): adiabatic, synchronous gauge, flat. Returns this lane's y.
static __device__ __noinline__ double initial_conditions(DevTables T, int has_cdm, int has_ur, double ci, double K, int ic, double ei, int gauge, int role,
                                                  int ell, double k, double tau) {
  // background row at tau (scalar lookup: executed once per mode)
  const int inf = bsearch_up(T.tau_table, T.bt_size, tau);
  const double hh = T.tau_table[inf + 1] - T.tau_table[inf], b = (tau - T.tau_table[inf]) / hh, aa = 1. - b, h2 = hh * hh / 6.;
  const double2* r0 = (const double2*)T.bg + (size_t)inf * BG_NCOL;
  const double2* r1 = r0 + BG_NCOL;
  const double a = spl2(r0[BG_A], r1[BG_A], aa, b, h2), rg = spl2(r0[BG_RHO_G], r1[BG_RHO_G], aa, b, h2),
               rb = spl2(r0[BG_RHO_B], r1[BG_RHO_B], aa, b, h2), rc = spl2(r0[BG_RHO_CDM], r1[BG_RHO_CDM], aa, b, h2),
               ru = spl2(r0[BG_RHO_UR], r1[BG_RHO_UR], aa, b, h2);
  double rho_r = rg, rho_m = rb, rho_nu = 0.;
  if (has_cdm) rho_m += rc;
  if (has_ur) { rho_r += ru; rho_nu += ru; }
  if (NCDM) {   // pm.cpp:4794-4799: the non-cold species count as relativistic relics at the initial time
    const double2* n0 = (const double2*)T.ncb + (size_t)inf * NCB_NCOL;
    for (int n = 0; n < CPT_MAX_NCDM; n++) { const double rn = spl2(n0[3 * n], n0[NCB_NCOL + 3 * n], aa, b, h2); rho_r += rn; rho_nu += rn; }
  }
  const double fracnu = rho_nu / rho_r, fracb = rb / rho_m;
  const double om = a * rho_m / sqrt(rho_r);
  const double kt2 = k * k * tau * tau, kt3 = k * tau * kt2;
  const double s2sq = 1. - 3. * K / (k * k);   // pm.cpp:4838: the curvature factors of the super-horizon series
  const double delta_g = -kt2 / 3. * (1. - om * tau / 5.) * ci * s2sq;
  const double theta_g = -k * kt3 / 36. * (1. - 3. * (1. + 5. * fracb - fracnu) / 20. / (1. - fracnu) * om * tau) * ci * s2sq;
  const double theta_ur = -k * kt3 / 36. / (4. * fracnu + 15.) *
                          (4. * fracnu + 11. + 12. * s2sq - 3. * (8. * fracnu * fracnu + 50. * fracnu + 275.) / 20. / (2. * fracnu + 15.) * tau * om) * ci * s2sq;
  const double shear_ur = kt2 / (45. + 12. * fracnu) * (3. * s2sq - 1.) * (1. + (4. * fracnu - 5.) / 4. / (2. * fracnu + 15.) * tau * om) * ci;
  const double l3_ur = kt3 * 2. / 7. / (12. * fracnu + 45.) * ci;
  const double eta = ci * (1. - kt2 / 12. / (15. + 4. * fracnu) *
                                    (5. + 4. * s2sq * fracnu - (16. * fracnu * fracnu + 280. * fracnu + 325) / 10. / (2. * fracnu + 15.) * tau * om));
  // synchronous-gauge values of the named variables (cdm velocity is zero by definition of that gauge)
  double dg, tg, db, dcdm = 0., dur, tur, sur, l3u = 0., et;
  if (ic == CPT_IC_AD) {
    dg = delta_g; tg = theta_g; db = 0.75 * delta_g; dcdm = 0.75 * delta_g; dur = delta_g; tur = theta_ur; sur = shear_ur; l3u = l3_ur; et = eta;
  } else {
    // isocurvature modes (pm.cpp:4956-5083; Bucher, Moodley & Turok 1999 with CLASS normalisation); l3_ur = 0
    const double fracg = rg / rho_r, fraccdm = 1. - fracb, kt = k * tau;
    if (ic == CPT_IC_CDI || ic == CPT_IC_BI) {
      const double f = (ic == CPT_IC_CDI) ? fraccdm : fracb;
      dg = ei * f * om * tau * (-2. / 3. + om * tau / 4.);
      tg = -ei * f * om * kt2 / 12.;
      db = 0.75 * dg + (ic == CPT_IC_BI ? ei : 0.);
      dcdm = 0.75 * dg + (ic == CPT_IC_CDI ? ei : 0.);
      dur = dg; tur = tg;
      sur = -ei * f * kt2 * tau * om / 6. / (2. * fracnu + 15.);
      et = -ei * f * om * tau * (1. / 6. - om * tau / 16.);
    } else if (ic == CPT_IC_NID) {
      dg = ei * fracnu / fracg * (-1. + kt2 / 6.);
      tg = -ei * fracnu / fracg * k * k * tau * (1. / 4. - fracb / fracg * 3. / 16. * om * tau);
      db = ei * fracnu / fracg / 8. * kt2;
      dcdm = -ei * fracnu * fracb / fracg / 80. * kt2 * om * tau;
      dur = ei * (1. - kt2 / 6.);
      tur = ei * k * k * tau / 4.;
      sur = ei * kt2 / (4. * fracnu + 15.) / 2.;
      et = -ei * fracnu / (4. * fracnu + 15.) / 6. * kt2;
    } else {  // CPT_IC_NIV
      dg = ei * kt * fracnu / fracg * (1. - 3. / 16. * fracb * (2. + fracg) / fracg * om * tau);
      tg = ei * fracnu / fracg * 3. / 4. * k *
           (-1. + 3. / 4. * fracb / fracg * om * tau + 3. / 16. * om * om * tau * tau * fracb / fracg / fracg * (fracg - 3. * fracb) + kt2 / 6.);
      db = 0.75 * dg;
      dcdm = -ei * 9. / 64. * fracnu * fracb / fracg * kt * om * tau;
      dur = -ei * kt * (1. + 3. / 16. * fracb * fracnu / fracg * om * tau);
      tur = ei * 3. / 4. * k * (1. - 1. / 6. * kt2 * (4. * fracnu + 9.) / (4. * fracnu + 5.));
      sur = ei / (4. * fracnu + 15.) * kt * (1. + 3. * om * tau * fracnu / (4. * fracnu + 15.));
      et = ei * fracnu * kt * (-1. / (4. * fracnu + 5.) + (-3. / 64. * fracb / fracg + 15. / 4. / (4. * fracnu + 15.) / (4. * fracnu + 5.) * om * tau));
    }
  }
  double tbv = tg, tcdm = 0.;
  if (!has_cdm) dcdm = 0.;
  if (!has_ur && !NCDM) { dur = tur = sur = l3u = 0.; }   // (NCDM: the chain lanes ask for the relic series, pm.cpp:5229-5256)
  if (gauge == CPT_GAUGE_NEWTONIAN) {  // gauge transformation of the synchronous series, pm.cpp:5095-5198
    const double H = spl2(r0[BG_H], r1[BG_H], aa, b, h2), aH = a * H, fracg = rg / rho_r, fraccdm = 1. - fracb, rmr = rho_m / rho_r;
    const double delta_tot = (fracg * dg + fracnu * dur + rmr * (fracb * db + fraccdm * dcdm)) / (1. + rmr);
    const double velocity_tot = ((4. / 3.) * (fracg * tg + fracnu * tur) + rmr * fracb * tbv) / (1. + rmr);
    const double alpha = (et + 1.5 * aH * aH / (k * k) / s2sq * (delta_tot + 3. * aH / (k * k) * velocity_tot)) / aH;
    et -= aH * alpha;   // phi
    dg -= 4. * aH * alpha; tg += k * k * alpha;
    db -= 3. * aH * alpha; tbv += k * k * alpha;
    if (has_cdm) { dcdm -= 3. * aH * alpha; tcdm = k * k * alpha; }
    if (has_ur) { dur -= 4. * aH * alpha; tur += k * k * alpha; }
  }
  switch (role) {
    case R_DELTA_G: return dg;
    case R_THETA_G: return tg;
    case R_DELTA_B: return db;
    case R_THETA_B: return tbv;
    case R_DELTA_CDM: return dcdm;
    case R_THETA_CDM: return tcdm;
    case R_DELTA_UR: return dur;
    case R_THETA_UR: return tur;
    case R_SHEAR_UR: return sur;
    case R_LUR: return ell == 3 ? l3u : 0.;
    case R_ETA: return et;
    default: return 0.;
  }
}

// the integration of one mode over its intervals of constant approximation scheme (the integrator wave of the two-wave kernels)
struct Sched { double tau_ini, tau_end, sw0, sw1, sw2, sw3; int nsw, ap0, ap1, ap2, ap3, fi0, fi1, fi2, fi3; };
static __device__ __forceinline__ int run_intervals(const PtParams& P, Ctx& C, const Sched& sc, double k, double inv_k2, int ik, int lane,
                                                    double2* bgw, double2* thw, double* jacw, double2* fww, Stat& st, int& n_regimes,
                                                    int& budget, unsigned long long* prof
#ifdef CPT_PROFILE
                                                    , unsigned long long t_begin
#endif
                                                    ) {
  static_assert(NCDM == 0, "the register-set kernels have their own driver (cpt_perturb_sets.inc)");
  int status = 0;
  const double tau_ini = sc.tau_ini, tau_end = sc.tau_end, sw0 = sc.sw0, sw1 = sc.sw1, sw2 = sc.sw2, sw3 = sc.sw3;
  const int nsw = sc.nsw, ap0 = sc.ap0, ap1 = sc.ap1, ap2 = sc.ap2, ap3 = sc.ap3, fi0 = sc.fi0, fi1 = sc.fi1, fi2 = sc.fi2;
  {
    Lookup Q;
    // (the integrator wave owns no table windows - its rows come from the helper wave through the mailbox; plain stores keep the
    //  struct in registers)
    Q.bgw = bgw; Q.thw = thw; Q.ncw = nullptr; Q.tau_cached = -1.; Q.bg_base = Q.th_base = 0; Q.bg_inf = Q.th_inf = -1;
    Q.bgx = Q.thx = Q.vbg = Q.vth = Q.vnc = 0.; Q.zmax = Q.xe_last = Q.taud_last = 0.;
    Q.bg_lo = Q.bg_hi = Q.th_lo = Q.th_hi = Q.nc_lo = Q.nc_hi = make_double2(0., 0.);
    Q.rg = Q.rb = Q.rc = Q.ru = Q.kap = Q.ddkappa = Q.cb2 = Q.a2 = Q.aH = Q.two_over_aH = Q.R = Q.inv_1pR = Q.inv_R = 0.;
    Q.tau_c = Q.dtau_c = Q.F = Q.Fp = Q.app = Q.inv_tau = Q.rg43 = Q.ru43 = Q.kcot = 0.;
    Q.mb = C.mb; Q.my_req = 0; Q.rq_tau0 = Q.rq_tau1 = -1.; Q.ans_seen = 0; Q.row = nullptr; Q.row_tau = -1.;
    lookup_set_mode(P, Q, k);
#ifdef CPT_PROFILE
    Q.prof = prof;
#endif
    Metric M;
    M.hp = M.etap = M.alpha = M.alphap = 0.;
    M.tca_shear_g = 0.; M.rsa_dg = M.rsa_tg = 0.;
    int f_tca = fi0, f_rsa = fi1, f_ufa = fi2;
    Layout L = make_layout(P, f_tca, f_rsa, f_ufa);
    LaneEq e = make_lane_eq(P, L, lane, k);
    double y;
    if (MODE) {  // tensors (pm.cpp:5386-5403): only the gravitational wave starts non-zero
      y = 0.;
      if (e.role == R_GW) {
        const double k2 = k * k;
        y = P.gw_ini / 2.449489742783178;
        if (CURV) {
          y *= sqrt(k2 * (k2 - P.K) / (k2 + 3. * P.K) / (k2 + 2. * P.K));
          if (P.K < 0.) y = (k2 + 3. * P.K >= 0.) ? y * sqrt(tanh(1.5707963267948966 * sqrt(k2 + 3. * P.K) / sqrt(-P.K))) : 0.;
        }
      }
    } else
      y = initial_conditions(P.tabs, P.has_cdm, P.has_ur, P.curvature_ini, P.K, P.ic, P.entropy_ini, GAUGE, e.role, e.ell, k, tau_ini);
#ifdef CPT_PROFILE
    prof[6] = clock64() - t_begin;  // schedule search + initial conditions
#endif
    for (int iv = 0; iv <= nsw && status == 0; iv++) {
      const double ta = (iv == 0) ? tau_ini : (iv == 1) ? sw0 : (iv == 2) ? sw1 : (iv == 3) ? sw2 : sw3;
      const double tb = (iv == nsw) ? tau_end : (iv == 0) ? sw0 : (iv == 1) ? sw1 : (iv == 2) ? sw2 : sw3;
      if (iv > 0) {
        // hand-over to the new scheme (pm.cpp:3777-4260): every variable keeps its lane; the ones the new scheme
        // drops are zeroed, the ones it adds are seeded
        const int was_tca = L.tca;
        const int ap = (iv == 1) ? ap0 : (iv == 2) ? ap1 : (iv == 3) ? ap2 : ap3;
        if (ap == 0) f_tca ^= 1; else if (ap == 1) f_rsa ^= 1; else f_ufa ^= 1;
        L = make_layout(P, f_tca, f_rsa, f_ufa);
        e = make_lane_eq(P, L, lane, k);
        double yn = (e.role == R_NONE) ? 0. : y;
        if (MODE) {  // tensors (pm.cpp:4640-4648): photons re-enter with delta_g = -4/3 gw'/kappa', pol0 = gw'/(3 kappa')
          if (was_tca && !L.tca) {
            const double gwd = bcast(y, TL_GWD);
            if (e.role == R_DELTA_G) yn = -4. / 3. * gwd * Q.tau_c;
            else if (e.role == R_POL && e.ell == 0) yn = 1. / 3. * gwd * Q.tau_c;
            else if (lane <= TL_P4 || e.chain == 1 || e.chain == 2) yn = 0.;
          }
        } else
        if (was_tca && !L.tca) {  // tight coupling switched off: seed shear, l=3 and polarisation (pm.cpp:3893-3916)
          const double sh = M.tca_shear_g, kod = k * Q.tau_c;
          if (e.role == R_SHEAR_G) yn = sh;
          const double s3 = CURV ? sqrt(fmax(1. - 8. * P.K / (k * k), 0.)) : 1.;
          if (e.role == R_LG) yn = (e.ell == 3) ? 6. / 7. * kod * s3 * sh : 0.;
          if (e.role == R_POL) {
            if (e.ell == 0) yn = 2.5 * sh;
            else if (e.ell == 1) yn = kod * (5. - 2. * Q.s2) / 6. * sh;
            else if (e.ell == 2) yn = 0.5 * sh;
            else if (e.ell == 3) yn = kod * 3. * s3 / 14. * sh;
            else yn = 0.;
          }
        }
        y = yn;
      }
      n_regimes++;
#ifdef CPT_PROFILE_INTERVALS
      const unsigned long long iv_t0 = clock64(); const int iv_s0 = st.steps;
#endif
      const int rc = ndf15s<0>(P, L, e, C, Q, M, k, inv_k2, ta, tb, y, st, lane, budget, jacw, fww, prof);
      if (rc) status = 10 + rc;
#ifdef CPT_PROFILE_INTERVALS
      if (lane == 0 && blockIdx.x == 0)
        printf("interval %d [%g, %g] tca %d rsa %d ufa %d: %d steps, %llu cycles\n", iv, ta, tb, L.tca, L.rsa, L.ufa, st.steps - iv_s0, clock64() - iv_t0);
#endif
    }
  }
  return status;
}

// the helper wave (see Mailbox): answers the integrator's table look-ups and evaluates perturb_sources (pm.cpp:6731-7285) for every
// sample the integrator posts.  Two sets of table windows: the samples walk monotonically through the sample times, the look-ups run
// one step ahead of the integration.
static __device__ __forceinline__ void run_helper(const PtParams& P, Mailbox* mb, double k, double inv_k2, int ik, int lane, double2* bgw_s, double2* thw_s,
                                                  double2* bgw_p, double2* thw_p, const double2* fw, double2* inv) {
  static_assert(NCDM == 0, "the register-set kernels have their own helper (cpt_perturb_sets.inc)");
  // (rows of the tail lanes and the padding column of the inverse: zero for good)
  for (int q = 0; q < FW_ACP; q++) inv[q * JS + lane] = make_double2(0., 0.);
  int inverted = 0;
  Lookup Q, Qp;
  lookup_init(P, Q, bgw_s, thw_s, lane);
  lookup_set_mode(P, Q, k);
  lookup_init(P, Qp, bgw_p, thw_p, lane);
  lookup_set_mode(P, Qp, k);
#ifdef CPT_PROFILE
  unsigned long long sprof[16];
  Q.prof = sprof; Qp.prof = sprof;
#endif
  Metric M;
  M.hp = M.etap = M.alpha = M.alphap = 0.; M.psi = M.phip = 0.;
  M.tca_shear_g = 0.; M.rsa_dg = M.rsa_tg = 0.;
  int flags = -1, tail = 0, answered = 0;
  Layout L = make_layout(P, 1, 0, 0);
  LaneEq e = make_lane_eq(P, L, lane, k);
#ifdef CPT_PROFILE
  unsigned long long hprof[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long th_begin = clock64();
#endif
  for (;;) {
    // look-ups first: the integrator may be waiting for one, a sample never holds it up while the ring has room
    const MbPoll pc = mb_poll(mb);
    if (pc.req_seq != answered) {
#ifdef CPT_PROFILE
      const unsigned long long th0 = clock64();
#endif
      const int rq = answered + 1;               // (in order: two requests may be waiting)
      const double tau = mb->req_tau[rq & 1];
      lookup(P, Qp, tau, lane);
      if (lane == 0) {
        double* a = mb->ans[rq & 1];
        a[0] = Qp.rg; a[1] = Qp.rb; a[2] = Qp.rc; a[3] = Qp.ru; a[4] = Qp.kap; a[5] = Qp.ddkappa; a[6] = Qp.cb2; a[7] = Qp.a2;
        a[8] = Qp.aH; a[9] = Qp.two_over_aH; a[10] = Qp.R; a[11] = Qp.inv_1pR; a[12] = Qp.inv_R; a[13] = Qp.tau_c; a[14] = Qp.dtau_c; a[15] = Qp.F;
        a[16] = Qp.Fp; a[17] = Qp.app; a[18] = Qp.inv_tau; a[19] = Qp.rg43; a[20] = Qp.ru43; a[21] = Qp.kcot;
#ifdef CPT_CHECK_ROWS
        a[31] = tau;
#endif
      }
      answered = rq;
      mb_store(&mb->ans_seq, rq);
#ifdef CPT_PROFILE
      hprof[0] += clock64() - th0; hprof[1]++;
#endif
      continue;
    }
    // then the inverse of a factorisation the integrator has posted (it will want it at its next Newton iteration but one)
    const int fs = pc.fact_seq;
    if (fs != inverted) {
#ifdef CPT_PROFILE
      const unsigned long long th0 = clock64();
#endif
      helper_inverse(fw, mb->rp, inv, lane);
      inverted = fs;
      mb_store(&mb->inv_seq, fs);
#ifdef CPT_PROFILE
      hprof[2] += clock64() - th0; hprof[3]++;
#endif
      continue;
    }
    const int head = pc.head;
    if (tail != head) {
#ifdef CPT_PROFILE
      const unsigned long long th0 = clock64();
#endif
      const int slot = tail & (MB_NSLOT - 1);
      const int f = mb->flags[slot], it = mb->it[slot];
      const double yi = mb->yi[slot][lane], ypi = mb->ypi[slot][lane], tca_keep = mb->tca_keep[slot];
      tail++;
      mb_store(&mb->tail, tail);                 // (the slot's content is in registers: the integrator may refill it)
      const double tn = P.tau_s[it];
      if (f != flags) {                          // the integrator entered another approximation scheme
        flags = f;
        L = make_layout(P, f & 1, (f >> 1) & 1, (f >> 2) & 1);
        e = make_lane_eq(P, L, lane, k);
      }
      (void)rhs<false>(P, L, e, Q, M, k, inv_k2, tn, yi, lane);   // leaves Q and M describing (tn, yi)
      store_sources(P, L, Q, M, k, inv_k2, yi, ypi, tca_keep, it, ik, lane);
#ifdef CPT_PROFILE
      hprof[4] += clock64() - th0; hprof[5]++;
#endif
      continue;
    }
    if (pc.done) {
      if (mb_load(&mb->head) == tail) break;     // finished and drained
      continue;
    }
#ifdef CPT_PROFILE
    hprof[6]++;
#endif
    __builtin_amdgcn_s_sleep(1);
  }
#ifdef CPT_PROFILE
  hprof[7] = clock64() - th_begin;
  if (lane == 0 && blockIdx.x == 0) for (int i = 0; i < 16; i++) g_prof_helper[i] = hprof[i];
#endif
}

// start of the integration (pm.cpp:2545-2635) and the schedule of approximation switches (pm.cpp:2940-3231) of one k-mode; every lane of the
// wave takes part in the searches, the result is wave-uniform.  Returns 0 or the status code of the mode (20 ... 23).
static __device__ __forceinline__ int make_schedule(const PtParams& P, double k, int lane, Sched& sc) {
  int status = 0;
  const double tau_end = P.tau_s[P.ntau - 1];
  // ---- start of integration: pm.cpp:2545-2635 ----
  double tau_ini;
  {
    const double tl = P.tabs.tau_table[0];
    const AHK q = lookup_aHk(P.tabs, P.n_e, tl);
    if ((q.a * q.H / q.dk > P.start_small_k) || (k / q.a / q.H > P.start_large_k) || (q.wdev > P.tol_ncdm_w)) status = 20;
    tau_ini = first(search_flip(P, k, tl, P.tau_s[0], 0., P.tol_tau_approx, 0, 0, lane));
  }
  // ---- regime schedule: pm.cpp:2940-3231 ----
  int fi0, fi1, fi2, fi3, fe0, fe1, fe2, fe3;
  approx_flags(P, k, tau_ini, &fi0, &fi1, &fi2, &fi3);
  approx_flags(P, k, tau_end, &fe0, &fe1, &fe2, &fe3);
  fi0 = ufirst(fi0); fi1 = ufirst(fi1); fi2 = ufirst(fi2); fi3 = ufirst(fi3);
  fe0 = ufirst(fe0); fe1 = ufirst(fe1); fe2 = ufirst(fe2); fe3 = ufirst(fe3);
  // (unused slots stay at +huge so that the sorting network below leaves them last)
  double sw0 = 1e300, sw1 = 1e300, sw2 = 1e300, sw3 = 1e300;
  int ap0 = 0, ap1 = 0, ap2 = 0, ap3 = 0, nsw = 0;
#pragma unroll 1
  for (int ap = 0; ap < (NCDM ? 4 : 3); ap++) {
    const int fi = (ap == 0) ? fi0 : (ap == 1) ? fi1 : (ap == 2) ? fi2 : fi3, fe = (ap == 0) ? fe0 : (ap == 1) ? fe1 : (ap == 2) ? fe2 : fe3;
    if (fi == fe) continue;
    const bool fwd = (ap == 0) ? (fi == 1 && fe == 0) : (fi == 0 && fe == 1);  // tca: on->off, rsa/ufa/ncdmfa: off->on
    if (!fwd) { status = 21; continue; }
    const double tsw = first(search_flip(P, k, tau_ini, tau_end, P.tol_tau_approx, 0., ap + 1, fi, lane));
    if (nsw == 0) { sw0 = tsw; ap0 = ap; } else if (nsw == 1) { sw1 = tsw; ap1 = ap; } else if (nsw == 2) { sw2 = tsw; ap2 = ap; } else { sw3 = tsw; ap3 = ap; }
    nsw++;
  }
  // sort the (at most 4) switches chronologically (scalars only: no private arrays)
  {
    auto cswap = [](double& x, double& y, int& a, int& b) { if (y < x) { const double t = x; x = y; y = t; const int u = a; a = b; b = u; } };
    cswap(sw0, sw1, ap0, ap1); cswap(sw2, sw3, ap2, ap3); cswap(sw0, sw2, ap0, ap2); cswap(sw1, sw3, ap1, ap3); cswap(sw1, sw2, ap1, ap2);
  }
  if ((nsw >= 2 && sw1 == sw0) || (nsw >= 3 && sw2 == sw1) || (nsw == 4 && sw3 == sw2)) status = 22;
  if (!(fi0 == 1 && fi1 == 0 && fi2 == 0 && fi3 == 0)) status = 23;  // pm.cpp:3720-3745

  sc = Sched{tau_ini, tau_end, sw0, sw1, sw2, sw3, nsw, ap0, ap1, ap2, ap3, fi0, fi1, fi2, fi3};
  return status;
}

// ---- the kernel: perturb_solve (pm.cpp:2463-2787) for one mode per workgroup of two wavefronts (integrator + helper) -------------
static __device__ __forceinline__ void body_perturb(const PtParams& P) {
  static_assert(NCDM == 0, "the register-set kernels have their own body (cpt_perturb_sets.inc)");
  __shared__ __attribute__((aligned(16))) double2 tabw[64 * (BG_NCOL + TH_NCOL)];      // the helper's windows of the look-ups ahead
  __shared__ __attribute__((aligned(16))) double2 tabw2[64 * (BG_NCOL + TH_NCOL)];     // ... and of the samples
  __shared__ double jacw[NC * JS];
  __shared__ __attribute__((aligned(16))) double2 fwsh[(FW_PAIRS + FW_ACP) * JS];   // the integrator's factors (LuReg::fw), then the helper's inverse of the core block (LuReg::inv)
  __shared__ __attribute__((aligned(16))) unsigned char mbox_s[sizeof(Mailbox)];
  Mailbox* mbox = (Mailbox*)mbox_s;
  const int lane = threadIdx.x & 63;
  const int wave = (int)(threadIdx.x >> 6);
  const int ik = P.order[blockIdx.x];
  const double k = P.k[ik];
  const double inv_k2 = 1.0 / (k * k);
  double2* bgw = tabw;
  double2* thw = tabw + 64 * BG_NCOL;
  Ctx C;
  C.mb = mbox; C.posted = 0; C.tail_seen = 0; C.fact_posted = 0; C.rowbuf = nullptr; C.cwbuf = nullptr;
  // (the only barrier of the kernel: the counters are zero before any wave looks at them)
  if (threadIdx.x == 0) { mbox->head = mbox->tail = mbox->done = mbox->req_seq = mbox->ans_seq = mbox->fact_seq = mbox->inv_seq = 0; mbox->req_tau[0] = mbox->req_tau[1] = -1.; }
  __syncthreads();
  if (wave == 1) {   // the helper polls `done` whatever happens to the integrator, drains the ring and leaves
    run_helper(P, mbox, k, inv_k2, ik, lane, tabw2, tabw2 + 64 * BG_NCOL, bgw, thw, fwsh, fwsh + FW_PAIRS * JS);
    return;
  }
  Stat st = {0, 0, 0, 0, 0, 0};
  unsigned long long prof[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#ifdef CPT_PROFILE
  const unsigned long long t_begin = clock64();
#endif
  int n_regimes = 0;
  int budget = P.max_steps;
  Sched sc;
  int status = make_schedule(P, k, lane, sc);
  if (status == 0) {
#ifdef CPT_PROFILE
    status = run_intervals(P, C, sc, k, inv_k2, ik, lane, bgw, thw, jacw, fwsh, st, n_regimes, budget, prof, t_begin);
#else
    status = run_intervals(P, C, sc, k, inv_k2, ik, lane, bgw, thw, jacw, fwsh, st, n_regimes, budget, prof);
#endif
  }
  mb_store(&mbox->done, 1);   // every path of the integrator ends here
#ifdef CPT_PROFILE
  prof[7] = clock64() - t_begin;
  if (lane == 0 && blockIdx.x == 0)  // the heaviest mode = the critical path
    for (int i = 0; i < 16; i++) g_prof[i] = prof[i];
#endif
  if (lane == 0) {
    if (P.status) P.status[ik] = status;
    if (P.stats) {
      cpt_stepstat s;
      s.steps = st.steps; s.failed = st.failed; s.fevals = st.fevals; s.jacobians = st.jacs; s.factorisations = st.lus;
      s.solves = st.solves; s.n_regimes = n_regimes; s.tau_ini = sc.tau_ini;
      P.stats[ik] = s;
    }
  }
}

// ---- unit-test kernels -----------------------------------------------------------------------------
static __device__ __forceinline__ void body_dbg_lookup(const PtParams& P, const double* tau, int n, double* out) {
  __shared__ __attribute__((aligned(16))) double2 w[64 * (BG_NCOL + TH_NCOL)];
  const int lane = threadIdx.x;
  Lookup Q;
  lookup_init(P, Q, w, w + 64 * BG_NCOL, lane);
#ifdef CPT_PROFILE
  unsigned long long dbg_prof[16];
  Q.prof = dbg_prof;
#endif
  lookup_set_mode(P, Q, 1.0);
  for (int i = 0; i < n; i++) {
    lookup(P, Q, tau[i], lane);
    if (lane == 0) {
      double* o = out + (size_t)i * 16;
      o[0] = bcast(Q.vbg, BG_A); o[1] = bcast(Q.vbg, BG_H); o[2] = bcast(Q.vbg, BG_HP); o[3] = Q.rg; o[4] = Q.rb; o[5] = Q.rc; o[6] = Q.ru;
      o[7] = bcast(Q.vth, TH_XE); o[8] = Q.kap; o[9] = bcast(Q.vth, TH_TAU_D); o[10] = Q.ddkappa; o[11] = bcast(Q.vth, TH_DDDKAPPA);
      o[12] = bcast(Q.vth, TH_EXPMK); o[13] = bcast(Q.vth, TH_G); o[14] = bcast(Q.vth, TH_DG); o[15] = Q.cb2;
    }
  }
}

// y and dy are exchanged in the REFERENCE's ordering of the regime (pm.cpp:3302-3481)
static __device__ __forceinline__ void body_dbg_derivs(const PtParams& P, double k, double tau, int tca, int rsa, int ufa, const double* y,
                                                   double* dy, int* neq) {
  __shared__ __attribute__((aligned(16))) double2 w[64 * (BG_NCOL + TH_NCOL)];
  const int lane = threadIdx.x;
  Lookup Q;
  lookup_init(P, Q, w, w + 64 * BG_NCOL, lane);
#ifdef CPT_PROFILE
  unsigned long long dbg_prof[16];
  Q.prof = dbg_prof;
#endif
  Metric M;
  M.hp = M.etap = M.alpha = M.alphap = 0.;
  M.tca_shear_g = 0.; M.rsa_dg = M.rsa_tg = 0.;
  lookup_set_mode(P, Q, k);
  Layout L = make_layout(P, tca, rsa, ufa);
  LaneEq e = make_lane_eq(P, L, lane, k);
  int nref;
  const int ri = ref_index_of(P, tca, rsa, ufa, e.role, e.ell, &nref);
  const double yl = (ri >= 0) ? y[ri] : 0.;
  const double d = rhs(P, L, e, Q, M, k, 1.0 / (k * k), tau, yl, lane);
  if (ri >= 0) dy[ri] = d;
  if (lane == 0) *neq = nref;
}

// (I - hg J(tau)) x = b through the structured factorisation, in the reference's ordering: unit test of the linear algebra
static __device__ __forceinline__ void body_dbg_solve(const PtParams& P, double k, double tau, int tca, int rsa, int ufa, double hg,
                                                  const double* b, double* x) {
  __shared__ __attribute__((aligned(16))) double2 w[64 * (BG_NCOL + TH_NCOL)];
  const int lane = threadIdx.x;
  Lookup Q;
  lookup_init(P, Q, w, w + 64 * BG_NCOL, lane);
#ifdef CPT_PROFILE
  unsigned long long dbg_prof[16];
  Q.prof = dbg_prof;
#endif
  Metric M;
  M.hp = M.etap = M.alpha = M.alphap = 0.;
  M.tca_shear_g = 0.; M.rsa_dg = M.rsa_tg = 0.;
  lookup_set_mode(P, Q, k);
  Layout L = make_layout(P, tca, rsa, ufa);
  LaneEq e = make_lane_eq(P, L, lane, k);
  const double inv_k2 = 1.0 / (k * k);
  __shared__ double jacw[NC * JS];
  Jac J;
  J.Jc = jacw;
  for (int j = 0; j < NC; j++) jc_store(J.Jc, j, lane, 0.);
  for (int r = 0; r < NC; r++) {
    if (!core_present(P, L, r)) continue;
    const double col = rhs(P, L, e, Q, M, k, inv_k2, tau, (lane == r) ? 1.0 : 0.0, lane);
    jc_store(J.Jc, r, lane, (lane < NC) ? col : 0.);
  }
  J.jdiag = -(e.D * Q.kap + e.G * Q.kcot + (CURV ? e.Gt * Q.inv_tau : 0.));
  __shared__ __attribute__((aligned(16))) double2 fwsh[(FW_PAIRS + FW_ACP) * JS];
  __shared__ double rps[32];
  LuReg F;
  F.fw = fwsh;
  F.inv = fwsh + FW_PAIRS * JS;
  const bool ok = factorise(e, J, hg, L.maxlen, lane, F);
  int nref;
  const int ri = ref_index_of(P, tca, rsa, ufa, e.role, e.ell, &nref);
  const double bl = (ri >= 0) ? b[ri] : 0.;
  double xl;
  if (INV && P.dbg_inverse && !F.permuted) {   // the product form of the core solve (helper_inverse), here by the one wave of the test kernel
    for (int q = 0; q < FW_ACP; q++) fwsh[(FW_PAIRS + q) * JS + lane] = make_double2(0., 0.);
    if (lane < NC) rps[lane] = F.rpivc;
    __syncthreads();
    helper_inverse(fwsh, rps, fwsh + FW_PAIRS * JS, lane);
    __syncthreads();
    xl = lu_solve<true>(e, F, L.maxlen, bl, lane);
    if (lane == 0) x[63] = 1.0;                // (tells the test that this form ran)
  } else xl = lu_solve(e, F, L.maxlen, bl, lane);
  if (ri >= 0) x[ri] = ok ? xl : nan("");
}

};  // struct PT<GAUGE>

static void fill_params(const cpt_handle* h, PtParams& P) {
  const cpt_config& c = h->cfg;
  P.tabs = h->tabs;
  P.has_cdm = c.has_cdm; P.has_ur = c.has_ur; P.tca_method = c.tight_coupling_approximation;
  P.rsa_method = c.radiation_streaming_approximation; P.ufa_method = c.ur_fluid_approximation;
  P.l_max_g = c.l_max_g; P.l_max_pol_g = c.l_max_pol_g; P.l_max_ur = c.l_max_ur;
  P.l_max_g_ten = c.l_max_g_ten; P.l_max_pol_g_ten = c.l_max_pol_g_ten; P.evolve_tensor_ur = c.evolve_tensor_ur; P.gw_ini = c.gw_ini;
  P.K = c.K; P.gauge = c.gauge; P.ic = c.ic; P.entropy_ini = c.entropy_ini; P.T_cmb = c.T_cmb; P.a_today = c.a_today; P.YHe = c.YHe; P.n_e = c.n_e; P.tau_free_streaming = c.tau_free_streaming;
  P.switch_sw = c.switch_sw; P.switch_eisw = c.switch_eisw; P.switch_lisw = c.switch_lisw; P.switch_dop = c.switch_dop;
  P.switch_pol = c.switch_pol; P.eisw_lisw_split_z = c.eisw_lisw_split_z;
  P.three_ceff2_ur = c.three_ceff2_ur; P.three_cvis2_ur = c.three_cvis2_ur;
  P.tp_size = c.tp_size; P.tp_t0 = c.index_tp_t0; P.tp_t1 = c.index_tp_t1; P.tp_t2 = c.index_tp_t2; P.tp_p = c.index_tp_p;
  P.tp_dm = c.index_tp_delta_m; P.tp_pp = c.index_tp_phi_plus_psi;
  for (int i = 0; i < CPT_NTK; i++) P.tp_tk[i] = c.has_transfers ? c.index_tp_transfer[i] : -1;
  P.tp_dn = (c.has_transfers && c.has_ncdm) ? c.index_tp_delta_ncdm1 : -1; P.tp_tn = (c.has_transfers && c.has_ncdm) ? c.index_tp_theta_ncdm1 : -1;
  P.start_small_k = c.start_small_k_at_tau_c_over_tau_h; P.start_large_k = c.start_large_k_at_tau_h_over_tau_k;
  P.tca_trig_h = c.tight_coupling_trigger_tau_c_over_tau_h; P.tca_trig_k = c.tight_coupling_trigger_tau_c_over_tau_k;
  P.rsa_trig = c.radiation_streaming_trigger_tau_over_tau_k; P.ufa_trig = c.ur_fluid_trigger_tau_over_tau_k;
  P.curvature_ini = c.curvature_ini; P.rtol = c.tol_perturb_integration; P.tol_tau_approx = c.tol_tau_approx;
  P.min_var = c.smallest_allowed_variation;
  P.has_ncdm = c.has_ncdm; P.nfa_method = c.ncdm_fluid_approximation; P.nfa_trig = c.ncdm_fluid_trigger_tau_over_tau_k;
  P.tol_ncdm_w = c.has_ncdm ? c.tol_ncdm_initial_w : 1e300; P.tp_dcb = c.has_ncdm ? c.index_tp_delta_cb : -1;
  P.nc = h->ncdm;
  P.max_steps = 400000;
  P.dbg_inverse = 0;
  // hierarchies longer than one wavefront (synchronous scalars without non-cold species): the tails go to chain waves of their own
  {
    const int lanes = (c.has_ncdm ? 13 + 3 * CPT_MAX_NCDM : 14) + (c.l_max_g - 2) + (c.l_max_pol_g - 2) + (c.has_ur ? c.l_max_ur - 2 : 0);
    P.long_tails = (c.mode == CPT_MODE_SCALARS && lanes > CPT_WAVE) ? 1 : 0;
    if (const char* e = getenv("CPT_LONG_TAILS")) P.long_tails = (c.mode == CPT_MODE_SCALARS && atoi(e) != 0) ? 1 : P.long_tails;
    P.long_len = max(c.l_max_g - 2, max(c.l_max_pol_g - 2, c.has_ur ? c.l_max_ur - 2 : 0));
  }
  // one tail per 16-lane row when each fits (defaults: 10 / 8 / 15 lanes), else the packed lane map with sequential sweeps
  P.rows = (c.mode == CPT_MODE_SCALARS && !c.has_ncdm && !P.long_tails && c.l_max_g - 2 <= 16 && c.l_max_pol_g - 2 <= 16 && (!c.has_ur || c.l_max_ur - 2 <= 16)) ? 1 : 0;
  if (const char* e = getenv("CPT_TAIL_ROWS")) P.rows = P.rows && atoi(e) != 0;
  P.k = nullptr; P.tau_s = nullptr; P.order = nullptr; P.nk = 0; P.ntau = 0; P.src = nullptr; P.stats = nullptr; P.status = nullptr;
}

}  // namespace
