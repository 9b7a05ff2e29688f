// Single-wave latency microbenchmarks for gfx950: what does one dependent FP64 / cross-lane instruction cost when a
// lone wave owns the SIMD (the k_perturb situation)?   hipcc --offload-arch=gfx950 -O3 -o ubench tools/ubench.hip
// Every body is inline asm, so the compiler cannot reshape it; 16 copies per loop iteration, 256 iterations.
#include <hip/hip_runtime.h>
#include <cstdio>
#define R16(X) X X X X X X X X X X X X X X X X
#define ITER 256
template <int SEL>
__global__ void __launch_bounds__(64) k(double* out, long long* cyc, double a, double b) {
  __shared__ double lds[64];
  const int lane = threadIdx.x;
  double x = a + lane, y = b + lane, z = a * 2, w = b * 3, va = a, vb = b;
  int addr = ((lane + 1) & 63) * 4, la = lane * 8;
  lds[lane] = 8 * (lane ^ 1);
  __syncthreads();
  asm volatile("" : "+v"(x), "+v"(y), "+v"(z), "+v"(w), "+v"(va), "+v"(vb), "+v"(addr));
  long long t0 = clock64();
  for (int i = 0; i < ITER; i++) {
    if (SEL == 0) { R16(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(va), "v"(vb));) }
    if (SEL == 1) { R16(asm volatile("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5"
                                     : "+v"(x), "+v"(y), "+v"(z), "+v"(w) : "v"(va), "v"(vb));) }
    if (SEL == 2) { R16(asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(va));) }
    if (SEL == 3) { R16(asm volatile("v_add_f64 %0, %0, %1" : "+v"(x) : "v"(va));) }
    if (SEL == 4) { R16(asm volatile("v_readlane_b32 s20, %0, 5\n v_readlane_b32 s21, %1, 5\n v_fma_f64 %2, s[20:21], %3, %2"
                                     : "+v"(((int*)&x)[0]), "+v"(((int*)&x)[1]), "+v"(x) : "v"(va) : "s20", "s21");) }
    if (SEL == 5) { R16(asm volatile("s_nop 1\n v_mov_b32_dpp %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32 %0, %1, %2" : "+v"(addr), "+v"(la) : "v"(lane));) }
    if (SEL == 6) { R16(asm volatile("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(((int*)&x)[0]) : "v"(addr));) }
    if (SEL == 7) { R16(asm volatile("v_rcp_f64 %0, %0" : "+v"(x));) }
    if (SEL == 8) { R16(asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)\n v_cvt_u32_f64 %1, %0" : "+v"(x), "+v"(la));) }
    if (SEL == 9) { R16(asm volatile("v_max_f64 %0, %0, %1" : "+v"(x) : "v"(va));) }
    if (SEL == 10) { R16(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(((float*)&x)[0]) : "v"(((float*)&va)[0]), "v"(((float*)&vb)[0]));) }
    if (SEL == 11) { R16(asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(((int*)&x)[0]) : "v"(((int*)&va)[0]) : "vcc");) }
    if (SEL == 12) { R16(asm volatile("v_readlane_b32 s20, %0, 5\n v_mov_b32 %0, s20" : "+v"(((int*)&x)[0]) : : "s20");) }
    if (SEL == 13) { R16(asm volatile("v_mov_b32 %0, %0" : "+v"(((int*)&x)[0]));) }
    if (SEL == 14) { R16(asm volatile("v_fma_f64 %0, %0, %1, %2\n v_mov_b32 %3, %3\n v_mov_b32 %3, %3\n v_mov_b32 %3, %3" : "+v"(x) : "v"(va), "v"(vb), "v"(addr));) }
    if (SEL == 15) { R16(asm volatile("s_mov_b32 s20, s20" : : : "s20");) }
    if (SEL == 16) { R16(asm volatile("v_cmp_lt_f64 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc" : : "v"(x), "v"(va), "v"(addr), "v"(la) : "vcc");) }
    if (SEL == 18) { R16(asm volatile("s_branch 1f\n v_mov_b32 %0, %0\n1:\n v_add_u32 %0, %0, %1" : "+v"(addr) : "v"(lane));) }
    if (SEL == 19) { R16(asm volatile("s_cmp_lg_u32 0, 0\n s_cbranch_scc1 1f\n v_add_u32 %0, %0, %1\n1:" : "+v"(addr) : "v"(lane) : "scc");) }
    if (SEL == 20) { R16(asm volatile("s_cmp_eq_u32 0, 0\n s_cbranch_scc1 1f\n .rept 100\n v_mov_b32 %0, %0\n .endr\n1:\n v_add_u32 %0, %0, %1" : "+v"(addr) : "v"(lane) : "scc");) }
    if (SEL == 21) { R16(asm volatile("v_add_u32 %0, %0, %1" : "+v"(addr) : "v"(lane));) }
    if (SEL == 22) { R16(asm volatile("v_cmp_lt_f64 vcc, %1, %2\n s_cbranch_vccz 1f\n .rept 100\n v_mov_b32 %0, %0\n .endr\n1:\n v_add_u32 %0, %0, %3" : "+v"(addr) : "v"(va), "v"(x), "v"(lane) : "vcc");) }
    if (SEL == 23) { R16(asm volatile("v_cmp_lt_f64 vcc, %1, %2\n s_cbranch_vccnz 1f\n v_add_u32 %0, %0, %3\n1:" : "+v"(addr) : "v"(va), "v"(x), "v"(lane) : "vcc");) }
    if (SEL == 17) { R16(asm volatile("v_mul_f64 %0, %0, %2\n v_mul_f64 %1, %1, %2" : "+v"(x), "+v"(y) : "v"(va));) }
  }
  long long t1 = clock64();
  out[lane] = x + y + z + w + addr + la;
  if (lane == 0) cyc[0] = t1 - t0;
}
template <int SEL>
void run(const char* name, double* out, long long* cyc, int per) {
  long long c = 0;
  for (int rep = 0; rep < 2; rep++) {
    hipLaunchKernelGGL(k<SEL>, dim3(1), dim3(64), 0, 0, out, cyc, 1.0000001, 1e-9);
    hipDeviceSynchronize();
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  }
  printf("%-44s %7.2f ticks per group (%d instr)\n", name, (double)c / (16.0 * ITER), per);
}
int main() {
  double* out; long long* cyc;
  hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 8);
  run<0>("dependent v_fma_f64", out, cyc, 1);
  run<1>("4 independent v_fma_f64", out, cyc, 4);
  run<2>("dependent v_mul_f64", out, cyc, 1);
  run<3>("dependent v_add_f64", out, cyc, 1);
  run<17>("2 independent v_mul_f64", out, cyc, 2);
  run<4>("2 readlane + fma on the SGPR pair (dep)", out, cyc, 3);
  run<5>("nop + dpp wave_shr + add (dep)", out, cyc, 3);
  run<6>("ds_bpermute_b32 + wait (dep)", out, cyc, 1);
  run<7>("dependent v_rcp_f64", out, cyc, 1);
  run<8>("ds_read_b64 + wait + cvt (dep)", out, cyc, 2);
  run<9>("dependent v_max_f64", out, cyc, 1);
  run<10>("dependent v_fma_f32", out, cyc, 1);
  run<11>("dependent v_cndmask_b32", out, cyc, 1);
  run<12>("readlane + v_mov from SGPR (dep)", out, cyc, 2);
  run<13>("dependent v_mov_b32", out, cyc, 1);
  run<14>("fma_f64 + 3 independent v_mov", out, cyc, 4);
  run<15>("s_mov_b32", out, cyc, 1);
  run<16>("v_cmp_f64 + cndmask", out, cyc, 2);
  run<21>("dependent v_add_u32 (reference for the next four)", out, cyc, 1);
  run<18>("taken s_branch over 1 instr + add", out, cyc, 2);
  run<19>("s_cmp + NOT taken s_cbranch + add", out, cyc, 3);
  run<20>("s_cmp + taken s_cbranch over 100 instr + add", out, cyc, 3);
  run<22>("v_cmp_f64 + taken cbranch_vccz over 100 + add", out, cyc, 3);
  run<23>("v_cmp_f64 + NOT taken cbranch_vccnz + add", out, cyc, 3);
  return 0;
}
