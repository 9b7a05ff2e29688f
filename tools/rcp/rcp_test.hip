// accuracy of v_rcp_f64 and of its Newton refinements (how many steps does fast_rcp need?)   hipcc --offload-arch=gfx950 -O3 -o rcp_test rcp_test.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstdlib>
__global__ void k(const double* x, double* r0, double* r1, double* r2, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double v = x[i];
  double r = __builtin_amdgcn_rcp(v);
  r0[i] = r;
  r = fma(r, fma(-v, r, 1.0), r);
  r1[i] = r;
  r = fma(r, fma(-v, r, 1.0), r);
  r2[i] = r;
}
int main() {
  const int n = 1 << 20;
  double *x = (double*)malloc(n * 8), *h0 = (double*)malloc(n * 8), *h1 = (double*)malloc(n * 8), *h2 = (double*)malloc(n * 8);
  srand(1);
  for (int i = 0; i < n; i++) x[i] = ldexp(1.0 + rand() / (double)RAND_MAX, (rand() % 200) - 100) * ((rand() & 1) ? 1 : -1);
  double *dx, *d0, *d1, *d2;
  hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8);
  hipMemcpy(dx, x, n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, d0, d1, d2, n);
  hipMemcpy(h0, d0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(h1, d1, n * 8, hipMemcpyDeviceToHost); hipMemcpy(h2, d2, n * 8, hipMemcpyDeviceToHost);
  double e0 = 0, e1 = 0, e2 = 0;
  for (int i = 0; i < n; i++) {
    const long double t = 1.0L / (long double)x[i];
    e0 = fmax(e0, (double)fabsl((h0[i] - t) / t)); e1 = fmax(e1, (double)fabsl((h1[i] - t) / t)); e2 = fmax(e2, (double)fabsl((h2[i] - t) / t));
  }
  printf("max relative error of 1/x: v_rcp_f64 %.3e, + 1 Newton step %.3e, + 2 steps %.3e  (2^-53 = %.3e)\n", e0, e1, e2, ldexp(1.0, -53));
  return 0;
}
