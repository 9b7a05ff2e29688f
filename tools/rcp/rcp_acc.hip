// Diagnostic: accuracy of v_rcp_f64 / v_rsq_f64 seeds and of one / two Newton refinements (decides the form of fast_rcp / fast_sqrt in cpt_perturb.hip).
//   hipcc --offload-arch=gfx950 -O3 -o rcp_acc rcp_acc.hip && ./rcp_acc
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const double* x, double* o, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double v = x[i];
  double r0 = __builtin_amdgcn_rcp(v);
  double r1 = fma(r0, fma(-v, r0, 1.0), r0);
  double r2 = fma(r1, fma(-v, r1, 1.0), r1);
  o[3 * i] = r0; o[3 * i + 1] = r1; o[3 * i + 2] = r2;
}
int main() {
  const int n = 1 << 22;
  std::vector<double> x(n), o(3 * n);
  unsigned long long s = 88172645463325252ull;
  for (int i = 0; i < n; i++) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    double m = 1.0 + (double)(s >> 11) / 9007199254740992.0;          // [1, 2)
    int e = (int)((s >> 3) % 600) - 300;
    x[i] = ldexp(m, e) * ((s & 1) ? 1 : -1);
  }
  double *dx, *dout;
  hipMalloc(&dx, n * 8); hipMalloc(&dout, 3 * n * 8);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
  hipMemcpy(o.data(), dout, 3 * n * 8, hipMemcpyDeviceToHost);
  double e0 = 0, e1 = 0, e2 = 0;
  for (int i = 0; i < n; i++) {
    long double t = 1.0L / (long double)x[i];
    e0 = fmax(e0, fabs((double)((o[3 * i] - t) / t)));
    e1 = fmax(e1, fabs((double)((o[3 * i + 1] - t) / t)));
    e2 = fmax(e2, fabs((double)((o[3 * i + 2] - t) / t)));
  }
  printf("max relative error: v_rcp_f64 %.3e   + 1 Newton step %.3e   + 2 Newton steps %.3e   (eps = 1.11e-16)\n", e0, e1, e2);
  return 0;
}
