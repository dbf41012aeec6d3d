// Developer tool: exp_nonpos() of csrc/flm_convt.hip (the library expf without its range tests) against expf, bit for bit,
// on 16.7 M arguments in [-110, 0] and the edge cases.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/exp_check.hip -o /tmp/exp_check && /tmp/exp_check
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include <string.h>
#include <stdlib.h>
__device__ __forceinline__ float exp_nonpos(float t) {
  const float ph = t * 0x1.715476p+0f;
  float pl = __builtin_fmaf(t, 0x1.715476p+0f, -ph);
  pl = __builtin_fmaf(t, 0x1.4ae0bep-26f, pl);
  const float e = __builtin_rintf(ph);
  const float a = (ph - e) + pl;
  return __builtin_ldexpf(__builtin_amdgcn_exp2f(a), (int)e);
}
__global__ void k(const float* x, float* y, float* z, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) { y[i] = expf(x[i]); z[i] = exp_nonpos(x[i]); } }
int main() {
  const int n = 1 << 24; float *x, *y, *z; hipMalloc(&x, n * 4); hipMalloc(&y, n * 4); hipMalloc(&z, n * 4);
  float* h = (float*)malloc(n * 4);
  for (int i = 0; i < n; ++i) { float u = (float)rand() / RAND_MAX; h[i] = (i & 3) == 0 ? -u * 110.f : ((i & 3) == 1 ? -u * 20.f : ((i & 3) == 2 ? -u : -u * 1e-3f)); }
  h[0] = 0.f; h[1] = -0.f; h[2] = -87.33f; h[3] = -88.f; h[4] = -103.9f; h[5] = -104.f; h[6] = -200.f; h[7] = -1e30f;
  hipMemcpy(x, h, n * 4, hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(x, y, z, n);
  float* hy = (float*)malloc(n * 4); float* hz = (float*)malloc(n * 4);
  hipMemcpy(hy, y, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hz, z, n * 4, hipMemcpyDeviceToHost);
  long diff = 0; for (int i = 0; i < n; ++i) if (memcmp(&hy[i], &hz[i], 4)) { if (diff < 8) printf("x=%g expf=%g mine=%g\n", h[i], hy[i], hz[i]); ++diff; }
  printf("%ld of %d differ\n", diff, n); return 0;
}
