// Developer tool: what the matrix pipe of THIS chip delivers in a bare loop -- the local ceiling the bf16 / fp32
// kernels are compared with in DESIGN.md section 5.  Operands live in registers (random bit patterns, not zeros:
// the chip holds a lower clock on random data), accumulators are independent, one or two waves per SIMD, every CU
// busy; reports TFLOP/s from wall time and the in-kernel clock (s_memtime / s_memrealtime, 100 MHz reference).
// The MFMAs are written as inline assembly with the accumulator tied in place: through the builtin hipcc moved the
// 16x16x32 accumulators between AGPRs and VGPRs inside the loop (a dozen v_accvgpr copies per MFMA), which made that
// row of round 1's table read 45 cycles per MFMA instead of the pipe's 16 (MI355X_MICROARCH.md).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND>
__global__ __launch_bounds__(256) void peak_kernel(const unsigned* __restrict__ seed, float* __restrict__ out,
                                                   unsigned long long* __restrict__ clk, int iters) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  uint4 ra = reinterpret_cast<const uint4*>(seed)[t & 1023], rb = reinterpret_cast<const uint4*>(seed)[(t + 333) & 1023];
  // keep exponents moderate so sums stay finite: clear the top exponent bits of every bf16 / fp32 lane
  ra.x &= 0x3fff3fff; ra.y &= 0x3fff3fff; ra.z &= 0x3fff3fff; ra.w &= 0x3fff3fff;
  rb.x &= 0x3fff3fff; rb.y &= 0x3fff3fff; rb.z &= 0x3fff3fff; rb.w &= 0x3fff3fff;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  float res = 0.f;
  if (KIND == 0) {  // v_mfma_f32_32x32x16_bf16, 4 independent accumulators
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const bf16x8 a = __builtin_bit_cast(bf16x8, ra), b = __builtin_bit_cast(bf16x8, rb);
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 4; ++i) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    for (int i = 0; i < 4; ++i) res += acc[i][0];
  } else if (KIND == 1) {  // v_mfma_f32_16x16x32_bf16, 8 independent accumulators
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const bf16x8 a = __builtin_bit_cast(bf16x8, ra), b = __builtin_bit_cast(bf16x8, rb);
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");  // (the last results land before they are read)
    for (int i = 0; i < 8; ++i) res += acc[i][0];
  } else {  // v_mfma_f32_32x32x2_f32, 4 independent accumulators
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const float a = __uint_as_float(ra.x), b = __uint_as_float(rb.x);
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int i = 0; i < 4; ++i) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    for (int i = 0; i < 4; ++i) res += acc[i][0];
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[t] = res;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int KIND>
static void run(const char* name, double flop_per_mfma, int mfma_per_iter, int waves_per_simd, const unsigned* seed) {
  const int blocks = 256 * waves_per_simd;  // 256 CUs x (1 or 2) workgroups of 4 waves
  const int iters = 1600000 / mfma_per_iter;  // ~25-50 ms per launch
  float* out; unsigned long long* clk;
  hipMalloc(&out, sizeof(float) * blocks * 256);
  hipMalloc(&clk, sizeof(unsigned long long) * 2 * blocks);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  double best = 0, ghz = 0;
  for (int rep = 0; rep < 6; ++rep) {  // ~1.5 s of back-to-back launches: the clock settles
    hipEventRecord(e0);
    for (int k = 0; k < 8; ++k) peak_kernel<KIND><<<blocks, 256>>>(seed, out, clk, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double tf = 8.0 * blocks * 4 * (double)iters * mfma_per_iter * flop_per_mfma / (ms * 1e-3) / 1e12;
    std::vector<unsigned long long> h(2 * blocks);
    hipMemcpy(h.data(), clk, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost);
    std::vector<double> g;
    for (int b = 0; b < blocks; ++b) if (h[2 * b + 1]) g.push_back((double)h[2 * b] / (double)h[2 * b + 1] * 0.1);
    std::sort(g.begin(), g.end());
    best = tf; ghz = g.empty() ? 0 : g[g.size() / 2];
  }
  printf("%-28s %d wave(s)/SIMD: %8.1f TFLOP/s at %.2f GHz in-kernel clock (last of 6 rounds)\n", name, waves_per_simd, best, ghz);
  hipFree(out); hipFree(clk);
}

int main() {
  std::vector<unsigned> h(4096);
  srand(7);
  for (auto& v : h) v = ((unsigned)rand() << 16) ^ (unsigned)rand();
  unsigned* seed; hipMalloc(&seed, sizeof(unsigned) * 4096);
  hipMemcpy(seed, h.data(), sizeof(unsigned) * 4096, hipMemcpyHostToDevice);
  for (int w = 1; w <= 2; ++w) {
    run<0>("v_mfma_f32_32x32x16_bf16", 2.0 * 32 * 32 * 16, 4, w, seed);
    run<1>("v_mfma_f32_16x16x32_bf16", 2.0 * 16 * 16 * 32, 8, w, seed);
    run<2>("v_mfma_f32_32x32x2_f32", 2.0 * 32 * 32 * 2, 4, w, seed);
  }
  return 0;
}
