// Developer tool: which bf16 MFMA shape suits the 256x256-tile implicit GEMM (csrc/flm_igemm_bf16.hip)?
// The inner step of that kernel, without its global side: 8 waves per CU (2 per SIMD), each wave a 128x64 sub-tile,
// operands read from LDS as 16-byte fragments (random bits, conflict-free lane-linear layout), per 64-deep k-step
//   shape A  v_mfma_f32_32x32x16_bf16: 4 slices x (4 im2col + 2 weight fragments) -> 32 MFMAs of 32 cycles,  8 x 16 accumulator registers
//   shape B  v_mfma_f32_16x16x32_bf16: 2 slices x (8 im2col + 4 weight fragments) -> 64 MFMAs of 16 cycles, 32 x  4 accumulator registers
// Same fragment bytes (24 ds_read_b128 per wave and step), same accumulator registers (128), same MFMA pipe cycles (1024);
// one __syncthreads per step as in the kernel.  Prints TFLOP/s and the in-kernel clock for both.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_shape_gemm.hip -o /tmp/mfma_shape_gemm && /tmp/mfma_shape_gemm
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int kLdsBytes = 96 * 1024;

template <int SHAPE>
__global__ __launch_bounds__(512, 1) void gemm_step_kernel(const uint4* __restrict__ seed, float* __restrict__ out,
                                                           unsigned long long* __restrict__ clk, int steps) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < kLdsBytes / 16; i += 512) {
    uint4 v = seed[(i + 37 * blockIdx.x) & 4095];
    v.x &= 0x3fff3fff; v.y &= 0x3fff3fff; v.z &= 0x3fff3fff; v.w &= 0x3fff3fff;  // moderate exponents
    reinterpret_cast<uint4*>(smem)[i] = v;
  }
  __syncthreads();
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  // fragment f of this wave: 1 KiB, lane-linear; 24 fragments per step and wave, walked through 48 KiB per wave pair
  const char* base = smem + (wave & 3) * 24 * 1024 + lane * 16;
  float res = 0.f;
  if (SHAPE == 0) {
    f32x16 acc[4][2];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int st = 0; st < steps; ++st) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        bf16x8 af[4], bf[2];
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const bf16x8*>(base + (s * 6 + i) * 1024);
#pragma unroll
        for (int j = 0; j < 2; ++j) bf[j] = *reinterpret_cast<const bf16x8*>(base + (s * 6 + 4 + j) * 1024);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
      }
      __syncthreads();
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) res += acc[i][j][r];
  } else {
    f32x4 acc[8][4];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int st = 0; st < steps; ++st) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bf16x8 af[8], bf[4];
#pragma unroll
        for (int i = 0; i < 8; ++i) af[i] = *reinterpret_cast<const bf16x8*>(base + (s * 12 + i) * 1024);
#pragma unroll
        for (int j = 0; j < 4; ++j) bf[j] = *reinterpret_cast<const bf16x8*>(base + (s * 12 + 8 + j) * 1024);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
      }
      __syncthreads();
    }
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) for (int r = 0; r < 4; ++r) res += acc[i][j][r];
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * 512 + tid] = res;
  if (tid == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int SHAPE>
static void run(const char* name, const uint4* seed) {
  const int blocks = 256, steps = 3000;  // one workgroup per CU; 3000 steps x 1024 pipe cycles ~ 1.7 ms
  float* out; unsigned long long* clk;
  hipMalloc(&out, sizeof(float) * blocks * 512);
  hipMalloc(&clk, sizeof(unsigned long long) * 2 * blocks);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_step_kernel<SHAPE>), hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  double tf = 0, ghz = 0;
  for (int rep = 0; rep < 6; ++rep) {
    hipEventRecord(e0);
    for (int k = 0; k < 40; ++k) gemm_step_kernel<SHAPE><<<blocks, 512, kLdsBytes>>>(seed, out, clk, steps);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // per step and wave: 128 x 64 x 64 MACs
    tf = 40.0 * blocks * 8 * (double)steps * 2.0 * 128 * 64 * 64 / (ms * 1e-3) / 1e12;
    std::vector<unsigned long long> h(2 * blocks);
    hipMemcpy(h.data(), clk, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost);
    std::vector<double> g;
    for (int b = 0; b < blocks; ++b) if (h[2 * b + 1]) g.push_back((double)h[2 * b] / (double)h[2 * b + 1] * 0.1);
    std::sort(g.begin(), g.end());
    ghz = g.empty() ? 0 : g[g.size() / 2];
  }
  hipError_t e = hipGetLastError();
  printf("%-44s %8.1f TFLOP/s at %.2f GHz in-kernel clock (last of 6 rounds)%s\n", name, tf, ghz, e == hipSuccess ? "" : "  [launch error]");
  hipFree(out); hipFree(clk);
}

int main() {
  std::vector<unsigned> h(4 * 4096);
  srand(11);
  for (auto& v : h) v = ((unsigned)rand() << 16) ^ (unsigned)rand();
  uint4* seed; hipMalloc(&seed, sizeof(unsigned) * h.size());
  hipMemcpy(seed, h.data(), sizeof(unsigned) * h.size(), hipMemcpyHostToDevice);
  run<0>("32x32x16_bf16: 32 MFMAs + 24 ds_read_b128 / step", seed);
  run<1>("16x16x32_bf16: 64 MFMAs + 24 ds_read_b128 / step", seed);
  run<0>("32x32x16_bf16 (again)", seed);
  run<1>("16x16x32_bf16 (again)", seed);
  return 0;
}
