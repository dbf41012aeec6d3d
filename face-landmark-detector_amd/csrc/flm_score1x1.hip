// 1x1 classifier convolutions of the bf16 configuration: Conv2D(C, 1x1) on f4 / f3 (networks/fcn.py:108, :117), 256 input
// channels, at most 80 stored columns, fp32 out.  At batch 512 score3 reads 268 MB and writes 151 MB for 18 GFLOP: a
// streaming kernel, and as a 128x128-tile implicit GEMM it is four k-steps of exposed load latency per tile with half
// the tile's columns empty (0.16 ms against a 0.07 ms HBM floor).  Here
//   * the whole weight matrix lives in REGISTERS: a wave holds the 5 x 8 fragments of v_mfma_f32_16x16x32_bf16 for
//     80 columns x 256 k (160 VGPRs), loaded once;
//   * a wave owns slices of 16 pixels: 8 fragment loads straight from global memory (lane = (pixel, 8 of the step's 32
//     channels): the eight loads of a slice cover its 16 x 512 bytes exactly once), 40 MFMAs, no LDS, no barrier;
//     eight waves per CU each with 8 KiB in flight cover the latency without software pipelining;
//   * the 16 x ldc outputs of a slice are one contiguous run: staged through a wave-private LDS line and written with
//     16-byte stores.
// k runs ascending in 32-deep MFMAs: the same sums as the implicit GEMM's 16-deep pairs (flm_igemm_bf16.hip, tested there),
// so the kernel is used at every batch size and a face's bits do not depend on which kernel a batch would have picked.
#include "flm_igemm_args.h"

namespace flm {

typedef float f32x4_s1 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_s1 __attribute__((ext_vector_type(8)));

constexpr int S1_K = 256, S1_STEPS = S1_K / 32, S1_TILES = 5, S1_MAXLD = 80;

__global__ __launch_bounds__(256, 2) void score1x1_bf16_kernel(const unsigned short* __restrict__ x,
                                                               const unsigned short* __restrict__ wt,
                                                               const float* __restrict__ scale, const float* __restrict__ shift,
                                                               float* __restrict__ y, int M, int cout, int ldc) {
  __shared__ __attribute__((aligned(16))) float stage[4][16 * S1_MAXLD];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l16 = lane & 15, kq = lane >> 4;
  // weights: column tile j, step s -> 8 bf16 of row 16j + l16 at k = 32s + 8kq (rows up to 79 exist: coutpad >= 128)
  float4 bw[S1_TILES][S1_STEPS];
#pragma unroll
  for (int j = 0; j < S1_TILES; ++j)
#pragma unroll
    for (int s = 0; s < S1_STEPS; ++s)
      bw[j][s] = *reinterpret_cast<const float4*>(wt + (size_t)(16 * j + l16) * S1_K + 32 * s + 8 * kq);
  float sc[S1_TILES], sh[S1_TILES];
#pragma unroll
  for (int j = 0; j < S1_TILES; ++j) {
    const int c = 16 * j + l16;
    sc[j] = c < cout ? scale[c] : 0.f;
    sh[j] = c < cout ? shift[c] : 0.f;
  }
  const int slices = (M + 15) >> 4;
  const int wave_g = blockIdx.x * 4 + wave, nwaves = gridDim.x * 4;
  float* st = stage[wave];
  for (int sl = wave_g; sl < slices; sl += nwaves) {
    const int m0 = sl << 4;
    const int row = min(m0 + l16, M - 1);  // (rows past M repeat the last one; they are not stored)
    const unsigned short* xr = x + (size_t)row * S1_K + 8 * kq;
    float4 af[S1_STEPS];
#pragma unroll
    for (int s = 0; s < S1_STEPS; ++s) af[s] = *reinterpret_cast<const float4*>(xr + 32 * s);
    f32x4_s1 acc[S1_TILES];
#pragma unroll
    for (int j = 0; j < S1_TILES; ++j) acc[j] = (f32x4_s1){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < S1_STEPS; ++s)
#pragma unroll
      for (int j = 0; j < S1_TILES; ++j)
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_s1, af[s]),
                                                         __builtin_bit_cast(bf16x8_s1, bw[j][s]), acc[j], 0, 0, 0);
    // accumulator: column 16j + l16, rows 4kq + r -> the slice's [16][ldc] image in LDS
#pragma unroll
    for (int j = 0; j < S1_TILES; ++j) {
      const int c = 16 * j + l16;
      if (c < ldc) {
#pragma unroll
        for (int r = 0; r < 4; ++r) st[(4 * kq + r) * ldc + c] = c < cout ? fmaf(acc[j][r], sc[j], sh[j]) : 0.f;
      }
    }
    __builtin_amdgcn_wave_barrier();
    const int nrows = min(16, M - m0);
    const int nf4 = (nrows * ldc) >> 2;  // ldc is a multiple of 4
    float* dst = y + (size_t)m0 * ldc;
    for (int i = lane; i < nf4; i += 64) *reinterpret_cast<float4*>(dst + 4 * i) = *reinterpret_cast<const float4*>(st + 4 * i);
    __builtin_amdgcn_wave_barrier();
  }
}

static std::atomic<int> g_score1x1{1};  // A/B knob: never changes results or layouts
void score1x1_enable(int on) { g_score1x1.store(on, std::memory_order_relaxed); }

// 1: launched; 0: shape left to the implicit-GEMM kernels
int launch_score1x1_bf16(hipStream_t s, const IgemmArgs& a, int relu, int pool, int posmajor, int coutpad) {
  if (!g_score1x1.load(std::memory_order_relaxed) || a.kh != 1 || a.kw != 1 || a.pad != 0 || a.stride != 1 || a.res ||
      relu || pool || posmajor || a.ksplit > 1 || !a.out_f32 || a.cin != S1_K || a.cout > S1_MAXLD || a.ldc > S1_MAXLD ||
      (a.ldc & 3) || a.cout > a.ldc || coutpad < S1_MAXLD || a.M < 1)
    return 0;
  const int slices = (a.M + 15) >> 4;
  int wgs = (slices + 3) / 4;
  if (wgs > 512) wgs = 512;  // two workgroups per CU
  score1x1_bf16_kernel<<<wgs, 256, 0, s>>>(static_cast<const unsigned short*>(a.x), static_cast<const unsigned short*>(a.wt),
                                           a.scale, a.shift, static_cast<float*>(a.y), a.M, a.cout, a.ldc);
  FLM_LAUNCH_CHECK("score1x1_bf16_kernel");
  return 1;
}

}  // namespace flm
