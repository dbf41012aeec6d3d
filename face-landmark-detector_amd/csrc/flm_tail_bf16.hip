// The second skip stage of the FCN-8 decoder in ONE launch, bf16 configuration, 68 classes:
//
//   seg_feats = crop(Conv2DTranspose(C, 4x4, stride 2)(fuse4)) + Conv2D(C, 1x1)(f3)          networks/fcn.py:114-119
//
// As two launches (score3: flm_score1x1.hip, then up4: the generic transposed-conv kernel of flm_convt.hip with the
// skip add) the stage took 0.078 + 0.135 ms per 512 faces: score3 wrote 151 MB that up4 read back, and up4 itself is a
// latency-bound launch (MFMA pipe 8 % busy, waves parked 75 %: four phases of a weight ring with three barriers each,
// for a layer of 0.04 GFLOP per face).  Here a wave owns slices of 16 output pixels of ONE phase (a0, b0) -- the
// workgroup's phase; its 45 KiB of transposed-conv weights sit in LDS for the workgroup's life -- and for each slice
//   * gathers the 2 x 2 input pixels of fuse4 (fp32 -> bf16, as the generic kernel does) and multiplies the phase's
//     9 x 5 weight fragments: the arithmetic of convt_kernel<5, 9, true, ...>, operand for operand (weights on the MFMA
//     rows, pixels on the lanes, k groups ascending), so the accumulators hold ITS bits;
//   * streams the slice's 16 x 512 bytes of f3 against the score3 matrix: the arithmetic of score1x1_bf16_kernel,
//     operand for operand;
//   * adds the two in a wave-private LDS image of the slice (the transposed conv's lanes hold pixels, the classifier's
//     lanes hold classes: the image is where they meet) and stores the slice's 16 x 288 bytes.
// The sum is the unfused stage's `up4 + (score3 + bias)` with the operands swapped: the same bits
// (tests/test_gpu_forward.py: seg_feats and landmarks equal with the knob "bf16_fused_tail" on and off).
// HBM: f3 268 MB + fuse4 38 MB read, seg_feats 151 MB written per 512 faces -- 0.46 GB, ~0.1 ms at streaming rate.
#include "flm_common.h"

namespace flm {

typedef float f32x4_t2 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t2 __attribute__((ext_vector_type(8)));

constexpr int T2_K3 = 256, T2_STEPS3 = T2_K3 / 32, T2_TILES = 5, T2_G = 9, T2_CP = 72, T2_C = 68;
constexpr int T2_PHASE_F4 = T2_G * T2_TILES * 64;   // float4 per phase of packed transposed-conv weights (45 KiB)
constexpr int T2_W3_F4 = T2_TILES * T2_STEPS3 * 64;  // float4 of the classifier's matrix in fragment order (40 KiB)
constexpr int T2_WAVES = 8;

// One workgroup of eight waves per CU.  Both weight sets live in LDS in fragment order (85 KiB): with the classifier's
// matrix in registers, as in flm_score1x1.hip, a wave had no room to keep a slice's 26 loads in flight and ran its nine
// gather-convert-multiply groups as one latency chain (0.17 ms per 512 faces; this form: see DESIGN).
__global__ __launch_bounds__(64 * T2_WAVES, 1) void seg_fused_bf16_kernel(const float* __restrict__ fuse4,       // [n,h4,w4,72]
                                                                          const float4* __restrict__ w4,          // [4][9][5][64] x 16 B
                                                                          const unsigned short* __restrict__ f3, // [n,2h4,2w4,256]
                                                                          const unsigned short* __restrict__ w3, // [>=80][256]
                                                                          const float* __restrict__ scale3,
                                                                          const float* __restrict__ shift3,
                                                                          float* __restrict__ seg,                // [n,2h4,2w4,72]
                                                                          int n, int h4, int w4d) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float4* lds4 = reinterpret_cast<float4*>(smem_raw);                                   // the phase's transposed-conv weights
  float4* lds3 = lds4 + T2_PHASE_F4;                                                    // score3: [tile j][step s][lane]
  float* stage_all = reinterpret_cast<float*>(lds3 + T2_W3_F4);                         // [8][16 * 72]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l16 = lane & 15, kq = lane >> 4;
  const int phase = blockIdx.x & 3, a0 = phase >> 1, b0 = phase & 1;
  const int wb = blockIdx.x >> 2, nbp = gridDim.x >> 2;
  for (int i = tid; i < T2_PHASE_F4; i += 64 * T2_WAVES) lds4[i] = w4[(size_t)phase * T2_PHASE_F4 + i];
  // column tile j, step s, lane (l16, kq) -> 8 bf16 of row 16j + l16 at k = 32s + 8kq (the B fragment of score1x1_bf16_kernel)
  for (int i = tid; i < T2_W3_F4; i += 64 * T2_WAVES) {
    const int ln = i & 63, js = i >> 6, j = js / T2_STEPS3, st_ = js - j * T2_STEPS3;
    lds3[i] = *reinterpret_cast<const float4*>(w3 + (size_t)(16 * j + (ln & 15)) * T2_K3 + 32 * st_ + 8 * (ln >> 4));
  }
  float sc[T2_TILES], sh[T2_TILES];
#pragma unroll
  for (int j = 0; j < T2_TILES; ++j) {
    const int c = 16 * j + l16;
    sc[j] = c < T2_C ? scale3[c] : 0.f;
    sh[j] = c < T2_C ? shift3[c] : 0.f;
  }
  __syncthreads();
  float* st = stage_all + wave * 16 * T2_CP;
  const int h3 = 2 * h4, w3d = 2 * w4d;
  const int cg = (w4d + 15) >> 4;
  const int slices = n * h4 * cg;
  // U slices per iteration with all their loads in flight before the first is consumed (U = 2: 252 registers and no
  // faster, 0.129 against 0.124 ms per 512 faces: eight waves x 26 loads already cover the latency)
  constexpr int U = 1;
  const int sstride = nbp * T2_WAVES;
  for (int sl0 = wb * T2_WAVES + wave; sl0 < slices; sl0 += U * sstride) {
    int img[U], i0[U], jc[U], oy[U];
    bool live[U], pv[U];
    float4 x0[U][T2_G], x1[U][T2_G], af3[U][T2_STEPS3];
    bool okg[U][T2_G];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int sl = sl0 + u * sstride;
      live[u] = sl < slices;                     // (wave-uniform)
      const int slc = live[u] ? sl : slices - 1;
      img[u] = slc / (h4 * cg);
      const int rem = slc - img[u] * (h4 * cg);
      i0[u] = rem / cg;
      jc[u] = rem - i0[u] * cg;
      const int j0 = 16 * jc[u] + l16;           // this lane's pixel column of the slice (both operand roles index it by l16)
      pv[u] = j0 < w4d;
      oy[u] = 2 * i0[u] + a0;
      const int ox = 2 * (pv[u] ? j0 : w4d - 1) + b0;  // (columns past the row repeat the last pixel; they are not stored)
#pragma unroll
      for (int g = 0; g < T2_G; ++g) {
        const int k0 = 32 * g + 8 * kq;
        const int tap = k0 / T2_CP, c = k0 - tap * T2_CP;
        const int ii = i0[u] - (tap >> 1), jj = j0 - (tap & 1);
        okg[u][g] = pv[u] && ii >= 0 && jj >= 0;
        const size_t off = okg[u][g] ? (((size_t)img[u] * h4 + ii) * w4d + jj) * T2_CP + c : 0;
        x0[u][g] = *reinterpret_cast<const float4*>(fuse4 + off);
        x1[u][g] = *reinterpret_cast<const float4*>(fuse4 + off + 4);
      }
      const unsigned short* xr = f3 + (((size_t)img[u] * h3 + oy[u]) * w3d + ox) * T2_K3 + 8 * kq;
#pragma unroll
      for (int s = 0; s < T2_STEPS3; ++s) af3[u][s] = *reinterpret_cast<const float4*>(xr + 32 * s);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (!live[u]) break;
      // ---- transposed conv, phase (a0, b0): D[class][pixel] += W[class][k] * X[k][pixel], k = tap * 72 + c ------------
      f32x4_t2 acc4[T2_TILES];
#pragma unroll
      for (int m = 0; m < T2_TILES; ++m) acc4[m] = (f32x4_t2){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int g = 0; g < T2_G; ++g) {
        const bool ok = okg[u][g];
        bf16x8_t2 t;
        t[0] = (__bf16)(ok ? x0[u][g].x : 0.f); t[1] = (__bf16)(ok ? x0[u][g].y : 0.f);
        t[2] = (__bf16)(ok ? x0[u][g].z : 0.f); t[3] = (__bf16)(ok ? x0[u][g].w : 0.f);
        t[4] = (__bf16)(ok ? x1[u][g].x : 0.f); t[5] = (__bf16)(ok ? x1[u][g].y : 0.f);
        t[6] = (__bf16)(ok ? x1[u][g].z : 0.f); t[7] = (__bf16)(ok ? x1[u][g].w : 0.f);
#pragma unroll
        for (int m = 0; m < T2_TILES; ++m) {
          const float4 af = lds4[(g * T2_TILES + m) * 64 + lane];
          acc4[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t2, af), t, acc4[m], 0, 0, 0);
        }
      }
      // result row 4kq + e of tile m is class 16m + 4kq + e (tile 4: row 4kq is class 64 + kq), column l16 the pixel
#pragma unroll
      for (int m = 0; m < 4; ++m)
        *reinterpret_cast<float4*>(st + l16 * T2_CP + 16 * m + 4 * kq) = make_float4(acc4[m][0], acc4[m][1], acc4[m][2], acc4[m][3]);
      st[l16 * T2_CP + 64 + kq] = acc4[4][0];
      st[l16 * T2_CP + 68 + kq] = 0.f;  // the pad channels of the 72-wide buffer
      __builtin_amdgcn_wave_barrier();
      // ---- 1x1 classifier on f3 at the slice's output pixels: D[pixel][class] --------------------------------------------
      f32x4_t2 acc3[T2_TILES];
#pragma unroll
      for (int j = 0; j < T2_TILES; ++j) acc3[j] = (f32x4_t2){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < T2_STEPS3; ++s)
#pragma unroll
        for (int j = 0; j < T2_TILES; ++j) {
          const float4 bwf = lds3[(j * T2_STEPS3 + s) * 64 + lane];
          acc3[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t2, af3[u][s]),
                                                            __builtin_bit_cast(bf16x8_t2, bwf), acc3[j], 0, 0, 0);
        }
      // accumulator: column 16j + l16 (class), rows 4kq + r (pixel): seg = up4 + (score3 * scale + shift)
#pragma unroll
      for (int j = 0; j < T2_TILES; ++j) {
        const int c = 16 * j + l16;
        if (c < T2_C) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float* p = st + (4 * kq + r) * T2_CP + c;
            *p = *p + fmaf(acc3[j][r], sc[j], sh[j]);
          }
        }
      }
      __builtin_amdgcn_wave_barrier();
      // ---- the slice's pixels (ox = 2 j0 + b0: every other pixel of the row), 288 contiguous bytes each --------------------
      const int npx = min(16, w4d - 16 * jc[u]);
      float* drow = seg + (((size_t)img[u] * h3 + oy[u]) * w3d + (size_t)(32 * jc[u] + b0)) * T2_CP;
      for (int i = lane; i < npx * (T2_CP / 4); i += 64) {
        const int p = i / (T2_CP / 4), f = i - p * (T2_CP / 4);
        *reinterpret_cast<float4*>(drow + (size_t)p * 2 * T2_CP + 4 * f) = *reinterpret_cast<const float4*>(st + p * T2_CP + 4 * f);
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
}

static std::atomic<int> g_fused_tail{1};  // A/B knob "bf16_fused_tail": same bits either way
void tail_fused_enable(int on) { g_fused_tail.store(on, std::memory_order_relaxed); }

// 1: launched (seg_feats written); 0: shape left to the two-launch form; < 0: error
int launch_seg_fused_bf16(hipStream_t s, const float* fuse4, const void* w4_packed, const void* f3, const void* w3,
                          const float* scale3, const float* shift3, float* seg, int n, int h4, int w4d, int C, int Cp,
                          int G, int cin3, int coutpad3) {
  if (!g_fused_tail.load(std::memory_order_relaxed) || C != T2_C || Cp != T2_CP || G != T2_G || cin3 != T2_K3 ||
      coutpad3 < 16 * T2_TILES || n < 1 || h4 < 1 || w4d < 1)
    return 0;
  // 32-bit slice arithmetic; the buffers themselves are indexed with size_t
  if ((long long)n * h4 * ((w4d + 15) / 16) >= (1ll << 30)) return 0;
  constexpr size_t lds = sizeof(float4) * (T2_PHASE_F4 + T2_W3_F4) + sizeof(float) * T2_WAVES * 16 * T2_CP;
  static FuncAttrOnce attr;
  FLM_FUNC_ATTR_ONCE(attr, (&seg_fused_bf16_kernel), lds);
  const long long slices = (long long)n * h4 * ((w4d + 15) / 16);  // per phase
  long long per_phase = (slices + T2_WAVES - 1) / T2_WAVES;          // workgroups of eight waves
  if (per_phase > 64) per_phase = 64;                                // 4 phases x 64 = one workgroup per CU
  seg_fused_bf16_kernel<<<dim3((unsigned)(4 * per_phase)), 64 * T2_WAVES, lds, s>>>(
      fuse4, static_cast<const float4*>(w4_packed), static_cast<const unsigned short*>(f3),
      static_cast<const unsigned short*>(w3), scale3, shift3, seg, n, h4, w4d);
  FLM_LAUNCH_CHECK("seg_fused_bf16_kernel");
  return 1;
}

}  // namespace flm
