// Transposed convolutions of the FCN-8 decoder, exact fp32 on the matrix cores
// (v_mfma_f32_16x16x4_f32), with the crop/add and the per-pixel softmax / argmax fused.
//
//   up5  Conv2DTranspose(C, 4x4, stride 2, valid, no bias)   networks/fcn.py:104-105
//        + crop to the skip map (keeps the top-left window)   fcn.py:55-86,110
//        + Add(score4)                                        fcn.py:112
//   up4  same, + Add(score3)                                  fcn.py:114-119
//   up3  Conv2DTranspose(C, 16x16, stride 8)                  fcn.py:121-122
//        + Reshape + softmax over classes                     networks/utils.py:28-30
//        (+ argmax over classes                               prediction.py:209)
//
// Kernel size = 2 * stride in all three, so in gather form every output pixel
// (s*i0+a0, s*j0+b0) sums exactly 2x2 input pixels (i0-di, j0-dj) with filter taps
// (a0+s*di, b0+s*dj): per phase (a0,b0) a GEMM  D[class][pixel] = W_phase[class][k] * X[k][pixel],
// k = (di,dj,c), K = 4*Cp.
//
// Classes sit on the MFMA ROW index, pixels on the lane: all classes of a pixel are then 4*MT
// registers of 4 lanes (lane, lane+16, +32, +48), so softmax/argmax over classes is an in-lane
// reduction plus two xor-shuffles.  One workgroup = 64 input positions (16 per wave) x one phase
// row a0 x all b0: each wave keeps its X fragments in registers for all phases (68 VGPRs at C=68)
// and streams the phase's weights, pre-packed in fragment order, through a double-buffered LDS
// ring (lane-linear ds_read_b128, conflict-free by construction).  For a fixed a0 consecutive b0
// write 8 consecutive output pixels = one 2176-byte run at C=68.
#include "flm_common.h"

namespace flm {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct ConvTArgs {
  const float* x;
  const void* wf;
  const float* skip;
  void* y;
  int n, hi, wi, ho, wo, s, ldy, epilogue;
  int C, Cp;
  int P;  // n*(hi+1)*(wi+1) input positions (one extra row/column: the far taps)
};

constexpr int GCH_F32 = 6, GCH_BF16 = 3;  // k groups per LDS chunk (bf16: smaller chunks, fewer staging registers)

// exp(t) for t <= 0 in the softmax.  fp32 path: the accurate library expf.  bf16 path: v_exp_f32 on
// t*log2(e) (about 1e-6 relative, far below the bf16 rounding the logits already carry); at 16x the MFMA
// rate the 20 accurate expf per lane per phase would cost more than the phase's matrix work.
template <bool BF>
__device__ __forceinline__ float softmax_exp(float t) {
  if constexpr (BF) return __builtin_amdgcn_exp2f(t * 1.44269504088896340736f);
  else return expf(t);
}

template <int MT, int G, bool BF, int NT>
__global__ __launch_bounds__(256, (!BF && MT >= 5 && G >= 20) ? 1 : 2) void convt_kernel(ConvTArgs a) {
  constexpr int GCH = BF ? GCH_BF16 : GCH_F32;
  constexpr int NCH = (G + GCH - 1) / GCH;
  constexpr int CHUNK_F4 = GCH * MT * 64;             // float4 per full chunk
  constexpr int NLD = (CHUNK_F4 + 255) / 256;         // staging loads per thread
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float4* lds = reinterpret_cast<float4*>(smem_raw);  // [2][CHUNK_F4]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int a0 = blockIdx.y;
  const int s = a.s;

  // ---- this lane's NT input positions (NT pixel tiles of 16 per wave: the phase's weights, streamed once
  //      per workgroup, then serve 64*NT positions) --------------------------------------------------
  const int wi1 = a.wi + 1, hi1 = a.hi + 1;
  bool pvalid[NT];
  int i0[NT], j0[NT], img[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int p = (blockIdx.x * 4 + wave) * 16 * NT + nt * 16 + r;
    pvalid[nt] = p < a.P;
    const int pp = pvalid[nt] ? p : 0;
    j0[nt] = pp % wi1;
    i0[nt] = (pp / wi1) % hi1;
    img[nt] = pp / (wi1 * hi1);
  }

  // ---- X fragments ------------------------------------------------------------------------------------
  //   fp32: xf[g] = x[tap(k4)][c(k4)..+3],  k4 = 16g + 4q   (4 floats)
  //   bf16: xf[g] = bf16(x[tap(k8)][c(k8)..+7]), k8 = 32g + 8q (8 floats converted, 16 bytes)
  float4 xf[NT][G];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
  for (int g = 0; g < G; ++g) {
    constexpr int EPL = BF ? 8 : 4;
    const int k0 = 4 * EPL * g + EPL * q;
    const int tap = k0 / a.Cp, c = k0 % a.Cp;
    const int ii = i0[nt] - (tap >> 1), jj = j0[nt] - (tap & 1);
    const bool ok = pvalid[nt] && tap < 4 && (unsigned)ii < (unsigned)a.hi && (unsigned)jj < (unsigned)a.wi;
    const size_t off = ok ? (((size_t)img[nt] * a.hi + ii) * a.wi + jj) * a.Cp + c : 0;
    // (component-wise selects: a float4 struct select goes through scratch memory)
    const float km = ok ? 1.f : 0.f;
    if constexpr (BF) {
      const float4 v0 = *reinterpret_cast<const float4*>(a.x + off);
      const float4 v1 = *reinterpret_cast<const float4*>(a.x + off + 4);
      bf16x8 t;
      t[0] = (__bf16)(ok ? v0.x : 0.f); t[1] = (__bf16)(ok ? v0.y : 0.f);
      t[2] = (__bf16)(ok ? v0.z : 0.f); t[3] = (__bf16)(ok ? v0.w : 0.f);
      t[4] = (__bf16)(ok ? v1.x : 0.f); t[5] = (__bf16)(ok ? v1.y : 0.f);
      t[6] = (__bf16)(ok ? v1.z : 0.f); t[7] = (__bf16)(ok ? v1.w : 0.f);
      xf[nt][g] = __builtin_bit_cast(float4, t);
    } else {
      const float4 v = *reinterpret_cast<const float4*>(a.x + off);
      xf[nt][g] = make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
    }
    (void)km;
  }

  // ---- weight stream: (b0, chunk) sequence for this phase row ------------------------------------
  const size_t phase_f4 = (size_t)G * MT * 64;
  const float4* wbase = reinterpret_cast<const float4*>(a.wf) + (size_t)a0 * s * phase_f4;
  const int total = s * NCH;
  // Staging registers are NAMED scalars: an indexed array here (even fully unrolled) is left in scratch
  // memory by hipcc when it is written and read under separate `if (more)` branches.
  static_assert(NLD <= 9, "staging covers at most 9 x 16 bytes per thread");
  float4 st0, st1, st2, st3, st4, st5, st6, st7, st8;
#define FLM_FOR_ST(X) X(0, st0) X(1, st1) X(2, st2) X(3, st3) X(4, st4) X(5, st5) X(6, st6) X(7, st7) X(8, st8)
#define FLM_LD1(I, R)                                \
  if constexpr (I < NLD) {                           \
    const int idx = tid + 256 * I;                   \
    R = src_[idx < cnt_ ? idx : 0];                  \
  }
#define FLM_ST1(I, R)                                \
  if constexpr (I < NLD) {                           \
    const int idx = tid + 256 * I;                   \
    if (idx < CHUNK_F4) lds[bf_ * CHUNK_F4 + idx] = R; \
  }
#define FLM_ISSUE(SEQ)                                                                  \
  {                                                                                     \
    const int sq_ = (SEQ);                                                              \
    const int b0_ = sq_ / NCH, ch_ = sq_ % NCH;                                         \
    const int ng_ = (G - ch_ * GCH) < GCH ? (G - ch_ * GCH) : GCH;                      \
    const int cnt_ = ng_ * MT * 64;                                                     \
    const float4* src_ = wbase + (size_t)b0_ * phase_f4 + (size_t)ch_ * CHUNK_F4;      \
    FLM_FOR_ST(FLM_LD1)                                                                 \
  }
#define FLM_STASH(BUF)                                                                  \
  {                                                                                     \
    const int bf_ = (BUF);                                                              \
    FLM_FOR_ST(FLM_ST1)                                                                 \
  }

  FLM_ISSUE(0)
  FLM_STASH(0)
  __syncthreads();

  // ---- phase loop with a software-pipelined epilogue ------------------------------------------------
  // The epilogue of phase b0-1 (softmax: 20 exp + reductions + stores per lane at C=68) is cut in three
  // parts that are issued INSIDE the MFMA stream of phase b0 (one part per weight chunk), so its VALU
  // work runs in the shadow of the matrix pipe instead of after it.
  //   part 1: class maximum (in-lane + two xor-shuffles), e = exp(x - max)
  //   part 2: sum, one reciprocal, p = e * (1/sum)
  //   part 3: stores (probabilities / class map / raw + skip)
  f32x4 pv[NT][MT];  // previous phase's accumulators, transformed in place by the parts
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int m = 0; m < MT; ++m) pv[nt][m] = (f32x4){0.f, 0.f, 0.f, 0.f};

#define FLM_EPI_PART1()                                                                           \
  if (a.epilogue != 0) {                                                                          \
    _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) {                                           \
      float mx = -3.402823466e38f;                                                                \
      _Pragma("unroll") for (int m = 0; m < MT; ++m) _Pragma("unroll") for (int e = 0; e < 4; ++e) \
        if (16 * m + 4 * q + e < a.C) mx = fmaxf(mx, pv[nt][m][e]);                               \
      mx = fmaxf(mx, __shfl_xor(mx, 16));                                                         \
      mx = fmaxf(mx, __shfl_xor(mx, 32));                                                         \
      _Pragma("unroll") for (int m = 0; m < MT; ++m) _Pragma("unroll") for (int e = 0; e < 4; ++e) \
        pv[nt][m][e] = (16 * m + 4 * q + e < a.C) ? softmax_exp<BF>(pv[nt][m][e] - mx) : 0.f;     \
    }                                                                                             \
  }
#define FLM_EPI_PART2()                                                                           \
  if (a.epilogue != 0) {                                                                          \
    _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) {                                           \
      float sum = 0.f;                                                                            \
      _Pragma("unroll") for (int m = 0; m < MT; ++m) _Pragma("unroll") for (int e = 0; e < 4; ++e) \
        sum += pv[nt][m][e];                                                                      \
      sum += __shfl_xor(sum, 16);                                                                 \
      sum += __shfl_xor(sum, 32);                                                                 \
      const float rs = BF ? __builtin_amdgcn_rcpf(sum) : 1.0f / sum;                              \
      _Pragma("unroll") for (int m = 0; m < MT; ++m) _Pragma("unroll") for (int e = 0; e < 4; ++e) \
        pv[nt][m][e] = pv[nt][m][e] * rs;                                                         \
    }                                                                                             \
  }
#define FLM_EPI_PART3(B0)                                                                         \
  _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) {                                             \
    const int oy = s * i0[nt] + a0, ox = s * j0[nt] + (B0);                                       \
    const bool ovalid = pvalid[nt] && oy < a.ho && ox < a.wo;                                     \
    const size_t opix = ((size_t)img[nt] * a.ho + oy) * a.wo + ox;                                \
    if (a.epilogue == 0) {                                                                        \
      if (ovalid) {                                                                               \
        float* y = reinterpret_cast<float*>(a.y) + opix * a.ldy;                                  \
        _Pragma("unroll") for (int m = 0; m < MT; ++m) {                                          \
          const int c4 = 16 * m + 4 * q;                                                          \
          if (c4 < a.ldy) { /* ldy is a multiple of 4 here (score buffers, Cp channels) */        \
            float4 v = make_float4(pv[nt][m][0], pv[nt][m][1], pv[nt][m][2], pv[nt][m][3]);                       \
            if (a.skip) {                                                                         \
              const float4 sk = *reinterpret_cast<const float4*>(a.skip + opix * a.Cp + c4);     \
              v.x += sk.x; v.y += sk.y; v.z += sk.z; v.w += sk.w;                                 \
            }                                                                                     \
            *reinterpret_cast<float4*>(y + c4) = v;                                               \
          }                                                                                       \
        }                                                                                         \
      }                                                                                           \
    } else if (a.epilogue == 1) {                                                                 \
      if (ovalid) {                                                                               \
        float* y = reinterpret_cast<float*>(a.y) + opix * a.ldy;                                  \
        if ((a.ldy & 3) == 0) {                                                                   \
          _Pragma("unroll") for (int m = 0; m < MT; ++m) {                                        \
            const int c4 = 16 * m + 4 * q;                                                        \
            if (c4 < a.C)                                                                         \
              *reinterpret_cast<float4*>(y + c4) = make_float4(pv[nt][m][0], pv[nt][m][1], pv[nt][m][2], pv[nt][m][3]); \
          }                                                                                       \
        } else {                                                                                  \
          _Pragma("unroll") for (int m = 0; m < MT; ++m) _Pragma("unroll") for (int e = 0; e < 4; ++e) { \
            const int c = 16 * m + 4 * q + e;                                                     \
            if (c < a.C) y[c] = pv[nt][m][e];                                                         \
          }                                                                                       \
        }                                                                                         \
      }                                                                                           \
    } else {                                                                                      \
      /* argmax over classes, first maximum wins (numpy argmax, prediction.py:209) */             \
      float bv = -1.f;                                                                            \
      int bi = 0x7fffffff;                                                                        \
      _Pragma("unroll") for (int m = 0; m < MT; ++m) _Pragma("unroll") for (int e = 0; e < 4; ++e) { \
        const int c = 16 * m + 4 * q + e; /* ascending within a lane */                           \
        if (c < a.C && pv[nt][m][e] > bv) { bv = pv[nt][m][e]; bi = c; }                                  \
      }                                                                                           \
      _Pragma("unroll") for (int sh = 16; sh <= 32; sh <<= 1) {                                   \
        const float ov = __shfl_xor(bv, sh);                                                      \
        const int oi = __shfl_xor(bi, sh);                                                        \
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }                               \
      }                                                                                           \
      if (ovalid && q == 0) reinterpret_cast<int*>(a.y)[opix] = bi;                               \
    }                                                                                             \
  }

  int seq = 0;
  for (int b0 = 0; b0 < s; ++b0) {
    f32x4 acc[NT][MT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[nt][m] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      const bool more = seq + 1 < total;
      if (more) FLM_ISSUE(seq + 1)
      const float4* wl = lds + (seq & 1) * CHUNK_F4;
      // epilogue part of the PREVIOUS phase, issued alongside this chunk's MFMAs
      if (b0 > 0) {
        if (ch == 0) FLM_EPI_PART1()
        if (ch == (NCH > 1 ? 1 : 0)) FLM_EPI_PART2()
        if (ch == NCH - 1) FLM_EPI_PART3(b0 - 1)
      }
#pragma unroll
      for (int gl = 0; gl < GCH; ++gl) {
        const int g = ch * GCH + gl;  // compile-time
        if (g < G) {
          float4 af[MT];
#pragma unroll
          for (int m = 0; m < MT; ++m) af[m] = wl[(gl * MT + m) * 64 + lane];
          if constexpr (BF) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
              const bf16x8 xb = __builtin_bit_cast(bf16x8, xf[nt][g]);
#pragma unroll
              for (int m = 0; m < MT; ++m)
                acc[nt][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[m]), xb, acc[nt][m],
                                                                      0, 0, 0);
            }
          } else {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
              for (int m = 0; m < MT; ++m)
                acc[nt][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m].x, xf[nt][g].x, acc[nt][m], 0, 0, 0);
#pragma unroll
              for (int m = 0; m < MT; ++m)
                acc[nt][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m].y, xf[nt][g].y, acc[nt][m], 0, 0, 0);
#pragma unroll
              for (int m = 0; m < MT; ++m)
                acc[nt][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m].z, xf[nt][g].z, acc[nt][m], 0, 0, 0);
#pragma unroll
              for (int m = 0; m < MT; ++m)
                acc[nt][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m].w, xf[nt][g].w, acc[nt][m], 0, 0, 0);
            }
          }
        }
      }
      if (more) FLM_STASH((seq + 1) & 1)
      __syncthreads();
      ++seq;
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int m = 0; m < MT; ++m) pv[nt][m] = acc[nt][m];
  }
  // drain: the last phase's epilogue
  FLM_EPI_PART1()
  FLM_EPI_PART2()
  FLM_EPI_PART3(s - 1)
}

#undef FLM_EPI_PART1
#undef FLM_EPI_PART2
#undef FLM_EPI_PART3
#undef FLM_ISSUE
#undef FLM_STASH
#undef FLM_FOR_ST
#undef FLM_LD1
#undef FLM_ST1

template <int MT, int G, bool BF, int NT = 1>
static int launch_t(hipStream_t st, const ConvTArgs& a) {
  constexpr int GCH = BF ? GCH_BF16 : GCH_F32;
  constexpr size_t lds = sizeof(float4) * 2 * GCH * MT * 64;
  static bool attr_done = false;
  if (!attr_done) {
    FLM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&convt_kernel<MT, G, BF, NT>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_done = true;
  }
  dim3 grid(cdiv(a.P, 64 * NT), a.s);
  convt_kernel<MT, G, BF, NT><<<grid, 256, lds, st>>>(a);
  FLM_LAUNCH_CHECK("convt_kernel");
  return FLM_OK;
}

int launch_convt(hipStream_t st, const ConvTDesc& d) {
  ConvTArgs a;
  a.x = d.x; a.wf = d.wf; a.skip = d.skip; a.y = d.y;
  a.n = d.n; a.hi = d.hi; a.wi = d.wi; a.ho = d.ho; a.wo = d.wo; a.s = d.s; a.ldy = d.ldy;
  a.epilogue = d.epilogue; a.C = d.g.C; a.Cp = d.g.Cp;
  const long long P = (long long)d.n * (d.hi + 1) * (d.wi + 1);
  if (P <= 0 || P > (1ll << 30) || d.ho > d.s * (d.hi + 1) || d.wo > d.s * (d.wi + 1)) {
    set_error("convt: bad geometry n=%d in=%dx%d out=%dx%d s=%d", d.n, d.hi, d.wi, d.ho, d.wo, d.s);
    return FLM_ERR_SHAPE;
  }
  a.P = (int)P;
  if (d.epilogue == 0 && (d.ldy & 3)) {
    set_error("convt: raw epilogue needs a channel stride that is a multiple of 4");
    return FLM_ERR_SHAPE;
  }
  if (d.g.bf16) {
    // two pixel tiles per wave: at 16x the matrix rate the phase weights (45 KiB per 64 positions) are the
    // stream to economise
    if (d.g.C == 68 && d.g.G == 9) return launch_t<5, 9, true, 2>(st, a);
    switch (d.g.MT) {
      case 1: return launch_t<1, 2, true>(st, a);
      case 2: return launch_t<2, 4, true>(st, a);
      case 3: return launch_t<3, 6, true>(st, a);
      case 4: return launch_t<4, 8, true>(st, a);
      case 5: return launch_t<5, 10, true>(st, a);
      case 6: return launch_t<6, 12, true>(st, a);
    }
  } else {
    if (d.g.C == 68 && d.g.G == 17) return launch_t<5, 17, false>(st, a);
    switch (d.g.MT) {
      case 1: return launch_t<1, 4, false>(st, a);
      case 2: return launch_t<2, 8, false>(st, a);
      case 3: return launch_t<3, 12, false>(st, a);
      case 4: return launch_t<4, 16, false>(st, a);
      case 5: return launch_t<5, 20, false>(st, a);
      case 6: return launch_t<6, 24, false>(st, a);
    }
  }
  set_error("convt: n_classes %d not supported (max %d)", d.g.C, kMaxClasses);
  return FLM_ERR_SHAPE;
}

}  // namespace flm
