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
//
// Landmark mode without the probability tensor (template CAND, epilogue 3).  The top-n decode
// (utils/metrics.py:66-77) needs, per face and class, only the n largest probabilities of the 264x264 map:
//   1. a sampling launch computes R of the 64 phases per tile of input positions (R/64 of the work; which
//      phases is a fixed function of the tile index, spread over all 64: the phases are separate filters, so
//      a sample from one phase alone is not representative of the map).  It writes no map either: every
//      wave keeps the per-class maximum of its 16*NT pixels of every sampled phase (epilogue 4), and
//      cand_tau_kernel (flm_decode.hip) takes tau[face][class] = the n-th largest of the face's 144..288 maxima.  Those
//      are values of n distinct pixels of the full map, so at least n pixels are >= tau and every member of
//      the true top n is;
//   2. the full launch keeps its probabilities in registers and appends (value, class, pixel) keys of the
//      pixels with p >= tau (and p > 0: zero weights cannot move a centroid) to an LDS list -- about 64*n of
//      the 69,696 pixels per class -- which the workgroup flushes to its face's list with one atomic;
//   3. cand_merge_kernel (flm_decode.hip) selects the exact top n from the list, same keys and tie rule as
//      the decode of the materialised map.
// A list that overflows (flat maps: everything ties with tau) raises a flag; the materialising launch and
// the ordinary decode follow in the stream, gated on that flag, so the result is exact in every case.
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
  int ppf;  // > 0: positions per face padded to a multiple of the workgroup's tile (a workgroup never spans two faces)
  int ls;   // log2(s): the strides of the reference's decoders are 2, 8 and 32
  int share;  // packed weights use the shared tile-4 layout (convt_share_layout): 68-class kernels, s % 4 == 0;
              // the main launches then run the SHARE = true instantiation (template parameter)
  int nb;   // phases b0 computed per phase row (s, or 1 for the sub-sampled launch)
  int sub;  // > 0: sampling launch: a workgroup computes `sub` phases chosen from its tile index (not a phase row);
            // epilogue 1 writes them compactly (pixel index (r, i0, j0) on a sub x (hi+1) x (wi+1) grid), epilogue 4
            // only the per-wave class maxima: y = unsigned [n][4 * tiles per face][16*MT] float bit patterns
  const float* tau;           // [n][C] candidate thresholds (epilogue 3)
  unsigned long long* cand;   // [n][cand_cap] keys: order_bits(p) << 32 | class << 17 | pixel
  unsigned* cand_cnt;         // [n] entries appended per face; cand_cnt[n] = overflow flag
  int cand_cap;
  const unsigned* gate;       // non-null: the launch does nothing unless *gate != 0
};

// candidate keys one wave can hold in LDS: 128 (fp32) or 256 (bf16, NT = 2) pixels x 68 classes pass through it;
// the fp32 kernel's 61 KiB weight ring leaves room for 512 per wave if two workgroups are to share a CU
#define kCandWaveCap (512 * NT)
constexpr int kMaxSamplePhases = 16;

__device__ __forceinline__ unsigned cand_order_bits(float v) {
  const unsigned u = __float_as_uint(v);
  return u ^ ((u >> 31) ? 0xffffffffu : 0x80000000u);
}

constexpr int GCH_F32 = 6, GCH_BF16 = 3;  // k groups per LDS chunk (bf16: smaller chunks, fewer staging registers)

// exp(t) for t <= 0 in the softmax.  fp32 path: the accurate library expf.  bf16 path: v_exp_f32 on
// t*log2(e) (about 1e-6 relative, far below the bf16 rounding the logits already carry); at 16x the MFMA
// rate the 20 accurate expf per lane per phase would cost more than the phase's matrix work.
// x: logit, mx: the pixel's maximum, nmxl = -mx * log2(e).  bf16: one fma + v_exp_f32.
template <bool BF>
__device__ __forceinline__ float softmax_exp(float x, float mx, float nmxl) {
  if constexpr (BF) return __builtin_amdgcn_exp2f(__builtin_fmaf(x, 1.44269504088896340736f, nmxl));
  else return expf(x - mx);
}
// max without the quiet-NaN canonicalisation fmaxf() drags in (two extra v_max per call on MFMA results);
// NaN logits give NaN probabilities either way.
__device__ __forceinline__ float max3_raw(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ float max_raw(float a, float b) {
  float r;
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

// Reductions over the four lane groups q = lane >> 4 (the 16*MT result rows of one pixel sit in lanes r, r+16, r+32,
// r+48).  v_permlane16_swap / v_permlane32_swap (gfx950) exchange 16- and 32-lane rows between two registers in the
// VALU: with the same value in both, {dst, src} come back as {[x0,x0,x2,x2], [x1,x1,x3,x3]} and {[lo,lo], [hi,hi]},
// so one swap + one max / add is the xor-16 / xor-32 butterfly step -- no ds_bpermute round trip, no lane-index
// arithmetic, no lgkmcnt(0) that would also drain the weight-fragment reads in flight.  Same operand pairs as the
// xor shuffles they replace (a + b in one lane, b + a in its partner): same bits.
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float reduce_q_max(float v) {
  u32x2 t = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = max_raw(__uint_as_float(t.x), __uint_as_float(t.y));
  t = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return max_raw(__uint_as_float(t.x), __uint_as_float(t.y));
}
__device__ __forceinline__ float reduce_q_sum(float v) {
  u32x2 t = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(t.x) + __uint_as_float(t.y);
  t = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(t.x) + __uint_as_float(t.y);
}

// Maximum over the 16 lanes r = lane & 15 of one lane group (the wave's 16 pixels of one class), in every lane of the
// group: four DPP steps in the VALU -- quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror (after
// the first two a quad is uniform, so the mirrors pair quads and then halves) -- instead of four ds_bpermute shuffles
// with their lane-index arithmetic and LDS round trips (34 values per wave and sampled phase).
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float reduce_r_max(float v) {
  v = max_raw(v, dpp_mov<0xB1>(v));
  v = max_raw(v, dpp_mov<0x4E>(v));
  v = max_raw(v, dpp_mov<0x141>(v));
  return max_raw(v, dpp_mov<0x140>(v));
}

// MODE: 0 = epilogues 0/1/2 (maps), 1 = epilogue 3 (top-n candidates), 2 = epilogue 4 (sampling launch: wave maxima)
template <int MT, int G, bool BF, int NT, int MODE, bool SHARE>
__global__ __launch_bounds__(256, (!BF && MT >= 5 && G >= 20) ? 1 : 2) void convt_kernel(ConvTArgs a) {
  constexpr bool CAND = MODE == 1;
  constexpr bool SAMPLE = MODE == 2;
  // The 68-class kernels know their class count, and their packed weights put classes 64..67 on rows 0, 4, 8, 12 of
  // the last 16-row tile (flm_pack.hip): result row 4q + e of a tile sits in register e of lane group q, so every
  // lane holds 16 + 1 classes -- which (tile, register) slots carry a class is known at compile time.
  constexpr bool C68 = (MT == 5 && (G == 9 || G == 17));
#define FLM_CVALID(M, E) (C68 ? ((M) < 4 || (E) == 0) : (16 * (M) + 4 * q + (E) < a.C))
#define FLM_CLS(M, E) ((C68 && (M) == 4) ? (64 + q) : (16 * (M) + 4 * q + (E)))
  if (a.gate && *a.gate == 0) return;
  constexpr int GCH = BF ? GCH_BF16 : GCH_F32;
  constexpr int NCH = (G + GCH - 1) / GCH;
  constexpr int CHUNK_F4 = GCH * MT * 64;             // float4 per full chunk
  constexpr int NLD = (CHUNK_F4 + 255) / 256;         // staging loads per thread
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float4* lds = reinterpret_cast<float4*>(smem_raw);  // [2][CHUNK_F4]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  // candidate keys: one private LDS region per wave, filled through a wave-uniform counter (no atomics)
  // (bf16 68-class kernels first keep the shared fifth tile's group sums here: [2 groups][NT][64 lanes] x 16 bytes per wave)
  constexpr size_t X4_BYTES = (BF && MT == 5 && G == 9) ? sizeof(float4) * 4 * 2 * NT * 64 : 0;
  float4* x4s = reinterpret_cast<float4*>(smem_raw + sizeof(float4) * 2 * CHUNK_F4) + wave * 2 * NT * 64;
  unsigned long long* cwave = reinterpret_cast<unsigned long long*>(smem_raw + sizeof(float4) * 2 * CHUNK_F4 + X4_BYTES) + wave * kCandWaveCap;
  float* tau_s = reinterpret_cast<float*>(smem_raw + sizeof(float4) * 2 * CHUNK_F4 + X4_BYTES + sizeof(unsigned long long) * 4 * kCandWaveCap);  // [16*MT]
  unsigned wcnt = 0;
  // sampling launch (epilogue 4): per-wave class maxima as float bit patterns (p >= 0: unsigned order = float order)
  unsigned* wmax = reinterpret_cast<unsigned*>(smem_raw + sizeof(float4) * 2 * CHUNK_F4 + X4_BYTES +
                                               (CAND ? sizeof(unsigned long long) * 4 * kCandWaveCap + sizeof(float) * 16 * MT : 0)) +
                   wave * kMaxSamplePhases * 16 * MT;  // [phase of the tile's list][result row]
  if (SAMPLE)
    for (int c = lane; c < a.sub * 16 * MT; c += 64) wmax[c] = 0u;

  const int s = a.s;
  // Phase (a0, b0) of iteration IT of this workgroup: phase row blockIdx.y, b0 = IT; in the sampling launch the
  // IT-th entry of the tile's list -- an odd stride walks all s*s phases before repeating.
  const int tile_pf = a.sub ? (int)(blockIdx.x % (a.ppf / (64 * NT))) : 0;
  // Shared tile-4 layout (template SHARE, packed weights per convt_share_layout): the four extra classes 64..67 of FOUR consecutive phases b0 = 4gb .. 4gb+3 are
  // rows 4q + j of the leader phase's tile 4 (X is the same for every phase, only the filter differs), so the short
  // phases 4gb+1..3 multiply and read 4 tiles instead of 5: 15 % fewer MFMAs.  Their fifth tile in the stream still
  // holds their own classes 64..67 (rows 4q): the sampling launch multiplies all five tiles of whatever phase it draws,
  // so its probabilities are the main launch's bit for bit.
#define FLM_PHASE(IT)                                                                                    \
  (a.sub ? (((tile_pf * a.sub + (IT)) * 23 + 5) & (s * s - 1)) : ((int)blockIdx.y * s + (IT)))


  // ---- this lane's NT input positions (NT pixel tiles of 16 per wave: the phase's weights, streamed once
  //      per workgroup, then serve 64*NT positions) --------------------------------------------------
  const int wi1 = a.wi + 1, hi1 = a.hi + 1;
  bool pvalid[NT];
  int i0[NT], j0[NT], img[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int p = (blockIdx.x * 4 + wave) * 16 * NT + nt * 16 + r;
    if (a.ppf > 0) {
      const int pl = p % a.ppf;
      img[nt] = p / a.ppf;
      pvalid[nt] = pl < wi1 * hi1 && img[nt] < a.n;
      const int pp = pvalid[nt] ? pl : 0;
      if (!pvalid[nt]) img[nt] = 0;
      j0[nt] = pp % wi1;
      i0[nt] = pp / wi1;
    } else {
      pvalid[nt] = p < a.P;
      const int pp = pvalid[nt] ? p : 0;
      j0[nt] = pp % wi1;
      i0[nt] = (pp / wi1) % hi1;
      img[nt] = pp / (wi1 * hi1);
    }
  }
  // candidate mode: thresholds of this workgroup's face for the 4*MT classes of this lane (16m + 4q + e)
  const int wg_img = CAND ? (blockIdx.x * 64 * NT) / (a.ppf > 0 ? a.ppf : 1) : 0;
  if (CAND) {
    // thresholds of this workgroup's face, clamped to FLT_MIN so that p >= tau implies p > 0 (zero weights cannot
    // move a centroid; a class left with fewer than n keys is caught by cand_merge_kernel)
    if (tid < 16 * MT) {  // indexed by result row: class of row p
      const int cls = (C68 && tid >= 64) ? (((tid & 3) == 0) ? 64 + ((tid - 64) >> 2) : a.C) : tid;
      tau_s[tid] = (cls < a.C && wg_img < a.n) ? fmaxf(a.tau[(size_t)wg_img * a.C + cls], 1.17549435e-38f) : 3.402823466e38f;
    }
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
  // raw buffer loads: per-thread offset tid*16 is a constant VGPR, phase / chunk / 4 KiB-step ride in the scalar
  // offset (no vector address arithmetic per load); reads past the packed array return zeros
  const __amdgpu_buffer_rsrc_t wsrd = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<void*>(a.wf), 0, (int)((size_t)s * s * phase_f4 * 16 < 0x7fffffffull ? (size_t)s * s * phase_f4 * 16 : 0x7fffffffull),
      0x00020000);
  const unsigned wvoff = (unsigned)tid * 16u;
  const int total = a.nb * NCH;
  // Staging registers are NAMED scalars: an indexed array here (even fully unrolled) is left in scratch
  // memory by hipcc when it is written and read under separate `if (more)` branches.
  static_assert(NLD <= 9, "staging covers at most 9 x 16 bytes per thread");
  float4 st0, st1, st2, st3, st4, st5, st6, st7, st8;
#define FLM_FOR_ST(X) X(0, st0) X(1, st1) X(2, st2) X(3, st3) X(4, st4) X(5, st5) X(6, st6) X(7, st7) X(8, st8)
#define FLM_LD1(I, R)                                \
  if constexpr (I < NLD)                             \
    R = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(wsrd, wvoff, soff_ + 4096 * I, 0));
#define FLM_ST1(I, R)                                \
  if constexpr (I < NLD) {                           \
    const int idx = tid + 256 * I;                   \
    if (idx < CHUNK_F4) lds[bf_ * CHUNK_F4 + idx] = R; \
  }
#define FLM_ISSUE(SEQ)                                                                  \
  {                                                                                     \
    const int sq_ = (SEQ);                                                              \
    const int b0_ = sq_ / NCH, ch_ = sq_ % NCH;                                         \
    /* (the last chunk of a phase is shorter: its tail reads the next phase's first groups, never used) */ \
    const int soff_ = (int)(((size_t)FLM_PHASE(b0_) * phase_f4 + (size_t)ch_ * CHUNK_F4) * 16);  \
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
  // Two accumulator sets: phase b0 accumulates into one while the parts transform the other (phase b0-1) in place.
  f32x4 accA[NT][MT], accB[NT][MT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int m = 0; m < MT; ++m) accA[nt][m] = accB[nt][m] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // Candidate keys of this wave: LDS region -> the face's global list (one atomic reserves the range).  Runs at the
  // end of the kernel and whenever the region would overflow (many classes peaking in the wave's 128 / 256 pixels).
#define FLM_CAND_FLUSH()                                                                          \
  {                                                                                               \
    const unsigned found_ = __builtin_amdgcn_readfirstlane(wcnt);                                 \
    if (wg_img < a.n && found_) {                                                                 \
      unsigned base_ = 0;                                                                         \
      if (lane == 0) {                                                                            \
        base_ = atomicAdd(&a.cand_cnt[wg_img], found_);                                           \
        if (base_ + found_ > (unsigned)a.cand_cap) atomicOr(&a.cand_cnt[a.n], 1u);                \
      }                                                                                           \
      base_ = __builtin_amdgcn_readfirstlane(base_);                                              \
      for (unsigned i_ = lane; i_ < found_; i_ += 64)                                             \
        if (base_ + i_ < (unsigned)a.cand_cap) a.cand[(size_t)wg_img * a.cand_cap + base_ + i_] = cwave[i_]; \
    }                                                                                             \
    wcnt = 0;                                                                                     \
  }
#define FLM_EPI_PART1(pv)                                                                         \
  if (a.epilogue != 0) {                                                                          \
    _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) {                                           \
      float mx = -3.402823466e38f;                                                                \
      _Pragma("unroll") for (int m = 0; m < MT; ++m) {                                            \
        const float x0 = FLM_CVALID(m, 0) ? pv[nt][m][0] : -3.402823466e38f;                      \
        const float x1 = FLM_CVALID(m, 1) ? pv[nt][m][1] : -3.402823466e38f;                      \
        const float x2 = FLM_CVALID(m, 2) ? pv[nt][m][2] : -3.402823466e38f;                      \
        const float x3 = FLM_CVALID(m, 3) ? pv[nt][m][3] : -3.402823466e38f;                      \
        mx = max3_raw(mx, x0, x1);                                                                \
        mx = max3_raw(mx, x2, x3);                                                                \
      }                                                                                           \
      mx = reduce_q_max(mx);                                                                      \
      const float nmxl = -mx * 1.44269504088896340736f;                                           \
      _Pragma("unroll") for (int m = 0; m < MT; ++m) _Pragma("unroll") for (int e = 0; e < 4; ++e) \
        pv[nt][m][e] = FLM_CVALID(m, e) ? softmax_exp<BF>(pv[nt][m][e], mx, nmxl) : 0.f;          \
    }                                                                                             \
  }
#define FLM_EPI_PART2(pv, EPI_B0)                                                                 \
  { const int epi_b0 = (EPI_B0);                                                                   \
  if (a.epilogue != 0) {                                                                          \
    _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) {                                           \
      float sum = 0.f;                                                                            \
      _Pragma("unroll") for (int m = 0; m < MT; ++m) _Pragma("unroll") for (int e = 0; e < 4; ++e) \
        sum += pv[nt][m][e];                                                                      \
      sum = reduce_q_sum(sum);                                                                    \
      float rs = BF ? __builtin_amdgcn_rcpf(sum) : 1.0f / sum;                                    \
      if (CAND && a.epilogue == 3) {                                                              \
        const int ph_ = FLM_PHASE(epi_b0);                                                        \
        const int oy_ = s * i0[nt] + (ph_ >> a.ls), ox_ = s * j0[nt] + (ph_ & (s - 1));                      \
        if (!(pvalid[nt] && oy_ < a.ho && ox_ < a.wo)) rs = 0.f;                                  \
      }                                                                                           \
      _Pragma("unroll") for (int m = 0; m < MT; ++m) _Pragma("unroll") for (int e = 0; e < 4; ++e) \
        pv[nt][m][e] = pv[nt][m][e] * rs;                                                         \
    }                                                                                             \
  } }
#define FLM_EPI_PART3(pv, B0)                                                                     \
  _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) {                                             \
    const int ph3_ = FLM_PHASE(B0);                                                               \
    const int oy = s * i0[nt] + (ph3_ >> a.ls), ox = s * j0[nt] + (ph3_ & (s - 1));                         \
    const bool ovalid = pvalid[nt] && oy < a.ho && ox < a.wo;                                     \
    if (SAMPLE) {                                                                                 \
      /* class maxima over the wave's 16 pixels (lanes r of one q), then one LDS max per class */  \
      _Pragma("unroll") for (int m = 0; m < MT; ++m) _Pragma("unroll") for (int e = 0; e < 4; ++e) \
       if (FLM_CVALID(m, e)) {                                                                    \
        float v = reduce_r_max(ovalid ? pv[nt][m][e] : 0.f);                                      \
        if (r == 0) atomicMax(&wmax[(B0) * 16 * MT + 16 * m + 4 * q + e], __float_as_uint(v));     \
      }                                                                                           \
    } else if (CAND) {                                                                            \
      /* invalid pixels carry p = 0 (part 2), thresholds are >= FLT_MIN: one compare per value decides */ \
      const unsigned pixbits = (unsigned)(oy * a.wo + ox);                                        \
      _Pragma("unroll") for (int m = 0; m < MT; ++m) {                                            \
        const float4 tq = *reinterpret_cast<const float4*>(tau_s + 16 * m + 4 * q);               \
        unsigned long long mk[4];                                                                 \
        mk[0] = __ballot(pv[nt][m][0] >= tq.x);                                                   \
        mk[1] = FLM_CVALID(m, 1) ? __ballot(pv[nt][m][1] >= tq.y) : 0ull;                         \
        mk[2] = FLM_CVALID(m, 2) ? __ballot(pv[nt][m][2] >= tq.z) : 0ull;                         \
        mk[3] = FLM_CVALID(m, 3) ? __ballot(pv[nt][m][3] >= tq.w) : 0ull;                         \
        if (mk[0] | mk[1] | mk[2] | mk[3]) { /* wave-uniform, taken for about one m in eight */   \
          /* room for the up to 4 x 64 keys of this group (one copy of the flush per group: the kernel's code   */ \
          /* already exceeds the instruction cache)                                                              */ \
          if (wcnt + 256u > (unsigned)kCandWaveCap) FLM_CAND_FLUSH()                              \
          _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                         \
            if (mk[e]) {                                                                          \
              const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(mk[e] >> 32),            \
                                                              __builtin_amdgcn_mbcnt_lo((unsigned)mk[e], 0u)); \
              const unsigned slot = wcnt + rank;                                                  \
              if (((mk[e] >> lane) & 1ull) && slot < (unsigned)kCandWaveCap)                      \
                cwave[slot] = ((unsigned long long)cand_order_bits(pv[nt][m][e]) << 32) |         \
                              ((unsigned long long)FLM_CLS(m, e) << 17) | pixbits;                \
              wcnt += __builtin_popcountll(mk[e]);                                                \
            }                                                                                     \
          }                                                                                       \
        }                                                                                         \
      }                                                                                           \
    } else {                                                                                      \
    const size_t opix = ((size_t)img[nt] * a.ho + oy) * a.wo + ox;                                \
    if (a.epilogue == 0) {                                                                        \
      if (ovalid) {                                                                               \
        float* y = reinterpret_cast<float*>(a.y) + opix * a.ldy;                                  \
        _Pragma("unroll") for (int m = 0; m < MT; ++m) {                                          \
          const int c4 = 16 * m + 4 * q;                                                          \
          if (C68 && m == 4) { /* one class per lane group + the zero pad channels of the Cp-wide buffers */ \
            float v = pv[nt][m][0];                                                               \
            if (a.skip) v += a.skip[opix * a.Cp + 64 + q];                                        \
            y[64 + q] = v;                                                                        \
            if (68 + q < a.ldy) y[68 + q] = 0.f;                                                  \
          } else                                                                                  \
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
            if (C68 && m == 4) y[64 + q] = pv[nt][m][0];                                          \
            else if (FLM_CVALID(m, 0))                                                            \
              *reinterpret_cast<float4*>(y + c4) = make_float4(pv[nt][m][0], pv[nt][m][1], pv[nt][m][2], pv[nt][m][3]); \
          }                                                                                       \
        } else {                                                                                  \
          _Pragma("unroll") for (int m = 0; m < MT; ++m) _Pragma("unroll") for (int e = 0; e < 4; ++e) { \
            const int c = FLM_CLS(m, e);                                                          \
            if (FLM_CVALID(m, e)) y[c] = pv[nt][m][e];                                            \
          }                                                                                       \
        }                                                                                         \
      }                                                                                           \
    } else {                                                                                      \
      /* argmax over classes, first maximum wins (numpy argmax, prediction.py:209) */             \
      float bv = -1.f;                                                                            \
      int bi = 0x7fffffff;                                                                        \
      _Pragma("unroll") for (int m = 0; m < MT; ++m) _Pragma("unroll") for (int e = 0; e < 4; ++e) { \
        const int c = FLM_CLS(m, e); /* ascending within a lane */                                \
        if (FLM_CVALID(m, e) && pv[nt][m][e] > bv) { bv = pv[nt][m][e]; bi = c; }                 \
      }                                                                                           \
      _Pragma("unroll") for (int sh = 16; sh <= 32; sh <<= 1) {                                   \
        const float ov = __shfl_xor(bv, sh);                                                      \
        const int oi = __shfl_xor(bi, sh);                                                        \
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }                               \
      }                                                                                           \
      if (ovalid && q == 0) reinterpret_cast<int*>(a.y)[opix] = bi;                               \
    }                                                                                             \
    }                                                                                             \
  }

  int seq = 0;
  // One phase: MFMAs into set ACC while the three parts finish the previous phase held in set PV.
#define FLM_PHASE_BODY(ACC, PV, B0, MTP, FILL, PJ, PPAR)                                            \
  {                                                                                                 \
    const int b0 = (B0);                                                                            \
    _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) _Pragma("unroll") for (int m = 0; m < MT; ++m) \
      ACC[nt][m] = (f32x4){0.f, 0.f, 0.f, 0.f};                                                     \
    if (FILL) { /* the previous phase was a short one: its class 64+q value waits in the group's tile-4 sums */ \
      _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) {                                           \
        if constexpr (BF) {                                                                         \
          PV[nt][MT - 1][0] = reinterpret_cast<const float*>(x4s + ((PPAR) * NT + nt) * 64 + lane)[PJ]; \
        } else {                                                                                    \
          const f32x4 xs_ = (PPAR) ? x4g[1][nt] : x4g[0][nt];                                       \
          PV[nt][MT - 1][0] = xs_[PJ];                                                              \
        }                                                                                           \
      }                                                                                             \
    }                                                                                               \
    _Pragma("unroll") for (int ch = 0; ch < NCH; ++ch) {                                            \
      const bool more = seq + 1 < total;                                                            \
      if (more) FLM_ISSUE(seq + 1)                                                                  \
      const float4* wl = lds + (seq & 1) * CHUNK_F4;                                                \
      if (b0 > 0) {                                                                                 \
        if (ch == 0) FLM_EPI_PART1(PV)                                                              \
        if (ch == (NCH > 1 ? 1 : 0)) FLM_EPI_PART2(PV, b0 - 1)                                      \
        if (ch == NCH - 1) FLM_EPI_PART3(PV, b0 - 1)                                                \
      }                                                                                             \
      _Pragma("unroll") for (int gl = 0; gl < GCH; ++gl) {                                          \
        const int g = ch * GCH + gl; /* compile-time */                                             \
        if (g < G) {                                                                                \
          float4 af[MT];                                                                            \
          _Pragma("unroll") for (int m = 0; m < (MTP); ++m) af[m] = wl[(gl * MT + m) * 64 + lane];  \
          if constexpr (BF) {                                                                       \
            _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) {                                     \
              const bf16x8 xb = __builtin_bit_cast(bf16x8, xf[nt][g]);                              \
              _Pragma("unroll") for (int m = 0; m < (MTP); ++m)                                     \
                ACC[nt][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[m]), xb, ACC[nt][m], 0, 0, 0); \
            }                                                                                       \
          } else {                                                                                  \
            _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) {                                     \
              _Pragma("unroll") for (int m = 0; m < (MTP); ++m)                                     \
                ACC[nt][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m].x, xf[nt][g].x, ACC[nt][m], 0, 0, 0); \
              _Pragma("unroll") for (int m = 0; m < (MTP); ++m)                                     \
                ACC[nt][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m].y, xf[nt][g].y, ACC[nt][m], 0, 0, 0); \
              _Pragma("unroll") for (int m = 0; m < (MTP); ++m)                                     \
                ACC[nt][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m].z, xf[nt][g].z, ACC[nt][m], 0, 0, 0); \
              _Pragma("unroll") for (int m = 0; m < (MTP); ++m)                                     \
                ACC[nt][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m].w, xf[nt][g].w, ACC[nt][m], 0, 0, 0); \
            }                                                                                       \
          }                                                                                         \
        }                                                                                           \
      }                                                                                             \
      if (more) FLM_STASH((seq + 1) & 1)                                                            \
      __syncthreads();                                                                              \
      ++seq;                                                                                        \
    }                                                                                               \
  }
  constexpr bool SHARE_OK = C68;         // kernels that may meet the shared tile-4 layout
  f32x4 x4g[2][BF ? 1 : NT];             // fp32: tile-4 sums of the current and the previous group of four phases (bf16: LDS)
#pragma unroll
  for (int nt = 0; nt < (BF ? 1 : NT); ++nt) x4g[0][nt] = x4g[1][nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if constexpr (SHARE_OK && SHARE && !SAMPLE) {
    // groups of four phases: leader (5 tiles, keeps the group's tile-4 sums), then three short phases
    for (int g4 = 0; g4 < (a.nb >> 2); ++g4) {
      const int par = g4 & 1;
      FLM_PHASE_BODY(accA, accB, 4 * g4, MT, (g4 > 0), 3, (par ^ 1))
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        if constexpr (BF) {
          x4s[(par * NT + nt) * 64 + lane] = make_float4(accA[nt][MT - 1][0], accA[nt][MT - 1][1], accA[nt][MT - 1][2], accA[nt][MT - 1][3]);
        } else {
          if (par) x4g[1][nt] = accA[nt][MT - 1];
          else x4g[0][nt] = accA[nt][MT - 1];
        }
      }
      FLM_PHASE_BODY(accB, accA, 4 * g4 + 1, MT - 1, false, 0, par)
      FLM_PHASE_BODY(accA, accB, 4 * g4 + 2, MT - 1, true, 1, par)
      FLM_PHASE_BODY(accB, accA, 4 * g4 + 3, MT - 1, true, 2, par)
    }
    {  // drain: phase nb-1 is the third short phase of the last group (set B)
      const int par = ((a.nb >> 2) - 1) & 1;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        if constexpr (BF) {
          accB[nt][MT - 1][0] = reinterpret_cast<const float*>(x4s + (par * NT + nt) * 64 + lane)[3];
        } else {
          const f32x4 xs_ = par ? x4g[1][nt] : x4g[0][nt];
          accB[nt][MT - 1][0] = xs_[3];
        }
      }
      FLM_EPI_PART1(accB)
      FLM_EPI_PART2(accB, a.nb - 1)
      FLM_EPI_PART3(accB, a.nb - 1)
    }
  } else {
  for (int bb = 0; bb < a.nb; bb += 2) {  // (a sampling launch multiplies all five tiles: a short phase's fifth is unused)
    FLM_PHASE_BODY(accA, accB, bb, MT, false, 0, 0)
    if (bb + 1 < a.nb) FLM_PHASE_BODY(accB, accA, bb + 1, MT, false, 0, 0)
  }
  // drain: the last phase's epilogue (even phase indices accumulate in set A)
  if (a.nb & 1) {
    FLM_EPI_PART1(accA)
    FLM_EPI_PART2(accA, a.nb - 1)
    FLM_EPI_PART3(accA, a.nb - 1)
  } else {
    FLM_EPI_PART1(accB)
    FLM_EPI_PART2(accB, a.nb - 1)
    FLM_EPI_PART3(accB, a.nb - 1)
  }
  }
#undef FLM_PHASE_BODY
  if (SAMPLE) {
    // (same-wave LDS atomics and reads are ordered; other waves never touch this region)
    const int tiles_pf = a.ppf / (64 * NT);
    const int face = blockIdx.x / tiles_pf;
    // one slot of 16*MT values per (wave, sampled phase), in class order
    unsigned* dst = reinterpret_cast<unsigned*>(a.y) + (((size_t)face * tiles_pf + tile_pf) * 4 + wave) * a.sub * (16 * MT);
    if (face < a.n)
      for (int i = lane; i < a.sub * 16 * MT; i += 64) {
        const int it = i / (16 * MT), c = i % (16 * MT);
        dst[i] = c < a.C ? wmax[it * 16 * MT + ((C68 && c >= 64) ? 64 + 4 * (c - 64) : c)] : 0u;
      }
  }
  if (CAND && a.epilogue == 3) FLM_CAND_FLUSH()
}

#undef FLM_PHASE
#undef FLM_CVALID
#undef FLM_CLS
#undef FLM_CAND_FLUSH
#undef FLM_EPI_PART1
#undef FLM_EPI_PART2
#undef FLM_EPI_PART3
#undef FLM_ISSUE
#undef FLM_STASH
#undef FLM_FOR_ST
#undef FLM_LD1
#undef FLM_ST1

template <int MT, int G, bool BF, int NT = 1, int MODE = 0, bool SHARE = false>
static int launch_t(hipStream_t st, ConvTArgs a) {
  constexpr bool CAND = MODE == 1;
  constexpr int GCH = BF ? GCH_BF16 : GCH_F32;
  constexpr size_t lds = sizeof(float4) * 2 * GCH * MT * 64 + ((BF && MT == 5 && G == 9) ? sizeof(float4) * 4 * 2 * NT * 64 : 0) +
                         (CAND ? sizeof(unsigned long long) * 4 * kCandWaveCap + sizeof(float) * 16 * MT : 0) +
                         (MODE == 2 ? sizeof(unsigned) * 4 * kMaxSamplePhases * 16 * MT : 0);
  // two workgroups per CU (160 KiB of LDS) is what the 68-class kernels are scheduled for: a key buffer that pushed the
  // fp32 candidate kernel to 94 KiB cost 22 % of up3
  static_assert(!(MT == 5 && (G == 9 || G == 17)) || MODE == 2 || lds <= 80 * 1024, "convt: LDS budget of two workgroups per CU");
  static FuncAttrOnce attr;
  FLM_FUNC_ATTR_ONCE(attr, (&convt_kernel<MT, G, BF, NT, MODE, SHARE>), lds);
  int xblocks = cdiv(a.P, 64 * NT);
  if (a.ppf > 0) {  // per-face padding: a workgroup's 64*NT positions belong to one face
    a.ppf = cdiv((a.hi + 1) * (a.wi + 1), 64 * NT) * 64 * NT;
    xblocks = a.n * (a.ppf / (64 * NT));
  }
  dim3 grid(xblocks, a.sub ? 1 : a.s);
  convt_kernel<MT, G, BF, NT, MODE, SHARE><<<grid, 256, lds, st>>>(a);
  FLM_LAUNCH_CHECK("convt_kernel");
  return FLM_OK;
}

// Maxima written by the sampling launch per face (epilogue 4): one per wave (4 per tile of 64*NT positions) and
// sampled phase.
int convt_sample_slots(const ConvTGeom& g, int hi, int wi, int sub) {
  const int nt = g.bf16 ? 2 : 1;
  return 4 * cdiv((hi + 1) * (wi + 1), 64 * nt) * sub;
}

int convt_candidates_supported(const ConvTGeom& g) {
  return g.C == 68 && ((g.bf16 && g.G == 9) || (!g.bf16 && g.G == 17));
}

int launch_convt(hipStream_t st, const ConvTDesc& d) {
  ConvTArgs a;
  a.x = d.x; a.wf = d.wf; a.skip = d.skip; a.y = d.y;
  a.n = d.n; a.hi = d.hi; a.wi = d.wi; a.ho = d.ho; a.wo = d.wo; a.s = d.s; a.ldy = d.ldy;
  a.epilogue = d.epilogue; a.C = d.g.C; a.Cp = d.g.Cp;
  a.sub = d.sub; a.nb = d.sub ? d.sub : d.s;
  a.share = convt_share_layout(d.g, d.s);
  a.ls = 0;
  while ((1 << a.ls) < d.s) ++a.ls;
  if ((1 << a.ls) != d.s) {
    set_error("convt: stride %d is not a power of two", d.s);
    return FLM_ERR_SHAPE;
  } a.ppf = (d.sub || d.epilogue == 3) ? 1 : 0;
  a.tau = d.tau; a.cand = d.cand; a.cand_cnt = d.cand_cnt; a.cand_cap = d.cand_cap; a.gate = d.gate;
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
  if (d.epilogue == 3) {
    if (!convt_candidates_supported(d.g) || !d.tau || !d.cand || !d.cand_cnt || d.cand_cap <= 0 ||
        (long long)d.ho * d.wo >= (1 << 17)) {
      set_error("convt: candidate epilogue needs the 68-class kernels, its buffers and a map below 2^17 pixels");
      return FLM_ERR_UNSUPPORTED;
    }
    if (!a.share) {
      set_error("convt: the candidate epilogue is built for strides that are multiples of 4");
      return FLM_ERR_UNSUPPORTED;
    }
    return d.g.bf16 ? launch_t<5, 9, true, 2, 1, true>(st, a) : launch_t<5, 17, false, 1, 1, true>(st, a);
  }
  if (d.epilogue == 4) {
    if (!convt_candidates_supported(d.g) || d.sub < 1 || d.sub > kMaxSamplePhases || !d.y) {
      set_error("convt: the sampling epilogue needs the 68-class kernels, sub > 0 and an output buffer");
      return FLM_ERR_UNSUPPORTED;
    }
    return d.g.bf16 ? launch_t<5, 9, true, 2, 2>(st, a) : launch_t<5, 17, false, 1, 2>(st, a);
  }
  if (d.g.bf16) {
    // two pixel tiles per wave: at 16x the matrix rate the phase weights (45 KiB per 64 positions) are the
    // stream to economise
    if (d.g.C == 68 && d.g.G == 9) return a.share ? launch_t<5, 9, true, 2, 0, true>(st, a) : launch_t<5, 9, true, 2, 0, false>(st, a);
    switch (d.g.MT) {
      case 1: return launch_t<1, 2, true>(st, a);
      case 2: return launch_t<2, 4, true>(st, a);
      case 3: return launch_t<3, 6, true>(st, a);
      case 4: return launch_t<4, 8, true>(st, a);
      case 5: return launch_t<5, 10, true>(st, a);
      case 6: return launch_t<6, 12, true>(st, a);
    }
  } else {
    if (d.g.C == 68 && d.g.G == 17) return a.share ? launch_t<5, 17, false, 1, 0, true>(st, a) : launch_t<5, 17, false, 1, 0, false>(st, a);
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
