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
#include <utility>

#include "flm_convt_dev.h"

namespace flm {


// Developer ablations (tools/ab_variants.py builds variants with -DFLM_ABLATE=<mask>; results are wrong, only timings
// mean anything): 1 no candidate test / stores (part 3), 2 no normalisation (part 2), 4 no max / exp (part 1),
// 8 no MFMAs, 16 no weight ring (loads, LDS stores, barriers); cand8 kernel: 32 prologue only, 64 no end-of-phase
// wait + barrier, 128 no hit loop, 256 no softmax / threshold ops, 512 no MFMAs, 1024 no LDS-DMA, 2048 the branchy
// per-piece form of the requests (results unchanged).  0 in every shipped build.
#ifndef FLM_ABLATE
#define FLM_ABLATE 0
#endif
constexpr int GCH_F32 = 6, GCH_BF16 = 3;  // k groups per LDS chunk (bf16: smaller chunks, fewer staging registers)


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
                                               (CAND ? sizeof(unsigned long long) * 4 * kCandWaveCap + sizeof(float) * 4 * 16 * MT : 0)) +
                   wave * kMaxSamplePhases * 16 * MT;  // [phase of the tile's list][result row]
  if (SAMPLE)
    for (int c = lane; c < a.sub * 16 * MT; c += 64) wmax[c] = 0u;

  const int s = a.s;
  // Phase (a0, b0) of iteration IT of this workgroup: packed phase blockIdx.y * nb + IT (nb = s for one phase row per
  // workgroup; a multiple of s when the workgroup walks several rows with the same X fragments, see convt_rows_per_wg);
  // in the sampling launch the IT-th entry of the tile's list -- an odd stride walks all s*s phases before repeating.
  const int tile_pf = a.sub ? (int)(blockIdx.x % (a.ppf / (64 * NT))) : 0;
  // Shared tile-4 layout (template SHARE, packed weights per convt_share_layout): the four extra classes 64..67 of FOUR consecutive phases b0 = 4gb .. 4gb+3 are
  // rows 4q + j of the leader phase's tile 4 (X is the same for every phase, only the filter differs), so the short
  // phases 4gb+1..3 multiply and read 4 tiles instead of 5: 15 % fewer MFMAs.  Their fifth tile in the stream still
  // holds their own classes 64..67 (rows 4q): the sampling launch multiplies all five tiles of whatever phase it draws,
  // so its probabilities are the main launch's bit for bit.
#define FLM_PHASE(IT)                                                                                    \
  (a.sub ? (((tile_pf * a.sub + (IT)) * 23 + 5) & (s * s - 1)) : ((int)blockIdx.y * a.nb + (IT)))


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
  // candidate mode: thresholds of this WAVE's face for the 4*MT classes of a lane (16m + 4q + e).  Faces are padded at
  // wave granularity there (16*NT positions: 1,104 of them per 256 x 256 face in fp32 where whole workgroups took 1,152,
  // 5.5 % of the launch's positions), so the four waves of a workgroup may belong to two faces: thresholds and flush per wave
  const int wg_img = CAND ? __builtin_amdgcn_readfirstlane(((blockIdx.x * 4 + wave) * 16 * NT) / (a.ppf > 0 ? a.ppf : 1)) : 0;
  if (CAND) {
    tau_s += wave * 16 * MT;
    // thresholds clamped to FLT_MIN so that p >= tau implies p > 0 (zero weights cannot move a centroid; a class left
    // with fewer than n keys is caught by cand_merge_kernel)
    for (int c = lane; c < 16 * MT; c += 64) {  // indexed by result row: class of row p
      const int cls = (C68 && c >= 64) ? (((c & 3) == 0) ? 64 + ((c - 64) >> 2) : a.C) : c;
      tau_s[c] = (cls < a.C && wg_img < a.n) ? fmaxf(a.tau[(size_t)wg_img * a.C + cls], 1.17549435e-38f) : 3.402823466e38f;
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
      const bool more = seq + 1 < total && !(FLM_ABLATE & 16);                                      \
      if (more) FLM_ISSUE(seq + 1)                                                                  \
      const float4* wl = lds + (seq & 1) * CHUNK_F4;                                                \
      if (b0 > 0) {                                                                                 \
        if (ch == 0 && !(FLM_ABLATE & 4)) FLM_EPI_PART1(PV)                                         \
        if (ch == (NCH > 1 ? 1 : 0) && !(FLM_ABLATE & 2)) FLM_EPI_PART2(PV, b0 - 1)                 \
        if (ch == NCH - 1 && !(FLM_ABLATE & 1)) FLM_EPI_PART3(PV, b0 - 1)                           \
      }                                                                                             \
      _Pragma("unroll") for (int gl = 0; gl < GCH; ++gl) {                                          \
        const int g = ch * GCH + gl; /* compile-time */                                             \
        if (g < G && !(FLM_ABLATE & 8)) {                                                           \
          float4 af[MT];                                                                            \
          _Pragma("unroll") for (int m = 0; m < (MTP); ++m) af[m] = wl[(gl * MT + m) * 64 + lane];  \
          if constexpr (BF) {                                                                       \
            _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) {                                     \
              const bf16x8 xb = __builtin_bit_cast(bf16x8, xf[nt][g]);                              \
              _Pragma("unroll") for (int m = 0; m < (MTP); ++m)                                     \
                ACC[nt][m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[m]), xb, ACC[nt][m], 0, 0, 0); \
            }                                                                                       \
          } else {                                                                                  \
            /* fp32: K = 4 * Cp products per output are NOT one fmaf chain: the SUBG k groups of an LDS chunk (96   */ \
            /* products) sum in a side accumulator from zero and the sub-chain sums are added in order -- 96 + 3    */ \
            /* roundings instead of 272 (up3's own rounding was the largest single share of the logits' error,      */ \
            /* DESIGN 2; sub-chains of 64 cost three registers more than the kernel has)                             */ \
            /* (the first sub-chain sums in the phase's accumulators themselves, cleared at the phase start)        */ \
            const bool sub_first = g % SUBG == 0;                                                   \
            const bool sub_last = (g == G - 1) || ((g + 1) % SUBG == 0);                            \
            if (g < SUBG) {                                                                         \
              _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) {                                   \
                _Pragma("unroll") for (int m = 0; m < (MTP); ++m)                                   \
                  ACC[nt][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m].x, xf[nt][g].x, ACC[nt][m], 0, 0, 0); \
                _Pragma("unroll") for (int m = 0; m < (MTP); ++m)                                   \
                  ACC[nt][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m].y, xf[nt][g].y, ACC[nt][m], 0, 0, 0); \
                _Pragma("unroll") for (int m = 0; m < (MTP); ++m)                                   \
                  ACC[nt][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m].z, xf[nt][g].z, ACC[nt][m], 0, 0, 0); \
                _Pragma("unroll") for (int m = 0; m < (MTP); ++m)                                   \
                  ACC[nt][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m].w, xf[nt][g].w, ACC[nt][m], 0, 0, 0); \
              }                                                                                     \
            } else {                                                                                \
            _Pragma("unroll") for (int nt = 0; nt < NT; ++nt) {                                     \
              _Pragma("unroll") for (int m = 0; m < (MTP); ++m)                                     \
                tmpacc[nt][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m].x, xf[nt][g].x, sub_first ? (f32x4){0.f, 0.f, 0.f, 0.f} : tmpacc[nt][m], 0, 0, 0); \
              _Pragma("unroll") for (int m = 0; m < (MTP); ++m)                                     \
                tmpacc[nt][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m].y, xf[nt][g].y, tmpacc[nt][m], 0, 0, 0); \
              _Pragma("unroll") for (int m = 0; m < (MTP); ++m)                                     \
                tmpacc[nt][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m].z, xf[nt][g].z, tmpacc[nt][m], 0, 0, 0); \
              _Pragma("unroll") for (int m = 0; m < (MTP); ++m)                                     \
                tmpacc[nt][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m].w, xf[nt][g].w, tmpacc[nt][m], 0, 0, 0); \
              if (sub_last) {                                                                       \
                _Pragma("unroll") for (int m = 0; m < (MTP); ++m) ACC[nt][m] += tmpacc[nt][m];      \
              }                                                                                     \
            }                                                                                       \
            }                                                                                       \
          }                                                                                         \
        }                                                                                           \
      }                                                                                             \
      if (more) FLM_STASH((seq + 1) & 1)                                                            \
      if (!(FLM_ABLATE & 16)) __syncthreads();                                                      \
      ++seq;                                                                                        \
    }                                                                                               \
  }
  constexpr int SUBG = GCH_F32;          // fp32: k groups per sub-chain of the two-level sum (= one LDS chunk: 96 products)
  f32x4 tmpacc[BF ? 1 : NT][MT];         // fp32: the running sub-chain
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


// Phase rows per workgroup.  A workgroup's X fragments (one burst of global loads that every CU issues at once, plus the
// index arithmetic behind it) serve rows * s phases: more rows amortise that prologue, fewer keep the last round of
// workgroups full.  Picks the cheaper by rounds x (prologue + phases), the prologue priced at kProlog phases (bf16 batch
// 512, 8-wave kernel: 2.75 ms with one row per workgroup, 2.40 / 2.18 / 2.15 with 2 / 4 / 8).  `forced` > 0 overrides.
static int convt_rows_per_wg(int xblocks, int s, int wg_slots, int forced = 0) {
  constexpr double kProlog = 2.5;
  int best = 1;
  double best_cost = 1e30;
  for (int rows = 1; rows <= s; rows *= 2) {
    if (forced == rows) return rows;
    const long long wgs = (long long)xblocks * (s / rows);
    const double cost = (double)((wgs + wg_slots - 1) / wg_slots) * (kProlog + (double)rows * s);
    if (cost < best_cost) {
      best = rows;
      best_cost = cost;
    }
  }
  return best;
}

// =====================================================================================================================
// up3 in landmark mode (epilogue 3) for the 68-class model: a kernel of its own (`cand8`), bf16 and fp32.
//
// What the generic kernel above spends its time on in this mode (round-1 ablations with -DFLM_ABLATE, bf16 batch 512: 2.63 ms): the
// weight ring alone -- global -> registers -> ds_write -> barrier, three chunks per phase, 13.6 GB per launch out of L2 --
// takes 1.40 ms with everything else compiled out; the ten threshold reads per phase each expose an LDS round trip; the
// candidate code, inlined at 160 sites, pushes the loop past the instruction cache (68 KB); and every workgroup reloads
// its X fragments for just one phase row.  Here:
//   * 8 waves per workgroup, one workgroup per CU: a ring step's weights are fetched once per 256 (fp32: 128) positions
//     -- half the bytes -- straight into LDS by LDS-DMA (no staging registers, no ds_write pass), in 45 KiB steps: a
//     whole phase in bf16 (ONE barrier per phase instead of three), half a phase in fp32 (k groups 0..8 | 9..16); the
//     fifth class tile of the three short phases of a group is not fetched;
//   * a workgroup walks several phase rows (up to all 64 phases) with the same X fragments (convt_rows_per_wg);
//   * per-face padding at wave granularity (bf16: 1089 -> 1120 positions per face, fp32: 1104, instead of 1152); a wave's
//     thresholds live in 20 registers (the staging registers' room), its face may differ from its neighbours';
//   * the softmax / threshold test of phase b-1 is cut into small ops dealt over the MFMA slots of phase b by cost
//     (compile-time schedule, order pinned): both waves of a SIMD run the same even mix of matrix and vector work;
//   * hits (about one value in 500) only set a bit per (pixel tile, class tile) group; one lean loop per phase then
//     re-tests the flagged groups and appends the keys -- the code that was inlined 160 times exists 20 times.
// Arithmetic per value is the generic kernel's, operation by operation (same MFMA order over k, same max / exp / sum
// order / reciprocal / product), so keys carry the very bits its materialising and sampling launches produce: thresholds
// taken from the sampling launch stay valid, and the landmark result is bit-identical (tests/test_gpu_candidates.py).
// bf16 batch 512: 2.63 -> 2.12 ms.
// =====================================================================================================================
namespace cand8 {
constexpr int MT = 5;
constexpr int PIECE = 1024;             // one (g, m) fragment tile: 64 lanes x 16 bytes
constexpr int KEY_CAP = 512;            // keys a wave holds in LDS between flushes
// FLM_CAND8_LANE (default 1): keys are appended INSIDE the slot stream, each lane to a list of its own (LANE_CAP entries,
// [entry][lane] in the wave's LDS region) -- a compare, a saved exec mask and a skipped body when no lane hits; no ballot,
// no mbcnt, no re-test, no branch ladder after the phase.  0: the hit loop (flag bits in the slots, one loop per phase).
#ifndef FLM_CAND8_LANE
#define FLM_CAND8_LANE 1
#endif
constexpr bool LANE = FLM_CAND8_LANE != 0;
constexpr int LANE_CAP = FLM_CAND8_LANE > 1 ? FLM_CAND8_LANE : 16;  // keys per lane between flushes; a flush is due when some lane holds LANE_CAP / 2 (macro values > 1: that capacity, for timing the flush)
constexpr int WAVE_LIST = LANE ? 64 * LANE_CAP : KEY_CAP;
// WAVES waves per workgroup share a ring of two steps of STEP_G k groups: (8, 9) = one workgroup per CU, 45 KiB steps
// (bf16: a whole phase); (4, 3) = two workgroups per CU that drift apart, 15 KiB steps, twice the weight traffic
constexpr size_t lds_bytes(int waves, int step_g) { return 2 * (size_t)step_g * MT * PIECE + (size_t)waves * WAVE_LIST * 8; }

template <bool BF>
struct Cfg {
  static constexpr int G = BF ? 9 : 17;      // k groups: 32 deep (bf16, one 16x16x32 MFMA) or 16 deep (fp32, four 16x16x4)
  static constexpr int NT = BF ? 2 : 1;      // pixel tiles of 16 per wave
  static constexpr int KM = BF ? 1 : 4;      // MFMAs per (k group, class tile, pixel tile)
  static constexpr int PHASE_BYTES = G * MT * PIECE;
  static constexpr int NSLOT = G * 4 * NT * KM;            // MFMAs of the four common class tiles per phase and wave
  // ---- the epilogue of one phase as a list of small ops (pixel tiles interleaved: op k of a stage works on nt = k % NT)
  //   A  9 NT  running class maximum, two values per v_max3     B  NT  cross-lane-group maximum, -max*log2(e)
  //   C 17 NT (bf16: 34 NT, the fma and the v_exp_f32 of a value two ops apart)  e = exp(x - max)
  //   D 17 NT  sum += e (the generic kernel's order)
  //   E    NT  cross-lane-group sum, reciprocal, pixel validity  F 17 NT  p = e * (1/sum)
  //   G  5 NT  p >= tau for the 4 (1) values of a class tile -> one bit per group
  static constexpr int CS = BF ? 2 : 1;  // ops per value in stage C
  static constexpr int B0 = 9 * NT, C0 = B0 + NT, D0 = C0 + 17 * NT * CS, E0 = D0 + 17 * NT, F0 = E0 + NT, G0 = F0 + 17 * NT,
                       NOPS = G0 + 5 * NT;
  static constexpr int op_cost(int k) {  // issue cycles / 4, roughly (bf16: v_exp_f32 8; fp32: the accurate expf, a true division)
    return k < B0 ? 1 : k < C0 ? 7 : k < D0 ? (BF ? (((k - C0) % (2 * NT)) < NT ? 1 : 2) : 20) : k < E0 ? 1 : k < F0 ? (BF ? 8 : 16)
                                                                                                  : k < G0 ? 1 : 5;
  }
};
template <bool BF>
struct EpiSched {
  int first[Cfg<BF>::NSLOT + 1];  // ops [first[s], first[s+1]) run beside MFMA slot s
  constexpr EpiSched() : first() {
    using C = Cfg<BF>;
    int total = 0;
    for (int k = 0; k < C::NOPS; ++k) total += C::op_cost(k);
    int k = 0, cum = 0;
    for (int sl = 0; sl < C::NSLOT; ++sl) {
      first[sl] = k;
      // slot sl takes ops while the cumulated cost stays within its share
      while (k < C::NOPS && (cum + C::op_cost(k)) * (long long)C::NSLOT <= (long long)total * (sl + 1)) cum += C::op_cost(k++);
    }
    first[C::NSLOT] = C::NOPS;
  }
};
template <bool BF>
struct Sched {
  static constexpr EpiSched<BF> k{};
};

}  // namespace cand8

template <bool BF, int WAVES, int STEP_G>
__global__ __launch_bounds__(WAVES * 64, 2) void up3_cand8_kernel(ConvTArgs a) {  // 2 waves per SIMD: 256 registers each
  using namespace cand8;
  using C = Cfg<BF>;
  constexpr int G = C::G, NT = C::NT, KM = C::KM, NOPS = C::NOPS;
  constexpr int NSTEP = (G + STEP_G - 1) / STEP_G;          // ring steps per phase
  constexpr int SLOT_BYTES = STEP_G * MT * PIECE;           // a ring step
  constexpr int NDMA = (STEP_G * MT + WAVES - 1) / WAVES;   // pieces a wave requests per step
  if (a.gate && *a.gate == 0) return;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  typedef __attribute__((address_space(3))) char lds_char;
  const unsigned ring_lds = (unsigned)(size_t)((lds_char*)smem_raw);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  unsigned long long* cwave = reinterpret_cast<unsigned long long*>(smem_raw + 2 * SLOT_BYTES) + wave * WAVE_LIST;
  unsigned wcnt = 0;
  unsigned lcnt = 0;       // LANE: keys in this lane's list (past LANE_CAP: keys were dropped)
  unsigned pixq_l[NT];     // LANE: pixel | (4q << 17) of the phase whose products are being tested
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) pixq_l[nt] = 0u;

  const int s = a.s;
  const int row0 = blockIdx.y * a.rpw;  // this workgroup walks phase rows row0 .. row0 + rpw - 1: phases t = 0 .. rpw*s - 1
  const int nph = a.rpw * s;
  const int ls = a.ls;
  // ---- this wave's 16 NT positions: one face (ppf is a multiple of 16 NT) --------------------------------------------
  const int wi1 = a.wi + 1, hi1 = a.hi + 1;
  const int p0 = (blockIdx.x * WAVES + wave) * 16 * NT;
  const int wimg = p0 / a.ppf;
  const bool wlive = wimg < a.n;
  bool pvalid[NT];
  int oy0[NT], ox0[NT];  // output pixel of phase (a0, b0): (oy0 + a0, ox0 + b0)
  int i0[NT], j0[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int pl = p0 % a.ppf + nt * 16 + r;
    pvalid[nt] = wlive && pl < wi1 * hi1;
    const int pp = pvalid[nt] ? pl : 0;
    j0[nt] = pp % wi1;
    i0[nt] = pp / wi1;
    oy0[nt] = s * i0[nt];
    ox0[nt] = s * j0[nt];
  }
  // thresholds of the wave's face, clamped to FLT_MIN so that p >= tau implies p > 0 (a zero weight cannot move a
  // centroid; a class left with fewer than n keys is caught by cand_merge_kernel), in registers
  float4 tq[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    float t[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int cls = m < 4 ? 16 * m + 4 * q + e : (e == 0 ? 64 + q : a.C);
      t[e] = (cls < a.C && wlive) ? fmaxf(a.tau[(size_t)wimg * a.C + cls], 1.17549435e-38f) : 3.402823466e38f;
    }
    tq[m] = make_float4(t[0], t[1], t[2], t[3]);
  }

  // ---- X fragments, as the generic kernel: bf16: xf[nt][g] = bf16(x[tap(k8)][c(k8)..+7]), k8 = 32g + 8q;
  //      fp32: xf[nt][g] = x[tap(k4)][c(k4)..+3], k4 = 16g + 4q ----------------------------------------------------------
  f32x4 xf[NT][G];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int g = 0; g < G; ++g) {
      constexpr int EPL = BF ? 8 : 4;
      const int k0 = 4 * EPL * g + EPL * q;
      const int tap = k0 / a.Cp, c = k0 % a.Cp;
      const int ii = i0[nt] - (tap >> 1), jj = j0[nt] - (tap & 1);
      const bool ok = wlive && tap < 4 && (unsigned)ii < (unsigned)a.hi && (unsigned)jj < (unsigned)a.wi &&
                      (p0 % a.ppf + nt * 16 + r) < wi1 * hi1;
      const size_t off = ok ? (((size_t)wimg * a.hi + ii) * a.wi + jj) * a.Cp + c : 0;
      const float4 v0 = *reinterpret_cast<const float4*>(a.x + off);
      if constexpr (BF) {
        const float4 v1 = *reinterpret_cast<const float4*>(a.x + off + 4);
        bf16x8 t;
        t[0] = (__bf16)(ok ? v0.x : 0.f); t[1] = (__bf16)(ok ? v0.y : 0.f);
        t[2] = (__bf16)(ok ? v0.z : 0.f); t[3] = (__bf16)(ok ? v0.w : 0.f);
        t[4] = (__bf16)(ok ? v1.x : 0.f); t[5] = (__bf16)(ok ? v1.y : 0.f);
        t[6] = (__bf16)(ok ? v1.z : 0.f); t[7] = (__bf16)(ok ? v1.w : 0.f);
        xf[nt][g] = __builtin_bit_cast(f32x4, t);
      } else {
        xf[nt][g] = (f32x4){ok ? v0.x : 0.f, ok ? v0.y : 0.f, ok ? v0.z : 0.f, ok ? v0.w : 0.f};
      }
    }

  // ---- weight ring: two 45 KiB steps filled by LDS-DMA; wave w moves pieces w, w+8, ... of the next step ---------------
  const unsigned long long wbase = reinterpret_cast<unsigned long long>(a.wf);
  const dma_srd wsrd = (dma_srd){(int)(unsigned)wbase, (int)(unsigned)((wbase >> 32) & 0xffffu), 0x7fffffff, 0x00020000};
  const unsigned voff = (unsigned)lane * 16u;
  // Ring step u = t * NSTEP + st: k groups [st * 9, min(G, st * 9 + 9)) of the workgroup's phase t (packed phase
  // row0*s + t: consecutive rows are consecutive in the packed array).  Piece i -> (g, m): a leader phase (t % 4 == 0)
  // moves five class tiles per k group, a short one the four common ones.
  auto dma_issue = [&](int u, int i) __attribute__((always_inline)) {
    const int t = u / NSTEP, st = u - t * NSTEP;
    const bool leader = (t & 3) == 0;
    const int ng = (st + 1) * STEP_G <= G ? STEP_G : G - st * STEP_G;
    const int np = ng * (leader ? MT : 4);
    if (i < np) {
      const int pc = leader ? i : (i >> 2) * MT + (i & 3);  // piece of the step: (g - g0) * 5 + m
      dma_piece(wsrd, ring_lds + (unsigned)((u & 1) * SLOT_BYTES + pc * PIECE), voff,
                (row0 * s + t) * C::PHASE_BYTES + (st * STEP_G * MT + pc) * PIECE);
    }
  };
#pragma unroll
  for (int k = 0; k < NDMA; ++k) dma_issue(0, wave + WAVES * k);
  __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): this wave's pieces have landed
  __syncthreads();
  // Whole-phase ring steps (NSTEP == 1: the bf16 8-wave shape): the byte offsets of this wave's NDMA pieces, for a leader
  // phase (5 class tiles per k group) and for a short one (4), are computed ONCE; a piece the wave does not have repeats
  // its first one (the same bytes to the same place).  The per-phase request is then two scalar selects and two adds per
  // piece and NO branch: the general form above decides `i < np` per piece -- six scalar branches and ~90 scalar
  // instructions per phase inside the slot-scheduled stream (round-3 ablation: 0.4 ms of the launch went with the requests).
  int pieceL[NDMA], pieceS[NDMA];
#pragma unroll
  for (int k = 0; k < NDMA; ++k) {
    const int i = wave + WAVES * k, i0 = wave;
    pieceL[k] = (i < G * MT ? i : i0) * PIECE;
    const int is_ = i < G * 4 ? i : i0;
    pieceS[k] = ((is_ >> 2) * MT + (is_ & 3)) * PIECE;
  }
  auto dma_issue_fast = [&](int u, auto kc) __attribute__((always_inline)) {   // requests piece k of ring step u (phase u)
    constexpr int k = decltype(kc)::value;
    const int t = u < nph ? u : nph - 1;                 // (past the last phase: its own weights again, into the idle slot)
    const int off = (t & 3) == 0 ? pieceL[k] : pieceS[k];
    dma_piece(wsrd, ring_lds + (unsigned)((u & 1) * SLOT_BYTES + off), voff, (row0 * s + t) * C::PHASE_BYTES + off);
  };

  f32x4 accA[NT][MT], accB[NT][MT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int m = 0; m < MT; ++m) accA[nt][m] = accB[nt][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float x4r[NT][3];  // class 64+q of the three short phases of the current group (rows 4q+1..3 of the leader's fifth tile)
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) x4r[nt][0] = x4r[nt][1] = x4r[nt][2] = 0.f;

  // keys of this wave: LDS region -> its face's global list (one atomic reserves the range)
  auto cand_flush = [&]() __attribute__((always_inline)) {
    const unsigned found = __builtin_amdgcn_readfirstlane(wcnt);
    if (wlive && found) {
      unsigned base = 0;
      if (lane == 0) {
        base = atomicAdd(&a.cand_cnt[wimg], found);
        if (base + found > (unsigned)a.cand_cap) atomicOr(&a.cand_cnt[a.n], 1u);
      }
      base = __builtin_amdgcn_readfirstlane(base);
      for (unsigned i = lane; i < found; i += 64)
        if (base + i < (unsigned)a.cand_cap) a.cand[(size_t)wimg * a.cand_cap + base + i] = cwave[i];
    }
    wcnt = 0;
  };

  // LANE: the lanes' lists -> the face's global list: an exclusive scan of the counts over the wave (one atomic reserves
  // the range), every lane copies its own entries.  Due when some lane is half full; a lane that ran out of room in
  // between has dropped keys: the overflow flag sends the batch through the materialising launch.
  auto lane_flush = [&]() __attribute__((always_inline)) {
    const bool over = lcnt > (unsigned)LANE_CAP;
    if (__any(over) && wlive && lane == 0) atomicOr(&a.cand_cnt[a.n], 1u);
    const unsigned mine = over ? (unsigned)LANE_CAP : lcnt;
    unsigned incl = mine;  // inclusive scan over the 64 lanes
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const unsigned o = __shfl_up(incl, d);
      if (lane >= d) incl += o;
    }
    const unsigned total = __builtin_amdgcn_readlane(incl, 63);
    if (wlive && total) {
      unsigned base = 0;
      if (lane == 0) {
        base = atomicAdd(&a.cand_cnt[wimg], total);
        if (base + total > (unsigned)a.cand_cap) atomicOr(&a.cand_cnt[a.n], 1u);
      }
      base = __builtin_amdgcn_readfirstlane(base) + incl - mine;
      for (unsigned i = 0; i < mine; ++i)
        if (base + i < (unsigned)a.cand_cap) a.cand[(size_t)wimg * a.cand_cap + base + i] = cwave[i * 64 + lane];
    }
    lcnt = 0;
  };

  // ---- epilogue state of the phase being finished --------------------------------------------------------------------
  float mx[NT], nmxl[NT], sum[NT], rs[NT];
  unsigned hitmask = 0;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) mx[nt] = nmxl[nt] = sum[nt] = rs[nt] = 0.f;

  // op K of the list above on the finished phase's values PV; bprev = its index t (< 0: no phase yet, nothing may hit)
  // (pin: an empty volatile asm on a value.  The ops are pure arithmetic, which LLVM places wherever the data flow
  //  allows -- it regrouped the exps and the products into blocks between the MFMAs -- and sched_barrier only fences the
  //  machine scheduler; a volatile asm keeps its place among the other side-effecting statements, so an op whose RESULT is
  //  pinned is issued no later than the slot it was written in, and nothing makes hipcc issue it earlier.  Pinning the
  //  inputs as well cost a hazard s_nop around every pin: 585 in the bf16 kernel against 265 this way.)
#define FLM_PIN(x) asm volatile("" : "+v"(x))
  auto epi_op = [&](auto kc, f32x4(&PV)[NT][MT], int bprev) __attribute__((always_inline)) {
    constexpr int K = decltype(kc)::value;
    if constexpr (K < C::B0) {
      constexpr int nt = K % NT, j = K / NT;
      if constexpr (j == 0) mx[nt] = max_raw(PV[nt][0][0], PV[nt][0][1]);
      else if constexpr (j < 8) mx[nt] = max3_raw(mx[nt], PV[nt][j >> 1][2 * (j & 1)], PV[nt][j >> 1][2 * (j & 1) + 1]);
      else mx[nt] = max_raw(mx[nt], PV[nt][4][0]);
      FLM_PIN(mx[nt]);
    } else if constexpr (K < C::C0) {
      constexpr int nt = (K - C::B0) % NT;
      mx[nt] = reduce_q_max(mx[nt]);
      nmxl[nt] = -mx[nt] * 1.44269504088896340736f;
      FLM_PIN(nmxl[nt]);
    } else if constexpr (K < C::D0) {
      if constexpr (BF) {
        // the exponent's argument and the v_exp_f32 of a value are separate ops with the other pixel tile's in between:
        // back to back, the dependent pair stalls the in-order wave (same two instructions as softmax_exp<true>)
        constexpr int c = K - C::C0, i = c / (2 * NT), rr = c % (2 * NT), nt = rr % NT, m = i < 16 ? i >> 2 : 4, e = i < 16 ? i & 3 : 0;
        float v;
        if constexpr (rr < NT) v = __builtin_fmaf(PV[nt][m][e], 1.44269504088896340736f, nmxl[nt]);
        else v = __builtin_amdgcn_exp2f(PV[nt][m][e]);
        FLM_PIN(v);
        PV[nt][m][e] = v;
      } else {
        constexpr int nt = (K - C::C0) % NT, i = (K - C::C0) / NT, m = i < 16 ? i >> 2 : 4, e = i < 16 ? i & 3 : 0;
        float v = softmax_exp<BF>(PV[nt][m][e], mx[nt], nmxl[nt]);
        FLM_PIN(v);
        PV[nt][m][e] = v;
      }
    } else if constexpr (K < C::E0) {
      constexpr int nt = (K - C::D0) % NT, i = (K - C::D0) / NT, m = i < 16 ? i >> 2 : 4, e = i < 16 ? i & 3 : 0;
      if constexpr (i == 0) sum[nt] = 0.f + PV[nt][m][e];
      else sum[nt] += PV[nt][m][e];
      FLM_PIN(sum[nt]);
    } else if constexpr (K < C::F0) {
      constexpr int nt = (K - C::E0) % NT;
      const float sq = reduce_q_sum(sum[nt]);
      const bool ok = pvalid[nt] && bprev >= 0 && oy0[nt] + row0 + (bprev >> ls) < a.ho && ox0[nt] + (bprev & (s - 1)) < a.wo;
      const float inv = BF ? __builtin_amdgcn_rcpf(sq) : 1.0f / sq;
      rs[nt] = ok ? inv : 0.f;
      FLM_PIN(rs[nt]);
      if constexpr (LANE)
        pixq_l[nt] = (unsigned)((oy0[nt] + row0 + (bprev >> ls)) * a.wo + ox0[nt] + (bprev & (s - 1))) + ((unsigned)(4 * q) << 17);
    } else if constexpr (K < C::G0) {
      constexpr int nt = (K - C::F0) % NT, i = (K - C::F0) / NT, m = i < 16 ? i >> 2 : 4, e = i < 16 ? i & 3 : 0;
      float v = PV[nt][m][e] * rs[nt];
      FLM_PIN(v);
      PV[nt][m][e] = v;
    } else {
      constexpr int nt = (K - C::G0) % NT, m = (K - C::G0) / NT;
      unsigned long long mk = __ballot(PV[nt][m][0] >= tq[m].x);
      if constexpr (m < 4) {
        mk |= __ballot(PV[nt][m][1] >= tq[m].y);
        mk |= __ballot(PV[nt][m][2] >= tq[m].z);
        mk |= __ballot(PV[nt][m][3] >= tq[m].w);
      }
      if constexpr (LANE) {
        if (mk) {  // (wave-uniform, about three groups in ten) the hitting lanes append to their own lists
          const float tv[4] = {tq[m].x, tq[m].y, tq[m].z, tq[m].w};
#pragma unroll
          for (int e = 0; e < (m < 4 ? 4 : 1); ++e) {
            const float pvv = PV[nt][m][e];
            if (pvv >= tv[e]) {  // (p >= tau > 0: its order bits are its bits with the sign set)
              const unsigned lo = m < 4 ? pixq_l[nt] + ((unsigned)(16 * m + e) << 17) : pixq_l[nt] + ((unsigned)(64 - 3 * q) << 17);
              if (lcnt < (unsigned)LANE_CAP)
                cwave[lcnt * 64 + lane] = ((unsigned long long)(__float_as_uint(pvv) | 0x80000000u) << 32) | lo;
              ++lcnt;
            }
          }
        }
        return;
      }
      hitmask |= mk ? 1u << (nt * MT + m) : 0u;  // (scalar: the products it tests are pinned in their slots)
    }
  };
  // The flagged groups of the finished phase: re-test, append keys (value order bits << 32 | class << 17 | pixel).  One
  // copy of the append code per group and phase body (compile-time registers) instead of the generic kernel's four per
  // group, each reached through one wave-uniform bit test and only when some group was flagged.  Kept lean -- per hit
  // value: the ballot, two mbcnt, the key halves (p > 0, so its order bits are its bits with the sign set; class and
  // pixel add up from per-lane terms computed once per call), one ds_write_b64.  Room is made once per call (a flush
  // when the wave's region is half full); a call that still runs out (more than 256 hits of one wave in one phase: flat
  // maps) drops the excess keys and raises the overflow flag, which sends the batch through the materialising launch.
  auto hit_loop = [&](f32x4(&PV)[NT][MT], int bprev) __attribute__((always_inline)) {
    if constexpr (LANE) {
      if (__any(lcnt >= (unsigned)(LANE_CAP / 2))) lane_flush();
      return;
    }
    const unsigned hm = __builtin_amdgcn_readfirstlane(hitmask);
    hitmask = 0;
    if (hm == 0) return;
    if (wcnt > (unsigned)(KEY_CAP - 256)) cand_flush();
    unsigned pixq[NT];  // pixel | (4q << 17): plus (16m + e) << 17 it is class << 17 | pixel for the four common tiles
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
      pixq[nt] = (unsigned)((oy0[nt] + row0 + (bprev >> ls)) * a.wo + ox0[nt] + (bprev & (s - 1))) + ((unsigned)(4 * q) << 17);
    static_for<NT * MT>([&](auto gc) __attribute__((always_inline)) {
      constexpr int grp = decltype(gc)::value, nt = grp / MT, m = grp % MT;
      if ((hm >> grp) & 1u) {
        const float tv[4] = {tq[m].x, tq[m].y, tq[m].z, tq[m].w};
#pragma unroll
        for (int e = 0; e < (m < 4 ? 4 : 1); ++e) {
          const float pvv = PV[nt][m][e];
          const bool hit = pvv >= tv[e];
          const unsigned long long mk = __ballot(hit);
          if (mk) {
            const unsigned slot = wcnt + __builtin_amdgcn_mbcnt_hi((unsigned)(mk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk, 0u));
            // class 16m + 4q + e, or 64 + q on the fifth tile: (64 + q) - 4q = 64 - 3q
            const unsigned lo = m < 4 ? pixq[nt] + ((unsigned)(16 * m + e) << 17) : pixq[nt] + ((unsigned)(64 - 3 * q) << 17);
            if (hit && slot < (unsigned)KEY_CAP)
              cwave[slot] = ((unsigned long long)(__float_as_uint(pvv) | 0x80000000u) << 32) | lo;
            wcnt += __builtin_popcountll(mk);
          }
        }
      }
    });
    if (wcnt > (unsigned)KEY_CAP) {  // keys were dropped: exactness now rests on the materialising launch
      if (lane == 0) atomicOr(&a.cand_cnt[a.n], 1u);
      wcnt = KEY_CAP;
    }
  };

  // one MFMA group of a (k group, class tile, pixel tile): bf16 one 16x16x32, fp32 component KC of four 16x16x4; FIRST:
  // the accumulator starts from zero
  auto mfma_step = [&](auto kcc, auto firstc, f32x4& acc, const f32x4& af, const f32x4& xv) __attribute__((always_inline)) {
    constexpr int KC = decltype(kcc)::value;
    constexpr bool FIRST = decltype(firstc)::value;
    const f32x4 zero = (f32x4){0.f, 0.f, 0.f, 0.f};
    if constexpr (BF) {
      (void)KC;
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af), __builtin_bit_cast(bf16x8, xv), FIRST ? zero : acc, 0, 0, 0);
    } else {
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(af[KC], xv[KC], FIRST ? zero : acc, 0, 0, 0);
    }
  };

  // ---- one ring step ST of phase b: its MFMA slots on the common tiles with the finished phase's ops beside them, the
  //      fifth tile of a leader phase, after the phase's last step the hit loop; then the wait and the barrier ----------
  // Slot order inside a k group: bf16 (m, nt); fp32 (component, m) -- each accumulator sees k in ascending order, as in the
  // generic kernel.
  auto step_body = [&](auto stc, f32x4(&ACC)[NT][MT], f32x4(&PV)[NT][MT], int b) __attribute__((always_inline)) {
    constexpr int ST = decltype(stc)::value;
    constexpr int g0 = ST * STEP_G, g1 = (g0 + STEP_G < G) ? g0 + STEP_G : G;
    constexpr int SPG = 4 * NT * KM;  // slots per k group
    const bool leader = (b & 3) == 0;
    const int bprev = b - 1;
    const int u = b * NSTEP + ST;
    if constexpr (ST == 0) {
      // the finished phase was a short one: its class 64+q value waits in the group's fifth-tile sums
      if ((bprev & 3) != 0 && bprev >= 0) {
        const int j = (bprev & 3) - 1;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) PV[nt][MT - 1][0] = j == 0 ? x4r[nt][0] : (j == 1 ? x4r[nt][1] : x4r[nt][2]);
      }
    }
    const f32x4* wl = reinterpret_cast<const f32x4*>(smem_raw + (u & 1) * SLOT_BYTES) + lane;
    f32x4 af[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) af[m] = wl[m * 64];
    static_for<(g1 - g0) * SPG>([&](auto ic) __attribute__((always_inline)) {
      constexpr int IL = decltype(ic)::value;        // slot inside the step
      constexpr int I = g0 * SPG + IL;               // slot inside the phase
      constexpr int g = I / SPG, w = I % SPG;
      constexpr int m = BF ? w / NT : w % 4, nt = BF ? w % NT : 0, kc = BF ? 0 : w / 4;
      constexpr bool first = g == 0 && kc == 0;
      // (the MFMA is pure arithmetic too: pinned from above by its own operand -- the fragment for the first MFMA of a
      //  step, the accumulator afterwards, whose pin a few slots later also bounds how far the MFMA before it may sink;
      //  that pin reads a result finished long ago, so it costs no hazard wait)
      if constexpr (FLM_ABLATE & 512) {
        FLM_PIN(af[m]);
      } else {
        if constexpr (first || (IL < 4 * NT && kc == 0)) { if constexpr (nt == 0) FLM_PIN(af[m]); }
        if constexpr (!first) FLM_PIN(ACC[nt][m]);
        mfma_step(std::integral_constant<int, kc>{}, std::integral_constant<bool, first>{}, ACC[nt][m], af[m], xf[nt][g]);
      }
      // the fragment is reloaded after its last use in the k group
      if constexpr ((BF ? nt == NT - 1 : kc == KM - 1) && g + 1 < g1) af[m] = wl[((g + 1 - g0) * MT + m) * 64];
      if constexpr (!(FLM_ABLATE & 256))
      static_for<Sched<BF>::k.first[I + 1] - Sched<BF>::k.first[I]>([&](auto jc) __attribute__((always_inline)) {
        epi_op(std::integral_constant<int, Sched<BF>::k.first[I] + decltype(jc)::value>{}, PV, bprev);
      });
      // the next step's pieces: requested in the first slots -- its ring slot has been free since this step's barrier,
      // and the requests then have the whole step to land
      if constexpr (IL < NDMA) {
        if constexpr (NSTEP == 1 && !(FLM_ABLATE & 2048)) {
          if (!(FLM_ABLATE & 1024)) dma_issue_fast(u + 1, std::integral_constant<int, IL>{});
        } else {
          if (u + 1 < nph * NSTEP && !(FLM_ABLATE & 1024)) dma_issue(u + 1, wave + WAVES * IL);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    });
    if (leader) {
      f32x4 t4 = wl[4 * 64];
#pragma unroll
      for (int g = g0; g < g1; ++g) {
        const f32x4 cur = t4;
        if (g + 1 < g1) t4 = wl[((g + 1 - g0) * MT + 4) * 64];
#pragma unroll
        for (int kc = 0; kc < KM; ++kc)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) {
            if constexpr (BF) {
              ACC[nt][4] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, cur), __builtin_bit_cast(bf16x8, xf[nt][g]),
                                                                   g == 0 ? (f32x4){0.f, 0.f, 0.f, 0.f} : ACC[nt][4], 0, 0, 0);
            } else {
              ACC[nt][4] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[kc], xf[nt][g][kc], (g == 0 && kc == 0) ? (f32x4){0.f, 0.f, 0.f, 0.f} : ACC[nt][4], 0, 0, 0);
            }
          }
      }
    }
    if constexpr (ST == NSTEP - 1) {
      if (!(FLM_ABLATE & 128)) hit_loop(PV, bprev);
      if (leader) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          x4r[nt][0] = ACC[nt][4][1];
          x4r[nt][1] = ACC[nt][4][2];
          x4r[nt][2] = ACC[nt][4][3];
        }
      }
    }
    if (!(FLM_ABLATE & 64)) {
      __builtin_amdgcn_s_waitcnt(0x0f70);  // this wave's pieces of the next step are in LDS
      __syncthreads();
    }
  };
  auto phase_body = [&](f32x4(&ACC)[NT][MT], f32x4(&PV)[NT][MT], int b) __attribute__((always_inline)) {
    static_for<NSTEP>([&](auto stc) __attribute__((always_inline)) { step_body(stc, ACC, PV, b); });
  };

  if (FLM_ABLATE & 32) return;  // (prologue only)
  for (int b = 0; b < nph; b += 2) {  // s is a multiple of 4 (shared fifth-tile layout): even phases accumulate in set A
    phase_body(accA, accB, b);
    phase_body(accB, accA, b + 1);
  }
  {  // drain: the last phase (a short one, set B)
    const int bprev = nph - 1, j = (bprev & 3) - 1;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) accB[nt][MT - 1][0] = j == 0 ? x4r[nt][0] : (j == 1 ? x4r[nt][1] : x4r[nt][2]);
    static_for<NOPS>([&](auto kc) __attribute__((always_inline)) { epi_op(kc, accB, bprev); });
    hit_loop(accB, bprev);
  }
  if constexpr (LANE) lane_flush();
  else cand_flush();
}
#undef FLM_PIN

static std::atomic<int> g_cand8_rpw{0};  // A/B knob "up3_cand8_rows": 0 = automatic, else phase rows per workgroup (1, 2, 4, 8)
void convt_cand8_rows(int rpw) { g_cand8_rpw.store(rpw, std::memory_order_relaxed); }

template <bool BF, int WAVES, int STEP_G>
static int launch_cand8(hipStream_t st, ConvTArgs a) {
  using namespace cand8;
  constexpr int NT = Cfg<BF>::NT;
  constexpr size_t lds = lds_bytes(WAVES, STEP_G);
  static_assert(lds * (8 / WAVES) <= 160 * 1024, "cand8: LDS budget of a CU");
  static FuncAttrOnce attr;
  FLM_FUNC_ATTR_ONCE(attr, (&up3_cand8_kernel<BF, WAVES, STEP_G>), lds);
  a.ppf = cdiv((a.hi + 1) * (a.wi + 1), 16 * NT) * 16 * NT;  // per-face padding at wave granularity
  const long long pos = (long long)a.n * a.ppf;
  const int xblocks = (int)((pos + WAVES * 16 * NT - 1) / (WAVES * 16 * NT));
  a.rpw = convt_rows_per_wg(xblocks, a.s, 256 * (8 / WAVES), g_cand8_rpw.load(std::memory_order_relaxed));
  up3_cand8_kernel<BF, WAVES, STEP_G><<<dim3(xblocks, a.s / a.rpw), WAVES * 64, lds, st>>>(a);
  FLM_LAUNCH_CHECK("up3_cand8_kernel");
  return FLM_OK;
}

template <int MT, int G, bool BF, int NT = 1, int MODE = 0, bool SHARE = false>
static int launch_t(hipStream_t st, ConvTArgs a) {
  constexpr bool CAND = MODE == 1;
  constexpr int GCH = BF ? GCH_BF16 : GCH_F32;
  constexpr size_t lds = sizeof(float4) * 2 * GCH * MT * 64 + ((BF && MT == 5 && G == 9) ? sizeof(float4) * 4 * 2 * NT * 64 : 0) +
                         (CAND ? sizeof(unsigned long long) * 4 * kCandWaveCap + sizeof(float) * 4 * 16 * MT : 0) +
                         (MODE == 2 ? sizeof(unsigned) * 4 * kMaxSamplePhases * 16 * MT : 0);
  // two workgroups per CU (160 KiB of LDS) is what the 68-class kernels are scheduled for: a key buffer that pushed the
  // fp32 candidate kernel to 94 KiB cost 22 % of up3
  static_assert(!(MT == 5 && (G == 9 || G == 17)) || MODE == 2 || lds <= 80 * 1024, "convt: LDS budget of two workgroups per CU");
  static FuncAttrOnce attr;
  FLM_FUNC_ATTR_ONCE(attr, (&convt_kernel<MT, G, BF, NT, MODE, SHARE>), lds);
  int xblocks = cdiv(a.P, 64 * NT);
  if (a.ppf > 0 && CAND) {  // per-face padding at wave granularity: a WAVE's 16*NT positions belong to one face
    a.ppf = cdiv((a.hi + 1) * (a.wi + 1), 16 * NT) * 16 * NT;
    xblocks = (int)(((long long)a.n * a.ppf + 64 * NT - 1) / (64 * NT));
  } else if (a.ppf > 0) {   // (sampling launch) a workgroup's 64*NT positions belong to one face
    a.ppf = cdiv((a.hi + 1) * (a.wi + 1), 64 * NT) * 64 * NT;
    xblocks = a.n * (a.ppf / (64 * NT));
  }
  int rows = 1;
  if (!a.sub) {
    rows = convt_rows_per_wg(xblocks, a.s, 512);  // two workgroups per CU
    a.nb = rows * a.s;
  }
  dim3 grid(xblocks, a.sub ? 1 : a.s / rows);
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

// A/B knob (flm_set_tuning "up3_cand8"): bit 0 the 8-wave kernel above for the bf16 candidate launch (default 1), bit 2
// its 4-wave shape, 0 the generic kernel; same keys either way, so it never changes results or layouts.  Bit 1 asked for
// an fp32 form (a tie at best); retired when the generic kernel's fp32 sum became two-level
static std::atomic<int> g_cand8{1};
void convt_cand8_enable(int on) { g_cand8.store(on, std::memory_order_relaxed); }

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
    const int c8 = g_cand8.load(std::memory_order_relaxed);  // bit 0: bf16, bit 1: fp32, bit 2: the 4-wave shape
    // (the 8-wave kernel decodes a phase as (b >> log2 s, b & (s - 1)): s a power of two, checked above)
    if ((d.s & 3) == 0 && d.s >= 4 && (d.s & (d.s - 1)) == 0 && (long long)d.s * d.s * cand8::Cfg<false>::PHASE_BYTES < 0x7fffffffll) {
      if (d.g.bf16 && (c8 & 1)) {
        // (knob "up3_wreg", default off: the weights-in-registers kernel where its shape and scratch conditions hold)
        const int wr = (c8 & 4) ? 0 : launch_up3_wreg(st, a, d.scratch, d.scratch_bytes);
        if (wr) return wr < 0 ? wr : FLM_OK;
        return (c8 & 4) ? launch_cand8<true, 4, 3>(st, a) : launch_cand8<true, 8, 9>(st, a);
      }
      // (bit 1, the fp32 form of that kernel, is retired: the generic kernel's fp32 sum is two-level since round 3 and
      // the 8-wave kernel, a tie in fp32 at best, was not given the same summation tree -- its keys would differ)
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
