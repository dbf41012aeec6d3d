// Implicit-GEMM convolution for bf16 operands on 256-row tiles (v_mfma_f32_32x32x16_bf16, fp32 accumulate):
// the large layers of BASELINE configs[2] (networks/fcn.py:33-48 enc3..5, :98 fc6, :100 fc7).
//
// Why a second tile shape: at 16x the fp32 matrix rate a 128x128 tile asks the memory side for 32 KiB per
// 512 MFMA cycles of each wave -- 64 B/clk/CU, the whole L1 port and twice what the L2 sustains chip-wide
// (about 70 GB/s per CU, MI355X_MICROARCH.md 'Indexed rows: gather into LDS').  A 256x256 tile moves half the
// bytes per MAC:
//   8 waves as 2x4, each 128x64 = 4x2 MFMA tiles (128 accumulator registers), one workgroup per CU;
//   k-step = 128 bytes of every row (64 bf16): 64 KiB per stage, two stages in LDS (128 KiB);
//   per step and wave 4 slices of 8 MFMAs; fragments are read by ds_read_b128 from the same XOR-swizzled image
//   as in flm_igemm.hip.  MFMA order inside a slice is im2col-fragment-major: fragment i is used by two
//   consecutive MFMAs and reloaded for the next slice right after them (6 MFMAs before its next use); the two
//   weight fragments are double-buffered and fetched at the start of the slice before;
//   global -> register -> LDS staging through raw buffer loads (no vector address arithmetic, zero fill of the
//   padding by the buffer bounds check): two register sets for the im2col rows (tile t+2 is loaded during
//   slice 0 of step t and written to LDS during slice 2 of step t+1), one for the weight rows (loaded in slice
//   3, written in slice 2 of the next step) -- 256 VGPRs exactly, nothing spills;
//   ONE barrier per step, after slice 2: by then every wave has issued its last fragment reads of the current
//   stage (slice 3's fragments are fetched during slice 2), so during slice 3 the next stage is already being
//   read and the step boundary has no bubble.
// Measured (batch 512): fc7 0.98 ms = 1.12 PFLOP/s, MFMA pipe 56 % busy at the 1.87 GHz the chip holds under
// this load on random data (MI355X_MICROARCH.md 'DVFS give-back': bf16 MFMA loops on random operands run at
// 1.5-1.9 GHz, not 2.4; its best random-data GEMM is 1.25 PFLOP/s).  An ablation without any global load, LDS
// write, barrier or fragment read still took 0.84 ms: the layer is within 15 % of what the matrix pipe gives
// at that clock.  Tile-group rasterisation and a staggered k order (L2 channel camping test) moved fc7 by < 4 %.
// 128-channel layers (enc2, 9 k-steps per tile) can use 256x128 tiles (8 waves as 4x2, each 64x64), but two
// co-resident 128x128 workgroups overlap each other's prologue and epilogue better: off by default.
#include "flm_igemm_args.h"

namespace flm {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int ROWB = 128;  // bytes of one operand row per k-step

// (raw buffer loads, the out-of-bounds offset and the LDS-DMA request: flm_igemm_args.h)

template <int MMAP, bool RELU, int WM, int WN, int TM, int TN, bool DMA, bool M16 = false>
__global__ __launch_bounds__(WM * WN * 64, 1) void igemm_bf16_big_kernel(IgemmArgs a) {
  static_assert(!M16 || DMA, "the 16x16x32 form exists for the LDS-DMA kernel");
  constexpr int BMt = WM * TM * 32, BNt = WN * TN * 32, NTHR = WM * WN * 64;
  constexpr int RPT = NTHR / 8;                 // rows one staging pass of the workgroup covers
  constexpr int AR = BMt / RPT, BR = BNt / RPT; // staged rows per thread
  constexpr int NLD = AR + BR;
  constexpr int SL = TM * TN;                   // MFMA slots per 16-deep slice
  constexpr int A_BYTES = BMt * ROWB, B_BYTES = BNt * ROWB, STAGE = A_BYTES + B_BYTES;
  static_assert(AR <= SL && BR <= SL, "im2col loads are issued in slice 0, weight loads in slice 3");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned long long* s_mask = reinterpret_cast<unsigned long long*>(smem + 2 * STAGE);  // [WM*WN]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int Lb = xcd_remap(blockIdx.x, gridDim.x);
  // Rasterisation: consecutive logical ids (= the workgroups resident on one XCD at a time, 32 CUs) form a group of
  // gm x gn tiles, so the XCD's L2 serves gm im2col panels and gn weight panels per k-step instead of 32 + 1: with
  // a linear order every byte of fc7's 268 MB input came from beyond L2 sixteen times.
  const int gs = a.gm * a.gn, mgroups = (a.mtiles + a.gm - 1) / a.gm;
  const int grp = Lb / gs, rin = Lb % gs;
  const int mslot = (grp % mgroups) * a.gm + rin % a.gm, nt = (grp / mgroups) * a.gn + rin / a.gm;
  if (mslot >= a.mtiles || nt >= a.ntiles) return;  // padding of the last groups (uniform: before any barrier)
  const int n0 = nt * BNt;
  int mt = mslot;
  if (MMAP == 2) {
    // Position-major tiles differ in work (20..49 taps of fc6's 7x7 on 8x8).  Workgroups are dealt to the CUs in
    // index order, so the tiles are handed out heaviest first (ties by index): the light tiles fill the tail.
    int* wk = reinterpret_cast<int*>(smem);
    int* ord = wk + a.mtiles;
    for (int t = tid; t < a.mtiles; t += NTHR)
      wk[t] = __builtin_popcountll(posmajor_tapmask(t, a.M, a.n, a.h, a.w, a.kh, a.kw, a.pad, BMt));
    __syncthreads();
    for (int t = tid; t < a.mtiles; t += NTHR) {
      const int wt = wk[t];
      int rank = 0;
      for (int u = 0; u < a.mtiles; ++u) rank += (wk[u] > wt) || (wk[u] == wt && u < t);
      ord[rank] = t;
    }
    __syncthreads();
    mt = __builtin_amdgcn_readfirstlane(ord[mslot]);
    __syncthreads();
  }

  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  const int wr = wave / WN, wc = wave % WN;
  // 32x32x16: lane (lr = row of the 32-row tile, lh = which 8 of the slice's 16 k); 16x16x32 (M16): lane (lr = row of the
  // 16-row tile, lh = which 8 of the slice's 32 k).  Either way the fragment is the 16-byte chunk (slice's first chunk +
  // lh) ^ swx of row lr, swx = (row >> 1) & 7 as the staging side wrote it.
  const int lr = M16 ? (lane & 15) : (lane & 31), lh = M16 ? (lane >> 4) : (lane >> 5);
  const int swx = (lr >> 1) & 7;
  // byte offset of this lane's fragment inside a row, per slice s (M16: two 32-deep slices, chunks 4s + lh; else four
  // 16-deep ones, chunks 2s + lh)
  int fc[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) fc[s] = (((M16 ? 4 * (s & 1) : 2 * s) + lh) ^ swx) << 4;
  const int farow = (wr * TM * 32 + lr) * ROWB;
  const int fbrow = A_BYTES + (wc * TN * 32 + lr) * ROWB;

  {
    const int m0 = mt * BMt;

    // ---- staging role: rows r0 + RPT*j, 16-byte chunk c8 ------------------------------------------------
    // DMA: the loads write LDS themselves, lane l of a wave at byte 16*l of the wave's 1 KiB piece (8 rows): the
    // XOR swizzle moves to the source side -- the lane at physical chunk tid&7 fetches the logical chunk that lives there
    const int c8 = DMA ? ((tid & 7) ^ ((tid >> 4) & 7)) : (tid & 7), r0 = tid >> 3;
    int pyx[AR];  // (py << 16) | px of the filter centre in input coordinates; py = 0x7000 for rows past M
    unsigned rowoff[AR];
#pragma unroll
    for (int j = 0; j < AR; ++j) {
      const int m = m0 + r0 + RPT * j;
      const bool pvj = m < a.M;
      const int mm = pvj ? m : 0;
      int px, py, pn;
      if (MMAP == 0) {
        px = (mm % a.wo) * a.stride;
        py = ((mm / a.wo) % a.ho) * a.stride;
        pn = mm / (a.wo * a.ho);
      } else if (MMAP == 1) {
        const int q = mm >> 2, d = mm & 3, wp = a.w >> 1, hp = a.h >> 1;
        px = 2 * (q % wp) + (d & 1);
        py = 2 * ((q / wp) % hp) + (d >> 1);
        pn = q / (wp * hp);
      } else {
        pn = mm % a.n;
        const int pos = mm / a.n;
        py = pos / a.w;
        px = pos % a.w;
      }
      pyx[j] = ((pvj ? py : 0x7000) << 16) | px;
      rowoff[j] = ((unsigned)(((pn * a.h + py) * a.w + px) * a.cin) + 8u * c8) * 2u;
    }

    // ---- filter taps that touch at least one in-bounds pixel of this tile ------------------------------
    const int ntaps = a.kh * a.kw;
    unsigned long long tapmask;
    if (ntaps == 1) {
      tapmask = 1ull;
    } else {
      unsigned long long mymask = 0;
      for (int t = 0; t < ntaps; ++t) {
        const int ky = t / a.kw - a.pad, kx = t % a.kw - a.pad;
        bool any = false;
#pragma unroll
        for (int j = 0; j < AR; ++j)
          any |= (unsigned)((pyx[j] >> 16) + ky) < (unsigned)a.h &&
                 (unsigned)((pyx[j] & 0xffff) + kx) < (unsigned)a.w;
        if (__any(any)) mymask |= 1ull << t;
      }
      if (lane == 0) s_mask[wave] = mymask;
      __syncthreads();
      tapmask = 0;
#pragma unroll
      for (int wv = 0; wv < WM * WN; ++wv) tapmask |= s_mask[wv];
      const unsigned tm_lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)tapmask);
      const unsigned tm_hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(tapmask >> 32));
      tapmask = ((unsigned long long)tm_hi << 32) | (unsigned long long)tm_lo;
    }
    const int nit = __builtin_popcountll(tapmask) * a.cpt;

    // weight rows n0 + r0 + RPT*j: one per-thread offset, the j term is wave-uniform (scalar offset of the load)
    const unsigned wrow0 = ((unsigned)r0 * (unsigned)a.K + 8u * c8) * 2u;
    const int xbias = (a.pad * a.w + a.pad) * a.cin * 2;  // largest negative tap displacement, in bytes
    const __amdgpu_buffer_rsrc_t xsrd = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(a.x)) - xbias, 0, 0x7fffffff, kSrdFlags);
    const __amdgpu_buffer_rsrc_t wsrd = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(a.wt)) + (size_t)n0 * a.K * 2, 0, 0x7fffffff, kSrdFlags);
    const int wstep = RPT * a.K * 2;  // bytes between the weight rows of consecutive j
    // LDS write offset of this thread's chunk (RPT is a multiple of 16: same XOR for every j).  Recomputed in
    // every step from an opaque copy of tid: as a loop invariant it would be the one value spilled to scratch,
    // and its reload's s_waitcnt vmcnt(0) would drain the loads in flight.
#define FLM_ST_A()                                                                  \
  int tid_ = tid;                                                                   \
  asm volatile("" : "+v"(tid_));                                                    \
  const int st_a = ((tid_ >> 3) * ROWB) + (((tid_ & 7) ^ ((tid_ >> 4) & 7)) << 4);

    // Staging registers: two sets for the im2col rows (their first touch of a pixel comes from beyond L2: a
    // full step of latency cover), one for the weight rows (an L2-resident panel every workgroup re-reads).
    float4 ra[2][AR], rb[BR];
#pragma unroll
    for (int z = 0; z < 2; ++z)
#pragma unroll
      for (int j = 0; j < AR; ++j) ra[z][j] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < BR; ++j) rb[j] = make_float4(0.f, 0.f, 0.f, 0.f);

    unsigned long long rem = tapmask;
    int cur_tap = __builtin_ctzll(rem);
    int cur_chunk = 0;
    int ld_ky = 0, ld_kx = 0, ld_delta = 0, ld_koff = 0, ld_c0 = 0;

#define FLM_TILE_PARAMS()                              \
  {                                                    \
    const int ty = (cur_tap * a.kw_magic) >> 16;       \
    ld_ky = ty - a.pad;                                \
    ld_kx = cur_tap - ty * a.kw - a.pad;               \
    ld_c0 = cur_chunk * 128;                           \
    ld_delta = (ld_ky * a.w + ld_kx) * a.cin * 2 + xbias + ld_c0; \
    ld_koff = cur_tap * a.cin * 2 + ld_c0;             \
    rem &= rem - 1;                                    \
    if (rem == 0) {                                    \
      rem = tapmask;                                   \
      if (++cur_chunk == a.cpt) cur_chunk = 0;         \
    }                                                  \
    cur_tap = __builtin_ctzll(rem);                    \
  }
#define FLM_LOAD_A(J, Z)                                                                              \
  {                                                                                                   \
    const int iy = (pyx[J] >> 16) + ld_ky, ix = (pyx[J] & 0xffff) + ld_kx;                            \
    const bool ok_ = (unsigned)iy < (unsigned)a.h && (unsigned)ix < (unsigned)a.w;                   \
    ra[Z][J] = __builtin_bit_cast(                                                                    \
        float4, __builtin_amdgcn_raw_buffer_load_b128(xsrd, ok_ ? rowoff[J] : kOobOffset, ld_delta, 0)); \
  }
#define FLM_LOAD_B(J) \
  rb[J] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(wsrd, wrow0, (J) * wstep + ld_koff, 0));
#define FLM_STORE_A(J, Z, STG) \
  *reinterpret_cast<float4*>(smem + (STG) * STAGE + st_a + (J) * RPT * ROWB) = ra[Z][J];
#define FLM_STORE_B(J, STG) \
  *reinterpret_cast<float4*>(smem + (STG) * STAGE + A_BYTES + st_a + (J) * RPT * ROWB) = rb[J];

    // 32x32x16: TM x TN tiles of 32 x 32 (16 registers each); M16: 2TM x 2TN tiles of 16 x 16 (4 registers each) -- the same
    // 128 accumulator registers, the same fragment bytes, twice the MFMAs at half the cycles each.  The 16x16x32 shape
    // holds a higher clock under this load (tools/mfma_ring_gemm.hip: +2.5..4.5 % on a GEMM of fc7's size, the bf16
    // outputs equal bit for bit).
    constexpr int TI = M16 ? 2 * TM : TM, TJ = M16 ? 2 * TN : TN, TR = M16 ? 16 : 32;  // tiles per wave, rows per tile
    typedef float accv_t __attribute__((ext_vector_type(M16 ? 4 : 16)));
    accv_t acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
      for (int j = 0; j < TJ; ++j)
#pragma unroll
        for (int r = 0; r < (M16 ? 4 : 16); ++r) acc[i][j][r] = 0.f;
    float4 af[TI], bfr[2][TJ];

    // One step on stage BUF: 4 slices x SL MFMAs.  W = register set written to the other stage (slice 2),
    // L = register set that receives the loads of the tile after that (slices 0-1).  Branch-free: past the
    // last tile the loads re-read tile 0 and the stores refill a stage nobody reads.
#define FLM_BIG_STEP(BUF, W, L)                                                                          \
  {                                                                                                      \
    FLM_TILE_PARAMS()                                                                                    \
    FLM_ST_A()                                                                                           \
    _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                                      \
      const int nstage = (s < 3) ? (BUF) : ((BUF) ^ 1);                                                  \
      _Pragma("unroll") for (int i = 0; i < TM; ++i) {                                                   \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                                 \
          const int slot = (s * TM + i) * TN + j;                                                        \
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[i]),          \
                                                              __builtin_bit_cast(bf16x8, bfr[s & 1][j]), acc[i][j], 0, 0, 0); \
          /* weight fragments of the next slice: fetched at the start of this one (the other buffer) */   \
          if (i == 0)                                                                                    \
            bfr[(s + 1) & 1][j] = *reinterpret_cast<const float4*>(smem + nstage * STAGE + fbrow + j * 32 * ROWB + fc[(s + 1) & 3]); \
          /* im2col fragment i: reloaded right after its last use, TM*TN - TN MFMAs before its next */    \
          if (j == TN - 1)                                                                               \
            af[i] = *reinterpret_cast<const float4*>(smem + nstage * STAGE + farow + i * 32 * ROWB + fc[(s + 1) & 3]);  \
          _Pragma("unroll") for (int k = 0; k < AR; ++k)                                                 \
            if (slot == k) FLM_LOAD_A(k, L)                                                              \
          _Pragma("unroll") for (int k = 0; k < BR; ++k)                                                 \
            if (slot == 3 * SL + k) FLM_LOAD_B(k)                                                        \
          _Pragma("unroll") for (int k = 0; k < NLD; ++k) {                                              \
            if (slot == 2 * SL + (k * SL) / NLD) {                                                       \
              if (k < AR) { FLM_STORE_A((k < AR ? k : 0), W, (BUF) ^ 1) }                                \
              else { FLM_STORE_B((k >= AR ? k - AR : 0), (BUF) ^ 1) }                                    \
            }                                                                                            \
          }                                                                                              \
          if (slot == 3 * SL - 1) __syncthreads();                                                       \
          __builtin_amdgcn_sched_barrier(0);                                                             \
        }                                                                                                \
      }                                                                                                  \
    }                                                                                                    \
  }

    // LDS-DMA form of the step (buffer_load_dwordx4 ... lds): no staging registers, no ds_write pass.  Tile t+2 is
    // requested during slice 3 of step t, straight into the stage whose last fragment reads the barrier of this step
    // has just retired, and must have landed by the barrier of step t+1 (its vmcnt(0)): three slices of cover.
    // Written as inline assembly: hipcc orders every ds_read behind a pending LDS-DMA of its own (s_waitcnt vmcnt(0)
    // before each fragment read -- the requests would land one by one); the assembly is opaque to that bookkeeping
    // and the one wait that is needed, vmcnt(0) ahead of the step's barrier, is written out.
    typedef __attribute__((address_space(3))) char lds_char;
    const unsigned dma_base = (unsigned)(size_t)((lds_char*)smem) + __builtin_amdgcn_readfirstlane(wave) * 8 * ROWB;
    const dma_srd xdma = dma_make_srd(reinterpret_cast<const char*>(a.x) - xbias);
    const dma_srd wdma = dma_make_srd(reinterpret_cast<const char*>(a.wt) + (size_t)n0 * a.K * 2);
#define FLM_DMA_A(J, STG)                                                                             \
  {                                                                                                   \
    const int iy = (pyx[J] >> 16) + ld_ky, ix = (pyx[J] & 0xffff) + ld_kx;                            \
    const bool ok_ = (unsigned)iy < (unsigned)a.h && (unsigned)ix < (unsigned)a.w;                   \
    dma_load16(xdma, dma_base + (STG) * STAGE + (J) * RPT * ROWB, ok_ ? rowoff[J] : kOobOffset, ld_delta); \
  }
#define FLM_DMA_B(J, STG) \
  dma_load16(wdma, dma_base + (STG) * STAGE + A_BYTES + (J) * RPT * ROWB, wrow0, (J) * wstep + ld_koff);
#define FLM_DMA_STEP(BUF)                                                                                \
  {                                                                                                      \
    FLM_TILE_PARAMS()                                                                                    \
    _Pragma("unroll") for (int s = 0; s < 4; ++s) {                                                      \
      const int nstage = (s < 3) ? (BUF) : ((BUF) ^ 1);                                                  \
      _Pragma("unroll") for (int i = 0; i < TM; ++i) {                                                   \
        _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                                 \
          const int slot = (s * TM + i) * TN + j;                                                        \
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[i]),          \
                                                              __builtin_bit_cast(bf16x8, bfr[s & 1][j]), acc[i][j], 0, 0, 0); \
          if (i == 0)                                                                                    \
            bfr[(s + 1) & 1][j] = *reinterpret_cast<const float4*>(smem + nstage * STAGE + fbrow + j * 32 * ROWB + fc[(s + 1) & 3]); \
          if (j == TN - 1)                                                                               \
            af[i] = *reinterpret_cast<const float4*>(smem + nstage * STAGE + farow + i * 32 * ROWB + fc[(s + 1) & 3]);  \
          _Pragma("unroll") for (int k = 0; k < NLD; ++k) {                                              \
            if (slot == 3 * SL + (k * SL) / NLD) {                                                       \
              if (k < AR) { FLM_DMA_A((k < AR ? k : 0), BUF) }                                           \
              else { FLM_DMA_B((k >= AR ? k - AR : 0), BUF) }                                            \
            }                                                                                            \
          }                                                                                              \
          if (slot == 3 * SL - 1) {                                                                      \
            __builtin_amdgcn_s_waitcnt(0x0f70); /* vmcnt(0): this wave's pieces of the next tile are in LDS */ \
            __syncthreads();                                                                             \
          }                                                                                              \
          __builtin_amdgcn_sched_barrier(0);                                                             \
        }                                                                                                \
      }                                                                                                  \
    }                                                                                                    \
  }

    // The 16x16x32 form of the step: two 32-deep slices of 2TM x 2TN MFMAs.  Every fragment read of the current stage is
    // issued by the end of slice 0 (slice 1's fragments are fetched during it), so the barrier sits in the MIDDLE of the
    // step, and the requests of tile t+2 -- into the stage that barrier has just retired -- spread over the second half.
#define FLM_DMA_STEP16(BUF)                                                                              \
  {                                                                                                      \
    FLM_TILE_PARAMS()                                                                                    \
    _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                                      \
      const int nstage = (s == 0) ? (BUF) : ((BUF) ^ 1);                                                 \
      _Pragma("unroll") for (int j = 0; j < TJ; ++j) {                                                   \
        _Pragma("unroll") for (int i = 0; i < TI; ++i) {                                                 \
          const int slot = (s * TJ + j) * TI + i;                                                        \
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[i]),          \
                                                              __builtin_bit_cast(bf16x8, bfr[0][j]), acc[i][j], 0, 0, 0); \
          /* weight-major order, every fragment single-buffered: weight fragment j is reloaded for the next slice  */ \
          /* after its eight MFMAs (24 before its next use), im2col fragment i after its last use of the slice      */ \
          /* (8 before its next) -- 48 fragment registers instead of 64, which spilled                             */ \
          if (i == TI - 1)                                                                               \
            bfr[0][j] = *reinterpret_cast<const float4*>(smem + nstage * STAGE + fbrow + j * 16 * ROWB + fc[(s + 1) & 1]); \
          if (j == TJ - 1)                                                                               \
            af[i] = *reinterpret_cast<const float4*>(smem + nstage * STAGE + farow + i * 16 * ROWB + fc[(s + 1) & 1]);  \
          _Pragma("unroll") for (int k = 0; k < NLD; ++k) {                                              \
            if (slot == TI * TJ + (k * TI * TJ) / NLD) {                                                 \
              if (k < AR) { FLM_DMA_A((k < AR ? k : 0), BUF) }                                           \
              else { FLM_DMA_B((k >= AR ? k - AR : 0), BUF) }                                            \
            }                                                                                            \
          }                                                                                              \
          if (slot == TI * TJ - 1) {                                                                     \
            __builtin_amdgcn_s_waitcnt(0x0f70); /* vmcnt(0): this wave's pieces of the next tile are in LDS */ \
            __syncthreads();                                                                             \
          }                                                                                              \
          __builtin_amdgcn_sched_barrier(0);                                                             \
        }                                                                                                \
      }                                                                                                  \
    }                                                                                                    \
  }

    if constexpr (DMA) {
      if (nit > 0) {
        FLM_TILE_PARAMS()
#pragma unroll
        for (int j = 0; j < AR; ++j) FLM_DMA_A(j, 0)
#pragma unroll
        for (int j = 0; j < BR; ++j) FLM_DMA_B(j, 0)
        FLM_TILE_PARAMS()  // tile 1 (past the end: tile 0 again, never read)
#pragma unroll
        for (int j = 0; j < AR; ++j) FLM_DMA_A(j, 1)
#pragma unroll
        for (int j = 0; j < BR; ++j) FLM_DMA_B(j, 1)
      }
      __builtin_amdgcn_s_waitcnt(0x0f70);
      __syncthreads();
    } else
    // prologue: tile 0 -> set 0 -> stage 0; tile 1 -> set 1 (written to stage 1 during step 0)
    if (nit > 0) {
      FLM_ST_A()
      FLM_TILE_PARAMS()
#pragma unroll
      for (int j = 0; j < AR; ++j) FLM_LOAD_A(j, 0)
#pragma unroll
      for (int j = 0; j < BR; ++j) FLM_LOAD_B(j)
#pragma unroll
      for (int j = 0; j < AR; ++j) FLM_STORE_A(j, 0, 0)
#pragma unroll
      for (int j = 0; j < BR; ++j) FLM_STORE_B(j, 0)
      FLM_TILE_PARAMS()  // tile 1 (past the end: tile 0 again, never read)
#pragma unroll
      for (int j = 0; j < AR; ++j) FLM_LOAD_A(j, 1)
#pragma unroll
      for (int j = 0; j < BR; ++j) FLM_LOAD_B(j)
    }
    if (!DMA) __syncthreads();
#pragma unroll
    for (int i = 0; i < TI; ++i) af[i] = *reinterpret_cast<const float4*>(smem + farow + i * TR * ROWB + fc[0]);
#pragma unroll
    for (int j = 0; j < TJ; ++j) bfr[0][j] = *reinterpret_cast<const float4*>(smem + fbrow + j * TR * ROWB + fc[0]);

    if constexpr (M16) {
      for (int it = 0; it < nit; it += 2) {
        FLM_DMA_STEP16(0)
        if (it + 1 < nit) FLM_DMA_STEP16(1)
      }
      __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): the requests past the last tile still write LDS
    } else if constexpr (DMA) {
      for (int it = 0; it < nit; it += 2) {
        FLM_DMA_STEP(0)
        if (it + 1 < nit) FLM_DMA_STEP(1)
      }
      __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): the requests past the last tile still write LDS
    } else {
      for (int it = 0; it < nit; it += 2) {
        FLM_BIG_STEP(0, 1, 0)
        if (it + 1 < nit) FLM_BIG_STEP(1, 0, 1)
      }
    }
#undef FLM_DMA_A
#undef FLM_DMA_B
#undef FLM_DMA_STEP
#undef FLM_DMA_STEP16

#undef FLM_TILE_PARAMS
#undef FLM_LOAD_A
#undef FLM_LOAD_B
#undef FLM_STORE_A
#undef FLM_STORE_B
#undef FLM_BIG_STEP
#undef FLM_ST_A

    // ---- epilogue of the 16x16x32 form: column = lane & 15, row = 4*(lane >> 4) + r of a 16 x 16 tile -- again four
    // consecutive rows of one channel per lane, so the same quad transpose gives 8-byte stores; a pooled layer's quad is the
    // four registers of one tile, and the four values a lane transposes come from four row tiles.
    if constexpr (M16) {
      // lane coordinates afresh (lane id from v_mbcnt, wave id from a scalar register): kept alive across the k-loop,
      // which uses all 256 registers, they -- in the end threadIdx.x itself -- were spilled to scratch
      const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
      const int wave = wave_s, wr = wave / WN, wc = wave % WN, lr = lane & 15, lh = lane >> 4;
      const int q = lane & 3;
      const bool packed = !a.out_f32 && !(a.cout & 3) && !(a.ldc & 3);
      const int mrow0 = m0 + wr * TM * 32 + 4 * lh;
      // (an opaque copy of the face count: the reciprocal the row set-up divides by would otherwise stay alive across
      // the k-loop for the divisions below -- in a vector register, i.e. in scratch)
      int faces_n = a.n;
      asm volatile("" : "+s"(faces_n));
      unsigned short* y16 = reinterpret_cast<unsigned short*>(a.y);
      float* y32 = reinterpret_cast<float*>(a.y);
#pragma unroll
      for (int j = 0; j < TJ; ++j) {
        const int col = n0 + wc * TN * 32 + 16 * j + lr;
        const bool cok = col < a.cout;
        const float sc = a.scale[cok ? col : 0], sh = a.shift[cok ? col : 0];
        const int col4 = col & ~3;
        if (MMAP == 1) {
#pragma unroll
          for (int ib = 0; ib < TI / 4; ++ib) {
            float v[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              v[g] = -3.402823466e38f;
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                float u = fmaf(acc[4 * ib + g][j][e], sc, sh);
                if (RELU) u = fminf(fmaxf(u, 0.f), a.relu_max);
                v[g] = fmaxf(v[g], u);
              }
            }
            if (packed) {
              const uint2 t = quad_transpose_bf16(v[0], v[1], v[2], v[3], q);
              const int m = mrow0 + 16 * (4 * ib + q);
              if (cok && m < a.M) *reinterpret_cast<uint2*>(y16 + (size_t)(m >> 2) * a.ldc + col4) = t;
            } else {
#pragma unroll
              for (int g = 0; g < 4; ++g) {
                const int m = mrow0 + 16 * (4 * ib + g);
                if (cok && m < a.M) {
                  const size_t o = (size_t)(m >> 2) * a.ldc + col;
                  if (!a.out_f32) y16[o] = f2bf(v[g]);
                  else y32[o] = v[g];
                }
              }
            }
          }
        } else {
#pragma unroll
          for (int i = 0; i < TI; ++i) {
            const int m_first = mrow0 + 16 * i;
            const int nn0 = MMAP == 2 ? m_first % faces_n : 0, pos0 = MMAP == 2 ? m_first / faces_n : 0;
            float u[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              u[e] = fmaf(acc[i][j][e], sc, sh);
              if (RELU) u[e] = fminf(fmaxf(u[e], 0.f), a.relu_max);
            }
            if (packed) {
              const uint2 t = quad_transpose_bf16(u[0], u[1], u[2], u[3], q);
              const int m = m_first + q;
              if (cok && m < a.M) {
                const size_t orow = MMAP == 2 ? posmajor_orow(m, m_first, nn0, pos0, faces_n, a.h * a.w) : (size_t)m;
                *reinterpret_cast<uint2*>(y16 + orow * a.ldc + col4) = t;
              }
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const int m = m_first + e;
                if (cok && m < a.M) {
                  const size_t orow = MMAP == 2 ? posmajor_orow(m, m_first, nn0, pos0, faces_n, a.h * a.w) : (size_t)m;
                  if (!a.out_f32) y16[orow * a.ldc + col] = f2bf(u[e]);
                  else y32[orow * a.ldc + col] = u[e];
                }
              }
            }
          }
        }
      }
      return;
    } else {
    // ---- epilogue: y = acc*scale + shift, ReLU, 2x2 max-pool (MMAP 1), store -------------------------------
    // accumulator layout: column = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    // bf16 outputs (every layer but the last of a chain) leave transposed over lane quads: 8-byte stores of four
    // channels of one pixel instead of four 2-byte stores (a quarter of the store instructions).
    if (!a.out_f32 && !(a.cout & 3) && !(a.ldc & 3)) {
      const int q = lane & 3;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = n0 + wc * TN * 32 + 32 * j + lr;
        const bool cok = col < a.cout;
        const float sc = a.scale[cok ? col : 0], sh = a.shift[cok ? col : 0];
        const int col4 = col & ~3;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const int mbase = m0 + wr * TM * 32 + 32 * i;
          if (MMAP == 1) {
            float v[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              v[g] = -3.402823466e38f;
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                float u = fmaf(acc[i][j][4 * g + e], sc, sh);
                if (RELU) u = fminf(fmaxf(u, 0.f), a.relu_max);
                v[g] = fmaxf(v[g], u);
              }
            }
            const uint2 t = quad_transpose_bf16(v[0], v[1], v[2], v[3], q);
            const int m = mbase + 8 * q + 4 * lh;  // pooled row g = q
            if (cok && m < a.M) *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(a.y) + (size_t)(m >> 2) * a.ldc + col4) = t;
          } else {
            const int m_first = mbase + 4 * lh;
            const int nn0 = MMAP == 2 ? m_first % a.n : 0, pos0 = MMAP == 2 ? m_first / a.n : 0;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              float u[4];
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                u[e] = fmaf(acc[i][j][4 * g + e], sc, sh);
                if (RELU) u[e] = fminf(fmaxf(u[e], 0.f), a.relu_max);
              }
              const uint2 t = quad_transpose_bf16(u[0], u[1], u[2], u[3], q);
              const int m = m_first + 8 * g + q;
              if (cok && m < a.M) {
                const size_t orow = MMAP == 2 ? posmajor_orow(m, m_first, nn0, pos0, a.n, a.h * a.w) : (size_t)m;
                *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(a.y) + orow * a.ldc + col4) = t;
              }
            }
          }
        }
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wc * TN * 32 + 32 * j + lr;
      const bool cok = col < a.cout;
      const float sc = a.scale[col], sh = a.shift[col];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int mbase = m0 + wr * TM * 32 + 32 * i;
        if (MMAP == 1) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            float v = -3.402823466e38f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float u = fmaf(acc[i][j][4 * g + e], sc, sh);
              if (RELU) u = fminf(fmaxf(u, 0.f), a.relu_max);
              v = fmaxf(v, u);
            }
            const int m = mbase + 8 * g + 4 * lh;
            if (cok && m < a.M) {
              const size_t o = (size_t)(m >> 2) * a.ldc + col;
              if (!a.out_f32) reinterpret_cast<unsigned short*>(a.y)[o] = f2bf(v);
              else reinterpret_cast<float*>(a.y)[o] = v;
            }
          }
        } else {
          const int m_first = mbase + 4 * lh;
          const int nn0 = MMAP == 2 ? m_first % a.n : 0, pos0 = MMAP == 2 ? m_first / a.n : 0;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int m = mbase + (r & 3) + 8 * (r >> 2) + 4 * lh;
            float u = fmaf(acc[i][j][r], sc, sh);
            if (cok && m < a.M) {
              const size_t orow = MMAP == 2 ? posmajor_orow(m, m_first, nn0, pos0, a.n, a.h * a.w) : (size_t)m;
              const size_t o = orow * a.ldc + col;
              if (RELU) u = fminf(fmaxf(u, 0.f), a.relu_max);
              if (!a.out_f32) reinterpret_cast<unsigned short*>(a.y)[o] = f2bf(u);
              else reinterpret_cast<float*>(a.y)[o] = u;
            }
          }
        }
      }
    }
    }  // (32x32x16 epilogue)
  }
}

// A/B knobs (flm_set_tuning): atomics read at launch time; they never change results or memory layouts
static std::atomic<int> g_big_enable{1};  // 0 never, 1 when the tile grid fills the chip, 2 whenever the shape allows (tests),
                              // 3 like 1 plus the 256x128 shape for 128-channel layers
static std::atomic<int> g_group_n{0};     // weight panels per tile group (0: default)
void igemm_bf16_big_enable(int on) { g_big_enable = on; }
void igemm_bf16_group_n(int gn) { g_group_n = gn; }

static std::atomic<int> g_big_dma{1};  // operands reach LDS by buffer_load ... lds (256x256 tiles); 0: through staging registers
void igemm_bf16_big_dma(int on) { g_big_dma = on; }

static std::atomic<int> g_big_m16{1};  // the 256x256 LDS-DMA kernel computes with v_mfma_f32_16x16x32_bf16 (1) or 32x32x16 (0); same bits
void igemm_bf16_big_m16(int on) { g_big_m16 = on; }

template <int MMAP, bool RELU, int WM, int WN, int TM, int TN, bool DMA, bool M16 = false>
static int launch_big_t(hipStream_t s, IgemmArgs a) {
  constexpr int BMt = WM * TM * 32, BNt = WN * TN * 32;
  constexpr size_t lds = 2 * (size_t)(BMt + BNt) * ROWB + 64;
  static FuncAttrOnce attr;
  FLM_FUNC_ATTR_ONCE(attr, (&igemm_bf16_big_kernel<MMAP, RELU, WM, WN, TM, TN, DMA, M16>), lds);
  a.mtiles = cdiv(a.M, BMt);
  a.ntiles = cdiv(a.cout, BNt);
  // fc6 (position-major): its 100 MB of weights are the big operand, one weight panel per group keeps it in L2
  const int group_n = g_group_n.load(std::memory_order_relaxed);
  a.gn = group_n > 0 ? group_n : (MMAP == 2 ? 1 : 4);
  if (a.gn > a.ntiles) a.gn = a.ntiles;
  a.gm = 32 / a.gn > 0 ? 32 / a.gn : 1;
  const int nblk = cdiv(a.mtiles, a.gm) * a.gm * cdiv(a.ntiles, a.gn) * a.gn;
  igemm_bf16_big_kernel<MMAP, RELU, WM, WN, TM, TN, DMA, M16><<<nblk, WM * WN * 64, lds, s>>>(a);
  FLM_LAUNCH_CHECK("igemm_bf16_big_kernel");
  return 1;
}

template <int MMAP, bool RELU, int WM, int WN, int TM, int TN>
static int launch_big(hipStream_t s, const IgemmArgs& a) {
  if (WM == 2 && g_big_dma && g_big_m16)
    return launch_big_t<MMAP, RELU, WM, WN, TM, TN, (WM == 2), (WM == 2)>(s, a);
  if (WM == 2 && g_big_dma) return launch_big_t<MMAP, RELU, WM, WN, TM, TN, (WM == 2)>(s, a);
  return launch_big_t<MMAP, RELU, WM, WN, TM, TN, false>(s, a);
}

template <int WM, int WN, int TM, int TN>
static int dispatch_big(hipStream_t s, const IgemmArgs& a, int relu, int pool, int posmajor) {
  if (pool) return relu ? launch_big<1, true, WM, WN, TM, TN>(s, a) : launch_big<1, false, WM, WN, TM, TN>(s, a);
  if (posmajor) return relu ? launch_big<2, true, WM, WN, TM, TN>(s, a) : launch_big<2, false, WM, WN, TM, TN>(s, a);
  return relu ? launch_big<0, true, WM, WN, TM, TN>(s, a) : launch_big<0, false, WM, WN, TM, TN>(s, a);
}

int launch_igemm_bf16_big(hipStream_t s, const IgemmArgs& a, int relu, int pool, int posmajor, int coutpad) {
  if (!g_big_enable || a.ksplit > 1 || a.stride != 1 || a.res) return 0;
  // buffer offsets at or above 0x80000000 mean "zero padding" here: operands must stay below 2 GiB
  if ((long long)a.n * a.h * a.w * a.cin * 2 + (1 << 20) >= (1ll << 31) || (long long)coutpad * a.K * 2 >= (1ll << 31)) return 0;
  // 256-wide N tiles need whole 256-row weight panels; 128-channel layers take the 256x128 shape
  const bool wide = (coutpad % 256 == 0) && a.cout > 128;
  const int bn = wide ? 256 : 128;
  if (!wide && a.cout > 128) return 0;
  const long long tiles = (long long)cdiv(a.M, 256) * cdiv(a.cout, bn);
  if (g_big_enable != 2) {
    if (tiles < 192) return 0;  // too few workgroups for 256 CUs: 128x128 tiles fill the chip better
    // 128-channel layers (enc2: 9 k-steps per tile) gain nothing from one 256x128 workgroup per CU over two
    // co-resident 128x128 ones, which overlap each other's prologue and epilogue
    if (!wide && g_big_enable != 3) return 0;
  }
  return wide ? dispatch_big<2, 4, 4, 2>(s, a, relu, pool, posmajor) : dispatch_big<4, 2, 2, 2>(s, a, relu, pool, posmajor);
}

}  // namespace flm
