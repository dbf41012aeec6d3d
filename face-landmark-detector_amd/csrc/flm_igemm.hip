// Implicit-GEMM convolution on the matrix cores, one kernel body for two arithmetic types:
//   fp32  v_mfma_f32_32x32x2_f32   (exact fp32 products and sums)                    -- the parity path
//   bf16  v_mfma_f32_32x32x16_bf16 (bf16 operands, fp32 accumulate, 16x the rate)  -- BASELINE configs[2]
// Both use 128-byte operand rows (32 floats / 64 bf16) and 16-byte fragments per lane, so the tile
// geometry, the LDS swizzle and the fragment addressing are shared; the MFMA slot and the output
// conversion differ, and since round 3 the fp32 path has its own operand ring and summation tree
// (template TWO: LDS-DMA staging, three accumulator sets -- chains of 32 + sqrt + sqrt roundings per
// output instead of one fmaf chain of K; the register-staged single-chain form stays selectable for A/B).
//
// Covers every Conv2D of the reference's fcn_8 + vanilla_encoder except enc1
// (networks/fcn.py:33-48 enc2..5 with ZeroPadding2D(1)+BN+ReLU+MaxPool fused; :98 fc6 7x7 'same';
// :100 fc7; :103,108,117 score convs).
//
//   C[m][o] = sum_k A[m][k] * Wt[o][k]      m = output pixel, k = (ky,kx,c), o = output channel
//
// Tile: 128 pixels x 128 channels per 256-thread workgroup, BK = 32 (one 128-byte run of input
// channels of one filter tap, so the im2col gather is a row of 16-byte loads).  4 waves as 2x2,
// each 64x64 = 2x2 MFMA tiles of 32x32.  Operands are staged global -> registers -> LDS with the
// next k-step's loads in flight under the current step's 64 MFMAs; LDS rows are 128 B with the
// 16-byte chunk index XOR-swizzled by (row>>1)&7 so ds_read_b128 by 32 consecutive rows is
// conflict-free.  K is consumed in a permuted order (lane half h of MFMA step s of group t reads
// k = 8t+4h+s) applied identically to both operands, so one ds_read_b128 feeds four MFMAs.
//
// Pixel order along M (template MMAP):
//   0  row-major (n, y, x)
//   1  2x2 quads (n, y/2, x/2, y&1, x&1): the four rows of a max-pool window are registers
//      4j..4j+3 of one lane in the 32x32 accumulator layout, so the pool is an in-lane max and
//      the pooled pixel index is m>>2.
//   2  position-major (y, x, n): a tile holds one or two neighbouring spatial positions of many faces,
//      so filter taps that fall outside the 8x8 map for the whole tile are skipped (fc6: 7x7 'same' on
//      8x8 -- 38 % of its dense MACs multiply zero padding).  Tiles then differ in work (20..49 taps):
//      every workgroup ranks the tiles by tap count (a few hundred integer ops); the bf16 / single-chain form runs
//      the i-th heaviest and the i-th lightest back to back, the fp32 parity form (TWO) one tile per workgroup,
//      handed out longest-first inside each XCD's share of the grid -- either way the generations end together.
#include <vector>

#include "flm_igemm_args.h"

// Developer variants (tools/ab_variants.py builds the file with -DFLM_IGEMM_VAR=<mask>; 0 in every shipped build):
// 4 no vmcnt wait before the step's barrier, 8 no LDS-DMA requests in the k-loop, 16 no barrier (4 / 8 / 16: timing
// ablations, results wrong), 1 no third accumulation level, 2 two fragment address registers + one v_xor per read instead of eight registers
// (chunk (2t + lh) ^ swx = ((lh ^ swx) ^ 2t): group t's address is group 0's with bits 5-6 flipped; frees six registers,
// costs 1.5 % of a layer).
#ifndef FLM_IGEMM_VAR
#define FLM_IGEMM_VAR 0
#endif

namespace flm {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int BM = 128, BN = 128, BK = 32;   // BK in 4-byte units: a k-step is 128 bytes of every row
constexpr int TILE_F = BM * BK;             // 4-byte units per operand tile (16 KiB)

__device__ __forceinline__ int swz(int row, int chunk) { return row * BK + ((chunk ^ ((row >> 1) & 7)) << 2); }

__device__ __forceinline__ float f4c(const float4& v, int i) { return i == 0 ? v.x : (i == 1 ? v.y : (i == 2 ? v.z : v.w)); }

// One of the 16 MFMA slots of a k-step.
//   fp32: component IDX of the four 16-byte fragments feeds all four 32x32 tiles (k = 8t+4h+IDX);
//   bf16: the whole fragments (8 bf16 = k 8h..8h+7 of a 16-deep step) feed tile IDX = 2*i + j.
template <bool BF, int IDX>
__device__ __forceinline__ void mfma_slot(const float4& a0, const float4& a1, const float4& b0, const float4& b1,
                                          f32x16& c00, f32x16& c01, f32x16& c10, f32x16& c11) {
  if constexpr (BF) {
    const float4& av = (IDX >> 1) ? a1 : a0;
    const float4& bv = (IDX & 1) ? b1 : b0;
    const bf16x8 af = __builtin_bit_cast(bf16x8, av), bfr = __builtin_bit_cast(bf16x8, bv);
    if constexpr (IDX == 0) c00 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, c00, 0, 0, 0);
    if constexpr (IDX == 1) c01 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, c01, 0, 0, 0);
    if constexpr (IDX == 2) c10 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, c10, 0, 0, 0);
    if constexpr (IDX == 3) c11 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, c11, 0, 0, 0);
  } else {
    c00 = __builtin_amdgcn_mfma_f32_32x32x2f32(f4c(a0, IDX), f4c(b0, IDX), c00, 0, 0, 0);
    c01 = __builtin_amdgcn_mfma_f32_32x32x2f32(f4c(a0, IDX), f4c(b1, IDX), c01, 0, 0, 0);
    c10 = __builtin_amdgcn_mfma_f32_32x32x2f32(f4c(a1, IDX), f4c(b0, IDX), c10, 0, 0, 0);
    c11 = __builtin_amdgcn_mfma_f32_32x32x2f32(f4c(a1, IDX), f4c(b1, IDX), c11, 0, 0, 0);
  }
}

template <bool BF, int MMAP, bool RELU, bool TWO = false>
__global__ __launch_bounds__(256, 2) void igemm_kernel(IgemmArgs a) {
  static_assert(!TWO || !BF, "the multi-level accumulation is the fp32 (parity) form");
  constexpr int ES = BF ? 2 : 4;    // operand element size
  constexpr int EPC = 16 / ES;      // elements per 16-byte chunk
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* As = reinterpret_cast<float*>(smem_raw);  // [2][TILE_F]
  float* Bs = As + 2 * TILE_F;                     // [2][TILE_F]
  unsigned long long* s_mask = reinterpret_cast<unsigned long long*>(Bs + 2 * TILE_F);  // [4]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int L = xcd_remap(blockIdx.x, gridDim.x);
  // MMAP 2: a workgroup owns the tile pair (i, MT-1-i); otherwise one tile.  (TWO: one tile per workgroup here as well,
  // handed out heaviest first -- workgroups are dealt to the CUs in index order, so the light tiles fill the tail, as in
  // flm_igemm_bf16.hip.  The loop over the pair kept set-up values alive across a k-loop that has no register to spare.)
  constexpr bool PAIR = (MMAP == 2) && !TWO;
  const int mslots = PAIR ? (a.mtiles + 1) / 2 : a.mtiles;
  int mslot = L % mslots, nt = L / mslots;
  if (MMAP == 2 && TWO && (a.ntiles & 7) == 0) {
    // An XCD is dealt the logical ids of ntiles / 8 weight panels x all tiles, in order.  Panel-major, heaviest first
    // inside each panel, that order starts the last panel's heavy tiles late: on the batch-64 tile weights
    // (49,49,49,49,42,...,20 taps) the list schedule ends at 84 units where 66 is ideal.  Rank-major across the XCD's
    // panels it is longest-first: 69, what the pairing of the other form reaches by construction.
    const int ppx = a.ntiles >> 3, per_xcd = mslots * ppx;
    const int xcd = L / per_xcd, l = L - xcd * per_xcd;
    mslot = l / ppx;
    nt = xcd * ppx + (l - mslot * ppx);
  }
  const int n0 = nt * BN;
  const int npass = (PAIR && mslot != a.mtiles - 1 - mslot) ? 2 : 1;
  int pair_mt[2] = {mslot, mslot};
  if (MMAP == 2) {
    // rank the tiles by work, heaviest first (ties by index); LDS scratch = the still unused A buffer
    // tap masks of all tiles, one (tile, position) pair per thread and round: a tile spans up to BM/n + 1 positions
    // -- every one of the map at a single face -- and a lone thread walking positions x taps of its tile cost 100 us
    // of the 180 us this layer took at one face
    unsigned long long* tmask = reinterpret_cast<unsigned long long*>(As);
    int* wk = reinterpret_cast<int*>(tmask + a.mtiles);
    int* ord = wk + a.mtiles;
    for (int t = tid; t < a.mtiles; t += 256) tmask[t] = 0ull;
    __syncthreads();
    const int npp = (BM - 1) / a.n + 2;
    for (int idx = tid; idx < a.mtiles * npp; idx += 256) {
      const int t = idx / npp, m_lo = t * BM, m_hi = (m_lo + BM < a.M ? m_lo + BM : a.M) - 1;
      const int p = m_lo / a.n + (idx - t * npp);
      if (p <= m_hi / a.n) atomicOr(&tmask[t], posmajor_posmask(posmajor_pos(a, p), a.h, a.w, a.kh, a.kw, a.pad));
    }
    __syncthreads();
    for (int t = tid; t < a.mtiles; t += 256) wk[t] = __builtin_popcountll(tmask[t]);
    __syncthreads();
    for (int t = tid; t < a.mtiles; t += 256) {
      const int wt = wk[t];
      int rank = 0;
      for (int u = 0; u < a.mtiles; ++u) rank += (wk[u] > wt) || (wk[u] == wt && u < t);
      ord[rank] = t;
    }
    __syncthreads();
    pair_mt[0] = TWO ? __builtin_amdgcn_readfirstlane(ord[mslot]) : ord[mslot];
    pair_mt[1] = ord[a.mtiles - 1 - mslot];
    __syncthreads();
  }
  for (int pass = 0; pass < npass; ++pass) {
  const int mt = pair_mt[pass];
  const int m0 = mt * BM;
  if (pass) __syncthreads();  // the previous tile's last fragment reads are done before LDS is refilled

  // ---- staging role: rows r0+32j, 16-byte chunk c8 of the 128-byte k-run ----------------------
  // (TWO: the loads write LDS themselves, lane l of a wave at byte 16*l of the wave's 1 KiB piece = 8 rows, so the XOR
  // swizzle moves to the source side: the lane at physical chunk tid&7 fetches the logical chunk that lives there)
  const int c8 = TWO ? ((tid & 7) ^ ((tid >> 4) & 7)) : (tid & 7), r0 = tid >> 3;
  int pn[4], py[4], px[4];
  bool pv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int m = m0 + r0 + 32 * j;
    pv[j] = m < a.M;
    const int mm = pv[j] ? m : 0;
    if (MMAP == 0) {  // (py, px) are INPUT coordinates of the filter centre: output pixel * stride
      px[j] = (mm % a.wo) * a.stride;
      py[j] = ((mm / a.wo) % a.ho) * a.stride;
      pn[j] = mm / (a.wo * a.ho);
    } else if (MMAP == 1) {
      const int q = mm >> 2, d = mm & 3, wp = a.w >> 1, hp = a.h >> 1;
      px[j] = 2 * (q % wp) + (d & 1);
      py[j] = 2 * ((q / wp) % hp) + (d >> 1);
      pn[j] = q / (wp * hp);
    } else {
      pn[j] = mm % a.n;
      const int pos = posmajor_pos(a, mm / a.n);
      py[j] = pos / a.w;
      px[j] = pos % a.w;
    }
  }

  // ---- which filter taps touch at least one in-bounds pixel of this tile ------------------------
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
      for (int j = 0; j < 4; ++j)
        any |= pv[j] && (unsigned)(py[j] + ky) < (unsigned)a.h && (unsigned)(px[j] + kx) < (unsigned)a.w;
      if (__any(any)) mymask |= 1ull << t;
    }
    if (lane == 0) s_mask[wave] = mymask;
    __syncthreads();
    tapmask = s_mask[0] | s_mask[1] | s_mask[2] | s_mask[3];
    // readfirstlane returns a signed int: go through unsigned or bit 31 smears over bits 32..63
    const unsigned tm_lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)tapmask);
    const unsigned tm_hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(tapmask >> 32));
    tapmask = ((unsigned long long)tm_hi << 32) | (unsigned long long)tm_lo;
  }
  // k-steps of this tile in iteration order (chunk-major, valid taps inner).  Split-K: blockIdx.y owns a contiguous
  // range of the DENSE (chunk, tap) sequence -- whole chunks whenever the slice count divides the chunks per tap; bf16
  // fc6 (4 chunks per tap, 8 slices) is cut inside its chunks -- and runs the valid taps inside it.  Cutting the dense
  // sequence (not the tile's own list of valid steps) keeps the partition of every output's sum independent of which
  // taps its tile may skip, i.e. of the batch the face sits in.
  const int ntv = __builtin_popcountll(tapmask);
  int nit = ntv * a.cpt, chunk0 = 0;
  unsigned long long rem0 = tapmask;
  if (a.ksplit > 1) {
    const int dense = ntaps * a.cpt;
    const int d0 = (int)blockIdx.y * dense / a.ksplit, d1 = ((int)blockIdx.y + 1) * dense / a.ksplit;
    const int c0 = d0 / ntaps, t0 = d0 - c0 * ntaps, c1 = d1 / ntaps, t1 = d1 - c1 * ntaps;
    const unsigned long long from0 = tapmask & (~0ull << t0);       // taps >= t0
    const unsigned long long upto1 = tapmask & ((1ull << t1) - 1);  // taps < t1
    if (c0 == c1) {
      rem0 = from0 & upto1;
      nit = __builtin_popcountll(rem0);
    } else {
      rem0 = from0;
      nit = __builtin_popcountll(from0) + (c1 - c0 - 1) * ntv + __builtin_popcountll(upto1);
    }
    chunk0 = c0;
    if (rem0 == 0ull) {  // no valid tap left in the first chunk of the range
      rem0 = tapmask;
      ++chunk0;
    }
  }

  // per-row BYTE offset of the centre pixel; per-tap displacement is wave-uniform
  unsigned rowoff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
    rowoff[j] = ((unsigned)(((pn[j] * a.h + py[j]) * a.w + px[j]) * a.cin) + EPC * c8) * ES;
  const char* wrow[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
    wrow[j] = reinterpret_cast<const char*>(a.wt) + ((size_t)(n0 + r0 + 32 * j) * a.K + EPC * c8) * ES;
  const char* xbase = reinterpret_cast<const char*>(a.x);

  // fragment read offsets (floats): rows 64*wr + 32*i + lr of A, 64*wc + 32*j + lr of B; the XOR term of
  // the swizzle depends on lr only, the chunk is 2*t + lh
  int fa0, fa1, fb0, fb1, fc0, fc1, fc2, fc3;
  {
    const int wr = wave >> 1, wc = wave & 1;
    const int lr = lane & 31, lh = lane >> 5;
    const int swx = (lr >> 1) & 7;
    fa0 = (64 * wr + lr) * BK, fa1 = fa0 + 32 * BK;
    fb0 = (64 * wc + lr) * BK, fb1 = fb0 + 32 * BK;
    fc0 = ((0 + lh) ^ swx) << 2, fc1 = ((2 + lh) ^ swx) << 2, fc2 = ((4 + lh) ^ swx) << 2, fc3 = ((6 + lh) ^ swx) << 2;
  }

  f32x16 acc00, acc01, acc10, acc11;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc00[r] = acc01[r] = acc10[r] = acc11[r] = 0.f;

  // ---- fp32 parity form: LDS-DMA operand ring + three-level accumulation -------------------------------------------
  // A v_mfma_f32_32x32x2_f32 stream is bit for bit ONE fmaf chain per output: K = 576 .. 12,544 sequential roundings,
  // 4-7e-7 relative RMS per layer against the float64 oracle where a blocked CPU summation leaves 2-5e-7 -- enough to
  // push 3 of 4,352 top-4 landmarks past the 1e-4 px bar.  Here every k-step (32 products: one
  // filter tap of one 32-channel chunk) sums in the MFMA accumulators from ZERO; the step sums are added, in the
  // order of the dense (chunk, tap) sequence, into a second accumulator set, and that set is added into a third and
  // cleared whenever the dense step index enters a new GROUP of grp ~ sqrt(K/32) steps: chains of
  // 32 + grp + K/(32 grp) roundings instead of K (fc6: 32 + 19 + 21 instead of 12,544).  Groups are cut on the DENSE
  // index and a step whose tap the tile skips contributes exact zeros either way, so an output's summation tree does
  // not depend on which padding taps its tile skips, i.e. on the batch the face sits in.
  // The 128 registers of the two extra sets are the staging registers of the other form and its slack: operands reach
  // LDS by buffer_load_dwordx4 ... lds (flm_igemm_args.h), zero padding by the buffer bounds check.  Tile t+2 is
  // requested in group 3 of step t, into the stage that step's barrier has just retired, and waited for (vmcnt(0)) at
  // the barrier of step t+1: three quarters of a step (3,072 MFMA cycles) of cover.
  // The third level is 128 VALU instructions behind a wave-uniform branch right after the barrier of a group's first
  // step (no LDS read is in flight there: at the end of the step the branch cost an lgkmcnt(0) per step, 3 % of the
  // layer).  The second level costs no matrix time: the first slot of a step adds each tile's accumulators to the second
  // set (16 VALU adds in the shadow of the neighbouring tile's MFMA; that tile's last MFMA was issued four MFMAs
  // earlier) and restarts the tile with C = 0.
  if constexpr (TWO) {
    f32x16 sum00, sum01, sum10, sum11;   // second level: step sums of the current group
    f32x16 top00, top01, top10, top11;   // third level: group sums
#pragma unroll
    for (int r = 0; r < 16; ++r) sum00[r] = sum01[r] = sum10[r] = sum11[r] = top00[r] = top01[r] = top10[r] = top11[r] = 0.f;
    f32x16 zero16;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero16[r] = 0.f;

    typedef __attribute__((address_space(3))) char lds_char;
    const int wave_s = __builtin_amdgcn_readfirstlane(wave);
    const unsigned dma_a = (unsigned)(size_t)((lds_char*)smem_raw) + wave_s * 8 * 128;  // this wave's 8 rows of A stage 0
    const unsigned dma_b = dma_a + 2 * TILE_F * 4;
    const int xbias = (a.pad * a.w + a.pad) * a.cin * 4;  // largest negative tap displacement, in bytes
    const dma_srd xdma = dma_make_srd(reinterpret_cast<const char*>(a.x) - xbias);
    const dma_srd wdma = dma_make_srd(reinterpret_cast<const char*>(a.wt) + (size_t)n0 * a.K * 4);
    const unsigned wrow0 = ((unsigned)r0 * (unsigned)a.K + 4u * c8) * 4u;
    const int wstep = 32 * a.K * 4;  // bytes between the weight rows of consecutive j
    // (py << 16) | px of the filter centre; rows past M get py = 0x7000: out of bounds for every tap, they never load
    int pyx[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) pyx[j] = ((pv[j] ? py[j] : 0x7000) << 16) | px[j];
    const int off_a = (fa0 + fc0) * 4, off_b = (fb0 + fc0) * 4;  // byte offsets of k-group 0 in an A / B stage

    unsigned long long rem = rem0;
    int cur_chunk = chunk0 < a.cpt ? chunk0 : 0;
    int cur_tap = __builtin_ctzll(rem);
    int ld_ky = 0, ld_kx = 0, ld_delta = 0, ld_koff = 0, ld_gid = 0;
    // group (of grp consecutive dense steps) of the step before the current one, the current one and the next
    int g_prev = 0, g_cur = 0, g_nx1 = 0;

#define FLM_TILE_PARAMS()                                                    \
  {                                                                          \
    const int ty = (cur_tap * a.kw_magic) >> 16;                             \
    ld_ky = ty - a.pad;                                                      \
    ld_kx = cur_tap - ty * a.kw - a.pad;                                     \
    ld_delta = (ld_ky * a.w + ld_kx) * a.cin * 4 + xbias + cur_chunk * 128;  \
    ld_koff = cur_tap * a.cin * 4 + cur_chunk * 128;                         \
    ld_gid = ((cur_chunk * ntaps + cur_tap) * a.grp_magic) >> 16;            \
    rem &= rem - 1;                                                          \
    if (rem == 0) {                                                          \
      rem = tapmask;                                                         \
      if (++cur_chunk == a.cpt) cur_chunk = 0; /* (past the slice's end: requests nobody reads) */ \
    }                                                                        \
    cur_tap = __builtin_ctzll(rem);                                          \
  }
#define FLM_DMA_A(J, STG)                                                                                    \
  {                                                                                                          \
    const int iy = (pyx[J] >> 16) + ld_ky, ix = (pyx[J] & 0xffff) + ld_kx;                                   \
    const bool ok_ = (unsigned)iy < (unsigned)a.h && (unsigned)ix < (unsigned)a.w;                          \
    dma_load16(xdma, dma_a + (STG) * (TILE_F * 4) + (J) * 32 * 128, ok_ ? rowoff[J] : kOobOffset, ld_delta); \
  }
#define FLM_DMA_B(J, STG) dma_load16(wdma, dma_b + (STG) * (TILE_F * 4) + (J) * 32 * 128, wrow0, (J) * wstep + ld_koff);
#define FLM_READ_FRAGS(AF0, AF1, BF0, BF1, T, STG)                                                          \
  {                                                                                                         \
    int oa_ = off_a, ob_ = off_b;                                                                           \
    if (FLM_IGEMM_VAR & 2) asm volatile("" : "+v"(oa_), "+v"(ob_)); /* (opaque, or the sums are hoisted) */ \
    const int fct_ = (T) == 0 ? fc0 : ((T) == 1 ? fc1 : ((T) == 2 ? fc2 : fc3));                            \
    const char* pa_ = !(FLM_IGEMM_VAR & 2) ? smem_raw + (fa0 + fct_) * 4 + (STG) * (TILE_F * 4)             \
                                           : smem_raw + ((oa_ ^ ((T) << 5)) + (STG) * (TILE_F * 4));        \
    const char* pb_ = !(FLM_IGEMM_VAR & 2) ? smem_raw + (fb0 + fct_) * 4 + (STG) * (TILE_F * 4) + 2 * TILE_F * 4 \
                                           : smem_raw + ((ob_ ^ ((T) << 5)) + (STG) * (TILE_F * 4) + 2 * TILE_F * 4); \
    AF0 = *reinterpret_cast<const float4*>(pa_);                                                            \
    AF1 = *reinterpret_cast<const float4*>(pa_ + 32 * BK * 4);                                              \
    BF0 = *reinterpret_cast<const float4*>(pb_);                                                            \
    BF1 = *reinterpret_cast<const float4*>(pb_ + 32 * BK * 4);                                              \
  }
#define FLM_SLOT_x 0
#define FLM_SLOT_y 1
#define FLM_SLOT_z 2
#define FLM_SLOT_w 3
#define FLM_MFMA4(AF0, AF1, BF0, BF1, E)                                                     \
  mfma_slot<false, FLM_SLOT_##E>(AF0, AF1, BF0, BF1, acc00, acc01, acc10, acc11);            \
  __builtin_amdgcn_sched_barrier(0);
  // first slot of a step: close the previous step tile by tile and restart the tile from C = 0
#define FLM_FLUSH1(SUM, ACC, AV, BV)                                                         \
  SUM += ACC;                                                                                \
  ACC = __builtin_amdgcn_mfma_f32_32x32x2f32(AV, BV, zero16, 0, 0, 0);                       \
  __builtin_amdgcn_sched_barrier(0);
#define FLM_MFMA4_FLUSH(AF0, AF1, BF0, BF1)                                                  \
  FLM_FLUSH1(sum00, acc00, AF0.x, BF0.x)                                                     \
  FLM_FLUSH1(sum01, acc01, AF0.x, BF1.x)                                                     \
  FLM_FLUSH1(sum10, acc10, AF1.x, BF0.x)                                                     \
  FLM_FLUSH1(sum11, acc11, AF1.x, BF1.x)

  // One step on stage BUF (compile-time).
#define FLM_STEP2(BUF)                                                                                \
  {                                                                                                   \
    FLM_TILE_PARAMS()                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                                \
    /* group 0 (fragments fetched during the previous step's group 3) */                              \
    FLM_MFMA4_FLUSH(afx0, afx1, bfx0, bfx1)                                                           \
    FLM_READ_FRAGS(afy0, afy1, bfy0, bfy1, 1, BUF)                                                    \
    FLM_MFMA4(afx0, afx1, bfx0, bfx1, y)                                                              \
    FLM_MFMA4(afx0, afx1, bfx0, bfx1, z)                                                              \
    FLM_MFMA4(afx0, afx1, bfx0, bfx1, w)                                                              \
    /* group 1 */                                                                                     \
    FLM_MFMA4(afy0, afy1, bfy0, bfy1, x) FLM_READ_FRAGS(afx0, afx1, bfx0, bfx1, 2, BUF)               \
    FLM_MFMA4(afy0, afy1, bfy0, bfy1, y)                                                              \
    FLM_MFMA4(afy0, afy1, bfy0, bfy1, z)                                                              \
    FLM_MFMA4(afy0, afy1, bfy0, bfy1, w)                                                              \
    /* group 2; then the step's only barrier: every wave has fetched the fragments of group 3 by now, so this */ \
    /* stage is dead, and this wave's pieces of the next one have landed (vmcnt(0)) */                \
    FLM_MFMA4(afx0, afx1, bfx0, bfx1, x) FLM_READ_FRAGS(afy0, afy1, bfy0, bfy1, 3, BUF)               \
    FLM_MFMA4(afx0, afx1, bfx0, bfx1, y)                                                              \
    FLM_MFMA4(afx0, afx1, bfx0, bfx1, z)                                                              \
    FLM_MFMA4(afx0, afx1, bfx0, bfx1, w)                                                              \
    if (!(FLM_IGEMM_VAR & 4)) __builtin_amdgcn_s_waitcnt(0x0f70);                                     \
    if (!(FLM_IGEMM_VAR & 16)) __syncthreads();                                                       \
    /* third level, right behind the barrier: the second set holds the steps before this one; if this step opened */ \
    /* a new group they are a closed group */                                                         \
    if (!(FLM_IGEMM_VAR & 1) && g_prev != g_cur) {                                                    \
      top00 += sum00; top01 += sum01; top10 += sum10; top11 += sum11;                                 \
      sum00 = zero16; sum01 = zero16; sum10 = zero16; sum11 = zero16;                                 \
    }                                                                                                 \
    /* group 3: the next step's first fragments come from the next stage; tile it+2 is requested into this one */ \
    {                                                                                                 \
      FLM_MFMA4(afy0, afy1, bfy0, bfy1, x)                                                            \
      FLM_READ_FRAGS(afx0, afx1, bfx0, bfx1, 0, (BUF) ^ 1)                                            \
      if (!(FLM_IGEMM_VAR & 8)) { FLM_DMA_A(0, BUF) FLM_DMA_B(0, BUF) }                               \
      FLM_MFMA4(afy0, afy1, bfy0, bfy1, y) if (!(FLM_IGEMM_VAR & 8)) { FLM_DMA_A(1, BUF) FLM_DMA_B(1, BUF) } \
      FLM_MFMA4(afy0, afy1, bfy0, bfy1, z) if (!(FLM_IGEMM_VAR & 8)) { FLM_DMA_A(2, BUF) FLM_DMA_B(2, BUF) } \
      FLM_MFMA4(afy0, afy1, bfy0, bfy1, w) if (!(FLM_IGEMM_VAR & 8)) { FLM_DMA_A(3, BUF) FLM_DMA_B(3, BUF) } \
    }                                                                                                 \
    g_prev = g_cur;                                                                                   \
    g_cur = g_nx1;                                                                                    \
    g_nx1 = ld_gid;                                                                                   \
  }

    // prologue: tile 0 -> stage 0, tile 1 -> stage 1 (past the end: a tile nobody reads)
    if (nit > 0) {
      FLM_TILE_PARAMS()
      g_prev = g_cur = ld_gid;
      FLM_DMA_A(0, 0) FLM_DMA_A(1, 0) FLM_DMA_A(2, 0) FLM_DMA_A(3, 0)
      FLM_DMA_B(0, 0) FLM_DMA_B(1, 0) FLM_DMA_B(2, 0) FLM_DMA_B(3, 0)
      FLM_TILE_PARAMS()
      g_nx1 = ld_gid;
      FLM_DMA_A(0, 1) FLM_DMA_A(1, 1) FLM_DMA_A(2, 1) FLM_DMA_A(3, 1)
      FLM_DMA_B(0, 1) FLM_DMA_B(1, 1) FLM_DMA_B(2, 1) FLM_DMA_B(3, 1)
    }
    __builtin_amdgcn_s_waitcnt(0x0f70);
    __syncthreads();
    float4 afx0, afx1, bfx0, bfx1, afy0, afy1, bfy0, bfy1;
    afy0 = afy1 = bfy0 = bfy1 = make_float4(0.f, 0.f, 0.f, 0.f);
    FLM_READ_FRAGS(afx0, afx1, bfx0, bfx1, 0, 0)
    for (int it = 0; it < nit; it += 2) {
      FLM_STEP2(0)
      if (it + 1 < nit) FLM_STEP2(1)
    }
    __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): the requests past the last tile still write LDS
    acc00 = top00 + (sum00 + acc00);
    acc01 = top01 + (sum01 + acc01);
    acc10 = top10 + (sum10 + acc10);
    acc11 = top11 + (sum11 + acc11);
#undef FLM_TILE_PARAMS
#undef FLM_DMA_A
#undef FLM_DMA_B
#undef FLM_READ_FRAGS
#undef FLM_MFMA4
#undef FLM_FLUSH1
#undef FLM_MFMA4_FLUSH
#undef FLM_SLOT_x
#undef FLM_SLOT_y
#undef FLM_SLOT_z
#undef FLM_SLOT_w
#undef FLM_STEP2
  }

  if constexpr (!TWO) {
  // ---- software pipeline ---------------------------------------------------------------------------
  // Step t computes tile t from LDS[t&1]; in the SAME step, spread between the 64 MFMAs, it issues the
  // global loads of tile t+2 into one register set and writes tile t+1 (loaded during step t-1, so long
  // landed) from the other set into LDS[(t+1)&1].  Every non-matrix instruction sits in the shadow of an
  // MFMA (one 64-cycle MFMA holds the SIMD's issue port for a few cycles only), the wave never waits on
  // memory it asked for less than a full step ago.  The step's one barrier sits after fragment group 2, when the
  // writes of tile t+1 are done and every wave has fetched its last fragments of tile t; group 3 then reads the
  // next step's first fragments from the other stage, so the step boundary itself has neither a barrier nor an
  // exposed LDS read (measured neutral next to a barrier at the step end: two co-resident workgroups already
  // cover each other).  Register sets are NAMED scalars (arrays under `if` go to scratch).
  float4 ra0P, ra1P, ra2P, ra3P, rb0P, rb1P, rb2P, rb3P;
  float4 ra0Q, ra1Q, ra2Q, ra3Q, rb0Q, rb1Q, rb2Q, rb3Q;
  bool ok0P = false, ok1P = false, ok2P = false, ok3P = false;
  bool ok0Q = false, ok1Q = false, ok2Q = false, ok3Q = false;
  ra0P = ra1P = ra2P = ra3P = rb0P = rb1P = rb2P = rb3P = make_float4(0.f, 0.f, 0.f, 0.f);
  ra0Q = ra1Q = ra2Q = ra3Q = rb0Q = rb1Q = rb2Q = rb3Q = make_float4(0.f, 0.f, 0.f, 0.f);

  // iterator over (valid tap, channel chunk): state of the NEXT tile to load
  unsigned long long rem = rem0;
  int cur_chunk = chunk0 < a.cpt ? chunk0 : 0;
  int cur_tap = __builtin_ctzll(rem);
  int ld_ky = 0, ld_kx = 0, ld_delta = 0, ld_koff = 0, ld_c0 = 0;

#define FLM_TILE_PARAMS()                                                    \
  {                                                                          \
    const int ty = (cur_tap * a.kw_magic) >> 16;                             \
    ld_ky = ty - a.pad;                                                      \
    ld_kx = cur_tap - ty * a.kw - a.pad;                                     \
    ld_delta = (ld_ky * a.w + ld_kx) * a.cin * ES;                           \
    ld_c0 = cur_chunk * 128;                                                 \
    ld_koff = cur_tap * a.cin * ES + ld_c0;                                  \
    /* taps are the INNER loop: consecutive k-steps re-read the same pixels shifted by one tap, so the */ \
    /* gathered rows are still in L1/L2 (chunk-inner order re-fetched them from beyond L2 nine times)    */ \
    rem &= rem - 1;                                                          \
    if (rem == 0) {                                                          \
      rem = tapmask;                                                         \
      if (++cur_chunk == a.cpt) cur_chunk = 0; /* (past the slice's end: loads nobody uses) */ \
    }                                                                        \
    cur_tap = __builtin_ctzll(rem);                                          \
  }
  // Out-of-bounds rows load a valid address (their own centre pixel) and are zeroed at the LDS write,
  // so loads issue with no branch around them.
#define FLM_LOAD_A(J, RA, OK)                                                                  \
  {                                                                                            \
    const int iy = py[J] + ld_ky, ix = px[J] + ld_kx;                                          \
    OK = pv[J] && (unsigned)iy < (unsigned)a.h && (unsigned)ix < (unsigned)a.w;               \
    const unsigned off = rowoff[J] + (OK ? (unsigned)ld_delta : 0u) + (unsigned)ld_c0;         \
    RA = *reinterpret_cast<const float4*>(xbase + off);                                        \
  }
#define FLM_LOAD_B(J, RB) RB = *reinterpret_cast<const float4*>(wrow[J] + ld_koff);
#define FLM_STORE_A(J, RA, OK) \
  *reinterpret_cast<float4*>(As + sbuf * TILE_F + swz(r0 + 32 * J, c8)) = OK ? RA : make_float4(0.f, 0.f, 0.f, 0.f);
#define FLM_STORE_B(J, RB) *reinterpret_cast<float4*>(Bs + sbuf * TILE_F + swz(r0 + 32 * J, c8)) = RB;


#define FLM_READ_FRAGS(AF0, AF1, BF0, BF1, FC)                                   \
  AF0 = *reinterpret_cast<const float4*>(Ab + fa0 + FC);                         \
  AF1 = *reinterpret_cast<const float4*>(Ab + fa1 + FC);                         \
  BF0 = *reinterpret_cast<const float4*>(Bb + fb0 + FC);                         \
  BF1 = *reinterpret_cast<const float4*>(Bb + fb1 + FC);
#define FLM_SLOT_x 0
#define FLM_SLOT_y 1
#define FLM_SLOT_z 2
#define FLM_SLOT_w 3
#define FLM_MFMA4(AF0, AF1, BF0, BF1, E)                                                     \
  mfma_slot<BF, FLM_SLOT_##E>(AF0, AF1, BF0, BF1, acc00, acc01, acc10, acc11);               \
  __builtin_amdgcn_sched_barrier(0);

  // One step.  W* = register set written to LDS now (tile it+1), L* = set receiving the loads of tile it+2.
  // No branches inside a step (hipcc drops to `s_waitcnt vmcnt(0)` in front of every load and LDS write
  // that sits in its own basic block): past the last tile the loads re-read tile (tap 0, chunk 0) and the
  // writes refill a buffer nobody reads any more.
#define FLM_STEP(IT, W, L)                                                                            \
  {                                                                                                   \
    const int it_ = (IT);                                                                             \
    const int buf = it_ & 1, sbuf = buf ^ 1;                                                          \
    const float* Ab = As + buf * TILE_F;                                                              \
    const float* Bb = Bs + buf * TILE_F;                                                              \
    FLM_TILE_PARAMS()                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                                \
    /* group 0 (fragments fetched during the previous step's group 3): global loads of tile it+2 */   \
    FLM_MFMA4(afx0, afx1, bfx0, bfx1, x) FLM_READ_FRAGS(afy0, afy1, bfy0, bfy1, fc1)                  \
    FLM_LOAD_A(0, ra0##L, ok0##L) FLM_LOAD_B(0, rb0##L)                                               \
    FLM_MFMA4(afx0, afx1, bfx0, bfx1, y) FLM_LOAD_A(1, ra1##L, ok1##L) FLM_LOAD_B(1, rb1##L)          \
    FLM_MFMA4(afx0, afx1, bfx0, bfx1, z) FLM_LOAD_A(2, ra2##L, ok2##L) FLM_LOAD_B(2, rb2##L)          \
    FLM_MFMA4(afx0, afx1, bfx0, bfx1, w) FLM_LOAD_A(3, ra3##L, ok3##L) FLM_LOAD_B(3, rb3##L)          \
    /* group 1: LDS writes of tile it+1, A rows */                                                    \
    FLM_MFMA4(afy0, afy1, bfy0, bfy1, x) FLM_READ_FRAGS(afx0, afx1, bfx0, bfx1, fc2)                  \
    FLM_STORE_A(0, ra0##W, ok0##W)                                                                    \
    FLM_MFMA4(afy0, afy1, bfy0, bfy1, y) FLM_STORE_A(1, ra1##W, ok1##W)                               \
    FLM_MFMA4(afy0, afy1, bfy0, bfy1, z) FLM_STORE_A(2, ra2##W, ok2##W)                               \
    FLM_MFMA4(afy0, afy1, bfy0, bfy1, w) FLM_STORE_A(3, ra3##W, ok3##W)                               \
    /* group 2: LDS writes, B rows; then the step's only barrier: every wave has fetched the fragments of */ \
    /* group 3 by now, so this stage is dead and the next one complete */                             \
    FLM_MFMA4(afx0, afx1, bfx0, bfx1, x) FLM_READ_FRAGS(afy0, afy1, bfy0, bfy1, fc3)                  \
    FLM_STORE_B(0, rb0##W)                                                                            \
    FLM_MFMA4(afx0, afx1, bfx0, bfx1, y) FLM_STORE_B(1, rb1##W)                                       \
    FLM_MFMA4(afx0, afx1, bfx0, bfx1, z) FLM_STORE_B(2, rb2##W)                                       \
    FLM_MFMA4(afx0, afx1, bfx0, bfx1, w) FLM_STORE_B(3, rb3##W)                                       \
    __syncthreads();                                                                                  \
    /* group 3: the next step's first fragments come from the next stage under these MFMAs: the step */ \
    /* boundary has no barrier and no exposed LDS latency */                                          \
    {                                                                                                 \
      const float* Abn = As + sbuf * TILE_F;                                                          \
      const float* Bbn = Bs + sbuf * TILE_F;                                                          \
      FLM_MFMA4(afy0, afy1, bfy0, bfy1, x)                                                            \
      afx0 = *reinterpret_cast<const float4*>(Abn + fa0 + fc0);                                       \
      afx1 = *reinterpret_cast<const float4*>(Abn + fa1 + fc0);                                       \
      bfx0 = *reinterpret_cast<const float4*>(Bbn + fb0 + fc0);                                       \
      bfx1 = *reinterpret_cast<const float4*>(Bbn + fb1 + fc0);                                       \
      FLM_MFMA4(afy0, afy1, bfy0, bfy1, y)                                                            \
      FLM_MFMA4(afy0, afy1, bfy0, bfy1, z)                                                            \
      FLM_MFMA4(afy0, afy1, bfy0, bfy1, w)                                                            \
    }                                                                                                 \
  }

  // prologue: tile 0 -> set P -> LDS[0]; tile 1 -> set Q (written during step 0)
  if (nit > 0) {
    FLM_TILE_PARAMS()
    FLM_LOAD_A(0, ra0P, ok0P) FLM_LOAD_A(1, ra1P, ok1P) FLM_LOAD_A(2, ra2P, ok2P) FLM_LOAD_A(3, ra3P, ok3P)
    FLM_LOAD_B(0, rb0P) FLM_LOAD_B(1, rb1P) FLM_LOAD_B(2, rb2P) FLM_LOAD_B(3, rb3P)
    if (nit > 1) {
      FLM_TILE_PARAMS()
      FLM_LOAD_A(0, ra0Q, ok0Q) FLM_LOAD_A(1, ra1Q, ok1Q) FLM_LOAD_A(2, ra2Q, ok2Q) FLM_LOAD_A(3, ra3Q, ok3Q)
      FLM_LOAD_B(0, rb0Q) FLM_LOAD_B(1, rb1Q) FLM_LOAD_B(2, rb2Q) FLM_LOAD_B(3, rb3Q)
    }
    {
      const int sbuf = 0;
      FLM_STORE_A(0, ra0P, ok0P) FLM_STORE_A(1, ra1P, ok1P) FLM_STORE_A(2, ra2P, ok2P) FLM_STORE_A(3, ra3P, ok3P)
      FLM_STORE_B(0, rb0P) FLM_STORE_B(1, rb1P) FLM_STORE_B(2, rb2P) FLM_STORE_B(3, rb3P)
    }
  }
  __syncthreads();
  float4 afx0, afx1, bfx0, bfx1, afy0, afy1, bfy0, bfy1;
  afy0 = afy1 = bfy0 = bfy1 = make_float4(0.f, 0.f, 0.f, 0.f);
  {
    const float* Ab = As;
    const float* Bb = Bs;
    FLM_READ_FRAGS(afx0, afx1, bfx0, bfx1, fc0)
  }

  // step it writes set Q (tile it+1) and loads tile it+2 into set P; the next step swaps the roles
  for (int it = 0; it < nit; it += 2) {
    FLM_STEP(it, Q, P)
    if (it + 1 < nit) FLM_STEP(it + 1, P, Q)
  }

#undef FLM_TILE_PARAMS
#undef FLM_LOAD_A
#undef FLM_LOAD_B
#undef FLM_STORE_A
#undef FLM_STORE_B
#undef FLM_READ_FRAGS
#undef FLM_MFMA4
#undef FLM_SLOT_x
#undef FLM_SLOT_y
#undef FLM_SLOT_z
#undef FLM_SLOT_w
#undef FLM_STEP

  }  // (!TWO)

  // ---- epilogue: y = acc*scale + shift, ReLU, 2x2 max-pool (MMAP 1), store ---------------------
  // accumulator layout: column = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  f32x16 acc[2][2];
  acc[0][0] = acc00; acc[0][1] = acc01; acc[1][0] = acc10; acc[1][1] = acc11;
  // (TWO: lane coordinates afresh -- lane id from v_mbcnt, wave id from a scalar register -- so that nothing derived from
  // threadIdx.x stays alive across the k-loop, which uses nearly all of the 256 registers)
  const int lane_e = TWO ? (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) : lane;
  const int wave_e = TWO ? __builtin_amdgcn_readfirstlane(wave) : wave;
  const int wr = wave_e >> 1, wc = wave_e & 1, lr = lane_e & 31, lh = lane_e >> 5;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = n0 + 64 * wc + 32 * j + lr;
    const bool cok = col < a.cout;
    const float sc = a.scale[col], sh = a.shift[col];  // coutpad-long arrays: always in bounds
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int mbase = m0 + 64 * wr + 32 * i;
      if (MMAP == 1 && a.ksplit > 1) {  // raw partial sums in quad order; BN / ReLU / pool happen in the reduce
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mbase + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (cok && m < a.M) a.part[((size_t)blockIdx.y * a.M + m) * a.ldc + col] = acc[i][j][r];
        }
      } else if (MMAP == 1) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float v = -3.402823466e38f;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float u = fmaf(acc[i][j][4 * g + e], sc, sh);
            if (RELU) u = fminf(fmaxf(u, 0.f), a.relu_max);
            v = fmaxf(v, u);
          }
          const int m = mbase + 8 * g + 4 * lh;  // first row of the quad
          if (cok && m < a.M) {
            const size_t o = (size_t)(m >> 2) * a.ldc + col;
            if (BF && !a.out_f32) reinterpret_cast<unsigned short*>(a.y)[o] = f2bf(v);
            else reinterpret_cast<float*>(a.y)[o] = v;
          }
        }
      } else {
        const int m_first = mbase + 4 * lh;
        const int nn0 = MMAP == 2 ? m_first % a.n : 0;
        const int pos0 = MMAP == 2 ? posmajor_pos(a, m_first / a.n) : 0, pos1 = MMAP == 2 ? posmajor_pos(a, m_first / a.n + 1) : 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mbase + (r & 3) + 8 * (r >> 2) + 4 * lh;
          float u = fmaf(acc[i][j][r], sc, sh);
          if (a.ksplit > 1) {  // raw partial sums, stored in output row order
            if (cok && m < a.M) {
              const size_t prow = (MMAP == 2) ? posmajor_orow(m, m_first, nn0, pos0, pos1, a.n, a.h * a.w) : (size_t)m;
              a.part[((size_t)blockIdx.y * a.M + prow) * a.ldc + col] = acc[i][j][r];
            }
            continue;
          }
          if (cok && m < a.M) {
            const size_t orow = MMAP == 2 ? posmajor_orow(m, m_first, nn0, pos0, pos1, a.n, a.h * a.w) : (size_t)m;
            const size_t o = orow * a.ldc + col;
            if (MMAP == 0 && a.res) {  // residual add before the activation (ResNet bottlenecks)
              if (BF) u += (float)__builtin_bit_cast(__bf16, reinterpret_cast<const unsigned short*>(a.res)[o]);
              else u += reinterpret_cast<const float*>(a.res)[o];
            }
            if (RELU) u = fminf(fmaxf(u, 0.f), a.relu_max);
            if (BF && !a.out_f32) reinterpret_cast<unsigned short*>(a.y)[o] = f2bf(u);
            else reinterpret_cast<float*>(a.y)[o] = u;
          }
        }
      }
    }
  }
  }  // pass
}

static std::atomic<int> g_posperm{1};  // A/B knob "posmajor_order": same results either way
void igemm_posperm_enable(int on) { g_posperm.store(on, std::memory_order_relaxed); }
int igemm_posperm_enabled() { return g_posperm.load(std::memory_order_relaxed); }
static std::mutex g_posperm_mu;
static std::vector<PospermEntry> g_posperm_cache;
static bool posperm_same(const PospermEntry& x, const PospermEntry& y) {
  return x.h == y.h && x.w == y.w && x.kh == y.kh && x.kw == y.kw && x.pad == y.pad && x.g == y.g;
}
bool posperm_cache_get(PospermEntry& e) {
  std::lock_guard<std::mutex> lock(g_posperm_mu);
  for (const PospermEntry& c : g_posperm_cache)
    if (posperm_same(c, e)) { e = c; return true; }
  return false;
}
void posperm_cache_put(const PospermEntry& e) {
  std::lock_guard<std::mutex> lock(g_posperm_mu);
  for (const PospermEntry& c : g_posperm_cache)
    if (posperm_same(c, e)) return;
  if (g_posperm_cache.size() < 64) g_posperm_cache.push_back(e);
}

template <bool BF, int MMAP, bool RELU, bool TWO = false>
static int launch_t(hipStream_t s, const IgemmArgs& a_in) {
  IgemmArgs a = a_in;
  if (MMAP == 2) posmajor_fill_perm(a, BM);
  const size_t lds = sizeof(float) * 4 * TILE_F + 64;
  static FuncAttrOnce attr;
  FLM_FUNC_ATTR_ONCE(attr, (&igemm_kernel<BF, MMAP, RELU, TWO>), lds);
  const int mslots = (MMAP == 2 && !TWO) ? (a.mtiles + 1) / 2 : a.mtiles;
  igemm_kernel<BF, MMAP, RELU, TWO><<<dim3(mslots * a.ntiles, a.ksplit > 1 ? a.ksplit : 1), 256, lds, s>>>(a);
  FLM_LAUNCH_CHECK("igemm_kernel");
  return FLM_OK;
}

// y[m][c] = act(scale[c] * sum_s part[s][m][c] + shift[c]); fixed summation order (deterministic)
__global__ void splitk_reduce_kernel(const float* __restrict__ part, const float* __restrict__ scale,
                                     const float* __restrict__ shift, void* __restrict__ y, int M, int ldc, int cout,
                                     int ksplit, int relu, int out_bf16) {
  const size_t total = (size_t)M * ldc;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % ldc);
    if (c >= cout) continue;
    float v = 0.f;
    for (int s = 0; s < ksplit; ++s) v += part[(size_t)s * total + i];
    v = fmaf(v, scale[c], shift[c]);
    if (relu) v = fmaxf(v, 0.f);
    if (out_bf16) reinterpret_cast<unsigned short*>(y)[i] = f2bf(v);
    else reinterpret_cast<float*>(y)[i] = v;
  }
}

// Pooled layers: y[q][c] = max over the quad's four rows of act(scale[c] * sum_s part[s][4q+e][c] + shift[c]).
__global__ void splitk_reduce_pool_kernel(const float* __restrict__ part, const float* __restrict__ scale,
                                          const float* __restrict__ shift, void* __restrict__ y, int M, int ldc, int cout,
                                          int ksplit, int relu, float relu_max, int out_bf16) {
  const size_t total = (size_t)(M >> 2) * ldc, plane = (size_t)M * ldc;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % ldc);
    if (c >= cout) continue;
    const size_t q = i / ldc;
    float best = -3.402823466e38f;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float v = 0.f;
      for (int s = 0; s < ksplit; ++s) v += part[(size_t)s * plane + (4 * q + e) * ldc + c];
      v = fmaf(v, scale[c], shift[c]);
      if (relu) v = fminf(fmaxf(v, 0.f), relu_max);
      best = fmaxf(best, v);
    }
    if (out_bf16) reinterpret_cast<unsigned short*>(y)[i] = f2bf(best);
    else reinterpret_cast<float*>(y)[i] = best;
  }
}

int igemm_occupancy(size_t lds_bytes) {
  int nb = -1;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(&igemm_kernel<false, 0, true>), 256,
                                                   lds_bytes) != hipSuccess)
    return -1;
  return nb;
}

template <bool BF, bool TWO = false>
static int dispatch(hipStream_t s, const IgemmDesc& d, const IgemmArgs& a) {
  if (d.pool) return d.relu ? launch_t<BF, 1, true, TWO>(s, a) : launch_t<BF, 1, false, TWO>(s, a);
  if (d.posmajor) return d.relu ? launch_t<BF, 2, true, TWO>(s, a) : launch_t<BF, 2, false, TWO>(s, a);
  return d.relu ? launch_t<BF, 0, true, TWO>(s, a) : launch_t<BF, 0, false, TWO>(s, a);
}

// fp32 summation: 0 = multi-level (igemm_kernel<.., TWO>, the default), -1 = the single-chain kernel with register
// staging (A/B of accuracy and speed, tools/).  Unlike the other knobs this one changes the fp32 summation order, i.e.
// result bits (never their validity).
static std::atomic<int> g_f32_group{0};
void igemm_f32_group(int steps) { g_f32_group = steps; }

int launch_igemm(hipStream_t s, const IgemmDesc& d) {
  const int es = d.bf16 ? 2 : 4;
  const int bke = 128 / es;  // operand elements per k-step
  if (d.cin % bke != 0 || d.coutpad % BN != 0 || d.cout > d.coutpad || d.kh * d.kw > 64) {
    set_error("igemm: unsupported shape cin=%d coutpad=%d cout=%d k=%dx%d (%s)", d.cin, d.coutpad, d.cout, d.kh, d.kw,
              d.bf16 ? "bf16" : "fp32");
    return FLM_ERR_SHAPE;
  }
  if (d.pool && ((d.h & 1) || (d.w & 1))) {
    set_error("igemm: pooled layer needs even h,w (got %dx%d)", d.h, d.w);
    return FLM_ERR_SHAPE;
  }
  const int stride = d.stride > 0 ? d.stride : 1;
  const int ho = (d.h + 2 * d.pad - d.kh) / stride + 1, wo = (d.w + 2 * d.pad - d.kw) / stride + 1;
  if ((stride != 1 || d.res) && (d.pool || d.posmajor)) {
    set_error("igemm: stride / residual are only built for the row-major pixel order");
    return FLM_ERR_UNSUPPORTED;
  }
  if (stride == 1 && (ho != d.h || wo != d.w)) {
    set_error("igemm: stride-1 layers must be 'same' (k=%dx%d pad=%d)", d.kh, d.kw, d.pad);
    return FLM_ERR_SHAPE;
  }
  const long long M = (long long)d.n * ho * wo;
  if (M <= 0 || M > (1ll << 30)) {
    set_error("igemm: pixel count %lld out of range", M);
    return FLM_ERR_SHAPE;
  }
  // byte offsets inside the kernel are 32-bit
  if ((long long)d.n * d.h * d.w * d.cin * es >= (1ll << 32) || (long long)d.coutpad * d.kh * d.kw * d.cin >= (1ll << 31)) {
    set_error("igemm: tensor exceeds the 32-bit offset range (split the batch)");
    return FLM_ERR_SHAPE;
  }
  IgemmArgs a;
  a.x = d.x; a.wt = d.wt; a.scale = d.scale; a.shift = d.shift; a.y = d.y;
  a.out_f32 = d.bf16 ? d.out_f32 : 1;
  a.relu_max = d.relu == 2 ? 6.0f : 3.402823466e38f;
  a.n = d.n; a.h = d.h; a.w = d.w; a.cin = d.cin; a.cout = d.cout; a.ldc = d.ldc;
  a.ho = ho; a.wo = wo; a.stride = stride; a.res = d.res;
  a.kh = d.kh; a.kw = d.kw; a.pad = d.pad;
  a.M = (int)M;
  a.K = d.kh * d.kw * d.cin;
  a.mtiles = cdiv(a.M, BM);
  a.cpt = d.cin / bke;
  a.kw_magic = (65536 + d.kw - 1) / d.kw;
  // split-K (over the channel chunks of every tap) for layers whose tile grid cannot fill the chip: score5
  // (32 workgroups at batch 64), fc6 / fc7 at batches of a few faces (32 workgroups walk 205 MB of weights)
  a.ksplit = 1;
  a.gm = a.gn = 1;
  a.grp_magic = 0;
  a.posperm_on = 0;
  for (int k = 0; k < 8; ++k) a.posperm[k] = 0ull;
  a.part = nullptr;
  const int tiles = cdiv(a.M, BM) * cdiv(d.cout, BN);
  // The slice count is a step function of the tile count, the same for fc6 and fc7 (<= 64 tiles, i.e. up to 4
  // faces: 8 slices; up to 8 faces: 4; up to 16: 2): batches inside one bracket sum in the same order, so a face's
  // result does not depend on its neighbours there.  3x3 encoder layers join in when they have 4 chunks per tap and at
  // most 16 tiles per face, so that one to four faces share the bracket (enc4, enc5: 16 and 4 workgroups for one face;
  // enc3 has 64 tiles per face: split at one face and whole at three, its sums would depend on the batch); their reduce applies BN / ReLU and the 2x2 max to the summed quads.
  // (a 3x3 encoder layer with up to 64 tiles per face -- fp32 enc3 -- splits for one to four faces, always 4 ways:
  // 16 workgroups x 36 k-steps for one face otherwise)
  const int faces = d.n > 0 ? d.n : 1;
  const bool few_faces = d.kh * d.kw > 1 && d.cout < 1024 && faces <= 4 && tiles <= 64 * faces;
  // a 1x1 classifier on the fc grid (score5: K = 4096 into one N tile) splits at EVERY batch, always 8 ways beyond 16
  // faces: with a tile cap its sums changed at 129 faces, the only layer whose bits depended on the batch beyond the
  // documented brackets (tools/debug_batch_invariance.py)
  // (fp32, the parity path, only: in bf16 at batch 512 the eight partial planes cost 0.05 ms the layer does not have)
  const bool narrow_1x1 = !d.bf16 && d.kh * d.kw == 1 && d.cin / bke >= 32 && d.cout <= BN;
  const int tile_cap = narrow_1x1 ? 0x7fffffff : ((d.cout >= 1024 || few_faces) ? 256 : 64);  // the wide fc layers keep splitting until they fill the chip
  if (d.splitk_ws && !d.res && stride == 1 && tiles <= tile_cap && (!d.pool || (a.M & 3) == 0) &&
      (d.kh * d.kw == 1 ? a.cpt >= 32 : (a.cpt >= 4 && (d.cout >= 1024 || tiles <= 16 * faces || few_faces))) &&
      (d.relu != 2 || d.pool)) {
    int ks = (tiles <= 64 || narrow_1x1) ? 8 : (tiles <= 128 ? 4 : 2);
    if (few_faces && tiles > 16 * faces) ks = 8;  // (capped to 9 * cpt / 8 below: the same count for 1..4 faces)
    int per = d.kh * d.kw == 1 ? 8 : 1;  // chunks a slice should at least hold
    if (d.kh * d.kw == 1 && tiles <= 8) {  // score5 up to 16 faces: 1..8 workgroups walking K = 4096 otherwise
      ks = 32;
      per = 4;
    }
    if (d.kh * d.kw > 1) {
      // the slices cut the dense (chunk, tap) sequence, not whole chunks, at least 8 steps each: fc6's 49 x 8 (bf16:
      // 49 x 4 -- by chunks half of its 8 slices had nothing to do; 16 slices measured no faster than 8); enc3 with
      // its 9 x 4 steps in fp32 takes 4 slices (one face: 16 workgroups otherwise)
      if (ks > d.kh * d.kw * a.cpt / 8) ks = d.kh * d.kw * a.cpt / 8;
    } else if (ks > a.cpt / per) {
      ks = a.cpt / per;
    }
    // (few_faces: the decision must not depend on how many of the 1..4 faces are present -- size for four)
    const size_t part_rows = few_faces ? (size_t)(a.M / faces) * 4 : (size_t)a.M;
    if (ks > 1 && (size_t)ks * part_rows * d.ldc * sizeof(float) <= d.splitk_ws_bytes) {
      a.ksplit = ks;
      a.part = d.splitk_ws;
    }
  }
  // only whole N tiles that hold stored columns are launched
  a.ntiles = cdiv(d.cout, BN);
  if (d.bf16) {
    const int sk = launch_score1x1_bf16(s, a, d.relu, d.pool, d.posmajor, d.coutpad);
    if (sk < 0) return sk;
    if (sk == 1) return FLM_OK;
    const int halo = launch_conv3_halo_bf16(s, a, d.relu, d.pool, d.posmajor);
    if (halo < 0) return halo;
    if (halo == 1) return FLM_OK;
    const int big = launch_igemm_bf16_big(s, a, d.relu, d.pool, d.posmajor, d.coutpad);
    if (big < 0) return big;
    if (big == 1) return FLM_OK;
  }
  int rc;
  const int f32_knob = g_f32_group.load(std::memory_order_relaxed);
  if (d.bf16) {
    rc = dispatch<true>(s, d, a);
  } else if (f32_knob < 0) {
    rc = dispatch<false>(s, d, a);
  } else {
    // groups of the third accumulation level: floor(sqrt(k-steps per output)) steps, at least 2
    const int dense = d.kh * d.kw * a.cpt;
    if (dense >= 8192) {
      set_error("igemm: %d k-steps per output exceed the group divider's range", dense);
      return FLM_ERR_SHAPE;
    }
    int grp = 2;
    while ((grp + 1) * (grp + 1) <= dense) ++grp;
    a.grp_magic = (65536 + grp - 1) / grp;
    // buffer offsets at or above 0x80000000 mean "zero padding", so the operand a launch addresses stays below 2 GiB:
    // larger batches go in slices of whole faces (faces are independent rows of the GEMM)
    const long long face_in = (long long)d.h * d.w * d.cin * 4,
                    face_out = (long long)(d.pool ? (ho / 2) * (wo / 2) : ho * wo) * d.ldc * 4;
    const long long lim = (1ll << 31) - (1ll << 22);
    if (face_in >= lim) {
      set_error("igemm: one face exceeds the 2 GiB buffer range of the fp32 kernel");
      return FLM_ERR_SHAPE;
    }
    const int max_faces = (int)(lim / face_in);
    if (d.n <= max_faces) {
      rc = dispatch<false, true>(s, d, a);
    } else {
      if (a.ksplit > 1) {  // (split-K is for a handful of faces: never reached)
        set_error("igemm: batch slicing is not built for split-K layers");
        return FLM_ERR_UNSUPPORTED;
      }
      rc = FLM_OK;
      for (int f0 = 0; f0 < d.n && rc == FLM_OK; f0 += max_faces) {
        IgemmArgs b = a;
        b.n = d.n - f0 < max_faces ? d.n - f0 : max_faces;
        b.x = reinterpret_cast<const char*>(a.x) + (size_t)f0 * face_in;
        b.y = reinterpret_cast<char*>(a.y) + (size_t)f0 * face_out;
        if (a.res) b.res = reinterpret_cast<const char*>(a.res) + (size_t)f0 * face_out;
        b.M = b.n * ho * wo;
        b.mtiles = cdiv(b.M, BM);
        rc = dispatch<false, true>(s, d, b);
      }
    }
  }
  if (rc || a.ksplit <= 1) return rc;
  const size_t total = (size_t)a.M * d.ldc;
  const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  if (d.pool) {
    splitk_reduce_pool_kernel<<<blocks, 256, 0, s>>>(a.part, a.scale, a.shift, a.y, a.M, d.ldc, d.cout, a.ksplit, d.relu,
                                                     a.relu_max, d.bf16 && !d.out_f32);
    FLM_LAUNCH_CHECK("splitk_reduce_pool_kernel");
    return FLM_OK;
  }
  splitk_reduce_kernel<<<blocks, 256, 0, s>>>(a.part, a.scale, a.shift, a.y, a.M, d.ldc, d.cout, a.ksplit, d.relu,
                                              d.bf16 && !d.out_f32);
  FLM_LAUNCH_CHECK("splitk_reduce_kernel");
  return FLM_OK;
}

}  // namespace flm
