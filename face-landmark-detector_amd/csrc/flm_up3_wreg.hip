// up3 in landmark mode (epilogue 3), bf16, 68 classes, stride 8: the WEIGHTS of a phase stay in registers and the input
// positions stream past them ("wreg"; networks/fcn.py:121-122 + networks/utils.py:28-30 + the top-n selection of
// utils/metrics.py:66-77, as flm_convt.hip describes it).
//
// up3_cand8_kernel (flm_convt.hip) keeps a wave's X fragments in registers and streams all 64 phases' weights past
// them: every CU pulls 64 x 45 KiB through an LDS ring per 256 positions (LDS-DMA requests inside the slot stream, one
// barrier per phase for all eight waves, every wave reading the whole ring step), and round 3's ablations priced the
// requests at 0.4 ms and the wait + barrier at 0.2 ms of its 1.9 ms.  Here the roles are swapped:
//   * a wave owns ONE phase (a0, b0) for its whole life: the 36 fragment pieces of the phase's four common class tiles
//     (9 k groups x 4: 128 accumulation registers, which the matrix instructions read where they lie, + 16 ordinary
//     ones: with two waves per SIMD hipcc splits the 256 registers evenly between the two files) are loaded once;
//     the eight waves of a workgroup are the eight phases of two leader groups, and a group's shared fifth tile (classes
//     64..67 of its four phases on rows 4q' + j, flm_pack.hip) sits in LDS, 9 KiB per group: a wave multiplies it like
//     the others and keeps register j of the result -- the bits up3_cand8_kernel gets;
//   * the input, converted ONCE per launch to bf16 with its zero ring ([face][hi+2][wi+2][72], up3_xpack_kernel),
//     streams through LDS in bands of position rows (two buffers; the next band arrives one 16-byte piece per lane and
//     tile while the current one is multiplied); a tile's X fragments are 9 ds_read_b128 from the band through a ring
//     of two register sets, one k group ahead of the MFMAs (FLM_WREG_AHEAD);
//   * two waves per SIMD and NO software pipeline inside a wave: a tile is 45 MFMAs, then its softmax / threshold
//     test on the results in place; the matrix work of one wave runs under the vector work of the other.  (The first
//     form -- one wave per SIMD with all 45 pieces in 180 registers and the epilogue of tile t-1 dealt over the MFMA
//     slots of tile t -- measured 2.8 ms: a lone wave issues one instruction per 4 cycles whatever its kind (tools/valu_latency.hip),
//     and its MFMAs and vector instructions simply added up, 1.0 + 1.5 + 0.3 ms.)
//   * one barrier per BAND, not per phase; a lane with some p >= tau among its 17 values stores them as a RECORD in
//     its wave's LDS list (one test per tile: the maximum of p - tau); the re-test against the thresholds
//     and the keys happen when the list is flushed, a few times per face.
// Arithmetic per value is the generic kernel's, operation by operation (same MFMA order over k, same max / exp / sum
// order / reciprocal / product), so the keys carry the bits of its materialising and sampling launches: thresholds stay
// valid and landmarks are bit-identical (tests/test_gpu_candidates.py).
#include "flm_convt_dev.h"

namespace flm {
namespace wreg {

constexpr int MT = 5, G = 9, WAVES = 8, CP = 72;
constexpr int POSB = CP * 2;              // bytes of one position (72 bf16)
constexpr int PIECE = 1024;               // one (g, m) fragment tile: 64 lanes x 16 bytes
constexpr int PHASE_BYTES = G * MT * PIECE;
constexpr int ROUND = WAVES * PIECE;      // bytes one round of the workgroup's band fetch moves (16 per lane)
constexpr int W5_BYTES = 2 * G * PIECE;   // the two leader groups' fifth tiles
constexpr int GA = 8;                     // k groups whose pieces live in accumulation registers (128 of them)
// Hit records: a lane with some p >= tau among its 17 values of a tile stores all 17 and its pixel (80 bytes) in its
// wave's record list; the re-test and the keys happen when the list is flushed, not in the stream
constexpr int REC_BYTES = 80;             // [0..16] the probabilities' bits, [17] pixel | 4q << 17, [18..19] unused
constexpr int REC_CAP = 64;               // records per wave: a tile adds at most 64; flushed when the next tile's do not fit
constexpr int LIST_BYTES = WAVES * REC_CAP * REC_BYTES;
constexpr int TAU_BYTES = 2 * 80 * 4;    // the clamped thresholds of two faces (by face parity), 68 + 12 floats each
constexpr int LDS_TOTAL = 160 * 1024;
constexpr int BAND_MAX = (LDS_TOTAL - LIST_BYTES - W5_BYTES - TAU_BYTES) / 2 / ROUND * ROUND;  // one band buffer, whole rounds

// k groups the fragment reads run ahead of the MFMAs (1: rings of two register sets; 2: rings of three -- the same 1.94 ms,
// and 12 bytes of scratch in the per-band code)
#ifndef FLM_WREG_AHEAD
#define FLM_WREG_AHEAD 1
#endif
constexpr int AH = FLM_WREG_AHEAD, RING = AH + 1;
// Developer ablations (-DFLM_WREG_ABLATE=<mask>; wrong results, timings only): 1 no softmax / threshold ops, 2 no MFMAs,
// 4 no fragment reads, 8 no band fetch, 16 no record test
#ifndef FLM_WREG_ABLATE
#define FLM_WREG_ABLATE 0
#endif

struct Args {
  const void* xp;             // [n][hi+2][wi+2][72] bf16: x with a zero ring (row / column 0 = index -1)
  const void* wf;             // packed up3 weights, [phase][g][m][lane][16 B]
  const float* tau;           // [n][68]
  unsigned long long* cand;
  unsigned* cand_cnt;
  const unsigned* gate;
  int cand_cap;
  int n, hi, wi, ho, wo;
  int wi1;                    // positions per row (wi + 1)
  unsigned wi1_magic;         // floor(2^32 / wi1) + 1: pl / wi1 = umulhi(pl, magic) for pl < 2^16
  int pitch;                  // (wi + 2) * 144
  int face_bytes;             // (hi + 2) * pitch
  int rb;                     // position rows per band
  int nband;                  // bands per face
  int band_stride;            // LDS bytes of one band buffer: (rb + 1) * pitch rounded up to whole rounds
  int ndma;                   // rounds per band (band_stride / ROUND)
  int chunks;                 // face chunks; workgroup = (chunk, phase octet)
  int xp_bytes;               // n * face_bytes
};

}  // namespace wreg

// x [n][hi][wi][72] fp32 -> bf16 with a zero ring, [n][hi+2][wi+2][72]: one thread per (position, 8 channels).  The
// conversion is the one the other kernels apply when they build their X fragments ((__bf16)float: round to nearest even).
__global__ __launch_bounds__(256) void up3_xpack_kernel(const float* __restrict__ x, uint4* __restrict__ xp, int n, int hi, int wi,
                                                        const unsigned* gate) {
  if (gate && *gate == 0) return;
  const long long total = (long long)n * (hi + 2) * (wi + 2) * 9;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int ch = (int)(i % 9);
  long long pos = i / 9;
  const int cc = (int)(pos % (wi + 2));
  pos /= (wi + 2);
  const int rr = (int)(pos % (hi + 2));
  const int f = (int)(pos / (hi + 2));
  uint4 o = make_uint4(0u, 0u, 0u, 0u);
  if (rr >= 1 && rr <= hi && cc >= 1 && cc <= wi) {
    const float4* src = reinterpret_cast<const float4*>(x + (((size_t)f * hi + (rr - 1)) * wi + (cc - 1)) * wreg::CP + 8 * ch);
    const float4 v0 = src[0], v1 = src[1];
    bf16x8 t;
    t[0] = (__bf16)v0.x; t[1] = (__bf16)v0.y; t[2] = (__bf16)v0.z; t[3] = (__bf16)v0.w;
    t[4] = (__bf16)v1.x; t[5] = (__bf16)v1.y; t[6] = (__bf16)v1.z; t[7] = (__bf16)v1.w;
    o = __builtin_bit_cast(uint4, t);
  }
  xp[i] = o;
}

__global__ __launch_bounds__(wreg::WAVES * 64, 2) void up3_wreg_kernel(wreg::Args a) {  // two waves per SIMD: 256 registers each
  using namespace wreg;
  if (a.gate && *a.gate == 0) return;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  typedef __attribute__((address_space(3))) char lds_char;
  typedef __attribute__((address_space(3))) f32x4 lds_f32x4w;
  typedef __attribute__((address_space(3))) const unsigned lds_u32;
  const unsigned lds0 = (unsigned)(size_t)((lds_char*)smem_raw);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;

  // ---- workgroup -> (face chunk, phase octet): the eight octets of a chunk sit on one XCD (consecutive workgroup ids go
  //      round the eight XCDs), so the chunk's input is fetched into that XCD's L2 once -----------------------------------
  const int L = blockIdx.x;
  const int oct = (L >> 3) & 7;
  const int chunk = (L >> 6) * 8 + (L & 7);
  if (chunk >= a.chunks) return;
  const int f0 = (int)(((long long)a.n * chunk) / a.chunks), f1 = (int)(((long long)a.n * (chunk + 1)) / a.chunks);
  const int nfaces = f1 - f0;
  if (nfaces <= 0) return;
  const int phase = 8 * oct + wave;       // packed phase a0 * 8 + b0: this wave's, for good
  const int a0 = phase >> 3, b0 = phase & 7;
  const int j5 = wave & 3;                // its row 4q' + j5 of the leader's fifth tile
  const int hi1 = a.hi + 1;

  // LDS: [band buffer 0][band buffer 1][fifth tiles: group][g][lane x 16 B][records: wave][REC_CAP][80 B][thresholds: 2][80]
  const unsigned w5_lds = lds0 + (unsigned)(2 * a.band_stride) + (unsigned)((wave >> 2) * G * PIECE) + (unsigned)lane * 16u;
  const unsigned rec0 = lds0 + (unsigned)(2 * a.band_stride + W5_BYTES + wave * REC_CAP * REC_BYTES);
  const unsigned tau_lds = lds0 + (unsigned)(2 * a.band_stride + W5_BYTES + LIST_BYTES);
  unsigned wcnt = 0;  // records in this wave's list (wave-uniform)

  // ---- this wave's weights: the 36 pieces of the common class tiles, lane-linear; the groups' fifth tiles into LDS ------
  f32x4 W[G][4];
  {
    const f32x4* wsrc = reinterpret_cast<const f32x4*>(static_cast<const char*>(a.wf) + (size_t)phase * PHASE_BYTES) + lane;
    // (three k groups at a time: the loads land in ordinary registers first, and there are 128 of those)
#pragma unroll
    for (int g0 = 0; g0 < G; g0 += 3) {
#pragma unroll
      for (int g = g0; g < g0 + 3; ++g)
#pragma unroll
        for (int m = 0; m < 4; ++m) W[g][m] = wsrc[(g * MT + m) * 64];
#pragma unroll
      for (int g = g0; g < g0 + 3; ++g)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          // from here on the piece lives in accumulation registers -- the first 32 of the 36: with two waves per SIMD
          // hipcc halves the 256 registers between the two files, so the last k group's four pieces stay ordinary ones
          if (g < GA) asm volatile("" : "+a"(W[g][m]));
          else asm volatile("" : "+v"(W[g][m]));
        }
    }
    for (int i = tid; i < 2 * G * 64; i += WAVES * 64) {  // piece (grp, g), 16 bytes of lane i & 63
      const int grp = i / (G * 64), rem = i - grp * (G * 64), g = rem >> 6, l = rem & 63;
      const f32x4 v = reinterpret_cast<const f32x4*>(static_cast<const char*>(a.wf) + (size_t)(8 * oct + 4 * grp) * PHASE_BYTES)[(g * MT + 4) * 64 + l];
      *reinterpret_cast<lds_f32x4w*>(lds0 + (unsigned)(2 * a.band_stride) + (unsigned)i * 16u) = v;
    }
  }
  // fragment offsets inside a band: k8 = 32 g + 8 q -> tap (di, dj) = k8 / 72, channel c = k8 % 72
  int delta[G];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const int k8 = 32 * g + 8 * q, tap = k8 / CP, c = k8 % CP;
    delta[g] = 2 * c - ((tap >> 1) * a.pitch + (tap & 1) * POSB);
  }

  // ---- input bands: global -> registers -> LDS, one 16-byte piece per lane and tile (the piece loaded during one tile is
  //      stored at the start of the next) --------------------------------------------------------------------------------
  const __amdgpu_buffer_rsrc_t xsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.xp), 0, a.xp_bytes, 0x00020000);
  const unsigned voff = (unsigned)tid * 16u;
  int dma_src = 0, dma_dst = 0, dma_round = a.ndma;  // the band being fetched: source offset in xp, LDS buffer offset, next round
  f32x4 stage = (f32x4){0.f, 0.f, 0.f, 0.f};
  unsigned stage_dst = 0xffffffffu;      // LDS address the staged piece goes to (none: ~0)
  auto dma_store = [&]() __attribute__((always_inline)) {
    if (stage_dst != 0xffffffffu) {
      *reinterpret_cast<lds_f32x4w*>(stage_dst) = stage;
      stage_dst = 0xffffffffu;
    }
  };
  auto dma_step = [&]() __attribute__((always_inline)) {
    dma_store();
    if (dma_round < a.ndma) {
      const int off = dma_round * ROUND;
      if (!(FLM_WREG_ABLATE & 8)) {
        stage = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xsrd, voff, __builtin_amdgcn_readfirstlane(dma_src + off), 0));  // (said to be uniform: hipcc built a waterfall loop round the load)
        stage_dst = lds0 + (unsigned)(dma_dst + off) + voff;
      }
      ++dma_round;
    }
  };
  auto band_rows = [&](int b) __attribute__((always_inline)) { return hi1 - b * a.rb < a.rb ? hi1 - b * a.rb : a.rb; };

  // The wave's records -> keys in the face's global list.  Lane l takes record base + l: its 17 probabilities against the
  // thresholds of their classes (the clamped tau the stream compared with, bit for bit), the hits counted, one atomic for
  // the wave's range, the keys written.  Runs a few times per face and wave.
  auto rec_flush = [&](int face, unsigned tau_tab) __attribute__((always_inline)) {  // tau_tab: the face's threshold table in LDS
    typedef __attribute__((address_space(3))) const float lds_f32r;
    for (unsigned base = 0; base < wcnt; base += 64) {
      const bool act = base + lane < wcnt;
      const unsigned ra = rec0 + (act ? base + lane : 0u) * REC_BYTES;
      const unsigned pixq = *reinterpret_cast<lds_u32*>(ra + 68);
      const unsigned q4 = (pixq >> 17) & 15u;   // 4 q of the lane that wrote the record
      unsigned hits = 0;
#pragma unroll
      for (int i = 0; i < 17; ++i) {
        const float pv = __uint_as_float(*reinterpret_cast<lds_u32*>(ra + 4 * i));
        const unsigned cls = i < 16 ? 16u * (i >> 2) + q4 + (i & 3) : 64u + (q4 >> 2);
        const float t = *reinterpret_cast<lds_f32r*>(tau_tab + 4u * cls);
        if (act && pv >= t) hits |= 1u << i;
      }
      const unsigned mine = __builtin_popcount(hits);
      unsigned incl = mine;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const unsigned o = __shfl_up(incl, d);
        if (lane >= d) incl += o;
      }
      const unsigned total = __builtin_amdgcn_readlane(incl, 63);
      if (total) {
        unsigned gbase = 0;
        if (lane == 0) {
          gbase = atomicAdd(&a.cand_cnt[face], total);
          if (gbase + total > (unsigned)a.cand_cap) atomicOr(&a.cand_cnt[a.n], 1u);
        }
        unsigned at = __builtin_amdgcn_readfirstlane(gbase) + incl - mine;
        const unsigned pix = pixq & 0x1ffffu;
#pragma unroll
        for (int i = 0; i < 17; ++i)
          if ((hits >> i) & 1u) {
            const unsigned pb = *reinterpret_cast<lds_u32*>(ra + 4 * i);
            const unsigned cls = i < 16 ? 16u * (i >> 2) + q4 + (i & 3) : 64u + (q4 >> 2);
            // (p >= tau > 0: its order bits are its bits with the sign set)
            if (at < (unsigned)a.cand_cap) a.cand[(size_t)face * a.cand_cap + at] = ((unsigned long long)(pb | 0x80000000u) << 32) | (cls << 17) | pix;
            ++at;
          }
      }
    }
    wcnt = 0;
  };

  // ---- first band in, then the stream ------------------------------------------------------------------------------------
  dma_src = f0 * a.face_bytes; dma_dst = 0; dma_round = 0;
  for (int k = 0; k < a.ndma; ++k) dma_step();

  int unit = 0;  // (face, band) in stream order
  for (int f = 0; f < nfaces; ++f) {
    const int face = f0 + f;
    // thresholds of the face, clamped to FLT_MIN so that p >= tau implies p > 0 (a zero weight cannot move a centroid; a
    // class left with fewer than n keys is caught by cand_merge_kernel): a table in LDS, by face parity -- a wave is at
    // most one band ahead of the others, and the band's barrier below publishes the table
    if (tid < 80) {
      typedef __attribute__((address_space(3))) float lds_f32w;
      *reinterpret_cast<lds_f32w*>(tau_lds + (unsigned)((f & 1) * 320 + tid * 4)) =
          tid < 68 ? fmaxf(a.tau[(size_t)face * 68 + tid], 1.17549435e-38f) : 3.402823466e38f;
    }
    const unsigned tau_q = tau_lds + (unsigned)((f & 1) * 320 + 16 * q);  // this lane's classes 16m + 4q + e at + 64 m; 64 + q at 256 + ...
    for (int b = 0; b < a.nband; ++b, ++unit) {
      dma_store();      // (a band of as many rounds as tiles: its last piece)
      __syncthreads();  // this band is in LDS (every wave's pieces; the first time also the fifth tiles), the face's
                        // thresholds are, and every wave is done with the band before
      {  // fetch the unit after this one into the other buffer
        int nf = f, nb = b + 1;
        if (nb == a.nband) { nb = 0; ++nf; }
        if (nf < nfaces) {
          dma_src = (f0 + nf) * a.face_bytes + nb * a.rb * a.pitch;
          dma_dst = ((unit + 1) & 1) * a.band_stride;
          dma_round = 0;
        }
      }
      const int rows = band_rows(b);
      const int npos = rows * a.wi1;
      const int ntile = (npos + 15) >> 4;
      const unsigned buf = lds0 + (unsigned)((unit & 1) * a.band_stride + a.pitch + POSB);
      // X fragments and the fifth tile's pieces go through rings of RING register sets, read AH k groups ahead of the MFMAs
      // that use them, across tiles: the last groups of a tile request the first of the next one, so those land under
      // the softmax (AH = 1: once group 8's MFMAs have been issued, into its set; AH = 2 measured the same).  Reads and
      // waits are by hand; the queue is in order, so "at most 2 AH outstanding" after a group's two requests means the
      // group's own pair has landed whatever else was queued between.
      f32x4 xr[RING], w5[RING];
      auto tile_pos = [&](int t, unsigned& ib, unsigned& jj) __attribute__((always_inline)) {
        const unsigned pl = (unsigned)(16 * t + r);
        ib = __umulhi(pl, a.wi1_magic);
        jj = pl - ib * (unsigned)a.wi1;
        return pl < (unsigned)npos;
      };
      unsigned posaddr, ib, jj;   // the current tile's LDS address and (row, column), carried from the tile before
      bool valid = tile_pos(0, ib, jj);
      posaddr = valid ? buf + ib * (unsigned)a.pitch + jj * (unsigned)POSB : buf;
#define FLM_XRD(SLOT, GG, PA) asm volatile("ds_read_b128 %0, %1" : "=v"(xr[SLOT]) : "v"((PA) + (unsigned)delta[GG]))
#define FLM_WRD(SLOT, GG) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(w5[SLOT]) : "v"(w5_lds), "n"((GG) * PIECE))
      if constexpr (!(FLM_WREG_ABLATE & 4)) {
        FLM_XRD(0, 0, posaddr); FLM_WRD(0, 0);
        if constexpr (AH == 2) { FLM_XRD(1, 1, posaddr); FLM_WRD(1, 1); }
      }
      for (int t = 0; t < ntile; ++t) {
        dma_step();
        // ---- this lane's position of the tile: (row, column), output pixel; the next tile's LDS address ---------------
        const unsigned oy = 8u * ((unsigned)(b * a.rb) + ib) + (unsigned)a0, ox = 8u * jj + (unsigned)b0;
        const bool okpix = valid && oy < (unsigned)a.ho && ox < (unsigned)a.wo;
        const unsigned pixq = oy * (unsigned)a.wo + ox + ((unsigned)(4 * q) << 17);
        const bool has_next = t + 1 < ntile;
        unsigned posnext;
        valid = tile_pos(t + 1, ib, jj) && has_next;   // (from here on: the next tile's)
        posnext = valid ? buf + ib * (unsigned)a.pitch + jj * (unsigned)POSB : buf;

        // ---- 45 MFMAs.  The weight operand of the common tiles is named as an accumulation register (k groups below GA);
        //      the first MFMA of an accumulator takes a zero srcC. -------------------------------------------------------
        f32x4 acc[MT];
        cand8::static_for<G>([&](auto gc) __attribute__((always_inline)) {
          constexpr int g = decltype(gc)::value;
          if constexpr (!(FLM_WREG_ABLATE & 4)) {
            // after a group's pair of requests at most 2 AH reads may be outstanding for its own pair to have landed
            if constexpr (g + AH < G) {
              FLM_XRD((g + AH) % RING, g + AH, posaddr); FLM_WRD((g + AH) % RING, g + AH);
              __builtin_amdgcn_s_waitcnt(AH == 2 ? 0xc47f : 0xc27f);
            } else if (AH == 2 && has_next) {   // (rings of three: 9 groups keep the slots in step from tile to tile)
              FLM_XRD((g + AH) % RING, g + AH - G, posnext); FLM_WRD((g + AH) % RING, g + AH - G);
              __builtin_amdgcn_s_waitcnt(0xc47f);
            } else if constexpr (g + 2 == G && AH == 2) {
              __builtin_amdgcn_s_waitcnt(0xc27f);    // only group 8's pair may still be out
            } else {
              __builtin_amdgcn_s_waitcnt(0xc07f);    // lgkmcnt(0)
            }
          }
          if constexpr (!(FLM_WREG_ABLATE & 2)) {
#pragma unroll
            for (int m = 0; m < 4; ++m) {
              if constexpr (g == 0) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(acc[m]) : "a"(W[g][m]), "v"(xr[g % RING]));
              else if constexpr (g < GA) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[m]) : "a"(W[g][m]), "v"(xr[g % RING]));
              else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[m]) : "v"(W[g][m]), "v"(xr[g % RING]));
            }
            if constexpr (g == 0) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(acc[4]) : "v"(w5[g % RING]), "v"(xr[g % RING]));
            else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[4]) : "v"(w5[g % RING]), "v"(xr[g % RING]));
          }
        });
        // (rings of two: group 0 of the next tile goes where group 8 was, once group 8's MFMAs have been issued)
        if constexpr (AH == 1 && !(FLM_WREG_ABLATE & 4))
          if (has_next) { FLM_XRD(0, 0, posnext); FLM_WRD(0, 0); }
        posaddr = posnext;
#undef FLM_XRD
#undef FLM_WRD
        // (the compiler cannot see into the MFMAs: the wait states between the last of them and the first vector
        //  instruction that reads a result are ours to leave)
        // (tied to the accumulators: nothing that reads one may be scheduled above it)
        asm volatile("s_nop 15" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]));

        if constexpr (!(FLM_WREG_ABLATE & 1)) {
          // ---- softmax over the 68 classes of the pixel: this lane's 17 values (classes 16m + 4q + e, and 64 + q from row
          //      4q + j5 of the shared tile), lane groups q = 0..3 joined by the swaps ----------------------------------
          float v[17];
#pragma unroll
          for (int i = 0; i < 16; ++i) v[i] = acc[i >> 2][i & 3];
          v[16] = j5 == 0 ? acc[4][0] : j5 == 1 ? acc[4][1] : j5 == 2 ? acc[4][2] : acc[4][3];
          float mx = max_raw(v[0], v[1]);
#pragma unroll
          for (int i = 2; i < 16; i += 2) mx = max3_raw(mx, v[i], v[i + 1]);
          mx = max_raw(mx, v[16]);
          mx = reduce_q_max(mx);
          const float nmxl = -mx * 1.44269504088896340736f;
          // (exponent arguments two at a time, v_pk_fma_f32: the same fma per value; an instruction of any kind costs a wave
          //  ~2.3 ns, tools/valu_latency.hip)
          typedef float f32x2 __attribute__((ext_vector_type(2)));
          const f32x2 l2 = {1.44269504088896340736f, 1.44269504088896340736f}, n2 = {nmxl, nmxl};
#pragma unroll
          for (int i = 0; i < 16; i += 2) {
            const f32x2 t2 = __builtin_elementwise_fma((f32x2){v[i], v[i + 1]}, l2, n2);
            v[i] = __builtin_amdgcn_exp2f(t2[0]);
            v[i + 1] = __builtin_amdgcn_exp2f(t2[1]);
          }
          v[16] = __builtin_amdgcn_exp2f(__builtin_fmaf(v[16], 1.44269504088896340736f, nmxl));
          float sum = 0.f + v[0];
#pragma unroll
          for (int i = 1; i < 17; ++i) sum += v[i];
          const float sq = reduce_q_sum(sum);
          const float rs = okpix ? __builtin_amdgcn_rcpf(sq) : 0.f;
#pragma unroll
          for (int i = 0; i < 17; ++i) v[i] *= rs;
          if constexpr (!(FLM_WREG_ABLATE & 16)) {
            // ---- any p >= tau?  The lanes whose largest p - tau is >= 0 store a record ------------------------------------
            typedef __attribute__((address_space(3))) const f32x4 lds_f32x4r;
            typedef __attribute__((address_space(3))) const float lds_f32r;
            f32x4 tq4[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) tq4[m] = *reinterpret_cast<lds_f32x4r*>(tau_q + 64 * m);
            const float tq16 = *reinterpret_cast<lds_f32r*>(tau_q + 256 - 12 * q);  // class 64 + q
            // (d = p - tau in float, two at a time: the sign of a float difference is exact, and a NaN p never wins a v_max3)
            float hmax = v[16] - tq16;
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
              const f32x2 d2 = (f32x2){v[i], v[i + 1]} - (f32x2){tq4[i >> 2][i & 3], tq4[i >> 2][(i & 3) + 1]};
              hmax = max3_raw(hmax, d2[0], d2[1]);
            }
            const unsigned long long mk = __ballot(hmax >= 0.f);
            if (mk) {
              const unsigned add = (unsigned)__builtin_popcountll(mk);
              if (wcnt + add > (unsigned)REC_CAP) rec_flush(face, tau_q - 16u * q);
              if (hmax >= 0.f) {
                const unsigned slot = wcnt + __builtin_amdgcn_mbcnt_hi((unsigned)(mk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk, 0u));
                const unsigned ra = rec0 + slot * REC_BYTES;
#pragma unroll
                for (int m = 0; m < 4; ++m) *reinterpret_cast<lds_f32x4w*>(ra + 16 * m) = (f32x4){v[4 * m], v[4 * m + 1], v[4 * m + 2], v[4 * m + 3]};
                *reinterpret_cast<lds_f32x4w*>(ra + 64) = (f32x4){v[16], __uint_as_float(pixq), 0.f, 0.f};
              }
              wcnt += add;
            }
          }
        }
      }
    }
    if (wcnt) rec_flush(face, tau_q - 16u * q);  // the face's last records
  }
}

static std::atomic<int> g_wreg{0};  // A/B knob "up3_wreg": 1 = the bf16 candidate launch of up3 runs this kernel (default 0: it ties with
                                    // up3_cand8_kernel, 1.94 against 1.94-1.97 ms per 512 faces, and that one takes every shape)
void convt_wreg_enable(int on) { g_wreg.store(on, std::memory_order_relaxed); }

// 1: launched; 0: not this kernel's case (the caller takes up3_cand8_kernel); < 0: error
int launch_up3_wreg(hipStream_t st, const ConvTArgs& c, void* scratch, size_t scratch_bytes) {
  using namespace wreg;
  if (!g_wreg.load(std::memory_order_relaxed)) return 0;
  if (c.s != 8 || c.C != 68 || c.Cp != CP || !scratch || c.hi < 2 || c.wi < 2) return 0;
  Args a;
  a.wf = c.wf; a.tau = c.tau; a.cand = c.cand; a.cand_cnt = c.cand_cnt; a.gate = c.gate; a.cand_cap = c.cand_cap;
  a.n = c.n; a.hi = c.hi; a.wi = c.wi; a.ho = c.ho; a.wo = c.wo;
  a.wi1 = c.wi + 1;
  a.wi1_magic = (unsigned)((1ull << 32) / (unsigned)a.wi1) + 1u;
  a.pitch = (c.wi + 2) * POSB;
  a.face_bytes = (c.hi + 2) * a.pitch;
  const long long xp_bytes = (long long)c.n * a.face_bytes;
  const int hi1 = c.hi + 1;
  if (xp_bytes > 0x7fffffffll - 2 * BAND_MAX || (size_t)xp_bytes > scratch_bytes || (long long)hi1 * a.wi1 >= 65536 || 2 * a.pitch > BAND_MAX) return 0;
  a.xp_bytes = (int)xp_bytes;
  a.nband = 1;
  for (;; ++a.nband) {
    a.rb = cdiv(hi1, a.nband);
    if ((a.rb + 1) * a.pitch <= BAND_MAX) break;
  }
  a.nband = cdiv(hi1, a.rb);
  a.band_stride = cdiv((a.rb + 1) * a.pitch, ROUND) * ROUND;
  a.ndma = a.band_stride / ROUND;
  const int last_rows = hi1 - (a.nband - 1) * a.rb;
  const int min_tiles = cdiv((last_rows < a.rb ? last_rows : a.rb) * a.wi1, 16);
  if (a.ndma > min_tiles) return 0;  // (a band's pieces arrive one per tile)
  // workgroups: 8 phase octets x face chunks; one workgroup per CU, so about (CUs / 8) chunks
  a.chunks = c.n < 32 ? c.n : 32;
  a.xp = scratch;
  {
    const long long total = (long long)c.n * (c.hi + 2) * (c.wi + 2) * 9;
    up3_xpack_kernel<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(c.x, static_cast<uint4*>(scratch), c.n, c.hi, c.wi, c.gate);
    FLM_LAUNCH_CHECK("up3_xpack_kernel");
  }
  const size_t lds = 2 * (size_t)a.band_stride + W5_BYTES + LIST_BYTES + TAU_BYTES;
  static FuncAttrOnce attr;
  FLM_FUNC_ATTR_ONCE(attr, (&up3_wreg_kernel), LDS_TOTAL);
  up3_wreg_kernel<<<8 * 8 * cdiv(a.chunks, 8), WAVES * 64, lds, st>>>(a);
  FLM_LAUNCH_CHECK("up3_wreg_kernel");
  return 1;
}

}  // namespace flm
