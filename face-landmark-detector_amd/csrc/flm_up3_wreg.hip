// up3 in landmark mode (epilogue 3), bf16, 68 classes, stride 8: the WEIGHTS of a phase stay in registers and the input
// positions stream past them ("wreg"; networks/fcn.py:121-122 + networks/utils.py:28-30 + the top-n selection of
// utils/metrics.py:66-77, as flm_convt.hip describes it).
//
// up3_cand8_kernel (flm_convt.hip) keeps a wave's X fragments in registers and streams all 64 phases' weights past
// them: every CU pulls 64 x 45 KiB through an LDS ring per 256 positions (LDS-DMA requests inside the slot stream, one
// barrier per phase for all eight waves, every wave reading the whole ring step: 36 % of the LDS bandwidth), and round 3's
// ablations priced the requests at 0.4 ms and the wait + barrier at 0.2 ms of its 1.9 ms.  Here the roles are swapped:
//   * a wave owns ONE phase (a0, b0) for its whole life: the phase's 45 fragment pieces (9 k groups x 5 class tiles,
//     180 registers -- the kernel runs one wave per SIMD with the 512-register file, the matrix instructions read
//     them where they lie) are loaded once; the four waves of a workgroup are the four phases of a leader group;
//   * the input, converted ONCE per launch to bf16 with its zero ring ([face][hi+2][wi+2][72], up3_xpack_kernel),
//     streams through LDS in bands of position rows (two buffers; the next band arrives by LDS-DMA, one request
//     per wave and tile, while the current one is multiplied); a tile's X fragments are 9 ds_read_b128 from the band
//     (position stride 144 B = 36 banks: 16 consecutive positions cover all 64 banks once);
//   * no barrier, no request burst and no LDS weight traffic inside the stream: one barrier per BAND (23 tiles at
//     256 x 256), LDS reads of 9 KiB per tile and wave where the ring cost 22.5 KiB per 16 positions;
//   * the softmax / threshold test of tile t-1 is dealt over the 45 MFMA slots of tile t by cost, with the next
//     tile's addressing and fragment reads ahead of it (compile-time schedule, order pinned as in up3_cand8_kernel);
//     hitting lanes append their keys to lane-private lists (no ballot arithmetic, no re-test).
// Every phase multiplies its own five class tiles (the short phases' fifth tile holds their classes 64..67 in rows 4q',
// flm_pack.hip): 45 MFMAs per tile instead of 38.25 with the shared fifth tile -- the matrix pipe is not what bounds
// this kernel.  Arithmetic per value is the generic kernel's, operation by operation (same MFMA order over k, same
// max / exp / sum order / reciprocal / product), so the keys carry the bits of its materialising and sampling launches:
// thresholds stay valid and landmarks are bit-identical (tests/test_gpu_candidates.py).
#include "flm_convt_dev.h"

namespace flm {
namespace wreg {

using cand8::static_for;

constexpr int MT = 5, G = 9, WAVES = 4, CP = 72;
constexpr int POSB = CP * 2;              // bytes of one position (72 bf16)
constexpr int PIECE = 1024;               // one (g, m) fragment tile: 64 lanes x 16 bytes
constexpr int PHASE_BYTES = G * MT * PIECE;
constexpr int ROUND = WAVES * PIECE;      // bytes one round of the workgroup's band fetch moves (16 per lane)
// Hit records: a lane with some p >= tau among its 17 values of a tile stores all 17 and its pixel (80 bytes) in its
// wave's record list; the re-test and the keys happen when the list is flushed (a few times per face), not in the stream
constexpr int REC_BYTES = 80;             // [0..16] the probabilities' bits, [17] pixel | 4q << 17, [18..19] unused
#ifndef FLM_WREG_REC_CAP
#define FLM_WREG_REC_CAP 128
#endif
constexpr int REC_CAP = FLM_WREG_REC_CAP; // records per wave; a flush is due when fewer than 64 slots are left
constexpr int LIST_BYTES = WAVES * REC_CAP * REC_BYTES;
constexpr int LDS_TOTAL = 160 * 1024;
constexpr int BAND_MAX = (LDS_TOTAL - LIST_BYTES) / 2 / ROUND * ROUND;  // one band buffer, whole rounds

// Developer ablations (-DFLM_WREG_ABLATE=<mask>; wrong results, timings only): 1 no softmax / threshold ops, 2 no MFMAs,
// 4 no X fragment reads, 8 no band fetch, 16 no candidate test (stage T)
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
  int dma_per_tile;           // rounds requested per tile step
  int chunks;                 // face chunks; workgroup = (chunk, phase group)
  int xp_bytes;               // n * face_bytes
};

// ---- the work beside the 45 MFMA slots of a tile, as a list of small ops -------------------------------------------
//   N   1   the next tile's positions: (row, column), LDS address, output pixel, validity
//   R   9   its X fragments: address + ds_read_b128 per k group
//   A   9   running class maximum of the finished tile, two values per v_max3
//   B   1   maximum over the four lane groups, -max * log2(e)
//   C  34   e = exp2(x * log2(e) - max * log2(e)): the fma and the v_exp_f32 of a value two ops apart
//   D  17   sum += e (the generic kernel's order)
//   E   1   sum over the lane groups, reciprocal, pixel validity
//   F  17   p = e * (1 / sum)
//   T  26   d = bits(p) - bits(tau) per value (p, tau > 0: the integer order is the float order), running maximum of the
//           d, three per v_max3_i32; one test of the maximum per tile, lanes with a hit store a record
struct Ops {
  static constexpr int N0 = 0, R0 = 1, A0 = R0 + G, B0 = A0 + 9, C0 = B0 + 1, D0 = C0 + 34, E0 = D0 + 17, F0 = E0 + 1, T0 = F0 + 17,
                       NOPS = T0 + 26;
  static constexpr int NSLOT = G * MT;
  static constexpr int cost(int k) {  // issue cycles / 4, roughly
    return k < R0 ? 12 : k < A0 ? 2 : k < B0 ? 1 : k < C0 ? 6 : k < D0 ? (cexp(k - C0) ? 2 : 1) : k < E0 ? 1 : k < F0 ? 8 : 1;
  }
  // stage T, op t: t = 3j, 3j + 1: d of values 2j, 2j + 1; t = 3j + 2: the maximum takes them in (j < 8); t = 24: d of
  // value 16; t = 25: the maximum takes it in
  static constexpr bool tmax(int t) { return t == 25 || (t < 24 && t % 3 == 2); }
  static constexpr int tval(int t) { return t >= 24 ? 16 : 2 * (t / 3) + t % 3; }
  // stage C, op c: the fma of value cval(c) or its v_exp_f32: F0 F1 E0 F2 E1 ... F16 E15 E16
  static constexpr bool cexp(int c) { return c == 33 || (c != 0 && (c & 1) == 0); }
  static constexpr int cval(int c) { return c == 0 ? 0 : c == 33 ? 16 : (c & 1) ? (c + 1) / 2 : c / 2 - 1; }
};
struct Sched {
  int first[Ops::NSLOT + 1];  // ops [first[s], first[s+1]) run beside MFMA slot s
  constexpr Sched() : first() {
    int total = 0;
    for (int k = 0; k < Ops::NOPS; ++k) total += Ops::cost(k);
    int k = 0, cum = 0;
    for (int sl = 0; sl < Ops::NSLOT; ++sl) {
      first[sl] = k;
      while (k < Ops::NOPS && (cum + Ops::cost(k)) * (long long)Ops::NSLOT <= (long long)total * (sl + 1)) cum += Ops::cost(k++);
    }
    first[Ops::NSLOT] = Ops::NOPS;
  }
};
static constexpr Sched kSched{};

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

__global__ __launch_bounds__(wreg::WAVES * 64, 1) void up3_wreg_kernel(wreg::Args a) {  // one wave per SIMD: 512 registers
  using namespace wreg;
  if (a.gate && *a.gate == 0) return;
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  typedef __attribute__((address_space(3))) char lds_char;
  const unsigned lds0 = (unsigned)(size_t)((lds_char*)smem_raw);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;

  // ---- workgroup -> (face chunk, phase group): the sixteen phase groups of a chunk sit on one XCD (consecutive
  //      workgroup ids go round the eight XCDs), so the chunk's input is fetched into that XCD's L2 once ----------------
  const int L = blockIdx.x;
  const int pg = (L >> 3) & 15;
  const int chunk = (L >> 7) * 8 + (L & 7);
  if (chunk >= a.chunks) return;
  const int f0 = (int)(((long long)a.n * chunk) / a.chunks), f1 = (int)(((long long)a.n * (chunk + 1)) / a.chunks);
  const int nfaces = f1 - f0;
  if (nfaces <= 0) return;
  const int phase = 4 * pg + wave;        // packed phase a0 * 8 + b0
  const int a0 = phase >> 3, b0 = phase & 7;
  const int hi1 = a.hi + 1;

  // ---- this wave's weights: 45 pieces, lane-linear -------------------------------------------------------------------
  f32x4 W[G][MT];
  {
    const f32x4* wsrc = reinterpret_cast<const f32x4*>(static_cast<const char*>(a.wf) + (size_t)phase * PHASE_BYTES) + lane;
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int m = 0; m < MT; ++m) W[g][m] = wsrc[(g * MT + m) * 64];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int m = 0; m < MT; ++m) asm volatile("" : "+a"(W[g][m]));  // from here on the piece lives in accumulation registers
  }
  // fragment offsets inside a band: k8 = 32 g + 8 q -> tap (di, dj) = k8 / 72, channel c = k8 % 72
  int delta[G];
#pragma unroll
  for (int g = 0; g < G; ++g) {
    const int k8 = 32 * g + 8 * q, tap = k8 / CP, c = k8 % CP;
    delta[g] = 2 * c - ((tap >> 1) * a.pitch + (tap & 1) * POSB);
  }

  const unsigned rec0 = lds0 + (unsigned)(2 * a.band_stride + wave * REC_CAP * REC_BYTES);  // this wave's records
  unsigned wcnt = 0;  // records in the list (wave-uniform)

  // ---- input bands: global -> registers -> LDS, one 16-byte piece per lane and tile step (the piece loaded in one step is
  //      stored in the next).  (LDS-DMA, the first form, cost 0.6 ms of 2.85: the wave's fragment reads queued up behind
  //      the request it had just made.) ------------------------------------------------------------------------------------
  const __amdgpu_buffer_rsrc_t xsrd = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.xp), 0, a.xp_bytes, 0x00020000);
  const unsigned voff = (unsigned)tid * 16u;
  const int nunits = nfaces * a.nband;   // unit = (face, band) in stream order
  int dma_src = 0, dma_dst = 0, dma_round = a.ndma;  // the band being fetched: source offset in xp, LDS buffer offset, next round
  f32x4 stage = (f32x4){0.f, 0.f, 0.f, 0.f};
  unsigned stage_dst = 0xffffffffu;      // LDS address the staged piece goes to (none: ~0)
  auto dma_store = [&]() __attribute__((always_inline)) {
    if (stage_dst != 0xffffffffu) {
      typedef __attribute__((address_space(3))) f32x4 lds_f32x4w;
      *reinterpret_cast<lds_f32x4w*>(stage_dst) = stage;
      stage_dst = 0xffffffffu;
    }
  };
  auto dma_step = [&]() __attribute__((always_inline)) {
    dma_store();
    if (dma_round < a.ndma) {
      const int off = dma_round * ROUND;
      if (!(FLM_WREG_ABLATE & 8)) {
        stage = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xsrd, voff, dma_src + off, 0));
        stage_dst = lds0 + (unsigned)(dma_dst + off) + voff;
      }
      ++dma_round;
    }
  };
  auto band_rows = [&](int b) __attribute__((always_inline)) { return hi1 - b * a.rb < a.rb ? hi1 - b * a.rb : a.rb; };
  auto band_tiles = [&](int b) __attribute__((always_inline)) { return (band_rows(b) * a.wi1 + 15) >> 4; };
  dma_src = f0 * a.face_bytes; dma_dst = 0; dma_round = 0;
  for (int k = 0; k < a.ndma; ++k) dma_step();
  dma_store();
  __syncthreads();
  // the unit after the one whose tiles are being prefetched: what the next fetch brings in
  int fu_f = a.nband > 1 ? 0 : 1, fu_b = a.nband > 1 ? 1 : 0;  // (face - f0, band) of unit 1
  auto fetch_begin = [&](int u) __attribute__((always_inline)) {  // start fetching unit u = (fu_f, fu_b) into buffer u & 1
    dma_src = (f0 + fu_f) * a.face_bytes + fu_b * a.rb * a.pitch;
    dma_dst = (u & 1) * a.band_stride;
    dma_round = 0;
    if (++fu_b == a.nband) { fu_b = 0; ++fu_f; }
  };
  if (nunits > 1) fetch_begin(1);

  // ---- the tile stream: the tile whose X fragments are read next: unit nu = (nf, nb), tile nt_ of its band ------------
  int nu = 0, nf = 0, nb = 0, nt_ = 0;
  int ntiles_nu = band_tiles(0);

  f32x4 accA[MT], accB[MT], xa[G], xb[G];
#pragma unroll
  for (int m = 0; m < MT; ++m) accA[m] = accB[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int g = 0; g < G; ++g) xa[g] = xb[g] = (f32x4){0.f, 0.f, 0.f, 0.f};
  unsigned pixq_n = 0xffffffffu, pixq_c = 0xffffffffu, pixq_p = 0xffffffffu;  // pixel | 4q << 17 of the next / multiplied / finished tile
  int face_n = -1, face_c = -1, face_p = -1, face_t = -1;                      // their faces; face_t: the thresholds in tq
  float4 tq[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) tq[m] = make_float4(3.402823466e38f, 3.402823466e38f, 3.402823466e38f, 3.402823466e38f);

  // The wave's records -> keys in the face's global list.  Lane l takes record base + l: its 17 probabilities against the
  // thresholds of their classes (the clamped tau the stream compared with, bit for bit), the hits counted, one atomic for
  // the wave's range, the keys written.  Runs a few times per face and wave.
  auto rec_flush = [&](int face) __attribute__((always_inline)) {
    typedef __attribute__((address_space(3))) const unsigned lds_u32;
    const float* tauf = a.tau + (size_t)(face < 0 ? 0 : face) * 68;
    for (unsigned base = 0; base < wcnt; base += 64) {
      const bool act = base + lane < wcnt && face >= 0;
      const unsigned ra = rec0 + (act ? base + lane : 0u) * REC_BYTES;
      const unsigned pixq = *reinterpret_cast<lds_u32*>(ra + 68);
      const unsigned q4 = (pixq >> 17) & 15u;   // 4 q of the lane that wrote the record
      unsigned hits = 0;
#pragma unroll
      for (int i = 0; i < 17; ++i) {
        const float pv = __uint_as_float(*reinterpret_cast<lds_u32*>(ra + 4 * i));
        const unsigned cls = i < 16 ? 16u * (i >> 2) + q4 + (i & 3) : 64u + (q4 >> 2);
        const float t = fmaxf(tauf[cls], 1.17549435e-38f);
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

  float mx = 0.f, nmxl = 0.f, sum = 0.f, rs = 0.f;
  int hmax = -1, dT0 = 0, dT1 = 0;
  unsigned posaddr = 0;
#define FLM_PIN(x) asm volatile("" : "+v"(x))
  // op K of the list on the finished tile's values PV (XN: the fragments of the tile after the multiplied one)
  auto epi_op = [&](auto kc, f32x4(&PV)[MT], f32x4(&XN)[G]) __attribute__((always_inline)) {
    constexpr int K = decltype(kc)::value;
    if constexpr (K < Ops::R0) {
      // next tile (nu, nt_): lane r holds position pl = 16 nt_ + r of the band
      const bool live = nu < nunits;
      const int b = nb;
      const int rows = band_rows(b);
      const unsigned pl = (unsigned)(16 * nt_ + r);
      const unsigned ib = __umulhi(pl, a.wi1_magic);
      const unsigned j = pl - ib * (unsigned)a.wi1;
      const bool valid = live && pl < (unsigned)(rows * a.wi1);
      const unsigned buf = lds0 + (unsigned)((nu & 1) * a.band_stride + a.pitch + POSB);
      posaddr = valid ? buf + ib * (unsigned)a.pitch + j * (unsigned)POSB : buf;
      const unsigned oy = 8u * ((unsigned)(b * a.rb) + ib) + (unsigned)a0, ox = 8u * j + (unsigned)b0;
      pixq_n = (valid && oy < (unsigned)a.ho && ox < (unsigned)a.wo) ? oy * (unsigned)a.wo + ox + ((unsigned)(4 * q) << 17) : 0xffffffffu;
      face_n = live ? f0 + nf : -1;
    } else if constexpr (K < Ops::A0) {
      constexpr int g = K - Ops::R0;
      if constexpr (!(FLM_WREG_ABLATE & 4)) {
        typedef __attribute__((address_space(3))) const f32x4 lds_f32x4;
        XN[g] = *reinterpret_cast<lds_f32x4*>(posaddr + (unsigned)delta[g]);
      }
    } else if constexpr (K < Ops::B0) {
      constexpr int j = K - Ops::A0;
      if constexpr (j == 0) mx = max_raw(PV[0][0], PV[0][1]);
      else if constexpr (j < 8) mx = max3_raw(mx, PV[j >> 1][2 * (j & 1)], PV[j >> 1][2 * (j & 1) + 1]);
      else mx = max_raw(mx, PV[4][0]);
      FLM_PIN(mx);
    } else if constexpr (K < Ops::C0) {
      mx = reduce_q_max(mx);
      nmxl = -mx * 1.44269504088896340736f;
      FLM_PIN(nmxl);
    } else if constexpr (K < Ops::D0) {
      constexpr int c = K - Ops::C0, i = Ops::cval(c), m = i < 16 ? i >> 2 : 4, e = i < 16 ? i & 3 : 0;
      float v;
      if constexpr (!Ops::cexp(c)) v = __builtin_fmaf(PV[m][e], 1.44269504088896340736f, nmxl);
      else v = __builtin_amdgcn_exp2f(PV[m][e]);
      FLM_PIN(v);
      PV[m][e] = v;
    } else if constexpr (K < Ops::E0) {
      constexpr int i = K - Ops::D0, m = i < 16 ? i >> 2 : 4, e = i < 16 ? i & 3 : 0;
      if constexpr (i == 0) sum = 0.f + PV[m][e];
      else sum += PV[m][e];
      FLM_PIN(sum);
    } else if constexpr (K < Ops::F0) {
      const float sq = reduce_q_sum(sum);
      const float inv = __builtin_amdgcn_rcpf(sq);
      rs = pixq_p != 0xffffffffu ? inv : 0.f;
      FLM_PIN(rs);
    } else if constexpr (K < Ops::T0) {
      constexpr int i = K - Ops::F0, m = i < 16 ? i >> 2 : 4, e = i < 16 ? i & 3 : 0;
      float v = PV[m][e] * rs;
      FLM_PIN(v);
      PV[m][e] = v;
    } else if constexpr (!(FLM_WREG_ABLATE & 16)) {
      constexpr int t = K - Ops::T0, i = Ops::tval(t), m = i < 16 ? i >> 2 : 4, e = i < 16 ? i & 3 : 0;
      const float tv = e == 0 ? tq[m].x : e == 1 ? tq[m].y : e == 2 ? tq[m].z : tq[m].w;
      if constexpr (!Ops::tmax(t)) {
        const int d = (int)(__float_as_uint(PV[m][e]) - __float_as_uint(tv));
        if constexpr (t == 24 || t % 3 == 0) dT0 = d;
        else dT1 = d;
      } else if constexpr (t == 2) {
        hmax = dT0 > dT1 ? dT0 : dT1;
      } else if constexpr (t == 25) {
        hmax = hmax > dT0 ? hmax : dT0;
      } else {
        const int mm = dT0 > dT1 ? dT0 : dT1;
        hmax = hmax > mm ? hmax : mm;
      }
    }
  };
  // after the slots: lanes with a hit (the maximum of the d is >= 0) store their record
  auto rec_store = [&](f32x4(&PV)[MT]) __attribute__((always_inline)) {
    if constexpr (FLM_WREG_ABLATE & 16) return;
    const unsigned long long mk = __ballot(hmax >= 0);
    if (mk) {
      if (hmax >= 0) {
        typedef __attribute__((address_space(3))) f32x4 lds_f32x4w;
        const unsigned slot = wcnt + __builtin_amdgcn_mbcnt_hi((unsigned)(mk >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mk, 0u));
        const unsigned ra = rec0 + slot * REC_BYTES;
#pragma unroll
        for (int m = 0; m < 4; ++m) *reinterpret_cast<lds_f32x4w*>(ra + 16 * m) = PV[m];
        *reinterpret_cast<lds_f32x4w*>(ra + 64) = (f32x4){PV[4][0], __uint_as_float(pixq_p), 0.f, 0.f};
      }
      wcnt += (unsigned)__builtin_popcountll(mk);
    }
  };

  // ---- one tile: its 45 MFMAs into ACC from XC, the ops of the finished tile PV and of the next tile's XN beside them -----
  auto tile_step = [&](f32x4(&ACC)[MT], f32x4(&PV)[MT], f32x4(&XC)[G], f32x4(&XN)[G]) __attribute__((always_inline)) {
    // the next tile opens a new band: its buffer must have landed, and every wave must be done with the band before
    // (whose buffer the fetch after this one overwrites)
    dma_step();
    if (nt_ == 0 && nu > 0 && nu < nunits) {
      dma_store();   // (a band of as many rounds as tiles: its last piece)
      __syncthreads();  // (with its lgkmcnt(0): this wave's pieces are in LDS, its reads of the old band have returned)
      if (nu + 1 < nunits) fetch_begin(nu + 1);
    }
    // thresholds of the finished tile's face (first use: stage T, forty slots from here)
    if (face_p != face_t) {
      face_t = face_p;
      if (face_p >= 0) {
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          float t[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int cls = m < 4 ? 16 * m + 4 * q + e : (e == 0 ? 64 + q : 68);
            t[e] = cls < 68 ? fmaxf(a.tau[(size_t)face_p * 68 + cls], 1.17549435e-38f) : 3.402823466e38f;
          }
          tq[m] = make_float4(t[0], t[1], t[2], t[3]);
        }
      }
    }
    static_for<Ops::NSLOT>([&](auto ic) __attribute__((always_inline)) {
      constexpr int I = decltype(ic)::value, g = I / MT, m = I % MT;
      // (inline assembly: the weight operand is named as an accumulation register -- left to hipcc, the accumulators and
      //  the X fragments went there and came back through ~400 v_accvgpr copies per tile; a volatile asm also keeps its
      //  place in the stream.  Hazards the compiler cannot see: a result is read 5 MFMAs later at the earliest (as
      //  srcC), by the VALU a whole tile later.)
      if constexpr (!(FLM_WREG_ABLATE & 2)) {
        if constexpr (g == 0)
          asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=&v"(ACC[m]) : "a"(W[g][m]), "v"(XC[g]));
        else
          asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(ACC[m]) : "a"(W[g][m]), "v"(XC[g]));
      }
      static_for<kSched.first[I + 1] - kSched.first[I]>([&](auto jc) __attribute__((always_inline)) {
        constexpr int K = kSched.first[I] + decltype(jc)::value;
        if constexpr (!(FLM_WREG_ABLATE & 1) || K < Ops::A0) epi_op(std::integral_constant<int, K>{}, PV, XN);
      });
      __builtin_amdgcn_sched_barrier(0);
    });
    rec_store(PV);
    // the finished tile was the last of its face, or fewer than a tile's worth of slots are left: flush
    if (wcnt > (unsigned)(REC_CAP - 64) || (face_c != face_p && wcnt)) rec_flush(face_p);
    // advance: finished <- multiplied <- next; next tile of the stream
    pixq_p = pixq_c; pixq_c = pixq_n;
    face_p = face_c; face_c = face_n;
    if (nu < nunits && ++nt_ == ntiles_nu) {
      nt_ = 0;
      ++nu;
      if (++nb == a.nband) { nb = 0; ++nf; }
      ntiles_nu = band_tiles(nb);
    }
  };

  // The stream: step k prefetches tile k, multiplies tile k-1, finishes tile k-2 (tiles < 0 and past the end are empty):
  // all tiles + 2 steps, rounded up to the two register sets
  int total = 0;
  for (int b = 0; b < a.nband; ++b) total += band_tiles(b);
  total = total * nfaces + 2;
  for (int k = 0; k < total; k += 2) {
    tile_step(accA, accB, xa, xb);
    tile_step(accB, accA, xb, xa);
  }
  rec_flush(face_t);
}
#undef FLM_PIN

static std::atomic<int> g_wreg{1};  // A/B knob "up3_wreg": 1 (default) the bf16 candidate launch of up3 runs this kernel
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
  a.dma_per_tile = 1;
  if (a.ndma > min_tiles) return 0;  // (a band's pieces arrive one per tile step)
  // workgroups: 16 phase groups x face chunks; one workgroup per CU, so about (CUs / 16) chunks
  a.chunks = c.n < 16 ? c.n : 16;
  a.xp = scratch;
  {
    const long long total = (long long)c.n * (c.hi + 2) * (c.wi + 2) * 9;
    up3_xpack_kernel<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(c.x, static_cast<uint4*>(scratch), c.n, c.hi, c.wi, c.gate);
    FLM_LAUNCH_CHECK("up3_xpack_kernel");
  }
  const size_t lds = 2 * (size_t)a.band_stride + LIST_BYTES;
  static FuncAttrOnce attr;
  FLM_FUNC_ATTR_ONCE(attr, (&up3_wreg_kernel), LDS_TOTAL);
  up3_wreg_kernel<<<8 * 16 * cdiv(a.chunks, 8), WAVES * 64, lds, st>>>(a);
  FLM_LAUNCH_CHECK("up3_wreg_kernel");
  return 1;
}

}  // namespace flm
