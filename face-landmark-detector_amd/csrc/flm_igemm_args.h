// Kernel-argument block and index helpers shared by the implicit-GEMM kernels (flm_igemm.hip: 128x128 tiles,
// fp32 and bf16; flm_igemm_bf16.hip: 256-row tiles for bf16).
#pragma once
#include "flm_common.h"

namespace flm {

struct IgemmArgs {
  const void* x;       // fp32 or bf16 [n,h,w,cin]
  const void* wt;      // same type, [coutpad][K]
  const float* scale;
  const float* shift;
  void* y;             // operand type, or fp32 when out_f32
  int out_f32;
  float relu_max;      // upper clamp applied with the ReLU (6 for ReLU6, +inf otherwise)
  int n, h, w, cin;    // input grid
  int ho, wo, stride;  // output grid = ((h + 2*pad - kh) / stride + 1, ...); stride > 1 only with MMAP 0
  const void* res;     // optional residual [n,ho,wo,ldc] (operand type) added before the ReLU
  int cout, ldc;
  int kh, kw, pad;
  int M;        // n*ho*wo
  int K;        // kh*kw*cin
  int mtiles, ntiles;
  int cpt;      // 128-byte channel chunks per tap = cin/32 (fp32) or cin/64 (bf16)
  int kw_magic; // ceil(65536 / kw): tap / kw == (tap * kw_magic) >> 16 for tap < 64
  int ksplit;   // > 1: blockIdx.y owns a slice of the k-steps and stores raw partial sums to `part`
  float* part;  // [ksplit][M][ldc]
  int grp_magic;  // fp32 three-level accumulation: ceil(65536 / steps per group); group = (dense step * grp_magic) >> 16
  int gm, gn;   // flm_igemm_bf16.hip: tiles are dealt in groups of gm x gn (the 32 workgroups resident on one XCD)
  // Position-major layers (fc6) in the fp32 kernel whose tiles span several positions of the map (faces per batch < rows per tile): the
  // positions are taken in the ORDER posperm gives (byte i of the 64-byte table = the map position at place i), chosen
  // on the host so that the positions sharing a tile have nearly the same filter taps in bounds -- a tile issues the UNION
  // of its positions' taps (posmajor_fill_perm below).  posperm_on == 0: places are positions.
  unsigned long long posperm[8];
  int posperm_on;
};

// place -> map position (see IgemmArgs::posperm).  The table sits in scalar registers: eight selects and a shift.
__device__ __forceinline__ int posmajor_pos(const IgemmArgs& a, int place) {
  if (!a.posperm_on) return place;
  unsigned long long w = a.posperm[0];
#pragma unroll
  for (int k = 1; k < 8; ++k) w = (place >> 3) == k ? a.posperm[k] : w;
  return (int)((w >> ((place & 7) * 8)) & 0xffull);
}

// Taps of a kh x kw 'same' filter that touch at least one in-bounds pixel for m-tile t in
// position-major order (positions t*BM/n .. of an h x w map): bit ky*kw+kx.
__device__ __forceinline__ unsigned long long posmajor_tapmask(int t, int M, int n, int h, int w, int kh, int kw,
                                                               int pad, int bm = 128) {
  const int m_lo = t * bm, m_hi = (m_lo + bm < M ? m_lo + bm : M) - 1;
  unsigned long long mask = 0;
  for (int p = m_lo / n; p <= m_hi / n; ++p) {
    const int y = p / w, x = p % w;
    for (int ky = 0; ky < kh; ++ky) {
      if ((unsigned)(y + ky - pad) >= (unsigned)h) continue;
      for (int kx = 0; kx < kw; ++kx)
        if ((unsigned)(x + kx - pad) < (unsigned)w) mask |= 1ull << (ky * kw + kx);
    }
  }
  return mask;
}

// Taps that see an in-bounds pixel from position p = y*w + x alone: rows ky of kx_lo..kx_hi.
__device__ __forceinline__ unsigned long long posmajor_posmask(int p, int h, int w, int kh, int kw, int pad) {
  const int y = p / w, x = p - y * w;
  const int kx_lo = pad - x > 0 ? pad - x : 0, kx_hi = pad + w - 1 - x < kw - 1 ? pad + w - 1 - x : kw - 1;
  const int ky_lo = pad - y > 0 ? pad - y : 0, ky_hi = pad + h - 1 - y < kh - 1 ? pad + h - 1 - y : kh - 1;
  if (kx_hi < kx_lo) return 0ull;
  const unsigned long long row = ((1ull << (kx_hi - kx_lo + 1)) - 1ull) << kx_lo;
  unsigned long long mask = 0ull;
  for (int ky = ky_lo; ky <= ky_hi; ++ky) mask |= row << (ky * kw);
  return mask;
}

// Output row (NHWC pixel index) of position-major GEMM row m = pos * n + face: face * (h*w) + pos.  The epilogues call it
// for the 16 rows a lane holds of a 32-row accumulator tile -- offsets 0..27 from the lane's first row m_first, whose
// (face, position) = (nn0, pos0) is divided ONCE: with n >= 32 a row wraps into the next position at most once.  (An
// integer division per stored element, 128 per lane and tile, was a fifth of the bf16 fc6 launch.)
__device__ __forceinline__ size_t posmajor_orow(int m, int m_first, int nn0, int pos0, int n, int hw) {
  if (n >= 32) {
    int nn = nn0 + (m - m_first), pos = pos0;
    if (nn >= n) {
      nn -= n;
      ++pos;
    }
    return (size_t)nn * hw + pos;
  }
  return (size_t)(m % n) * hw + m / n;
}
// The same under IgemmArgs::posperm (the fp32 kernel): pos0, pos1 = the MAP positions of m_first's place and of the next
// place (posmajor_pos); below 32 faces places are positions (posmajor_fill_perm leaves those batches alone).
__device__ __forceinline__ size_t posmajor_orow(int m, int m_first, int nn0, int pos0, int pos1, int n, int hw) {
  if (n >= 32) {
    int nn = nn0 + (m - m_first), pos = pos0;
    if (nn >= n) {
      nn -= n;
      pos = pos1;
    }
    return (size_t)nn * hw + pos;
  }
  return (size_t)(m % n) * hw + m / n;
}

// Host side of IgemmArgs::posperm for a launch with rows-per-tile bm.  A tile of a position-major layer issues the UNION
// of the taps its positions need (zero rows for the positions that do not); when n faces < bm rows, g = bm / n positions
// share a tile.  Map order pairs (y, x) with (y, x + 1): at fc6 (7x7 'same' on 8x8) and 64 faces the union is 48 taps per
// 44 used (0.917).  The order chosen here starts from the positions sorted by the RANGES of taps they have in bounds --
// equal ranges (the middle rows / columns) become neighbours --, from 2 x 2 blocks offset by one, or from the tap counts,
// each improved by swapping two positions of different groups while that lowers sum(|union| x rows); the best of the
// three is kept: 0.917 -> 0.955 at g = 2 (the optimum of a perfect matching), 0.786 -> 0.874 at g = 4.  (The bf16 256-row kernel keeps map order: it has no register to spare for the lookups, and its
// headline batch, 512 faces, has one position per tile anyway.)  Only the ORDER in which tiles take positions changes: every output is the same sum in the
// same order (the skipped and the not-skipped taps of a row outside its own set are zero products either way), the same
// bits (tests/test_gpu_forward.py).  Maps of more than 64 positions, n >= bm, n < 32 (split-K territory) or n not dividing bm: left alone.
void igemm_posperm_enable(int on);
int igemm_posperm_enabled();
struct PospermEntry {
  int h, w, kh, kw, pad, g, on;
  unsigned long long table[8];
};
// (flm_igemm.hip: a small cache behind a mutex -- the search below runs once per layer geometry and group size)
bool posperm_cache_get(PospermEntry& e);
void posperm_cache_put(const PospermEntry& e);

inline void posmajor_fill_perm(IgemmArgs& a, int bm) {
  a.posperm_on = 0;
  for (int k = 0; k < 8; ++k) a.posperm[k] = 0ull;
  const int P = a.h * a.w;
  if (!igemm_posperm_enabled() || P > 64 || P < 2 || a.n < 32 || a.n >= bm || bm % a.n != 0 || a.kh * a.kw > 64) return;
  const int g = bm / a.n;
  PospermEntry ent = {a.h, a.w, a.kh, a.kw, a.pad, g, 0, {0, 0, 0, 0, 0, 0, 0, 0}};
  if (!posperm_cache_get(ent)) {
    unsigned long long mask[64];
    long long key[3][64];
    for (int p = 0; p < P; ++p) {
      const int y = p / a.w, x = p % a.w;
      const int kx_lo = a.pad - x > 0 ? a.pad - x : 0, kx_hi = a.pad + a.w - 1 - x < a.kw - 1 ? a.pad + a.w - 1 - x : a.kw - 1;
      const int ky_lo = a.pad - y > 0 ? a.pad - y : 0, ky_hi = a.pad + a.h - 1 - y < a.kh - 1 ? a.pad + a.h - 1 - y : a.kh - 1;
      unsigned long long m = 0ull;
      for (int ky = ky_lo; ky <= ky_hi; ++ky)
        for (int kx = kx_lo; kx <= kx_hi; ++kx) m |= 1ull << (ky * a.kw + kx);
      mask[p] = m;
      // three starting orders: by the tap ranges in bounds; by 2 x 2 blocks offset by one (the middle rows / columns, whose
      // ranges are equal, share a block); by tap count, most first
      key[0][p] = ((((long long)ky_lo * 64 + ky_hi) * 64 + kx_lo) * 64 + kx_hi) * 64 + p;
      key[1][p] = ((long long)((y + 1) / 2) * 64 + (x + 1) / 2) * 64 + p;
      key[2][p] = (long long)(64 - __builtin_popcountll(m)) * (1ll << 40) + key[0][p];
    }
    long long best_cost = -1;
    int best[64];
    for (int start = 0; start < 3; ++start) {
      int order[64];
      for (int p = 0; p < P; ++p) order[p] = p;
      for (int i = 1; i < P; ++i)  // insertion sort by key
        for (int j = i; j > 0 && key[start][order[j]] < key[start][order[j - 1]]; --j) {
          const int t = order[j]; order[j] = order[j - 1]; order[j - 1] = t;
        }
      auto group_cost = [&](int gi) {
        unsigned long long u = 0ull;
        int rows = 0;
        for (int i = gi * g; i < (gi + 1) * g && i < P; ++i) { u |= mask[order[i]]; ++rows; }
        return (long long)__builtin_popcountll(u) * rows;
      };
      for (int sweep = 0; sweep < 64; ++sweep) {  // swap two positions of different groups while that lowers the cost
        bool better = false;
        for (int i = 0; i < P; ++i)
          for (int j = i + 1; j < P; ++j) {
            if (i / g == j / g) continue;
            const long long before = group_cost(i / g) + group_cost(j / g);
            int t = order[i]; order[i] = order[j]; order[j] = t;
            if (group_cost(i / g) + group_cost(j / g) < before) { better = true; continue; }
            t = order[i]; order[i] = order[j]; order[j] = t;
          }
        if (!better) break;
      }
      long long cost = 0;
      for (int gi = 0; gi * g < P; ++gi) cost += group_cost(gi);
      if (best_cost < 0 || cost < best_cost) {
        best_cost = cost;
        for (int p = 0; p < P; ++p) best[p] = order[p];
      }
    }
    bool identity = true;
    for (int i = 0; i < P; ++i) {
      identity = identity && best[i] == i;
      ent.table[i >> 3] |= (unsigned long long)best[i] << ((i & 7) * 8);
    }
    ent.on = identity ? 0 : 1;
    posperm_cache_put(ent);
  }
  a.posperm_on = ent.on;
  for (int k = 0; k < 8; ++k) a.posperm[k] = ent.table[k];
}

__device__ __forceinline__ unsigned short f2bf(float v) { return __builtin_bit_cast(unsigned short, (__bf16)v); }

// 4x4 transpose of bf16 values over a lane quad, for the epilogues of the 32x32 MFMA tiles: a lane holds four rows of
// ONE column (2-byte stores, 64 B per row and instruction); afterwards lane q of the quad holds row q of the quad's
// FOUR columns, 8 bytes it stores at once.  Two exchange steps on packed pairs: 16-bit halves with lane^1 (v_perm),
// whole dwords with lane^2.  Every lane of the wave must be active.
__device__ __forceinline__ uint2 quad_transpose_bf16(float r0, float r1, float r2, float r3, int q) {
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  const unsigned p01 = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){r0, r1}, bf16x2_t));
  const unsigned p23 = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){r2, r3}, bf16x2_t));
  const unsigned o01 = (unsigned)__builtin_amdgcn_mov_dpp((int)p01, 0xB1, 0xf, 0xf, true);  // quad_perm [1,0,3,2]
  const unsigned o23 = (unsigned)__builtin_amdgcn_mov_dpp((int)p23, 0xB1, 0xf, 0xf, true);
  const unsigned sel = (q & 1) ? 0x03020706u : 0x05040100u;  // odd: (other.hi, own.hi); even: (own.lo, other.lo)
  const unsigned t01 = __builtin_amdgcn_perm(o01, p01, sel);  // row q&1, the lane pair's two columns
  const unsigned t23 = __builtin_amdgcn_perm(o23, p23, sel);  // row 2 + (q&1)
  const bool low = q < 2;
  const unsigned recv = (unsigned)__builtin_amdgcn_mov_dpp((int)(low ? t23 : t01), 0x4E, 0xf, 0xf, true);  // [2,3,0,1]
  return make_uint2(low ? t01 : recv, low ? recv : t23);
}

// Operands are fetched with raw buffer loads (buffer_load_dwordx4 v, voffset, srd, soffset offen):
//   * the per-row byte offset of the centre pixel is a constant VGPR; the filter tap and channel chunk of the
//     k-step are wave-uniform and ride in the scalar offset, so a load costs no vector address arithmetic;
//   * rows of the im2col operand that fall on zero padding (or past M) get the offset 0x80000000, beyond
//     num_records: the buffer unit returns zeros, so no validity flag travels from the load to the LDS write
//     and the write needs no select.
// The resource base is moved back by the largest negative tap displacement so that the scalar offset is
// never negative.
constexpr unsigned kOobOffset = 0x80000000u;
constexpr int kSrdFlags = 0x00020000;  // raw dword buffer, gfx9 DATA_FORMAT field

// LDS-DMA: buffer_load_dwordx4 ... lds writes lane l's 16 bytes to LDS address M0 + 16*l, no VGPR destination.
// Inline assembly: through the builtin hipcc orders every later ds_read behind the pending request (vmcnt(0)).
typedef int dma_srd __attribute__((ext_vector_type(4)));
__device__ __forceinline__ dma_srd dma_make_srd(const void* base) {
  const unsigned long long b = reinterpret_cast<unsigned long long>(base);
  return (dma_srd){(int)(unsigned)b, (int)(unsigned)((b >> 32) & 0xffffu), 0x7fffffff, kSrdFlags};
}
__device__ __forceinline__ void dma_load16(dma_srd srd, unsigned lds_addr, unsigned voffset, int soffset) {
  // M0 (the LDS base of the request) is an operand the compiler sets itself ("{m0}"), so it knows the register is
  // written; the s_nop is the wait state the ISA asks for between a scalar write of M0 and a buffer_load ... lds
  asm volatile("s_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds"
               :
               : "v"(voffset), "s"(srd), "s"(soffset), "{m0}"(lds_addr)
               : "memory");
}

// bf16 layers whose tile grid fills the chip with 256-row tiles (flm_igemm_bf16.hip); returns 1 when it launched,
// 0 when the shape is left to the 128x128 kernel, < 0 on error.
int launch_igemm_bf16_big(hipStream_t s, const IgemmArgs& a, int relu, int pool, int posmajor, int coutpad);
// bf16 1x1 classifiers with 256 input channels and at most 80 columns, fp32 out (flm_score1x1.hip); same return convention.
int launch_score1x1_bf16(hipStream_t s, const IgemmArgs& a, int relu, int pool, int posmajor, int coutpad);
// bf16 3x3 'same' layers with 64 input channels (flm_conv3_halo.hip); same return convention.
int launch_conv3_halo_bf16(hipStream_t s, const IgemmArgs& a, int relu, int pool, int posmajor);

}  // namespace flm
