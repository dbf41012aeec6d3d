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
};

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
