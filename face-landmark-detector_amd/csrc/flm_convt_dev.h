// Device-side pieces shared by the transposed-conv kernels (flm_convt.hip) and the weights-in-registers form of the
// last one (flm_up3_wreg.hip): launch arguments, the softmax helpers (same instructions = same bits in every kernel),
// compile-time loops and the LDS-DMA request.
#pragma once
#include <utility>

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
  int rpw;                    // cand8 kernel: phase rows a0 one workgroup walks with the same X fragments (divides s)
};

// candidate keys one wave can hold in LDS: 128 (fp32) or 256 (bf16, NT = 2) pixels x 68 classes pass through it;
// the fp32 kernel's 61 KiB weight ring leaves room for 512 per wave if two workgroups are to share a CU
#define kCandWaveCap (512 * NT)
constexpr int kMaxSamplePhases = 16;

__device__ __forceinline__ unsigned cand_order_bits(float v) {
  const unsigned u = __float_as_uint(v);
  return u ^ ((u >> 31) ? 0xffffffffu : 0x80000000u);
}

// The library expf for arguments t <= 0: its instruction sequence (t*log2(e) split into a rounded head and an fma'd
// tail, v_exp_f32 of the fraction, v_ldexp_f32 by the integer part) without the two range tests -- overflow cannot
// happen, and v_ldexp_f32 underflows by itself.  9 instructions instead of 14 per class and pixel; the same bits as
// expf for every argument (tools/exp_check.hip: 16.7 M arguments in [-110, 0]) except -103.97 < t < -103.28, where expf
// cuts to 0 and this returns the smallest denormal, 1.4e-45 (fp32 up3 at batch 64: 1.54 -> 1.515 ms).
__device__ __forceinline__ float exp_nonpos(float t) {
  const float ph = t * 0x1.715476p+0f;
  float pl = __builtin_fmaf(t, 0x1.715476p+0f, -ph);
  pl = __builtin_fmaf(t, 0x1.4ae0bep-26f, pl);
  const float e = __builtin_rintf(ph);
  const float a = (ph - e) + pl;
  return __builtin_ldexpf(__builtin_amdgcn_exp2f(a), (int)e);
}

// exp(t) for t <= 0 in the softmax.  fp32 path: the accurate expf above.  bf16 path: v_exp_f32 on
// t*log2(e) (about 1e-6 relative, far below the bf16 rounding the logits already carry); at 16x the MFMA
// rate the 20 accurate expf per lane per phase would cost more than the phase's matrix work.
// x: logit, mx: the pixel's maximum, nmxl = -mx * log2(e).  bf16: one fma + v_exp_f32.
template <bool BF>
__device__ __forceinline__ float softmax_exp(float x, float mx, float nmxl) {
  if constexpr (BF) return __builtin_amdgcn_exp2f(__builtin_fmaf(x, 1.44269504088896340736f, nmxl));
  else return exp_nonpos(x - mx);
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

// up3 in landmark mode, bf16, weights in registers (flm_up3_wreg.hip): 1 launched, 0 not its case, < 0 error.  scratch:
// room for the bf16 copy of x with its zero ring, n * (hi + 2) * (wi + 2) * 144 bytes
int launch_up3_wreg(hipStream_t st, const ConvTArgs& c, void* scratch, size_t scratch_bytes);

namespace cand8 {
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// LDS-DMA: buffer_load_dwordx4 ... lds writes lane l's 16 bytes to LDS address M0 + 16*l, no VGPR destination.  Inline
// assembly: through the builtin hipcc would order every later ds_read behind the pending request (vmcnt(0)).
typedef int dma_srd __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void dma_piece(dma_srd srd, unsigned lds_addr, unsigned voffset, int soffset) {
  // M0 (the LDS base of the request) is an operand the compiler sets itself ("{m0}"), so it knows the register is
  // written; the s_nop is the wait state the ISA asks for between a scalar write of M0 and a buffer_load ... lds
  asm volatile("s_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds"
               :
               : "v"(voffset), "s"(srd), "s"(soffset), "{m0}"(lds_addr)
               : "memory");
}
}  // namespace cand8

}  // namespace flm
