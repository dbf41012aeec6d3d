// 3x3 'same' convolution with 64 input channels, bf16 operands (v_mfma_f32_32x32x16_bf16, fp32 accumulate),
// everything but the input halo resident on chip: the vanilla encoder's enc2 (networks/fcn.py:33-38,
// ZeroPadding2D(1) + Conv2D(128, 3x3) + BatchNormalization + ReLU + MaxPooling2D) and VGG's 64-channel layers.
//
// Why not the implicit GEMM of flm_igemm*.hip: with K = 9 taps x 64 channels the GEMM tiles re-fetch every input
// pixel nine times and the same 147 KB of weights once per tile -- 32 KiB from L2 per 1 M MACs, which is what
// bounds enc2 at a third of the bf16 rate.  Here
//   * a workgroup owns 64 output channels for its whole life: their 9 x 64 x 64 weights (72 KiB) are loaded into
//     LDS once, in the XOR-swizzled row image the fragment reads want;
//   * it walks over 16x16-pixel output tiles; per tile only the 18x18x64 input halo (40.5 KiB) comes from
//     memory, prefetched into registers under the previous tile's MFMAs (raw buffer loads: pixels outside the
//     image read as zero) and written to the second halo buffer at the tile's end -- one barrier per tile;
//   * the nine taps are nine shifted reads of the same halo: fragment address = halo pixel (y+ky, x+kx), 16-byte
//     chunk XOR-swizzled by ((hx >> 1) + 4*(hy & 1)) & 7, which keeps the 16 pixels (2 rows x 8 columns) a
//     ds_read_b128 lane group touches on 16 distinct bank quads for every tap shift;
//   * pixel order inside the tile is 2x2 quads (as flm_igemm.hip MMAP 1): the max-pool is an in-lane max;
//     weight rows are permuted so that a lane owns channels 2*lr and 2*lr+1: one dword store per pixel pair.
// 4 waves (one per SIMD, up to 512 registers), each 64 pixels x 64 channels = 2x2 MFMA tiles; fragments are read
// two 16-deep slices ahead of their MFMAs.  k order = tap-major, then 16-deep slices: the same as the implicit
// GEMM's, so results are bit-identical to it.
#include "flm_igemm_args.h"

namespace flm {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int HT = 16;                        // output tile edge
constexpr int HH = HT + 2;                    // halo edge
constexpr int HALO_BYTES = HH * HH * 128;     // 41,472
constexpr int WSLICE_BYTES = 9 * 64 * 128;    // 73,728
constexpr int HALO_CHUNKS = HH * HH * 8;      // 16-byte chunks per halo
constexpr int HLD = (HALO_CHUNKS + 255) / 256;  // staging loads per thread (11)

__device__ __forceinline__ int halo_off(int hy, int hx, int chunk) {
  return ((hy * HH + hx) << 7) + ((chunk ^ (((hx >> 1) + 4 * (hy & 1)) & 7)) << 4);
}

typedef float f32x4_h __attribute__((ext_vector_type(4)));

// M16: v_mfma_f32_16x16x32_bf16 -- the wave's 64 pixels x 64 channels are 4 x 4 tiles of 16 x 16, a tap two 32-deep slices
// of 16 MFMAs (18 slices per tile instead of 36 of four 32x32x16); same halo and weight images, same fragment bytes, same
// 128 accumulator registers, and the same bits (the shape sums a k-run of 32 as two 16-runs; flm_igemm_bf16.hip).
template <bool POOL, bool RELU, bool M16>
__global__ __launch_bounds__(256, 1) void conv3_halo_bf16_kernel(IgemmArgs a, int wg_per_slice) {
  constexpr int NI = M16 ? 4 : 2;      // pixel tiles (and channel tiles) per wave
  constexpr int NSL = M16 ? 18 : 36;   // slices per output tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* wlds = smem;                       // [9][64][128 B]
  char* halo0 = smem + WSLICE_BYTES;       // [2][HH*HH][128 B]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = M16 ? (lane & 15) : (lane & 31), lh = M16 ? (lane >> 4) : (lane >> 5);
  const int slice = blockIdx.x / wg_per_slice, wg = blockIdx.x % wg_per_slice;
  const int n0 = slice * 64;
  const int tx_n = a.w / HT, ty_n = a.h / HT;
  const int tiles = a.n * ty_n * tx_n;

  // ---- weights of this slice -> LDS, row 32j + l <- channel n0 + 2l + j ------------------------------------
  {
    const char* wsrc = reinterpret_cast<const char*>(a.wt);
    for (int e = tid; e < 9 * 64 * 8; e += 256) {
      const int c8 = e & 7, row = (e >> 3) & 63, tap = e >> 9;
      const int ch = n0 + 2 * (row & 31) + (row >> 5);
      const float4 v = *reinterpret_cast<const float4*>(wsrc + ((size_t)ch * a.K + tap * 64) * 2 + c8 * 16);
      *reinterpret_cast<float4*>(wlds + tap * 8192 + row * 128 + ((c8 ^ ((row >> 1) & 7)) << 4)) = v;
    }
  }
  // a lane stores channel pairs (2L, 2L + 1): L = lr (32x32x16); M16: L = lr and L = 16 + lr (channel tiles jb and jb + 2)
  float sc0[2], sc1[2], sh0[2], sh1[2];
#pragma unroll
  for (int jb = 0; jb < 2; ++jb) {
    const int L = (M16 ? 16 * jb : 0) + lr;
    sc0[jb] = a.scale[n0 + 2 * L]; sc1[jb] = a.scale[n0 + 2 * L + 1];
    sh0[jb] = a.shift[n0 + 2 * L]; sh1[jb] = a.shift[n0 + 2 * L + 1];
  }

  // ---- halo staging role: chunk e = tid + 256*k of the 18x18x8 chunk grid -----------------------------------
  const __amdgpu_buffer_rsrc_t xsrd =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, 0x7fffffff, 0x00020000);
  int st_off[HLD];   // LDS byte offset inside a halo buffer (-1: no chunk)
  int st_rel[HLD];   // (hy << 16) | hx of the chunk's pixel, chunk index in bits 28..30
#pragma unroll
  for (int k = 0; k < HLD; ++k) {
    const int e = tid + 256 * k;
    const bool has = e < HALO_CHUNKS;
    const int pix = has ? (e >> 3) : 0, c8 = e & 7;
    const int hy = pix / HH, hx = pix % HH;
    st_off[k] = has ? halo_off(hy, hx, c8) : -1;
    st_rel[k] = (c8 << 28) | (hy << 16) | hx;
  }
  float4 pf[HLD];
  // one chunk of the halo of tile (TX, TY, IMG): pixels outside the image (or past the last tile) read as zero
#define FLM_HALO_LOAD1(K, TX, TY, IMG, TVALID)                                                    \
  {                                                                                               \
    const int iy = (TY) * HT - 1 + ((st_rel[K] >> 16) & 0xfff), ix = (TX) * HT - 1 + (st_rel[K] & 0xffff); \
    const bool ok = st_off[K] >= 0 && (TVALID) && (unsigned)iy < (unsigned)a.h && (unsigned)ix < (unsigned)a.w; \
    const unsigned off = (unsigned)((((IMG) * a.h + iy) * a.w + ix) * 128 + ((st_rel[K] >> 28) & 7) * 16); \
    pf[K] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xsrd, ok ? off : 0x80000000u, 0, 0)); \
  }
#define FLM_HALO_STORE1(K, BUF) \
  if (st_off[K] >= 0) *reinterpret_cast<float4*>(halo0 + (BUF) * HALO_BYTES + st_off[K]) = pf[K];

  // ---- fragment addressing ------------------------------------------------------------------------------
  // wave w: rows 64w + 32i + lr -> quad q = 16w + 8i + (lr >> 2), (dy, dx) = ((lr >> 1) & 1, lr & 1)
  int py[NI], px[NI];  // pixel inside the tile (M16: rows 64w + 16i + lr -> quad 16w + 4i + (lr >> 2))
#pragma unroll
  for (int i = 0; i < NI; ++i) {
    const int q = 16 * wave + (M16 ? 4 : 8) * i + (lr >> 2);
    py[i] = 2 * (q >> 3) + ((lr >> 1) & 1);
    px[i] = 2 * (q & 7) + (lr & 1);
  }
  const int swx = (lr >> 1) & 7;
  int fcb[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) fcb[s] = lr * 128 + ((((M16 ? 4 * (s & 1) : 2 * s) + lh) ^ swx) << 4);
  unsigned int* yw = reinterpret_cast<unsigned int*>(a.y);

  // One (i, g) piece of the epilogue of the tile whose sums sit in accumulator set Q: BN / bias, ReLU, (pool),
  // channel pairs as dwords.  Rows 8g + 4lh .. +3 of the 32-row tile = quad q = 16*wave + 8*i + 2*g + lh.
#define FLM_EPI_PIECE(Q, I, GG, TX, TY, IMG)                                                      \
  {                                                                                               \
    /* 32x32x16: rows 8g + 4lh .. +3 of pixel tile I = quad 16w + 8I + 2g + lh, registers 4g .. 4g+3, channel tiles 0 / 1 */ \
    /* M16: (I, GG) = (pixel tile, jb): rows 4lh .. +3 of tile I = quad 16w + 4I + lh, registers 0..3, channel tiles jb / jb+2 */ \
    const int q_ = M16 ? 16 * wave + 4 * (I) + lh : 16 * wave + 8 * (I) + 2 * (GG) + lh;          \
    const int qy_ = q_ >> 3, qx_ = q_ & 7;                                                        \
    const int jb_ = M16 ? (GG) : 0, r0_ = M16 ? 0 : 4 * (GG), ja_ = M16 ? (GG) : 0, jc_ = M16 ? (GG) + 2 : 1;     \
    const int L_ = (M16 ? 16 * jb_ : 0) + lr;                                                     \
    if (POOL) {                                                                                   \
      float v0 = -3.402823466e38f, v1 = -3.402823466e38f;                                         \
      _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                             \
        float u0 = fmaf(acc[Q][I][ja_][r0_ + e], sc0[jb_], sh0[jb_]), u1 = fmaf(acc[Q][I][jc_][r0_ + e], sc1[jb_], sh1[jb_]); \
        if (RELU) { u0 = fminf(fmaxf(u0, 0.f), a.relu_max); u1 = fminf(fmaxf(u1, 0.f), a.relu_max); } \
        v0 = fmaxf(v0, u0);                                                                       \
        v1 = fmaxf(v1, u1);                                                                       \
      }                                                                                           \
      const size_t opix = ((size_t)(IMG) * (a.h >> 1) + (TY) * (HT / 2) + qy_) * (a.w >> 1) + (TX) * (HT / 2) + qx_; \
      yw[(opix * a.ldc + n0) / 2 + L_] = (unsigned)f2bf(v0) | ((unsigned)f2bf(v1) << 16);        \
    } else {                                                                                      \
      _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                             \
        float u0 = fmaf(acc[Q][I][ja_][r0_ + e], sc0[jb_], sh0[jb_]), u1 = fmaf(acc[Q][I][jc_][r0_ + e], sc1[jb_], sh1[jb_]); \
        if (RELU) { u0 = fminf(fmaxf(u0, 0.f), a.relu_max); u1 = fminf(fmaxf(u1, 0.f), a.relu_max); } \
        const int y_ = (TY) * HT + 2 * qy_ + (e >> 1), x_ = (TX) * HT + 2 * qx_ + (e & 1);        \
        const size_t opix = ((size_t)(IMG) * a.h + y_) * a.w + x_;                                \
        yw[(opix * a.ldc + n0) / 2 + L_] = (unsigned)f2bf(u0) | ((unsigned)f2bf(u1) << 16);       \
      }                                                                                           \
    }                                                                                             \
  }

  // slice sl = 4*tap + s; fragments of slice sl live in set sl % 3 and are fetched two slices ahead
#define FLM_FRAGS(SL, SET)                                                                        \
  {                                                                                               \
    const int tap_ = M16 ? (SL) >> 1 : (SL) >> 2, s_ = M16 ? (SL) & 1 : (SL) & 3;                 \
    const int ky_ = tap_ / 3, kx_ = tap_ % 3;                                                     \
    _Pragma("unroll") for (int i = 0; i < NI; ++i)                                                \
      af[SET][i] = *reinterpret_cast<const float4*>(hb + halo_off(py[i] + ky_, px[i] + kx_, (M16 ? 4 : 2) * s_ + lh)); \
    _Pragma("unroll") for (int j = 0; j < NI; ++j)                                                \
      bfr[SET][j] = *reinterpret_cast<const float4*>(wlds + tap_ * 8192 + j * (M16 ? 2048 : 4096) + fcb[s_]); \
  }

  // One tile: 36 slices of 4 MFMAs into accumulator set P.  With one wave per SIMD nothing else hides the scalar
  // and vector work around the MFMAs, so it is dealt into the slices: the epilogue of the PREVIOUS tile (set Q,
  // EPI = 1) in slices 0-7, the loads of the NEXT tile's halo in slices 8-18, their LDS writes in slices 25-35.
#define FLM_TILE(P, Q, EPI)                                                                       \
  {                                                                                               \
    const char* hb = halo0 + buf * HALO_BYTES;                                                    \
    const int tn_ = t + wg_per_slice;                                                             \
    const int ntx_ = tn_ % tx_n, nty_ = (tn_ / tx_n) % ty_n, nimg_ = tn_ / (tx_n * ty_n);         \
    const bool nvalid_ = tn_ < tiles;                                                             \
    _Pragma("unroll") for (int i = 0; i < NI; ++i) _Pragma("unroll") for (int j = 0; j < NI; ++j)   \
      _Pragma("unroll") for (int r = 0; r < (M16 ? 4 : 16); ++r) acc[P][i][j][r] = 0.f;          \
    /* fragment sets: three, fetched two slices ahead (128-cycle slices); M16: two, one 256-cycle slice ahead */ \
    constexpr int NSET = M16 ? 2 : 3;                                                             \
    float4 af[NSET][NI], bfr[NSET][NI];                                                           \
    FLM_FRAGS(0, 0)                                                                               \
    if constexpr (!M16) FLM_FRAGS(1, 1)                                                           \
    _Pragma("unroll") for (int sl = 0; sl < NSL; ++sl) {                                          \
      if (sl + NSET - 1 < NSL) FLM_FRAGS(sl + NSET - 1, (sl + NSET - 1) % NSET)                   \
      _Pragma("unroll") for (int i = 0; i < NI; ++i) _Pragma("unroll") for (int j = 0; j < NI; ++j) { \
        if constexpr (M16)                                                                        \
          acc[P][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[sl % NSET][i]), \
                                                                 __builtin_bit_cast(bf16x8, bfr[sl % NSET][j]), acc[P][i][j], 0, 0, 0); \
        else                                                                                      \
          acc[P][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[sl % NSET][i]), \
                                                                 __builtin_bit_cast(bf16x8, bfr[sl % NSET][j]), acc[P][i][j], 0, 0, 0); \
      }                                                                                           \
      if constexpr (M16) {                                                                        \
        /* 18 slices of 256 cycles: epilogue pieces (pixel tile, jb) in 0-7, two halo loads per slice in 0-5, their LDS */ \
        /* writes two per slice in 12-17 */                                                       \
        if (EPI && sl < 8) FLM_EPI_PIECE(Q, (sl >> 1), (sl & 1), ptx, pty, pimg)                  \
        if (sl < 6) {                                                                             \
          if (2 * sl < HLD) FLM_HALO_LOAD1((2 * sl < HLD ? 2 * sl : 0), ntx_, nty_, nimg_, nvalid_) \
          if (2 * sl + 1 < HLD) FLM_HALO_LOAD1((2 * sl + 1 < HLD ? 2 * sl + 1 : 0), ntx_, nty_, nimg_, nvalid_) \
        }                                                                                         \
        if (sl >= 12) {                                                                           \
          if (2 * (sl - 12) < HLD) { FLM_HALO_STORE1((2 * (sl - 12) < HLD ? 2 * (sl - 12) : 0), buf ^ 1) } \
          if (2 * (sl - 12) + 1 < HLD) { FLM_HALO_STORE1((2 * (sl - 12) + 1 < HLD ? 2 * (sl - 12) + 1 : 0), buf ^ 1) } \
        }                                                                                         \
      } else {                                                                                    \
        if (EPI && sl < 8) FLM_EPI_PIECE(Q, (sl >> 2), (sl & 3), ptx, pty, pimg)                  \
        if (sl >= 8 && sl < 8 + HLD) FLM_HALO_LOAD1((sl - 8 < HLD ? sl - 8 : 0), ntx_, nty_, nimg_, nvalid_) \
        if (sl >= 36 - HLD) { FLM_HALO_STORE1((sl - (36 - HLD)), buf ^ 1) }                       \
      }                                                                                           \
      __builtin_amdgcn_sched_barrier(0);                                                          \
    }                                                                                             \
    ptx = t % tx_n; pty = (t / tx_n) % ty_n; pimg = t / (tx_n * ty_n);                            \
    __syncthreads();                                                                              \
    buf ^= 1;                                                                                     \
    t += wg_per_slice;                                                                            \
  }
#define FLM_FINAL_EPI(Q)                                                                          \
  _Pragma("unroll") for (int i = 0; i < NI; ++i) _Pragma("unroll") for (int g = 0; g < (M16 ? 2 : 4); ++g) \
    FLM_EPI_PIECE(Q, i, g, ptx, pty, pimg)

  int t = wg;
  if (t >= tiles) return;
  {
    const int tx0 = t % tx_n, ty0 = (t / tx_n) % ty_n, img0 = t / (tx_n * ty_n);
#pragma unroll
    for (int k = 0; k < HLD; ++k) FLM_HALO_LOAD1(k, tx0, ty0, img0, true)
#pragma unroll
    for (int k = 0; k < HLD; ++k) { FLM_HALO_STORE1(k, 0) }
  }
  __syncthreads();

  typedef float accv_h __attribute__((ext_vector_type(M16 ? 4 : 16)));
  accv_h acc[2][NI][NI];
  int buf = 0;
  int ptx = 0, pty = 0, pimg = 0;  // tile whose sums wait in the other accumulator set
  FLM_TILE(0, 1, 0)
  for (;;) {
    if (t >= tiles) {
      FLM_FINAL_EPI(0)
      break;
    }
    FLM_TILE(1, 0, 1)
    if (t >= tiles) {
      FLM_FINAL_EPI(1)
      break;
    }
    FLM_TILE(0, 1, 1)
  }
#undef FLM_HALO_LOAD1
#undef FLM_HALO_STORE1
#undef FLM_EPI_PIECE
#undef FLM_FRAGS
#undef FLM_TILE
#undef FLM_FINAL_EPI
}

static std::atomic<int> g_halo_enable{1};  // A/B knob: never changes results or layouts
void conv3_halo_enable(int on) { g_halo_enable.store(on, std::memory_order_relaxed); }
static std::atomic<int> g_halo_m16{1};     // 16x16x32 MFMAs (1) or 32x32x16 (0); same bits
void conv3_halo_m16(int on) { g_halo_m16.store(on, std::memory_order_relaxed); }

template <bool POOL, bool RELU, bool M16>
static int launch_halo_t(hipStream_t s, const IgemmArgs& a, int slices) {
  constexpr size_t lds = WSLICE_BYTES + 2 * HALO_BYTES;
  static FuncAttrOnce attr;
  FLM_FUNC_ATTR_ONCE(attr, (&conv3_halo_bf16_kernel<POOL, RELU, M16>), lds);
  const int tiles = a.n * (a.h / HT) * (a.w / HT);
  int wg_per_slice = 256 / slices;  // one workgroup per CU
  if (wg_per_slice > tiles) wg_per_slice = tiles;
  if (wg_per_slice < 1) wg_per_slice = 1;
  conv3_halo_bf16_kernel<POOL, RELU, M16><<<slices * wg_per_slice, 256, lds, s>>>(a, wg_per_slice);
  FLM_LAUNCH_CHECK("conv3_halo_bf16_kernel");
  return 1;
}

// 1: launched; 0: shape left to the implicit-GEMM kernels
int launch_conv3_halo_bf16(hipStream_t s, const IgemmArgs& a, int relu, int pool, int posmajor) {
  if (!g_halo_enable || a.kh != 3 || a.kw != 3 || a.pad != 1 || a.stride != 1 || a.cin != 64 || a.res || posmajor ||
      a.ksplit > 1 || a.out_f32 || (a.cout % 64) || (a.h % HT) || (a.w % HT) || (a.ldc & 1) ||
      (long long)a.n * a.h * a.w * 128 >= (1ll << 31))
    return 0;
  const int slices = a.cout / 64;
  if (slices > 256) return 0;
  // few tiles: the implicit GEMM fills the chip better (g_halo_enable == 2 forces this kernel: tests)
  if (g_halo_enable != 2 && (long long)a.n * (a.h / HT) * (a.w / HT) * slices < 1024) return 0;
  if (g_halo_m16.load(std::memory_order_relaxed)) {
    if (pool) return relu ? launch_halo_t<true, true, true>(s, a, slices) : launch_halo_t<true, false, true>(s, a, slices);
    return relu ? launch_halo_t<false, true, true>(s, a, slices) : launch_halo_t<false, false, true>(s, a, slices);
  }
  if (pool) return relu ? launch_halo_t<true, true, false>(s, a, slices) : launch_halo_t<true, false, false>(s, a, slices);
  return relu ? launch_halo_t<false, true, false>(s, a, slices) : launch_halo_t<false, false, false>(s, a, slices);
}

}  // namespace flm
