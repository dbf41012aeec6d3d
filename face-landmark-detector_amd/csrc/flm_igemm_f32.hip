// Implicit-GEMM convolution, exact fp32 on the matrix cores (v_mfma_f32_32x32x2_f32).
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
//   2  position-major (y, x, n): a tile holds one or two spatial positions of many faces, so
//      filter taps that fall outside the 8x8 map for the whole tile are skipped (fc6: 7x7 'same'
//      on 8x8 -- 38 % of its dense MACs multiply zero padding).
#include "flm_common.h"

namespace flm {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct IgemmArgs {
  const float* x;
  const float* wt;
  const float* scale;
  const float* shift;
  float* y;
  int n, h, w, cin;
  int cout, ldc;
  int kh, kw, pad;
  int M;        // n*h*w
  int K;        // kh*kw*cin
  int mtiles, ntiles;
  int cpt;      // 32-channel chunks per tap = cin/32
  int stagger;  // start delay of the odd co-resident workgroup, x64 cycles
  int dbg;      // diagnosis only (wrong results): 1 no global loads in loop, 2 no LDS writes, 4 no barrier
};

int g_igemm_debug = 0;
int g_igemm_stagger = 40;  // x64 cycles (tunable through flm_set_tuning)

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int TILE_F = BM * BK;  // floats per operand tile

__device__ __forceinline__ int swz(int row, int chunk) { return row * BK + ((chunk ^ ((row >> 1) & 7)) << 2); }

template <int MMAP, bool RELU>
__global__ __launch_bounds__(256, 2) void igemm_f32_kernel(IgemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* As = reinterpret_cast<float*>(smem_raw);  // [2][TILE_F]
  float* Bs = As + 2 * TILE_F;                     // [2][TILE_F]
  unsigned long long* s_mask = reinterpret_cast<unsigned long long*>(Bs + 2 * TILE_F);  // [4]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int L = xcd_remap(blockIdx.x, gridDim.x);
  const int mt = L % a.mtiles, nt = L / a.mtiles;
  const int m0 = mt * BM, n0 = nt * BN;

  // ---- staging role: rows r0+32j, 16-byte chunk c8 of the 128-byte k-run ----------------------
  const int c8 = tid & 7, r0 = tid >> 3;
  int pn[4], py[4], px[4];
  bool pv[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int m = m0 + r0 + 32 * j;
    pv[j] = m < a.M;
    const int mm = pv[j] ? m : 0;
    if (MMAP == 0) {
      px[j] = mm % a.w;
      py[j] = (mm / a.w) % a.h;
      pn[j] = mm / (a.w * a.h);
    } else if (MMAP == 1) {
      const int q = mm >> 2, d = mm & 3, wp = a.w >> 1, hp = a.h >> 1;
      px[j] = 2 * (q % wp) + (d & 1);
      py[j] = 2 * ((q / wp) % hp) + (d >> 1);
      pn[j] = q / (wp * hp);
    } else {
      pn[j] = mm % a.n;
      const int pos = mm / a.n;
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
  const int nit = __builtin_popcountll(tapmask) * a.cpt;

  // per-row element offset of the centre pixel; per-tap displacement is wave-uniform
  unsigned rowoff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) rowoff[j] = (unsigned)(((pn[j] * a.h + py[j]) * a.w + px[j]) * a.cin) + 4 * c8;
  const float* wrow[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) wrow[j] = a.wt + (size_t)(n0 + r0 + 32 * j) * a.K + 4 * c8;

  float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
  bool ok0 = false, ok1 = false, ok2 = false, ok3 = false;

  // iterator over (valid tap, channel chunk)
  unsigned long long rem = tapmask;
  int cur_tap = __builtin_ctzll(rem);
  int cur_chunk = 0;

  // Out-of-bounds rows load a valid address (their own centre pixel) and are zeroed at the LDS write,
  // so the eight loads issue back to back with no branch around them.
#define FLM_LOAD_A(J, RA, OK)                                                                        \
  {                                                                                                  \
    const int iy = py[J] + ky, ix = px[J] + kx;                                                      \
    OK = pv[J] && (unsigned)iy < (unsigned)a.h && (unsigned)ix < (unsigned)a.w;                     \
    const unsigned off = rowoff[J] + (OK ? (unsigned)tapdelta : 0u) + (unsigned)(cur_chunk * BK);  \
    RA = *reinterpret_cast<const float4*>(a.x + off);                                                \
  }
#define FLM_ISSUE_LOADS()                                                             \
  {                                                                                   \
    const int ky = cur_tap / a.kw - a.pad, kx = cur_tap % a.kw - a.pad;               \
    const int tapdelta = (ky * a.w + kx) * a.cin;                                     \
    const int koff = cur_tap * a.cin + cur_chunk * BK;                                \
    FLM_LOAD_A(0, ra0, ok0) FLM_LOAD_A(1, ra1, ok1) FLM_LOAD_A(2, ra2, ok2) FLM_LOAD_A(3, ra3, ok3) \
    rb0 = *reinterpret_cast<const float4*>(wrow[0] + koff);                           \
    rb1 = *reinterpret_cast<const float4*>(wrow[1] + koff);                           \
    rb2 = *reinterpret_cast<const float4*>(wrow[2] + koff);                           \
    rb3 = *reinterpret_cast<const float4*>(wrow[3] + koff);                           \
    if (++cur_chunk == a.cpt) {                                                       \
      cur_chunk = 0;                                                                  \
      rem &= rem - 1;                                                                 \
      cur_tap = rem ? __builtin_ctzll(rem) : 0;                                       \
    }                                                                                 \
  }
#define FLM_STORE_ROW(J, RA, RB, OK)                                                                  \
  {                                                                                                   \
    const int row = r0 + 32 * J;                                                                      \
    *reinterpret_cast<float4*>(As + sbuf * TILE_F + swz(row, c8)) = OK ? RA : make_float4(0.f, 0.f, 0.f, 0.f); \
    *reinterpret_cast<float4*>(Bs + sbuf * TILE_F + swz(row, c8)) = RB;                               \
  }
#define FLM_STORE_LDS(BUF)                                                                            \
  {                                                                                                   \
    const int sbuf = (BUF);                                                                           \
    FLM_STORE_ROW(0, ra0, rb0, ok0) FLM_STORE_ROW(1, ra1, rb1, ok1) FLM_STORE_ROW(2, ra2, rb2, ok2)   \
    FLM_STORE_ROW(3, ra3, rb3, ok3)                                                                   \
  }

  const int wr = wave >> 1, wc = wave & 1;
  const int lr = lane & 31, lh = lane >> 5;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (nit > 0) {
    FLM_ISSUE_LOADS()
    stagger_odd_workgroup(a.stagger);
    FLM_STORE_LDS(0)
  }
  __syncthreads();

  for (int it = 0; it < nit; ++it) {
    const int buf = it & 1;
    const bool more = it + 1 < nit;
    if (more && !(a.dbg & 1)) FLM_ISSUE_LOADS()
    const float* Ab = As + buf * TILE_F;
    const float* Bb = Bs + buf * TILE_F;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      float4 af[2], bf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        af[i] = *reinterpret_cast<const float4*>(Ab + swz(64 * wr + 32 * i + lr, 2 * t + lh));
        bf[i] = *reinterpret_cast<const float4*>(Bb + swz(64 * wc + 32 * i + lr, 2 * t + lh));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].x, bf[j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].y, bf[j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].z, bf[j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i].w, bf[j].w, acc[i][j], 0, 0, 0);
        }
    }
    if (more && !(a.dbg & 2)) FLM_STORE_LDS(buf ^ 1)
    if (!(a.dbg & 4)) __syncthreads();
  }

#undef FLM_LOAD_A
#undef FLM_ISSUE_LOADS
#undef FLM_STORE_ROW
#undef FLM_STORE_LDS

  // ---- epilogue: y = acc*scale + shift, ReLU, 2x2 max-pool (MMAP 1), store ---------------------
  // accumulator layout: column = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = n0 + 64 * wc + 32 * j + lr;
    const bool cok = col < a.cout;
    const float sc = a.scale[col], sh = a.shift[col];  // coutpad-long arrays: always in bounds
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int mbase = m0 + 64 * wr + 32 * i;
      if (MMAP == 1) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float v = -3.402823466e38f;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float u = fmaf(acc[i][j][4 * g + e], sc, sh);
            if (RELU) u = fmaxf(u, 0.f);
            v = fmaxf(v, u);
          }
          const int m = mbase + 8 * g + 4 * lh;  // first row of the quad
          if (cok && m < a.M) a.y[(size_t)(m >> 2) * a.ldc + col] = v;
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = mbase + (r & 3) + 8 * (r >> 2) + 4 * lh;
          float u = fmaf(acc[i][j][r], sc, sh);
          if (RELU) u = fmaxf(u, 0.f);
          if (cok && m < a.M) {
            size_t orow;
            if (MMAP == 2) {
              const int nn = m % a.n, pos = m / a.n;
              orow = (size_t)nn * (a.h * a.w) + pos;
            } else {
              orow = (size_t)m;
            }
            a.y[orow * a.ldc + col] = u;
          }
        }
      }
    }
  }
}

template <int MMAP, bool RELU>
static int launch_t(hipStream_t s, const IgemmArgs& a) {
  const size_t lds = sizeof(float) * 4 * TILE_F + 64;
  static bool attr_done = false;
  if (!attr_done) {
    FLM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_f32_kernel<MMAP, RELU>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_done = true;
  }
  igemm_f32_kernel<MMAP, RELU><<<a.mtiles * a.ntiles, 256, lds, s>>>(a);
  FLM_LAUNCH_CHECK("igemm_f32_kernel");
  return FLM_OK;
}

int igemm_occupancy(size_t lds_bytes) {
  int nb = -1;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(&igemm_f32_kernel<0, true>), 256,
                                                   lds_bytes) != hipSuccess)
    return -1;
  return nb;
}

int launch_igemm_f32(hipStream_t s, const IgemmDesc& d) {
  if (d.cin % 32 != 0 || d.coutpad % BN != 0 || d.cout > d.coutpad || d.kh * d.kw > 64) {
    set_error("igemm_f32: unsupported shape cin=%d coutpad=%d cout=%d k=%dx%d", d.cin, d.coutpad, d.cout, d.kh, d.kw);
    return FLM_ERR_SHAPE;
  }
  if (d.pool && ((d.h & 1) || (d.w & 1))) {
    set_error("igemm_f32: pooled layer needs even h,w (got %dx%d)", d.h, d.w);
    return FLM_ERR_SHAPE;
  }
  const long long M = (long long)d.n * d.h * d.w;
  if (M <= 0 || M > (1ll << 30)) {
    set_error("igemm_f32: pixel count %lld out of range", M);
    return FLM_ERR_SHAPE;
  }
  if ((long long)M * d.cin >= (1ll << 31) || (long long)d.coutpad * d.kh * d.kw * d.cin >= (1ll << 31)) {
    set_error("igemm_f32: tensor exceeds 2^31 elements (split the batch)");
    return FLM_ERR_SHAPE;
  }
  IgemmArgs a;
  a.x = d.x; a.wt = d.wt; a.scale = d.scale; a.shift = d.shift; a.y = d.y;
  a.n = d.n; a.h = d.h; a.w = d.w; a.cin = d.cin; a.cout = d.cout; a.ldc = d.ldc;
  a.kh = d.kh; a.kw = d.kw; a.pad = d.pad;
  a.M = (int)M;
  a.K = d.kh * d.kw * d.cin;
  a.mtiles = cdiv(a.M, BM);
  a.ntiles = d.coutpad / BN;
  a.cpt = d.cin / BK;
  a.stagger = g_igemm_stagger;
  a.dbg = g_igemm_debug;
  // only whole N tiles that hold stored columns are launched
  a.ntiles = cdiv(d.cout, BN);
  if (d.pool) return d.relu ? launch_t<1, true>(s, a) : launch_t<1, false>(s, a);
  if (d.posmajor) return d.relu ? launch_t<2, true>(s, a) : launch_t<2, false>(s, a);
  return d.relu ? launch_t<0, true>(s, a) : launch_t<0, false>(s, a);
}

}  // namespace flm
