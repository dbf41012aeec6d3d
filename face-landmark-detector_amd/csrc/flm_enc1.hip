// Encoder level 1 with the preprocess fused into its loader.
//
//   get_image_array sub_mean (data/generator.py:52-61): float32(u8) - [103.939,116.779,123.68] on
//   BGR, then channel reversal -> RGB;  ZeroPadding2D(1) AFTER that (networks/fcn.py:25), so the
//   mean cannot be folded into the bias; Conv2D(64,3x3) + BatchNormalization + ReLU + MaxPool 2x2
//   (networks/fcn.py:26-30).
//
// fp32 (enc1_kernel) and bf16 (enc1_bf16_kernel) variants below share the scheme: a workgroup owns a few pooled
// output rows of one face, stages the 4-row input halo of a pooled row once in LDS (preprocess fused), and
// every strip of 32 pooled pixels (64 input columns x 2 rows = 128 conv outputs) reads its im2col fragments
// straight from the halo; BN/ReLU/pool happen on the accumulator (the 2x2 window is registers 4j..4j+3 of a
// lane).  The filter matrix lives in registers.
#include "flm_common.h"

namespace flm {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// One 256-thread workgroup handles kEnc1RowsPerWgF32 pooled output rows of one face.  Per pooled row the whole
// 4-row input halo (4 x (w+2) x 3) is staged once in LDS as preprocessed fp32 (12 bytes = 3 aligned dwords per
// thread and step when the input is uint8), then every strip of 32 pooled pixels builds its im2col fragments
// straight from the halo (K = 27 padded to 32) and runs 32 v_mfma_f32_32x32x2_f32 per wave.
constexpr int kEnc1RowsPerWgF32 = 2;

__host__ __device__ inline int enc1_halo_stride(int w) {  // floats per halo row: = 16 (mod 32) keeps the two image rows
  const int need = (((w / 2 + 31) / 32) * 64 + 2) * 3;     // of a lane group on disjoint LDS banks
  return need + ((16 - need % 32) + 32) % 32;
}

template <bool U8, bool OBF, bool POOL>
__global__ __launch_bounds__(256) void enc1_kernel(const void* __restrict__ xin, const float* __restrict__ w1p,
                                                   const float* __restrict__ scale, const float* __restrict__ shift,
                                                   void* __restrict__ f1, int n, int h, int w) {
  extern __shared__ __attribute__((aligned(16))) float halo[];  // [4][hs]
  const int hs = enc1_halo_stride(w);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hp = h >> 1, wp = w >> 1;
  const int rgroups = (hp + kEnc1RowsPerWgF32 - 1) / kEnc1RowsPerWgF32;
  const int img = blockIdx.x / rgroups, yp_first = (blockIdx.x % rgroups) * kEnc1RowsPerWgF32;
  const int lr = lane & 31, lh = lane >> 5;

  // filter fragments: B[k][o] with o = 32*j + lr, k = 8t + 4*lh + e
  float4 bf[2][4];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int t = 0; t < 4; ++t)
      bf[j][t] = *reinterpret_cast<const float4*>(w1p + (32 * j + lr) * 32 + 8 * t + 4 * lh);
  float sc[2], sh[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    sc[j] = scale[32 * j + lr];
    sh[j] = shift[32 * j + lr];
  }

  // this lane's conv-output pixel inside a strip: row r = 32*wave + lr -> quad q, (dy,dx)
  const int r = 32 * wave + lr;
  const int q = r >> 2, dy = (r >> 1) & 1, dx = r & 1;
  const float mean_rgb[3] = {123.68f, 116.779f, 103.939f};  // means of B,G,R reversed to R,G,B order

  for (int yp = yp_first; yp < yp_first + kEnc1RowsPerWgF32 && yp < hp; ++yp) {
    if (yp != yp_first) __syncthreads();  // the previous row's fragment reads are done
    // ---- stage halo: rows 2yp-1..2yp+2, halo column c <-> input column c-1, 3 channels, preprocessed ------
    if (U8 && (w & 3) == 0) {
      const int gpr = w >> 2;  // groups of 4 pixels = 12 bytes = 3 aligned dwords
      for (int e = tid; e < 4 * gpr; e += 256) {
        const int hr = e / gpr, g = e % gpr;
        const int iy = 2 * yp - 1 + hr;
        unsigned d0 = 0, d1 = 0, d2 = 0;
        const bool ok = (unsigned)iy < (unsigned)h;
        if (ok) {
          const unsigned int* p = reinterpret_cast<const unsigned int*>(reinterpret_cast<const uint8_t*>(xin) +
                                                                        (((size_t)img * h + iy) * w + 4 * g) * 3);
          d0 = p[0]; d1 = p[1]; d2 = p[2];
        }
        const unsigned by[12] = {d0 & 255, (d0 >> 8) & 255, (d0 >> 16) & 255, d0 >> 24, d1 & 255, (d1 >> 8) & 255,
                                 (d1 >> 16) & 255, d1 >> 24, d2 & 255, (d2 >> 8) & 255, (d2 >> 16) & 255, d2 >> 24};
        float* dst = halo + hr * hs + (4 * g + 1) * 3;
#pragma unroll
        for (int k = 0; k < 4; ++k) {  // output channel ch (RGB order) = input channel 2-ch (BGR) minus that channel's mean
          dst[3 * k + 0] = ok ? (float)by[3 * k + 2] - mean_rgb[0] : 0.f;
          dst[3 * k + 1] = ok ? (float)by[3 * k + 1] - mean_rgb[1] : 0.f;
          dst[3 * k + 2] = ok ? (float)by[3 * k + 0] - mean_rgb[2] : 0.f;
        }
      }
      const int nz = hs - 3 * w;  // zero borders: column 0 and the floats past column w
      for (int e = tid; e < 4 * nz; e += 256) {
        const int hr = e / nz, z = e % nz;
        halo[hr * hs + (z < 3 ? z : 3 * w + z)] = 0.f;
      }
    } else {
      for (int e = tid; e < 4 * hs; e += 256) {
        const int hr = e / hs, rest = e % hs;
        const int hc = rest / 3, ch = rest % 3;
        const int iy = 2 * yp - 1 + hr, ix = hc - 1;
        float v = 0.f;
        if ((unsigned)iy < (unsigned)h && (unsigned)ix < (unsigned)w) {
          const size_t pix = ((size_t)img * h + iy) * w + ix;
          if (U8) v = (float)reinterpret_cast<const uint8_t*>(xin)[pix * 3 + (2 - ch)] - mean_rgb[ch];
          else v = reinterpret_cast<const float*>(xin)[pix * 3 + ch];
        }
        halo[e] = v;
      }
    }
    __syncthreads();

    for (int xp0 = 0; xp0 < wp; xp0 += 32) {
      const int pbase = dy * hs + (2 * (xp0 + q) + dx) * 3;
      f32x16 acc[2];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

#pragma unroll
      for (int t = 0; t < 4; ++t) {
        float af[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int k0 = 8 * t + e, k1 = 8 * t + 4 + e;  // compile-time after unroll
          const int off0 = (k0 < 27) ? (k0 / 9) * hs + (k0 % 9) : 0, off1 = (k1 < 27) ? (k1 / 9) * hs + (k1 % 9) : 0;
          const bool v0 = k0 < 27, v1 = k1 < 27;
          const float x = halo[pbase + (lh ? off1 : off0)];
          af[e] = (lh ? v1 : v0) ? x : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0], bf[j][t].x, acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[1], bf[j][t].y, acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[2], bf[j][t].z, acc[j], 0, 0, 0);
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[3], bf[j][t].w, acc[j], 0, 0, 0);
        }
      }

      // epilogue: scale/shift (BN or bias), ReLU, then either the 2x2 max over registers 4g..4g+3 (pooled pixel
      // xp0 + 8*wave + 2g + lh) or, for an un-pooled first layer (VGG block1_conv1), every conv output itself
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int xp = xp0 + 8 * wave + 2 * g + lh;
          if (POOL) {
            float v = 0.f;  // ReLU floor
#pragma unroll
            for (int e = 0; e < 4; ++e) v = fmaxf(v, fmaf(acc[j][4 * g + e], sc[j], sh[j]));
            if (xp < wp) {
              const size_t o = (((size_t)img * hp + yp) * wp + xp) * 64 + 32 * j + lr;
              if (OBF) reinterpret_cast<unsigned short*>(f1)[o] = __builtin_bit_cast(unsigned short, (__bf16)v);
              else reinterpret_cast<float*>(f1)[o] = v;
            }
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {  // row 4g+e of the quad block: (dy, dx) = (e>>1, e&1)
              const float v = fmaxf(0.f, fmaf(acc[j][4 * g + e], sc[j], sh[j]));
              const int y = 2 * yp + (e >> 1), x = 2 * xp + (e & 1);
              if (xp < wp) {
                const size_t o = (((size_t)img * h + y) * w + x) * 64 + 32 * j + lr;
                if (OBF) reinterpret_cast<unsigned short*>(f1)[o] = __builtin_bit_cast(unsigned short, (__bf16)v);
                else reinterpret_cast<float*>(f1)[o] = v;
              }
            }
          }
        }
    }
  }
}

// ---- bf16 variant (BASELINE configs[2]) ----------------------------------------------------------------
// Same layer on v_mfma_f32_32x32x16_bf16: the fp32 kernel above spends 0.9 ms of matrix time per 512 faces on
// a layer that is 1.3 % of the network's FLOPs; in bf16 it is a streaming kernel.
//   * one workgroup = kEnc1RowsPerWg pooled output rows of one face; the image rows they read pass through a
//     four-row ring in LDS as bf16 pixels padded to 4 channels (8 bytes), preprocess fused (x - mean in fp32, then
//     one rounding);
//   * K = 3 filter rows x 16: k = 16*ky + 4*kx + c with kx = 3 and c = 3 carrying zero weights, so the
//     fragment of lane (row, half) for filter row ky is the 16 bytes of halo pixels x+2*half, x+2*half+1:
//     two ds_read_b64 (x may be odd), no gather arithmetic; 3 MFMAs per 32x32 tile;
//   * halo row stride = 16 pixels (mod 32) puts the two image rows of a lane group on disjoint LDS banks.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

__host__ __device__ inline int enc1_bf16_halo_stride(int w) {
  int need = ((w / 2 + 31) / 32) * 64 + 4;  // whole strips of 32 pooled pixels + the kx = 2,3 reach of the last lane
  return need + ((16 - need % 32) + 32) % 32;
}

typedef float f32x2_e1 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_e1 __attribute__((ext_vector_type(2)));

// Epilogue of one strip of 32 pooled pixels: BN + ReLU (+ 2x2 max) on the lane's 2 x 16 accumulators, channel pairs
// stored as dwords.  Written for the instruction count -- the kernel streams 1 GB of f1 per 512 faces and its VALU
// work, not the matrix pipe, sets the pace: register pairs through v_pk_fma_f32, v_max3, one v_cvt_pk_bf16_f32 per
// stored dword, one base address per strip (the four pixel pairs of a lane are 256 bytes apart), and no bounds test
// when the row is whole strips (FULL).  `dst` points at the lane's dword of the strip's first pixel pair.
template <bool POOL, bool FULL>
__device__ __forceinline__ void enc1_bf16_store_strip(const f32x16 (&acc)[2], const float (&sc)[2], const float (&sh)[2],
                                                      unsigned int* __restrict__ dst, int xp, int wp, int w) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    f32x2_e1 u[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 2; ++e)
        u[j][e] = __builtin_elementwise_fma((f32x2_e1){acc[j][4 * g + 2 * e], acc[j][4 * g + 2 * e + 1]},
                                            (f32x2_e1){sc[j], sc[j]}, (f32x2_e1){sh[j], sh[j]});
    if (POOL) {
      const float v0 = fmaxf(fmaxf(fmaxf(fmaxf(u[0][0].x, 0.f), u[0][0].y), u[0][1].x), u[0][1].y);  // two v_max3
      const float v1 = fmaxf(fmaxf(fmaxf(fmaxf(u[1][0].x, 0.f), u[1][0].y), u[1][1].x), u[1][1].y);
      const unsigned pk = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_e1){v0, v1}, bf16x2_e1));
      if (FULL || xp + 2 * g < wp) dst[g * 64] = pk;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float v0 = fmaxf(0.f, u[0][e >> 1][e & 1]), v1 = fmaxf(0.f, u[1][e >> 1][e & 1]);
        const unsigned pk = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_e1){v0, v1}, bf16x2_e1));
        // conv pixel (2*yp + (e>>1), 2*xp + (e&1)) of the lane's quad; dst = its (2*yp, 2*xp) pixel
        if (FULL || xp + 2 * g < wp) dst[g * 128 + (e >> 1) * w * 32 + (e & 1) * 32] = pk;
      }
    }
  }
}

constexpr int kEnc1RowsPerWg = 8;  // pooled rows per workgroup: the filter fragments (36 loads per lane) are built once, and
                                   // 18 image rows are fetched for 16 (the two above the first pooled row a second time)

template <bool U8, bool POOL>
__global__ __launch_bounds__(256) void enc1_bf16_kernel(const void* __restrict__ xin, const float* __restrict__ w1p,
                                                        const float* __restrict__ scale, const float* __restrict__ shift,
                                                        unsigned short* __restrict__ f1, int n, int h, int w) {
  extern __shared__ __attribute__((aligned(16))) char smem_e1[];
  uint2* halo = reinterpret_cast<uint2*>(smem_e1);  // [4][hs] pixels of 4 bf16
  const int hs = enc1_bf16_halo_stride(w);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hp = h >> 1, wp = w >> 1;
  const int rgroups = (hp + kEnc1RowsPerWg - 1) / kEnc1RowsPerWg;
  const int img = blockIdx.x / rgroups, yp_first = (blockIdx.x % rgroups) * kEnc1RowsPerWg;
  const int lr = lane & 31, lh = lane >> 5;
  const float mean_rgb[3] = {123.68f, 116.779f, 103.939f};

  // ---- filter fragments: B[k][o], k = 16*ky + 8*lh + e -> (kx = 2*lh + (e>>2), c = e&3).  Column tile j holds the
  //      channels o = 2*lr + j: a lane then owns two neighbouring channels of a pixel and stores them as one dword
  //      (a pixel's 64 channels = one 128-byte line per half-wave) ------------------------------------------------
  bf16x8 bw[3][2];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int kx = 2 * lh + (e >> 2), c = e & 3;
        const float wv = (kx < 3 && c < 3) ? w1p[(2 * lr + j) * 32 + ky * 9 + kx * 3 + c] : 0.f;
        bw[ky][j][e] = (__bf16)wv;
      }
  float sc[2], sh[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    sc[j] = scale[2 * lr + j];
    sh[j] = shift[2 * lr + j];
  }

  // this lane's conv-output pixel inside a strip of 32 pooled pixels: row r -> quad q, (dy, dx)
  const int r = 32 * wave + lr;
  const int q = r >> 2, dy = (r >> 1) & 1, dx = r & 1;
  unsigned int* f1w = reinterpret_cast<unsigned int*>(f1);  // channel pairs

  // ---- the halo is a ring of four image rows: row iy lives in slot (iy + 1) & 3.  A pooled row reads rows 2yp-1..2yp+2;
  //      the next one keeps two of them and replaces the other two, so every image row is fetched and converted once per
  //      workgroup (plus the two rows above its first pooled row).  The two new rows are requested BEFORE the strips of
  //      the current pooled row are computed and written to LDS after them: their latency hides behind the matrix work.
  //      Halo column c <-> input column c-1; column 0 and columns w+1.. stay zero (written once).
  const uint8_t* xu8 = reinterpret_cast<const uint8_t*>(xin);
  const float* xf32 = reinterpret_cast<const float*>(xin);
  const int gpr = w >> 2;                                 // groups of 4 pixels (12 bytes = 3 aligned dwords) per row
  const bool quads = U8 && (w & 3) == 0;                  // the dword path
  const bool prefetch = quads && 2 * gpr <= 256;          // one group per thread covers both new rows
  auto px_u8 = [&](unsigned b, unsigned g_, unsigned r_) {  // BGR bytes -> RGB - mean, 4 bf16 (one rounding)
    bf16x4 v = {(__bf16)((float)r_ - mean_rgb[0]), (__bf16)((float)g_ - mean_rgb[1]), (__bf16)((float)b - mean_rgb[2]), (__bf16)0.f};
    return __builtin_bit_cast(uint2, v);
  };
  auto quad_load = [&](int iy, int g, unsigned (&d)[3]) {
    d[0] = d[1] = d[2] = 0u;
    if ((unsigned)iy < (unsigned)h) {
      const unsigned int* p = reinterpret_cast<const unsigned int*>(xu8 + (((size_t)img * h + iy) * w + 4 * g) * 3);
      d[0] = p[0]; d[1] = p[1]; d[2] = p[2];
    }
  };
  auto quad_store = [&](int iy, int g, const unsigned (&d)[3]) {
    uint2* o = halo + ((iy + 1) & 3) * hs + 4 * g + 1;
    if ((unsigned)iy < (unsigned)h) {
      o[0] = px_u8(d[0] & 255, (d[0] >> 8) & 255, (d[0] >> 16) & 255);
      o[1] = px_u8(d[0] >> 24, d[1] & 255, (d[1] >> 8) & 255);
      o[2] = px_u8((d[1] >> 16) & 255, d[1] >> 24, d[2] & 255);
      o[3] = px_u8((d[2] >> 8) & 255, (d[2] >> 16) & 255, d[2] >> 24);
    } else {
      o[0] = o[1] = o[2] = o[3] = make_uint2(0u, 0u);  // rows above / below the image
    }
  };
  auto stage_two_rows = [&](int iy0) {  // rows iy0, iy0 + 1, synchronously
    if (quads) {
      for (int e = tid; e < 2 * gpr; e += 256) {
        const int hr = e >= gpr, g = e - hr * gpr;
        unsigned d[3];
        quad_load(iy0 + hr, g, d);
        quad_store(iy0 + hr, g, d);
      }
    } else {
      for (int hr = 0; hr < 2; ++hr) {
        const int iy = iy0 + hr;
        uint2* o = halo + ((iy + 1) & 3) * hs + 1;
        for (int ix = tid; ix < w; ix += 256) {
          bf16x4 v = {(__bf16)0.f, (__bf16)0.f, (__bf16)0.f, (__bf16)0.f};
          if ((unsigned)iy < (unsigned)h) {
            const size_t pix = ((size_t)img * h + iy) * w + ix;
            if (U8) {
              o[ix] = px_u8(xu8[pix * 3], xu8[pix * 3 + 1], xu8[pix * 3 + 2]);
              continue;
            }
            v[0] = (__bf16)xf32[pix * 3]; v[1] = (__bf16)xf32[pix * 3 + 1]; v[2] = (__bf16)xf32[pix * 3 + 2];
          }
          o[ix] = __builtin_bit_cast(uint2, v);
        }
      }
    }
  };
  for (int hr = 0; hr < 4; ++hr) {  // zero borders of the four slots
    for (int z = tid; z < hs - w; z += 256) halo[hr * hs + (z == 0 ? 0 : w + z)] = make_uint2(0u, 0u);
  }
  stage_two_rows(2 * yp_first - 1);
  stage_two_rows(2 * yp_first + 1);
  __syncthreads();
  const int yp_end = min(yp_first + kEnc1RowsPerWg, hp);
  const int pf_hr = tid >= gpr, pf_g = tid - pf_hr * gpr;  // this thread's group of the two new rows (prefetch form)

  for (int yp = yp_first; yp < yp_end; ++yp) {
    const bool more = yp + 1 < yp_end;
    unsigned pf[3] = {0u, 0u, 0u};
    if (more && prefetch && tid < 2 * gpr) quad_load(2 * yp + 3 + pf_hr, pf_g, pf);
    const int slot0 = 2 * yp + dy;  // slot of filter row ky: (slot0 + ky) & 3

    for (int xp0 = 0; xp0 < wp; xp0 += 32) {
      const int x = 2 * (xp0 + q) + dx;  // conv column; halo columns x..x+3
      f32x16 acc[2];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const uint2* hp_ = halo + ((slot0 + ky) & 3) * hs + x + 2 * lh;
        const uint2 p0 = hp_[0], p1 = hp_[1];
        const uint4 av = make_uint4(p0.x, p0.y, p1.x, p1.y);
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), bw[ky][j], acc[j], 0, 0, 0);
      }
      const int xp = xp0 + 8 * wave + lh;  // pooled pixel of accumulator group g: xp + 2*g
      unsigned int* dst = POOL ? f1w + (((size_t)img * hp + yp) * wp + xp) * 32 + lr
                               : f1w + (((size_t)img * h + 2 * yp) * w + 2 * xp) * 32 + lr;
      if ((wp & 31) == 0) enc1_bf16_store_strip<POOL, true>(acc, sc, sh, dst, xp, wp, w);
      else enc1_bf16_store_strip<POOL, false>(acc, sc, sh, dst, xp, wp, w);
    }
    if (more) {
      __syncthreads();  // every wave has read the two rows that leave the ring
      if (prefetch) {
        if (tid < 2 * gpr) quad_store(2 * yp + 3 + pf_hr, pf_g, pf);
      } else {
        stage_two_rows(2 * yp + 3);
      }
      __syncthreads();
    }
  }
}

template <bool U8>
static void launch_bf16_u(hipStream_t s, int blocks, const void* x, const float* w1p, const float* scale,
                          const float* shift, void* f1, int n, int h, int w, int pool) {
  const size_t lds = (size_t)4 * enc1_bf16_halo_stride(w) * 8;
  blocks = n * (((h >> 1) + kEnc1RowsPerWg - 1) / kEnc1RowsPerWg);
  if (pool) enc1_bf16_kernel<U8, true><<<blocks, 256, lds, s>>>(x, w1p, scale, shift, (unsigned short*)f1, n, h, w);
  else enc1_bf16_kernel<U8, false><<<blocks, 256, lds, s>>>(x, w1p, scale, shift, (unsigned short*)f1, n, h, w);
}

template <bool U8>
static void launch_u(hipStream_t s, int blocks, const void* x, const float* w1p, const float* scale, const float* shift,
                     void* f1, int n, int h, int w, int out_bf16, int pool) {
  (void)out_bf16;  // bf16 output: enc1_bf16_kernel
  const size_t lds = sizeof(float) * 4 * enc1_halo_stride(w);
  blocks = n * (((h >> 1) + kEnc1RowsPerWgF32 - 1) / kEnc1RowsPerWgF32);
  if (pool) enc1_kernel<U8, false, true><<<blocks, 256, lds, s>>>(x, w1p, scale, shift, f1, n, h, w);
  else enc1_kernel<U8, false, false><<<blocks, 256, lds, s>>>(x, w1p, scale, shift, f1, n, h, w);
}

int launch_enc1(hipStream_t s, const void* x, int in_format, int n, int h, int w, const float* w1p,
                const float* scale, const float* shift, void* f1, int out_bf16, int pool) {
  if ((h & 1) || (w & 1)) {
    set_error("enc1: h,w must be even");
    return FLM_ERR_SHAPE;
  }
  const int blocks = n * (h >> 1);
  if (in_format != FLM_IN_U8_BGR && in_format != FLM_IN_F32_RGB) {
    set_error("enc1: unknown input format %d", in_format);
    return FLM_ERR_ARG;
  }
  if (out_bf16) {
    if ((size_t)4 * enc1_bf16_halo_stride(w) * 8 > 64 * 1024) {
      set_error("enc1: input width %d exceeds the bf16 kernel's halo buffer", w);
      return FLM_ERR_SHAPE;
    }
    if (in_format == FLM_IN_U8_BGR) launch_bf16_u<true>(s, blocks, x, w1p, scale, shift, f1, n, h, w, pool);
    else launch_bf16_u<false>(s, blocks, x, w1p, scale, shift, f1, n, h, w, pool);
    FLM_LAUNCH_CHECK("enc1_bf16_kernel");
    return FLM_OK;
  }
  if (sizeof(float) * 4 * enc1_halo_stride(w) > 64 * 1024) {
    set_error("enc1: input width %d exceeds the halo buffer", w);
    return FLM_ERR_SHAPE;
  }
  if (in_format == FLM_IN_U8_BGR) {
    launch_u<true>(s, blocks, x, w1p, scale, shift, f1, n, h, w, out_bf16, pool);
  } else if (in_format == FLM_IN_F32_RGB) {
    launch_u<false>(s, blocks, x, w1p, scale, shift, f1, n, h, w, out_bf16, pool);
  } else {
    set_error("enc1: unknown input format %d", in_format);
    return FLM_ERR_ARG;
  }
  FLM_LAUNCH_CHECK("enc1_kernel");
  return FLM_OK;
}

}  // namespace flm
