// Encoder level 1 with the preprocess fused into its loader.
//
//   get_image_array sub_mean (data/generator.py:52-61): float32(u8) - [103.939,116.779,123.68] on
//   BGR, then channel reversal -> RGB;  ZeroPadding2D(1) AFTER that (networks/fcn.py:25), so the
//   mean cannot be folded into the bias; Conv2D(64,3x3) + BatchNormalization + ReLU + MaxPool 2x2
//   (networks/fcn.py:26-30).
//
// One 256-thread workgroup handles one pooled output row of one face: per strip of 32 pooled
// pixels (64 input columns x 2 input rows = 128 conv outputs) it stages the 4 x 66 x 3 input halo
// in LDS as preprocessed fp32, builds the im2col fragments straight from the halo (K = 27 padded
// to 32) and runs 32 v_mfma_f32_32x32x2_f32 per wave; BN/ReLU/pool happen on the accumulator
// (the 2x2 window is registers 4j..4j+3 of a lane).  The 64x32 filter matrix lives in registers.
#include "flm_common.h"

namespace flm {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int HALO_W = 66;              // 64 columns + 1 each side
constexpr int HALO_STRIDE = 208;        // floats per halo row: 198 used; 208 = 16 (mod 32) keeps the two
                                        // image rows of a lane group on disjoint LDS banks
constexpr int HALO_F = 4 * HALO_STRIDE;

__device__ __forceinline__ int koff_of(int k) {  // halo offset of patch element k = ky*9 + kx*3 + c
  return (k < 27) ? (k / 9) * HALO_STRIDE + (k % 9) : -1;
}

template <bool U8, bool OBF, bool POOL>
__global__ __launch_bounds__(256) void enc1_kernel(const void* __restrict__ xin, const float* __restrict__ w1p,
                                                   const float* __restrict__ scale, const float* __restrict__ shift,
                                                   void* __restrict__ f1, int n, int h, int w) {
  __shared__ __attribute__((aligned(16))) float halo[HALO_F];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hp = h >> 1, wp = w >> 1;
  const int img = blockIdx.x / hp, yp = blockIdx.x % hp;
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

  // this lane's conv-output pixel inside the strip: row r = 32*wave + lr -> quad q, (dy,dx)
  const int r = 32 * wave + lr;
  const int q = r >> 2, dy = (r >> 1) & 1, dx = r & 1;
  const int pbase = dy * HALO_STRIDE + (2 * q + dx) * 3;

  const float mean_rgb[3] = {123.68f, 116.779f, 103.939f};  // means of B,G,R reversed to R,G,B order

  for (int xp0 = 0; xp0 < wp; xp0 += 32) {
    __syncthreads();  // previous strip's fragment reads are done
    // ---- stage halo: rows 2yp-1..2yp+2, cols 2xp0-1..2xp0+64, 3 channels, preprocessed ----------
    for (int e = tid; e < 4 * HALO_W * 3; e += 256) {
      const int hr = e / (HALO_W * 3), rest = e % (HALO_W * 3);
      const int hc = rest / 3, ch = rest % 3;
      const int iy = 2 * yp - 1 + hr, ix = 2 * xp0 - 1 + hc;
      float v = 0.f;
      if ((unsigned)iy < (unsigned)h && (unsigned)ix < (unsigned)w) {
        const size_t pix = ((size_t)img * h + iy) * w + ix;
        if (U8) {
          // output channel ch (RGB order) = input channel 2-ch (BGR) minus that channel's mean
          v = (float)reinterpret_cast<const uint8_t*>(xin)[pix * 3 + (2 - ch)] - mean_rgb[ch];
        } else {
          v = reinterpret_cast<const float*>(xin)[pix * 3 + ch];
        }
      }
      halo[hr * HALO_STRIDE + hc * 3 + ch] = v;
    }
    __syncthreads();

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
        const int k0 = koff_of(8 * t + e), k1 = koff_of(8 * t + 4 + e);  // compile-time after unroll
        const int off = lh ? k1 : k0;
        af[e] = (off >= 0) ? halo[pbase + off] : 0.f;
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

template <bool U8>
static void launch_u(hipStream_t s, int blocks, const void* x, const float* w1p, const float* scale, const float* shift,
                     void* f1, int n, int h, int w, int out_bf16, int pool) {
  if (out_bf16) {
    if (pool) enc1_kernel<U8, true, true><<<blocks, 256, 0, s>>>(x, w1p, scale, shift, f1, n, h, w);
    else enc1_kernel<U8, true, false><<<blocks, 256, 0, s>>>(x, w1p, scale, shift, f1, n, h, w);
  } else {
    if (pool) enc1_kernel<U8, false, true><<<blocks, 256, 0, s>>>(x, w1p, scale, shift, f1, n, h, w);
    else enc1_kernel<U8, false, false><<<blocks, 256, 0, s>>>(x, w1p, scale, shift, f1, n, h, w);
  }
}

int launch_enc1(hipStream_t s, const void* x, int in_format, int n, int h, int w, const float* w1p,
                const float* scale, const float* shift, void* f1, int out_bf16, int pool) {
  if ((h & 1) || (w & 1)) {
    set_error("enc1: h,w must be even");
    return FLM_ERR_SHAPE;
  }
  const int blocks = n * (h >> 1);
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
