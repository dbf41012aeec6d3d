// MobileNet-v1 encoder pieces (reference networks/mobilenet.py:16-114, alpha = 1, depth_multiplier = 1):
//   conv1     ZeroPadding2D(1) + Conv2D(32, 3x3, stride 2, valid, no bias) + BN + ReLU6       (:16-29, :79)
//   dw block  ZeroPadding2D(1) + DepthwiseConv2D(3x3, stride s, valid, no bias) + BN + ReLU6  (:38-47)
//             (the 1x1 pointwise conv + BN + ReLU6 of the block, :49-56, runs on igemm_kernel)
// Both are bandwidth-bound (3 input channels / 9 MACs per element): plain coalesced kernels, fp32.
#include "flm_common.h"

namespace flm {

// Activations are fp32 or, in the bf16 configuration, bf16 (arithmetic stays fp32).
template <bool BF>
__device__ __forceinline__ float4 load4(const void* base, size_t idx) {
  if (BF) {
    const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(base) + idx);
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                       __uint_as_float(u.y & 0xffff0000u));
  }
  return *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + idx);
}
__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {
  return (unsigned)__builtin_bit_cast(unsigned short, (__bf16)a) | ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)b) << 16);
}
template <bool BF>
__device__ __forceinline__ void store4(void* base, size_t idx, float4 v) {
  if (BF) *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(base) + idx) = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
  else *reinterpret_cast<float4*>(reinterpret_cast<float*>(base) + idx) = v;
}

// conv1: one thread per (output pixel, 8 output channels); the 27x32 filter sits in LDS.
template <bool U8, bool BF>
__global__ __launch_bounds__(256) void mb_conv1_kernel(const void* __restrict__ xin, const float* __restrict__ wgt,
                                                       const float* __restrict__ scale,
                                                       const float* __restrict__ shift, void* __restrict__ y, int n,
                                                       int h, int w) {
  __shared__ float wl[27 * 32];
  for (int i = threadIdx.x; i < 27 * 32; i += 256) wl[i] = wgt[i];
  __syncthreads();
  const int ho = h >> 1, wo = w >> 1;
  const size_t total = (size_t)n * ho * wo * 4;
  const float mean_rgb[3] = {123.68f, 116.779f, 103.939f};
  for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (size_t)gridDim.x * 256) {
    const int og = (int)(t & 3);
    const size_t pix = t >> 2;
    const int x = (int)(pix % wo), yy = (int)((pix / wo) % ho), img = (int)(pix / ((size_t)wo * ho));
    float acc[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) acc[o] = 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int iy = 2 * yy + ky - 1, ix = 2 * x + kx - 1;
        if ((unsigned)iy >= (unsigned)h || (unsigned)ix >= (unsigned)w) continue;  // zero padding
        const size_t ip = ((size_t)img * h + iy) * w + ix;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          float v;
          if (U8) v = (float)reinterpret_cast<const uint8_t*>(xin)[ip * 3 + (2 - c)] - mean_rgb[c];
          else v = reinterpret_cast<const float*>(xin)[ip * 3 + c];
          const float* wr = wl + ((ky * 3 + kx) * 3 + c) * 32 + og * 8;
#pragma unroll
          for (int o = 0; o < 8; ++o) acc[o] = fmaf(v, wr[o], acc[o]);
        }
      }
    float r[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) r[o] = fminf(fmaxf(fmaf(acc[o], scale[og * 8 + o], shift[og * 8 + o]), 0.f), 6.f);
    store4<BF>(y, pix * 32 + og * 8, make_float4(r[0], r[1], r[2], r[3]));
    store4<BF>(y, pix * 32 + og * 8 + 4, make_float4(r[4], r[5], r[6], r[7]));
  }
}

// depthwise 3x3, stride 1 or 2, pad 1: one thread per (output pixel, 4 channels)
template <bool BF>
__global__ __launch_bounds__(256) void mb_depthwise_kernel(const void* __restrict__ x, const float* __restrict__ wgt,
                                                           const float* __restrict__ scale,
                                                           const float* __restrict__ shift, void* __restrict__ y,
                                                           int n, int h, int w, int c, int stride) {
  const int ho = h / stride, wo = w / stride, c4 = c >> 2;
  const size_t total = (size_t)n * ho * wo * c4;
  for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (size_t)gridDim.x * 256) {
    const int cq = (int)(t % c4);
    const size_t pix = t / c4;
    const int ox = (int)(pix % wo), oy = (int)((pix / wo) % ho), img = (int)(pix / ((size_t)wo * ho));
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int iy = stride * oy + ky - 1, ix = stride * ox + kx - 1;
        if ((unsigned)iy >= (unsigned)h || (unsigned)ix >= (unsigned)w) continue;
        const float4 v = load4<BF>(x, (((size_t)img * h + iy) * w + ix) * c + 4 * cq);
        const float4 k = *reinterpret_cast<const float4*>(wgt + (size_t)(ky * 3 + kx) * c + 4 * cq);
        acc.x = fmaf(v.x, k.x, acc.x);
        acc.y = fmaf(v.y, k.y, acc.y);
        acc.z = fmaf(v.z, k.z, acc.z);
        acc.w = fmaf(v.w, k.w, acc.w);
      }
    const float4 sc = *reinterpret_cast<const float4*>(scale + 4 * cq);
    const float4 sh = *reinterpret_cast<const float4*>(shift + 4 * cq);
    float4 o;
    o.x = fminf(fmaxf(fmaf(acc.x, sc.x, sh.x), 0.f), 6.f);
    o.y = fminf(fmaxf(fmaf(acc.y, sc.y, sh.y), 0.f), 6.f);
    o.z = fminf(fmaxf(fmaf(acc.z, sc.z, sh.z), 0.f), 6.f);
    o.w = fminf(fmaxf(fmaf(acc.w, sc.w, sh.w), 0.f), 6.f);
    store4<BF>(y, pix * c + 4 * cq, o);
  }
}

// ---- ResNet50 stem (reference networks/resnet50.py:145-152) ------------------------------------------
//   conv1: ZeroPadding2D(3) + Conv2D(64, 7x7, stride 2, bias) + BN + ReLU; then MaxPooling2D(3x3, stride 2, valid)
// conv1: one thread per (output pixel, 8 output channels); the 147x64 filter sits in LDS (37.6 KB).
template <bool U8, bool BF>
__global__ __launch_bounds__(256) void rn_conv1_kernel(const void* __restrict__ xin, const float* __restrict__ wgt,
                                                       const float* __restrict__ scale,
                                                       const float* __restrict__ shift, void* __restrict__ y, int n,
                                                       int h, int w) {
  extern __shared__ __attribute__((aligned(16))) float wl[];  // [147][64]
  for (int i = threadIdx.x; i < 147 * 64; i += 256) wl[i] = wgt[i];
  __syncthreads();
  const int ho = h >> 1, wo = w >> 1;
  const size_t total = (size_t)n * ho * wo * 8;
  const float mean_rgb[3] = {123.68f, 116.779f, 103.939f};
  for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (size_t)gridDim.x * 256) {
    const int og = (int)(t & 7);
    const size_t pix = t >> 3;
    const int x = (int)(pix % wo), yy = (int)((pix / wo) % ho), img = (int)(pix / ((size_t)wo * ho));
    float acc[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) acc[o] = 0.f;
    for (int ky = 0; ky < 7; ++ky) {
      const int iy = 2 * yy + ky - 3;
      if ((unsigned)iy >= (unsigned)h) continue;
      for (int kx = 0; kx < 7; ++kx) {
        const int ix = 2 * x + kx - 3;
        if ((unsigned)ix >= (unsigned)w) continue;
        const size_t ip = ((size_t)img * h + iy) * w + ix;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          float v;
          if (U8) v = (float)reinterpret_cast<const uint8_t*>(xin)[ip * 3 + (2 - c)] - mean_rgb[c];
          else v = reinterpret_cast<const float*>(xin)[ip * 3 + c];
          const float* wr = wl + ((ky * 7 + kx) * 3 + c) * 64 + og * 8;
#pragma unroll
          for (int o = 0; o < 8; ++o) acc[o] = fmaf(v, wr[o], acc[o]);
        }
      }
    }
    float r[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) r[o] = fmaxf(fmaf(acc[o], scale[og * 8 + o], shift[og * 8 + o]), 0.f);
    store4<BF>(y, pix * 64 + og * 8, make_float4(r[0], r[1], r[2], r[3]));
    store4<BF>(y, pix * 64 + og * 8 + 4, make_float4(r[4], r[5], r[6], r[7]));
  }
}

// MaxPooling2D(3x3, stride 2, 'valid'): one thread per (output pixel, 4 channels)
template <bool BF>
__global__ __launch_bounds__(256) void maxpool3_kernel(const void* __restrict__ x, void* __restrict__ y, int n, int h,
                                                       int w, int c) {
  const int ho = (h - 3) / 2 + 1, wo = (w - 3) / 2 + 1, c4 = c >> 2;
  const size_t total = (size_t)n * ho * wo * c4;
  for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (size_t)gridDim.x * 256) {
    const int cq = (int)(t % c4);
    const size_t pix = t / c4;
    const int ox = (int)(pix % wo), oy = (int)((pix / wo) % ho), img = (int)(pix / ((size_t)wo * ho));
    float4 m = make_float4(-3.402823466e38f, -3.402823466e38f, -3.402823466e38f, -3.402823466e38f);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const float4 v = load4<BF>(x, (((size_t)img * h + 2 * oy + ky) * w + 2 * ox + kx) * c + 4 * cq);
        m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
      }
    store4<BF>(y, pix * c + 4 * cq, m);
  }
}

int launch_rn_conv1(hipStream_t s, const void* x, int in_format, int n, int h, int w, const float* wgt,
                    const float* scale, const float* shift, void* y, int bf16) {
  const size_t total = (size_t)n * (h / 2) * (w / 2) * 8;
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  const size_t lds = sizeof(float) * 147 * 64;
  if (in_format == FLM_IN_U8_BGR) {
    if (bf16) rn_conv1_kernel<true, true><<<blocks, 256, lds, s>>>(x, wgt, scale, shift, y, n, h, w);
    else rn_conv1_kernel<true, false><<<blocks, 256, lds, s>>>(x, wgt, scale, shift, y, n, h, w);
  } else if (in_format == FLM_IN_F32_RGB) {
    if (bf16) rn_conv1_kernel<false, true><<<blocks, 256, lds, s>>>(x, wgt, scale, shift, y, n, h, w);
    else rn_conv1_kernel<false, false><<<blocks, 256, lds, s>>>(x, wgt, scale, shift, y, n, h, w);
  } else {
    set_error("resnet conv1: unknown input format %d", in_format);
    return FLM_ERR_ARG;
  }
  FLM_LAUNCH_CHECK("rn_conv1_kernel");
  return FLM_OK;
}

int launch_maxpool3(hipStream_t s, const void* x, int n, int h, int w, int c, void* y, int bf16) {
  if ((c & 3) || h < 3 || w < 3) {
    set_error("maxpool3: unsupported shape c=%d %dx%d", c, h, w);
    return FLM_ERR_SHAPE;
  }
  const size_t total = (size_t)n * ((h - 3) / 2 + 1) * ((w - 3) / 2 + 1) * (c / 4);
  const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
  if (bf16) maxpool3_kernel<true><<<blocks, 256, 0, s>>>(x, y, n, h, w, c);
  else maxpool3_kernel<false><<<blocks, 256, 0, s>>>(x, y, n, h, w, c);
  FLM_LAUNCH_CHECK("maxpool3_kernel");
  return FLM_OK;
}

int launch_mb_conv1(hipStream_t s, const void* x, int in_format, int n, int h, int w, const float* wgt,
                    const float* scale, const float* shift, void* y, int bf16) {
  const size_t total = (size_t)n * (h / 2) * (w / 2) * 4;
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  if (in_format == FLM_IN_U8_BGR) {
    if (bf16) mb_conv1_kernel<true, true><<<blocks, 256, 0, s>>>(x, wgt, scale, shift, y, n, h, w);
    else mb_conv1_kernel<true, false><<<blocks, 256, 0, s>>>(x, wgt, scale, shift, y, n, h, w);
  } else if (in_format == FLM_IN_F32_RGB) {
    if (bf16) mb_conv1_kernel<false, true><<<blocks, 256, 0, s>>>(x, wgt, scale, shift, y, n, h, w);
    else mb_conv1_kernel<false, false><<<blocks, 256, 0, s>>>(x, wgt, scale, shift, y, n, h, w);
  } else {
    set_error("mobilenet conv1: unknown input format %d", in_format);
    return FLM_ERR_ARG;
  }
  FLM_LAUNCH_CHECK("mb_conv1_kernel");
  return FLM_OK;
}

int launch_mb_depthwise(hipStream_t s, const void* x, int n, int h, int w, int c, int stride, const float* wgt,
                        const float* scale, const float* shift, void* y, int bf16) {
  if ((c & 3) || (stride != 1 && stride != 2) || (h % stride) || (w % stride)) {
    set_error("depthwise: unsupported shape c=%d stride=%d %dx%d", c, stride, h, w);
    return FLM_ERR_SHAPE;
  }
  const size_t total = (size_t)n * (h / stride) * (w / stride) * (c / 4);
  const int blocks = (int)((total + 255) / 256 < 16384 ? (total + 255) / 256 : 16384);
  if (bf16) mb_depthwise_kernel<true><<<blocks, 256, 0, s>>>(x, wgt, scale, shift, y, n, h, w, c, stride);
  else mb_depthwise_kernel<false><<<blocks, 256, 0, s>>>(x, wgt, scale, shift, y, n, h, w, c, stride);
  FLM_LAUNCH_CHECK("mb_depthwise_kernel");
  return FLM_OK;
}

}  // namespace flm
