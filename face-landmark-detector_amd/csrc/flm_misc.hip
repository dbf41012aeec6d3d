// Bandwidth-bound helpers around the forward: standalone preprocess, similarity estimate,
// alignment warp, crop front-end.
#include "flm_common.h"

#include <atomic>

namespace flm {

// ---- get_image_array (reference data/generator.py:29-69) for crops already at model size ------------
//   sub_mean        float32(u8) - [103.939,116.779,123.68] per BGR channel, then channel reversal (:52-61)
//   sub_and_divide  float32(u8)/127.5 - 1   (:50-51, no channel reversal)
//   divide          float32(u8)/255         (:62-65, no channel reversal)
__global__ void preprocess_kernel(const uint8_t* __restrict__ img, float* __restrict__ out, size_t npix, int norm) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < npix; i += (size_t)gridDim.x * blockDim.x) {
    const float b = (float)img[3 * i + 0], g = (float)img[3 * i + 1], r = (float)img[3 * i + 2];
    float o0, o1, o2;
    if (norm == FLM_NORM_SUB_MEAN) {
      o0 = r - 123.68f;
      o1 = g - 116.779f;
      o2 = b - 103.939f;
    } else if (norm == FLM_NORM_SUB_AND_DIVIDE) {
      o0 = b / 127.5f - 1.f;
      o1 = g / 127.5f - 1.f;
      o2 = r / 127.5f - 1.f;
    } else {
      o0 = b / 255.0f;
      o1 = g / 255.0f;
      o2 = r / 255.0f;
    }
    out[3 * i + 0] = o0;
    out[3 * i + 1] = o1;
    out[3 * i + 2] = o2;
  }
}

int launch_preprocess(hipStream_t s, const uint8_t* img, int n, int h, int w, int norm, float* out) {
  if (norm < 0 || norm > 2) {
    set_error("preprocess: unknown imgNorm %d", norm);
    return FLM_ERR_ARG;
  }
  const size_t npix = (size_t)n * h * w;
  const int blocks = (int)((npix + 255) / 256 < 4096 ? (npix + 255) / 256 : 4096);
  preprocess_kernel<<<blocks > 0 ? blocks : 1, 256, 0, s>>>(img, out, npix, norm);
  FLM_LAUNCH_CHECK("preprocess_kernel");
  return FLM_OK;
}

// ---- least-squares similarity (no reflection) from K landmark pairs -----------------------------------
// Complex form: dst ~ alpha*src + beta, alpha = sum(conj(p')q') / sum|p'|^2 over centred points.
// Landmarks the decode rejected ((-1,-1), utils/metrics.py:78-79) are left out; fewer than two
// usable points (or a degenerate cloud) gives the identity.  float64, sequential sums in landmark order
// (oracle/warp_ref.py restates them one by one), so the arithmetic stays with ONE thread per face; the other 63
// lanes of its wave stage the face's landmarks (times the grid-to-crop scale sx, sy; rejected points keep their
// negative marker) and the template in LDS, so the serial loops run on LDS latency, not on a global load per term.
__global__ __launch_bounds__(64) void similarity_kernel(const double* __restrict__ lm, const double* __restrict__ tmpl,
                                                        int n, int k, double sx, double sy, float* __restrict__ m) {
  extern __shared__ double sim_s[];  // [k][2] landmarks, [k][2] template
  const int f = blockIdx.x;
  double* p = sim_s;
  double* t = sim_s + 2 * k;
  for (int i = threadIdx.x; i < 2 * k; i += 64) {
    const double v = lm[(size_t)f * k * 2 + i];
    p[i] = v < 0.0 ? v : v * ((i & 1) ? sy : sx);
    t[i] = tmpl[i];
  }
  __syncthreads();
  if (threadIdx.x != 0) return;
  double mpx = 0, mpy = 0, mqx = 0, mqy = 0;
  int cnt = 0;
  for (int i = 0; i < k; ++i) {
    if (p[2 * i] < 0.0 || p[2 * i + 1] < 0.0) continue;
    mpx += p[2 * i]; mpy += p[2 * i + 1];
    mqx += t[2 * i]; mqy += t[2 * i + 1];
    ++cnt;
  }
  double a = 1.0, b = 0.0, tx = 0.0, ty = 0.0;
  if (cnt >= 2) {
    mpx /= cnt; mpy /= cnt; mqx /= cnt; mqy /= cnt;
    double sa = 0, sb = 0, var = 0;
    for (int i = 0; i < k; ++i) {
      if (p[2 * i] < 0.0 || p[2 * i + 1] < 0.0) continue;
      const double px = p[2 * i] - mpx, py = p[2 * i + 1] - mpy;
      const double qx = t[2 * i] - mqx, qy = t[2 * i + 1] - mqy;
      sa += px * qx + py * qy;
      sb += px * qy - py * qx;
      var += px * px + py * py;
    }
    if (var > 0.0) {
      a = sa / var;
      b = sb / var;
      tx = mqx - (a * mpx - b * mpy);
      ty = mqy - (b * mpx + a * mpy);
    }
  }
  float* o = m + (size_t)f * 6;
  o[0] = (float)a; o[1] = (float)(-b); o[2] = (float)tx;
  o[3] = (float)b; o[4] = (float)a;    o[5] = (float)ty;
}

int launch_similarity(hipStream_t s, const double* lm, const double* tmpl, int n, int k, double sx, double sy, float* m) {
  if (n <= 0 || k <= 0 || k > 1024) {
    set_error("similarity: bad sizes n=%d k=%d", n, k);
    return FLM_ERR_SHAPE;
  }
  similarity_kernel<<<n, 64, sizeof(double) * 4 * k, s>>>(lm, tmpl, n, k, sx, sy, m);
  FLM_LAUNCH_CHECK("similarity_kernel");
  return FLM_OK;
}

// ---- alignment warp: inverse-map bilinear, edge clamp, explicit fma order ------------------------------
// M maps source pixel coords to aligned coords; each output pixel samples the source at M^-1 (xd, yd).
// The arithmetic below is the specification (oracle/warp_ref.py restates it operation by operation):
//   det  = fma(m00, m11, -(m01*m10));  idet = 1/det
//   i00 = m11*idet; i01 = -m01*idet; i10 = -m10*idet; i11 = m00*idet
//   i02 = -fma(i00, m02, i01*m12);     i12 = -fma(i10, m02, i11*m12)
//   xs = fma(i00, xd, fma(i01, yd, i02));  ys = fma(i10, xd, fma(i11, yd, i12))
//   clamp xs to [0, Ws-1], ys to [0, Hs-1]   (skimage warp mode="edge", data/generator.py:200)
//   x0 = floor(xs), fx = xs-x0, x1 = min(x0+1, Ws-1)  (same for y)
//   top = fma(fx, p01-p00, p00); bot = fma(fx, p11-p10, p10); out = fma(fy, bot-top, top)
// The aligned faces are written once and read by a later launch (or by the host): their stores carry the non-temporal
// hint, so that the 786,432 B a face writes stream past the L2 instead of evicting the source lines the gathers of
// the neighbouring pixels are about to re-read (batch 512: 0.161 -> 0.104 ms; the PMC's FETCH_SIZE had shown the
// source fetched twice).
__device__ __forceinline__ void store_stream16(float* p, const float4& v) {
  typedef float f4v __attribute__((ext_vector_type(4)));
  const f4v vv = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(vv, reinterpret_cast<f4v*>(p));
}

template <bool U8>
__global__ __launch_bounds__(256) void warp_kernel(const void* __restrict__ src, int hs, int ws,
                                                   const float* __restrict__ m, float* __restrict__ dst, int hd,
                                                   int wd) {
  const int f = blockIdx.y;
  const float* mm = m + (size_t)f * 6;
  const float m00 = mm[0], m01 = mm[1], m02 = mm[2], m10 = mm[3], m11 = mm[4], m12 = mm[5];
  const float det = fmaf(m00, m11, -(m01 * m10));
  const float idet = 1.0f / det;
  const float i00 = m11 * idet, i01 = -m01 * idet, i10 = -m10 * idet, i11 = m00 * idet;
  const float i02 = -fmaf(i00, m02, i01 * m12), i12 = -fmaf(i10, m02, i11 * m12);
  const int npix = hd * wd;
  // per-face bases are wave-uniform; everything below them is 32-bit (the launcher checks that a face fits)
  const uint8_t* s8 = reinterpret_cast<const uint8_t*>(src) + (size_t)f * hs * ws * 3;
  const float* sf = reinterpret_cast<const float*>(src) + (size_t)f * hs * ws * 3;
  float* dface = dst + (size_t)f * npix * 3;
  // A wave's 64 pixels are 768 contiguous bytes of the output: the three floats of a pixel go through a per-wave LDS
  // line and leave as 48 x 16-byte stores instead of 64 x (8 + 4)-byte ones.
  __shared__ float stage[4][192];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const bool wide = (npix & 3) == 0;
  const int pend = (npix + 63) & ~63;   // whole waves run the loop together (the staging needs every lane's pixel)
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < pend; p += gridDim.x * blockDim.x) {
    const bool live = p < npix;
    const int pc = live ? p : npix - 1;
    const int py = pc / wd;
    const float xd = (float)(pc - py * wd), yd = (float)py;
    float xs = fmaf(i00, xd, fmaf(i01, yd, i02));
    float ys = fmaf(i10, xd, fmaf(i11, yd, i12));
    xs = fminf(fmaxf(xs, 0.f), (float)(ws - 1));
    ys = fminf(fmaxf(ys, 0.f), (float)(hs - 1));
    const float xf = floorf(xs), yf = floorf(ys);
    const float fx = xs - xf, fy = ys - yf;
    const int x0 = (int)xf, y0 = (int)yf;
    const int x1 = min(x0 + 1, ws - 1), y1 = min(y0 + 1, hs - 1);
    float o3[3];
    if (U8 && ws >= 2) {
      // The two pixels of a source row are six contiguous bytes: two (unaligned) dword loads per row instead of six
      // byte loads -- the kernel was bound by the gather's load instructions, not by the 983,040 B per face it moves
      // (batch 512: 0.235 -> 0.190 ms).  The pair starts at xl = min(x0, ws - 2), so the second dword, bytes 2..5 of the
      // pair, ends inside the row; x0 = ws - 1 happens only for xs = ws - 1 exactly (fx = 0, x1 = x0): both samples are
      // then the pair's second pixel.
      const int xl = min(x0, ws - 2);
      const bool second = x0 != xl;
      const int ot = (y0 * ws + xl) * 3, ob = (y1 * ws + xl) * 3;
      unsigned ta, tb, ba, bb;
      __builtin_memcpy(&ta, s8 + ot, 4);
      __builtin_memcpy(&tb, s8 + ot + 2, 4);
      __builtin_memcpy(&ba, s8 + ob, 4);
      __builtin_memcpy(&bb, s8 + ob + 2, 4);
      // pixel 0 = bytes 0,1,2 of the first dword; pixel 1 = byte 3 of the first, bytes 2,3 of the second
      const float t0[3] = {(float)(ta & 0xffu), (float)((ta >> 8) & 0xffu), (float)((ta >> 16) & 0xffu)};
      const float t1[3] = {(float)(ta >> 24), (float)((tb >> 16) & 0xffu), (float)(tb >> 24)};
      const float b0[3] = {(float)(ba & 0xffu), (float)((ba >> 8) & 0xffu), (float)((ba >> 16) & 0xffu)};
      const float b1[3] = {(float)(ba >> 24), (float)((bb >> 16) & 0xffu), (float)(bb >> 24)};
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float p00 = second ? t1[c] : t0[c], p01 = t1[c];
        const float p10 = second ? b1[c] : b0[c], p11 = b1[c];
        const float top = fmaf(fx, p01 - p00, p00);
        const float bot = fmaf(fx, p11 - p10, p10);
        o3[c] = fmaf(fy, bot - top, top);
      }
    } else {
    const int o00 = (y0 * ws + x0) * 3, o01 = (y0 * ws + x1) * 3;
    const int o10 = (y1 * ws + x0) * 3, o11 = (y1 * ws + x1) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float p00, p01, p10, p11;
      if (U8) {
        p00 = (float)s8[o00 + c]; p01 = (float)s8[o01 + c]; p10 = (float)s8[o10 + c]; p11 = (float)s8[o11 + c];
      } else {
        p00 = sf[o00 + c]; p01 = sf[o01 + c]; p10 = sf[o10 + c]; p11 = sf[o11 + c];
      }
      const float top = fmaf(fx, p01 - p00, p00);
      const float bot = fmaf(fx, p11 - p10, p10);
      o3[c] = fmaf(fy, bot - top, top);
    }
    }
    const int pbase = p - lane;   // first pixel of the wave
    if (wide && pbase + 64 <= npix) {
      stage[wv][3 * lane + 0] = o3[0];
      stage[wv][3 * lane + 1] = o3[1];
      stage[wv][3 * lane + 2] = o3[2];
      __builtin_amdgcn_wave_barrier();
      if (lane < 48) {
        const float4 v = *reinterpret_cast<const float4*>(&stage[wv][4 * lane]);
        store_stream16(dface + pbase * 3 + 4 * lane, v);
      }
      __builtin_amdgcn_wave_barrier();
    } else if (live) {
      float* d = dface + p * 3;
      d[0] = o3[0]; d[1] = o3[1]; d[2] = o3[2];
    }
  }
}

// uint8 sources (ws >= 2): UNR pixels per thread with all of their gathers in flight before the first is consumed --
// with one pixel at a time the waves spent 72 % of their cycles parked on the four dword loads of a pixel (PMC), and
// a CU's 32 waves x 4 loads x 256 B in flight bound the kernel by latency, not by bytes.  Same arithmetic per pixel as
// warp_kernel<true>, operation for operation.
// Developer timing ablations (tools/ab_variants.py, -DFLM_WARP_VAR=<mask>; results wrong; 0 in shipped builds):
// 1 no output stores, 2 every gather reads the face's first bytes (perfect locality), 4 no gathers at all.
#ifndef FLM_WARP_VAR
#define FLM_WARP_VAR 0
#endif
template <int UNR>
__global__ __launch_bounds__(256) void warp_u8_kernel(const uint8_t* __restrict__ src, int hs, int ws,
                                                      const float* __restrict__ m, float* __restrict__ dst, int hd,
                                                      int wd) {
  const int f = blockIdx.y;
  const float* mm = m + (size_t)f * 6;
  const float m00 = mm[0], m01 = mm[1], m02 = mm[2], m10 = mm[3], m11 = mm[4], m12 = mm[5];
  const float det = fmaf(m00, m11, -(m01 * m10));
  const float idet = 1.0f / det;
  const float i00 = m11 * idet, i01 = -m01 * idet, i10 = -m10 * idet, i11 = m00 * idet;
  const float i02 = -fmaf(i00, m02, i01 * m12), i12 = -fmaf(i10, m02, i11 * m12);
  const int npix = hd * wd;
  const uint8_t* s8 = src + (size_t)f * hs * ws * 3;
  float* dface = dst + (size_t)f * npix * 3;
  __shared__ float stage[4][192];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const bool wide = (npix & 3) == 0;
  const int pend = (npix + 63) & ~63;  // whole waves run the loop together (the staging needs every lane's pixel)
  const int stride = gridDim.x * blockDim.x;
  for (int p0 = blockIdx.x * blockDim.x + threadIdx.x; p0 < pend; p0 += UNR * stride) {
    float fx[UNR], fy[UNR];
    unsigned ta[UNR], tb[UNR], ba[UNR], bb[UNR];
    bool second[UNR];
#pragma unroll
    for (int k = 0; k < UNR; ++k) {
      const int p = p0 + k * stride;
      const int pc = p < npix ? p : npix - 1;
      const int py = pc / wd;
      const float xd = (float)(pc - py * wd), yd = (float)py;
      float xs = fmaf(i00, xd, fmaf(i01, yd, i02));
      float ys = fmaf(i10, xd, fmaf(i11, yd, i12));
      xs = fminf(fmaxf(xs, 0.f), (float)(ws - 1));
      ys = fminf(fmaxf(ys, 0.f), (float)(hs - 1));
      const float xf = floorf(xs), yf = floorf(ys);
      fx[k] = xs - xf;
      fy[k] = ys - yf;
      const int x0 = (int)xf, y0 = (int)yf;
      const int y1 = min(y0 + 1, hs - 1);
      const int xl = min(x0, ws - 2);  // the pair (xl, xl + 1): see warp_kernel
      second[k] = x0 != xl;
      int ot = (y0 * ws + xl) * 3, ob = (y1 * ws + xl) * 3;
      if (FLM_WARP_VAR & 2) { ot = (ot & 3) + 4 * lane; ob = (ob & 3) + 4 * lane + 512; }
      if (FLM_WARP_VAR & 4) {
        ta[k] = (unsigned)ot; tb[k] = (unsigned)ob; ba[k] = (unsigned)(ot + ob); bb[k] = (unsigned)(ot ^ ob);
      } else {
      __builtin_memcpy(&ta[k], s8 + ot, 4);
      __builtin_memcpy(&tb[k], s8 + ot + 2, 4);
      __builtin_memcpy(&ba[k], s8 + ob, 4);
      __builtin_memcpy(&bb[k], s8 + ob + 2, 4);
      }
    }
#pragma unroll
    for (int k = 0; k < UNR; ++k) {
      const int p = p0 + k * stride;
      if (p - lane >= pend) break;  // (wave-uniform)
      const float t0[3] = {(float)(ta[k] & 0xffu), (float)((ta[k] >> 8) & 0xffu), (float)((ta[k] >> 16) & 0xffu)};
      const float t1[3] = {(float)(ta[k] >> 24), (float)((tb[k] >> 16) & 0xffu), (float)(tb[k] >> 24)};
      const float b0[3] = {(float)(ba[k] & 0xffu), (float)((ba[k] >> 8) & 0xffu), (float)((ba[k] >> 16) & 0xffu)};
      const float b1[3] = {(float)(ba[k] >> 24), (float)((bb[k] >> 16) & 0xffu), (float)(bb[k] >> 24)};
      float o3[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float p00 = second[k] ? t1[c] : t0[c], p01 = t1[c];
        const float p10 = second[k] ? b1[c] : b0[c], p11 = b1[c];
        const float top = fmaf(fx[k], p01 - p00, p00);
        const float bot = fmaf(fx[k], p11 - p10, p10);
        o3[c] = fmaf(fy[k], bot - top, top);
      }
      const int pbase = p - lane;  // first pixel of the wave
      if (wide && pbase + 64 <= npix) {
        stage[wv][3 * lane + 0] = o3[0];
        stage[wv][3 * lane + 1] = o3[1];
        stage[wv][3 * lane + 2] = o3[2];
        __builtin_amdgcn_wave_barrier();
        if (lane < 48) {
          const float4 v = *reinterpret_cast<const float4*>(&stage[wv][4 * lane]);
          if (!(FLM_WARP_VAR & 1) || v.x == 12345.678f) store_stream16(dface + pbase * 3 + 4 * lane, v);
        }
        __builtin_amdgcn_wave_barrier();
      } else if (p < npix) {
        float* d = dface + p * 3;
        d[0] = o3[0]; d[1] = o3[1]; d[2] = o3[2];
      }
    }
  }
}

static std::atomic<int> g_warp_rows{1};  // A/B knob "warp_rows": same results either way
void warp_rows_enable(int on) { g_warp_rows.store(on, std::memory_order_relaxed); }

// uint8 sources, destination width a multiple of 64: one wave = 64 consecutive pixels of ROWS consecutive output rows, a
// workgroup = a strip of 256 columns x ROWS rows (grid: strips across, row groups, faces).  What the FLM_WARP_VAR
// ablations of warp_u8_kernel showed at batch 512 (0.163 ms as shipped): 0.092 ms without the stores, 0.090 ms without
// the gathers, 0.080 ms with neither -- (a) gathers and stores each cost little alone and nearly their sum together: the
// written faces were displacing the source from the L2 (store_stream16 above: 0.163 -> 0.109 ms), and (b) the floor
// is the kernel's own VALU work, about 150 instructions per pixel with the quarter-rate 32-bit integer multiplies of
// the pixel -> (row, column) division and of the byte offsets.  Here
//   - the row and the segment come from the block and wave indices: no division, the row terms are one per wave,
//   - the byte offsets use 24-bit multiplies (full rate; the launcher checks hs, ws < 2^24),
//   - the x0 = ws - 1 case is expressed through the weight instead of six selects: the pair starts at xl = ws - 2 and
//     fx becomes 1, fmaf(1, t1 - t0, t0) = t1 exactly (small integers), the value warp_kernel computes.
// Per pixel the arithmetic is warp_kernel<true>'s, the results the same bits (tests/test_gpu_align.py).  Back-to-back
// launches (bench.py's hbm_kernels protocol; the source stays in the Infinity Cache): 0.109 -> 0.100 ms per 512 faces
// (0.63 of 8 TB/s by the algorithmic 983,040 B per face), 0.0165 -> 0.0131 per 64.  Inside the step (source cold) 0.150 against
// 0.153 ms.  The workgroup is a STRIP, not a 64 x 8 patch: equal on near-identity transforms, but on the degenerate
// transforms the bench pipeline produces (landmarks of random weights fit the template with scale ~0.1 and any
// rotation: a destination row is a diagonal through the source, most samples clamp to a border COLUMN at varying rows)
// the patch form took 0.197 ms inside the step where the strip takes 0.164 and the pixel-list kernel 0.162.
template <int ROWS>
__global__ __launch_bounds__(256) void warp_u8_rows_kernel(const uint8_t* __restrict__ src, int hs, int ws,
                                                           const float* __restrict__ m, float* __restrict__ dst,
                                                           int hd, int wd) {
  const int f = blockIdx.z;
  const float* mm = m + (size_t)f * 6;
  const float m00 = mm[0], m01 = mm[1], m02 = mm[2], m10 = mm[3], m11 = mm[4], m12 = mm[5];
  const float det = fmaf(m00, m11, -(m01 * m10));
  const float idet = 1.0f / det;
  const float i00 = m11 * idet, i01 = -m01 * idet, i10 = -m10 * idet, i11 = m00 * idet;
  const float i02 = -fmaf(i00, m02, i01 * m12), i12 = -fmaf(i10, m02, i11 * m12);
  const uint8_t* s8 = src + (size_t)f * hs * ws * 3;
  float* dface = dst + (size_t)f * hd * wd * 3;
  __shared__ float stage[4][192];
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // a workgroup's four waves lie side by side: a strip of 256 columns x ROWS rows
  const int xseg = (blockIdx.x * 4 + wv) * 64;
  const int row0 = blockIdx.y * ROWS;
  if (xseg >= wd) return;  // (wave-uniform; no workgroup barrier below)
  const float xd = (float)(xseg + lane);
  const float xmax = (float)(ws - 1), ymax = (float)(hs - 1);
  const unsigned ws3 = (unsigned)ws * 3u;
  float fx[ROWS], fy[ROWS];
  unsigned ta[ROWS], tb[ROWS], ba[ROWS], bb[ROWS];
#pragma unroll
  for (int k = 0; k < ROWS; ++k) {
    const int row = min(row0 + k, hd - 1);  // rows past the image repeat the last one (loaded, never stored)
    const float yd = (float)row;
    float xs = fmaf(i00, xd, fmaf(i01, yd, i02));
    float ys = fmaf(i10, xd, fmaf(i11, yd, i12));
    xs = fminf(fmaxf(xs, 0.f), xmax);
    ys = fminf(fmaxf(ys, 0.f), ymax);
    const float xf = floorf(xs), yf = floorf(ys);
    fy[k] = ys - yf;
    const int x0 = (int)xf, y0 = (int)yf;
    const int xl = min(x0, ws - 2);
    fx[k] = x0 != xl ? 1.0f : xs - xf;
    const unsigned ot = (__umul24((unsigned)y0, (unsigned)ws) + (unsigned)xl) * 3u;
    const unsigned ob = ot + (y0 + 1 < hs ? ws3 : 0u);
    if (FLM_WARP_VAR & 4) {
      ta[k] = ot; tb[k] = ob; ba[k] = ot + ob; bb[k] = ot ^ ob;
    } else {
    __builtin_memcpy(&ta[k], s8 + ot, 4);
    __builtin_memcpy(&tb[k], s8 + ot + 2, 4);
    __builtin_memcpy(&ba[k], s8 + ob, 4);
    __builtin_memcpy(&bb[k], s8 + ob + 2, 4);
    }
  }
#pragma unroll
  for (int k = 0; k < ROWS; ++k) {
    const int row = row0 + k;
    if (row >= hd) break;  // (wave-uniform)
    const float t0[3] = {(float)(ta[k] & 0xffu), (float)((ta[k] >> 8) & 0xffu), (float)((ta[k] >> 16) & 0xffu)};
    const float t1[3] = {(float)(ta[k] >> 24), (float)((tb[k] >> 16) & 0xffu), (float)(tb[k] >> 24)};
    const float b0[3] = {(float)(ba[k] & 0xffu), (float)((ba[k] >> 8) & 0xffu), (float)((ba[k] >> 16) & 0xffu)};
    const float b1[3] = {(float)(ba[k] >> 24), (float)((bb[k] >> 16) & 0xffu), (float)(bb[k] >> 24)};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float top = fmaf(fx[k], t1[c] - t0[c], t0[c]);
      const float bot = fmaf(fx[k], b1[c] - b0[c], b0[c]);
      stage[wv][3 * lane + c] = fmaf(fy[k], bot - top, top);
    }
    __builtin_amdgcn_wave_barrier();
    if (lane < 48) {
      const float4 v = *reinterpret_cast<const float4*>(&stage[wv][4 * lane]);
      if (!(FLM_WARP_VAR & 1) || v.x == 12345.678f) store_stream16(dface + ((size_t)row * wd + xseg) * 3 + 4 * lane, v);
    }
    __builtin_amdgcn_wave_barrier();
  }
}

int launch_warp(hipStream_t s, const void* src, int src_is_u8, int n, int hs, int ws, const float* m, float* dst,
                int hd, int wd) {
  if (n <= 0 || hs <= 0 || ws <= 0 || hd <= 0 || wd <= 0 || n > 65535 || (long long)hs * ws * 12 >= (1ll << 31) ||
      (long long)hd * wd * 12 >= (1ll << 31)) {
    set_error("warp: bad sizes (a face must stay below 2^31 bytes on either side)");
    return FLM_ERR_SHAPE;
  }
  if (src_is_u8 && ws >= 2 && (wd & 63) == 0 && hs < (1 << 24) && ws < (1 << 24) &&
      g_warp_rows.load(std::memory_order_relaxed)) {
    // rows per wave: 2 by default (knob value 4: four)
    const int rows = g_warp_rows.load(std::memory_order_relaxed) == 4 ? 4 : 2;
    const int gy = cdiv(hd, rows);
    if (gy <= 65535) {
      const dim3 grid(cdiv(wd / 64, 4), gy, n);
      const uint8_t* s8 = static_cast<const uint8_t*>(src);
      if (rows == 4) warp_u8_rows_kernel<4><<<grid, 256, 0, s>>>(s8, hs, ws, m, dst, hd, wd);
      else warp_u8_rows_kernel<2><<<grid, 256, 0, s>>>(s8, hs, ws, m, dst, hd, wd);
      FLM_LAUNCH_CHECK("warp_u8_rows_kernel");
      return FLM_OK;
    }
  }
  if (src_is_u8 && ws >= 2) {
    constexpr int UNR = 4;  // (2: 0.165 ms, 4: 0.161-0.163, 8: 0.164, 16: 0.192 per 512 faces)
    int bu = cdiv(hd * wd, 256 * UNR);
    if (bu > 1024) bu = 1024;
    warp_u8_kernel<UNR><<<dim3(bu, n), 256, 0, s>>>(static_cast<const uint8_t*>(src), hs, ws, m, dst, hd, wd);
    FLM_LAUNCH_CHECK("warp_u8_kernel");
    return FLM_OK;
  }
  int bx = cdiv(hd * wd, 256 * 2);  // two pixels per thread (batch 512: 0.194 -> 0.179 ms; 4, 8, 16 no better)
  if (bx > 1024) bx = 1024;
  dim3 grid(bx, n);
  if (src_is_u8) warp_kernel<true><<<grid, 256, 0, s>>>(src, hs, ws, m, dst, hd, wd);
  else warp_kernel<false><<<grid, 256, 0, s>>>(src, hs, ws, m, dst, hd, wd);
  FLM_LAUNCH_CHECK("warp_kernel");
  return FLM_OK;
}

// ---- crop front-end: K boxes of one frame -> model-size uint8 BGR crops ------------------------------
// detect_marks' `img[y0:y1, x0:x1]` + `cv2.resize(face_img, (W, H))` (reference prediction.py:80-82) and the
// `cv2.resize(img, (width, height))` of get_image_array (data/generator.py:53), both with cv2's default
// INTER_LINEAR on uint8.  The arithmetic lives in OpenCV (unpinned dependency, absent here): this kernel restates
// the published generic algorithm of imgproc/resize.cpp for 8-bit INTER_LINEAR in INTEGER fixed point, so that
// both sides of the parity test (oracle/warp_ref.py) are exact:
//   region  = box clipped to the frame (numpy slicing clips the high side; the reference's wrap-around for
//             negative starts is a defect that is not reproduced), cw x ch pixels at (cx0, cy0)
//   scale_x = 1.0 / ((double)ow / cw)                                           (double)
//   fx      = (float)((x + 0.5) * scale_x - 0.5);  sx = floor(fx);  fx -= sx    (product and sum in double)
//   sx < 0 -> sx = 0, fx = 0;   sx >= cw-1 -> sx = cw-1, fx = 0
//   a1 = (int)rint(fx * 2048),  a0 = (int)rint((1.f - fx) * 2048)               (11-bit weights, ties to even)
//   row r:  h_r = S[r][sx]*a0 + S[r][min(sx+1, cw-1)]*a1                        (int32; rows clamp(sy), clamp(sy+1))
//   out = (((b0 * (h_0 >> 4)) >> 16) + ((b1 * (h_1 >> 4)) >> 16) + 2) >> 2      (b0, b1: the weights of the UNCLAMPED
//                                                                                 fy: only the row indices clamp)
//   exact 2x downscale in both axes (cw == 2*ow, ch == 2*oh): cv2 switches INTER_LINEAR to the area average
//   out = (S[2y][2x] + S[2y][2x+1] + S[2y+1][2x] + S[2y+1][2x+1] + 2) >> 2.
__device__ __forceinline__ void resize_coef(int d, double scale, int n_src, int& s0, int& s1, int& w0, int& w1) {
  float f = (float)(((double)d + 0.5) * scale - 0.5);
  int s = (int)floorf(f);
  f -= (float)s;
  if (s < 0) { s = 0; f = 0.f; }
  if (s >= n_src - 1) { s = n_src - 1; f = 0.f; }
  s0 = s;
  s1 = min(s + 1, n_src - 1);
  w0 = (int)rintf((1.f - f) * 2048.f);
  w1 = (int)rintf(f * 2048.f);
}

// Along y OpenCV clamps only the ROW INDICES (clip(sy + k, 0, h)) and keeps the split weights of the unclamped
// position: on the first / last output rows of an upscale both rows are the border row, weighted b0 and b1 separately
// -- floor(b0*v >> 16) + floor(b1*v >> 16) is not always (2048*v) >> 16, so folding the weights there is off by one LSB.
__device__ __forceinline__ void resize_coef_y(int d, double scale, int n_src, int& s0, int& s1, int& w0, int& w1) {
  float f = (float)(((double)d + 0.5) * scale - 0.5);
  const int s = (int)floorf(f);
  f -= (float)s;
  s0 = min(max(s, 0), n_src - 1);
  s1 = min(max(s + 1, 0), n_src - 1);
  w0 = (int)rintf((1.f - f) * 2048.f);
  w1 = (int)rintf(f * 2048.f);
}

// frame_idx (may be null): box k is cut from frame frame_idx[k] of `nframes` frames laid out `frame_stride` bytes apart
// (a ring of stream frames in one allocation): the faces of a whole group of frames in one launch.
__global__ __launch_bounds__(256) void crop_resize_kernel(const uint8_t* __restrict__ frame, int fh, int fw,
                                                          const int32_t* __restrict__ boxes, uint8_t* __restrict__ out,
                                                          int oh, int ow, const int32_t* __restrict__ frame_idx,
                                                          size_t frame_stride, int nframes) {
  const int k = blockIdx.y;
  if (frame_idx) {
    const int fi = frame_idx[k];
    if ((unsigned)fi >= (unsigned)nframes) {  // an index outside the ring: zeros, like a box outside its frame
      uint8_t* dz = out + (size_t)k * oh * ow * 3;
      for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < oh * ow * 3; p += gridDim.x * blockDim.x) dz[p] = 0;
      return;
    }
    frame += (size_t)fi * frame_stride;
  }
  const int cx0 = min(max(boxes[4 * k + 0], 0), fw), cy0 = min(max(boxes[4 * k + 1], 0), fh);
  const int cx1 = min(max(boxes[4 * k + 2], 0), fw), cy1 = min(max(boxes[4 * k + 3], 0), fh);
  const int cw = cx1 - cx0, ch = cy1 - cy0;
  const int npix = oh * ow;
  uint8_t* dst = out + (size_t)k * npix * 3;
  if (cw <= 0 || ch <= 0) {  // box outside the frame: nothing to sample (cv2.resize raises on an empty crop)
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < npix * 3; p += gridDim.x * blockDim.x) dst[p] = 0;
    return;
  }
  const uint8_t* src = frame + ((size_t)cy0 * fw + cx0) * 3;
  const size_t pitch = (size_t)fw * 3;
  const bool area2 = (cw == 2 * ow) && (ch == 2 * oh);
  const double scale_x = 1.0 / ((double)ow / (double)cw), scale_y = 1.0 / ((double)oh / (double)ch);
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < npix; p += gridDim.x * blockDim.x) {
    const int x = p % ow, y = p / ow;
    uint8_t* d = dst + (size_t)p * 3;
    if (area2) {
      const uint8_t* r0 = src + (size_t)(2 * y) * pitch + (size_t)(2 * x) * 3;
      const uint8_t* r1 = r0 + pitch;
#pragma unroll
      for (int c = 0; c < 3; ++c) d[c] = (uint8_t)(((int)r0[c] + (int)r0[3 + c] + (int)r1[c] + (int)r1[3 + c] + 2) >> 2);
      continue;
    }
    int x0, x1, a0, a1, y0, y1, b0, b1;
    resize_coef(x, scale_x, cw, x0, x1, a0, a1);
    resize_coef_y(y, scale_y, ch, y0, y1, b0, b1);
    const uint8_t* r0 = src + (size_t)y0 * pitch;
    const uint8_t* r1 = src + (size_t)y1 * pitch;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int h0 = (int)r0[x0 * 3 + c] * a0 + (int)r0[x1 * 3 + c] * a1;
      const int h1 = (int)r1[x0 * 3 + c] * a0 + (int)r1[x1 * 3 + c] * a1;
      d[c] = (uint8_t)((((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2);
    }
  }
}

int launch_crop_resize(hipStream_t s, const uint8_t* frame, int fh, int fw, const int32_t* boxes, int k,
                       uint8_t* out, int oh, int ow, const int32_t* frame_idx, size_t frame_stride, int nframes) {
  if (k <= 0 || fh <= 0 || fw <= 0 || oh <= 0 || ow <= 0 || k > 65535 ||
      (frame_idx && (nframes <= 0 || frame_stride < (size_t)fh * fw * 3))) {
    set_error("crop_resize: bad sizes");
    return FLM_ERR_SHAPE;
  }
  int bx = cdiv(oh * ow, 256);
  if (bx > 1024) bx = 1024;
  crop_resize_kernel<<<dim3(bx, k), 256, 0, s>>>(frame, fh, fw, boxes, out, oh, ow, frame_idx, frame_stride, nframes);
  FLM_LAUNCH_CHECK("crop_resize_kernel");
  return FLM_OK;
}

}  // namespace flm
