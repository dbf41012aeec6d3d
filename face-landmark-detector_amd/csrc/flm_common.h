// Shared declarations of libflm_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <atomic>
#include <mutex>

#include "flm.h"

namespace flm {

void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what);

#define FLM_HIP(call)                                            \
  do {                                                           \
    hipError_t e_ = (call);                                      \
    if (e_ != hipSuccess) return ::flm::hip_fail(e_, #call);     \
  } while (0)

#define FLM_LAUNCH_CHECK(what)                                   \
  do {                                                           \
    hipError_t e_ = hipGetLastError();                           \
    if (e_ != hipSuccess) return ::flm::hip_fail(e_, what);      \
  } while (0)

// hipFuncSetAttribute (dynamic LDS above 64 KiB) once per kernel instantiation AND device, safe under concurrent
// callers: a launcher keeps one static FuncAttrOnce and calls FLM_FUNC_ATTR_ONCE(flag, kernel, lds) before launching.
struct FuncAttrOnce {
  std::atomic<unsigned> done{0};  // one bit per device ordinal
  std::mutex mu;
};
#define FLM_FUNC_ATTR_ONCE(flag, kernel, lds_bytes)                                                        \
  do {                                                                                                     \
    int dev_ = 0;                                                                                          \
    FLM_HIP(hipGetDevice(&dev_));                                                                          \
    const unsigned bit_ = 1u << (dev_ & 31);                                                               \
    if (!((flag).done.load(std::memory_order_acquire) & bit_)) {                                           \
      std::lock_guard<std::mutex> lock_((flag).mu);                                                        \
      if (!((flag).done.load(std::memory_order_relaxed) & bit_)) {                                         \
        FLM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),                                 \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds_bytes)));        \
        (flag).done.fetch_or(bit_, std::memory_order_release);                                             \
      }                                                                                                    \
    }                                                                                                      \
  } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

constexpr float kBnEps = 1e-3f;  // keras BatchNormalization default epsilon

// ---- architecture constants (networks/fcn.py:13,34,43,98,100) ------------------------------
constexpr int kFc = 4096;
constexpr int kMaxClasses = 96;
constexpr int kMaxEnc = 64;

// Encoder description: a list of layers; layer i reads the output of layer `src` (default: i-1, or the
// network input for i = 0) and may add the output of layer `res` before its activation.
enum EncKind {
  ENC_FIRST3 = 0,    // 3-channel Conv2D 3x3 'same' + (BN) + ReLU (+ MaxPool 2x2): enc1_kernel
  ENC_CONV3 = 1,     // Conv2D 3x3 'same' + (BN) + ReLU (+ MaxPool 2x2): igemm_kernel
  ENC_MB_CONV1 = 2,  // MobileNet conv1: pad 1, 3x3 stride 2, no bias, BN, ReLU6 (3 -> 32)
  ENC_MB_DW = 3,     // MobileNet depthwise 3x3 (stride 1|2), no bias, BN, ReLU6
  ENC_MB_PW = 4,     // MobileNet pointwise 1x1, no bias, BN, ReLU6: igemm_kernel
  ENC_RN_CONV1 = 5,  // ResNet conv1: pad 3, 7x7 stride 2, bias, BN, ReLU (3 -> 64)
  ENC_MAXPOOL3 = 6,  // MaxPooling2D 3x3 stride 2 'valid' (no parameters)
  ENC_CONV = 7       // generic Conv2D k x k (k = 1|3, stride 1|2, 'same' when stride 1) + bias + BN (+ residual)
                     // (+ ReLU): igemm_kernel
};
struct EncLayer {
  int cin, cout, bn, pool;
  int kind, stride;
  int k, relu;   // ENC_CONV: kernel size and whether a ReLU follows
  int src, res;  // layer indices (-1: previous layer / none)
};
struct ArchSpec {
  int n_enc;
  EncLayer enc[kMaxEnc];
  int f_idx[5];  // index of the layer whose (pooled) output is f1..f5 (f1, f2 are not used by the decoders)
  int fcn32;     // one 64x64 stride-32 transposed conv instead of the FCN-8 skip decoder
  int valid;
};
// Output grid of every encoder layer for an h x w input.
void enc_dims(const ArchSpec& A, int h, int w, int* hs, int* ws);
inline int enc_has_params(const EncLayer& e) { return e.kind != ENC_MAXPOOL3; }
// FLM_ARCH_FCN8 / FCN32: vanilla_encoder (networks/fcn.py:10-51); *_VGG: get_vgg_encoder (networks/vgg16.py:17-81);
// *_MOBILENET: get_mobilenet_encoder (networks/mobilenet.py:59-114); *_RESNET50: get_resnet50_encoder
// (networks/resnet50.py:122-182)
ArchSpec arch_spec(int arch);

// Geometry of the transposed-conv kernels for a class count C.
struct ConvTGeom {
  int C;   // classes
  int Cp;  // channel stride of the (fp32) score / fuse buffers: multiple of 4 (fp32) or 8 (bf16)
  int MT;  // 16-row class tiles
  int G;   // k groups over K = 4 taps * Cp: 16 deep (fp32, 16x16x4 MFMA x4) or 32 deep (bf16, 16x16x32)
  int bf16;
};
ConvTGeom convt_geom(int C, int dtype);

// One implicit-GEMM conv layer inside the packed blob (offsets in bytes).
struct ConvPack {
  size_t w;      // fp32 or bf16 [coutpad][kh*kw*cin], k = (ky*kw+kx)*cin + c
  size_t scale;  // float [coutpad]
  size_t shift;  // float [coutpad]
  int cin, cout, coutpad, kh, kw, pad;
};

struct Fcn8Pack {
  size_t enc1_w;  // float [64][32], k = ky*9+kx*3+c (c in RGB order), zero for k >= 27
  size_t enc1_scale, enc1_shift;
  ConvPack enc[kMaxEnc];  // enc[i] packs encoder layer i for i >= 1 (layer 0 is the 3-channel first conv)
  ConvPack fc6, fc7, score5, score4, score3;
  size_t up5, up4, up3;  // [s*s phases][G][MT][64 lanes][16 bytes: 4 fp32 or 8 bf16]
  ConvTGeom g;
  int dtype;
  int arch;
  ArchSpec spec;
  size_t total;
};
Fcn8Pack fcn8_pack_layout(int C, int dtype, int arch = 0);

// Workspace of the forward (offsets in bytes).
struct Fcn8Ws {
  size_t act[kMaxEnc];  // output of every encoder layer
  size_t f[5];          // = act[spec.f_idx[k]]
  size_t fc6, fc7, score5, fuse4, seg;
  size_t splitk;  // split-K partial sums of the score convs
  size_t splitk_bytes;
  size_t probs;   // logits/probs when they are not the call's output (else == SIZE_MAX)
  size_t decode;  // decode partials
  // landmark mode without the probability tensor (flm_convt.hip); cand == SIZE_MAX when not used
  size_t sub, tau, cand, cand_cnt;
  int cand_cap;
  size_t total;
  int oh, ow;
};
// `opts` (NULL = defaults) carries the per-call options that change the layout (include/flm.h: flm_forward_opts).
Fcn8Ws fcn8_ws_layout(int n, int h, int w, int C, int dtype, int out_mode, int decode_mode, int n_points,
                      int arch = 0, const flm_forward_opts* opts = nullptr);

// ---- kernel launchers (each returns FLM_OK or an error) --------------------------------------
int launch_pack_fcn(hipStream_t s, const flm_fcn_params& p, int C, const Fcn8Pack& L, char* blob);

int launch_enc1(hipStream_t s, const void* x, int in_format, int n, int h, int w, const float* w1p,
                const float* scale, const float* shift, void* f1, int out_bf16, int pool);

struct IgemmDesc {
  const void* x;       // [n,h,w,cin] fp32, or bf16 when `bf16`
  const void* wt;      // [coutpad][K], same type as x
  const float* scale;  // [coutpad]
  const float* shift;  // [coutpad]
  void* y;             // [n,ho,wo,ldc]  (ho,wo = h,w or h/2,w/2 when pooled); bf16 unless out_f32
  int bf16;            // operands are bf16 (v_mfma_f32_32x32x16_bf16), accumulate fp32
  int out_f32;         // bf16 path: store fp32 (score convs feeding the fp32 decoder buffers)
  int n, h, w, cin;
  int cout;     // columns stored
  int coutpad;  // rows of wt (multiple of 128)
  int ldc;      // channel stride of y
  int kh, kw, pad;
  int relu, pool, posmajor;  // relu: 0 none, 1 ReLU, 2 ReLU6
  int stride;               // 1 (0 = 1); 2 only in the row-major order (ResNet's strided 1x1 convs)
  const void* res;          // optional residual [n,ho,wo,ldc] added before the activation
  float* splitk_ws;        // optional scratch for split-K partial sums (null: never split)
  size_t splitk_ws_bytes;
};
int launch_igemm(hipStream_t s, const IgemmDesc& d);
int igemm_occupancy(size_t lds_bytes);
void igemm_f32_group(int steps);
void igemm_bf16_big_enable(int on);
void igemm_bf16_group_n(int gn);
void igemm_bf16_big_dma(int on);
void igemm_bf16_big_m16(int on);
void conv3_halo_enable(int on);
void conv3_halo_m16(int on);
void score1x1_enable(int on);
void tail_fused_enable(int on);
void decode_dma_enable(int on);
void warp_rows_enable(int on);
void igemm_posperm_enable(int on);
// seg_feats = crop(up4(fuse4)) + score3(f3) in one launch (flm_tail_bf16.hip; bf16, 68 classes, 256-channel f3):
// 1 launched, 0 shape left to the two-launch form, < 0 error
int launch_seg_fused_bf16(hipStream_t s, const float* fuse4, const void* w4_packed, const void* f3, const void* w3,
                          const float* scale3, const float* shift3, float* seg, int n, int h4, int w4d, int C, int Cp,
                          int G, int cin3, int coutpad3);

struct ConvTDesc {
  const float* x;     // [n,hi,wi,Cp]
  const void* wf;     // fragment-packed weights (fp32 or bf16 per g.bf16)
  const float* skip;  // [n,ho,wo,Cp] added to the result, or null
  void* y;            // logits/probs float [n,ho,wo,ldy] or int32 class map [n,ho,wo]
  int n, hi, wi;      // input grid
  int ho, wo;         // output grid after the crop (<= s*(hi+1))
  int s;              // stride; kernel = 2s
  int ldy;            // channel stride of y (C for the final layer, Cp for score buffers)
  int epilogue;       // 0 raw (+skip), 1 softmax probs, 2 argmax class map, 3 top-n candidates (no map written),
                      // 4 per-wave class maxima of a sampling launch (sub > 0)
  ConvTGeom g;
  // landmark mode without the probability tensor (see flm_convt.hip); all zero / null otherwise
  int sub = 0;                         // > 0: sampling launch, `sub` phases per tile, compact [n][sub][hi+1][wi+1][C] output
  const float* tau = nullptr;          // [n][C] thresholds (epilogue 3)
  unsigned long long* cand = nullptr;  // [n][cand_cap] candidate keys
  unsigned* cand_cnt = nullptr;        // [n] + overflow flag at [n]
  int cand_cap = 0;
  const unsigned* gate = nullptr;      // launch is a no-op unless *gate != 0
  void* scratch = nullptr;             // epilogue 3: memory the launch may use (the bf16 input copy of up3_wreg_kernel), or null
  size_t scratch_bytes = 0;
};
int launch_convt(hipStream_t s, const ConvTDesc& d);
int convt_candidates_supported(const ConvTGeom& g);
// 68-class kernels with a stride that is a multiple of 4 (up3: 8, fcn_32: 32) share the fifth class tile between
// four consecutive phases (flm_convt.hip, flm_pack.hip).
__host__ __device__ inline int convt_share_layout(const ConvTGeom& g, int s) {
  return g.C == 68 && (g.bf16 ? g.G == 9 : g.G == 17) && (s % 4) == 0;
}
int convt_sample_slots(const ConvTGeom& g, int hi, int wi, int sub);
void convt_cand8_enable(int mask);  // A/B knob "up3_cand8"
void convt_cand8_rows(int rpw);     // A/B knob "up3_cand8_rows"
void convt_wreg_enable(int on);     // A/B knob "up3_wreg"
int launch_cand_tau(hipStream_t s, const unsigned* wave_max, int n, int slots, int ld, int l, int n_points, float* tau);

// (activations fp32, or bf16 when `bf16`)
int launch_mb_conv1(hipStream_t s, const void* x, int in_format, int n, int h, int w, const float* wgt,
                    const float* scale, const float* shift, void* y, int bf16);
int launch_mb_depthwise(hipStream_t s, const void* x, int n, int h, int w, int c, int stride, const float* wgt,
                        const float* scale, const float* shift, void* y, int bf16);
int launch_rn_conv1(hipStream_t s, const void* x, int in_format, int n, int h, int w, const float* wgt,
                    const float* scale, const float* shift, void* y, int bf16);
int launch_maxpool3(hipStream_t s, const void* x, int n, int h, int w, int c, void* y, int bf16);

size_t decode_ws_bytes(int n, int h, int w, int l, int mode, int n_points);
int launch_decode(hipStream_t s, const float* hm, int n, int h, int w, int l, int ld, int mode, int n_points,
                  float thresh, double* out, void* ws, size_t ws_bytes, float* tau_out = nullptr,
                  const unsigned* gate = nullptr);
int launch_cand_merge(hipStream_t s, const unsigned long long* cand, unsigned* cand_cnt, int n, int w, int l,
                      int n_points, float thresh, int cap, double* out);

int launch_preprocess(hipStream_t s, const uint8_t* img, int n, int h, int w, int norm, float* out);
int launch_similarity(hipStream_t s, const double* lm, const double* tmpl, int n, int k, double sx, double sy, float* m);
int launch_warp(hipStream_t s, const void* src, int src_is_u8, int n, int hs, int ws, const float* m, float* dst,
                int hd, int wd);
int launch_crop_resize(hipStream_t s, const uint8_t* frame, int fh, int fw, const int32_t* boxes, int k,
                       uint8_t* out, int oh, int ow, const int32_t* frame_idx = nullptr, size_t frame_stride = 0,
                       int nframes = 0);

// Bijective XCD-aware remap of a 1-D grid: blocks that the dispatcher deals to the same XCD
// (b % 8) receive consecutive logical ids, so neighbours in logical order share an L2.
__device__ __forceinline__ int xcd_remap(int b, int nblk) {
  const int q = nblk >> 3, r = nblk & 7;
  const int xcd = b & 7, loc = b >> 3;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + loc;
}

}  // namespace flm
