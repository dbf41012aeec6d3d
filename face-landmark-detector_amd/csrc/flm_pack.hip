// Weight repacking: Keras layouts -> the layouts the gfx950 kernels read.
// Replaces keras `model.load_weights` (reference prediction.py:128) on the device side.
#include "flm_common.h"

namespace flm {

ConvTGeom convt_geom(int C, int dtype) {
  ConvTGeom g;
  g.C = C;
  g.bf16 = dtype == FLM_BF16;
  g.MT = cdiv(C, 16);
  if (C == 68) {
    // exact paths for the 68-landmark model: fp32 K = 4*68 = 17 groups of 16; bf16 K = 4*72 = 9 groups of 32
    g.Cp = g.bf16 ? 72 : 68;
    g.G = g.bf16 ? 9 : 17;
  } else {  // generic path: channels padded to the class-tile size
    g.Cp = 16 * g.MT;
    g.G = g.bf16 ? 2 * g.MT : 4 * g.MT;
  }
  return g;
}

static EncLayer mk(int kind, int cin, int cout, int bn, int pool, int stride, int k = 3, int relu = 1, int src = -1,
                   int res = -1) {
  EncLayer e;
  e.kind = kind; e.cin = cin; e.cout = cout; e.bn = bn; e.pool = pool; e.stride = stride;
  e.k = k; e.relu = relu; e.src = src; e.res = res;
  return e;
}

ArchSpec arch_spec(int arch) {
  ArchSpec A;
  A.valid = 1;
  A.n_enc = 0;
  A.fcn32 = (arch == FLM_ARCH_FCN32 || arch == FLM_ARCH_FCN32_VGG || arch == FLM_ARCH_FCN32_MOBILENET ||
             arch == FLM_ARCH_FCN32_RESNET50);
  if (arch == FLM_ARCH_FCN8 || arch == FLM_ARCH_FCN32) {
    // vanilla_encoder, networks/fcn.py:10-51: 5 x (pad 1, conv 3x3, BN, ReLU, pool); F = 64,128,256,256,256
    const int f[6] = {3, 64, 128, 256, 256, 256};
    A.n_enc = 5;
    for (int i = 0; i < 5; ++i) {
      A.enc[i] = mk(i == 0 ? ENC_FIRST3 : ENC_CONV3, f[i], f[i + 1], 1, 1, 1);
      A.f_idx[i] = i;
    }
  } else if (arch == FLM_ARCH_FCN8_VGG || arch == FLM_ARCH_FCN32_VGG) {
    // get_vgg_encoder, networks/vgg16.py:27-72: conv 3x3 'same' + ReLU, pool closes each block
    const int blocks[5] = {2, 2, 3, 3, 3};
    const int ch[5] = {64, 128, 256, 512, 512};
    int cin = 3, k = 0;
    for (int bl = 0; bl < 5; ++bl)
      for (int c = 0; c < blocks[bl]; ++c) {
        A.enc[k] = mk(k == 0 ? ENC_FIRST3 : ENC_CONV3, cin, ch[bl], 0, c == blocks[bl] - 1, 1);
        if (c == blocks[bl] - 1) A.f_idx[bl] = k;
        cin = ch[bl];
        ++k;
      }
    A.n_enc = k;  // 13
  } else if (arch == FLM_ARCH_FCN8_MOBILENET || arch == FLM_ARCH_FCN32_MOBILENET) {
    // get_mobilenet_encoder, networks/mobilenet.py:79-102: conv1 (stride 2), then 13 depthwise-separable blocks;
    // strides 2 at blocks 2, 4, 6, 12; f1..f5 = outputs of blocks 1, 3, 5, 11, 13
    const int pw[13] = {64, 128, 128, 256, 256, 512, 512, 512, 512, 512, 512, 1024, 1024};
    const int st[13] = {1, 2, 1, 2, 1, 2, 1, 1, 1, 1, 1, 2, 1};
    int k = 0, cin = 32;
    A.enc[k++] = mk(ENC_MB_CONV1, 3, 32, 1, 0, 2);
    for (int b = 0; b < 13; ++b) {
      A.enc[k++] = mk(ENC_MB_DW, cin, cin, 1, 0, st[b]);
      A.enc[k++] = mk(ENC_MB_PW, cin, pw[b], 1, 0, 1, 1, 2);
      cin = pw[b];
    }
    A.n_enc = k;  // 27
    const int fb[5] = {1, 3, 5, 11, 13};
    for (int i = 0; i < 5; ++i) A.f_idx[i] = 2 * fb[i];  // index of conv_pw_<block>
  } else if (arch == FLM_ARCH_FCN8_RESNET50 || arch == FLM_ARCH_FCN32_RESNET50) {
    // get_resnet50_encoder, networks/resnet50.py:145-170: conv1 7x7 s2 + BN + ReLU, MaxPooling2D(3x3, s2, valid),
    // stages 2..5 of bottleneck blocks [3,4,6,3]; a stage's first block has a 1x1 shortcut conv and (stages 3..5)
    // stride 2 on its first 1x1 and on the shortcut (:81-118); f3/f4/f5 = outputs of stages 3/4/5
    int k = 0;
    A.enc[k++] = mk(ENC_RN_CONV1, 3, 64, 1, 0, 2, 7, 1);
    A.enc[k++] = mk(ENC_MAXPOOL3, 64, 64, 0, 0, 2);
    const int nblk[4] = {3, 4, 6, 3};
    const int width[4] = {64, 128, 256, 512};
    int cin = 64, last = k - 1;
    for (int st = 0; st < 4; ++st) {
      const int f1 = width[st], f3 = 4 * width[st];
      for (int b = 0; b < nblk[st]; ++b) {
        const int s = (b == 0 && st > 0) ? 2 : 1;
        int shortcut = last;  // identity block: the block input
        if (b == 0) {         // conv_block: 1x1 (stride s) + BN on the shortcut, no ReLU
          A.enc[k] = mk(ENC_CONV, cin, f3, 1, 0, s, 1, 0, last, -1);
          shortcut = k++;
        }
        A.enc[k] = mk(ENC_CONV, cin, f1, 1, 0, s, 1, 1, last, -1); ++k;   // branch2a
        A.enc[k] = mk(ENC_CONV, f1, f1, 1, 0, 1, 3, 1, k - 1, -1); ++k;   // branch2b
        A.enc[k] = mk(ENC_CONV, f1, f3, 1, 0, 1, 1, 1, k - 1, shortcut);  // branch2c + add + ReLU
        last = k++;
        cin = f3;
      }
      A.f_idx[st + 1] = last;
    }
    A.f_idx[0] = 0;
    A.n_enc = k;  // 2 + 16*3 + 4 = 54
  } else {
    A.valid = 0;
  }
  return A;
}

void enc_dims(const ArchSpec& A, int h, int w, int* hs, int* ws) {
  for (int i = 0; i < A.n_enc; ++i) {
    const EncLayer& e = A.enc[i];
    const int src = e.src >= 0 ? e.src : i - 1;
    const int hi = src >= 0 ? hs[src] : h, wi = src >= 0 ? ws[src] : w;
    int ho = hi, wo = wi;
    switch (e.kind) {
      case ENC_FIRST3:
      case ENC_CONV3:
        if (e.pool) { ho = hi / 2; wo = wi / 2; }
        break;
      case ENC_MB_CONV1:
      case ENC_RN_CONV1:
        ho = hi / 2; wo = wi / 2;  // pad k/2, stride 2 on an even grid
        break;
      case ENC_MB_DW:
        ho = hi / e.stride; wo = wi / e.stride;
        break;
      case ENC_MAXPOOL3:
        ho = (hi - 3) / 2 + 1; wo = (wi - 3) / 2 + 1;  // 'valid'
        break;
      case ENC_CONV: {
        const int pad = e.k / 2;
        ho = (hi + 2 * pad - e.k) / e.stride + 1;
        wo = (wi + 2 * pad - e.k) / e.stride + 1;
        break;
      }
      default:
        break;
    }
    hs[i] = ho;
    ws[i] = wo;
  }
}

static size_t take(size_t& cur, size_t bytes) {
  size_t o = cur;
  cur = align_up(cur + bytes, 256);
  return o;
}

static ConvPack conv_pack(size_t& cur, int kh, int kw, int pad, int cin, int cout, int coutpad, int es) {
  ConvPack c;
  c.kh = kh; c.kw = kw; c.pad = pad; c.cin = cin; c.cout = cout; c.coutpad = coutpad;
  c.w = take(cur, (size_t)es * coutpad * kh * kw * cin);
  c.scale = take(cur, sizeof(float) * coutpad);
  c.shift = take(cur, sizeof(float) * coutpad);
  return c;
}

Fcn8Pack fcn8_pack_layout(int C, int dtype, int arch) {
  Fcn8Pack L;
  L.g = convt_geom(C, dtype);
  L.dtype = dtype;
  L.arch = arch;
  L.spec = arch_spec(arch);
  const ArchSpec& A = L.spec;
  const int fcn32 = A.fcn32;
  const int es = dtype == FLM_BF16 ? 2 : 4;
  size_t cur = 0;
  L.enc1_w = take(cur, sizeof(float) * 64 * 147);  // 64x32 (3x3 first convs) .. 147x64 (ResNet conv1)
  L.enc1_scale = take(cur, sizeof(float) * 64);
  L.enc1_shift = take(cur, sizeof(float) * 64);
  for (int i = 1; i < A.n_enc; ++i) {
    const EncLayer& e = A.enc[i];
    if (e.kind == ENC_MB_DW) {  // [9][C] filter + scale/shift, fp32
      ConvPack c;
      c.kh = c.kw = 3; c.pad = 1; c.cin = c.cout = c.coutpad = e.cin;
      c.w = take(cur, sizeof(float) * 9 * e.cin);
      c.scale = take(cur, sizeof(float) * e.cin);
      c.shift = take(cur, sizeof(float) * e.cin);
      L.enc[i] = c;
    } else if (e.kind == ENC_MB_PW) {
      // bf16 implicit GEMMs take input channels in multiples of 64: the 32-channel pointwise conv (MobileNet block 1)
      // is run on PAIRS of neighbouring pixels -- [M][32] read as [M/2][64] against the block-diagonal filter
      // [[W,0],[0,W]] gives [M/2][2*cout], which is [M][cout] in memory
      const int pair = (dtype == FLM_BF16 && e.cin == 32) ? 2 : 1;
      L.enc[i] = conv_pack(cur, 1, 1, 0, e.cin * pair, e.cout * pair, (int)align_up(e.cout * pair, 128), es);
    } else if (e.kind == ENC_CONV) {
      L.enc[i] = conv_pack(cur, e.k, e.k, e.k / 2, e.cin, e.cout, (int)align_up(e.cout, 128), es);
    } else if (e.kind == ENC_MAXPOOL3) {
      L.enc[i] = ConvPack{};
    } else {
      L.enc[i] = conv_pack(cur, 3, 3, 1, e.cin, e.cout, (int)align_up(e.cout, 128), es);
    }
  }
  const int c3 = A.enc[A.f_idx[2]].cout, c4 = A.enc[A.f_idx[3]].cout, c5 = A.enc[A.f_idx[4]].cout;
  L.fc6 = conv_pack(cur, 7, 7, 3, c5, kFc, kFc, es);
  L.fc7 = conv_pack(cur, 1, 1, 0, kFc, kFc, kFc, es);
  // score convs write Cp channels (pad channels come out as exact zeros)
  L.score5 = conv_pack(cur, 1, 1, 0, kFc, L.g.Cp, 128, es);
  L.score4 = conv_pack(cur, 1, 1, 0, c4, L.g.Cp, 128, es);
  L.score3 = conv_pack(cur, 1, 1, 0, c3, L.g.Cp, 128, es);
  const size_t frag = (size_t)16 * L.g.G * L.g.MT * 64;  // 16 bytes per lane per (group, class tile)
  L.up5 = take(cur, fcn32 ? 0 : frag * 4);
  L.up4 = take(cur, fcn32 ? 0 : frag * 4);
  L.up3 = take(cur, frag * (fcn32 ? 1024 : 64));  // fcn_32: 32x32 phases of the 64x64 kernel
  L.total = cur;
  return L;
}

// dst[o][(ky*kw+kx)*cin + c] = src[ky][kx][c][o]   (HWIO -> OHWI rows), zero rows for o >= cout
__device__ __forceinline__ void put(float* d, size_t i, float v) { d[i] = v; }
__device__ __forceinline__ void put(unsigned short* d, size_t i, float v) {
  d[i] = __builtin_bit_cast(unsigned short, (__bf16)v);  // round to nearest even
}

template <typename T>
__global__ void pack_conv_kernel(const float* __restrict__ src, T* __restrict__ dst, int kh, int kw, int cin,
                                 int cout, int coutpad) {
  const size_t K = (size_t)kh * kw * cin;
  const size_t total = K * coutpad;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t o = i / K, k = i % K;
    put(dst, i, (o < (size_t)cout) ? src[k * cout + o] : 0.f);
  }
}

// scale/shift of conv -> (BN) : y = conv*scale + shift
//   with BN:  scale = gamma/sqrt(var+eps), shift = (bias-mean)*scale + beta
//   without:  scale = 1, shift = bias
__global__ void pack_affine_kernel(flm_conv_params p, float* __restrict__ scale, float* __restrict__ shift, int cout,
                                   int coutpad) {
  const int o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= coutpad) return;
  if (o >= cout) {
    scale[o] = 0.f;
    shift[o] = 0.f;
    return;
  }
  const double b = p.bias ? (double)p.bias[o] : 0.0;
  if (p.gamma) {
    const double s = (double)p.gamma[o] / sqrt((double)p.var[o] + (double)kBnEps);
    scale[o] = (float)s;
    shift[o] = (float)((b - (double)p.mean[o]) * s + (double)p.beta[o]);
  } else {
    scale[o] = 1.f;
    shift[o] = (float)b;
  }
}

// enc1: dst[o][k], k = ky*9 + kx*3 + c, padded to 32
__global__ void pack_enc1_kernel(const float* __restrict__ src /*[3][3][3][64]*/, float* __restrict__ dst) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 64 * 32) return;
  const int o = i >> 5, k = i & 31;
  dst[i] = (k < 27) ? src[k * 64 + o] : 0.f;
}

// Transposed conv (kernel = 2s, stride s), Keras layout src[a][b][o][c] with a,b in [0,2s).
// Output pixel (s*i0+a0, s*j0+b0) = sum_{di,dj in {0,1}} sum_c x[i0-di][j0-dj][c] * src[a0+s*di][b0+s*dj][o][c].
// Fragment order: dst[phase=a0*s+b0][g][mt][lane][e], lane = (r = lane&15, q = lane>>4), EPL = 4 (fp32) or
// 8 (bf16) elements per lane:  class o = 16*mt + r, k = 4*EPL*g + EPL*q + e, tap = k / Cp = 2*di+dj, c = k % Cp.
// The 68-class kernels (flm_convt.hip, C68) spread classes 64..67 over the last tile's rows 0, 4, 8, 12: in the
// MFMA result row 4q + e belongs to lane group q, so every lane then holds 17 classes (16 + one) instead of lane
// group 0 holding 20 and the others 16 + 4 dead values.  With convt_share_layout(g, s) those four rows carry the
// four phases of a group instead (see flm_convt.hip).
template <typename T>
__global__ void pack_convt_kernel(const float* __restrict__ src, T* __restrict__ dst, int s, ConvTGeom g) {
  constexpr int EPL = 16 / (int)sizeof(T);
  const size_t per_phase = (size_t)g.G * g.MT * 64 * EPL;
  const size_t total = per_phase * s * s;
  const int ks = 2 * s;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int phase = (int)(i / per_phase);
    size_t rem = i % per_phase;
    const int e = (int)(rem % EPL);
    rem /= EPL;
    const int lane = (int)(rem & 63);
    rem >>= 6;
    const int mt = (int)(rem % g.MT);
    const int gg = (int)(rem / g.MT);
    const int r = lane & 15, q = lane >> 4;
    int o = 16 * mt + r;
    int a0 = phase / s, b0 = phase % s;
    if (g.C == 68 && (g.G == 9 || g.G == 17) && mt == 4) {
      if (convt_share_layout(g, s)) {
        // shared fifth tile: in a leader phase (b0 % 4 == 0) row 4q' + j holds class 64+q' of phase b0 + j.  The
        // other phases keep their own four classes in rows 4q' (as without sharing): the main launch does not
        // multiply that tile, the sampling launch (any phase, all five tiles, register 0) does
        if ((b0 & 3) == 0) {
          o = 64 + (r >> 2);
          b0 += r & 3;
        } else {
          o = (r & 3) == 0 ? 64 + (r >> 2) : g.C;
        }
      } else {
        o = (r & 3) == 0 ? 64 + (r >> 2) : g.C;  // g.C: no class
      }
    }
    const int k = 4 * EPL * gg + EPL * q + e;
    const int tap = k / g.Cp, c = k % g.Cp;
    float v = 0.f;
    if (o < g.C && c < g.C && tap < 4) {
      const int a = a0 + s * (tap >> 1), b = b0 + s * (tap & 1);
      v = src[(((size_t)a * ks + b) * g.C + o) * g.C + c];
    }
    put(dst, i, v);
  }
}

// Pixel-pair form of a 1x1 conv (see fcn8_pack_layout): dst[o'][c'] = src[c' % cin][o' % cout] when o' / cout == c' / cin.
__global__ void pack_pw_pair_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst, int cin, int cout,
                                    int coutpad) {
  const int K = 2 * cin;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= coutpad * K) return;
  const int o = i / K, c = i % K;
  float v = 0.f;
  if (o < 2 * cout && o / cout == c / cin) v = src[(size_t)(c % cin) * cout + (o % cout)];
  dst[i] = __builtin_bit_cast(unsigned short, (__bf16)v);
}
// the two halves share one set of per-channel scale / shift
__global__ void dup_affine_kernel(float* __restrict__ scale, float* __restrict__ shift, int cout, int coutpad) {
  const int o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o >= cout && o < 2 * cout) {
    scale[o] = scale[o - cout];
    shift[o] = shift[o - cout];
  } else if (o >= 2 * cout && o < coutpad) {
    scale[o] = 0.f;
    shift[o] = 0.f;
  }
}

static int pack_conv(hipStream_t s, const flm_conv_params& p, const ConvPack& c, char* blob, int dtype) {
  if (!p.kernel) {
    set_error("flm_fcn_pack: conv layer lacks its kernel");
    return FLM_ERR_ARG;
  }
  const size_t total = (size_t)c.coutpad * c.kh * c.kw * c.cin;
  const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  if (dtype == FLM_BF16)
    pack_conv_kernel<unsigned short><<<blocks, 256, 0, s>>>(p.kernel, (unsigned short*)(blob + c.w), c.kh, c.kw, c.cin,
                                                            c.cout, c.coutpad);
  else
    pack_conv_kernel<float><<<blocks, 256, 0, s>>>(p.kernel, (float*)(blob + c.w), c.kh, c.kw, c.cin, c.cout, c.coutpad);
  FLM_LAUNCH_CHECK("pack_conv_kernel");
  // pad columns of the score convs (cout = Cp > C) must stay zero: affine kernel writes zeros for o >= cout_real
  return FLM_OK;
}

int launch_pack_fcn(hipStream_t s, const flm_fcn_params& p, int C, const Fcn8Pack& L, char* blob) {
  const ArchSpec& A = L.spec;
  int n_params = 0;
  for (int i = 0; i < A.n_enc; ++i) n_params += enc_has_params(A.enc[i]);
  if (!p.enc || p.n_enc != n_params) {
    set_error("flm_fcn_pack: this architecture has %d encoder convs, got %d", n_params, p.n_enc);
    return FLM_ERR_ARG;
  }
  int pi = 0;
  for (int i = 0; i < A.n_enc; ++i) {
    const EncLayer& e = A.enc[i];
    if (!enc_has_params(e)) continue;
    flm_conv_params q = p.enc[pi++];
    const bool needs_bias = e.kind == ENC_FIRST3 || e.kind == ENC_CONV3 || e.kind == ENC_RN_CONV1 || e.kind == ENC_CONV;
    if (!q.kernel || (needs_bias && !q.bias)) {
      set_error("flm_fcn_pack: encoder conv %d lacks kernel or bias", pi);
      return FLM_ERR_ARG;
    }
    if (e.bn && (!q.gamma || !q.beta || !q.mean || !q.var)) {
      set_error("flm_fcn_pack: encoder conv %d lacks BatchNormalization tensors", pi);
      return FLM_ERR_ARG;
    }
    if (!e.bn) q.gamma = q.beta = q.mean = q.var = nullptr;
    if (i == 0) {  // first conv (3 input channels): its own kernel layouts
      if (e.kind == ENC_MB_CONV1) {  // Keras (3,3,3,32) is already [27][32]
        FLM_HIP(hipMemcpyAsync(blob + L.enc1_w, q.kernel, sizeof(float) * 27 * 32, hipMemcpyDeviceToDevice, s));
        pack_affine_kernel<<<1, 64, 0, s>>>(q, (float*)(blob + L.enc1_scale), (float*)(blob + L.enc1_shift), 32, 32);
      } else if (e.kind == ENC_RN_CONV1) {  // Keras (7,7,3,64) is already [147][64]
        FLM_HIP(hipMemcpyAsync(blob + L.enc1_w, q.kernel, sizeof(float) * 147 * 64, hipMemcpyDeviceToDevice, s));
        pack_affine_kernel<<<1, 64, 0, s>>>(q, (float*)(blob + L.enc1_scale), (float*)(blob + L.enc1_shift), 64, 64);
      } else {
        pack_enc1_kernel<<<cdiv(64 * 32, 256), 256, 0, s>>>(q.kernel, (float*)(blob + L.enc1_w));
        FLM_LAUNCH_CHECK("pack_enc1_kernel");
        pack_affine_kernel<<<1, 64, 0, s>>>(q, (float*)(blob + L.enc1_scale), (float*)(blob + L.enc1_shift), 64, 64);
      }
      FLM_LAUNCH_CHECK("pack_affine_kernel");
      continue;
    }
    if (e.kind == ENC_MB_DW) {  // Keras depthwise kernel (3,3,C,1) is already [9][C]
      FLM_HIP(hipMemcpyAsync(blob + L.enc[i].w, q.kernel, sizeof(float) * 9 * e.cin, hipMemcpyDeviceToDevice, s));
      pack_affine_kernel<<<cdiv(e.cin, 256), 256, 0, s>>>(q, (float*)(blob + L.enc[i].scale),
                                                          (float*)(blob + L.enc[i].shift), e.cin, e.cin);
      FLM_LAUNCH_CHECK("pack_affine_kernel");
      continue;
    }
    if (e.kind == ENC_MB_PW && L.enc[i].cin != e.cin) {  // pixel-pair form (bf16, 32 input channels)
      if (!q.kernel) {
        set_error("flm_fcn_pack: conv layer lacks its kernel");
        return FLM_ERR_ARG;
      }
      const int total = L.enc[i].coutpad * 2 * e.cin;
      pack_pw_pair_kernel<<<cdiv(total, 256), 256, 0, s>>>(q.kernel, (unsigned short*)(blob + L.enc[i].w), e.cin, e.cout,
                                                           L.enc[i].coutpad);
      pack_affine_kernel<<<cdiv(L.enc[i].coutpad, 256), 256, 0, s>>>(q, (float*)(blob + L.enc[i].scale),
                                                                     (float*)(blob + L.enc[i].shift), e.cout, e.cout);
      dup_affine_kernel<<<cdiv(L.enc[i].coutpad, 256), 256, 0, s>>>((float*)(blob + L.enc[i].scale),
                                                                    (float*)(blob + L.enc[i].shift), e.cout, L.enc[i].coutpad);
      FLM_LAUNCH_CHECK("pack_pw_pair_kernel");
      continue;
    }
    int rc = pack_conv(s, q, L.enc[i], blob, L.dtype);
    if (rc) return rc;
    pack_affine_kernel<<<cdiv(L.enc[i].coutpad, 256), 256, 0, s>>>(q, (float*)(blob + L.enc[i].scale),
                                                                   (float*)(blob + L.enc[i].shift), L.enc[i].cout,
                                                                   L.enc[i].coutpad);
    FLM_LAUNCH_CHECK("pack_affine_kernel");
  }
  struct Item { const flm_conv_params* p; const ConvPack* c; int cout_real; };
  const Item items[5] = {{&p.fc6, &L.fc6, kFc}, {&p.fc7, &L.fc7, kFc}, {&p.score5, &L.score5, C},
                         {&p.score4, &L.score4, C}, {&p.score3, &L.score3, C}};
  for (int ii = 0; ii < (A.fcn32 ? 3 : 5); ++ii) {
    const Item& it = items[ii];
    // Keras kernels of the score convs have C output columns; the packed rows C..coutpad-1 are zero.
    ConvPack c = *it.c;
    c.cout = it.cout_real;
    int rc = pack_conv(s, *it.p, c, blob, L.dtype);
    if (rc) return rc;
    flm_conv_params q = *it.p;
    q.gamma = q.beta = q.mean = q.var = nullptr;
    pack_affine_kernel<<<cdiv(c.coutpad, 256), 256, 0, s>>>(q, (float*)(blob + c.scale), (float*)(blob + c.shift),
                                                            it.cout_real, c.coutpad);
    FLM_LAUNCH_CHECK("pack_affine_kernel");
  }
  if (A.fcn32) {  // fcn.py:145-146: one Conv2DTranspose(C, 64x64, stride 32), passed in the up3 slot
    if (!p.up3) {
      set_error("flm_fcn32_pack: transposed-conv kernel missing");
      return FLM_ERR_ARG;
    }
    if (L.dtype == FLM_BF16)
      pack_convt_kernel<unsigned short><<<4096, 256, 0, s>>>(p.up3, (unsigned short*)(blob + L.up3), 32, L.g);
    else
      pack_convt_kernel<float><<<4096, 256, 0, s>>>(p.up3, (float*)(blob + L.up3), 32, L.g);
    FLM_LAUNCH_CHECK("pack_convt_kernel");
    return FLM_OK;
  }
  if (!p.up5 || !p.up4 || !p.up3) {
    set_error("flm_fcn8_pack: transposed-conv kernels missing");
    return FLM_ERR_ARG;
  }
  if (L.dtype == FLM_BF16) {
    pack_convt_kernel<unsigned short><<<256, 256, 0, s>>>(p.up5, (unsigned short*)(blob + L.up5), 2, L.g);
    pack_convt_kernel<unsigned short><<<256, 256, 0, s>>>(p.up4, (unsigned short*)(blob + L.up4), 2, L.g);
    pack_convt_kernel<unsigned short><<<2048, 256, 0, s>>>(p.up3, (unsigned short*)(blob + L.up3), 8, L.g);
  } else {
    pack_convt_kernel<float><<<256, 256, 0, s>>>(p.up5, (float*)(blob + L.up5), 2, L.g);
    pack_convt_kernel<float><<<256, 256, 0, s>>>(p.up4, (float*)(blob + L.up4), 2, L.g);
    pack_convt_kernel<float><<<2048, 256, 0, s>>>(p.up3, (float*)(blob + L.up3), 8, L.g);
  }
  FLM_LAUNCH_CHECK("pack_convt_kernel");
  return FLM_OK;
}

}  // namespace flm
