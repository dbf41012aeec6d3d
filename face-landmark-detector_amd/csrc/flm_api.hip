// extern "C" entry points of libflm_hip.so (see include/flm.h) and the forward's launch sequence.
#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "flm_common.h"

namespace flm {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int hip_fail(hipError_t e, const char* what) {
  set_error("HIP error %d (%s) at %s", (int)e, hipGetErrorString(e), what);
  return FLM_ERR_HIP;
}

// ---- optional per-launch timing (bench.py): hipEvent pairs around every launch of the forward --------
// Off by default; when enabled the forward records two events per layer on the caller's stream and
// never synchronises -- flm_profile_read() does, after the caller's own timed region has ended.
struct ProfRec {
  const char* name;
  hipEvent_t a, b;
};
static ProfRec* g_prof = nullptr;
static int g_prof_cap = 0, g_prof_n = 0;
static char g_prof_filter[32] = "";  // non-empty: only launches of this layer are bracketed

struct ProfScope {
  hipStream_t s;
  ProfRec* r;
  ProfScope(hipStream_t st, const char* name) : s(st), r(nullptr) {
    if (g_prof && g_prof_n < g_prof_cap && (!g_prof_filter[0] || !strcmp(g_prof_filter, name))) {
      r = &g_prof[g_prof_n++];
      r->name = name;
      (void)hipEventRecord(r->a, s);
    }
  }
  ~ProfScope() {
    if (r) (void)hipEventRecord(r->b, s);
  }
};

// Landmark mode without the probability tensor (flm_convt.hip): for the 68-class FCN-8 kernels, top-n decode
// with n <= 32 (the threshold is the n-th of a face's 144..288 sampled maxima; list capacities grow with n), maps
// below 2^17 pixels.  On by default for both types: bf16 batch
// 512 saves the tensor's write + re-read (up3 + decode 6.2 -> 3.5 ms); fp32 batch 64 gains 1.5 % of the step (the
// fp32 up3 is bound by the matrix pipe and writes its map for free; the gain is the decode).  0 = off (tests, A/B).
// These three are per-call options (flm_forward_opts): they change the workspace layout, so they travel with the call
// and with its workspace query instead of living in process state.
struct CandOpts {
  int enable, sub, cap_div;
};
static bool resolve_opts(const flm_forward_opts* o, CandOpts* c) {
  c->enable = 1;
  c->sub = 0;
  c->cap_div = 1;
  if (!o) return true;
  if (o->struct_size < sizeof(flm_forward_opts)) {
    set_error("flm_forward_opts: struct_size %u is smaller than this library's %zu (initialise with flm_forward_opts_init)",
              o->struct_size, sizeof(flm_forward_opts));
    return false;
  }
  if (o->candidate_sub_phases < 0 || o->candidate_sub_phases > 16 || o->candidate_cap_div < 1) {
    set_error("flm_forward_opts: candidate_sub_phases must be in [0,16] and candidate_cap_div >= 1");
    return false;
  }
  c->enable = o->landmark_candidates != 0;
  c->sub = o->candidate_sub_phases;
  c->cap_div = o->candidate_cap_div;
  return true;
}
// More sampled phases cost 1/64 of up3 each and tighten the threshold: the key lists shrink about in proportion.  At
// n = 4 the lists are short anyway (4 phases: 8-10 k keys per face); at n >= 16 their merge costs more than the extra
// phases (batch 512 bf16, n = 25: 12.1 ms with 4 phases, 11.5 with 8; tools/ab_sub.py).
// In fp32 a sampled phase costs sixteen times the matrix time it costs in bf16 while a key costs the same: at n <= 8 two
// phases do (batch 64, n = 4: 8.335 / 8.318 / 8.29 ms per step with 4 / 3 / 2 phases, 9.6 k / 13.4 k / 20.2 k keys per face of
// the 69.6 k the lists hold; bf16 batch 512: 7.76-7.93 / 7.84-8.15 / 8.01-8.03).
static int cand_sub_for(const CandOpts& c, int n_points, bool bf16) {
  if (c.sub > 0) return c.sub;
  return n_points <= 8 ? (bf16 ? 4 : 2) : (n_points <= 15 ? 6 : 8);
}
static bool landmark_candidates_enabled(const CandOpts& c, const ConvTGeom& g, int fcn32, int decode_mode, int n_points,
                                        int oh, int ow) {
  return c.enable && !fcn32 && decode_mode == FLM_DECODE_TOPN && n_points >= 1 && n_points <= 32 &&
         convt_candidates_supported(g) && (long long)oh * ow < (1 << 17);
}

struct EncNames {
  char s[kMaxEnc][8];
  EncNames() {
    for (int i = 0; i < kMaxEnc; ++i) snprintf(s[i], sizeof(s[i]), "enc%d", i + 1);
  }
};

static size_t take(size_t& cur, size_t bytes) {
  size_t o = cur;
  cur = align_up(cur + bytes, 256);
  return o;
}

Fcn8Ws fcn8_ws_layout(int n, int h, int w, int C, int dtype, int out_mode, int decode_mode, int n_points,
                      int arch, const flm_forward_opts* opts) {
  Fcn8Ws W;
  CandOpts co;
  if (!resolve_opts(opts, &co)) {  // callers validate first (check_opts); an invalid struct sizes nothing
    W = Fcn8Ws();
    W.total = 0;
    return W;
  }
  const ConvTGeom g = convt_geom(C, dtype);
  const ArchSpec A = arch_spec(arch);
  const size_t es = dtype == FLM_BF16 ? 2 : 4;  // encoder activations, fc6, fc7 are stored in the operand type
  size_t cur = 0;
  int hs[kMaxEnc], wsz[kMaxEnc];
  enc_dims(A, h, w, hs, wsz);
  for (int i = 0; i < A.n_enc; ++i) W.act[i] = take(cur, es * (size_t)n * hs[i] * wsz[i] * A.enc[i].cout);
  for (int k = 0; k < 5; ++k) W.f[k] = W.act[A.f_idx[k]];
  const int h5 = h / 32, w5 = w / 32, h4 = h / 16, w4 = w / 16, h3 = h / 8, w3 = w / 8;
  W.fc6 = take(cur, es * (size_t)n * h5 * w5 * kFc);
  W.fc7 = take(cur, es * (size_t)n * h5 * w5 * kFc);
  W.score5 = take(cur, sizeof(float) * (size_t)n * h5 * w5 * g.Cp);
  W.fuse4 = take(cur, sizeof(float) * (size_t)n * h4 * w4 * g.Cp);
  W.seg = take(cur, sizeof(float) * (size_t)n * h3 * w3 * g.Cp);
  // split-K partial sums: the score convs (8 slices of [n*h5*w5][Cp]) and fc6 / fc7 while they have at most 64
  // tiles (8 slices of [<= 256 rows][4096])
  W.splitk_bytes = sizeof(float) * 8 * (size_t)n * h5 * w5 * g.Cp;
  {
    const size_t rows = (size_t)n * h5 * w5 < 256 ? (size_t)n * h5 * w5 : 256;
    const size_t fc_bytes = sizeof(float) * 8 * rows * kFc;
    if (fc_bytes > W.splitk_bytes) W.splitk_bytes = fc_bytes;
  }
  if (n <= 4) {  // a 3x3 layer on the 1/4-resolution grid with 256 outputs (vanilla enc3, fp32) in 4 slices, 4 faces
    const size_t enc3_bytes = sizeof(float) * 4 * (size_t)4 * (h / 4) * (w / 4) * 256;
    if (enc3_bytes > W.splitk_bytes) W.splitk_bytes = enc3_bytes;
  }
  W.splitk = take(cur, W.splitk_bytes);
  W.oh = h + (A.fcn32 ? 32 : 8);  // (h/32 - 1)*32 + 64 (fcn.py:145) vs (h/8 - 1)*8 + 16 (fcn.py:121)
  W.ow = w + (A.fcn32 ? 32 : 8);
  W.probs = SIZE_MAX;
  W.decode = SIZE_MAX;
  W.sub = W.tau = W.cand = W.cand_cnt = SIZE_MAX;
  W.cand_cap = 0;
  if (out_mode == FLM_OUT_LANDMARKS) {
    W.probs = take(cur, sizeof(float) * (size_t)n * W.oh * W.ow * C);
    W.decode = take(cur, decode_ws_bytes(n, W.oh, W.ow, C, decode_mode, n_points));
    if (landmark_candidates_enabled(co, g, A.fcn32, decode_mode, n_points, W.oh, W.ow)) {
      const int h3 = h / 8, w3 = w / 8;
      W.sub = take(cur, sizeof(unsigned) * (size_t)n * convt_sample_slots(g, h3, w3, cand_sub_for(co, n_points, g.bf16 != 0)) * 16 * g.MT);  // sampled maxima
      W.tau = take(cur, sizeof(float) * (size_t)n * C);
      // expected keys per class: the n-th of 1/64 of the pixels ranks about 64*n-th overall; x4 head room
      // (never below one 64-key block: a huge cap_div then still takes the documented overflow fallback instead of
      // sizing an empty list, which the candidate launch rejects)
      W.cand_cap = (int)align_up((size_t)C * 64 * n_points * 4 / co.cap_div, 64);
      if (W.cand_cap < 64) W.cand_cap = 64;
      W.cand = take(cur, sizeof(unsigned long long) * (size_t)n * W.cand_cap);
      W.cand_cnt = take(cur, sizeof(unsigned) * ((size_t)n + 1));
    }
  }
  W.total = cur;
  return W;
}

static int check_fcn8_shape(int n, int h, int w, int C, int dtype) {
  if (dtype != FLM_F32 && dtype != FLM_BF16) {
    set_error("fcn8: unknown dtype %d (FLM_F32 = 0, FLM_BF16 = 1)", dtype);
    return FLM_ERR_UNSUPPORTED;
  }
  if (n <= 0 || h <= 0 || w <= 0 || (h % 32) || (w % 32)) {
    set_error("fcn8: input must be [N>0, H, W, 3] with H and W multiples of 32 (got n=%d h=%d w=%d)", n, h, w);
    return FLM_ERR_SHAPE;
  }
  if (C < 1 || C > kMaxClasses) {
    set_error("fcn8: n_classes must be in [1,%d] (got %d)", kMaxClasses, C);
    return FLM_ERR_SHAPE;
  }
  // (factors bounded first: the product below then fits 64 bits -- found by the UBSan sweep, tests/test_abi_sanitized.py)
  if (n > (1 << 24) || h > (1 << 15) || w > (1 << 15) || (long long)n * (h + 32) * (w + 32) * C >= (1ll << 40)) {
    set_error("fcn8: batch too large (n=%d h=%d w=%d: at most 2^40 output values per call)", n, h, w);
    return FLM_ERR_SHAPE;
  }
  return FLM_OK;
}

static int conv_layer(hipStream_t s, const char* blob, const ConvPack& c, const void* x, void* y, int n, int h,
                      int w, int relu, int pool, int posmajor, int dtype, int out_f32 = 0,
                      float* splitk_ws = nullptr, size_t splitk_bytes = 0, int stride = 1,
                      const void* res = nullptr) {
  IgemmDesc d;
  d.bf16 = dtype == FLM_BF16;
  d.out_f32 = out_f32;
  d.stride = stride;
  d.res = res;
  d.splitk_ws = splitk_ws;
  d.splitk_ws_bytes = splitk_bytes;
  d.x = x;
  d.wt = blob + c.w;
  d.scale = reinterpret_cast<const float*>(blob + c.scale);
  d.shift = reinterpret_cast<const float*>(blob + c.shift);
  d.y = y;
  d.n = n; d.h = h; d.w = w; d.cin = c.cin;
  d.cout = c.cout; d.coutpad = c.coutpad; d.ldc = c.cout;
  d.kh = c.kh; d.kw = c.kw; d.pad = c.pad;
  d.relu = relu; d.pool = pool; d.posmajor = posmajor;
  return launch_igemm(s, d);
}

}  // namespace flm

using namespace flm;

extern "C" {

int flm_abi_version(void) { return FLM_ABI_VERSION; }

int flm_debug_query(const char* key, int arg) {
  if (key && !strcmp(key, "igemm_occupancy")) return igemm_occupancy((size_t)arg);
  return -1;
}

int flm_set_tuning(const char* key, int value) {
  if (!key) return FLM_ERR_ARG;
  if (!strcmp(key, "none")) return FLM_OK;
  if (!strcmp(key, "bf16_big_tiles")) {  // 256-row bf16 implicit-GEMM tiles on/off (A/B runs of tools/tune.py)
    flm::igemm_bf16_big_enable(value);
    return FLM_OK;
  }
  if (!strcmp(key, "f32_two_level")) {  // fp32 implicit GEMMs: two-level accumulation (1) or one chain per output (0)
    flm::igemm_f32_group(value ? 0 : -1);
    return FLM_OK;
  }
  if (!strcmp(key, "bf16_mfma16")) {  // 256x256 LDS-DMA tiles on v_mfma_f32_16x16x32_bf16 (1) or 32x32x16 (0)
    flm::igemm_bf16_big_m16(value);
    return FLM_OK;
  }
  if (!strcmp(key, "bf16_halo_mfma16")) {  // the halo-resident 3x3 kernel on v_mfma_f32_16x16x32_bf16 (1) or 32x32x16 (0)
    flm::conv3_halo_m16(value);
    return FLM_OK;
  }
  if (!strcmp(key, "bf16_lds_dma")) {  // 256x256 tiles: operands by buffer_load ... lds (1) or through registers (0)
    flm::igemm_bf16_big_dma(value);
    return FLM_OK;
  }
  if (!strcmp(key, "landmark_candidates") || !strcmp(key, "candidate_sub_phases") || !strcmp(key, "candidate_cap_div")) {
    set_error("flm_set_tuning: '%s' changes the workspace layout and is a per-call option now: pass flm_forward_opts to "
              "flm_fcn_workspace_bytes_opts / flm_fcn_forward_opts", key);
    return FLM_ERR_ARG;
  }
  if (!strcmp(key, "up3_cand8")) {  // candidate launch of up3: bit 0 = bf16 takes the 8-wave kernel (default 1), bit 2 = its
                                    // 4-wave shape; bit 1 (the fp32 form) is accepted and ignored since round 3
    if (value < 0 || value > 7) {
      set_error("flm_set_tuning: up3_cand8 must be in [0,7]");
      return FLM_ERR_ARG;
    }
    flm::convt_cand8_enable(value);
    return FLM_OK;
  }
  if (!strcmp(key, "up3_cand8_rows")) {  // phase rows per workgroup of that kernel: 0 automatic, else 1, 2, 4 or 8
    if (value < 0 || value > 8 || (value & (value - 1))) {
      set_error("flm_set_tuning: up3_cand8_rows must be 0, 1, 2, 4 or 8");
      return FLM_ERR_ARG;
    }
    flm::convt_cand8_rows(value);
    return FLM_OK;
  }
  if (!strcmp(key, "up3_wreg")) {  // bf16 candidate launch of up3: 1 the weights-in-registers kernel (flm_up3_wreg.hip)
                                   // where its conditions hold, 0 (default) always the 8-wave kernel.  Same keys
    if (value < 0 || value > 1) {
      set_error("flm_set_tuning: up3_wreg must be 0 or 1");
      return FLM_ERR_ARG;
    }
    flm::convt_wreg_enable(value);
    return FLM_OK;
  }
  if (!strcmp(key, "bf16_score1x1")) {  // register-resident 1x1 classifier kernel for 256-channel inputs: 0 off, 1 on
    flm::score1x1_enable(value);
    return FLM_OK;
  }
  if (!strcmp(key, "bf16_fused_tail")) {  // score3 + up4 as one launch (flm_tail_bf16.hip): 0 off, 1 on.  Same bits
    flm::tail_fused_enable(value);
    return FLM_OK;
  }
  if (!strcmp(key, "decode_lds_dma")) {  // standalone decode of 68-landmark maps: tiles by LDS-DMA (1) or register prefetch (0)
    flm::decode_dma_enable(value);
    return FLM_OK;
  }
  if (!strcmp(key, "posmajor_order")) {  // fc6 at batches below a tile's rows: positions sharing a tile chosen by their taps (1) or in map order (0)
    flm::igemm_posperm_enable(value);
    return FLM_OK;
  }
  if (!strcmp(key, "warp_rows")) {  // uint8 warp, destination width % 64 == 0: a wave per row segment, 2 rows per wave (1;
                                    // 4: four rows) or the pixel-list kernel (0)
    if (value != 0 && value != 1 && value != 4) {
      set_error("flm_set_tuning: warp_rows must be 0, 1 or 4");
      return FLM_ERR_ARG;
    }
    flm::warp_rows_enable(value);
    return FLM_OK;
  }
  if (!strcmp(key, "bf16_conv3_halo")) {  // halo-resident 3x3 kernel for 64-channel inputs: 0 off, 1 auto, 2 always
    flm::conv3_halo_enable(value);
    return FLM_OK;
  }
  if (!strcmp(key, "bf16_group_n")) {  // weight panels per tile group of the 256-row kernel (0: default)
    if (value < 0 || value > 32 || (value & (value - 1))) {
      set_error("flm_set_tuning: bf16_group_n must be 0 or a power of two <= 32");
      return FLM_ERR_ARG;
    }
    flm::igemm_bf16_group_n(value);
    return FLM_OK;
  }
  set_error("flm_set_tuning: unknown key '%s'", key);
  return FLM_ERR_ARG;
}

int flm_profile_enable(int max_records) {
  if (g_prof) return FLM_OK;
  if (max_records <= 0 || max_records > (1 << 20)) {
    set_error("flm_profile_enable: bad record count");
    return FLM_ERR_ARG;
  }
  g_prof = new ProfRec[max_records];
  for (int i = 0; i < max_records; ++i) {
    FLM_HIP(hipEventCreate(&g_prof[i].a));
    FLM_HIP(hipEventCreate(&g_prof[i].b));
  }
  g_prof_cap = max_records;
  g_prof_n = 0;
  return FLM_OK;
}

int flm_profile_reset(void) {
  g_prof_n = 0;
  return FLM_OK;
}

int flm_profile_read(int index, char* name_out, int name_cap, float* ms_out) {
  if (!g_prof || index < 0 || index >= g_prof_n) return 1;  // past the end
  ProfRec& r = g_prof[index];
  FLM_HIP(hipEventSynchronize(r.b));
  FLM_HIP(hipEventElapsedTime(ms_out, r.a, r.b));
  if (name_out && name_cap > 0) {
    strncpy(name_out, r.name, name_cap - 1);
    name_out[name_cap - 1] = 0;
  }
  return FLM_OK;
}

int flm_profile_disable(void) {
  if (!g_prof) return FLM_OK;
  for (int i = 0; i < g_prof_cap; ++i) {
    (void)hipEventDestroy(g_prof[i].a);
    (void)hipEventDestroy(g_prof[i].b);
  }
  delete[] g_prof;
  g_prof = nullptr;
  g_prof_cap = g_prof_n = 0;
  return FLM_OK;
}
int flm_profile_filter(const char* layer) {
  if (layer && strlen(layer) >= sizeof(g_prof_filter)) {
    set_error("flm_profile_filter: layer name too long");
    return FLM_ERR_ARG;
  }
  strcpy(g_prof_filter, layer ? layer : "");
  return FLM_OK;
}
const char* flm_last_error(void) { return g_err; }

static size_t packed_bytes_impl(int n_classes, int dtype, int arch) {
  if ((dtype != FLM_F32 && dtype != FLM_BF16) || n_classes < 1 || n_classes > kMaxClasses) return 0;
  if (!arch_spec(arch).valid) return 0;
  return fcn8_pack_layout(n_classes, dtype, arch).total;
}
size_t flm_fcn_packed_bytes(int arch, int n_classes, int dtype) { return packed_bytes_impl(n_classes, dtype, arch); }
size_t flm_fcn8_packed_bytes(int n_classes, int dtype) { return packed_bytes_impl(n_classes, dtype, FLM_ARCH_FCN8); }
size_t flm_fcn32_packed_bytes(int n_classes, int dtype) { return packed_bytes_impl(n_classes, dtype, FLM_ARCH_FCN32); }

static int pack_impl(flm_stream_t stream, const flm_fcn_params* p, int n_classes, int dtype, void* packed_dev,
                     size_t packed_bytes, int arch) {
  if (!arch_spec(arch).valid) {
    set_error("flm_fcn_pack: unknown architecture %d", arch);
    return FLM_ERR_ARG;
  }
  if (!p || !packed_dev) {
    set_error("flm_fcn8_pack: null argument");
    return FLM_ERR_ARG;
  }
  if (dtype != FLM_F32 && dtype != FLM_BF16) {
    set_error("flm_fcn8_pack: unknown dtype %d", dtype);
    return FLM_ERR_UNSUPPORTED;
  }
  if (n_classes < 1 || n_classes > kMaxClasses) {
    set_error("flm_fcn8_pack: n_classes must be in [1,%d]", kMaxClasses);
    return FLM_ERR_SHAPE;
  }
  const Fcn8Pack L = fcn8_pack_layout(n_classes, dtype, arch);
  if (packed_bytes < L.total) {
    set_error("flm_fcn8_pack: packed buffer too small (%zu < %zu)", packed_bytes, L.total);
    return FLM_ERR_WORKSPACE;
  }
  return launch_pack_fcn(static_cast<hipStream_t>(stream), *p, n_classes, L, static_cast<char*>(packed_dev));
}
static flm_fcn_params from_fcn8(const flm_fcn8_params* p) {
  flm_fcn_params q;
  q.enc = p->enc;
  q.n_enc = 5;
  q.fc6 = p->fc6; q.fc7 = p->fc7; q.score5 = p->score5; q.score4 = p->score4; q.score3 = p->score3;
  q.up5 = p->up5; q.up4 = p->up4; q.up3 = p->up3;
  return q;
}
int flm_fcn_pack(flm_stream_t stream, int arch, const flm_fcn_params* p, int n_classes, int dtype, void* packed_dev,
                 size_t packed_bytes) {
  return pack_impl(stream, p, n_classes, dtype, packed_dev, packed_bytes, arch);
}
int flm_fcn8_pack(flm_stream_t stream, const flm_fcn8_params* p, int n_classes, int dtype, void* packed_dev,
                  size_t packed_bytes) {
  if (!p) {
    set_error("flm_fcn8_pack: null argument");
    return FLM_ERR_ARG;
  }
  const flm_fcn_params q = from_fcn8(p);
  return pack_impl(stream, &q, n_classes, dtype, packed_dev, packed_bytes, FLM_ARCH_FCN8);
}
int flm_fcn32_pack(flm_stream_t stream, const flm_fcn8_params* p, int n_classes, int dtype, void* packed_dev,
                   size_t packed_bytes) {
  if (!p) {
    set_error("flm_fcn32_pack: null argument");
    return FLM_ERR_ARG;
  }
  const flm_fcn_params q = from_fcn8(p);
  return pack_impl(stream, &q, n_classes, dtype, packed_dev, packed_bytes, FLM_ARCH_FCN32);
}

size_t flm_fcn8_workspace_bytes(int n, int h, int w, int n_classes, int dtype, int out_mode, int decode_mode,
                                int n_points) {
  if (check_fcn8_shape(n, h, w, n_classes, dtype)) return 0;
  return fcn8_ws_layout(n, h, w, n_classes, dtype, out_mode, decode_mode, n_points, FLM_ARCH_FCN8).total;
}
size_t flm_fcn32_workspace_bytes(int n, int h, int w, int n_classes, int dtype, int out_mode, int decode_mode,
                                 int n_points) {
  if (check_fcn8_shape(n, h, w, n_classes, dtype)) return 0;
  return fcn8_ws_layout(n, h, w, n_classes, dtype, out_mode, decode_mode, n_points, FLM_ARCH_FCN32).total;
}
size_t flm_fcn_workspace_bytes_opts(int arch, int n, int h, int w, int n_classes, int dtype, int out_mode,
                                    int decode_mode, int n_points, const flm_forward_opts* opts) {
  if (!arch_spec(arch).valid || check_fcn8_shape(n, h, w, n_classes, dtype)) return 0;
  return fcn8_ws_layout(n, h, w, n_classes, dtype, out_mode, decode_mode, n_points, arch, opts).total;
}
size_t flm_fcn_workspace_bytes(int arch, int n, int h, int w, int n_classes, int dtype, int out_mode, int decode_mode,
                               int n_points) {
  return flm_fcn_workspace_bytes_opts(arch, n, h, w, n_classes, dtype, out_mode, decode_mode, n_points, nullptr);
}
void flm_forward_opts_init(flm_forward_opts* opts) {
  if (!opts) return;
  opts->struct_size = (uint32_t)sizeof(flm_forward_opts);
  opts->landmark_candidates = 1;
  opts->candidate_sub_phases = 0;
  opts->candidate_cap_div = 1;
}

int64_t flm_fcn8_workspace_offset(const char* name, int n, int h, int w, int n_classes, int dtype, int out_mode,
                                  int decode_mode, int n_points) {
  return flm_fcn8_workspace_offset_opts(name, n, h, w, n_classes, dtype, out_mode, decode_mode, n_points, nullptr);
}
int64_t flm_fcn8_workspace_offset_opts(const char* name, int n, int h, int w, int n_classes, int dtype, int out_mode,
                                       int decode_mode, int n_points, const flm_forward_opts* opts) {
  if (!name || check_fcn8_shape(n, h, w, n_classes, dtype)) return -1;
  const Fcn8Ws W = fcn8_ws_layout(n, h, w, n_classes, dtype, out_mode, decode_mode, n_points, FLM_ARCH_FCN8, opts);
  if (W.total == 0) return -1;
  if (name[0] == 'f' && name[1] >= '1' && name[1] <= '5' && name[2] == 0) return (int64_t)W.f[name[1] - '1'];
  if (!strcmp(name, "cand_sub")) return W.sub == SIZE_MAX ? -1 : (int64_t)W.sub;
  if (!strcmp(name, "cand_tau")) return W.tau == SIZE_MAX ? -1 : (int64_t)W.tau;
  if (!strcmp(name, "cand_keys")) return W.cand == SIZE_MAX ? -1 : (int64_t)W.cand;
  if (!strcmp(name, "cand_cnt")) return W.cand_cnt == SIZE_MAX ? -1 : (int64_t)W.cand_cnt;
  if (!strcmp(name, "cand_cap")) return W.cand == SIZE_MAX ? -1 : (int64_t)W.cand_cap;
  if (!strcmp(name, "fc6")) return (int64_t)W.fc6;
  if (!strcmp(name, "fc7")) return (int64_t)W.fc7;
  if (!strcmp(name, "score5")) return (int64_t)W.score5;
  if (!strcmp(name, "fuse4")) return (int64_t)W.fuse4;
  if (!strcmp(name, "seg_feats")) return (int64_t)W.seg;
  if (!strcmp(name, "probs")) return W.probs == SIZE_MAX ? -1 : (int64_t)W.probs;
  return -1;
}

static int forward_impl(flm_stream_t stream, const void* packed_dev, const void* x_dev, int in_format, int n, int h,
                        int w, int C, int dtype, int out_mode, int decode_mode, int n_points, float thresh,
                        void* out_dev, void* workspace_dev, size_t workspace_bytes, int arch,
                        const flm_forward_opts* opts = nullptr) {
  const ArchSpec A = arch_spec(arch);
  CandOpts co;
  if (!resolve_opts(opts, &co)) return FLM_ERR_ARG;
  if (!A.valid) {
    set_error("flm_fcn_forward: unknown architecture %d", arch);
    return FLM_ERR_ARG;
  }
  const int fcn32 = A.fcn32;
  if (!packed_dev || !x_dev || !out_dev || !workspace_dev) {
    set_error("flm_fcn8_forward: null argument");
    return FLM_ERR_ARG;
  }
  int rc = check_fcn8_shape(n, h, w, C, dtype);
  if (rc) return rc;
  if (out_mode < FLM_OUT_PROBS || out_mode > FLM_OUT_LOGITS) {
    set_error("flm_fcn8_forward: unknown output mode %d", out_mode);
    return FLM_ERR_ARG;
  }
  const Fcn8Ws W = fcn8_ws_layout(n, h, w, C, dtype, out_mode, decode_mode, n_points, arch, opts);
  if (workspace_bytes < W.total) {
    set_error("flm_fcn8_forward: workspace too small (%zu < %zu)", workspace_bytes, W.total);
    return FLM_ERR_WORKSPACE;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  const Fcn8Pack L = fcn8_pack_layout(C, dtype, arch);
  const int bf = dtype == FLM_BF16;
  const char* blob = static_cast<const char*>(packed_dev);
  char* ws = static_cast<char*>(workspace_dev);
  void* f[5];
  for (int i = 0; i < 5; ++i) f[i] = ws + W.f[i];
  void* fc6 = ws + W.fc6;
  void* fc7 = ws + W.fc7;
  float* score5 = reinterpret_cast<float*>(ws + W.score5);
  float* fuse4 = reinterpret_cast<float*>(ws + W.fuse4);
  float* seg = reinterpret_cast<float*>(ws + W.seg);

  // encoder: vanilla (networks/fcn.py:10-51) or VGG16 (networks/vgg16.py:27-72)
  static const EncNames enc_name_table;  // "enc1".."enc64" (profile record labels); built once, thread-safe (C++11 statics)
  const char (*enc_names)[8] = enc_name_table.s;
  int hs[kMaxEnc], wsz[kMaxEnc];
  enc_dims(A, h, w, hs, wsz);
  {
    int h8 = hs[A.f_idx[2]], h16 = hs[A.f_idx[3]], h32 = hs[A.f_idx[4]];
    if (h8 != h / 8 || h16 != h / 16 || h32 != h / 32) {
      set_error("flm_fcn_forward: encoder grid %d/%d/%d does not match the decoder's H/8, H/16, H/32", h8, h16, h32);
      return FLM_ERR_SHAPE;
    }
  }
  for (int i = 0; i < A.n_enc; ++i) {
    const EncLayer& e = A.enc[i];
    const int src = e.src >= 0 ? e.src : i - 1;
    const void* xin = src >= 0 ? static_cast<const void*>(ws + W.act[src]) : x_dev;
    const int hi = src >= 0 ? hs[src] : h, wi = src >= 0 ? wsz[src] : w;
    void* yout = ws + W.act[i];
    const float* w0 = reinterpret_cast<const float*>(blob + (i == 0 ? L.enc1_w : L.enc[i].w));
    const float* sc0 = reinterpret_cast<const float*>(blob + (i == 0 ? L.enc1_scale : L.enc[i].scale));
    const float* sh0 = reinterpret_cast<const float*>(blob + (i == 0 ? L.enc1_shift : L.enc[i].shift));
    ProfScope ps(s, enc_names[i]);
    switch (e.kind) {
      case ENC_FIRST3:
        rc = launch_enc1(s, xin, in_format, n, hi, wi, w0, sc0, sh0, yout, bf, e.pool);
        break;
      case ENC_MB_CONV1:
        rc = launch_mb_conv1(s, xin, in_format, n, hi, wi, w0, sc0, sh0, yout, bf);
        break;
      case ENC_RN_CONV1:
        rc = launch_rn_conv1(s, xin, in_format, n, hi, wi, w0, sc0, sh0, yout, bf);
        break;
      case ENC_MAXPOOL3:
        rc = launch_maxpool3(s, xin, n, hi, wi, e.cin, yout, bf);
        break;
      case ENC_MB_DW:
        rc = launch_mb_depthwise(s, xin, n, hi, wi, e.cin, e.stride, w0, sc0, sh0, yout, bf);
        break;
      case ENC_CONV3:
        rc = conv_layer(s, blob, L.enc[i], xin, yout, n, hi, wi, /*relu*/ 1, e.pool, 0, dtype, 0,
                        reinterpret_cast<float*>(ws + W.splitk), W.splitk_bytes);
        break;
      case ENC_MB_PW:
        if (L.enc[i].cin != e.cin) {
          // pixel-pair form (flm_pack.hip): half as many "pixels", twice the channels.  A 1x1 conv sees a flat list
          // of pixels, so pairs may run across row ends: any even dimension can be the one that is halved
          int pn = n, ph = hi, pw = wi;
          if (!(pw & 1)) pw >>= 1;
          else if (!(ph & 1)) ph >>= 1;
          else if (!(pn & 1)) pn >>= 1;
          else {
            set_error("flm_fcn_forward: the paired pointwise conv needs an even number of pixels per batch");
            rc = FLM_ERR_SHAPE;
            break;
          }
          rc = conv_layer(s, blob, L.enc[i], xin, yout, pn, ph, pw, /*relu6*/ 2, 0, 0, dtype);
        } else {
          rc = conv_layer(s, blob, L.enc[i], xin, yout, n, hi, wi, /*relu6*/ 2, 0, 0, dtype);
        }
        break;
      case ENC_CONV:
        rc = conv_layer(s, blob, L.enc[i], xin, yout, n, hi, wi, e.relu, 0, 0, dtype, 0, nullptr, 0, e.stride,
                        e.res >= 0 ? ws + W.act[e.res] : nullptr);
        break;
      default:
        set_error("flm_fcn_forward: unknown encoder layer kind %d", e.kind);
        rc = FLM_ERR_ARG;
    }
    if (rc) return rc;
  }
  const int h5 = h / 32, w5 = w / 32, h4 = h / 16, w4 = w / 16, h3 = h / 8, w3 = w / 8;
  // head (fcn.py:98-103); Dropout is the identity at inference
  { ProfScope ps(s, "fc6");
  rc = conv_layer(s, blob, L.fc6, f[4], fc6, n, h5, w5, 1, 0, /*posmajor*/ 1, dtype, 0,
                  reinterpret_cast<float*>(ws + W.splitk), W.splitk_bytes); }
  if (rc) return rc;
  { ProfScope ps(s, "fc7");
  rc = conv_layer(s, blob, L.fc7, fc6, fc7, n, h5, w5, 1, 0, 0, dtype, 0,
                  reinterpret_cast<float*>(ws + W.splitk), W.splitk_bytes); }
  if (rc) return rc;
  { ProfScope ps(s, "score5");
  rc = conv_layer(s, blob, L.score5, fc7, score5, n, h5, w5, 0, 0, 0, dtype, /*out_f32*/ 1,
                  reinterpret_cast<float*>(ws + W.splitk), W.splitk_bytes); }
  if (rc) return rc;
  ConvTDesc t;
  t.g = L.g;
  t.n = n;
  if (!fcn32) {
  // skip branches: score4 on f4 -> fuse4 buffer, score3 on f3 -> seg buffer, then the transposed
  // convs add themselves onto those (crop keeps the top-left window, fcn.py:76-84)
  { ProfScope ps(s, "score4");
  rc = conv_layer(s, blob, L.score4, f[3], fuse4, n, h4, w4, 0, 0, 0, dtype, 1); }
  if (rc) return rc;
  // (bf16, 68 classes: score3 and up4 run as ONE launch after up5, flm_tail_bf16.hip -- same bits)
  const bool fuse_seg = bf && L.g.C == 68 && L.score3.cin == 256 && L.score3.kh == 1;
  int fused_seg = 0;
  if (!fuse_seg) {
  { ProfScope ps(s, "score3");
  rc = conv_layer(s, blob, L.score3, f[2], seg, n, h3, w3, 0, 0, 0, dtype, 1); }
  if (rc) return rc;
  }
  // up5 (fcn.py:104) + crop + Add (fcn.py:110-112), in place on fuse4
  t.x = score5; t.wf = blob + L.up5; t.skip = fuse4; t.y = fuse4;
  t.hi = h5; t.wi = w5; t.ho = h4; t.wo = w4; t.s = 2; t.ldy = L.g.Cp; t.epilogue = 0;
  { ProfScope ps(s, "up5");
  rc = launch_convt(s, t); }
  if (rc) return rc;
  // up4 (fcn.py:114) + crop + Add (fcn.py:118-119), in place on seg ("seg_feats")
  if (fuse_seg) {
    ProfScope ps(s, "seg_fused");
    fused_seg = launch_seg_fused_bf16(s, fuse4, blob + L.up4, f[2], blob + L.score3.w,
                                      reinterpret_cast<const float*>(blob + L.score3.scale),
                                      reinterpret_cast<const float*>(blob + L.score3.shift), seg, n, h4, w4, L.g.C, L.g.Cp,
                                      L.g.G, L.score3.cin, L.score3.coutpad);
    if (fused_seg < 0) return fused_seg;
  }
  if (fuse_seg && !fused_seg) {  // (knob off, or a shape the fused kernel leaves alone: the two-launch form)
    ProfScope ps(s, "score3");
    rc = conv_layer(s, blob, L.score3, f[2], seg, n, h3, w3, 0, 0, 0, dtype, 1);
    if (rc) return rc;
  }
  t.x = fuse4; t.wf = blob + L.up4; t.skip = seg; t.y = seg;
  t.hi = h4; t.wi = w4; t.ho = h3; t.wo = w3;
  if (!fused_seg) {
  { ProfScope ps(s, "up4");
  rc = launch_convt(s, t); }
  if (rc) return rc;
  }
  }  // !fcn32
  // last upsampling + softmax (networks/utils.py:30) / argmax (prediction.py:209):
  //   fcn_8 : Conv2DTranspose(16x16, s8) on seg_feats  (fcn.py:121)
  //   fcn_32: Conv2DTranspose(64x64, s32) on the 1x1 classifier output  (fcn.py:143-146)
  t.wf = blob + L.up3; t.skip = nullptr; t.ho = W.oh; t.wo = W.ow; t.ldy = C;
  if (fcn32) { t.x = score5; t.hi = h5; t.wi = w5; t.s = 32; }
  else { t.x = seg; t.hi = h3; t.wi = w3; t.s = 8; }
  if (out_mode == FLM_OUT_LOGITS || out_mode == FLM_OUT_PROBS) {
    t.y = out_dev;
    t.epilogue = (out_mode == FLM_OUT_PROBS) ? 1 : 0;
    if (out_mode == FLM_OUT_LOGITS && (C & 3)) {
      set_error("flm_fcn8_forward: FLM_OUT_LOGITS needs n_classes %% 4 == 0");
      return FLM_ERR_UNSUPPORTED;
    }
    ProfScope ps(s, "up3");
    return launch_convt(s, t);
  }
  if (out_mode == FLM_OUT_CLASSMAP) {
    t.y = out_dev;
    t.epilogue = 2;
    ProfScope ps(s, "up3");
    return launch_convt(s, t);
  }
  // landmarks (utils/metrics.py:102-109)
  float* probs = reinterpret_cast<float*>(ws + W.probs);
  const unsigned* gate = nullptr;
  if (W.cand != SIZE_MAX) {
    // top-n without the probability tensor (flm_convt.hip): thresholds from the phase-(0,0) sub-map, candidate
    // keys from the full launch, exact selection; then the materialising path below runs gated on the overflow flag
    float* sub = reinterpret_cast<float*>(ws + W.sub);
    float* tau = reinterpret_cast<float*>(ws + W.tau);
    unsigned* cnt = reinterpret_cast<unsigned*>(ws + W.cand_cnt);
    unsigned long long* cand = reinterpret_cast<unsigned long long*>(ws + W.cand);
    FLM_HIP(hipMemsetAsync(cnt, 0, sizeof(unsigned) * ((size_t)n + 1), s));
    ConvTDesc ts = t;
    ts.y = sub; ts.epilogue = 4; ts.sub = cand_sub_for(co, n_points, L.g.bf16 != 0);
    { ProfScope ps(s, "up3_sub");
    rc = launch_convt(s, ts); }
    if (rc) return rc;
    { ProfScope ps(s, "tau");
    rc = launch_cand_tau(s, reinterpret_cast<const unsigned*>(sub), n, convt_sample_slots(L.g, t.hi, t.wi, cand_sub_for(co, n_points, L.g.bf16 != 0)), 16 * L.g.MT, C,
                         n_points, tau); }
    if (rc) return rc;
    ConvTDesc tc = t;
    tc.y = nullptr; tc.epilogue = 3; tc.tau = tau; tc.cand = cand; tc.cand_cnt = cnt; tc.cand_cap = W.cand_cap;
    // (the probability region is written only by the gated fallback below, after this launch: its scratch until then)
    tc.scratch = probs; tc.scratch_bytes = sizeof(float) * (size_t)n * W.oh * W.ow * C;
    { ProfScope ps(s, "up3");
    rc = launch_convt(s, tc); }
    if (rc) return rc;
    { ProfScope ps(s, "decode");
    rc = launch_cand_merge(s, cand, cnt, n, W.ow, C, n_points, thresh, W.cand_cap, static_cast<double*>(out_dev)); }
    if (rc) return rc;
    gate = cnt + n;  // overflow flag: non-zero -> redo this batch through the probability tensor
    t.gate = gate;
  }
  t.y = probs;
  t.epilogue = 1;
  { ProfScope ps(s, gate ? "up3_fallback" : "up3");
  rc = launch_convt(s, t); }
  if (rc) return rc;
  ProfScope ps(s, gate ? "decode_fallback" : "decode");
  return launch_decode(s, probs, n, W.oh, W.ow, C, C, decode_mode, n_points, thresh,
                       static_cast<double*>(out_dev), ws + W.decode,
                       decode_ws_bytes(n, W.oh, W.ow, C, decode_mode, n_points), nullptr, gate);
}

int flm_fcn8_run_layer(flm_stream_t stream, const void* packed_dev, const char* layer, const void* x_dev,
                       void* y_dev, int n, int h, int w, int C, int dtype) {
  if (!packed_dev || !layer || !x_dev || !y_dev || n <= 0 || h <= 0 || w <= 0) {
    set_error("flm_fcn8_run_layer: bad argument");
    return FLM_ERR_ARG;
  }
  if ((dtype != FLM_F32 && dtype != FLM_BF16) || C < 1 || C > kMaxClasses) {
    set_error("flm_fcn8_run_layer: unsupported dtype/n_classes");
    return FLM_ERR_UNSUPPORTED;
  }
  const Fcn8Pack L = fcn8_pack_layout(C, dtype);
  const char* blob = static_cast<const char*>(packed_dev);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (!strncmp(layer, "enc", 3) && layer[3] >= '2' && layer[3] <= '5' && layer[4] == 0)
    return conv_layer(s, blob, L.enc[layer[3] - '1'], x_dev, y_dev, n, h, w, 1, 1, 0, dtype);
  if (!strcmp(layer, "fc6")) return conv_layer(s, blob, L.fc6, x_dev, y_dev, n, h, w, 1, 0, 1, dtype);
  if (!strcmp(layer, "fc7")) return conv_layer(s, blob, L.fc7, x_dev, y_dev, n, h, w, 1, 0, 0, dtype);
  if (!strcmp(layer, "score5")) return conv_layer(s, blob, L.score5, x_dev, y_dev, n, h, w, 0, 0, 0, dtype, 1);
  if (!strcmp(layer, "score4")) return conv_layer(s, blob, L.score4, x_dev, y_dev, n, h, w, 0, 0, 0, dtype, 1);
  if (!strcmp(layer, "score3")) return conv_layer(s, blob, L.score3, x_dev, y_dev, n, h, w, 0, 0, 0, dtype, 1);
  set_error("flm_fcn8_run_layer: unknown layer '%s'", layer);
  return FLM_ERR_ARG;
}

int flm_fcn8_forward(flm_stream_t stream, const void* packed_dev, const void* x_dev, int in_format, int n, int h,
                     int w, int C, int dtype, int out_mode, int decode_mode, int n_points, float thresh,
                     void* out_dev, void* workspace_dev, size_t workspace_bytes) {
  return forward_impl(stream, packed_dev, x_dev, in_format, n, h, w, C, dtype, out_mode, decode_mode, n_points, thresh,
                      out_dev, workspace_dev, workspace_bytes, FLM_ARCH_FCN8);
}
int flm_fcn_forward_opts(flm_stream_t stream, int arch, const void* packed_dev, const void* x_dev, int in_format, int n,
                         int h, int w, int C, int dtype, int out_mode, int decode_mode, int n_points, float thresh,
                         void* out_dev, void* workspace_dev, size_t workspace_bytes, const flm_forward_opts* opts) {
  return forward_impl(stream, packed_dev, x_dev, in_format, n, h, w, C, dtype, out_mode, decode_mode, n_points, thresh,
                      out_dev, workspace_dev, workspace_bytes, arch, opts);
}
int flm_fcn_forward(flm_stream_t stream, int arch, const void* packed_dev, const void* x_dev, int in_format, int n,
                    int h, int w, int C, int dtype, int out_mode, int decode_mode, int n_points, float thresh,
                    void* out_dev, void* workspace_dev, size_t workspace_bytes) {
  return forward_impl(stream, packed_dev, x_dev, in_format, n, h, w, C, dtype, out_mode, decode_mode, n_points, thresh,
                      out_dev, workspace_dev, workspace_bytes, arch);
}
int flm_fcn32_forward(flm_stream_t stream, const void* packed_dev, const void* x_dev, int in_format, int n, int h,
                      int w, int C, int dtype, int out_mode, int decode_mode, int n_points, float thresh,
                      void* out_dev, void* workspace_dev, size_t workspace_bytes) {
  return forward_impl(stream, packed_dev, x_dev, in_format, n, h, w, C, dtype, out_mode, decode_mode, n_points, thresh,
                      out_dev, workspace_dev, workspace_bytes, FLM_ARCH_FCN32);
}

int flm_preprocess(flm_stream_t stream, const uint8_t* img, int n, int h, int w, int norm, float* out) {
  if (!img || !out || n <= 0 || h <= 0 || w <= 0) {
    set_error("flm_preprocess: bad argument");
    return FLM_ERR_ARG;
  }
  return launch_preprocess(static_cast<hipStream_t>(stream), img, n, h, w, norm, out);
}

size_t flm_decode_workspace_bytes(int n, int h, int w, int l, int mode, int n_points) {
  if (n <= 0 || h <= 0 || w <= 0 || l <= 0) return 0;
  return decode_ws_bytes(n, h, w, l, mode, n_points);
}

int flm_decode(flm_stream_t stream, const float* hm, int n, int h, int w, int l, int mode, int n_points,
               float thresh, double* out, void* ws, size_t ws_bytes) {
  if (!hm || !out || !ws) {
    set_error("flm_decode: null argument");
    return FLM_ERR_ARG;
  }
  return launch_decode(static_cast<hipStream_t>(stream), hm, n, h, w, l, l, mode, n_points, thresh, out, ws,
                       ws_bytes);
}

int flm_similarity_from_landmarks(flm_stream_t stream, const double* lm, const double* tmpl, int n, int k,
                                  float* m) {
  if (!lm || !tmpl || !m) {
    set_error("flm_similarity_from_landmarks: null argument");
    return FLM_ERR_ARG;
  }
  return launch_similarity(static_cast<hipStream_t>(stream), lm, tmpl, n, k, 1.0, 1.0, m);
}

int flm_similarity_from_landmarks_scaled(flm_stream_t stream, const double* lm, const double* tmpl, int n, int k,
                                         double sx, double sy, float* m) {
  if (!lm || !tmpl || !m) {
    set_error("flm_similarity_from_landmarks_scaled: null argument");
    return FLM_ERR_ARG;
  }
  return launch_similarity(static_cast<hipStream_t>(stream), lm, tmpl, n, k, sx, sy, m);
}

int flm_warp_affine(flm_stream_t stream, const void* src, int src_is_u8, int n, int hs, int ws, const float* m,
                    float* dst, int hd, int wd) {
  if (!src || !m || !dst) {
    set_error("flm_warp_affine: null argument");
    return FLM_ERR_ARG;
  }
  return launch_warp(static_cast<hipStream_t>(stream), src, src_is_u8, n, hs, ws, m, dst, hd, wd);
}

int flm_crop_resize(flm_stream_t stream, const uint8_t* frame, int fh, int fw, const int32_t* boxes, int k,
                    uint8_t* out, int oh, int ow) {
  if (!frame || !boxes || !out) {
    set_error("flm_crop_resize: null argument");
    return FLM_ERR_ARG;
  }
  return launch_crop_resize(static_cast<hipStream_t>(stream), frame, fh, fw, boxes, k, out, oh, ow);
}

int flm_crop_resize_frames(flm_stream_t stream, const uint8_t* frames, size_t frame_stride, int nframes, int fh, int fw,
                           const int32_t* boxes, const int32_t* frame_idx, int k, uint8_t* out, int oh, int ow) {
  if (!frames || !boxes || !frame_idx || !out) {
    set_error("flm_crop_resize_frames: null argument");
    return FLM_ERR_ARG;
  }
  return launch_crop_resize(static_cast<hipStream_t>(stream), frames, fh, fw, boxes, k, out, oh, ow, frame_idx,
                            frame_stride, nframes);
}

}  // extern "C"
