/*
 * flm.h -- C ABI of the MI355X-native facial-landmark hot path (libflm_hip.so).
 *
 * The reference (sandyz1000/face-landmark-detector) is pure Python on
 * TensorFlow/Keras: it has no FFI, plugin or operator registry.  The boundary it
 * offers is a duck-typed model object plus a few functions; each entry point below
 * names the reference interface it replaces (paths relative to the reference root).
 * The Python host side (package `face-landmark-detector_amd`) binds these with
 * ctypes; INTEGRATION.md shows the binding a reference maintainer would add.
 *
 * Conventions
 *  - plain pointers and sizes only; every pointer named *_dev is DEVICE memory
 *    borrowed for the duration of the call; nothing is retained.
 *  - every launch goes to the caller's stream (hipStream_t passed as void*);
 *    no call synchronises the device or allocates memory.
 *  - scratch memory is a caller-owned workspace sized by the matching
 *    *_workspace_bytes query.
 *  - return value: 0 = FLM_OK, negative = error; flm_last_error() returns a
 *    thread-local message for the last failing call on this thread.
 *  - reentrancy: calls on different streams may run concurrently from different threads.
 *    Process state is limited to (1) the thread-local error string, (2) the A/B
 *    performance knobs of flm_set_tuning -- atomic integers read at launch time that
 *    never change results or memory layouts, (3) the measurement hook flm_profile_*
 *    (off by default; a measurement aid, NOT thread-safe).  Everything that changes
 *    results' provenance or the workspace layout is an argument (flm_forward_opts).
 *  - layouts are NHWC ("channels_last", networks/config.py:5) throughout.
 */
#ifndef FLM_H_
#define FLM_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FLM_ABI_VERSION 2

typedef void* flm_stream_t; /* hipStream_t */

enum flm_status {
  FLM_OK = 0,
  FLM_ERR_ARG = -1,         /* null pointer / bad enum */
  FLM_ERR_SHAPE = -2,       /* shape outside what the kernels cover */
  FLM_ERR_WORKSPACE = -3,   /* workspace too small */
  FLM_ERR_HIP = -4,         /* a HIP runtime call failed */
  FLM_ERR_UNSUPPORTED = -5  /* valid request this build does not implement */
};

/* Arithmetic of the conv stack.  FLM_F32: exact fp32 on v_mfma_f32_*_f32 (the parity path).
 * FLM_BF16: bf16 operands (weights and the activations f1..f5/fc6/fc7 stored as bf16), fp32
 * accumulation, BatchNorm/bias/softmax/decode in fp32 (BASELINE configs[2]); inputs and outputs of
 * flm_fcn8_forward keep the same types in both modes. */
enum flm_dtype { FLM_F32 = 0, FLM_BF16 = 1 };

/* Input formats of the forward. */
enum flm_input_format {
  FLM_IN_U8_BGR = 0,  /* raw cv2 crop, uint8 [N,H,W,3] BGR: the sub_mean preprocess of
                         get_image_array (data/generator.py:52-61) is fused into the
                         first conv's loader */
  FLM_IN_F32_RGB = 1  /* already preprocessed float32 [N,H,W,3], what model.predict
                         receives at prediction.py:208 */
};

/* What flm_fcn8_forward writes to `out_dev`. */
enum flm_output_mode {
  FLM_OUT_PROBS = 0,     /* float32 [N, H'*W', C]: model.predict (networks/utils.py:28-30) */
  FLM_OUT_CLASSMAP = 1,  /* int32 [N, H', W']: pr.argmax(axis=2) (prediction.py:209) */
  FLM_OUT_LANDMARKS = 2, /* float64 [N, C, 2] (x,y): transfer_target (utils/metrics.py:102-109) */
  FLM_OUT_LOGITS = 3     /* float32 [N, H', W', C] before the softmax (debug / tests) */
};

enum flm_decode_mode {
  FLM_DECODE_ALL = 0, /* n_points < 1: full-map weighted centroid (utils/metrics.py:58-64) */
  FLM_DECODE_TOPN = 1 /* weighted centroid of the n largest pixels (utils/metrics.py:66-77) */
};

enum flm_image_norm { /* imgNorm of get_image_array (data/generator.py:50-65) */
  FLM_NORM_SUB_MEAN = 0,
  FLM_NORM_SUB_AND_DIVIDE = 1,
  FLM_NORM_DIVIDE = 2
};

int flm_abi_version(void);
const char* flm_last_error(void);

/* ---- weights -------------------------------------------------------------------
 * Replaces keras `model.load_weights` (prediction.py:128).  Parameters arrive as
 * device tensors in the KERAS layouts and are repacked on the device into one blob
 * in the layouts the kernels read (OHWI rows for the implicit GEMMs, MFMA fragment
 * order for the transposed convs, BatchNorm folded to scale/shift with eps = 1e-3).
 */
typedef struct flm_conv_params {
  const float* kernel; /* Conv2D: HWIO (kh,kw,in,out) */
  const float* bias;   /* [out] */
  const float* gamma;  /* BatchNormalization, NULL for layers without BN */
  const float* beta;
  const float* mean;   /* moving_mean */
  const float* var;    /* moving_variance */
} flm_conv_params;

typedef struct flm_fcn8_params {
  flm_conv_params enc[5];  /* vanilla_encoder, networks/fcn.py:10-51: 3x3, F=64,128,256,256,256 */
  flm_conv_params fc6;     /* fcn.py:98  7x7x256x4096 */
  flm_conv_params fc7;     /* fcn.py:100 1x1x4096x4096 */
  flm_conv_params score5;  /* fcn.py:103 1x1x4096xC */
  flm_conv_params score4;  /* fcn.py:108 1x1x256xC on f4 */
  flm_conv_params score3;  /* fcn.py:117 1x1x256xC on f3 */
  const float* up5;        /* fcn.py:104 Conv2DTranspose (4,4,C,C) = (kh,kw,out,in), stride 2 */
  const float* up4;        /* fcn.py:114 (4,4,C,C), stride 2 */
  const float* up3;        /* fcn.py:121 (16,16,C,C), stride 8 */
} flm_fcn8_params;

size_t flm_fcn8_packed_bytes(int n_classes, int dtype);
int flm_fcn8_pack(flm_stream_t stream, const flm_fcn8_params* params_dev_ptrs, int n_classes,
                  int dtype, void* packed_dev, size_t packed_bytes);

/* ---- forward -------------------------------------------------------------------
 * Replaces `model.predict(x)` of the model built by fcn_8 + vanilla_encoder +
 * get_segmentation_model (networks/fcn.py:89-126, networks/utils.py:6-39;
 * call site prediction.py:208), optionally continued through the argmax of
 * prediction.py:209 or the decode of utils/metrics.py:102-109.
 * H and W must be multiples of 32; the output grid is H' = H+8, W' = W+8.
 */
size_t flm_fcn8_workspace_bytes(int n, int h, int w, int n_classes, int dtype, int out_mode,
                                int decode_mode, int n_points);
int flm_fcn8_forward(flm_stream_t stream, const void* packed_dev, const void* x_dev, int in_format,
                     int n, int h, int w, int n_classes, int dtype, int out_mode, int decode_mode,
                     int n_points, float thresh, void* out_dev, void* workspace_dev,
                     size_t workspace_bytes);
/* Byte offset inside the workspace of a named intermediate ("f1".."f5","fc6","fc7",
 * "score5","fuse4","seg_feats","probs"), or -1: lets tests compare layer by layer. */
int64_t flm_fcn8_workspace_offset(const char* name, int n, int h, int w, int n_classes, int dtype,
                                  int out_mode, int decode_mode, int n_points);

/* ---- fcn_32 (networks/fcn.py:129-150) ------------------------------------------------------------
 * Same encoder and head; no skip branches; one Conv2DTranspose(C, 64x64, stride 32) then the
 * softmax: output grid H' = H+32, W' = W+32.  Parameters: flm_fcn8_params with `up3` holding the
 * (64,64,C,C) kernel; score4, score3, up5, up4 are ignored.  Same contracts as the fcn8 calls. */
size_t flm_fcn32_packed_bytes(int n_classes, int dtype);
int flm_fcn32_pack(flm_stream_t stream, const flm_fcn8_params* params_dev_ptrs, int n_classes, int dtype,
                   void* packed_dev, size_t packed_bytes);
size_t flm_fcn32_workspace_bytes(int n, int h, int w, int n_classes, int dtype, int out_mode,
                                 int decode_mode, int n_points);
int flm_fcn32_forward(flm_stream_t stream, const void* packed_dev, const void* x_dev, int in_format,
                      int n, int h, int w, int n_classes, int dtype, int out_mode, int decode_mode,
                      int n_points, float thresh, void* out_dev, void* workspace_dev,
                      size_t workspace_bytes);

/* ---- architecture-generic entry points ------------------------------------------------------------
 * The registry of the reference (networks/basic_models.py:59-64, networks/fcn.py:153-192) builds the same
 * FCN head on several encoders.  `arch` selects the graph; the flm_fcn8_* / flm_fcn32_* calls above are
 * these with arch = FLM_ARCH_FCN8 / FLM_ARCH_FCN32.  Encoder convs arrive in network order in `enc`:
 * 5 layers for the vanilla encoder (BatchNorm tensors required), 13 for VGG16 (block1_conv1 ..
 * block5_conv3, networks/vgg16.py:27-72, no BatchNorm: gamma..var NULL), 27 for MobileNet-v1 (conv1, then
 * conv_dw_i / conv_pw_i for i = 1..13, networks/mobilenet.py:79-102; no biases: bias NULL; the depthwise
 * kernels are the Keras (3,3,C,1) tensors), 53 for ResNet50 (conv1, then per bottleneck block the shortcut
 * conv `res<stage><block>_branch1` when the block has one, then branch2a, 2b, 2c; networks/resnet50.py:
 * 145-170; every conv has bias and BatchNorm). */
enum flm_arch {
  FLM_ARCH_FCN8 = 0, FLM_ARCH_FCN32 = 1, FLM_ARCH_FCN8_VGG = 2, FLM_ARCH_FCN32_VGG = 3,
  FLM_ARCH_FCN8_MOBILENET = 4, FLM_ARCH_FCN32_MOBILENET = 5,
  FLM_ARCH_FCN8_RESNET50 = 6, FLM_ARCH_FCN32_RESNET50 = 7     /* every architecture builds in fp32 and bf16 */
};
typedef struct flm_fcn_params {
  const flm_conv_params* enc; /* host array of n_enc entries (the pointers inside are device pointers) */
  int n_enc;
  flm_conv_params fc6, fc7, score5, score4, score3;
  const float *up5, *up4, *up3; /* FCN-32 variants: only up3 = the (64,64,C,C) kernel */
} flm_fcn_params;
size_t flm_fcn_packed_bytes(int arch, int n_classes, int dtype);
int flm_fcn_pack(flm_stream_t stream, int arch, const flm_fcn_params* params, int n_classes, int dtype,
                 void* packed_dev, size_t packed_bytes);
size_t flm_fcn_workspace_bytes(int arch, int n, int h, int w, int n_classes, int dtype, int out_mode,
                               int decode_mode, int n_points);
int flm_fcn_forward(flm_stream_t stream, int arch, const void* packed_dev, const void* x_dev, int in_format,
                    int n, int h, int w, int n_classes, int dtype, int out_mode, int decode_mode,
                    int n_points, float thresh, void* out_dev, void* workspace_dev, size_t workspace_bytes);

/* Per-call options of the forward that change the WORKSPACE LAYOUT (never the results): pass the same struct to the
 * workspace query and to the forward.  NULL = defaults.  Initialise with flm_forward_opts_init (sets struct_size, which
 * lets the struct grow without breaking callers).
 *   landmark_candidates   1 (default): FLM_OUT_LANDMARKS with top-n, n <= 32, on the 68-class FCN-8 kernels selects
 *                         from candidate keys emitted by the last transposed conv instead of materialising the
 *                         [N,H'*W',C] probabilities (bit-identical landmarks, gated fallback); 0: always materialise
 *                         and decode (the decode of utils/metrics.py:102-109 on model.predict's output, literally)
 *   candidate_sub_phases  phases per tile in that path's sampling launch (1..16; 0 = by n_points: 4 up to n = 8 (fp32: 2),
 *                         6 up to 15, 8 beyond)
 *   candidate_cap_div     shrink the candidate lists by this factor (>= 1; tests of the overflow fallback) */
typedef struct flm_forward_opts {
  uint32_t struct_size;
  int32_t landmark_candidates;
  int32_t candidate_sub_phases;
  int32_t candidate_cap_div;
} flm_forward_opts;
void flm_forward_opts_init(flm_forward_opts* opts);
size_t flm_fcn_workspace_bytes_opts(int arch, int n, int h, int w, int n_classes, int dtype, int out_mode,
                                    int decode_mode, int n_points, const flm_forward_opts* opts);
int flm_fcn_forward_opts(flm_stream_t stream, int arch, const void* packed_dev, const void* x_dev, int in_format,
                         int n, int h, int w, int n_classes, int dtype, int out_mode, int decode_mode,
                         int n_points, float thresh, void* out_dev, void* workspace_dev, size_t workspace_bytes,
                         const flm_forward_opts* opts);
int64_t flm_fcn8_workspace_offset_opts(const char* name, int n, int h, int w, int n_classes, int dtype,
                                       int out_mode, int decode_mode, int n_points, const flm_forward_opts* opts);

/* One named Conv2D layer of the model in isolation ("enc2".."enc5" with BN+ReLU+pool fused,
 * "fc6", "fc7", "score5", "score4", "score3"): x_dev float32 [n,h,w,Cin] -> y_dev.  For tests and
 * developer tools that run or time one layer in isolation. */
int flm_fcn8_run_layer(flm_stream_t stream, const void* packed_dev, const char* layer, const void* x_dev,
                       void* y_dev, int n, int h, int w, int n_classes, int dtype);

/* ---- measurement hook (bench.py) --------------------------------------------------
 * When enabled, flm_fcn8_forward brackets each of its kernel launches with a hipEvent pair on
 * the caller's stream (no synchronisation).  flm_profile_read(i, ...) waits for record i and
 * returns its layer name and duration in ms; it returns 1 past the last record.
 * Process-global, not thread-safe: a measurement aid, off by default. */
int flm_profile_enable(int max_records);
/* Bracket only the launches of one layer ("fc6", ...; NULL or "" = every launch).  Every event pair costs a few
 * microseconds of stream time, so a timed region that only needs the dominant kernel's duration filters on it. */
int flm_profile_filter(const char* layer);
/* A/B performance knobs: they never change memory layouts, and -- with the one exception of "f32_two_level", which
 * selects between two fp32 summation orders -- never results; key "none" is always accepted, unknown keys
 * fail.  Atomic integers read at launch time; meant for A/B runs (tools/tune.py) and for tests that force a code path:
 *   "bf16_big_tiles"        0 off | 1 auto (default) | 2 whenever the shape allows | 3 auto + 256x128 tiles
 *                           256-row bf16 implicit-GEMM tiles (csrc/flm_igemm_bf16.hip)
 *   "bf16_lds_dma"          1 (default): the 256x256 tiles fetch their operands with buffer_load ... lds (no staging
 *                           registers, no LDS write pass); 0: global -> registers -> LDS
 *   "bf16_mfma16"           1 (default): the LDS-DMA form of those tiles computes with v_mfma_f32_16x16x32_bf16;
 *                           0: with 32x32x16.  Same bits
 *   "bf16_halo_mfma16"      1 (default): the halo-resident 3x3 kernel (enc2) computes with v_mfma_f32_16x16x32_bf16 too;
 *                           0: with 32x32x16.  Same bits
 *   "f32_two_level"         1 (default): the fp32 implicit GEMMs sum every 32-product k-step from zero and add the step
 *                           sums into a second accumulator set (chains of 32 + K/32 roundings, operands by LDS-DMA;
 *                           csrc/flm_igemm.hip); 0: one fmaf chain of K per output (round 2's kernel).  Both are valid
 *                           fp32 evaluations of the layer; the bits differ
 *   "bf16_group_n"          weight panels per tile group of that kernel (0 default, else a power of two <= 32)
 *   "bf16_conv3_halo"       0 off | 1 auto (default) | 2 always: halo-resident 3x3 kernel for 64-channel inputs
 *   "bf16_score1x1"         1 (default): 1x1 classifiers on 256-channel bf16 maps (score4, score3) run the kernel that
 *                           keeps the weights in registers (csrc/flm_score1x1.hip); 0: the implicit GEMM.  Same bits
 *   "bf16_fused_tail"       1 (default): seg_feats = crop(up4(fuse4)) + score3(f3) is ONE launch in the bf16 configuration
 *                           of the 68-class models (csrc/flm_tail_bf16.hip); 0: score3, then up4 with the skip add.
 *                           Same bits
 *   "decode_lds_dma"        1 (default): the standalone decode of 68-landmark maps brings its tiles into a three-slot LDS
 *                           ring by buffer_load ... lds (csrc/flm_decode.hip, decode_partial_dma_kernel); 0: one tile of
 *                           register prefetch.  Same results
 *   "posmajor_order"        1 (default): position-major layers (fc6) at batches smaller than a tile's rows take the map
 *                           positions that share a tile in an order chosen for their common filter taps
 *                           (csrc/flm_igemm_args.h, posmajor_fill_perm); 0: map order.  Same bits
 *   "warp_rows"             1 (default): uint8 warps whose destination width is a multiple of 64 run a wave per
 *                           64-pixel row segment, 2 rows per wave (csrc/flm_misc.hip, warp_u8_rows_kernel; 4: four rows
 *                           per wave); 0: the pixel-list kernel.  Same bits
 *   "up3_cand8"             bit 0: the bf16 candidate launch of the last transposed conv runs the 8-wave kernel
 *                           (csrc/flm_convt.hip, up3_cand8_kernel); bit 2: its 4-wave x 2-workgroup shape; default 1;
 *                           0: the generic kernel.  Same keys either way.  Bit 1 (an fp32 form of that kernel) is
 *                           accepted and ignored: the fp32 path always runs the generic kernel
 *   "up3_cand8_rows"        phase rows one workgroup of that kernel walks with the same input fragments: 0 (default)
 *                           chosen from the batch, else 1, 2, 4 or 8
 *   "up3_wreg"              1: that launch runs the weights-in-registers kernel instead (csrc/flm_up3_wreg.hip: a wave keeps
 *                           one phase's weights for its whole life, the input streams past through LDS) where its
 *                           conditions hold (stride 8, 68 classes; the probability region of the workspace is its
 *                           scratch); 0 (default).  Same keys
 * The options that change the workspace layout ("landmark_candidates", "candidate_*") are per-call arguments:
 * flm_forward_opts above. */
int flm_set_tuning(const char* key, int value);
/* Diagnostics for developers ("igemm_occupancy", arg = dynamic LDS bytes -> workgroups per CU). */
int flm_debug_query(const char* key, int arg);
int flm_profile_reset(void);
int flm_profile_read(int index, char* name_out, int name_cap, float* ms_out);
int flm_profile_disable(void);

/* ---- preprocess ----------------------------------------------------------------
 * get_image_array (data/generator.py:29-69) for crops already at model size:
 * uint8 BGR [N,H,W,3] -> float32 [N,H,W,3] (RGB for sub_mean). */
int flm_preprocess(flm_stream_t stream, const uint8_t* img_bgr_dev, int n, int h, int w, int norm,
                   float* out_dev);

/* ---- decode --------------------------------------------------------------------
 * transfer_target / transfer_xy_coord / get_average_xy (utils/metrics.py:46-109):
 * float32 heatmaps [N,H,W,L] -> float64 [N,L,2] (x,y) in heatmap pixel units,
 * (-1,-1) where mean(selected) <= thresh.  Ties at the n-th place: pixels are ordered
 * by (value, flat index), the n largest are kept. */
size_t flm_decode_workspace_bytes(int n, int h, int w, int l, int mode, int n_points);
int flm_decode(flm_stream_t stream, const float* hm_dev, int n, int h, int w, int l, int mode,
               int n_points, float thresh, double* out_dev, void* workspace_dev,
               size_t workspace_bytes);

/* ---- alignment (no reference implementation: README.md:1 states the intent only) --
 * Least-squares similarity (Umeyama, 4 dof) mapping each face's K landmarks onto a
 * template, returned as the 2x3 matrix M that maps SOURCE pixel coords to ALIGNED
 * coords; and the inverse-map bilinear warp with edge clamp (the skimage
 * `warp(..., mode="edge")` shape of data/generator.py:192-200). */
int flm_similarity_from_landmarks(flm_stream_t stream, const double* lm_dev /*[N,K,2]*/,
                                  const double* tmpl_dev /*[K,2]*/, int n, int k,
                                  float* m_dev /*[N,2,3]*/);
/* Same fit with the landmarks first taken from output-grid to crop pixel units: x * sx, y * sy in float64 (the
 * reference leaves decoded coordinates in grid units); points the decode rejected, (-1,-1), stay rejected. */
int flm_similarity_from_landmarks_scaled(flm_stream_t stream, const double* lm_dev /*[N,K,2]*/,
                                         const double* tmpl_dev /*[K,2]*/, int n, int k, double sx, double sy,
                                         float* m_dev /*[N,2,3]*/);
int flm_warp_affine(flm_stream_t stream, const void* src_dev /*[N,Hs,Ws,3]*/, int src_is_u8, int n,
                    int hs, int ws, const float* m_dev /*[N,2,3] src->dst*/,
                    float* dst_dev /*[N,Hd,Wd,3]*/, int hd, int wd);

/* ---- crop front-end (detect_marks pre-processing, prediction.py:76-83; get_image_array's resize,
 * data/generator.py:53) ---------------------------------------------------------------------------
 * `img[y0:y1, x0:x1]` + `cv2.resize(., (out_w, out_h))` (default INTER_LINEAR on uint8) for K boxes of one
 * uint8 BGR frame, into the model's input batch (uint8 BGR [K,out_h,out_w,3]); boxes are (x0,y0,x1,y1)
 * int32, already squared by the host-side box maths, clipped to the frame here (a box that misses the frame
 * gives zeros).  Integer fixed point: OpenCV's 11-bit-weight algorithm restated (csrc/flm_misc.hip states
 * every operation), bit-exact against oracle/warp_ref.py; parity with the cv2 binary itself is unpinned. */
int flm_crop_resize(flm_stream_t stream, const uint8_t* frame_dev, int fh, int fw,
                    const int32_t* boxes_dev /*[K,4]*/, int k, uint8_t* out_dev, int out_h, int out_w);
/* The same for the faces of SEVERAL frames in one launch (the multi-face stream of prediction.py:99-113, one launch
 * sequence per group of frames): `frames_dev` holds `nframes` uint8 BGR frames of fh x fw, `frame_stride` bytes apart
 * (a ring of stream frames in one allocation); box k is cut from frame frame_idx_dev[k] (an index outside [0, nframes)
 * gives zeros). */
int flm_crop_resize_frames(flm_stream_t stream, const uint8_t* frames_dev, size_t frame_stride, int nframes, int fh, int fw,
                           const int32_t* boxes_dev /*[K,4]*/, const int32_t* frame_idx_dev /*[K]*/, int k,
                           uint8_t* out_dev, int out_h, int out_w);

#ifdef __cplusplus
}
#endif
#endif /* FLM_H_ */
