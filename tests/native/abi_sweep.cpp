// Host-side sweep of the C ABI's query and argument-checking entry points (include/flm.h), built and run under
// AddressSanitizer + UndefinedBehaviorSanitizer by tests/test_abi_sanitized.py.  No GPU is touched: workspace / packed
// layouts, workspace offsets, option-struct validation and the null / bad-enum / bad-shape rejections are pure host
// arithmetic (size_t products, alignment, table lookups) -- the part of the library a caller can drive out of range.
#include <cstdio>
#include <cstring>

#include "flm.h"

static long g_checks = 0;
#define CHECK(cond)                                                         \
  do {                                                                      \
    ++g_checks;                                                             \
    if (!(cond)) {                                                          \
      std::fprintf(stderr, "abi_sweep: %s failed at line %d\n", #cond, __LINE__); \
      return 1;                                                             \
    }                                                                       \
  } while (0)

int main() {
  CHECK(flm_abi_version() == FLM_ABI_VERSION);
  const int ns[] = {1, 2, 3, 7, 16, 64, 130, 512};
  const int hw[][2] = {{32, 32}, {64, 96}, {96, 160}, {256, 256}, {416, 608}};
  const int cs[] = {1, 5, 21, 68, 96};
  const int npts[] = {0, 1, 4, 9, 25, 32, 33, 64};
  flm_forward_opts o_def, o_off, o_small, o_sub, o_tiny;
  flm_forward_opts_init(&o_def);
  flm_forward_opts_init(&o_off);
  o_off.landmark_candidates = 0;
  flm_forward_opts_init(&o_small);
  o_small.candidate_cap_div = 4096;
  flm_forward_opts_init(&o_sub);
  o_sub.candidate_sub_phases = 16;
  flm_forward_opts_init(&o_tiny);
  o_tiny.candidate_cap_div = 0x7fffffff;  // lists of one 64-key block: always the overflow fallback, never an empty list
  CHECK(o_def.struct_size == sizeof(flm_forward_opts) && o_def.landmark_candidates == 1 && o_def.candidate_cap_div == 1);
  const flm_forward_opts* opts[] = {nullptr, &o_def, &o_off, &o_small, &o_sub, &o_tiny};
  const char* names[] = {"f1", "f2", "f3", "f4", "f5", "fc6", "fc7", "score5", "fuse4", "seg_feats", "probs",
                         "cand_sub", "cand_tau", "cand_keys", "cand_cnt", "cand_cap", "nonsense", ""};
  for (int arch = 0; arch < 8; ++arch)
    for (int dt = 0; dt < 2; ++dt)
      for (int c : cs) {
        const size_t pb = flm_fcn_packed_bytes(arch, c, dt);
        CHECK(pb > 0 && pb < (size_t)1 << 32);
        for (int n : ns)
          for (auto& d : hw)
            for (int om = 0; om < 4; ++om)
              for (int dm = 0; dm < 2; ++dm)
                for (int np : npts) {
                  size_t last = 0;
                  for (const flm_forward_opts* o : opts) {
                    const size_t b = flm_fcn_workspace_bytes_opts(arch, n, d[0], d[1], c, dt, om, dm, np, o);
                    CHECK(b > 0 && b < (size_t)1 << 42);
                    if (o == nullptr) CHECK(b == flm_fcn_workspace_bytes(arch, n, d[0], d[1], c, dt, om, dm, np));
                    if (o == &o_def) CHECK(b == last);  // NULL means the defaults
                    if (o == &o_off) CHECK(b <= last);  // without the candidate lists the workspace never grows
                    last = b;
                    if (arch == 0 && n <= 7)
                      for (const char* nm : names) {
                        const int64_t off = flm_fcn8_workspace_offset_opts(nm, n, d[0], d[1], c, dt, om, dm, np, o);
                        CHECK(off >= -1 && (off < 0 || (size_t)off < b || !std::strcmp(nm, "cand_cap")));
                        if (!std::strcmp(nm, "cand_cap") && off >= 0) CHECK(off >= 64);
                      }
                  }
                }
      }
  for (int n : ns)
    for (auto& d : hw)
      for (int l : {1, 6, 68, 96})
        for (int dm = 0; dm < 2; ++dm)
          for (int np : npts) CHECK(flm_decode_workspace_bytes(n, d[0] + 8, d[1] + 8, l, dm, np) > 0);
  // out-of-range requests are refused with a message, never computed with wrapped sizes
  CHECK(flm_fcn_packed_bytes(8, 68, 0) == 0 && flm_fcn_packed_bytes(-1, 68, 0) == 0 && flm_fcn_packed_bytes(0, 0, 0) == 0);
  CHECK(flm_fcn_packed_bytes(0, 97, 0) == 0 && flm_fcn_packed_bytes(0, 68, 2) == 0 && flm_fcn8_packed_bytes(68, -3) == 0);
  CHECK(flm_fcn_workspace_bytes(0, 0, 256, 256, 68, 0, 0, 0, 0) == 0);
  CHECK(flm_fcn_workspace_bytes(0, -5, 256, 256, 68, 0, 0, 0, 0) == 0);
  CHECK(flm_fcn_workspace_bytes(0, 1, 250, 256, 68, 0, 0, 0, 0) == 0 && std::strstr(flm_last_error(), "multiples of 32"));
  CHECK(flm_fcn_workspace_bytes(0, 1, 256, 256, 0, 0, 0, 0, 0) == 0);
  CHECK(flm_fcn_workspace_bytes(0, 1 << 30, 1 << 14, 1 << 14, 68, 0, 0, 0, 0) == 0);  // would overflow 2^40 outputs
  CHECK(flm_fcn_workspace_bytes(99, 1, 256, 256, 68, 0, 0, 0, 0) == 0);
  CHECK(flm_fcn8_workspace_bytes(1, 256, 256, 68, 7, 0, 0, 0) == 0 && flm_fcn32_workspace_bytes(1, 256, 256, 68, 0, 0, 0, 0) > 0);
  CHECK(flm_decode_workspace_bytes(0, 8, 8, 1, 0, 0) == 0 && flm_decode_workspace_bytes(1, -8, 8, 1, 0, 0) == 0);
  flm_forward_opts bad;
  flm_forward_opts_init(&bad);
  bad.struct_size = 4;
  CHECK(flm_fcn_workspace_bytes_opts(0, 1, 256, 256, 68, 0, 2, 1, 4, &bad) == 0 && std::strstr(flm_last_error(), "struct_size"));
  flm_forward_opts_init(&bad);
  bad.candidate_sub_phases = 17;
  CHECK(flm_fcn_workspace_bytes_opts(0, 1, 256, 256, 68, 0, 2, 1, 4, &bad) == 0);
  bad.candidate_sub_phases = 0;
  bad.candidate_cap_div = 0;
  CHECK(flm_fcn_workspace_bytes_opts(0, 1, 256, 256, 68, 0, 2, 1, 4, &bad) == 0);
  flm_forward_opts_init(nullptr);  // tolerated
  CHECK(flm_fcn8_workspace_offset(nullptr, 1, 256, 256, 68, 0, 0, 0, 0) == -1);
  // null pointers and bad enums reach no kernel launch
  CHECK(flm_fcn_forward(nullptr, 0, nullptr, nullptr, 0, 1, 32, 32, 68, 0, 0, 0, 0, 0.f, nullptr, nullptr, 0) == FLM_ERR_ARG);
  CHECK(flm_fcn_forward_opts(nullptr, 0, nullptr, nullptr, 0, 1, 32, 32, 68, 0, 0, 0, 0, 0.f, nullptr, nullptr, 0, &bad) < 0);
  CHECK(flm_fcn_forward(nullptr, 42, &bad, &bad, 0, 1, 32, 32, 68, 0, 0, 0, 0, 0.f, &bad, &bad, 16) == FLM_ERR_ARG);
  CHECK(flm_fcn8_forward(nullptr, &bad, &bad, 0, 1, 32, 32, 68, 0, 9, 0, 0, 0.f, &bad, &bad, 16) == FLM_ERR_ARG);
  CHECK(flm_fcn8_forward(nullptr, &bad, &bad, 0, 1, 32, 32, 68, 0, 0, 0, 0, 0.f, &bad, &bad, 16) == FLM_ERR_WORKSPACE);
  CHECK(flm_fcn8_forward(nullptr, &bad, &bad, 0, 1, 33, 32, 68, 0, 0, 0, 0, 0.f, &bad, &bad, 16) == FLM_ERR_SHAPE);
  CHECK(flm_fcn8_pack(nullptr, nullptr, 68, 0, nullptr, 0) == FLM_ERR_ARG && flm_fcn32_pack(nullptr, nullptr, 68, 0, nullptr, 0) == FLM_ERR_ARG);
  flm_fcn_params fp;
  std::memset(&fp, 0, sizeof(fp));
  CHECK(flm_fcn_pack(nullptr, 0, &fp, 68, 0, &bad, 16) == FLM_ERR_WORKSPACE);
  CHECK(flm_fcn_pack(nullptr, 17, &fp, 68, 0, &bad, 16) == FLM_ERR_ARG && flm_fcn_pack(nullptr, 0, &fp, 68, 5, &bad, 16) == FLM_ERR_UNSUPPORTED);
  CHECK(flm_fcn_pack(nullptr, 0, &fp, 0, 0, &bad, 16) == FLM_ERR_SHAPE);
  CHECK(flm_decode(nullptr, nullptr, 1, 8, 8, 1, 0, 0, 0.f, nullptr, nullptr, 0) == FLM_ERR_ARG);
  CHECK(flm_preprocess(nullptr, nullptr, 1, 8, 8, 0, nullptr) == FLM_ERR_ARG);
  CHECK(flm_similarity_from_landmarks(nullptr, nullptr, nullptr, 1, 68, nullptr) == FLM_ERR_ARG);
  CHECK(flm_similarity_from_landmarks_scaled(nullptr, nullptr, nullptr, 1, 68, 1.0, 1.0, nullptr) == FLM_ERR_ARG);
  CHECK(flm_warp_affine(nullptr, nullptr, 1, 1, 8, 8, nullptr, nullptr, 8, 8) == FLM_ERR_ARG);
  CHECK(flm_crop_resize(nullptr, nullptr, 8, 8, nullptr, 1, nullptr, 8, 8) == FLM_ERR_ARG);
  CHECK(flm_crop_resize_frames(nullptr, nullptr, 192, 1, 8, 8, nullptr, nullptr, 1, nullptr, 8, 8) == FLM_ERR_ARG);
  CHECK(flm_fcn8_run_layer(nullptr, nullptr, "fc6", nullptr, nullptr, 1, 8, 8, 68, 0) == FLM_ERR_ARG);
  // tuning keys: known accepted, unknown / out of range refused; the layout options are no longer process state
  CHECK(flm_set_tuning("none", 0) == 0 && flm_set_tuning(nullptr, 0) == FLM_ERR_ARG && flm_set_tuning("nope", 1) == FLM_ERR_ARG);
  CHECK(flm_set_tuning("bf16_group_n", 3) == FLM_ERR_ARG && flm_set_tuning("bf16_group_n", 0) == 0);
  CHECK(flm_set_tuning("up3_cand8_rows", 3) == FLM_ERR_ARG && flm_set_tuning("up3_cand8_rows", 0) == 0);
  CHECK(flm_set_tuning("up3_wreg", 2) == FLM_ERR_ARG && flm_set_tuning("up3_wreg", 1) == 0 && flm_set_tuning("up3_wreg", 0) == 0);
  CHECK(flm_set_tuning("landmark_candidates", 0) == FLM_ERR_ARG && std::strstr(flm_last_error(), "flm_forward_opts"));
  CHECK(flm_profile_read(0, nullptr, 0, nullptr) == 1 && flm_profile_filter("a-layer-name-that-is-much-too-long-for-the-filter") == FLM_ERR_ARG);
  CHECK(flm_debug_query(nullptr, 0) == -1);
  std::printf("abi_sweep ok: %ld checks\n", g_checks);
  return 0;
}
