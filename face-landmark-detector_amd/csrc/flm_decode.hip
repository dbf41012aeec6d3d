// Heatmap -> landmark coordinates ("soft-argmax"), HBM-bound single pass over the heatmaps.
//
// Restates get_average_xy / transfer_xy_coord / transfer_target (reference utils/metrics.py:46-109):
//   n_points < 1 : full-map weighted centroid                                  (:58-64)
//   n_points >= 1: weighted centroid of the n largest pixels                   (:66-77)
//   reject -> (-1,-1) when  hsum / n_points <= thresh                          (:78-79)
// Numeric types follow the reference's numpy behaviour: in the top-n branch `hsum` is a sequential
// float32 sum in ascending value order and the index-weighted sums are float64; the coordinates are
// float64.  Ties at the n-th place: pixels are ordered by (value, flat index), the n largest kept.
//
// Pass 1 (decode_partial): grid = (chunks, faces).  A workgroup streams its pixel range in tiles of
// 64 pixels x L channels: coalesced 16-byte loads -> LDS (row stride odd, so a wave reading one
// channel of 64 pixels is bank-conflict-free) -> each wave owns L/4 channels with lane = pixel.
//   ALL : per-lane float64 partial sums in registers, one wave reduction at the end.
//   TOPN: per channel a descending list of the n best (value,index) keys spread over the wave's
//         lanes (lane i = i-th best); a 64-pixel batch is tested against the list's n-th key with one
//         compare + ballot, insertions (rare after warm-up) are a ballot/popcount + one lane shift.
// Pass 2 (decode_merge): one wave per (face, landmark) merges the chunk partials and finishes the
// arithmetic in the reference's order.
#include "flm_common.h"

namespace flm {

constexpr int PT = 64;  // pixels per tile

struct DecodeArgs {
  const float* hm;
  int n, h, w, l;
  int chunks, chunk_px;  // chunk_px multiple of 64
  int mode, n_points;
  int vec;  // face stride is a multiple of 16 bytes: 16-byte loads allowed
  float thresh;
  void* part;   // ALL: double [n][chunks][l][3]; TOPN: u64 [n][chunks][l][n_points]
  double* out;  // [n][l][2]
  float* tau_out;        // TOPN only, non-null: write the n-th largest VALUE per (face, landmark) instead of coordinates
  const unsigned* gate;  // non-null: the launch does nothing unless *gate != 0
};

// 16 bytes of a map that is read exactly once: the non-temporal hint keeps the stream from displacing everything else in
// the L2 (standalone top-4 decode of 68-landmark maps, LDS-DMA form: 0.253 -> 0.234 ms at batch 64, 1.63 -> 1.49 ms at 512)
__device__ __forceinline__ float4 load_stream16(const float* p) {
  typedef float f4v __attribute__((ext_vector_type(4)));
  const f4v v = __builtin_nontemporal_load(reinterpret_cast<const f4v*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}

__device__ __forceinline__ unsigned order_bits(float v) {
  const unsigned u = __float_as_uint(v);
  return u ^ ((u >> 31) ? 0xffffffffu : 0x80000000u);
}
__device__ __forceinline__ float from_order_bits(unsigned o) {
  const unsigned u = (o & 0x80000000u) ? (o ^ 0x80000000u) : ~o;
  return __uint_as_float(u);
}
__device__ __forceinline__ unsigned long long readlane64(unsigned long long v, int srclane) {
  const unsigned lo = __builtin_amdgcn_readlane((unsigned)v, srclane);
  const unsigned hi = __builtin_amdgcn_readlane((unsigned)(v >> 32), srclane);
  return ((unsigned long long)hi << 32) | lo;
}
// lane i <- lane i-1 across the whole wave, lane 0 <- 0: the gfx9 DPP wave shift (wave_shr:1, one VALU move per half)
// instead of __shfl_up's ds_bpermute round trip -- this sits on the serial chain of every list insertion.
__device__ __forceinline__ unsigned long long shfl_up64(unsigned long long v, int lane) {
  const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)v, 0x138, 0xf, 0xf, false);
  const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(v >> 32), 0x138, 0xf, 0xf, false);
  (void)lane;
  return ((unsigned long long)hi << 32) | lo;
}

// Insert every key of `cand` (one per lane, 0 = none) that beats the list's n-th entry.
// list: descending across lanes 0..n-1 (0 = empty slot); tau = list[n-1].
__device__ __forceinline__ void insert_candidates(unsigned long long& list, unsigned long long& tau,
                                                  unsigned long long cand, int n, int lane) {
  unsigned long long mask = __ballot(cand > tau);
  while (mask) {
    const int src = __builtin_ctzll(mask);
    const unsigned long long k = readlane64(cand, src);
    if (lane == src) cand = 0;
    const int pos = __builtin_popcountll(__ballot(list > k));  // entries that stay ahead of k
    const unsigned long long up = shfl_up64(list, lane);
    list = (lane < pos) ? list : (lane == pos ? k : up);
    if (lane >= n) list = 0;
    tau = readlane64(list, n - 1);
    mask = __ballot(cand > tau);
  }
}

// The same for 64 < n <= 128 (the reference's own sweep decodes n = k*k up to 81, utils/metrics.py:130-133): the list
// takes two registers per lane, ranks 0..63 in `l0` and 64..127 in `l1`; an insertion shifts both, the last entry of
// `l0` carrying into lane 0 of `l1`.  tau = entry n-1 (in `l1`).
__device__ __forceinline__ void insert_candidates_wide(unsigned long long& l0, unsigned long long& l1,
                                                       unsigned long long& tau, unsigned long long cand, int n, int lane) {
  unsigned long long mask = __ballot(cand > tau);
  while (mask) {
    const int src = __builtin_ctzll(mask);
    const unsigned long long k = readlane64(cand, src);
    if (lane == src) cand = 0;
    const int pos = __builtin_popcountll(__ballot(l0 > k)) + __builtin_popcountll(__ballot(l1 > k));
    const unsigned long long carry = readlane64(l0, 63);
    const unsigned long long up0 = shfl_up64(l0, lane), up1 = shfl_up64(l1, lane);
    if (pos < 64) {  // wave-uniform
      l0 = (lane < pos) ? l0 : (lane == pos ? k : up0);
      l1 = lane == 0 ? carry : up1;
    } else {
      const int q = pos - 64;
      l1 = (lane < q) ? l1 : (lane == q ? k : up1);
    }
    if (lane + 64 >= n) l1 = 0;
    tau = readlane64(l1, n - 65);
    mask = __ballot(cand > tau);
  }
}

// utils/metrics.py:69-77 on a finished list: float32 sum in ascending value order (= list lanes n-1 .. 0),
// float64 index-weighted sums, reject when hsum / n_points <= thresh.
__device__ __forceinline__ void finish_topn(unsigned long long list, int n_points, int w, float thresh, int lane,
                                            double* out, unsigned long long list_hi = 0ull) {
  float hsum = 0.f;
  double i0 = 0.0, i1 = 0.0;
  for (int i = n_points - 1; i >= 0; --i) {
    const unsigned long long k = i >= 64 ? readlane64(list_hi, i - 64) : readlane64(list, i);
    if (k == 0ull) continue;
    const float hv = from_order_bits((unsigned)(k >> 32));
    const unsigned idx = (unsigned)k;
    hsum += hv;
    i0 += (double)(idx / (unsigned)w) * (double)hv;
    i1 += (double)(idx % (unsigned)w) * (double)hv;
  }
  double x = i1 / (double)hsum, y = i0 / (double)hsum;
  if (hsum / (float)n_points <= thresh) { x = -1.0; y = -1.0; }
  if (lane == 0) {
    out[0] = x;
    out[1] = y;
  }
}

template <int MODE, int CPW, bool WIDE = false>
__global__ __launch_bounds__(256) void decode_partial_kernel(DecodeArgs a) {
  if (a.gate && *a.gate == 0) return;
  extern __shared__ __attribute__((aligned(16))) float tile[];  // [PT][LS]
  const int L = a.l, LS = L | 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int face = blockIdx.y, chunk = blockIdx.x;
  const int HW = a.h * a.w;
  const int p_begin = chunk * a.chunk_px;
  const int p_end = min(p_begin + a.chunk_px, HW);
  const float* src = a.hm + (size_t)face * HW * L;
  const int c_first = wave * CPW;

  double s0[CPW], sx[CPW], sy[CPW];                 // ALL
  unsigned long long list[CPW], tau[CPW];           // TOPN
  unsigned long long list_hi[WIDE ? CPW : 1];       // TOPN, 64 < n <= 128: ranks 64..127
#pragma unroll
  for (int i = 0; i < CPW; ++i) {
    if (MODE == FLM_DECODE_ALL) {
      s0[i] = 0.0; sx[i] = 0.0; sy[i] = 0.0;
    } else {
      list[i] = 0ull; tau[i] = 0ull;
      if (WIDE) list_hi[i] = 0ull;
    }
  }

  const int tile_f = PT * L;  // floats per full tile (multiple of 4 because PT is)
  // (the map is read once: its loads carry the non-temporal hint, here and in the LDS-DMA form)
  // Register prefetch of the NEXT tile (named registers: up to 6 x 16 bytes per thread cover L <= 96),
  // issued before the current tile is processed so the HBM latency hides behind the selection work.
  float4 pf0, pf1, pf2, pf3, pf4, pf5;
  pf0 = pf1 = pf2 = pf3 = pf4 = pf5 = make_float4(0.f, 0.f, 0.f, 0.f);
#define FLM_PF_LOAD(I, R)                                                         \
  {                                                                               \
    const int e4 = tid * 4 + 1024 * I;                                            \
    if (e4 < tile_f) R = load_stream16(nsrc + e4);                                \
  }
#define FLM_PF_STORE(I, R)                                                        \
  {                                                                               \
    const int e4 = tid * 4 + 1024 * I;                                            \
    if (e4 < tile_f) {                                                            \
      int p = e4 / L, c = e4 - p * L;                                             \
      tile[p * LS + c] = R.x; if (++c == L) { c = 0; ++p; }                       \
      tile[p * LS + c] = R.y; if (++c == L) { c = 0; ++p; }                       \
      tile[p * LS + c] = R.z; if (++c == L) { c = 0; ++p; }                       \
      tile[p * LS + c] = R.w;                                                     \
    }                                                                             \
  }
  bool pf_valid = false;
  if (a.vec && p_begin + PT <= p_end) {
    const float* nsrc = src + (size_t)p_begin * L;
    FLM_PF_LOAD(0, pf0) FLM_PF_LOAD(1, pf1) FLM_PF_LOAD(2, pf2) FLM_PF_LOAD(3, pf3) FLM_PF_LOAD(4, pf4)
    FLM_PF_LOAD(5, pf5)
    pf_valid = true;
  }
  for (int p0 = p_begin; p0 < p_end; p0 += PT) {
    const int npx = min(PT, p_end - p0);
    const int nf = npx * L;
    __syncthreads();
    if (pf_valid) {
      FLM_PF_STORE(0, pf0) FLM_PF_STORE(1, pf1) FLM_PF_STORE(2, pf2) FLM_PF_STORE(3, pf3) FLM_PF_STORE(4, pf4)
      FLM_PF_STORE(5, pf5)
    } else {
      // partial or unaligned tile: plain loads, zero fill
      const float* tsrc = src + (size_t)p0 * L;
      for (int e = tid; e < tile_f; e += 256) {
        const int p = e / L, c = e - p * L;
        tile[p * LS + c] = (e < nf) ? tsrc[e] : 0.f;
      }
    }
    pf_valid = a.vec && p0 + 2 * PT <= p_end;
    if (pf_valid) {
      const float* nsrc = src + (size_t)(p0 + PT) * L;
      FLM_PF_LOAD(0, pf0) FLM_PF_LOAD(1, pf1) FLM_PF_LOAD(2, pf2) FLM_PF_LOAD(3, pf3) FLM_PF_LOAD(4, pf4)
      FLM_PF_LOAD(5, pf5)
    }
    __syncthreads();

    const int pix = p0 + lane;
    const bool pvalid = lane < npx;
    if (MODE == FLM_DECODE_ALL) {
      const double dx = (double)(pix % a.w), dy = (double)(pix / a.w);
#pragma unroll
      for (int i = 0; i < CPW; ++i) {
        const int c = c_first + i;
        if (c < L) {
          const double hv = pvalid ? (double)tile[lane * LS + c] : 0.0;
          s0[i] += hv;
          sx[i] = fma(hv, dx, sx[i]);
          sy[i] = fma(hv, dy, sy[i]);
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < CPW; ++i) {
        const int c = c_first + i;
        if (c < L) {  // wave-uniform
          const float hv = tile[lane * LS + c];
          const unsigned long long key =
              pvalid ? (((unsigned long long)order_bits(hv) << 32) | (unsigned)pix) : 0ull;
          if (__any(key > tau[i])) {
            if constexpr (WIDE) insert_candidates_wide(list[i], list_hi[i], tau[i], key, a.n_points, lane);
            else insert_candidates(list[i], tau[i], key, a.n_points, lane);
          }
        }
      }
    }
  }

#undef FLM_PF_LOAD
#undef FLM_PF_STORE

  // ---- write partials ---------------------------------------------------------------------------
  if (MODE == FLM_DECODE_ALL) {
    double* part = reinterpret_cast<double*>(a.part) + ((size_t)face * a.chunks + chunk) * L * 3;
#pragma unroll
    for (int i = 0; i < CPW; ++i) {
      const int c = c_first + i;
      if (c < L) {
        double v0 = s0[i], v1 = sx[i], v2 = sy[i];
#pragma unroll
        for (int sh = 32; sh >= 1; sh >>= 1) {  // fixed-order butterfly: deterministic
          v0 += __shfl_xor(v0, sh);
          v1 += __shfl_xor(v1, sh);
          v2 += __shfl_xor(v2, sh);
        }
        if (lane == 0) {
          part[c * 3 + 0] = v0;
          part[c * 3 + 1] = v1;
          part[c * 3 + 2] = v2;
        }
      }
    }
  } else {
    unsigned long long* part =
        reinterpret_cast<unsigned long long*>(a.part) + ((size_t)face * a.chunks + chunk) * L * a.n_points;
#pragma unroll
    for (int i = 0; i < CPW; ++i) {
      const int c = c_first + i;
      if (c < L && lane < a.n_points) part[(size_t)c * a.n_points + lane] = list[i];
      if (WIDE && c < L && lane + 64 < a.n_points) part[(size_t)c * a.n_points + 64 + lane] = list_hi[i];
    }
  }
}

// ---- the same pass for 68-landmark maps with the tiles brought in by LDS-DMA (round 3) --------------------------------
// What bounded the kernel above at batch 64 (0.31 ms, 0.49 of 8 TB/s) is bytes in flight: one 17 KiB tile of register
// prefetch per workgroup, three workgroups per CU, 51 KiB per CU against ~3 us of loaded HBM latency.  Here a tile goes
// from HBM to LDS by buffer_load_dwordx4 ... lds (17 requests of 1 KiB; inline assembly as in flm_igemm_args.h, so that
// hipcc does not order the tile's ds_reads behind every pending request) into a ring of three slots: while tile t is
// processed, tiles t+1 and t+2 are in flight -- twice the bytes, no prefetch registers.  The image of a tile is then the
// plain [pixel][68] array (rows of 272 bytes: an odd row stride is not available to a DMA): a wave owns channels
// 16w .. 16w+15 and 64+w and reads its pixel's values as four ds_read_b128 + one b32 -- lanes 272 bytes apart cover all
// 32 banks once per 8 lanes, conflict-free.  The tail of a chunk is zero-filled by the buffer bounds check
// (num_records = the chunk's bytes; the tile offset rides in the VECTOR offset, the one the check looks at).
// The requests carry `nt`: the map is read once (0.253 -> 0.234 ms at batch 64, 1.63 -> 1.49 ms at 512).
// One barrier per tile: a wave waits for its own requests of tile t (vmcnt), the barrier makes every wave's pieces
// visible and proves that tile t-1 has been read by all, then tile t+2 is requested into t-1's slot.
constexpr int DL = 68, D_TILE_B = PT * DL * 4, D_PIECES = D_TILE_B / 1024, D_RING = 3, D_PPW = (D_PIECES + 3) / 4;
static_assert(D_TILE_B % 1024 == 0 && D_PPW == 5, "17 pieces of 1 KiB: five per wave, the missing ones repeat piece w");

template <int MODE>
__global__ __launch_bounds__(256) void decode_partial_dma_kernel(DecodeArgs a) {
  if (a.gate && *a.gate == 0) return;
  extern __shared__ __attribute__((aligned(16))) char ring[];  // [D_RING][PT][DL] floats
  constexpr int CPW = 17;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int face = blockIdx.y, chunk = blockIdx.x;
  const int HW = a.h * a.w;
  const int p_begin = chunk * a.chunk_px;
  const int p_end = min(p_begin + a.chunk_px, HW);
  const int ntiles = (p_end - p_begin + PT - 1) / PT;
  if (ntiles <= 0) return;  // (uniform)

  double s0[CPW], sx[CPW], sy[CPW];                 // ALL
  unsigned long long list[CPW], tau[CPW];           // TOPN
  float tauf[CPW];                                  // TOPN: the value of the list's n-th key (NaN while the list is not full)
#pragma unroll
  for (int i = 0; i < CPW; ++i) {
    if (MODE == FLM_DECODE_ALL) {
      s0[i] = 0.0; sx[i] = 0.0; sy[i] = 0.0;
    } else {
      list[i] = 0ull; tau[i] = 0ull;
      tauf[i] = from_order_bits(0u);
    }
  }

  typedef int dsrd_t __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) char lds_char;
  const unsigned ring_lds = (unsigned)(size_t)((lds_char*)ring);
  const unsigned long long cb = reinterpret_cast<unsigned long long>(a.hm + ((size_t)face * HW + p_begin) * DL);
  const dsrd_t srd = (dsrd_t){(int)(unsigned)cb, (int)(unsigned)((cb >> 32) & 0xffffu), (p_end - p_begin) * DL * 4, 0x00020000};
  // piece k of a tile: bytes [1024 k, 1024 k + 1024); this wave's pieces wave, wave + 4, ... (five requests per tile and
  // wave so that the vmcnt arithmetic is the same in every wave: a piece past the 17th repeats piece `wave`)
  auto issue = [&](int t) __attribute__((always_inline)) {
    const unsigned slot = ring_lds + (unsigned)(t % D_RING) * D_TILE_B;
#pragma unroll
    for (int j = 0; j < D_PPW; ++j) {
      const int k = wave + 4 * j < D_PIECES ? wave + 4 * j : wave;
      asm volatile("s_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen nt lds"
                   :
                   : "v"((unsigned)t * D_TILE_B + (unsigned)k * 1024u + (unsigned)lane * 16u), "s"(srd), "{m0}"(slot + k * 1024)
                   : "memory");
    }
  };
  issue(0);
  if (ntiles > 1) issue(1);
  const int c16 = 16 * wave;  // this wave's channels: c16 .. c16 + 15 and 64 + wave
  for (int t = 0; t < ntiles; ++t) {
    if (t + 1 < ntiles) __builtin_amdgcn_s_waitcnt(0x0f75);  // vmcnt(5): tile t's five requests done, tile t+1's may fly
    else __builtin_amdgcn_s_waitcnt(0x0f70);
    __syncthreads();
    if (t + 2 < ntiles) issue(t + 2);
    const char* tile = ring + (size_t)(t % D_RING) * D_TILE_B + lane * (DL * 4);
    float v[CPW];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 q = *reinterpret_cast<const float4*>(tile + (c16 + 4 * j) * 4);
      v[4 * j] = q.x; v[4 * j + 1] = q.y; v[4 * j + 2] = q.z; v[4 * j + 3] = q.w;
    }
    v[16] = *reinterpret_cast<const float*>(tile + (64 + wave) * 4);
    const int p0 = p_begin + t * PT;
    const int pix = p0 + lane;
    const bool pvalid = pix < p_end;
    if (MODE == FLM_DECODE_ALL) {
      const double dx = (double)(pix % a.w), dy = (double)(pix / a.w);
#pragma unroll
      for (int i = 0; i < CPW; ++i) {
        const double hv = pvalid ? (double)v[i] : 0.0;
        s0[i] += hv;
        sx[i] = fma(hv, dx, sx[i]);
        sy[i] = fma(hv, dy, sy[i]);
      }
    } else {
      // The per-value test is ONE float compare against the list's n-th VALUE (a scalar): "not less than" lets every true
      // candidate through -- an equal value may still win on the pixel index, a NaN orders above everything as its order
      // bits do, and while the list is not full its n-th value reads as NaN, which nothing is less than -- and the 64-bit
      // key is only built for a class that has a candidate.  The sweep over n said that this test, not the insertions,
      // is what separates top-4 from the all-pixel mode's streaming rate (0.283 -> 0.26 ms at batch 64).
#pragma unroll
      for (int i = 0; i < CPW; ++i) {
        if (__any(!(v[i] < tauf[i]))) {
          const unsigned long long key = pvalid ? (((unsigned long long)order_bits(v[i]) << 32) | (unsigned)pix) : 0ull;
          if (__any(key > tau[i])) {
            insert_candidates(list[i], tau[i], key, a.n_points, lane);
            tauf[i] = from_order_bits((unsigned)(tau[i] >> 32));
          }
        }
      }
    }
  }

  // ---- partials, in the layout of decode_partial_kernel (channel index c) --------------------------------------------
  if (MODE == FLM_DECODE_ALL) {
    double* part = reinterpret_cast<double*>(a.part) + ((size_t)face * a.chunks + chunk) * DL * 3;
#pragma unroll
    for (int i = 0; i < CPW; ++i) {
      const int c = i < 16 ? c16 + i : 64 + wave;
      double v0 = s0[i], v1 = sx[i], v2 = sy[i];
#pragma unroll
      for (int sh = 32; sh >= 1; sh >>= 1) {  // fixed-order butterfly: deterministic
        v0 += __shfl_xor(v0, sh);
        v1 += __shfl_xor(v1, sh);
        v2 += __shfl_xor(v2, sh);
      }
      if (lane == 0) {
        part[c * 3 + 0] = v0;
        part[c * 3 + 1] = v1;
        part[c * 3 + 2] = v2;
      }
    }
  } else {
    unsigned long long* part =
        reinterpret_cast<unsigned long long*>(a.part) + ((size_t)face * a.chunks + chunk) * DL * a.n_points;
#pragma unroll
    for (int i = 0; i < CPW; ++i) {
      const int c = i < 16 ? c16 + i : 64 + wave;
      if (lane < a.n_points) part[(size_t)c * a.n_points + lane] = list[i];
    }
  }
}

static std::atomic<int> g_decode_dma{1};  // A/B knob "decode_lds_dma": same results either way
void decode_dma_enable(int on) { g_decode_dma.store(on, std::memory_order_relaxed); }

// one wave per (face, landmark)
template <int MODE, bool WIDE = false>
__global__ __launch_bounds__(64) void decode_merge_kernel(DecodeArgs a) {
  if (a.gate && *a.gate == 0) return;
  const int lane = threadIdx.x;
  const int c = blockIdx.x, face = blockIdx.y;
  const int L = a.l;
  double* out = a.out + ((size_t)face * L + c) * 2;
  if (MODE == FLM_DECODE_ALL) {
    if (lane != 0) return;
    const double* part = reinterpret_cast<const double*>(a.part) + (size_t)face * a.chunks * L * 3;
    double v0 = 0.0, v1 = 0.0, v2 = 0.0;
    for (int s = 0; s < a.chunks; ++s) {
      v0 += part[((size_t)s * L + c) * 3 + 0];
      v1 += part[((size_t)s * L + c) * 3 + 1];
      v2 += part[((size_t)s * L + c) * 3 + 2];
    }
    // utils/metrics.py:60: hsum is float32 (np.sum of a float32 map), n_points = H*W
    const float hsum = (float)v0;
    double x = v1 / (double)hsum, y = v2 / (double)hsum;
    if (hsum / (float)(a.h * a.w) <= a.thresh) { x = -1.0; y = -1.0; }
    out[0] = x;
    out[1] = y;
  } else {
    const unsigned long long* part =
        reinterpret_cast<const unsigned long long*>(a.part) + (size_t)face * a.chunks * L * a.n_points;
    unsigned long long list = 0ull, tau = 0ull;
    if constexpr (WIDE) {  // 64 < n <= 128: a chunk's list arrives in two batches of up to 64 keys
      unsigned long long list_hi = 0ull;
      for (int s = 0; s < a.chunks; ++s)
        for (int r0 = 0; r0 < a.n_points; r0 += 64) {
          const unsigned long long cand = r0 + lane < a.n_points ? part[((size_t)s * L + c) * a.n_points + r0 + lane] : 0ull;
          if (__any(cand > tau)) insert_candidates_wide(list, list_hi, tau, cand, a.n_points, lane);
        }
      if (a.tau_out) {
        const unsigned long long k = readlane64(list_hi, a.n_points - 65);
        if (lane == 0) a.tau_out[(size_t)face * L + c] = k ? from_order_bits((unsigned)(k >> 32)) : -3.402823466e38f;
        return;
      }
      finish_topn(list, a.n_points, a.w, a.thresh, lane, out, list_hi);
      return;
    }
    // 64 / n_points chunk lists are merged per pass (lane -> (chunk offset, rank))
    const int per = 64 / a.n_points;
    for (int s0 = 0; s0 < a.chunks; s0 += per) {
      const int s = s0 + lane / a.n_points, rk = lane % a.n_points;
      const unsigned long long cand =
          (lane < per * a.n_points && s < a.chunks) ? part[((size_t)s * L + c) * a.n_points + rk] : 0ull;
      if (__any(cand > tau)) insert_candidates(list, tau, cand, a.n_points, lane);
    }
    if (a.tau_out) {  // threshold pass of the candidate path (flm_convt.hip): the n-th largest value, or -max
      const unsigned long long k = readlane64(list, a.n_points - 1);
      if (lane == 0) a.tau_out[(size_t)face * L + c] = k ? from_order_bits((unsigned)(k >> 32)) : -3.402823466e38f;
      return;
    }
    finish_topn(list, a.n_points, a.w, a.thresh, lane, out);
  }
}

// Exact top n of a face's candidate keys (flm_convt.hip, epilogue 3): key = order_bits(p) << 32 | class << 17 |
// pixel, in the order the workgroups of the candidate launch flushed them.  grid = (G, faces): workgroup g owns the
// classes g*cpg .. g*cpg + cpg - 1, wave w of its NW the classes g*cpg + w + NW*k.  The keys are first BUCKETED by class in
// LDS (counting sort: histogram, prefix, scatter; kMergeKeys per pass, a longer list takes several passes with the
// lists kept in registers), then every wave feeds only the ~cnt/68 keys of each of its classes to the same descending
// (value, pixel) lists as the decode of a materialised map, so ties resolve identically -- the keys are distinct and
// the lists order-independent, so the bucket order does not matter.  (Round 1 had every wave scan ALL keys of the face
// once per class it owned: 0.16 ms per 512 faces, a serial chain of cnt/64 steps x 5 classes per wave.)
struct CandMergeArgs {
  const unsigned long long* cand;
  unsigned* cand_cnt;  // [n] fill counts, [n] = fallback flag
  int n, w, l, n_points, cap;
  float thresh;
  double* out;
  int cpg;  // classes per workgroup
};

constexpr int kMergeKeys = 6144;     // 48 KiB of keys per pass: three workgroups per CU
constexpr int kCandFineBatch = 128;  // below: four workgroups per face (the chip would sit empty with one)

template <int CPW, int NW>  // NW waves, CPW = ceil(cpg / NW) classes per wave
__global__ __launch_bounds__(NW * 64) void cand_merge_kernel(CandMergeArgs a) {
  __shared__ unsigned long long keys[kMergeKeys];
  __shared__ int hist[NW * CPW + 1], off[NW * CPW + 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int face = blockIdx.y;
  const int cfirst = a.cpg * blockIdx.x, cend = min(cfirst + a.cpg, a.l);
  const int nc = cend - cfirst;
  const unsigned cnt = min(a.cand_cnt[face], (unsigned)a.cap);
  const unsigned long long* src = a.cand + (size_t)face * a.cap;
  unsigned long long list[CPW], tau[CPW];
#pragma unroll
  for (int k = 0; k < CPW; ++k) { list[k] = 0ull; tau[k] = 0ull; }
  constexpr int KPT = kMergeKeys / (NW * 64);  // keys per thread and pass, all loads in flight at once
  for (unsigned base = 0; base < cnt; base += kMergeKeys) {
    unsigned long long kreg[KPT];
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
      const unsigned i = base + tid + NW * 64 * j;
      kreg[j] = i < cnt ? src[i] : 0ull;
    }
    if (tid <= NW * CPW) hist[tid] = 0;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
      const int rel = (int)((kreg[j] >> 17) & 127u) - cfirst;
      if (kreg[j] != 0ull && (unsigned)rel < (unsigned)nc) atomicAdd(&hist[rel], 1);
      else kreg[j] = 0ull;
    }
    __syncthreads();
    if (wave == 0) {  // exclusive prefix over the classes; hist becomes the write cursor
      const int v = lane < nc ? hist[lane] : 0;
      int incl = v;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int t = __shfl_up(incl, d);
        if (lane >= d) incl += t;
      }
      if (lane < nc) {
        off[lane] = incl - v;
        hist[lane] = incl - v;
      }
      if (lane == 63 && nc >= 64) {
        int run = incl;
        for (int c = 64; c < nc; ++c) {
          off[c] = run;
          const int h = hist[c];
          hist[c] = run;
          run += h;
        }
        off[nc] = run;
      }
      if (nc < 64 && lane == nc) off[nc] = incl;  // (incl of lane nc = the total: its own v is 0)
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < KPT; ++j) {
      if (kreg[j] != 0ull) {
        const int rel = (int)((kreg[j] >> 17) & 127u) - cfirst;
        keys[atomicAdd(&hist[rel], 1)] = (kreg[j] & 0xffffffff00000000ull) | (kreg[j] & 0x1ffffull);
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < CPW; ++k) {
      const int rel = wave + NW * k;
      if (rel < nc) {  // wave-uniform
        const int lo = off[rel], hi = off[rel + 1];
        for (int i0 = lo; i0 < hi; i0 += 64) {
          const unsigned long long cand = (i0 + lane < hi) ? keys[i0 + lane] : 0ull;
          if (__any(cand > tau[k])) insert_candidates(list[k], tau[k], cand, a.n_points, lane);
        }
      }
    }
    __syncthreads();  // the next pass overwrites the buckets
  }
#pragma unroll
  for (int k = 0; k < CPW; ++k) {
    const int c = cfirst + wave + NW * k;
    if (c < cend) {
      // fewer than n keys: the threshold did not have n pixels above it (or the class has fewer than n non-zero
      // pixels), so the list may not hold the whole top n -> let the materialising path redo the batch
      if (readlane64(list[k], a.n_points - 1) == 0ull && lane == 0) atomicOr(&a.cand_cnt[a.n], 1u);
      finish_topn(list[k], a.n_points, a.w, a.thresh, lane, a.out + ((size_t)face * a.l + c) * 2);
    }
  }
}

// Chunks per face.  The kernel holds 155 registers: three workgroups per CU, 768 on the chip at a time.  What counts is
// that the launch is whole rounds of those 768 -- 1024 or 1152 workgroups run a second, mostly empty round (batch 512:
// 4.4-4.7 TB/s instead of 5.6) -- and, among whole rounds, as few chunks per face as possible: every chunk starts with
// empty lists (~30 serial inserts per class until its threshold has risen) and adds n entries per class to the merge
// (batch 64: 12 chunks per face = one round, 3.8 TB/s; 24 = two rounds, 3.3; 48: 2.9; tools/bench_hbm.py).
static void decode_plan(int n, int h, int w, int* chunks, int* chunk_px) {
  const int HW = h * w, faces = n > 0 ? n : 1;
  constexpr int kResident = 768, kMaxChunks = 32;
  int best_s = 1;
  double best_u = 0.0;
  for (int s = 1; s <= kMaxChunks; ++s) {
    const long long wgs = (long long)faces * s;
    const long long rounds = (wgs + kResident - 1) / kResident;
    const double u = (double)wgs / (double)(rounds * kResident);  // how full the rounds are
    if (u > best_u * 1.05) {  // (ties and near-ties go to the fewer chunks)
      best_u = u;
      best_s = s;
    }
  }
  const int s = best_s;
  int px = (HW + s - 1) / s;
  px = (px + PT - 1) / PT * PT;
  *chunk_px = px;
  *chunks = (HW + px - 1) / px;
}

size_t decode_ws_bytes(int n, int h, int w, int l, int mode, int n_points) {
  int chunks, chunk_px;
  decode_plan(n, h, w, &chunks, &chunk_px);
  const size_t per = (mode == FLM_DECODE_ALL) ? sizeof(double) * 3
                                              : sizeof(unsigned long long) * (n_points > 0 ? n_points : 1);
  return align_up((size_t)n * chunks * l * per, 256);
}

// tau[face][class] = n-th largest of the face's wave maxima (flm_convt.hip, epilogue 4); 0 when fewer than n are
// non-zero (the consumer clamps to FLT_MIN and cand_merge_kernel checks that n keys arrived).
__global__ __launch_bounds__(256) void cand_tau_kernel(const unsigned* __restrict__ wave_max, int slots, int ld, int l,
                                                       int n_points, float* __restrict__ tau) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int face = blockIdx.x;
  const unsigned* src = wave_max + (size_t)face * slots * ld;
  for (int c = blockIdx.y * 4 + wave; c < l; c += 4 * gridDim.y) {  // one class per wave and round
    unsigned long long list = 0ull, t = 0ull;
    for (int s0 = 0; s0 < slots; s0 += 64) {
      const int sl = s0 + lane;
      const unsigned v = sl < slots ? src[(size_t)sl * ld + c] : 0u;
      const unsigned long long key = v ? (((unsigned long long)v << 32) | (unsigned)sl) : 0ull;
      if (__any(key > t)) insert_candidates(list, t, key, n_points, lane);
    }
    const unsigned long long k = readlane64(list, n_points - 1);
    if (lane == 0) tau[(size_t)face * l + c] = __uint_as_float((unsigned)(k >> 32));
  }
}

// The same threshold for n <= 8 with the reads coalesced: lanes run along the classes (the ld values of a slot are
// contiguous), three groups of threads share the slots, every thread keeps its n largest maxima in registers (a sorted
// insertion, values with multiplicity, zeros never enter), and one thread per class merges the three short lists.  The
// wave-per-class kernel above reads a slot column with a stride of ld words: 64 cache lines per load (bf16 batch 512:
// 45 -> 23 us).
template <int NMAX>
__global__ __launch_bounds__(256) void cand_tau_small_kernel(const unsigned* __restrict__ wave_max, int slots, int ld, int l,
                                                             int n_points, float* __restrict__ tau) {
  __shared__ unsigned part[3][NMAX][96];
  const int face = blockIdx.x, tid = threadIdx.x;
  const int g = tid / ld, c = tid - g * ld;   // ld <= 85: three groups fit 256 threads
  const unsigned* src = wave_max + (size_t)face * slots * ld;
  unsigned top[NMAX];
#pragma unroll
  for (int k = 0; k < NMAX; ++k) top[k] = 0u;
  if (g < 3) {
#pragma unroll 8
    for (int sl = g; sl < slots; sl += 3) {
      unsigned v = src[(size_t)sl * ld + c];
      if (v > top[NMAX - 1]) {
#pragma unroll
        for (int k = 0; k < NMAX; ++k) {  // descending; v sinks to its place, the smallest falls out
          const unsigned hi = v > top[k] ? v : top[k], lo = v > top[k] ? top[k] : v;
          top[k] = hi;
          v = lo;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < NMAX; ++k) part[g][k][c] = top[k];
  }
  __syncthreads();
  if (g == 0 && c < l) {
#pragma unroll
    for (int gg = 1; gg < 3; ++gg)
#pragma unroll
      for (int j = 0; j < NMAX; ++j) {
        unsigned v = part[gg][j][c];
        if (v > top[NMAX - 1]) {
#pragma unroll
          for (int k = 0; k < NMAX; ++k) {
            const unsigned hi = v > top[k] ? v : top[k], lo = v > top[k] ? top[k] : v;
            top[k] = hi;
            v = lo;
          }
        }
      }
    unsigned t = 0u;
#pragma unroll
    for (int k = 0; k < NMAX; ++k)
      if (k == n_points - 1) t = top[k];
    tau[(size_t)face * l + c] = __uint_as_float(t);
  }
}

int launch_cand_tau(hipStream_t s, const unsigned* wave_max, int n, int slots, int ld, int l, int n_points, float* tau) {
  if (n_points < 1 || n_points > 64 || slots < 1) {
    set_error("cand_tau: unsupported n_points=%d slots=%d", n_points, slots);
    return FLM_ERR_UNSUPPORTED;
  }
  // (one workgroup per face: below ~200 faces it leaves the chip empty and the wave-per-class kernel, 24 waves per
  // face, is faster -- 64 faces: 18 us against 32)
  if (n >= 192 && n_points <= 8 && ld <= 85 && l <= ld) {
    // NMAX = n_points would do; two instantiations keep the code small (lists longer than n only cost compares)
    if (n_points <= 4) cand_tau_small_kernel<4><<<n, 256, 0, s>>>(wave_max, slots, ld, l, n_points, tau);
    else cand_tau_small_kernel<8><<<n, 256, 0, s>>>(wave_max, slots, ld, l, n_points, tau);
    FLM_LAUNCH_CHECK("cand_tau_small_kernel");
    return FLM_OK;
  }
  cand_tau_kernel<<<dim3(n, 6), 256, 0, s>>>(wave_max, slots, ld, l, n_points, tau);
  FLM_LAUNCH_CHECK("cand_tau_kernel");
  return FLM_OK;
}

int launch_cand_merge(hipStream_t s, const unsigned long long* cand, unsigned* cand_cnt, int n, int w, int l,
                      int n_points, float thresh, int cap, double* out) {
  if (l > 68 || n_points < 1 || n_points > 64) {
    set_error("cand_merge: unsupported l=%d n_points=%d", l, n_points);
    return FLM_ERR_UNSUPPORTED;
  }
  CandMergeArgs a;
  a.cand = cand; a.cand_cnt = cand_cnt; a.n = n; a.w = w; a.l = l; a.n_points = n_points; a.cap = cap;
  a.thresh = thresh; a.out = out;
  if (n < kCandFineBatch) {
    a.cpg = 17;
    cand_merge_kernel<5, 4><<<dim3(cdiv(l, 17), n), 256, 0, s>>>(a);
  } else {
    a.cpg = 68;
    cand_merge_kernel<9, 8><<<dim3(1, n), 512, 0, s>>>(a);
  }
  FLM_LAUNCH_CHECK("cand_merge_kernel");
  return FLM_OK;
}

int launch_decode(hipStream_t s, const float* hm, int n, int h, int w, int l, int ld, int mode, int n_points,
                  float thresh, double* out, void* ws, size_t ws_bytes, float* tau_out, const unsigned* gate) {
  if (n <= 0 || h <= 0 || w <= 0 || l <= 0 || l > kMaxClasses || ld != l) {
    set_error("decode: unsupported shape n=%d h=%d w=%d l=%d (max %d landmarks)", n, h, w, l, kMaxClasses);
    return FLM_ERR_SHAPE;
  }
  if ((long long)h * w >= (1ll << 31)) {
    set_error("decode: map too large");
    return FLM_ERR_SHAPE;
  }
  if (mode == FLM_DECODE_TOPN && (n_points < 1 || n_points > 128)) {
    set_error("decode: top-n mode supports 1 <= n_points <= 128 (got %d)", n_points);
    return FLM_ERR_UNSUPPORTED;
  }
  if (mode != FLM_DECODE_ALL && mode != FLM_DECODE_TOPN) {
    set_error("decode: unknown mode %d", mode);
    return FLM_ERR_ARG;
  }
  if (reinterpret_cast<uintptr_t>(hm) & 15) {
    set_error("decode: heatmap pointer must be 16-byte aligned");
    return FLM_ERR_ARG;
  }
  if (ws_bytes < decode_ws_bytes(n, h, w, l, mode, n_points)) {
    set_error("decode: workspace too small");
    return FLM_ERR_WORKSPACE;
  }
  DecodeArgs a;
  a.hm = hm; a.n = n; a.h = h; a.w = w; a.l = l;
  decode_plan(n, h, w, &a.chunks, &a.chunk_px);
  a.mode = mode; a.n_points = n_points; a.thresh = thresh; a.part = ws; a.out = out;
  a.tau_out = (mode == FLM_DECODE_TOPN) ? tau_out : nullptr;
  a.gate = gate;
  a.vec = (((long long)h * w * l) & 3) == 0;
  const size_t lds = sizeof(float) * PT * (l | 1);
  dim3 grid(a.chunks, n);
  const bool small = l <= 68;  // 17 channels per wave
  // 68-landmark maps, 16-byte-aligned faces, n <= 64: the LDS-DMA form (a chunk stays below the 2 GiB buffer range)
  const bool dma = g_decode_dma.load(std::memory_order_relaxed) && l == DL && a.vec && (mode == FLM_DECODE_ALL || n_points <= 64) &&
                   (long long)a.chunk_px * DL * 4 < (1ll << 31);
  if (dma) {
    constexpr size_t dlds = (size_t)D_RING * D_TILE_B;
    if (mode == FLM_DECODE_ALL) {
      static FuncAttrOnce attr;
      FLM_FUNC_ATTR_ONCE(attr, (&decode_partial_dma_kernel<FLM_DECODE_ALL>), dlds);
      decode_partial_dma_kernel<FLM_DECODE_ALL><<<grid, 256, dlds, s>>>(a);
      FLM_LAUNCH_CHECK("decode_partial_dma_kernel");
      decode_merge_kernel<FLM_DECODE_ALL><<<dim3(l, n), 64, 0, s>>>(a);
    } else {
      static FuncAttrOnce attr;
      FLM_FUNC_ATTR_ONCE(attr, (&decode_partial_dma_kernel<FLM_DECODE_TOPN>), dlds);
      decode_partial_dma_kernel<FLM_DECODE_TOPN><<<grid, 256, dlds, s>>>(a);
      FLM_LAUNCH_CHECK("decode_partial_dma_kernel");
      decode_merge_kernel<FLM_DECODE_TOPN><<<dim3(l, n), 64, 0, s>>>(a);
    }
    FLM_LAUNCH_CHECK("decode_merge_kernel");
    return FLM_OK;
  }
  if (mode == FLM_DECODE_ALL) {
    if (small) decode_partial_kernel<FLM_DECODE_ALL, 17><<<grid, 256, lds, s>>>(a);
    else decode_partial_kernel<FLM_DECODE_ALL, 24><<<grid, 256, lds, s>>>(a);
    FLM_LAUNCH_CHECK("decode_partial_kernel");
    decode_merge_kernel<FLM_DECODE_ALL><<<dim3(l, n), 64, 0, s>>>(a);
  } else if (n_points > 64) {  // two list registers per lane (the reference's sweep reaches n = 81)
    if (small) decode_partial_kernel<FLM_DECODE_TOPN, 17, true><<<grid, 256, lds, s>>>(a);
    else decode_partial_kernel<FLM_DECODE_TOPN, 24, true><<<grid, 256, lds, s>>>(a);
    FLM_LAUNCH_CHECK("decode_partial_kernel");
    decode_merge_kernel<FLM_DECODE_TOPN, true><<<dim3(l, n), 64, 0, s>>>(a);
  } else {
    if (small) decode_partial_kernel<FLM_DECODE_TOPN, 17><<<grid, 256, lds, s>>>(a);
    else decode_partial_kernel<FLM_DECODE_TOPN, 24><<<grid, 256, lds, s>>>(a);
    FLM_LAUNCH_CHECK("decode_partial_kernel");
    decode_merge_kernel<FLM_DECODE_TOPN><<<dim3(l, n), 64, 0, s>>>(a);
  }
  FLM_LAUNCH_CHECK("decode_merge_kernel");
  return FLM_OK;
}

}  // namespace flm
