// Developer tool: issue rate of dependent vs independent vector instructions for one or two waves per SIMD (what a
// softmax chain costs an in-order wave).  hipcc --offload-arch=gfx950 -O3 tools/valu_latency.hip -o valu_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int ITERS = 2048, UNROLL = 32;

template <int MODE>
__global__ void k(float* out, float seed) {
  float a = seed + threadIdx.x, b = seed * 2.f, c = seed * 3.f, d = seed * 4.f;
  typedef float f2 __attribute__((ext_vector_type(2)));
  f2 pa = {a, b}, pb = {b, c}, pc = {c, d}, pd = {d, a}, pe = {seed, seed};
  unsigned sc = blockIdx.x;
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int u = 0; u < UNROLL / 4; ++u) {
      if constexpr (MODE == 0) {  // one dependent chain of v_add_f32
        asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1" : "+v"(a) : "v"(b));
      } else if constexpr (MODE == 1) {  // four independent chains
        asm volatile("v_add_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_add_f32 %2, %2, %4\n v_add_f32 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(seed));
      } else if constexpr (MODE == 2) {  // dependent fma -> exp pairs (each exp reads the fma before it), pairs independent
        asm volatile("v_fma_f32 %0, %2, %3, %3\n v_exp_f32 %0, %0\n v_fma_f32 %1, %2, %3, %3\n v_exp_f32 %1, %1" : "+v"(a), "+v"(b) : "v"(c), "v"(d));
      } else if constexpr (MODE == 3) {  // two dependent chains interleaved
        asm volatile("v_add_f32 %0, %0, %2\n v_add_f32 %1, %1, %2\n v_add_f32 %0, %0, %2\n v_add_f32 %1, %1, %2" : "+v"(a), "+v"(b) : "v"(c));
      } else if constexpr (MODE == 4) {  // dependent v_max3
        asm volatile("v_max3_f32 %0, %0, %1, %2\n v_max3_f32 %0, %0, %1, %2\n v_max3_f32 %0, %0, %1, %2\n v_max3_f32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));
      } else if constexpr (MODE == 5) {  // dependent exp chain
        asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %0, %0\n v_exp_f32 %0, %0\n v_exp_f32 %0, %0" : "+v"(a));
      } else if constexpr (MODE == 6) {  // independent exps
        asm volatile("v_exp_f32 %0, %4\n v_exp_f32 %1, %4\n v_exp_f32 %2, %4\n v_exp_f32 %3, %4" : "=v"(a), "=v"(b), "=v"(c), "=v"(d) : "v"(seed));
      } else if constexpr (MODE == 8) {  // packed fma, four independent register pairs
        asm volatile("v_pk_fma_f32 %0, %0, %4, %4\n v_pk_fma_f32 %1, %1, %4, %4\n v_pk_fma_f32 %2, %2, %4, %4\n v_pk_fma_f32 %3, %3, %4, %4" : "+v"(pa), "+v"(pb), "+v"(pc), "+v"(pd) : "v"(pe));
      } else if constexpr (MODE == 9) {  // packed mul
        asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4" : "+v"(pa), "+v"(pb), "+v"(pc), "+v"(pd) : "v"(pe));
      } else if constexpr (MODE == 10) {  // v_sub_u32 + v_max3_i32 mix (the record test)
        asm volatile("v_sub_u32 %0, %2, %3\n v_sub_u32 %1, %3, %2\n v_max3_i32 %2, %2, %0, %1\n v_sub_u32 %0, %2, %3" : "+v"(a), "+v"(b), "+v"(c) : "v"(d));
      } else if constexpr (MODE == 11) {  // s_nop 0 between adds
        asm volatile("v_add_f32 %0, %0, %1\n s_nop 0\n v_add_f32 %0, %0, %1\n s_nop 0" : "+v"(a) : "v"(b));
      } else if constexpr (MODE == 12) {  // scalar instructions only
        asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1\n s_add_u32 %0, %0, 1" : "+s"(sc));
      } else if constexpr (MODE == 13) {  // vector / scalar alternating (independent)
        asm volatile("v_add_f32 %0, %0, %2\n s_add_u32 %1, %1, 1\n v_add_f32 %0, %0, %2\n s_add_u32 %1, %1, 1" : "+v"(a), "+s"(sc) : "v"(b));
      } else if constexpr (MODE == 7) {  // exp then an independent add, alternating
        asm volatile("v_exp_f32 %0, %2\n v_add_f32 %1, %1, %2\n v_exp_f32 %0, %2\n v_add_f32 %1, %1, %2" : "+v"(a), "+v"(b) : "v"(c));
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + pa[0] + pb[1] + pc[0] + pd[1] + (float)sc;
}

template <int MODE>
void run(const char* what, int threads) {
  float* out;
  CHECK(hipMalloc(&out, 256 * 1024 * 4));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  k<MODE><<<256, threads>>>(out, 1.0f);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  k<MODE><<<256, threads>>>(out, 1.0f);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double n = (double)ITERS * UNROLL;
  printf("%-52s %d waves/SIMD: %.3f ms  = %.2f ns per instruction and wave (x clock GHz = cycles)\n", what, threads / 256, ms, ms * 1e6 / n);
  CHECK(hipFree(out));
}

int main() {
  for (int t : {256, 512, 1024}) {
    run<0>("dependent v_add_f32 chain", t);
    run<1>("four independent v_add_f32 chains", t);
    run<3>("two dependent v_add_f32 chains interleaved", t);
    run<4>("dependent v_max3_f32 chain", t);
    run<2>("fma -> exp pairs (exp reads the fma before it)", t);
    run<5>("dependent v_exp_f32 chain", t);
    run<6>("independent v_exp_f32", t);
    run<7>("v_exp_f32 / independent v_add_f32 alternating", t);
    run<8>("independent v_pk_fma_f32 (two values each)", t);
    run<9>("independent v_pk_mul_f32 (two values each)", t);
    run<10>("v_sub_u32 x3 + v_max3_i32", t);
    run<11>("v_add_f32 / s_nop 0 alternating (per instruction, nops counted)", t);
    run<12>("s_add_u32 chain", t);
    run<13>("v_add_f32 / s_add_u32 alternating", t);
  }
  return 0;
}
