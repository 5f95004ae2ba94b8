// Experiment (not part of libprl.so): how fast can 5 x int64 + 7 x fp32 output columns of 33.5 M
// tokens be WRITTEN on MI355X, as a function of the store pattern?  Build: hipcc -O3 --offload-arch=gfx950
//   A: every lane owns 4 consecutive tokens; int64 columns as two 16-B stores at a 32-B lane stride
//      (what pack_collate_kernel does), fp32 columns as one 16-B store
//   B: int64 columns written as two wave-contiguous 1 KB runs (lane L writes tokens 2L, 2L+1 of the
//      first / second half of the wave's 256 tokens)
//   C: 2 tokens per lane: int64 one 16-B store, fp32 one 8-B store (all wave-contiguous)
// each with plain and non-temporal stores.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef long l2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

struct Cols { long* i[5]; float* f[7]; long n; };

template <bool NT, class V> __device__ __forceinline__ void st(V* p, V v) {
  if constexpr (NT) __builtin_nontemporal_store(v, p); else *p = v;
}

template <int MODE, bool NT>
__global__ __launch_bounds__(256) void k(Cols c) {
  const long nthreads = (long)gridDim.x * 256;
  const int lane = threadIdx.x & 63;
  if (MODE == 2) {
    for (long g = (long)blockIdx.x * 256 + threadIdx.x; g < c.n / 2; g += nthreads) {
      const long t = g * 2;
#pragma unroll
      for (int a = 0; a < 5; ++a) st<NT>((l2*)(c.i[a] + t), l2{t, t + 1});
#pragma unroll
      for (int a = 0; a < 7; ++a) st<NT>((f2*)(c.f[a] + t), f2{(float)t, 1.0f});
    }
    return;
  }
  for (long g = (long)blockIdx.x * 256 + threadIdx.x; g < c.n / 4; g += nthreads) {
    const long t = g * 4;
    const long wave_t = (g - lane) * 4;
#pragma unroll
    for (int a = 0; a < 5; ++a) {
      if (MODE == 0) {
        st<NT>((l2*)(c.i[a] + t), l2{t, t + 1});
        st<NT>((l2*)(c.i[a] + t + 2), l2{t + 2, t + 3});
      } else {
        st<NT>((l2*)(c.i[a] + wave_t + 2 * lane), l2{t, t + 1});
        st<NT>((l2*)(c.i[a] + wave_t + 128 + 2 * lane), l2{t + 2, t + 3});
      }
    }
#pragma unroll
    for (int a = 0; a < 7; ++a) st<NT>((f4*)(c.f[a] + t), f4{(float)t, 1.0f, 2.0f, 3.0f});
  }
}

template <int MODE, bool NT> float run(Cols c, int blocks) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  float best = 1e9f;
  for (int it = 0; it < 6; ++it) {
    hipEventRecord(a); hipLaunchKernelGGL((k<MODE, NT>), dim3(blocks), dim3(256), 0, 0, c); hipEventRecord(b);
    hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); if (it && ms < best) best = ms;
  }
  return best;
}

int main() {
  Cols c; c.n = 33554432;
  for (int a = 0; a < 5; ++a) hipMalloc(&c.i[a], c.n * 8);
  for (int a = 0; a < 7; ++a) hipMalloc(&c.f[a], c.n * 4);
  const double bytes = (double)c.n * 68;
  for (int blocks : {2048, 8192}) {
    printf("blocks %d\n", blocks);
    float t;
    t = run<0, false>(c, blocks); printf("  A plain  %7.1f us  %6.0f GB/s\n", t * 1e3, bytes / t / 1e6);
    t = run<0, true>(c, blocks);  printf("  A nt     %7.1f us  %6.0f GB/s\n", t * 1e3, bytes / t / 1e6);
    t = run<1, false>(c, blocks); printf("  B plain  %7.1f us  %6.0f GB/s\n", t * 1e3, bytes / t / 1e6);
    t = run<1, true>(c, blocks);  printf("  B nt     %7.1f us  %6.0f GB/s\n", t * 1e3, bytes / t / 1e6);
    t = run<2, false>(c, blocks); printf("  C plain  %7.1f us  %6.0f GB/s\n", t * 1e3, bytes / t / 1e6);
    t = run<2, true>(c, blocks);  printf("  C nt     %7.1f us  %6.0f GB/s\n", t * 1e3, bytes / t / 1e6);
  }
  return 0;
}
