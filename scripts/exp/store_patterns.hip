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

struct Cols { long* i[5]; float* f[7]; long n; const int* src[4]; };  // src: four 4-byte input columns (ids, labels, old, ref)

template <bool NT, class V> __device__ __forceinline__ void st(V* p, V v) {
  if constexpr (NT) __builtin_nontemporal_store(v, p); else *p = v;
}

template <int MODE, bool NT>
__global__ __launch_bounds__(256) void k(Cols c) {
  const long nthreads = (long)gridDim.x * 256;
  const int lane = threadIdx.x & 63;
  if (MODE == 3) {  // C + the kernel's 16 bytes of input per token, contiguous (round 4: the read + write floor of this store order)
    typedef int i2 __attribute__((ext_vector_type(2)));
    for (long g = (long)blockIdx.x * 256 + threadIdx.x; g < c.n / 2; g += nthreads) {
      const long t = g * 2;
      const i2 a0 = *(const i2*)(c.src[0] + t), a1 = *(const i2*)(c.src[1] + t), a2 = *(const i2*)(c.src[2] + t), a3 = *(const i2*)(c.src[3] + t);
      st<NT>((l2*)(c.i[0] + t), l2{a0.x, a0.y});
      st<NT>((l2*)(c.i[1] + t), l2{a1.x, a1.y});
#pragma unroll
      for (int a = 2; a < 5; ++a) st<NT>((l2*)(c.i[a] + t), l2{t, t + 1});
      st<NT>((f2*)(c.f[0] + t), f2{__int_as_float(a2.x), __int_as_float(a2.y)});
      st<NT>((f2*)(c.f[1] + t), f2{__int_as_float(a3.x), __int_as_float(a3.y)});
#pragma unroll
      for (int a = 2; a < 7; ++a) st<NT>((f2*)(c.f[a] + t), f2{(float)t, 1.0f});
    }
    return;
  }
  if (MODE == 4) {  // column bursts: a workgroup owns 2048 consecutive tokens and writes them one column after the other
    const long chunks = c.n / 2048;
    for (long ch = blockIdx.x; ch < chunks; ch += gridDim.x) {
      const long t0 = ch * 2048;
#pragma unroll
      for (int a = 0; a < 5; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {  // 2048 int64 = 16 KB = 4 x (256 lanes x 16 B)
          const long t = t0 + r * 512 + threadIdx.x * 2;
          st<NT>((l2*)(c.i[a] + t), l2{t, t + 1});
        }
#pragma unroll
      for (int a = 0; a < 7; ++a)
#pragma unroll
        for (int r = 0; r < 2; ++r) {  // 2048 fp32 = 8 KB = 2 x (256 lanes x 16 B)
          const long t = t0 + r * 1024 + threadIdx.x * 4;
          st<NT>((f4*)(c.f[a] + t), f4{(float)t, 1.0f, 2.0f, 3.0f});
        }
    }
    return;
  }
  if (MODE == 5 || MODE == 6) {  // D2: column bursts with ONE token mapping for all columns: lane owns tokens r * 512 + 2 * tid + {0, 1}, r = 0..3
    typedef int i2 __attribute__((ext_vector_type(2)));
    const long chunks = c.n / 2048;
    for (long ch = blockIdx.x; ch < chunks; ch += gridDim.x) {
      const long t0 = ch * 2048 + threadIdx.x * 2;
      i2 in[4][4];
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) in[a][r] = MODE == 6 ? *(const i2*)(c.src[a] + t0 + r * 512) : i2{(int)t0, r};
#pragma unroll
      for (int a = 0; a < 5; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const long t = t0 + r * 512;
          st<NT>((l2*)(c.i[a] + t), a < 2 ? l2{in[a][r].x, in[a][r].y} : l2{t, t + 1});
        }
#pragma unroll
      for (int a = 0; a < 7; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const long t = t0 + r * 512;
          st<NT>((f2*)(c.f[a] + t), a < 2 ? f2{__int_as_float(in[2 + a][r].x), __int_as_float(in[2 + a][r].y)} : f2{(float)t, 1.0f});
        }
    }
    return;
  }
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
  for (int a = 0; a < 4; ++a) { hipMalloc((void**)&c.src[a], c.n * 4); hipMemset((void*)c.src[a], 0, c.n * 4); }
  const double bytes = (double)c.n * 68;
  for (int blocks : {2048, 4096, 16384}) {
    printf("blocks %d\n", blocks);
    float t;
    t = run<0, false>(c, blocks); printf("  A plain  %7.1f us  %6.0f GB/s\n", t * 1e3, bytes / t / 1e6);
    t = run<0, true>(c, blocks);  printf("  A nt     %7.1f us  %6.0f GB/s\n", t * 1e3, bytes / t / 1e6);
    t = run<1, false>(c, blocks); printf("  B plain  %7.1f us  %6.0f GB/s\n", t * 1e3, bytes / t / 1e6);
    t = run<1, true>(c, blocks);  printf("  B nt     %7.1f us  %6.0f GB/s\n", t * 1e3, bytes / t / 1e6);
    t = run<2, false>(c, blocks); printf("  C plain  %7.1f us  %6.0f GB/s\n", t * 1e3, bytes / t / 1e6);
    t = run<2, true>(c, blocks);  printf("  C nt     %7.1f us  %6.0f GB/s\n", t * 1e3, bytes / t / 1e6);
    t = run<3, true>(c, blocks);  printf("  C nt + 16 B/token of contiguous reads  %7.1f us  %6.0f GB/s (84 B/token)\n", t * 1e3, (double)c.n * 84 / t / 1e6);
    t = run<3, false>(c, blocks); printf("  C plain + reads                        %7.1f us  %6.0f GB/s (84 B/token)\n", t * 1e3, (double)c.n * 84 / t / 1e6);
    t = run<4, true>(c, blocks);  printf("  D nt (column bursts of 2048 tokens)    %7.1f us  %6.0f GB/s\n", t * 1e3, bytes / t / 1e6);
    t = run<5, true>(c, blocks);  printf("  D2 nt (bursts, 2 tokens per lane everywhere) %7.1f us  %6.0f GB/s\n", t * 1e3, bytes / t / 1e6);
    t = run<5, false>(c, blocks); printf("  D2 plain                                     %7.1f us  %6.0f GB/s\n", t * 1e3, bytes / t / 1e6);
    t = run<6, true>(c, blocks);  printf("  D2 nt + 16 B/token of reads                  %7.1f us  %6.0f GB/s (84 B/token)\n", t * 1e3, (double)c.n * 84 / t / 1e6);
    t = run<6, false>(c, blocks); printf("  D2 plain + reads                             %7.1f us  %6.0f GB/s (84 B/token)\n", t * 1e3, (double)c.n * 84 / t / 1e6);
    t = run<4, false>(c, blocks); printf("  D plain                                %7.1f us  %6.0f GB/s\n", t * 1e3, bytes / t / 1e6);
  }
  return 0;
}
