// Stage-0 probe for the mixed-precision head (round 3): does an MX-scaled fp8 MFMA (v_mfma_scale_f32_32x32x64_f8f6f4)
// accumulate into the SAME 32 x 32 accumulator as an f16 / bf16 MFMA, with any lane-consistent operand layout, and what
// does it cost next to them?  Also: semantics of v_cvt_scalef32_pk_fp8_f16.   hipcc --offload-arch=gfx950 -O3 -o bin/mx_probe
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));

#define CK(x)                                                                  \
  do {                                                                         \
    hipError_t e = (x);                                                        \
    if (e != hipSuccess) {                                                     \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
      exit(1);                                                                 \
    }                                                                          \
  } while (0)

// ---- correctness: one wave, C[32 x 32] = A16 B16^T (K = 16) + 2^(sa + sb - 254) A8 B8^T (K = 64)
// A16 / B16: [32][16] f16 row-major, lane (r, h) takes 8 halves at k = 8 h.  A8 / B8: [32][64] fp8 row-major, lane (r, h)
// takes the 32 bytes at k = 32 h.
__global__ void mixed_kernel(const _Float16* a16, const _Float16* b16, const uint8_t* a8, const uint8_t* b8, float* c, int sa, int sb) {
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  f16x8 fa = *reinterpret_cast<const f16x8*>(a16 + r * 16 + 8 * h);
  f16x8 fb = *reinterpret_cast<const f16x8*>(b16 + r * 16 + 8 * h);
  i32x8 qa = *reinterpret_cast<const i32x8*>(a8 + r * 64 + 32 * h);
  i32x8 qb = *reinterpret_cast<const i32x8*>(b8 + r * 64 + 32 * h);
  f32x16 acc = {};
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(qa, qb, acc, 0, 0, 0, sa, 0, sb);
  for (int g = 0; g < 16; ++g) {
    const int row = (g & 3) + 8 * (g >> 2) + 4 * h;  // A row index i; column = lane & 31 = B row index j
    c[row * 32 + r] = acc[g];
  }
}

__global__ void cvt_kernel(const _Float16* x, uint32_t* out, float scale) {
  s16x2 old = {0, 0};
  f16x2 v = {x[2 * threadIdx.x], x[2 * threadIdx.x + 1]};
  auto lo = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(old, v, scale, false);
  auto hi = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(old, v, scale, true);
  out[2 * threadIdx.x] = (uint16_t)lo[0] | ((uint32_t)(uint16_t)lo[1] << 16);
  out[2 * threadIdx.x + 1] = (uint16_t)hi[0] | ((uint32_t)(uint16_t)hi[1] << 16);
}

// ---- ds_read_b64_tr_b16: LDS holds element index e at halfword e; lane l reads the 8 bytes at byte address addr[l]
__global__ void tr_kernel(const int* addr, uint32_t* out) {
  __shared__ uint16_t lds[4096];
  for (int e = threadIdx.x; e < 4096; e += 64) lds[e] = (uint16_t)e;
  __syncthreads();
  typedef int i32x2 __attribute__((ext_vector_type(2)));
  i32x2 r;
  const unsigned a = (unsigned)(uintptr_t)lds + (unsigned)addr[threadIdx.x];
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(a) : "memory");
  out[2 * threadIdx.x] = (uint32_t)r[0];
  out[2 * threadIdx.x + 1] = (uint32_t)r[1];
}

// ---- throughput: every wave issues `iters` rounds of 8 MFMAs on 8 independent accumulators
template <int KIND>  // 0 bf16 32x32x16, 1 f16 32x32x16, 2 mx fp8 32x32x64, 3 interleaved: 4 x f16 then 1 x mx (1.5 products)
__global__ __launch_bounds__(512) void rate_kernel(float* out, int iters, int scale) {
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i)
    for (int g = 0; g < 16; ++g) acc[i][g] = 0.0f;
  f16x8 fa, fb;
  bf16x8 ba, bb;
  i32x8 qa, qb;
  for (int i = 0; i < 8; ++i) {
    fa[i] = (_Float16)(0.001f * (threadIdx.x + i));
    fb[i] = (_Float16)(0.002f * (threadIdx.x - i));
    ba[i] = (short)(0x3f80 + threadIdx.x + i);
    bb[i] = (short)(0x3f00 + threadIdx.x - i);
    qa[i] = 0x38383838 + threadIdx.x * 0x01010101 + i;
    qb[i] = 0x30303030 + threadIdx.x * 0x01000100 + i;
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if constexpr (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ba, bb, acc[i], 0, 0, 0);
      if constexpr (KIND == 1) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc[i], 0, 0, 0);
      if constexpr (KIND == 2) acc[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(qa, qb, acc[i], 0, 0, 0, scale, 0, scale);
      if constexpr (KIND == 3) {
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fb, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb, fa, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa, fa, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb, fb, acc[i], 0, 0, 0);
        acc[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(qa, qb, acc[i], 0, 0, 0, scale, 0, scale);
      }
    }
  }
  float s = 0.0f;
  for (int i = 0; i < 8; ++i)
    for (int g = 0; g < 16; ++g) s += acc[i][g];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static float fp8_e4m3_decode(uint8_t v) {
  const int sign = v >> 7, e = (v >> 3) & 15, m = v & 7;
  float x;
  if (e == 0) x = ldexpf((float)m, -9);  // subnormal: m * 2^-3 * 2^-6
  else if (e == 15 && m == 7) x = NAN;
  else x = ldexpf(1.0f + m / 8.0f, e - 7);
  return sign ? -x : x;
}

template <int KIND>
static void rate(const char* name, int threads, double flop_per_mfma_round) {
  float* out;
  const int blocks = 256 * 4;
  CK(hipMalloc(&out, (size_t)blocks * threads * 4));
  const int iters = 2000;
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  hipLaunchKernelGGL(rate_kernel<KIND>, dim3(blocks), dim3(threads), 0, 0, out, 10, 127);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  hipLaunchKernelGGL(rate_kernel<KIND>, dim3(blocks), dim3(threads), 0, 0, out, iters, 127);
  CK(hipEventRecord(b));
  CK(hipDeviceSynchronize());
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  const double waves = (double)blocks * threads / 64;
  const double flop = waves * iters * 8 * flop_per_mfma_round;
  printf("%-34s %4d thr/WG: %8.3f ms  %8.1f TFLOP/s (bf16-equivalent flops)\n", name, threads, ms, flop / (ms * 1e-3) / 1e12);
  CK(hipFree(out));
}

int main() {
  srand(1);
  // ---------------- correctness
  std::vector<_Float16> a16(32 * 16), b16(32 * 16);
  std::vector<uint8_t> a8(32 * 64), b8(32 * 64);
  for (auto& x : a16) x = (_Float16)((rand() % 2001 - 1000) / 500.0f);
  for (auto& x : b16) x = (_Float16)((rand() % 2001 - 1000) / 500.0f);
  for (auto& x : a8) {
    do x = (uint8_t)(rand() & 255); while ((x & 0x7f) == 0x7f);  // no NaN
  }
  for (auto& x : b8) {
    do x = (uint8_t)(rand() & 255); while ((x & 0x7f) == 0x7f);
  }
  _Float16 *da16, *db16;
  uint8_t *da8, *db8;
  float* dc;
  CK(hipMalloc(&da16, a16.size() * 2));
  CK(hipMalloc(&db16, b16.size() * 2));
  CK(hipMalloc(&da8, a8.size()));
  CK(hipMalloc(&db8, b8.size()));
  CK(hipMalloc(&dc, 32 * 32 * 4));
  CK(hipMemcpy(da16, a16.data(), a16.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(db16, b16.data(), b16.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(da8, a8.data(), a8.size(), hipMemcpyHostToDevice));
  CK(hipMemcpy(db8, b8.data(), b8.size(), hipMemcpyHostToDevice));
  for (int trial = 0; trial < 3; ++trial) {
    const int ea = trial == 0 ? 127 : trial == 1 ? 120 : 110, eb = trial == 2 ? 118 : 127;  // E8M0 exponents
    const int sa = ea * 0x01010101, sb = eb * 0x01010101;
    hipLaunchKernelGGL(mixed_kernel, dim3(1), dim3(64), 0, 0, da16, db16, da8, db8, dc, sa, sb);
    CK(hipDeviceSynchronize());
    std::vector<float> c(32 * 32);
    CK(hipMemcpy(c.data(), dc, c.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0, scale_c = 0;
    for (int i = 0; i < 32; ++i)
      for (int j = 0; j < 32; ++j) {
        double hi = 0, lo = 0;
        for (int k = 0; k < 16; ++k) hi += (double)(float)a16[i * 16 + k] * (double)(float)b16[j * 16 + k];
        for (int k = 0; k < 64; ++k) lo += (double)fp8_e4m3_decode(a8[i * 64 + k]) * (double)fp8_e4m3_decode(b8[j * 64 + k]);
        const double want = hi + lo * ldexp(1.0, ea - 127 + eb - 127);
        worst = fmax(worst, fabs(want - c[i * 32 + j]));
        scale_c = fmax(scale_c, fabs(want));
      }
    printf("mixed f16 + mx-fp8 accumulate, scales 2^%d 2^%d: max abs err %.3e of max |C| %.3e  -> %s\n", ea - 127, eb - 127, worst, scale_c,
           worst <= 2e-5 * scale_c ? "OK" : "MISMATCH");
  }
  // ---------------- cvt semantics
  {
    std::vector<_Float16> x = {(_Float16)1.0f, (_Float16)-1.5f, (_Float16)0.0625f, (_Float16)300.0f, (_Float16)0.001f, (_Float16)448.0f, (_Float16)17.0f, (_Float16)-0.3f};
    _Float16* dx;
    uint32_t* dout;
    CK(hipMalloc(&dx, x.size() * 2));
    CK(hipMalloc(&dout, x.size() * 4));
    CK(hipMemcpy(dx, x.data(), x.size() * 2, hipMemcpyHostToDevice));
    for (float scale : {1.0f, 4.0f, 0.25f}) {
      hipLaunchKernelGGL(cvt_kernel, dim3(1), dim3(x.size() / 2), 0, 0, dx, dout, scale);
      CK(hipDeviceSynchronize());
      std::vector<uint32_t> o(x.size());
      CK(hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost));
      printf("cvt_scalef32_pk_fp8_f16 scale %.2f:", scale);
      for (size_t t = 0; t < x.size() / 2; ++t) {
        const uint32_t lo = o[2 * t], hi = o[2 * t + 1];
        printf("  (%g, %g) -> sel_lo word %08x = (%g, %g) | sel_hi word %08x", (float)x[2 * t], (float)x[2 * t + 1], lo, fp8_e4m3_decode(lo & 255),
               fp8_e4m3_decode((lo >> 8) & 255), hi);
      }
      printf("\n");
    }
  }
  // ---------------- transpose read semantics
  {
    int* daddr;
    uint32_t* dout;
    CK(hipMalloc(&daddr, 64 * 4));
    CK(hipMalloc(&dout, 128 * 4));
    for (int pattern = 0; pattern < 2; ++pattern) {
      std::vector<int> addr(64);
      // pattern 0: lane l reads bytes [8 l, 8 l + 8) (linear).  pattern 1: a [t][v] tile with 64-byte rows (32 v): group g = l >> 4,
      // lane q = l & 15 reads row t = (q >> 2) + 4 * (g >> 1)... kept simple: row = q >> 2, v chunk = (q & 3) + 4 * (g & 1), + 1024 bytes for g >= 2
      for (int l = 0; l < 64; ++l) {
        const int g = l >> 4, q = l & 15;
        addr[l] = pattern == 0 ? 8 * l : (q >> 2) * 64 + ((q & 3) + 4 * (g & 1)) * 8 + (g >> 1) * 1024;
      }
      CK(hipMemcpy(daddr, addr.data(), 256, hipMemcpyHostToDevice));
      hipLaunchKernelGGL(tr_kernel, dim3(1), dim3(64), 0, 0, daddr, dout);
      CK(hipDeviceSynchronize());
      std::vector<uint32_t> o(128);
      CK(hipMemcpy(o.data(), dout, 512, hipMemcpyDeviceToHost));
      printf("ds_read_b64_tr_b16 pattern %d (element indices each lane received):\n", pattern);
      for (int l = 0; l < 64; ++l) {
        printf("  l%02d@%4d: %4u %4u %4u %4u%s", l, addr[l], o[2 * l] & 0xffff, o[2 * l] >> 16, o[2 * l + 1] & 0xffff, o[2 * l + 1] >> 16, (l & 3) == 3 ? "\n" : "");
      }
    }
  }
  // ---------------- rates (flops per round of one MFMA each, in real flops; the mixed kind counts 1.5 products of 32x32x64)
  for (int threads : {256, 512}) {
    rate<0>("bf16 32x32x16", threads, 2.0 * 32 * 32 * 16);
    rate<1>("f16 32x32x16", threads, 2.0 * 32 * 32 * 16);
    rate<2>("mx fp8 32x32x64 (real flops)", threads, 2.0 * 32 * 32 * 64);
    rate<3>("4 x f16 + 1 x mx fp8 (K = 64 + 64)", threads, 2.0 * 32 * 32 * 128);
  }
  return 0;
}
