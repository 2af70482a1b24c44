// Micro-benchmark: how many VALU / transcendental / LDS-read instructions of the SAME wave hide behind an
// fp32 MFMA on gfx950?  (Design input for software-pipelining the edge kernels: the epilogue of one tile and the
// A-operand SiLU of the next K step interleaved between the MFMAs of the current one.)
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_shadow.hip -o tools/bin/mfma_shadow && tools/bin/mfma_shadow
//
// Every variant runs ITER iterations of 8 independent MFMAs (8 accumulators), each followed by NV filler
// instructions on independent registers, inside one asm block (the compiler cannot reorder it).  Reported: shader
// cycles per MFMA (s_memtime) for one wave per SIMD (256 threads / CU) and two (512 threads / CU).
// 64 = the matrix pipe's issue interval for v_mfma_f32_32x32x2_f32, 32 for v_mfma_f32_16x16x4_f32.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

// filler kinds
enum { K_FMA = 0, K_EXP = 1, K_SILU = 2, K_LDS = 3, K_LDSW = 4, K_PKFMA = 5, K_PKMUL = 6, K_RCP = 7 };

template <int KIND, int NV>
__device__ __forceinline__ void fill(float& f0, float& f1, float& f2, float& f3, float f4, f32x4& l0, unsigned la, f32x2& p0, f32x2& p1, f32x2 p2) {
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (KIND == K_FMA) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if (i % 4 == 0) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f0) : "v"(f4));
      if (i % 4 == 1) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f1) : "v"(f4));
      if (i % 4 == 2) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f2) : "v"(f4));
      if (i % 4 == 3) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f3) : "v"(f4));
    }
  } else if constexpr (KIND == K_EXP) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if (i % 4 == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(f0));
      if (i % 4 == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(f1));
      if (i % 4 == 2) asm volatile("v_exp_f32 %0, %0" : "+v"(f2));
      if (i % 4 == 3) asm volatile("v_exp_f32 %0, %0" : "+v"(f3));
    }
  } else if constexpr (KIND == K_SILU) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      float& t = (i & 1) ? f2 : f0;
      float& u = (i & 1) ? f3 : f1;
      asm volatile("v_mul_f32 %0, 0xbfb8aa3b, %2\n v_exp_f32 %0, %0\n v_add_f32 %0, 1.0, %0\n v_rcp_f32 %0, %0\n v_mul_f32 %1, %2, %0"
                   : "+v"(t), "+v"(u) : "v"(f4));
    }
  } else if constexpr (KIND == K_PKFMA || KIND == K_PKMUL) {
    // packed fp32: two values per lane and instruction (64-bit aligned register pairs)
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      f32x2& t = (i & 1) ? p1 : p0;
      if constexpr (KIND == K_PKFMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(t) : "v"(p2));
      else asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(t) : "v"(p2));
    }
  } else if constexpr (KIND == K_RCP) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      if (i % 4 == 0) asm volatile("v_rcp_f32 %0, %0" : "+v"(f0));
      if (i % 4 == 1) asm volatile("v_rcp_f32 %0, %0" : "+v"(f1));
      if (i % 4 == 2) asm volatile("v_rcp_f32 %0, %0" : "+v"(f2));
      if (i % 4 == 3) asm volatile("v_rcp_f32 %0, %0" : "+v"(f3));
    }
  } else if constexpr (KIND == K_LDS) {
#pragma unroll
    for (int i = 0; i < NV; ++i) asm volatile("ds_read_b128 %0, %1" : "+v"(l0) : "v"(la));
  } else {
#pragma unroll
    for (int i = 0; i < NV; ++i) asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "+v"(l0) : "v"(la));
  }
  __builtin_amdgcn_sched_barrier(0);
}

template <int KIND, int NV>
__global__ __launch_bounds__(512) void k32(float* out, unsigned long long* cyc, int iters) {
  __shared__ float lds[4096];
  lds[threadIdx.x] = threadIdx.x;
  __syncthreads();
  f32x16 c[8];
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) c[i][r] = 0.f;
  float a = out[threadIdx.x & 63], b = out[(threadIdx.x & 63) + 64];
  float f0 = a, f1 = b, f2 = a + 1.f, f3 = b + 1.f, f4 = 0.999f;
  f32x4 l0 = {0.f, 0.f, 0.f, 0.f};
  f32x2 p0 = {a, b}, p1 = {b, a}, p2 = {0.999f, 0.998f};
  unsigned la = (threadIdx.x & 63) * 16;
  const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      c[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c[i], 0, 0, 0);
      fill<KIND, NV>(f0, f1, f2, f3, f4, l0, la, p0, p1, p2);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n s_nop 7\n s_nop 7" ::: "memory");
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = f0 + f1 + f2 + f3 + l0[0] + p0[0] + p0[1] + p1[0] + p1[1];
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += c[i][r];
  if (s == 12345.678f) out[threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

// the same with v_mfma_f32_16x16x4_f32 (4 accumulator registers, 32-cycle issue interval)
template <int KIND, int NV>
__global__ __launch_bounds__(512) void k16(float* out, unsigned long long* cyc, int iters) {
  __shared__ float lds[4096];
  lds[threadIdx.x] = threadIdx.x;
  __syncthreads();
  f32x4 c[8];
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 4; ++r) c[i][r] = 0.f;
  float a = out[threadIdx.x & 63], b = out[(threadIdx.x & 63) + 64];
  float f0 = a, f1 = b, f2 = a + 1.f, f3 = b + 1.f, f4 = 0.999f;
  f32x4 l0 = {0.f, 0.f, 0.f, 0.f};
  f32x2 p0 = {a, b}, p1 = {b, a}, p2 = {0.999f, 0.998f};
  unsigned la = (threadIdx.x & 63) * 16;
  const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      c[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c[i], 0, 0, 0);
      fill<KIND, NV>(f0, f1, f2, f3, f4, l0, la, p0, p1, p2);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n s_nop 7\n s_nop 7" ::: "memory");
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = f0 + f1 + f2 + f3 + l0[0] + p0[0] + p0[1] + p1[0] + p1[1];
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 4; ++r) s += c[i][r];
  if (s == 12345.678f) out[threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

typedef void (*kern_t)(float*, unsigned long long*, int);

int main() {
  float* out;
  unsigned long long* cyc;
  CK(hipMalloc(&out, 4096 * 4));
  CK(hipMemset(out, 0, 4096 * 4));
  CK(hipMalloc(&cyc, 256 * 16 * 8));
  const int iters = 2000;
  struct V { const char* name; kern_t k; int fill; } vs[] = {
      {"32x32x2: bare", k32<K_FMA, 0>, 0}, {"32x32x2 + 1 v_fma", k32<K_FMA, 1>, 1}, {"32x32x2 + 2 v_fma", k32<K_FMA, 2>, 2},
      {"32x32x2 + 4 v_fma", k32<K_FMA, 4>, 4}, {"32x32x2 + 8 v_fma", k32<K_FMA, 8>, 8}, {"32x32x2 + 12 v_fma", k32<K_FMA, 12>, 12},
      {"32x32x2 + 16 v_fma", k32<K_FMA, 16>, 16}, {"32x32x2 + 24 v_fma", k32<K_FMA, 24>, 24},
      {"32x32x2 + 1 v_pk_fma", k32<K_PKFMA, 1>, 1}, {"32x32x2 + 2 v_pk_fma", k32<K_PKFMA, 2>, 2}, {"32x32x2 + 4 v_pk_fma", k32<K_PKFMA, 4>, 4},
      {"32x32x2 + 8 v_pk_fma", k32<K_PKFMA, 8>, 8}, {"32x32x2 + 16 v_pk_fma", k32<K_PKFMA, 16>, 16},
      {"32x32x2 + 4 v_pk_mul", k32<K_PKMUL, 4>, 4}, {"32x32x2 + 8 v_pk_mul", k32<K_PKMUL, 8>, 8},
      {"32x32x2 + 4 v_rcp", k32<K_RCP, 4>, 4}, {"32x32x2 + 8 v_exp", k32<K_EXP, 8>, 8},
      {"32x32x2 + 1 v_exp", k32<K_EXP, 1>, 1}, {"32x32x2 + 2 v_exp", k32<K_EXP, 2>, 2}, {"32x32x2 + 4 v_exp", k32<K_EXP, 4>, 4},
      {"32x32x2 + 1 SiLU (mul exp add rcp mul)", k32<K_SILU, 1>, 5}, {"32x32x2 + 2 SiLU", k32<K_SILU, 2>, 10},
      {"32x32x2 + 1 ds_read_b128 (no wait)", k32<K_LDS, 1>, 1}, {"32x32x2 + 1 ds_read_b128 + lgkmcnt(0)", k32<K_LDSW, 1>, 1},
      {"16x16x4: bare", k16<K_FMA, 0>, 0}, {"16x16x4 + 2 v_fma", k16<K_FMA, 2>, 2}, {"16x16x4 + 4 v_fma", k16<K_FMA, 4>, 4},
      {"16x16x4 + 8 v_fma", k16<K_FMA, 8>, 8}, {"16x16x4 + 1 SiLU", k16<K_SILU, 1>, 5},
      {"16x16x4 + 1 ds_read_b128 (no wait)", k16<K_LDS, 1>, 1}};
  printf("| variant | fillers / MFMA | cycles / MFMA, 1 wave per SIMD | cycles / MFMA per wave, 2 waves per SIMD | pipe interval seen with 2 waves |\n|---|---|---|---|---|\n");
  for (auto& v : vs) {
    double res[2];
    for (int w = 0; w < 2; ++w) {
      const int threads = w == 0 ? 256 : 512;
      hipLaunchKernelGGL(v.k, dim3(256), dim3(threads), 0, 0, out, cyc, 10);
      CK(hipDeviceSynchronize());
      hipLaunchKernelGGL(v.k, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
      CK(hipDeviceSynchronize());
      std::vector<unsigned long long> h(256 * (threads / 64));
      CK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
      double s = 0;
      for (auto c : h) s += (double)c;
      res[w] = s / h.size() / (8.0 * iters);
    }
    printf("| %s | %d | %.1f | %.1f | %.1f |\n", v.name, v.fill, res[0], res[1], res[1] / 2.0);
  }
  return 0;
}
