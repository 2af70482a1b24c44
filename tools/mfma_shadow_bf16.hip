// Micro-benchmark: what hides behind v_mfma_f32_32x32x16_bf16 on gfx950 (design input of the emulated-fp32 edge kernels)?
//   (a) fillers of the SAME wave after every MFMA (1 wave / SIMD and 2 waves / SIMD running the same stream),
//   (b) ROLES: waves 0-3 of a 512-thread workgroup issue only MFMAs, waves 4-7 only vector instructions -- each role's
//       time alone and together (do the two waves of a SIMD overlap their matrix and vector work at all?).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_shadow_bf16.hip -o tools/bin/mfma_shadow_bf16
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

enum { K_FMA = 0, K_EXP = 1, K_PKFMA = 2, K_LDS = 3, K_CVT = 4, K_MIX = 5 };

template <int KIND, int NV>
__device__ __forceinline__ void fill(float& f0, float& f1, float& f2, float& f3, float f4, f32x4& l0, unsigned la, f32x2& p0, f32x2& p1, f32x2 p2,
                                     unsigned& u0) {
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    float& t = (i % 4 == 0) ? f0 : (i % 4 == 1) ? f1 : (i % 4 == 2) ? f2 : f3;
    if constexpr (KIND == K_FMA) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(t) : "v"(f4));
    else if constexpr (KIND == K_EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(t));
    else if constexpr (KIND == K_PKFMA) { f32x2& q = (i & 1) ? p1 : p0; asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(q) : "v"(p2)); }
    else if constexpr (KIND == K_LDS) asm volatile("ds_read_b128 %0, %1" : "+v"(l0) : "v"(la));
    else if constexpr (KIND == K_CVT) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u0) : "v"(f0), "v"(f1));
    else {   // the activation mix of the edge kernel: per value add, 2 fma, add, mul, exp, add, rcp, mul (+ split)
      if (i % 8 == 5) asm volatile("v_exp_f32 %0, %0" : "+v"(t));
      else if (i % 8 == 7) asm volatile("v_rcp_f32 %0, %0" : "+v"(t));
      else asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(t) : "v"(f4));
    }
  }
  __builtin_amdgcn_sched_barrier(0);
}

// NACC accumulators used round robin: NACC = 8 independent streams, 2 = the emulated kernel's dependency distance
template <int KIND, int NV, int NACC>
__global__ __launch_bounds__(512) void kb(float* out, unsigned long long* cyc, int iters) {
  __shared__ float lds[4096];
  lds[threadIdx.x] = threadIdx.x;
  __syncthreads();
  f32x16 c[8];
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) c[i][r] = 0.f;
  float a = out[threadIdx.x & 63], b = out[(threadIdx.x & 63) + 64];
  bf16x8 av, bv;
  for (int i = 0; i < 8; ++i) { av[i] = (__bf16)(a + i); bv[i] = (__bf16)(b - i); }
  float f0 = a, f1 = b, f2 = a + 1.f, f3 = b + 1.f, f4 = 0.999f;
  f32x4 l0 = {0.f, 0.f, 0.f, 0.f};
  f32x2 p0 = {a, b}, p1 = {b, a}, p2 = {0.999f, 0.998f};
  unsigned la = (threadIdx.x & 63) * 16, u0 = 0;
  const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      c[i % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, c[i % NACC], 0, 0, 0);
      fill<KIND, NV>(f0, f1, f2, f3, f4, l0, la, p0, p1, p2, u0);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n s_nop 7\n s_nop 7" ::: "memory");
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = f0 + f1 + f2 + f3 + l0[0] + p0[0] + p0[1] + p1[0] + p1[1] + (float)u0;
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += c[i][r];
  if (s == 12345.678f) out[threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

// roles: mode 0 = waves 0-3 MFMA only (waves 4-7 idle), 1 = waves 4-7 vector only (0-3 idle), 2 = both.  Per iteration
// 8 MFMAs (256 pipe cycles) resp. 8 x NV vector instructions.
template <int KIND, int NV>
__global__ __launch_bounds__(512) void kroles(float* out, unsigned long long* cyc, int iters, int mode) {
  __shared__ float lds[4096];
  lds[threadIdx.x] = threadIdx.x;
  __syncthreads();
  const int w = threadIdx.x >> 6;
  const bool mf = w < 4;
  f32x16 c[8];
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) c[i][r] = 0.f;
  float a = out[threadIdx.x & 63], b = out[(threadIdx.x & 63) + 64];
  bf16x8 av, bv;
  for (int i = 0; i < 8; ++i) { av[i] = (__bf16)(a + i); bv[i] = (__bf16)(b - i); }
  float f0 = a, f1 = b, f2 = a + 1.f, f3 = b + 1.f, f4 = 0.999f;
  f32x4 l0 = {0.f, 0.f, 0.f, 0.f};
  f32x2 p0 = {a, b}, p1 = {b, a}, p2 = {0.999f, 0.998f};
  unsigned la = (threadIdx.x & 63) * 16, u0 = 0;
  unsigned long long t0 = 0, t1 = 0;
  if (mf && mode != 1) {
    t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, c[i], 0, 0, 0);
    }
    asm volatile("s_nop 7\n s_nop 7" ::: "memory");
    t1 = __builtin_readcyclecounter();
  } else if (!mf && mode != 0) {
    t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) fill<KIND, NV>(f0, f1, f2, f3, f4, l0, la, p0, p1, p2, u0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n s_nop 7" ::: "memory");
    t1 = __builtin_readcyclecounter();
  }
  float s = f0 + f1 + f2 + f3 + l0[0] + p0[0] + p0[1] + p1[0] + p1[1] + (float)u0;
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += c[i][r];
  if (s == 12345.678f) out[threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + w] = t1 - t0;
}

typedef void (*kern_t)(float*, unsigned long long*, int);
typedef void (*kern_r)(float*, unsigned long long*, int, int);

int main() {
  float* out;
  unsigned long long* cyc;
  CK(hipMalloc(&out, 4096 * 4));
  CK(hipMemset(out, 0, 4096 * 4));
  CK(hipMalloc(&cyc, 256 * 16 * 8));
  const int iters = 2000;
  struct V { const char* name; kern_t k; int fill; } vs[] = {
      {"bare, 8 accumulators", kb<K_FMA, 0, 8>, 0}, {"bare, 2 accumulators (dependency distance 2)", kb<K_FMA, 0, 2>, 0},
      {"bare, 1 accumulator (back-to-back dependent)", kb<K_FMA, 0, 1>, 0},
      {"+ 1 v_fma", kb<K_FMA, 1, 8>, 1}, {"+ 2 v_fma", kb<K_FMA, 2, 8>, 2}, {"+ 3 v_fma", kb<K_FMA, 3, 8>, 3}, {"+ 4 v_fma", kb<K_FMA, 4, 8>, 4},
      {"+ 5 v_fma", kb<K_FMA, 5, 8>, 5}, {"+ 6 v_fma", kb<K_FMA, 6, 8>, 6}, {"+ 8 v_fma", kb<K_FMA, 8, 8>, 8},
      {"+ 2 v_fma, 2 accumulators", kb<K_FMA, 2, 2>, 2}, {"+ 4 v_fma, 2 accumulators", kb<K_FMA, 4, 2>, 4},
      {"+ 1 v_pk_fma", kb<K_PKFMA, 1, 8>, 1}, {"+ 2 v_pk_fma", kb<K_PKFMA, 2, 8>, 2}, {"+ 4 v_pk_fma", kb<K_PKFMA, 4, 8>, 4},
      {"+ 1 v_exp", kb<K_EXP, 1, 8>, 1}, {"+ 2 v_exp", kb<K_EXP, 2, 8>, 2}, {"+ 4 v_exp", kb<K_EXP, 4, 8>, 4},
      {"+ 2 v_cvt_pk_bf16_f32", kb<K_CVT, 2, 8>, 2},
      {"+ 2 of the activation mix", kb<K_MIX, 2, 8>, 2}, {"+ 3 of the activation mix", kb<K_MIX, 3, 8>, 3}, {"+ 4 of the activation mix", kb<K_MIX, 4, 8>, 4},
      {"+ 1 ds_read_b128 (no wait)", kb<K_LDS, 1, 8>, 1}, {"+ 2 ds_read_b128 (no wait)", kb<K_LDS, 2, 8>, 2}};
  printf("| same wave: v_mfma_f32_32x32x16_bf16 ... | fillers / MFMA | cycles / MFMA, 1 wave per SIMD | cycles / MFMA per wave, 2 waves per SIMD | pipe interval seen with 2 waves |\n|---|---|---|---|---|\n");
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
  struct R { const char* name; kern_r k; int nv; } rs[] = {
      {"4 v_fma per slot (32 / iteration)", kroles<K_FMA, 4>, 4}, {"8 v_fma per slot (64 / iteration)", kroles<K_FMA, 8>, 8},
      {"4 v_pk_fma per slot", kroles<K_PKFMA, 4>, 4}, {"4 of the activation mix per slot", kroles<K_MIX, 4>, 4},
      {"8 of the activation mix per slot", kroles<K_MIX, 8>, 8}, {"2 ds_read_b128 per slot", kroles<K_LDS, 2>, 2}};
  printf("\n| roles (waves 0-3: 8 MFMAs / iteration; waves 4-7: vector only) | cycles / iteration: MFMA waves alone | vector waves alone | MFMA waves, both running | vector waves, both running |\n|---|---|---|---|---|\n");
  for (auto& r : rs) {
    double res[3][2];
    for (int mode = 0; mode < 3; ++mode) {
      hipLaunchKernelGGL(r.k, dim3(256), dim3(512), 0, 0, out, cyc, 10, mode);
      CK(hipDeviceSynchronize());
      hipLaunchKernelGGL(r.k, dim3(256), dim3(512), 0, 0, out, cyc, iters, mode);
      CK(hipDeviceSynchronize());
      std::vector<unsigned long long> h(256 * 8);
      CK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
      double s[2] = {0, 0};
      for (size_t i = 0; i < h.size(); ++i) s[(i & 7) >= 4] += (double)h[i];
      res[mode][0] = s[0] / (h.size() / 2) / iters; res[mode][1] = s[1] / (h.size() / 2) / iters;
    }
    printf("| %s | %.0f | %.0f | %.0f | %.0f |\n", r.name, res[0][0], res[1][1], res[2][0], res[2][1]);
  }
  return 0;
}
