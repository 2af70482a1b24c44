// Probe: why do the two waves of a SIMD not overlap in the emulated edge kernel?  A synthetic persistent loop with the
// kernel's phase structure per "k step" -- V: ~100 vector instructions (incl. 16 transcendentals), M: 48 bf16 MFMAs -- on
// 256-thread workgroups, ONE or TWO per CU (dynamic LDS sized so that only that many fit), with the ingredients switched
// on one at a time:  barrier per step, B operands read from LDS in front of their MFMAs, global loads feeding the vector
// phase, LDS staging writes, packed fp32 in the vector phase.  Prints shader-clock cycles per step per workgroup.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/emu_phase_probe.hip -o tools/bin/emu_phase_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

enum { F_BARRIER = 1, F_LDSB = 2, F_GLOAD = 4, F_STAGE = 8, F_PACKED = 16, F_NOV = 32, F_NOM = 64,
       F_NOTRANS = 128, F_NOSPLIT = 256, F_NOARITH = 512, F_PRIO_M = 1024, F_PRIO_V = 2048, F_AGPR = 4096 };

template <int FLAGS, int DIST = 1>
__global__ __launch_bounds__(256, 2) void probe(const float* gsrc, float* out, unsigned long long* cyc, int steps) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int t = threadIdx.x, lane = t & 63;
  for (int i = t; i < 12288; i += 256) lds[i] = (float)(i & 255) * 0.001f;     // 48 KB of "B slices"
  __syncthreads();
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = gsrc[(t * 8 + i) & 4095];
  bf16x8 a_h, a_m, a_l;
  for (int i = 0; i < 8; ++i) { a_h[i] = (__bf16)v[i]; a_m[i] = (__bf16)(v[i] * 0.01f); a_l[i] = (__bf16)(v[i] * 0.0001f); }
  const float* gp = gsrc + (size_t)(blockIdx.x * 256 + t) * 16 % 65536;
  f32x4 g0 = {0, 0, 0, 0}, g1 = g0, g2 = g0, g3 = g0;
  const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int s = 0; s < steps; ++s) {
    // ---- V phase: the activation arithmetic of 8 values (add, 2 fma, add, mul, exp, add, rcp, mul) + the bf16 split
    if (!(FLAGS & F_NOV)) {
      if (FLAGS & F_PRIO_V) __builtin_amdgcn_s_setprio(2);
      if (FLAGS & F_GLOAD) {                      // the P / Q chunk of this step (requested one step earlier)
        v[0] += g0[0]; v[1] += g0[1]; v[2] += g1[0]; v[3] += g1[1]; v[4] += g2[0]; v[5] += g2[1]; v[6] += g3[0]; v[7] += g3[1];
        const float* q = gp + ((s * 64) & 16383);
        g0 = *reinterpret_cast<const f32x4*>(q); g1 = *reinterpret_cast<const f32x4*>(q + 4);
        g2 = *reinterpret_cast<const f32x4*>(q + 8); g3 = *reinterpret_cast<const f32x4*>(q + 12);
      }
      if (FLAGS & F_PACKED) {
#pragma unroll
        for (int i = 0; i < 8; i += 2) {
          f32x2 z = {v[i], v[i + 1]};
          z = z + f32x2{0.1f, 0.2f};
          z = __builtin_elementwise_fma(z, f32x2{0.9f, 0.9f}, f32x2{0.01f, 0.01f});
          z = __builtin_elementwise_fma(z, f32x2{1.1f, 1.1f}, f32x2{0.02f, 0.02f});
          z = z + f32x2{0.3f, 0.3f};
          f32x2 e = z * f32x2{-1.44f, -1.44f};
          e = f32x2{__builtin_amdgcn_exp2f(e.x), __builtin_amdgcn_exp2f(e.y)} + f32x2{1.f, 1.f};
          z = z * f32x2{__builtin_amdgcn_rcpf(e.x), __builtin_amdgcn_rcpf(e.y)};
          v[i] = z.x; v[i + 1] = z.y;
        }
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          float z = v[i];
          if (!(FLAGS & F_NOARITH)) {
            z = z + 0.1f;
            z = __builtin_fmaf(z, 0.9f, 0.01f);
            z = __builtin_fmaf(z, 1.1f, 0.02f);
            z = z + 0.3f;
          }
          if (!(FLAGS & F_NOTRANS)) {
            float e = __builtin_amdgcn_exp2f(z * -1.44f) + 1.f;
            z = z * __builtin_amdgcn_rcpf(e);
          }
          v[i] = z;
        }
      }
      if (!(FLAGS & F_NOSPLIT)) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const __bf16 h1 = (__bf16)v[i];
          const float r1 = v[i] - (float)h1;
          const __bf16 m1 = (__bf16)r1;
          const float r2 = r1 - (float)m1;
          a_h[i] = h1; a_m[i] = m1; a_l[i] = (__bf16)r2;
        }
      } else {
        a_h[0] = (__bf16)v[0]; a_m[1] = (__bf16)v[1]; a_l[2] = (__bf16)(v[2] + v[3] + v[4] + v[5] + v[6] + v[7]);
      }
      if (FLAGS & F_PRIO_V) __builtin_amdgcn_s_setprio(0);
    }
    // ---- M phase: 8 column tiles x 6 products
    if (!(FLAGS & F_NOM)) {
      if (FLAGS & F_PRIO_M) __builtin_amdgcn_s_setprio(2);
      const float* bl = lds + ((s & 1) ? 6144 : 0) + lane * 4;
#pragma unroll
      for (int c0 = 0; c0 < 8; c0 += DIST) {               // DIST column tiles in flight: the same accumulator every DIST-th MFMA
        bf16x8 bh[DIST], bm[DIST], bo[DIST];
#pragma unroll
        for (int u = 0; u < DIST; ++u) {
          bh[u] = a_h; bm[u] = a_m; bo[u] = a_l;
          if (FLAGS & F_LDSB) {
            bh[u] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(bl + (c0 + u) * 768));
            bm[u] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(bl + (c0 + u) * 768 + 256));
            bo[u] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(bl + (c0 + u) * 768 + 512));
          }
        }
#define MM(a, b) _Pragma("unroll") for (int u = 0; u < DIST; ++u) { \
          if (FLAGS & F_AGPR) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[c0 + u]) : "v"(a), "v"(b[u])); \
          else acc[c0 + u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b[u], acc[c0 + u], 0, 0, 0); }
        MM(a_l, bh); MM(a_m, bm); MM(a_h, bo); MM(a_m, bh); MM(a_h, bm); MM(a_h, bh);
#undef MM
      }
      if (FLAGS & F_PRIO_M) __builtin_amdgcn_s_setprio(0);
    }
    if (FLAGS & F_STAGE) {                        // the next slice: 6 x 16 bytes per thread from L2 into the other LDS buffer
      const float* q = gsrc + ((s * 6144 + t * 4) & 65535);
      float* d = lds + ((s & 1) ? 0 : 6144) + t * 4;
#pragma unroll
      for (int i = 0; i < 6; ++i) *reinterpret_cast<f32x4*>(d + 1024 * i) = *reinterpret_cast<const f32x4*>(q + 1024 * i);
    }
    if (FLAGS & F_BARRIER) __syncthreads();
  }
  if (FLAGS & F_AGPR) asm volatile("s_nop 15\n s_nop 15\n s_nop 15" ::: "memory");
  const unsigned long long t1 = __builtin_readcyclecounter();
  float sum = 0.f;
  for (int i = 0; i < 8; ++i) { sum += v[i]; for (int r = 0; r < 16; ++r) sum += acc[i][r]; }
  sum += (float)a_h[0] + g0[0];
  if (sum == 12345.678f) out[t] = sum;
  if (t == 0) cyc[blockIdx.x] = t1 - t0;
}

typedef void (*kern_t)(const float*, float*, unsigned long long*, int);

int main() {
  float *src, *out;
  unsigned long long* cyc;
  std::vector<float> h(65536 + 4096);
  for (size_t i = 0; i < h.size(); ++i) h[i] = 0.001f * (float)(i % 997);
  CK(hipMalloc(&src, h.size() * 4));
  CK(hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&out, 4096 * 4));
  CK(hipMalloc(&cyc, 1024 * 8));
  const int steps = 512;
  struct V { const char* name; kern_t k; } vs[] = {
      {"M only (48 MFMAs per step)", probe<F_NOV, 2>},
      {"V only: arith + trans + split", probe<F_NOM, 2>},
      {"V only: arith", probe<F_NOM | F_NOTRANS | F_NOSPLIT, 2>},
      {"V only: trans", probe<F_NOM | F_NOARITH | F_NOSPLIT, 2>},
      {"V only: split", probe<F_NOM | F_NOARITH | F_NOTRANS, 2>},
      {"V + M", probe<0, 2>},
      {"V(arith) + M", probe<F_NOTRANS | F_NOSPLIT, 2>},
      {"V(trans) + M", probe<F_NOARITH | F_NOSPLIT, 2>},
      {"V(split) + M", probe<F_NOARITH | F_NOTRANS, 2>},
      {"V + M, M phase at priority 2", probe<F_PRIO_M, 2>},
      {"V + M, V phase at priority 2", probe<F_PRIO_V, 2>},
      {"kernel's step", probe<F_LDSB | F_GLOAD | F_STAGE | F_BARRIER, 2>},
      {"kernel's step, M phase at priority 2", probe<F_LDSB | F_GLOAD | F_STAGE | F_BARRIER | F_PRIO_M, 2>},
      {"kernel's step, V phase at priority 2", probe<F_LDSB | F_GLOAD | F_STAGE | F_BARRIER | F_PRIO_V, 2>},
      {"kernel's step without V", probe<F_NOV | F_LDSB | F_GLOAD | F_STAGE | F_BARRIER, 2>},
      {"AGPR accumulators: M only", probe<F_AGPR | F_NOV, 2>},
      {"AGPR accumulators: V + M", probe<F_AGPR, 2>},
      {"AGPR accumulators: V(arith) + M", probe<F_AGPR | F_NOTRANS | F_NOSPLIT, 2>},
      {"AGPR accumulators: V(trans) + M", probe<F_AGPR | F_NOARITH | F_NOSPLIT, 2>},
      {"AGPR accumulators: kernel's step", probe<F_AGPR | F_LDSB | F_GLOAD | F_STAGE | F_BARRIER, 2>},
      {"AGPR accumulators: kernel's step without V", probe<F_AGPR | F_NOV | F_LDSB | F_GLOAD | F_STAGE | F_BARRIER, 2>},
      {"M + B from LDS", probe<F_NOV | F_LDSB, 2>},
      {"M + staging + barrier", probe<F_NOV | F_STAGE | F_BARRIER, 2>}};
  printf("| per k step (shader cycles, mean over workgroups) | 1 workgroup per CU | 2 workgroups per CU | per-CU throughput gain of the second |\n|---|---|---|---|\n");
  for (auto& v : vs) {
    double res[2];
    for (int w = 0; w < 2; ++w) {
      const int grid = w == 0 ? 256 : 512;
      // 49 KB of LDS used; 2 per CU: request 64 KB each, 1 per CU: 96 KB each
      const size_t lds_bytes = w == 0 ? 96 * 1024 : 64 * 1024;
      CK(hipFuncSetAttribute(reinterpret_cast<const void*>(v.k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
      hipLaunchKernelGGL(v.k, dim3(grid), dim3(256), lds_bytes, 0, (const float*)src, out, cyc, 8);
      CK(hipDeviceSynchronize());
      hipLaunchKernelGGL(v.k, dim3(grid), dim3(256), lds_bytes, 0, (const float*)src, out, cyc, steps);
      CK(hipDeviceSynchronize());
      std::vector<unsigned long long> hc(grid);
      CK(hipMemcpy(hc.data(), cyc, grid * 8, hipMemcpyDeviceToHost));
      double s = 0;
      for (auto c : hc) s += (double)c;
      res[w] = s / grid / steps;
    }
    printf("| %s | %.0f | %.0f | %.2f x |\n", v.name, res[0], res[1], 2.0 * res[0] / res[1]);
  }
  return 0;
}
