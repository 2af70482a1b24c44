// Common device helpers for the gfx950 (CDNA4, wave64) kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dsbdd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int kThreads = 256;  // 4 waves of 64 lanes per workgroup

// SiLU(x) = x * sigmoid(x)  (nn.SiLU; egnn_new.py:8,16-19) as v_exp_f32 +
// v_rcp_f32 (~1 ulp each; `x / (1+e)` would expand to the 10-instruction IEEE
// division sequence).  Saturates correctly: x << 0 -> exp = inf -> rcp = 0 -> -0.
__device__ __forceinline__ float sigmoidf_fast(float x) {
  return __builtin_amdgcn_rcpf(1.0f + __expf(-x));
}

__device__ __forceinline__ float silu(float x) { return x * sigmoidf_fast(x); }

// Two values per lane and instruction (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 on 64-bit register pairs; measured
// next to fp32 MFMAs, tools/mfma_shadow.hip: 4.0 cycles per packed instruction against 2.9 per scalar one, i.e. 0.7 x
// the cost per value).  silu2 is the same five operations per value as silu(), so the results are bitwise equal.
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2 splat2(float v) { return f32x2{v, v}; }
__device__ __forceinline__ f32x2 silu2(f32x2 x) {
  const f32x2 s = x * splat2(-1.44269504088896340736f);               // exp(-x) = exp2(-x log2 e), as __expf
  const f32x2 e = {__builtin_amdgcn_exp2f(s.x), __builtin_amdgcn_exp2f(s.y)};
  const f32x2 d = e + splat2(1.0f);
  const f32x2 r = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
  return x * r;
}

// v_mfma_f32_32x32x2_f32: D[32x32] += A[32x2] * B[2x32], exact fp32 (one fmaf
// chain per output).  Lane l supplies A[i = l&31][k = l>>5] and
// B[k = l>>5][j = l&31]; D register r of lane l is
// D[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31].
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ int mfma_row(int r, int lane) {
  return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
}

// Bijective XCD-aware remap of a linear work index: workgroup b runs on XCD
// b % 8 (observed dispatch order; speed only), so give every XCD a contiguous
// chunk of the index space -> neighbouring tiles share that XCD's L2.
__device__ __forceinline__ int xcd_remap(int b, int n) {
  constexpr int X = 8;
  int q = n / X, r = n % X;
  int xcd = b % X, k = b / X;
  int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + k;
}

__device__ __forceinline__ float4 ld4(const float* p) {
  return *reinterpret_cast<const float4*>(p);
}

__device__ __forceinline__ f32x4 ldv4(const float* p) {
  return *reinterpret_cast<const f32x4*>(p);
}

}  // namespace dsbdd
