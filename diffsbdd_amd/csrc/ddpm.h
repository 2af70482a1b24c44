// Per-step DDPM posterior updates and the sharding-invariant noise generator.
//   ConditionalDDPM.sample_p_zs_given_zt      conditional_model.py:448-464
//   ConditionalDDPM.sample_normal_zero_com    conditional_model.py:140-160
//   EnVariationalDiffusion.sample_p_zs_given_zt  en_diffusion.py:530-556
// One workgroup per sample (a sample is <= a few hundred nodes); the ~20 tiny
// PyTorch kernels + 2-4 host syncs of the reference step become one launch.
#pragma once
#include "common.h"
#include "graph.h"

namespace dsbdd {

__device__ __forceinline__ float block_sum(float v, float* red) {
  const int t = threadIdx.x;
  red[t] = v;
  __syncthreads();
  for (int o = kThreads / 2; o > 0; o >>= 1) {
    if (t < o) red[t] += red[t + o];
    __syncthreads();
  }
  const float r = red[0];
  __syncthreads();
  return r;
}

// z_lig <- z_lig / alpha_ts - c_eps * eps + sigma * noise ; ligand COM removed
// from ligand x and pocket x (conditional_model.py:688-696).
__global__ __launch_bounds__(kThreads) void cond_update_kernel(
    float* z_lig, float* xh_poc, const float* eps, const float* noise, const int64_t* mask_lig,
    int n_lig, const int64_t* mask_poc, int n_poc, int dl, int dp, float alpha_ts, float c_eps,
    float sigma, int remove_com) {
  __shared__ float red[kThreads];
  const int b = blockIdx.x, t = threadIdx.x;
  const int l0 = lower_bound_i64(mask_lig, n_lig, b), l1 = lower_bound_i64(mask_lig, n_lig, b + 1);
  const int p0 = lower_bound_i64(mask_poc, n_poc, b), p1 = lower_bound_i64(mask_poc, n_poc, b + 1);
  const int nl = l1 - l0;
  float s[3] = {0.f, 0.f, 0.f};
  for (int idx = t; idx < nl * dl; idx += kThreads) {
    const int i = l0 + idx / dl, c = idx % dl;
    const size_t o = (size_t)i * dl + c;
    // mu = zt / alpha_{t|s} - (sigma^2_{t|s} / alpha_{t|s} / sigma_t) * eps ; zs = mu + sigma * noise
    const float v = (z_lig[o] / alpha_ts - c_eps * eps[o]) + sigma * noise[o];
    z_lig[o] = v;
    if (c < 3) s[c] += v;
  }
  if (!remove_com) return;
  const float cnt = nl > 0 ? (float)nl : 1.f;
  float m[3];
  for (int c = 0; c < 3; ++c) m[c] = block_sum(s[c], red) / cnt;
  for (int idx = t; idx < nl * 3; idx += kThreads) z_lig[(size_t)(l0 + idx / 3) * dl + idx % 3] -= m[idx % 3];
  for (int idx = t; idx < (p1 - p0) * 3; idx += kThreads)
    xh_poc[(size_t)(p0 + idx / 3) * dp + idx % 3] -= m[idx % 3];
}

// Joint model: both node sets are denoised, then the COM over ligand+pocket is
// removed (en_diffusion.py:547-556).
__global__ __launch_bounds__(kThreads) void joint_update_kernel(
    float* z_lig, float* z_poc, const float* eps_l, const float* eps_p, const float* noise_l,
    const float* noise_p, const int64_t* mask_lig, int n_lig, const int64_t* mask_poc, int n_poc,
    int dl, int dp, float alpha_ts, float c_eps, float sigma) {
  __shared__ float red[kThreads];
  const int b = blockIdx.x, t = threadIdx.x;
  const int l0 = lower_bound_i64(mask_lig, n_lig, b), l1 = lower_bound_i64(mask_lig, n_lig, b + 1);
  const int p0 = lower_bound_i64(mask_poc, n_poc, b), p1 = lower_bound_i64(mask_poc, n_poc, b + 1);
  const int nl = l1 - l0, np = p1 - p0;
  float s[3] = {0.f, 0.f, 0.f};
  for (int idx = t; idx < nl * dl; idx += kThreads) {
    const int c = idx % dl;
    const size_t o = (size_t)(l0 + idx / dl) * dl + c;
    const float v = (z_lig[o] / alpha_ts - c_eps * eps_l[o]) + sigma * noise_l[o];
    z_lig[o] = v;
    if (c < 3) s[c] += v;
  }
  for (int idx = t; idx < np * dp; idx += kThreads) {
    const int c = idx % dp;
    const size_t o = (size_t)(p0 + idx / dp) * dp + c;
    const float v = (z_poc[o] / alpha_ts - c_eps * eps_p[o]) + sigma * noise_p[o];
    z_poc[o] = v;
    if (c < 3) s[c] += v;
  }
  const float cnt = (nl + np) > 0 ? (float)(nl + np) : 1.f;
  float m[3];
  for (int c = 0; c < 3; ++c) m[c] = block_sum(s[c], red) / cnt;
  for (int idx = t; idx < nl * 3; idx += kThreads) z_lig[(size_t)(l0 + idx / 3) * dl + idx % 3] -= m[idx % 3];
  for (int idx = t; idx < np * 3; idx += kThreads) z_poc[(size_t)(p0 + idx / 3) * dp + idx % 3] -= m[idx % 3];
}

// ---- Philox4x32-10 (Salmon et al. 2011) + Box-Muller -----------------------
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
  const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
  const uint32_t n1 = (uint32_t)p1;
  const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
  const uint32_t n3 = (uint32_t)p0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}

__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}

// out[i][c] ~ N(0,1), a pure function of (seed, draw_index, stream_id, global
// sample id, row within sample, column).
__global__ void randn_keyed_kernel(float* out, const int64_t* mask, int n_rows, int n_cols,
                                   int64_t sample_offset, uint64_t seed, uint64_t draw_index,
                                   uint32_t stream_id) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_rows * n_cols) return;
  const int i = idx / n_cols, c = idx % n_cols;
  const int64_t b = mask[i];
  const int first = lower_bound_i64(mask, n_rows, b);
  const uint64_t gs = (uint64_t)(b + sample_offset);
  uint32_t ctr[4] = {(uint32_t)gs, (uint32_t)((i - first) * n_cols + c), (uint32_t)draw_index,
                     (uint32_t)(draw_index >> 32) ^ (stream_id * 0x9E3779B1u) ^ (uint32_t)(gs >> 32)};
  philox4x32_10(ctr, (uint32_t)seed, (uint32_t)(seed >> 32));
  // Box-Muller on two 32-bit uniforms; u1 in (0,1]
  const float u1 = ((float)(ctr[0] >> 8) + 1.0f) * (1.0f / 16777216.0f);
  const float u2 = (float)(ctr[1] >> 8) * (1.0f / 16777216.0f);
  out[idx] = sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
}

}  // namespace dsbdd
