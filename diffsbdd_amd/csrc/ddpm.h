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

// One standard normal: a pure function of (seed, draw_index, stream_id, global sample id, element index inside the sample's
// [rows][n_cols] block).  randn_keyed_kernel writes these values to memory; the fused step kernel (cond_step_keyed_kernel)
// evaluates the SAME function in place of the load -- the same bits either way.
__device__ __forceinline__ float randn_keyed_value(uint64_t seed, uint64_t draw_index, uint32_t stream_id, uint64_t gs,
                                                   uint32_t elem) {
  uint32_t ctr[4] = {(uint32_t)gs, elem, (uint32_t)draw_index,
                     (uint32_t)(draw_index >> 32) ^ (stream_id * 0x9E3779B1u) ^ (uint32_t)(gs >> 32)};
  philox4x32_10(ctr, (uint32_t)seed, (uint32_t)(seed >> 32));
  // Box-Muller on two 32-bit uniforms; u1 in (0,1]
  const float u1 = ((float)(ctr[0] >> 8) + 1.0f) * (1.0f / 16777216.0f);
  const float u2 = (float)(ctr[1] >> 8) * (1.0f / 16777216.0f);
  return sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
}

// noise sources of the per-sample kernels: a tensor [n_lig][dl] (injected noise, separate randn launch) or the keyed
// generator evaluated in place (o = flat index into the tensor, l0 = first ligand row of the sample)
struct NoiseTensor {
  const float* p;
  __device__ __forceinline__ float operator()(size_t o, int, int) const { return p[o]; }
};
struct NoiseKeyed {
  uint64_t seed, draw; uint64_t gs;
  __device__ __forceinline__ float operator()(size_t, int row_in_sample, int elem_in_sample) const {
    (void)row_in_sample;
    return randn_keyed_value(seed, draw, 0u, gs, (uint32_t)elem_in_sample);
  }
};

// z_lig <- z_lig / alpha_ts - c_eps * eps + sigma * noise ; ligand COM removed
// from ligand x and pocket x (conditional_model.py:688-696).
template <class Noise>
__device__ __forceinline__ void cond_update_body(float* z_lig, float* xh_poc, const float* eps, const Noise& noise, int l0, int l1,
                                                 int p0, int p1, int dl, int dp, float alpha_ts, float c_eps, float sigma,
                                                 int remove_com, float* red) {
  const int t = threadIdx.x;
  const int nl = l1 - l0;
  float s[3] = {0.f, 0.f, 0.f};
  for (int idx = t; idx < nl * dl; idx += kThreads) {
    const int i = l0 + idx / dl, c = idx % dl;
    const size_t o = (size_t)i * dl + c;
    // mu = zt / alpha_{t|s} - (sigma^2_{t|s} / alpha_{t|s} / sigma_t) * eps ; zs = mu + sigma * noise
    // (explicit fma: the SAME two roundings whether the noise is a load or evaluated in place -- left to the compiler,
    //  the two instantiations contracted this expression differently: 1 ulp apart)
    const float v = __builtin_fmaf(sigma, noise(o, idx / dl, idx), __builtin_fmaf(-c_eps, eps[o], z_lig[o] / alpha_ts));
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

__global__ __launch_bounds__(kThreads) void cond_update_kernel(
    float* z_lig, float* xh_poc, const float* eps, const float* noise, const int64_t* mask_lig,
    int n_lig, const int64_t* mask_poc, int n_poc, int dl, int dp, float alpha_ts, float c_eps,
    float sigma, int remove_com) {
  __shared__ float red[kThreads];
  const int b = blockIdx.x;
  const int l0 = lower_bound_i64(mask_lig, n_lig, b), l1 = lower_bound_i64(mask_lig, n_lig, b + 1);
  const int p0 = lower_bound_i64(mask_poc, n_poc, b), p1 = lower_bound_i64(mask_poc, n_poc, b + 1);
  cond_update_body(z_lig, xh_poc, eps, NoiseTensor{noise}, l0, l1, p0, p1, dl, dp, alpha_ts, c_eps, sigma, remove_com, red);
}

// Joint model: both node sets are denoised, then the COM over ligand+pocket is
// removed (en_diffusion.py:547-556).
__global__ __launch_bounds__(kThreads) void joint_update_kernel(
    float* z_lig, float* z_poc, const float* eps_l, const float* eps_p, const float* noise_l,
    const float* noise_p, const int64_t* mask_lig, int n_lig, const int64_t* mask_poc, int n_poc,
    int dl, int dp, float alpha_ts, float c_eps, float sigma, int center_noise) {
  __shared__ float red[kThreads];
  const int b = blockIdx.x, t = threadIdx.x;
  const int l0 = lower_bound_i64(mask_lig, n_lig, b), l1 = lower_bound_i64(mask_lig, n_lig, b + 1);
  const int p0 = lower_bound_i64(mask_poc, n_poc, b), p1 = lower_bound_i64(mask_poc, n_poc, b + 1);
  const int nl = l1 - l0, np = p1 - p0;
  const float cnt = (nl + np) > 0 ? (float)(nl + np) : 1.f;
  float mn[3] = {0.f, 0.f, 0.f};
  if (center_noise) {   // x part of the noise COM-free over ligand + pocket (en_diffusion.py:932-942)
    float sl[3] = {0.f, 0.f, 0.f};
    for (int i = l0 + t; i < l1; i += kThreads)
      for (int c = 0; c < 3; ++c) sl[c] += noise_l[(size_t)i * dl + c];
    for (int i = p0 + t; i < p1; i += kThreads)
      for (int c = 0; c < 3; ++c) sl[c] += noise_p[(size_t)i * dp + c];
    for (int c = 0; c < 3; ++c) mn[c] = block_sum(sl[c], red) / cnt;
  }
  float s[3] = {0.f, 0.f, 0.f};
  for (int idx = t; idx < nl * dl; idx += kThreads) {
    const int c = idx % dl;
    const size_t o = (size_t)(l0 + idx / dl) * dl + c;
    const float nz = c < 3 ? noise_l[o] - mn[c] : noise_l[o];
    const float v = (z_lig[o] / alpha_ts - c_eps * eps_l[o]) + sigma * nz;
    z_lig[o] = v;
    if (c < 3) s[c] += v;
  }
  for (int idx = t; idx < np * dp; idx += kThreads) {
    const int c = idx % dp;
    const size_t o = (size_t)(p0 + idx / dp) * dp + c;
    const float nz = c < 3 ? noise_p[o] - mn[c] : noise_p[o];
    const float v = (z_poc[o] / alpha_ts - c_eps * eps_p[o]) + sigma * nz;
    z_poc[o] = v;
    if (c < 3) s[c] += v;
  }
  float m[3];
  for (int c = 0; c < 3; ++c) m[c] = block_sum(s[c], red) / cnt;
  for (int idx = t; idx < nl * 3; idx += kThreads) z_lig[(size_t)(l0 + idx / 3) * dl + idx % 3] -= m[idx % 3];
  for (int idx = t; idx < np * 3; idx += kThreads) z_poc[(size_t)(p0 + idx / 3) * dp + idx % 3] -= m[idx % 3];
}

// ---- building blocks of the sampling loops around the reverse step ------------------------
// All of them: one workgroup per sample, per-sample reductions by block_sum (a fixed reduction
// tree over the sample's own rows), so the result for a sample is a pure function of that
// sample's data -- bitwise reproducible and independent of the batch composition.  They replace
// chains of ~20-100 small torch kernels (index_add_ with float atomics among them).

struct SampleRows { int l0, l1, p0, p1; };
__device__ __forceinline__ SampleRows sample_rows(const int64_t* mask_lig, int n_lig, const int64_t* mask_poc,
                                                  int n_poc, int b) {
  return SampleRows{lower_bound_i64(mask_lig, n_lig, b), lower_bound_i64(mask_lig, n_lig, b + 1),
                    lower_bound_i64(mask_poc, n_poc, b), lower_bound_i64(mask_poc, n_poc, b + 1)};
}

// sum over rows [r0, r1) of columns 0..2 of a [rows][ld] matrix (optionally only rows with sel != 0)
__device__ __forceinline__ void rows_sum3(const float* a, int ld, int r0, int r1, const float* sel, float* red,
                                          float (&out)[3], float* count = nullptr) {
  float s[3] = {0.f, 0.f, 0.f}, cnt = 0.f;
  for (int i = r0 + (int)threadIdx.x; i < r1; i += kThreads) {
    if (sel && sel[i] == 0.f) continue;
    s[0] += a[(size_t)i * ld]; s[1] += a[(size_t)i * ld + 1]; s[2] += a[(size_t)i * ld + 2];
    cnt += 1.f;
  }
  for (int c = 0; c < 3; ++c) out[c] = block_sum(s[c], red);
  if (count) *count = block_sum(cnt, red);
}

// out[b][0..2] = mean over the rows of sample b of x[:, 0..2]  (scatter_mean, count clamped to >= 1)
__global__ __launch_bounds__(kThreads) void segment_mean3_kernel(const float* x, int ld, const int64_t* mask,
                                                                 int n_rows, float* out) {
  __shared__ float red[kThreads];
  const int b = blockIdx.x;
  const int r0 = lower_bound_i64(mask, n_rows, b), r1 = lower_bound_i64(mask, n_rows, b + 1);
  float s[3];
  rows_sum3(x, ld, r0, r1, nullptr, red, s);
  const float cnt = r1 > r0 ? (float)(r1 - r0) : 1.f;
  if (threadIdx.x < 3) out[3 * b + threadIdx.x] = s[threadIdx.x] / cnt;
}

// Conditional model:  z_lig <- a * z_lig + sigma * noise ; then (remove_com) the ligand centre of mass is
// subtracted from ligand and pocket x.  Covers sample_normal_zero_com (a = 1: z_lig holds mu),
// noised_representation (a = alpha_t), sample_p_zt_given_zs (a = alpha_t|s)
// (conditional_model.py:140-183,420-430).
__global__ __launch_bounds__(kThreads) void cond_affine_noise_kernel(
    float* z_lig, float* xh_poc, const float* noise, const int64_t* mask_lig, int n_lig,
    const int64_t* mask_poc, int n_poc, int dl, int dp, float a, float sigma, int remove_com) {
  __shared__ float red[kThreads];
  const int t = threadIdx.x;
  const SampleRows r = sample_rows(mask_lig, n_lig, mask_poc, n_poc, blockIdx.x);
  const int nl = r.l1 - r.l0;
  float s[3] = {0.f, 0.f, 0.f};
  for (int idx = t; idx < nl * dl; idx += kThreads) {
    const int c = idx % dl;
    const size_t o = (size_t)(r.l0 + idx / dl) * dl + c;
    const float v = a * z_lig[o] + sigma * noise[o];
    z_lig[o] = v;
    if (c < 3) s[c] += v;
  }
  if (!remove_com) return;
  const float cnt = nl > 0 ? (float)nl : 1.f;
  float m[3];
  for (int c = 0; c < 3; ++c) m[c] = block_sum(s[c], red) / cnt;
  for (int idx = t; idx < nl * 3; idx += kThreads) z_lig[(size_t)(r.l0 + idx / 3) * dl + idx % 3] -= m[idx % 3];
  for (int idx = t; idx < (r.p1 - r.p0) * 3; idx += kThreads)
    xh_poc[(size_t)(r.p0 + idx / 3) * dp + idx % 3] -= m[idx % 3];
}

// Joint model:  z <- a * z + sigma * noise for both node sets, where (center_noise) the x part of the
// noise is first made COM-free over the sample's ligand + pocket rows
// (sample_center_gravity_zero_gaussian_batch, en_diffusion.py:932-942) and (remove_com) the joint COM of
// the result is removed (en_diffusion.py:479-501).  a = 0 / sigma = 1 draws z_T.
__global__ __launch_bounds__(kThreads) void joint_affine_noise_kernel(
    float* z_lig, float* z_poc, const float* noise_l, const float* noise_p, const int64_t* mask_lig,
    int n_lig, const int64_t* mask_poc, int n_poc, int dl, int dp, float a, float sigma,
    int center_noise, int remove_com) {
  __shared__ float red[kThreads];
  const int t = threadIdx.x;
  const SampleRows r = sample_rows(mask_lig, n_lig, mask_poc, n_poc, blockIdx.x);
  const int nl = r.l1 - r.l0, np = r.p1 - r.p0;
  const float cnt = (nl + np) > 0 ? (float)(nl + np) : 1.f;
  float mn[3] = {0.f, 0.f, 0.f};
  if (center_noise) {
    float sl[3], sp[3];
    rows_sum3(noise_l, dl, r.l0, r.l1, nullptr, red, sl);
    rows_sum3(noise_p, dp, r.p0, r.p1, nullptr, red, sp);
    for (int c = 0; c < 3; ++c) mn[c] = (sl[c] + sp[c]) / cnt;
  }
  float s[3] = {0.f, 0.f, 0.f};
  for (int idx = t; idx < nl * dl; idx += kThreads) {
    const int c = idx % dl;
    const size_t o = (size_t)(r.l0 + idx / dl) * dl + c;
    const float nz = c < 3 ? noise_l[o] - mn[c] : noise_l[o];
    const float v = (a == 0.f ? 0.f : a * z_lig[o]) + sigma * nz;
    z_lig[o] = v;
    if (c < 3) s[c] += v;
  }
  for (int idx = t; idx < np * dp; idx += kThreads) {
    const int c = idx % dp;
    const size_t o = (size_t)(r.p0 + idx / dp) * dp + c;
    const float nz = c < 3 ? noise_p[o] - mn[c] : noise_p[o];
    const float v = (a == 0.f ? 0.f : a * z_poc[o]) + sigma * nz;
    z_poc[o] = v;
    if (c < 3) s[c] += v;
  }
  if (!remove_com) return;
  float m[3];
  for (int c = 0; c < 3; ++c) m[c] = block_sum(s[c], red) / cnt;
  for (int idx = t; idx < nl * 3; idx += kThreads) z_lig[(size_t)(r.l0 + idx / 3) * dl + idx % 3] -= m[idx % 3];
  for (int idx = t; idx < np * 3; idx += kThreads) z_poc[(size_t)(r.p0 + idx / 3) * dp + idx % 3] -= m[idx % 3];
}

// One RePaint iteration of the conditional model after the reverse step (conditional_model.py:600-660):
// on entry z_lig holds the denoised ("unknown") state z_s and xh_poc the pocket moved by that step.
//   known part : x_known = x0 + (COM(pocket) - com_pocket0);  z_known = alpha_s [x_known | h0] + sigma_s noise1,
//                ligand COM of z_known removed from z_known and the pocket;
//   alignment  : dx = COM_fixed(z_unknown) - COM_fixed(z_known);  z_known.x += dx;  pocket.x += dx;
//   blend      : z = z_known * fixed + z_unknown * (1 - fixed);
//   resample   : (renoise) z <- alpha_t|s z + sigma_t|s noise2, ligand COM removed from z and the pocket.
// zk_tmp [n_lig][dl] is scratch.
struct CondRepaintArgs {
  float* z_lig; float* xh_poc; float* zk_tmp;
  const float* xh0_lig;        // [n_lig][dl] the given ligand, normalised
  const float* com_pocket0;    // [B][3]
  const float* fixed;          // [n_lig] 1 = known atom
  const float* noise1; const float* noise2;
  const int64_t* mask_lig; const int64_t* mask_poc;
  int n_lig, n_poc, dl, dp;
  float alpha_s, sigma_s, alpha_ts, sigma_ts;
  int renoise, remove_com;
};

template <class Noise>
__device__ __forceinline__ void cond_repaint_body(const CondRepaintArgs& p, const SampleRows& r, int b, const Noise& noise1,
                                                  const Noise& noise2, float* red) {
  const int t = threadIdx.x, dl = p.dl, dp = p.dp;
  const int nl = r.l1 - r.l0, np = r.p1 - r.p0;
  float cp[3];
  rows_sum3(p.xh_poc, dp, r.p0, r.p1, nullptr, red, cp);
  const float npc = np > 0 ? (float)np : 1.f;
  float shift[3];
  for (int c = 0; c < 3; ++c) shift[c] = cp[c] / npc - p.com_pocket0[3 * b + c];
  // z_known before centring
  float s[3] = {0.f, 0.f, 0.f};
  for (int idx = t; idx < nl * dl; idx += kThreads) {
    const int c = idx % dl;
    const size_t o = (size_t)(r.l0 + idx / dl) * dl + c;
    const float base = c < 3 ? p.xh0_lig[o] + shift[c] : p.xh0_lig[o];
    const float v = __builtin_fmaf(p.sigma_s, noise1(o, idx / dl, idx), p.alpha_s * base);
    p.zk_tmp[o] = v;
    if (c < 3) s[c] += v;
  }
  float m1[3] = {0.f, 0.f, 0.f};
  if (p.remove_com) {
    const float cnt = nl > 0 ? (float)nl : 1.f;
    for (int c = 0; c < 3; ++c) m1[c] = block_sum(s[c], red) / cnt;
  } else {
    __syncthreads();
  }
  for (int idx = t; idx < nl * 3; idx += kThreads) p.zk_tmp[(size_t)(r.l0 + idx / 3) * dl + idx % 3] -= m1[idx % 3];
  __syncthreads();   // zk_tmp rows of this sample are re-read by other threads below
  // centres of mass of the fixed atoms of both parts
  float ck[3], cu[3], nf = 0.f;
  rows_sum3(p.zk_tmp, dl, r.l0, r.l1, p.fixed, red, ck, &nf);
  rows_sum3(p.z_lig, dl, r.l0, r.l1, p.fixed, red, cu);
  const float nfc = nf > 0.f ? nf : 1.f;
  float dx[3];
  for (int c = 0; c < 3; ++c) dx[c] = cu[c] / nfc - ck[c] / nfc;
  // blend (+ optional q(z_t | z_s))
  float s2[3] = {0.f, 0.f, 0.f};
  for (int idx = t; idx < nl * dl; idx += kThreads) {
    const int i = r.l0 + idx / dl, c = idx % dl;
    const size_t o = (size_t)i * dl + c;
    const float f = p.fixed[i];
    float zk = p.zk_tmp[o];
    if (c < 3) zk += dx[c];
    float v = zk * f + p.z_lig[o] * (1.f - f);
    if (p.renoise) {
      v = __builtin_fmaf(p.sigma_ts, noise2(o, idx / dl, idx), p.alpha_ts * v);
      if (c < 3) s2[c] += v;
    }
    p.z_lig[o] = v;
  }
  float m2[3] = {0.f, 0.f, 0.f};
  if (p.renoise && p.remove_com) {
    const float cnt = nl > 0 ? (float)nl : 1.f;
    for (int c = 0; c < 3; ++c) m2[c] = block_sum(s2[c], red) / cnt;
    for (int idx = t; idx < nl * 3; idx += kThreads) p.z_lig[(size_t)(r.l0 + idx / 3) * dl + idx % 3] -= m2[idx % 3];
  }
  // the pocket follows every translation of the ligand frame, in the reference's order
  for (int idx = t; idx < np * 3; idx += kThreads) {
    const int c = idx % 3;
    float* q = p.xh_poc + (size_t)(r.p0 + idx / 3) * dp + c;
    float v = (*q - m1[c]) + dx[c];
    if (p.renoise && p.remove_com) v -= m2[c];
    *q = v;
  }
}

__global__ __launch_bounds__(kThreads) void cond_repaint_kernel(CondRepaintArgs p) {
  __shared__ float red[kThreads];
  const SampleRows r = sample_rows(p.mask_lig, p.n_lig, p.mask_poc, p.n_poc, blockIdx.x);
  cond_repaint_body(p, r, blockIdx.x, NoiseTensor{p.noise1}, NoiseTensor{p.noise2}, red);
}

// ONE launch per reverse step of a pocket-conditioned chain with the keyed generator (round 5): the posterior update
// (cond_update_body, draw d), optionally the RePaint iteration behind it (cond_repaint_body, draws d + 1 and d + 2), the
// noise evaluated in place instead of 1 - 3 randn_keyed launches, and the NEXT denoiser call's time written to its device
// word (instead of a fill launch).  Same arithmetic as the separate kernels: the results are bitwise those.
struct CondStepArgs {
  CondRepaintArgs rp;            // (noise1 / noise2 unused; zk_tmp, xh0_lig, com_pocket0, fixed may be null without repaint)
  const float* eps;
  float u_alpha_ts, u_c_eps, u_sigma;   // the reverse step's coefficients
  int repaint;                   // 0: update only
  uint64_t seed, draw; int64_t sample_offset; const int64_t* sample_ids;
  float* t_word; float t_next;   // optional
};

__global__ __launch_bounds__(kThreads) void cond_step_keyed_kernel(CondStepArgs a) {
  __shared__ float red[kThreads];
  const int b = blockIdx.x;
  const CondRepaintArgs& p = a.rp;
  const SampleRows r = sample_rows(p.mask_lig, p.n_lig, p.mask_poc, p.n_poc, b);
  const uint64_t gs = (uint64_t)(a.sample_ids ? a.sample_ids[b] : b + a.sample_offset);
  cond_update_body(p.z_lig, p.xh_poc, a.eps, NoiseKeyed{a.seed, a.draw, gs}, r.l0, r.l1, r.p0, r.p1, p.dl, p.dp, a.u_alpha_ts,
                   a.u_c_eps, a.u_sigma, p.remove_com, red);
  if (a.repaint) {
    __threadfence_block();
    __syncthreads();             // the sample's rows of z_lig / xh_poc written above are read by other threads below
    cond_repaint_body(p, r, b, NoiseKeyed{a.seed, a.draw + 1, gs}, NoiseKeyed{a.seed, a.draw + 2, gs}, red);
  }
  if (a.t_word && b == 0 && threadIdx.x == 0) *a.t_word = a.t_next;
}

// One RePaint iteration of the joint model after the reverse step (en_diffusion.py:742-809):
// on entry z_* hold the denoised ("unknown") state.
//   known part : z_known = alpha_s xh0 + sigma_s noise1   (x part of noise1 made COM-free here)
//   alignment  : z_known.x += COM_known(z_unknown) - COM_known(z_known)   (known = fixed ligand + pocket nodes)
//   blend      : z = z_known * fixed + z_unknown * (1 - fixed)
//   jump back  : (renoise) z <- alpha_t|s z + sigma_t|s noise2 (COM-free), joint COM removed.
struct JointRepaintArgs {
  float* z_lig; float* z_poc; float* zk_lig; float* zk_poc;   // zk_*: scratch
  const float* xh0_lig; const float* xh0_poc;
  const float* fixed_l; const float* fixed_p;
  const float* n1_l; const float* n1_p; const float* n2_l; const float* n2_p;
  const int64_t* mask_lig; const int64_t* mask_poc;
  int n_lig, n_poc, dl, dp;
  float alpha_s, sigma_s, alpha_ts, sigma_ts;
  int renoise;
};

__global__ __launch_bounds__(kThreads) void joint_repaint_kernel(JointRepaintArgs p) {
  __shared__ float red[kThreads];
  const int t = threadIdx.x, dl = p.dl, dp = p.dp;
  const SampleRows r = sample_rows(p.mask_lig, p.n_lig, p.mask_poc, p.n_poc, blockIdx.x);
  const int nl = r.l1 - r.l0, np = r.p1 - r.p0;
  const float cnt = (nl + np) > 0 ? (float)(nl + np) : 1.f;
  float a3[3], b3[3], mn[3];
  rows_sum3(p.n1_l, dl, r.l0, r.l1, nullptr, red, a3);
  rows_sum3(p.n1_p, dp, r.p0, r.p1, nullptr, red, b3);
  for (int c = 0; c < 3; ++c) mn[c] = (a3[c] + b3[c]) / cnt;
  for (int idx = t; idx < nl * dl; idx += kThreads) {
    const int c = idx % dl;
    const size_t o = (size_t)(r.l0 + idx / dl) * dl + c;
    const float nz = c < 3 ? p.n1_l[o] - mn[c] : p.n1_l[o];
    p.zk_lig[o] = p.alpha_s * p.xh0_lig[o] + p.sigma_s * nz;
  }
  for (int idx = t; idx < np * dp; idx += kThreads) {
    const int c = idx % dp;
    const size_t o = (size_t)(r.p0 + idx / dp) * dp + c;
    const float nz = c < 3 ? p.n1_p[o] - mn[c] : p.n1_p[o];
    p.zk_poc[o] = p.alpha_s * p.xh0_poc[o] + p.sigma_s * nz;
  }
  __syncthreads();
  float kl[3], kp[3], ul[3], up[3], nfl = 0.f, nfp = 0.f;
  rows_sum3(p.zk_lig, dl, r.l0, r.l1, p.fixed_l, red, kl, &nfl);
  rows_sum3(p.zk_poc, dp, r.p0, r.p1, p.fixed_p, red, kp, &nfp);
  rows_sum3(p.z_lig, dl, r.l0, r.l1, p.fixed_l, red, ul);
  rows_sum3(p.z_poc, dp, r.p0, r.p1, p.fixed_p, red, up);
  const float nk = (nfl + nfp) > 0.f ? (nfl + nfp) : 1.f;
  float dx[3];
  for (int c = 0; c < 3; ++c) dx[c] = (ul[c] + up[c]) / nk - (kl[c] + kp[c]) / nk;
  float mn2[3] = {0.f, 0.f, 0.f};
  if (p.renoise) {
    rows_sum3(p.n2_l, dl, r.l0, r.l1, nullptr, red, a3);
    rows_sum3(p.n2_p, dp, r.p0, r.p1, nullptr, red, b3);
    for (int c = 0; c < 3; ++c) mn2[c] = (a3[c] + b3[c]) / cnt;
  }
  float s[3] = {0.f, 0.f, 0.f};
  for (int idx = t; idx < nl * dl; idx += kThreads) {
    const int i = r.l0 + idx / dl, c = idx % dl;
    const size_t o = (size_t)i * dl + c;
    const float f = p.fixed_l[i];
    float zk = p.zk_lig[o];
    if (c < 3) zk += dx[c];
    float v = zk * f + p.z_lig[o] * (1.f - f);
    if (p.renoise) {
      const float nz = c < 3 ? p.n2_l[o] - mn2[c] : p.n2_l[o];
      v = p.alpha_ts * v + p.sigma_ts * nz;
      if (c < 3) s[c] += v;
    }
    p.z_lig[o] = v;
  }
  for (int idx = t; idx < np * dp; idx += kThreads) {
    const int i = r.p0 + idx / dp, c = idx % dp;
    const size_t o = (size_t)i * dp + c;
    const float f = p.fixed_p[i];
    float zk = p.zk_poc[o];
    if (c < 3) zk += dx[c];
    float v = zk * f + p.z_poc[o] * (1.f - f);
    if (p.renoise) {
      const float nz = c < 3 ? p.n2_p[o] - mn2[c] : p.n2_p[o];
      v = p.alpha_ts * v + p.sigma_ts * nz;
      if (c < 3) s[c] += v;
    }
    p.z_poc[o] = v;
  }
  if (!p.renoise) return;
  float m[3];
  for (int c = 0; c < 3; ++c) m[c] = block_sum(s[c], red) / cnt;
  for (int idx = t; idx < nl * 3; idx += kThreads) p.z_lig[(size_t)(r.l0 + idx / 3) * dl + idx % 3] -= m[idx % 3];
  for (int idx = t; idx < np * 3; idx += kThreads) p.z_poc[(size_t)(r.p0 + idx / 3) * dp + idx % 3] -= m[idx % 3];
}

// out[i][c] ~ N(0,1), a pure function of (seed, draw_index, stream_id, global
// sample id, row within sample, column).
__global__ void randn_keyed_kernel(float* out, const int64_t* mask, int n_rows, int n_cols,
                                   int64_t sample_offset, const int64_t* sample_ids, uint64_t seed,
                                   uint64_t draw_index, uint32_t stream_id) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_rows * n_cols) return;
  const int i = idx / n_cols, c = idx % n_cols;
  const int64_t b = mask[i];
  const int first = lower_bound_i64(mask, n_rows, b);
  // global sample id: an explicit table (batches packed from several pockets) or offset + local id
  const uint64_t gs = (uint64_t)(sample_ids ? sample_ids[b] : b + sample_offset);
  out[idx] = randn_keyed_value(seed, draw_index, stream_id, gs, (uint32_t)((i - first) * n_cols + c));
}

}  // namespace dsbdd
