// The loss terms of the pocket-conditioned training step around the network call, as three launches (round 6).
//
// ConditionalDDPM.forward (conditional_model.py:202-330 of the reference; diffsbdd_amd/conditional_model.py mirrors it in
// torch) normalises the batch, removes the ligand's centre of mass, noises the ligand (z_t = alpha_t xh0 + sigma_t eps,
// centred again), evaluates the network, and returns twelve per-sample terms.  In torch that is ~ 250 launches of a few
// microseconds around a 10-ms step -- 0.8 ms of GPU time in launch gaps (profiles/r6l_*).  Here:
//
//   loss_cond_pre_kernel   one workgroup per sample: normalisation, both centrings, z_t, the centred pocket, and every
//                          term that does not read the network's output: SNR weight, -log Z of p(x | z0), KL of the prior,
//                          the categorical likelihood term L0_h of z_t (training mode evaluates it on z_t), log p(N_lig |
//                          N_pocket), delta log p(x);
//   loss_cond_post_kernel  the squared-error terms (error_t, L0_x), xh_hat, the two logged means;
//   loss_cond_post_bwd_kernel  their gradient w.r.t. the network's output.
//
// Every per-sample sum is a fixed-order reduction inside the sample's workgroup (torch: index_add_ with atomics).
// Formulas and their order follow the torch mirror line by line (cited below by the mirror's method names, which cite the
// reference); tests/test_gpu_train.py compares all twelve terms and the parameter gradients between the two paths.
#pragma once
#include "common.h"
#include "ddpm.h"

namespace dsbdd {

enum {
  LS_T = 0, LS_GAMMA_T, LS_GAMMA_S, LS_ALPHA_T, LS_SIGMA_T, LS_SNR_W, LS_NEG_LOG_C, LS_KL, LS_L0_H, LS_LOG_PN, LS_DELTA_LOG_PX,
  LS_T_IS_ZERO, LS_ROWS
};
enum { LO_ERR_T = 0, LO_L0_X, LO_INFO_X, LO_INFO_H, LO_ROWS };

struct LossCfg {
  int batch, n_lig, n_pocket, atom_nf, residue_nf, T;
  int remove_com;          // ConditionalDDPM: 1 (ligand COM removed from ligand and pocket); SimpleConditionalDDPM: 0
  int vnode_idx;           // class index of the virtual atom, or -1
  float nv0, nv1, nb1;     // norm_values[0], norm_values[1], norm_biases[1]
  int n1_tab, n2_tab;      // log p(n_lig | n_pocket) table [n1_tab][n2_tab]
};

constexpr int kLossThreads = 256;

__device__ __forceinline__ int lower_bound_i64(const long long* a, int n, long long v) {
  int lo = 0, hi = n;
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (a[mid] < v) lo = mid + 1; else hi = mid; }
  return lo;
}
// (block_sum of ddpm.h: fixed-order sum of one value per thread, every thread gets the total)
static_assert(kLossThreads == kThreads, "block_sum reduces kThreads values");
__device__ __forceinline__ float sigmoid_ref(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float cdf_gauss(float x) { return 0.5f * (1.0f + erff(x * 0.70710678118654752440f)); }
// PredefinedNoiseSchedule.forward: gamma[round(t * T)] with python's negative indexing (s = -1 / T at t = 0)
__device__ __forceinline__ float gamma_at(const float* table, int T, float t) {
  int i = (int)rintf(t * (float)T);
  if (i < 0) i += T + 1;
  if (i > T) i = T;
  return table[i];
}

// engine.edge_capacity in one launch: out[0] = the masks are sorted ascending with ids in [0, batch), out[1] = sum over the
// samples of the complete graph's edges with every (sample, node set) segment rounded up to 32 (csrc/graph.h).  One
// workgroup; ~ 25 torch launches (two bincounts, the comparisons, their reductions) before every training forward otherwise.
__global__ __launch_bounds__(kLossThreads) void edge_capacity_kernel(const long long* ml, int n_l, const long long* mp, int n_p,
                                                                      int batch, long long* out) {
  __shared__ long long cap_s[kLossThreads];
  __shared__ int ok_s[kLossThreads];
  const int t = threadIdx.x;
  int ok = 1;
  for (int i = t; i + 1 < n_l; i += kLossThreads) ok &= ml[i + 1] >= ml[i];
  for (int i = t; i + 1 < n_p; i += kLossThreads) ok &= mp[i + 1] >= mp[i];
  if (t == 0) {
    if (n_l > 0) ok &= ml[0] >= 0 && ml[n_l - 1] < batch;
    if (n_p > 0) ok &= mp[0] >= 0 && mp[n_p - 1] < batch;
  }
  long long cap = 0;
  for (int b = t; b < batch; b += kLossThreads) {
    const long long nl = lower_bound_i64(ml, n_l, b + 1) - lower_bound_i64(ml, n_l, b);
    const long long np = lower_bound_i64(mp, n_p, b + 1) - lower_bound_i64(mp, n_p, b);
    const long long n = nl + np;
    cap += (nl * n + 31) / 32 * 32 + (np * n + 31) / 32 * 32;
  }
  cap_s[t] = cap; ok_s[t] = ok;
  __syncthreads();
  for (int o = kLossThreads / 2; o > 0; o >>= 1) {
    if (t < o) { cap_s[t] += cap_s[t + o]; ok_s[t] &= ok_s[t + o]; }
    __syncthreads();
  }
  if (t == 0) { out[0] = ok_s[0]; out[1] = cap_s[0]; }
}

__global__ __launch_bounds__(kLossThreads) void loss_cond_pre_kernel(
    LossCfg c, const float* lig_x, const float* lig_h, const long long* lig_mask, const float* poc_x, const float* poc_h,
    const long long* poc_mask, const float* eps, const float* t_int, const float* gamma_table, const float* logpn_table,
    float* z_t, float* xh_pocket, float* ps, float* lig_xn, float* lig_hn, float* poc_xn, float* poc_hn) {
  __shared__ float red[kLossThreads];
  __shared__ int seg[4];
  const int b = blockIdx.x, t = threadIdx.x;
  const int a = c.atom_nf, r = c.residue_nf, ldl = 3 + a, ldp = 3 + r;
  if (t == 0) {
    seg[0] = lower_bound_i64(lig_mask, c.n_lig, b); seg[1] = lower_bound_i64(lig_mask, c.n_lig, b + 1);
    seg[2] = lower_bound_i64(poc_mask, c.n_pocket, b); seg[3] = lower_bound_i64(poc_mask, c.n_pocket, b + 1);
  }
  __syncthreads();
  const int l0 = seg[0], l1 = seg[1], p0 = seg[2], p1 = seg[3];
  const int nl = l1 - l0, np = p1 - p0;
  const float inv0 = 1.0f / c.nv0, inv1 = 1.0f / c.nv1;          // normalize(): a division by a python scalar multiplies by its inverse
  const float cnt = (float)(nl > 1 ? nl : 1);                     // seg_mean: count clamped to >= 1

  // per-sample scalars (forward(): t, s, gamma; alpha / sigma)
  const float ti = t_int[b];
  const float tt = ti / (float)c.T, ss = (ti - 1.0f) / (float)c.T;
  const float g_t = gamma_at(gamma_table, c.T, tt), g_s = gamma_at(gamma_table, c.T, ss);
  const float alpha_t = sqrtf(sigmoid_ref(-g_t)), sigma_t = sqrtf(sigmoid_ref(g_t));
  const float g_T = gamma_table[c.T], g_0 = gamma_table[0];
  const float alpha_T = sqrtf(sigmoid_ref(-g_T)), sigma_T = sqrtf(sigmoid_ref(g_T));

  // 1. ligand centre of mass of the normalised coordinates (_remove_lig_com)
  float m1[3] = {0.f, 0.f, 0.f};
  if (c.remove_com) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int i = l0 + t; i < l1; i += kLossThreads) { s0 += lig_x[3 * i] * inv0; s1 += lig_x[3 * i + 1] * inv0; s2 += lig_x[3 * i + 2] * inv0; }
    m1[0] = block_sum(s0, red) / cnt; m1[1] = block_sum(s1, red) / cnt; m1[2] = block_sum(s2, red) / cnt;
  }
  // 2. KL sums of the prior (kl_prior) and the centre of mass of the noised coordinates (noised_representation)
  float sx = 0.f, sh = 0.f, z0 = 0.f, z1 = 0.f, z2 = 0.f;
  for (int i = l0 + t; i < l1; i += kLossThreads) {
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const float x = lig_x[3 * i + d] * inv0 - m1[d];
      const float mu = alpha_T * x;
      sx += mu * mu;
      const float z = alpha_t * x + sigma_t * eps[(size_t)i * ldl + d];
      if (d == 0) z0 += z; else if (d == 1) z1 += z; else z2 += z;
    }
    for (int k = 0; k < a; ++k) {
      const float h = (lig_h[(size_t)i * a + k] - c.nb1) * inv1;
      const float mu = alpha_T * h;
      sh += mu * mu;
    }
  }
  sx = block_sum(sx, red); sh = block_sum(sh, red);
  float m2[3] = {0.f, 0.f, 0.f};
  if (c.remove_com) { m2[0] = block_sum(z0, red) / cnt; m2[1] = block_sum(z1, red) / cnt; m2[2] = block_sum(z2, red) / cnt; }

  // 3. z_t and the categorical term -log p(h | z_t) (_log_ph_given_z0 on z_t: training mode, conditional_model.py:285-302)
  const float sig_cat = sigma_t * c.nv1;
  float lh = 0.f;
  for (int i = l0 + t; i < l1; i += kLossThreads) {
    float* zr = z_t + (size_t)i * ldl;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const float xn = lig_x[3 * i + d] * inv0;
      if (lig_xn) lig_xn[3 * i + d] = xn;                       // normalize() leaves the normalised batch in the caller's dicts
      const float x = xn - m1[d];
      zr[d] = (alpha_t * x + sigma_t * eps[(size_t)i * ldl + d]) - m2[d];
    }
    float mx = -INFINITY;
    for (int k = 0; k < a; ++k) {
      const float h = (lig_h[(size_t)i * a + k] - c.nb1) * inv1;
      if (lig_hn) lig_hn[(size_t)i * a + k] = h;
      const float z = alpha_t * h + sigma_t * eps[(size_t)i * ldl + 3 + k];
      zr[3 + k] = z;
      const float cen = (z * c.nv1 + c.nb1) - 1.0f;
      const float lp = logf(cdf_gauss((cen + 0.5f) / sig_cat) - cdf_gauss((cen - 0.5f) / sig_cat) + 1e-10f);
      mx = fmaxf(mx, lp);
    }
    float se = 0.f;
    for (int k = 0; k < a; ++k) {
      const float cen = (zr[3 + k] * c.nv1 + c.nb1) - 1.0f;
      const float lp = logf(cdf_gauss((cen + 0.5f) / sig_cat) - cdf_gauss((cen - 0.5f) / sig_cat) + 1e-10f);
      se += expf(lp - mx);
    }
    const float lse = mx + logf(se);
    for (int k = 0; k < a; ++k) {
      const float cen = (zr[3 + k] * c.nv1 + c.nb1) - 1.0f;
      const float lp = logf(cdf_gauss((cen + 0.5f) / sig_cat) - cdf_gauss((cen - 0.5f) / sig_cat) + 1e-10f);
      const float onehot = ((lig_h[(size_t)i * a + k] - c.nb1) * inv1) * c.nv1 + c.nb1;
      lh += (lp - lse) * onehot;
    }
  }
  lh = block_sum(lh, red);
  // 4. the pocket: normalised, shifted by both ligand centres
  for (int i = p0 + t; i < p1; i += kLossThreads) {
    float* pr = xh_pocket + (size_t)i * ldp;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const float xn = poc_x[3 * i + d] * inv0;
      if (poc_xn) poc_xn[3 * i + d] = xn;
      pr[d] = (xn - m1[d]) - m2[d];
    }
    for (int k = 0; k < r; ++k) {
      const float h = (poc_h[(size_t)i * r + k] - c.nb1) * inv1;
      if (poc_hn) poc_hn[(size_t)i * r + k] = h;
      pr[3 + k] = h;
    }
  }
  if (t == 0) {
    const int B = c.batch;
    const float dof = c.remove_com ? (float)((nl - 1) * 3) : (float)(nl * 3);       // subspace_dimensionality
    const float tz = ti == 0.0f ? 1.0f : 0.0f;
    ps[LS_T * B + b] = tt; ps[LS_GAMMA_T * B + b] = g_t; ps[LS_GAMMA_S * B + b] = g_s;
    ps[LS_ALPHA_T * B + b] = alpha_t; ps[LS_SIGMA_T * B + b] = sigma_t;
    ps[LS_SNR_W * B + b] = 1.0f - expf(-(g_s - g_t));                                // 1 - SNR(gamma_s - gamma_t)
    ps[LS_NEG_LOG_C * B + b] = -(dof * (-(0.5f * g_0) - 0.91893853320467274178f));    // -log_constants_p_x_given_z0
    // gaussian_KL(mu2, sigma_T, 1, d) = d log(1 / sigma_T) + 0.5 (d sigma_T^2 + mu2) - 0.5 d;  d = dof for x, 1 for h
    const float lq = logf(1.0f / sigma_T), q2 = sigma_T * sigma_T;
    const float kl_x = dof * lq + 0.5f * (dof * q2 + sx) - 0.5f * dof;
    const float kl_h = lq + 0.5f * (q2 + sh) - 0.5f;
    ps[LS_KL * B + b] = kl_x + kl_h;
    ps[LS_L0_H * B + b] = -lh * tz;
    float lpn = 0.f;
    if (logpn_table) {
      const int i1 = nl < c.n1_tab ? nl : c.n1_tab - 1, i2 = np < c.n2_tab ? np : c.n2_tab - 1;
      lpn = logpn_table[(size_t)i1 * c.n2_tab + i2];
    }
    ps[LS_LOG_PN * B + b] = lpn;
    ps[LS_DELTA_LOG_PX * B + b] = -dof * logf(c.nv0);
    ps[LS_T_IS_ZERO * B + b] = tz;
  }
}

// error terms after the network call: error_t = sum (eps - net)^2 (x (1 - [t = 0])), L0_x = 0.5 sum over the coordinates
// (x [t = 0]), xh_hat = z_t / alpha_t - net sigma_t / alpha_t, and the logged per-sample means of |net|
__global__ __launch_bounds__(kLossThreads) void loss_cond_post_kernel(
    LossCfg c, const float* net, const float* eps, const float* z_t, const float* lig_h, const long long* lig_mask,
    const float* ps, float* xh_hat, float* out) {
  __shared__ float red[kLossThreads];
  __shared__ int seg[2];
  const int b = blockIdx.x, t = threadIdx.x, a = c.atom_nf, ld = 3 + a, B = c.batch;
  if (t == 0) { seg[0] = lower_bound_i64(lig_mask, c.n_lig, b); seg[1] = lower_bound_i64(lig_mask, c.n_lig, b + 1); }
  __syncthreads();
  const int l0 = seg[0], l1 = seg[1];
  const float alpha_t = ps[LS_ALPHA_T * B + b], sigma_t = ps[LS_SIGMA_T * B + b], tz = ps[LS_T_IS_ZERO * B + b];
  float e_all = 0.f, e_x = 0.f, ax = 0.f, ah = 0.f;
  for (int i = l0 + t; i < l1; i += kLossThreads) {
    const bool virt = c.vnode_idx >= 0 && lig_h[(size_t)i * a + c.vnode_idx] != c.nb1;   // ((one_hot - nb) / nv).bool()
    float sx_ = 0.f, sa = 0.f, mx_ = 0.f, mh = 0.f;
    for (int k = 0; k < ld; ++k) {
      const size_t o = (size_t)i * ld + k;
      const float n = net[o], d = eps[o] - n;
      float sq = d * d;
      if (k < 3 && virt) sq = 0.f;
      sa += sq;
      if (k < 3) { sx_ += sq; mx_ += fabsf(n); } else mh += fabsf(n);
      xh_hat[o] = z_t[o] / alpha_t - n * sigma_t / alpha_t;
    }
    e_all += sa; e_x += sx_;
    ax += mx_ / 3.0f; ah += mh / (float)a;
  }
  e_all = block_sum(e_all, red); e_x = block_sum(e_x, red); ax = block_sum(ax, red); ah = block_sum(ah, red);
  if (t == 0) {
    const float cnt = (float)(l1 - l0 > 1 ? l1 - l0 : 1);
    out[LO_ERR_T * B + b] = e_all * (1.0f - tz);
    out[LO_L0_X * B + b] = (0.5f * e_x) * tz;
    out[LO_INFO_X * B + b] = ax / cnt;
    out[LO_INFO_H * B + b] = ah / cnt;
  }
}

// d net = -2 (eps - net) [g_err (1 - tz) + 0.5 g_l0x tz on the coordinates] (zero on the coordinates of virtual atoms)
//         - g_hat sigma_t / alpha_t
__global__ __launch_bounds__(kLossThreads) void loss_cond_post_bwd_kernel(
    LossCfg c, const float* net, const float* eps, const float* lig_h, const long long* lig_mask, const float* ps,
    const float* g_err, const float* g_l0x, const float* g_hat, float* d_net) {
  const int a = c.atom_nf, ld = 3 + a, B = c.batch;
  const size_t n = (size_t)c.n_lig * ld;
  for (size_t o = (size_t)blockIdx.x * kLossThreads + threadIdx.x; o < n; o += (size_t)gridDim.x * kLossThreads) {
    const int i = (int)(o / ld), k = (int)(o % ld);
    const int b = (int)lig_mask[i];
    const float tz = ps[LS_T_IS_ZERO * B + b];
    float w = (g_err ? g_err[b] : 0.f) * (1.0f - tz);
    if (k < 3) {
      w += 0.5f * (g_l0x ? g_l0x[b] : 0.f) * tz;
      if (c.vnode_idx >= 0 && lig_h[(size_t)i * a + c.vnode_idx] != c.nb1) w = 0.f;
    }
    float g = -2.0f * (eps[o] - net[o]) * w;
    if (g_hat) g -= g_hat[o] * ps[LS_SIGMA_T * B + b] / ps[LS_ALPHA_T * B + b];
    d_net[o] = g;
  }
}

}  // namespace dsbdd
