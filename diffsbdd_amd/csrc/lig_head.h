// Output head of a ligand-output-only call in pocket-conditioning mode, ONE launch instead of three
// (embedding_out GEMM + decoder MLP + finalize: 21.6 + 12.8 + 4.2 us for 1472 ligand rows, all launch-latency):
//
//     hout = h W_out + b_out                    egnn_new.py:241   (the time column, dynamics.py:147, is never formed)
//     eps_h = W1 SiLU(W0 hout[:, :J] + b0) + b1   dynamics.py:152 (atom_decoder)
//     eps_x = x - x_in, NaN flag                dynamics.py:136,155-159
//
// 50 MFLOP in all: plain fp32 FMAs, one workgroup per 16 ligand rows; every output is one fmaf chain over k in ascending
// order (a pure function of the row: bitwise independent of the batch composition).  The pocket rows do not move in this
// mode (velocity 0); their input coordinates are still screened for NaN, as the reference's check over all nodes does.
#pragma once
#include "common.h"

namespace dsbdd {

struct LigHeadArgs {
  const float* h; int H;                  // [n_lig ..][H]
  const float* WoT; int ldo; const float* bo; int J;      // embedding_out^T [H][ldo], bias; J = joint_nf (columns used)
  const float* W0T; int ld0; const float* b0; int n_hid;  // decoder layer 1 ^T [J][ld0]
  const float* W1T; int ld1; const float* b1; int n_out;  // decoder layer 2 ^T [n_hid][ld1]
  const float* x; const float* x_in; int n_lig; int n_nodes;
  float* eps_lig; int dl; int* status;
};

constexpr int kHeadRows = 16;
constexpr int kHeadKC = 32;      // k rows of W_out^T staged in LDS at a time

__global__ __launch_bounds__(kThreads) void lig_head_kernel(LigHeadArgs p) {
  __shared__ __attribute__((aligned(16))) float sH[kHeadRows * 256];     // rows of h, then hout
  __shared__ float sO[kHeadRows * 128];
  __shared__ float sW[kHeadKC * 128];
  __shared__ float sD[kHeadRows * 64];
  const int t = threadIdx.x;
  const int r0 = blockIdx.x * kHeadRows;
  const int nr = min(kHeadRows, p.n_lig - r0);
  const int H = p.H, J = p.J;
  for (int i = t; i < kHeadRows * H / 4; i += kThreads) {
    const int r = i / (H / 4), k4 = i - r * (H / 4);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < nr) v = ld4(p.h + (size_t)(r0 + r) * H + 4 * k4);
    *reinterpret_cast<float4*>(sH + r * H + 4 * k4) = v;
  }
  __syncthreads();
  // embedding_out: thread (c, half) -> column c of 8 rows; W_out^T goes through LDS in chunks of 32 k (one coalesced
  // burst per chunk: the loop is a latency chain, not a bandwidth problem)
  {
    const int c = t & 127, rb = (t >> 7) * 8;
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
    for (int k0 = 0; k0 < H; k0 += kHeadKC) {
      {   // all of a thread's loads first (independent, one latency), then the LDS stores
        constexpr int NL = kHeadKC * 128 / kThreads;
        float wv[NL];
#pragma unroll
        for (int u = 0; u < NL; ++u) {
          const int i = t + u * kThreads, kk = i >> 7, cc = i & 127;
          wv[u] = p.WoT[(size_t)(k0 + kk) * p.ldo + (cc < J ? cc : 0)];
        }
#pragma unroll
        for (int u = 0; u < NL; ++u) sW[t + u * kThreads] = wv[u];
      }
      __syncthreads();
#pragma unroll 2
      for (int k = 0; k < kHeadKC; k += 4) {
        const float w0 = sW[k * 128 + c], w1 = sW[(k + 1) * 128 + c], w2 = sW[(k + 2) * 128 + c], w3 = sW[(k + 3) * 128 + c];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float4 hv = *reinterpret_cast<const float4*>(sH + (rb + i) * H + k0 + k);
          acc[i] = fmaf(hv.w, w3, fmaf(hv.z, w2, fmaf(hv.y, w1, fmaf(hv.x, w0, acc[i]))));
        }
      }
      __syncthreads();
    }
    if (c < J) {
      const float b = p.bo[c];
#pragma unroll
      for (int i = 0; i < 8; ++i) sO[(rb + i) * 128 + c] = acc[i] + b;
    }
  }
  __syncthreads();
  for (int idx = t; idx < kHeadRows * p.n_hid; idx += kThreads) {
    const int r = idx / p.n_hid, j = idx - r * p.n_hid;
    float a = 0.f;
    for (int k = 0; k < J; ++k) a = fmaf(sO[r * 128 + k], p.W0T[(size_t)k * p.ld0 + j], a);
    sD[r * 64 + j] = silu(a + p.b0[j]);
  }
  __syncthreads();
  for (int idx = t; idx < kHeadRows * p.n_out; idx += kThreads) {
    const int r = idx / p.n_out, c = idx - r * p.n_out;
    if (r >= nr) continue;
    float a = 0.f;
    for (int j = 0; j < p.n_hid; ++j) a = fmaf(sD[r * 64 + j], p.W1T[(size_t)j * p.ld1 + c], a);
    p.eps_lig[(size_t)(r0 + r) * p.dl + 3 + c] = a + p.b1[c];
  }
  bool nan = false;
  if (t < 3 * nr) {
    const int r = t / 3, k = t - 3 * r;
    const float v = p.x[3 * (size_t)(r0 + r) + k] - p.x_in[3 * (size_t)(r0 + r) + k];
    nan = v != v;
    p.eps_lig[(size_t)(r0 + r) * p.dl + k] = v;
  }
  // pocket rows: velocity 0 in this mode unless an input coordinate is not finite (inf - inf); this workgroup's slice
  {
    const int n_poc3 = 3 * (p.n_nodes - p.n_lig);
    const int per = (n_poc3 + gridDim.x - 1) / gridDim.x;
    const int lo = 3 * p.n_lig + blockIdx.x * per, hi = min(lo + per, 3 * p.n_nodes);
    for (int i = lo + t; i < hi; i += kThreads) {
      const float v = p.x[i] - p.x_in[i];
      nan = nan || (v != v);
    }
  }
  if (nan) atomicOr(p.status, 1);
}

inline bool lig_head_fits(const LigHeadArgs& a) {
  return a.H % kHeadKC == 0 && a.H <= 256 && a.J <= 128 && a.n_hid <= 64 && a.n_out <= 64 && a.n_lig > 0;
}

inline hipError_t launch_lig_head(hipStream_t s, const LigHeadArgs& a) {
  hipLaunchKernelGGL(lig_head_kernel, dim3((a.n_lig + kHeadRows - 1) / kHeadRows), dim3(kThreads), 0, s, a);
  return hipGetLastError();
}

}  // namespace dsbdd
