// Backward pass of the fused edge stages ("hand-written backward for the fused edge kernels") and the
// node-level weight-gradient GEMM, so that the reference's training step (lightning_modules.py:337-363 ->
// conditional_model.py:202-330 / en_diffusion.py:336-469 -> dynamics.py:87-167 -> egnn_new.py:31-58,96-122) runs on
// hand-written gfx950 kernels.
//
// Forward of an edge MLP (edge_mlp.h), per edge e = (i, j):
//     z1 = P[i] + Q[j] + d wd + d0 wd0 + tab[type]      a1 = SiLU(z1)
//     z2 = a1 W2^T + b2
//   GCL   (egnn_new.py:31-52):  m = SiLU(z2), att = sigmoid(m . wa + ba), out = m att, agg[i] = sum_e out / nf
//   COORD (egnn_new.py:96-122): phi = w3 . SiLU(z2), trans = u T(phi) (+ cross T(phi_x)), xagg[i] = sum_e trans / nf
//
// Backward = three GEMM-shaped passes per MLP, the [E][H] activations are RECOMPUTED from the per-node projections
// (nothing of size [E][H] is kept from the forward pass):
//   kernel A (edge_bwd_a_kernel): recompute a1 (lane = edge, as edge_wave.h) and z2 = a1 W2^T + b2 on the matrix cores,
//       the epilogue turns the accumulators into dz2 (GCL: attention + SiLU backward from the gathered d_agg rows;
//       COORD: scalar head, tanh and the geometry of u / cross backward) -> dz2 [E][H] and a1 [E][H] to HBM, bias /
//       head gradients as per-(wave, half) partial vectors, the coordinate gradients of the geometry per edge.
//   wgrad (wgrad_kernel): dW2 = dz2^T a1, a split-K "TN" GEMM over the edges with an ordered reduction (no atomics).
//   kernel B (edge_bwd_b_kernel): da1 = dz2 W2 (A operand = the lane's dz2 row, straight from global memory), the epilogue
//       recomputes z1 in accumulator layout, dz1 = da1 SiLU'(z1) -> dz1 [E][H], the gradients of wd / wd0 / tab as
//       partial vectors, the gradient w.r.t. |d|^2 and d0 per edge.
//   rows_gather_kernel: dP[i] = sum over row i's edges of dz1[e], dQ[i] = sum of dz1[rev(e)] (rev(e) = the edge (j, i):
//       the radius graph is symmetric, so "edges with column i" = the reverses of row i's edges) -- fixed order, no atomics.
//   edge_to_node3_kernel: coordinate gradient d_x[i] from the per-edge pieces, the same way.
// Every sum over edges is taken in a fixed order: the gradients are bitwise reproducible.
#pragma once
#include "common.h"
#include "edge_mlp.h"
#include "edge_wave.h"

namespace dsbdd {

struct TrainEdgeArgs {
  const int* erow; const int* ecol; const float* ed0; int E;   // E: number of list entries this launch walks
  const float* x; int n_lig; int n_nodes;
  const float* P; const float* Q; int ldpq;
  const float* wd; const float* wd0; const float* table; const float* b2;
  const float* Bmat;     // kernel A: W2^T [H][H] ([k = in][out]); kernel B: W2 [H][H] ([k = out][in])
  const float* head;     // GCL: att_w (nullptr: no attention); COORD: w3
  const float* head_b;   // GCL: att_b
  const float* d_agg;    // GCL: [N][H]  gradient w.r.t. the normalised aggregate
  const float* d_xagg;   // COORD: [N][3] gradient w.r.t. the normalised coordinate aggregate
  const int* node_batch; const float* mean;
  float norm_constant, coords_range; int use_tanh; int which;   // which: 0 = u phi term, 1 = cross phi_x term
  float norm_factor;
  float* a1_out;         // kernel A: [E][H]
  const float* dz_in;    // kernel B: dz2 [E][H]
  float* dz_out;         // kernel A: dz2 [E][H]; kernel B: dz1 [E][H]
  float* part;           // [gridDim.x][8][H] partial vectors, one slot per workgroup (its 8 (wave, half) sums added in order):
                         // kernel B writes rows 0 .. 4, kernel A rows 5 .. 7 (same grid for both)
  float* gxr; float* gxc; float* gm;   // COORD kernel A: [E][3] gradient pieces for x[row], x[col], the sample mean
  float* gd; float* gd0; // kernel B: [E] gradient w.r.t. the current and the input squared distance
};

constexpr int kPartA = 3;   // kernel A: d_b2, d_head (att_w / w3), [0] = d_att_b
constexpr int kPartB = 5;   // kernel B: d_wd, d_wd0, d_tab[0..2]
constexpr int kPartAll = kPartA + kPartB;   // one [8][H] slot per workgroup, B's rows first: the layout of dsbdd_train_mlp_grad::d_vec,
                                            // so ONE ordered reduction behind kernel B serves both kernels (round 6)

__device__ __forceinline__ float dsilu_from(float z, float sg) { return sg * (1.0f + z * (1.0f - sg)); }

// Stream of [32][H] slices of a row-major [H][H] matrix through a two-slot LDS buffer.
template <int H>
struct SliceStream {
  static constexpr int BK = 32;
  static constexpr int NQ = H / 4;
  static constexpr int BI = BK * NQ / kThreads;
  f32x4 stg[BI];
  __device__ __forceinline__ void load(const float* mat, int ks, int t) {
    const float* src = mat + (size_t)ks * BK * H;
#pragma unroll
    for (int i = 0; i < BI; ++i) stg[i] = ldv4(src + (size_t)(i * kThreads + t) * 4);
  }
  __device__ __forceinline__ void store(float* buf, int t) {
#pragma unroll
    for (int i = 0; i < BI; ++i) *reinterpret_cast<f32x4*>(buf + (size_t)(i * kThreads + t) * 4) = stg[i];
  }
};

// ---------------------------------------------------------------------------------------------------------------------
// kernel A
template <int H, int MODE>
__global__ __launch_bounds__(kThreads, 1) void edge_bwd_a_kernel(TrainEdgeArgs p) {
  constexpr int BK = 32, CT = H / 32, NK = H / BK;
  __shared__ __attribute__((aligned(16))) float sB[2 * BK * H];
  __shared__ __attribute__((aligned(16))) float sV[7 * H];
  __shared__ __attribute__((aligned(16))) float sS[4][96];
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int half = lane >> 5, j = lane & 31;
  float* s_phi = sS[w];            // [32]
  float* s_aux = sS[w] + 32;       // [32]
  int* s_row = reinterpret_cast<int*>(sS[w] + 64);   // [32]
  const bool attention = MODE == MODE_GCL && p.head != nullptr;

  for (int i = t; i < H; i += kThreads) {
    sV[i] = p.wd[i];
    sV[H + i] = p.wd0[i];
    sV[2 * H + i] = p.table[i];
    sV[3 * H + i] = p.table[H + i];
    sV[4 * H + i] = p.table[2 * H + i];
    sV[5 * H + i] = p.b2[i];
    sV[6 * H + i] = p.head ? p.head[i] : 0.f;
  }
  const float att_b = attention ? p.head_b[0] : 0.f;
  const float inv_norm = 1.0f / p.norm_factor;
  const int ntiles = (p.E + 127) / 128;

  float pb2[CT], pv1[CT], ps = 0.f;
#pragma unroll
  for (int c = 0; c < CT; ++c) { pb2[c] = 0.f; pv1[c] = 0.f; }

  SliceStream<H> st;
  st.load(p.Bmat, 0, t);
  st.store(sB, t);
  __syncthreads();
  int bslice = 0;

#pragma unroll 1
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int e_base = tile * 128 + w * 32;
    const int e = e_base + j;
    bool active = e < p.E;
    int r = 0, cidx = 0;
    float d0 = 0.f;
    if (active) {
      r = p.erow[e]; cidx = p.ecol[e]; d0 = p.ed0[e];
      if ((unsigned)r >= (unsigned)p.n_nodes || (unsigned)cidx >= (unsigned)p.n_nodes) { active = false; r = 0; cidx = 0; d0 = 0.f; }
    }
    float xr[3], xc[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { xr[k] = p.x[3 * r + k]; xc[k] = p.x[3 * cidx + k]; }
    const float ddx = xr[0] - xc[0], ddy = xr[1] - xc[1], ddz = xr[2] - xc[2];
    const float d = active ? ddx * ddx + ddy * ddy + ddz * ddz : 0.f;
    const bool rl = r < p.n_lig, cl = cidx < p.n_lig;
    const int ty = (rl && cl) ? 1 : ((!rl && !cl) ? 2 : 0);

    f32x16 acc[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      const float bv = sV[5 * H + 32 * c + j];
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[c][q] = bv;
    }
    const float* Pp = p.P + (size_t)r * p.ldpq + 4 * half;
    const float* Qp = p.Q + (size_t)cidx * p.ldpq + 4 * half;
    f32x4 pc = ldv4(Pp), qc = ldv4(Qp), pn = pc, qn = qc;
    const f32x2 dd2 = splat2(d), dz2v = splat2(d0);

#pragma unroll 1
    for (int kt = 0; kt < NK; ++kt) {
      // next slice of the stream (after the tile's last slice: slice 0 again, for the next tile)
      st.load(p.Bmat, kt + 1 < NK ? kt + 1 : 0, t);
      const float* bcur = sB + (bslice & 1) * BK * H + (4 * half) * H + j;
      const float* vk = sV + kt * BK + 4 * half;
      const float* vt = vk + (2 + ty) * H;
#pragma unroll
      for (int g = 0; g < BK / 8; ++g) {
        const int kb = kt * BK + 8 * g;
        if (g + 1 < BK / 8 || kt + 1 < NK) { pn = ldv4(Pp + kb + 8); qn = ldv4(Qp + kb + 8); }
        const f32x4 wd4 = *reinterpret_cast<const f32x4*>(vk + 8 * g);
        const f32x4 wz4 = *reinterpret_cast<const f32x4*>(vk + H + 8 * g);
        const f32x4 tb4 = *reinterpret_cast<const f32x4*>(vt + 8 * g);
        f32x2 alo = pk_fma(dz2v, wz4.xy, pk_fma(dd2, wd4.xy, pc.xy + qc.xy)) + tb4.xy;
        f32x2 ahi = pk_fma(dz2v, wz4.zw, pk_fma(dd2, wd4.zw, pc.zw + qc.zw)) + tb4.zw;
        alo = silu2(alo);
        ahi = silu2(ahi);
        const float a[4] = {alo.x, alo.y, ahi.x, ahi.y};
        if (e < p.E) {
          const f32x4 av = active ? f32x4{a[0], a[1], a[2], a[3]} : f32x4{0.f, 0.f, 0.f, 0.f};
          *reinterpret_cast<f32x4*>(p.a1_out + (size_t)e * H + kb + 4 * half) = av;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float* brow = bcur + (8 * g + i) * H;
#pragma unroll
          for (int c = 0; c < CT; ++c) acc[c] = mfma32(a[i], brow[32 * c], acc[c]);
        }
        pc = pn; qc = qn;
      }
      st.store(sB + ((bslice + 1) & 1) * BK * H, t);
      ++bslice;
      __syncthreads();
    }

    // ================= epilogue (wave-private) =================
    if (half == 0) s_row[j] = active ? r : -1;
    wave_lds_fence();
    int rowm[16];
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      const int4 v = *reinterpret_cast<const int4*>(s_row + 8 * q4 + 4 * half);
      rowm[4 * q4] = v.x; rowm[4 * q4 + 1] = v.y; rowm[4 * q4 + 2] = v.z; rowm[4 * q4 + 3] = v.w;
    }
    // from here on rowm[q] is the BYTE offset of this lane's first column in row q's d_agg row (-1: inactive entry): the
    // row-gathered loads of the epilogue take it as it is, everything else only looks at its sign
#pragma unroll
    for (int q = 0; q < 16; ++q) rowm[q] = rowm[q] >= 0 ? (rowm[q] * H + j) * 4 : -1;
    auto bcast16 = [&](const float* src, float (&dst)[16]) {     // dst[q] = src[mfma_row(q, lane)]
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const float4 v = *reinterpret_cast<const float4*>(src + 8 * q4 + 4 * half);
        dst[4 * q4] = v.x; dst[4 * q4 + 1] = v.y; dst[4 * q4 + 2] = v.z; dst[4 * q4 + 3] = v.w;
      }
    };

    if (MODE == MODE_GCL) {
      float att[16], tt[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) { att[q] = 1.f; tt[q] = 0.f; }
      // d_out = d_agg[row] / nf in accumulator layout: 16 row-gathered loads per column tile.  They are requested one
      // column tile AHEAD of their use (round 6: the epilogue's waves were waiting half of their cycles on these loads in
      // 16 dependent batches, profiles/r6q_train_pmc_2.md), and the gate's dot product and s = sum_f d_out m share ONE
      // SiLU pass over the accumulators (same operations in the same order as the two passes they replace: same bits).
      // (buffer loads: one 32-bit byte offset per register row, the column tile as scalar offset; an inactive row's offset
      // lies outside the descriptor and reads as 0 -- no per-lane 64-bit addresses, no selects)
      const __amdgpu_buffer_rsrc_t dagg_rs = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<float*>(p.d_agg), 0, (int)min((size_t)p.n_nodes * H * 4, (size_t)0x7FFFFFFF), 0x00020000);
      auto load_dout = [&](int c, float (&dst)[16]) {
#pragma unroll
        for (int q = 0; q < 16; ++q)
          dst[q] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(dagg_rs, rowm[q], 32 * c * 4, 0)) * inv_norm;
      };
      if (attention) {
        float part[16], part_s[16], dcur[16], dnxt[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) { part[q] = 0.f; part_s[q] = 0.f; }
        load_dout(0, dcur);
#pragma unroll
        for (int c = 0; c < CT; ++c) {
          if (c + 1 < CT) load_dout(c + 1, dnxt);
          const float aw = sV[6 * H + 32 * c + j];
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const float m = silu(acc[c][q]);
            part[q] = fmaf(m, aw, part[q]);
            part_s[q] = fmaf(dcur[q], m, part_s[q]);
          }
#pragma unroll
          for (int q = 0; q < 16; ++q) dcur[q] = dnxt[q];
          __builtin_amdgcn_sched_barrier(0);
        }
        const float gate = sigmoidf_fast(reduce16_half_wave(part, j) + att_b);   // register j >> 1 of this half
        const float stot = reduce16_half_wave(part_s, j);                        // s = sum_f d_out m
        // the 16 registers of this half are rows mfma_row(q, lane) = (q & 3) + 8 (q >> 2) + 4 half
        s_phi[mfma_row(j >> 1, lane)] = gate;
        s_aux[mfma_row(j >> 1, lane)] = stot * gate * (1.0f - gate);
        wave_lds_fence();
        bcast16(s_phi, att);
        bcast16(s_aux, tt);
        wave_lds_fence();
        if (j == 0) {
#pragma unroll
          for (int q = 0; q < 16; ++q) ps += tt[q];
        }
      }
      float dcur3[16], dnxt3[16];
      load_dout(0, dcur3);
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        if (c + 1 < CT) load_dout(c + 1, dnxt3);
        const float aw = sV[6 * H + 32 * c + j];
        float b2p = 0.f, awp = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const float z = acc[c][q];
          const float sg = sigmoidf_fast(z);
          const float m = z * sg;
          const float dout = dcur3[q];
          const float dm = fmaf(tt[q], aw, dout * att[q]);
          const float dz = dm * dsilu_from(z, sg);
          acc[c][q] = dz;
          b2p += dz;
          awp = fmaf(tt[q], m, awp);
        }
        pb2[c] += b2p; pv1[c] += awp;
#pragma unroll
        for (int q = 0; q < 16; ++q) dcur3[q] = dnxt3[q];
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      // phi = w3 . SiLU(z2)
      float part[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) part[q] = 0.f;
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        const float wv = sV[6 * H + 32 * c + j];
#pragma unroll
        for (int q = 0; q < 16; ++q) part[q] = fmaf(silu(acc[c][q]), wv, part[q]);
      }
      s_phi[mfma_row(j >> 1, lane)] = reduce16_half_wave(part, j);
      wave_lds_fence();
      const float ph = s_phi[j];
      wave_lds_fence();
      // lane = edge: d_phi and the geometry gradients (egnn_new.py:100-109, 296-316)
      float g0 = 0.f, g1 = 0.f, g2 = 0.f;
      if (active) { g0 = p.d_xagg[3 * r] * inv_norm; g1 = p.d_xagg[3 * r + 1] * inv_norm; g2 = p.d_xagg[3 * r + 2] * inv_norm; }
      float T = ph, dT = 1.f;
      if (p.use_tanh) { const float th = tanhf(ph); T = th * p.coords_range; dT = (1.0f - th * th) * p.coords_range; }
      float dphi = 0.f, ar[3] = {0.f, 0.f, 0.f}, ac[3] = {0.f, 0.f, 0.f}, am[3] = {0.f, 0.f, 0.f};
      if (active) {
        if (p.which == 0) {
          const float nrm = sqrtf(d + 1e-8f), den = nrm + p.norm_constant;
          const float ux = ddx / den, uy = ddy / den, uz = ddz / den;
          dphi = (g0 * ux + g1 * uy + g2 * uz) * dT;
          const float du0 = g0 * T, du1 = g1 * T, du2 = g2 * T;
          const float sdot = (du0 * ddx + du1 * ddy + du2 * ddz) / (nrm * den * den);
          ar[0] = du0 / den - ddx * sdot; ar[1] = du1 / den - ddy * sdot; ar[2] = du2 / den - ddz * sdot;
          ac[0] = -ar[0]; ac[1] = -ar[1]; ac[2] = -ar[2];
        } else {
          const int b = p.node_batch[r];
          const float m0 = p.mean[3 * b], m1 = p.mean[3 * b + 1], m2 = p.mean[3 * b + 2];
          const float a0 = xr[0] - m0, a1 = xr[1] - m1, a2 = xr[2] - m2;
          const float b0 = xc[0] - m0, b1 = xc[1] - m1, b2 = xc[2] - m2;
          const float c0 = a1 * b2 - a2 * b1, c1 = a2 * b0 - a0 * b2, c2 = a0 * b1 - a1 * b0;
          const float cn = sqrtf(c0 * c0 + c1 * c1 + c2 * c2), cden = cn + p.norm_constant;
          dphi = (g0 * c0 + g1 * c1 + g2 * c2) / cden * dT;
          const float dc0 = g0 * T, dc1 = g1 * T, dc2 = g2 * T;
          const float sdot = cn > 0.f ? (dc0 * c0 + dc1 * c1 + dc2 * c2) / (cn * cden * cden) : 0.f;
          const float e0 = dc0 / cden - c0 * sdot, e1 = dc1 / cden - c1 * sdot, e2 = dc2 / cden - c2 * sdot;   // d cr
          // cr = a x b:  d a = b x d_cr,  d b = d_cr x a
          ar[0] = b1 * e2 - b2 * e1; ar[1] = b2 * e0 - b0 * e2; ar[2] = b0 * e1 - b1 * e0;
          ac[0] = e1 * a2 - e2 * a1; ac[1] = e2 * a0 - e0 * a2; ac[2] = e0 * a1 - e1 * a0;
          am[0] = -(ar[0] + ac[0]); am[1] = -(ar[1] + ac[1]); am[2] = -(ar[2] + ac[2]);
        }
      }
      if (half == 0 && e < p.E) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          p.gxr[3 * (size_t)e + k] = ar[k];
          p.gxc[3 * (size_t)e + k] = ac[k];
          if (p.gm) p.gm[3 * (size_t)e + k] = am[k];
        }
      }
      if (half == 0) s_aux[j] = dphi;
      wave_lds_fence();
      float dph[16];
      bcast16(s_aux, dph);
      wave_lds_fence();
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        const float wv = sV[6 * H + 32 * c + j];
        float b2p = 0.f, w3p = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const float z = acc[c][q];
          const float sg = sigmoidf_fast(z);
          const float dz = dph[q] * wv * dsilu_from(z, sg);
          w3p = fmaf(dph[q], z * sg, w3p);
          acc[c][q] = dz;
          b2p += dz;
        }
        pb2[c] += b2p; pv1[c] += w3p;
      }
    }
    // dz2 -> HBM (rows of inactive list entries are written as zeros: the weight-gradient GEMM sums over every row)
#pragma unroll
    for (int c = 0; c < CT; ++c) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int em = e_base + mfma_row(q, lane);
        if (em < p.E) p.dz_out[(size_t)em * H + 32 * c + j] = rowm[q] >= 0 ? acc[c][q] : 0.f;
      }
    }
    wave_lds_fence();
  }
  // partial vectors: the 8 (wave, half) sums of this workgroup meet in LDS (the slice buffers are free now), slot order
  __syncthreads();
  float* sp = sB + (size_t)((w * 2 + half) * kPartA) * H;
#pragma unroll
  for (int c = 0; c < CT; ++c) {
    sp[32 * c + j] = pb2[c];
    sp[H + 32 * c + j] = pv1[c];
    sp[2 * H + 32 * c + j] = (c == 0 && j == 0) ? ps : 0.f;
  }
  __syncthreads();
  for (int i = t; i < kPartA * H; i += kThreads) {
    float v = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) v += sB[(size_t)q * kPartA * H + i];
    p.part[(size_t)blockIdx.x * kPartAll * H + (size_t)kPartB * H + i] = v;      // rows 5 .. 7 of the workgroup's [8][H] slot
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// kernel E (round 6): the message stage's kernel A WITHOUT its H x H layer.  The forward pass of the network path keeps
// z2 = W2 a1 + b2 of every edge (edge_wave_kernel<.., STORE>, 4 E H bytes per message stage -- the device has 288 GB), so
// what is left of kernel A is element-wise per edge: a1 = SiLU(z1) from the row-gathered projections (for the weight
// gradient), m = SiLU(z2), the attention gate and its backward, dz2.  One WAVE per edge, lane l holds columns
// V l .. V l + V - 1 (V = H / 64: 16-byte accesses at H = 256, every row a contiguous 4 H bytes), 16 waves per workgroup,
// the grid of kernel B (one [8][H] partial-vector slot per workgroup: rows 5 .. 7 here, as kernel A).  Row sums are wave
// butterflies in a fixed order: bitwise reproducible, and within rounding of kernel A's (which sums the same products in
// accumulator-tile order).
constexpr int kThreadsE = 1024;
template <int V>
__device__ __forceinline__ void ld_cols(const float* row, int lane, float (&v)[V]) {
  if constexpr (V == 4) { const float4 q = *reinterpret_cast<const float4*>(row + 4 * lane); v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w; }
  else if constexpr (V == 2) { const float2 q = *reinterpret_cast<const float2*>(row + 2 * lane); v[0] = q.x; v[1] = q.y; }
  else {
#pragma unroll
    for (int k = 0; k < V; ++k) v[k] = row[V * lane + k];
  }
}
template <int V>
__device__ __forceinline__ void st_cols(float* row, int lane, const float (&v)[V]) {
  if constexpr (V == 4) *reinterpret_cast<float4*>(row + 4 * lane) = float4{v[0], v[1], v[2], v[3]};
  else if constexpr (V == 2) *reinterpret_cast<float2*>(row + 2 * lane) = float2{v[0], v[1]};
  else {
#pragma unroll
    for (int k = 0; k < V; ++k) row[V * lane + k] = v[k];
  }
}
__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <int H>
__global__ __launch_bounds__(kThreadsE) void edge_bwd_e_kernel(TrainEdgeArgs p, const float* __restrict__ z2) {
  constexpr int V = H / 64, NW = kThreadsE / 64;
  __shared__ __attribute__((aligned(16))) float sP[NW][2 * H + 4];
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const bool attention = p.head != nullptr;
  float wd[V], wd0[V], tb[3][V], aw[V];
  ld_cols<V>(p.wd, lane, wd);
  ld_cols<V>(p.wd0, lane, wd0);
#pragma unroll
  for (int y = 0; y < 3; ++y) ld_cols<V>(p.table + y * H, lane, tb[y]);
#pragma unroll
  for (int k = 0; k < V; ++k) aw[k] = 0.f;
  if (attention) ld_cols<V>(p.head, lane, aw);
  const float att_b = attention ? p.head_b[0] : 0.f;
  const float inv_norm = 1.0f / p.norm_factor;
  float pb2[V], pv1[V], ps = 0.f;
#pragma unroll
  for (int k = 0; k < V; ++k) { pb2[k] = 0.f; pv1[k] = 0.f; }

#pragma unroll 1
  for (int e = blockIdx.x * NW + w; e < p.E; e += gridDim.x * NW) {
    int r = __builtin_amdgcn_readfirstlane(p.erow[e]), c = __builtin_amdgcn_readfirstlane(p.ecol[e]);
    const float d0 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, p.ed0[e])));
    const bool active = (unsigned)r < (unsigned)p.n_nodes && (unsigned)c < (unsigned)p.n_nodes;
    float zv[V], a1[V], dz[V];
    if (!active) {          // (wave-uniform) rows of inactive list entries are zeros: the weight-gradient GEMM sums every row
#pragma unroll
      for (int k = 0; k < V; ++k) zv[k] = 0.f;
      st_cols<V>(p.a1_out + (size_t)e * H, lane, zv);
      st_cols<V>(p.dz_out + (size_t)e * H, lane, zv);
      continue;
    }
    float pv[V], qv[V], dout[V];
    ld_cols<V>(z2 + (size_t)e * H, lane, zv);
    ld_cols<V>(p.P + (size_t)r * p.ldpq, lane, pv);
    ld_cols<V>(p.Q + (size_t)c * p.ldpq, lane, qv);
    ld_cols<V>(p.d_agg + (size_t)r * H, lane, dout);
    const float ddx = p.x[3 * r] - p.x[3 * c], ddy = p.x[3 * r + 1] - p.x[3 * c + 1], ddz = p.x[3 * r + 2] - p.x[3 * c + 2];
    const float d = ddx * ddx + ddy * ddy + ddz * ddz;
    const bool rl = r < p.n_lig, cl = c < p.n_lig;
    const int ty = (rl && cl) ? 1 : ((!rl && !cl) ? 2 : 0);
    // a1 = SiLU(z1), the first layer as kernel A evaluates it
#pragma unroll
    for (int k = 0; k < V; ++k) a1[k] = silu(fmaf(d0, wd0[k], fmaf(d, wd[k], pv[k] + qv[k])) + tb[ty][k]);
    st_cols<V>(p.a1_out + (size_t)e * H, lane, a1);
    float m[V], sg[V], dot = 0.f, sdot = 0.f;
#pragma unroll
    for (int k = 0; k < V; ++k) {
      dout[k] *= inv_norm;
      sg[k] = sigmoidf_fast(zv[k]);
      m[k] = zv[k] * sg[k];
      dot = fmaf(m[k], aw[k], dot);
      sdot = fmaf(dout[k], m[k], sdot);
    }
    float att = 1.f, tt = 0.f;
    if (attention) {
      const float gate = sigmoidf_fast(wave_sum_f(dot) + att_b);
      att = gate;
      tt = wave_sum_f(sdot) * gate * (1.0f - gate);
      ps += tt;
    }
#pragma unroll
    for (int k = 0; k < V; ++k) {
      const float dm = fmaf(tt, aw[k], dout[k] * att);
      dz[k] = dm * dsilu_from(zv[k], sg[k]);
      pb2[k] += dz[k];
      pv1[k] = fmaf(tt, m[k], pv1[k]);
    }
    st_cols<V>(p.dz_out + (size_t)e * H, lane, dz);
  }
  // partial vectors of the workgroup: the 16 waves' sums in wave order -> rows 5 .. 7 of its slot
  st_cols<V>(sP[w], lane, pb2);
  st_cols<V>(sP[w] + H, lane, pv1);
  if (lane == 0) sP[w][2 * H] = ps;
  __syncthreads();
  for (int i = t; i < kPartA * H; i += kThreadsE) {
    float v = 0.f;
    if (i < 2 * H) {
#pragma unroll
      for (int q = 0; q < NW; ++q) v += sP[q][i];
    } else if (i == 2 * H) {
#pragma unroll
      for (int q = 0; q < NW; ++q) v += sP[q][2 * H];
    }
    p.part[(size_t)blockIdx.x * kPartAll * H + (size_t)kPartB * H + i] = v;
  }
}

// kernel E of the coordinate stage: phi = w3 . SiLU(z2) from the kept z2, the geometry gradients of kernel A's
// MODE_COORD epilogue (every lane evaluates the edge's scalars; lane 0 stores them), dz2 = d_phi w3 SiLU'(z2).
template <int H>
__global__ __launch_bounds__(kThreadsE) void edge_bwd_ec_kernel(TrainEdgeArgs p, const float* __restrict__ z2) {
  constexpr int V = H / 64, NW = kThreadsE / 64;
  __shared__ __attribute__((aligned(16))) float sP[NW][2 * H + 4];
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  float wd[V], wd0[V], tb[3][V], w3[V];
  ld_cols<V>(p.wd, lane, wd);
  ld_cols<V>(p.wd0, lane, wd0);
#pragma unroll
  for (int y = 0; y < 3; ++y) ld_cols<V>(p.table + y * H, lane, tb[y]);
  ld_cols<V>(p.head, lane, w3);
  const float inv_norm = 1.0f / p.norm_factor;
  float pb2[V], pv1[V];
#pragma unroll
  for (int k = 0; k < V; ++k) { pb2[k] = 0.f; pv1[k] = 0.f; }

#pragma unroll 1
  for (int e = blockIdx.x * NW + w; e < p.E; e += gridDim.x * NW) {
    const int r = __builtin_amdgcn_readfirstlane(p.erow[e]), c = __builtin_amdgcn_readfirstlane(p.ecol[e]);
    const float d0 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, p.ed0[e])));
    const bool active = (unsigned)r < (unsigned)p.n_nodes && (unsigned)c < (unsigned)p.n_nodes;
    float zv[V], a1[V], dz[V];
    if (!active) {
#pragma unroll
      for (int k = 0; k < V; ++k) zv[k] = 0.f;
      st_cols<V>(p.a1_out + (size_t)e * H, lane, zv);
      st_cols<V>(p.dz_out + (size_t)e * H, lane, zv);
      if (lane < 3) {
        p.gxr[3 * (size_t)e + lane] = 0.f; p.gxc[3 * (size_t)e + lane] = 0.f;
        if (p.gm) p.gm[3 * (size_t)e + lane] = 0.f;
      }
      continue;
    }
    float pv[V], qv[V];
    ld_cols<V>(z2 + (size_t)e * H, lane, zv);
    ld_cols<V>(p.P + (size_t)r * p.ldpq, lane, pv);
    ld_cols<V>(p.Q + (size_t)c * p.ldpq, lane, qv);
    const float xr[3] = {p.x[3 * r], p.x[3 * r + 1], p.x[3 * r + 2]}, xc[3] = {p.x[3 * c], p.x[3 * c + 1], p.x[3 * c + 2]};
    const float ddx = xr[0] - xc[0], ddy = xr[1] - xc[1], ddz = xr[2] - xc[2];
    const float d = ddx * ddx + ddy * ddy + ddz * ddz;
    const bool rl = r < p.n_lig, cl = c < p.n_lig;
    const int ty = (rl && cl) ? 1 : ((!rl && !cl) ? 2 : 0);
#pragma unroll
    for (int k = 0; k < V; ++k) a1[k] = silu(fmaf(d0, wd0[k], fmaf(d, wd[k], pv[k] + qv[k])) + tb[ty][k]);
    st_cols<V>(p.a1_out + (size_t)e * H, lane, a1);
    float m[V], sg[V], dot = 0.f;
#pragma unroll
    for (int k = 0; k < V; ++k) {
      sg[k] = sigmoidf_fast(zv[k]);
      m[k] = zv[k] * sg[k];
      dot = fmaf(m[k], w3[k], dot);
    }
    const float ph = wave_sum_f(dot);
    // d_phi and the geometry gradients (egnn_new.py:100-109, 296-316), as kernel A's MODE_COORD epilogue
    const float g0 = p.d_xagg[3 * r] * inv_norm, g1 = p.d_xagg[3 * r + 1] * inv_norm, g2 = p.d_xagg[3 * r + 2] * inv_norm;
    float T = ph, dT = 1.f;
    if (p.use_tanh) { const float th = tanhf(ph); T = th * p.coords_range; dT = (1.0f - th * th) * p.coords_range; }
    float dphi = 0.f, ar[3] = {0.f, 0.f, 0.f}, ac[3] = {0.f, 0.f, 0.f}, am[3] = {0.f, 0.f, 0.f};
    if (p.which == 0) {
      const float nrm = sqrtf(d + 1e-8f), den = nrm + p.norm_constant;
      const float ux = ddx / den, uy = ddy / den, uz = ddz / den;
      dphi = (g0 * ux + g1 * uy + g2 * uz) * dT;
      const float du0 = g0 * T, du1 = g1 * T, du2 = g2 * T;
      const float sdot = (du0 * ddx + du1 * ddy + du2 * ddz) / (nrm * den * den);
      ar[0] = du0 / den - ddx * sdot; ar[1] = du1 / den - ddy * sdot; ar[2] = du2 / den - ddz * sdot;
      ac[0] = -ar[0]; ac[1] = -ar[1]; ac[2] = -ar[2];
    } else {
      const int b = p.node_batch[r];
      const float m0 = p.mean[3 * b], m1 = p.mean[3 * b + 1], m2 = p.mean[3 * b + 2];
      const float a0 = xr[0] - m0, a1_ = xr[1] - m1, a2 = xr[2] - m2;
      const float b0 = xc[0] - m0, b1 = xc[1] - m1, b2 = xc[2] - m2;
      const float c0 = a1_ * b2 - a2 * b1, c1 = a2 * b0 - a0 * b2, c2 = a0 * b1 - a1_ * b0;
      const float cn = sqrtf(c0 * c0 + c1 * c1 + c2 * c2), cden = cn + p.norm_constant;
      dphi = (g0 * c0 + g1 * c1 + g2 * c2) / cden * dT;
      const float dc0 = g0 * T, dc1 = g1 * T, dc2 = g2 * T;
      const float sdot = cn > 0.f ? (dc0 * c0 + dc1 * c1 + dc2 * c2) / (cn * cden * cden) : 0.f;
      const float e0 = dc0 / cden - c0 * sdot, e1 = dc1 / cden - c1 * sdot, e2 = dc2 / cden - c2 * sdot;   // d cr
      ar[0] = b1 * e2 - b2 * e1; ar[1] = b2 * e0 - b0 * e2; ar[2] = b0 * e1 - b1 * e0;
      ac[0] = e1 * a2 - e2 * a1_; ac[1] = e2 * a0 - e0 * a2; ac[2] = e0 * a1_ - e1 * a0;
      am[0] = -(ar[0] + ac[0]); am[1] = -(ar[1] + ac[1]); am[2] = -(ar[2] + ac[2]);
    }
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        p.gxr[3 * (size_t)e + k] = ar[k];
        p.gxc[3 * (size_t)e + k] = ac[k];
        if (p.gm) p.gm[3 * (size_t)e + k] = am[k];
      }
    }
#pragma unroll
    for (int k = 0; k < V; ++k) {
      dz[k] = dphi * w3[k] * dsilu_from(zv[k], sg[k]);
      pb2[k] += dz[k];
      pv1[k] = fmaf(dphi, m[k], pv1[k]);
    }
    st_cols<V>(p.dz_out + (size_t)e * H, lane, dz);
  }
  st_cols<V>(sP[w], lane, pb2);
  st_cols<V>(sP[w] + H, lane, pv1);
  __syncthreads();
  for (int i = t; i < kPartA * H; i += kThreadsE) {
    float v = 0.f;
    if (i < 2 * H) {
#pragma unroll
      for (int q = 0; q < NW; ++q) v += sP[q][i];
    }
    p.part[(size_t)blockIdx.x * kPartAll * H + (size_t)kPartB * H + i] = v;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// kernel B
template <int H>
__global__ __launch_bounds__(kThreads, 1) void edge_bwd_b_kernel(TrainEdgeArgs p) {
  constexpr int BK = 32, CT = H / 32, NK = H / BK;
  __shared__ __attribute__((aligned(16))) float sB[2 * BK * H];
  __shared__ __attribute__((aligned(16))) float sV[5 * H];
  __shared__ __attribute__((aligned(16))) float sS[4][5 * 32];
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int half = lane >> 5, j = lane & 31;
  int* s_row = reinterpret_cast<int*>(sS[w]);
  int* s_col = reinterpret_cast<int*>(sS[w] + 32);
  float* s_d = sS[w] + 64;
  float* s_d0 = sS[w] + 96;
  int* s_ty = reinterpret_cast<int*>(sS[w] + 128);

  for (int i = t; i < H; i += kThreads) {
    sV[i] = p.wd[i];
    sV[H + i] = p.wd0[i];
    sV[2 * H + i] = p.table[i];
    sV[3 * H + i] = p.table[H + i];
    sV[4 * H + i] = p.table[2 * H + i];
  }
  const int ntiles = (p.E + 127) / 128;
  float pwd[CT], pwd0[CT], pt0[CT], pt1[CT], pt2[CT];
#pragma unroll
  for (int c = 0; c < CT; ++c) { pwd[c] = 0.f; pwd0[c] = 0.f; pt0[c] = 0.f; pt1[c] = 0.f; pt2[c] = 0.f; }

  SliceStream<H> st;
  st.load(p.Bmat, 0, t);
  st.store(sB, t);
  __syncthreads();
  int bslice = 0;
  float wdv[CT], wd0v[CT];
#pragma unroll
  for (int c = 0; c < CT; ++c) { wdv[c] = sV[32 * c + j]; wd0v[c] = sV[H + 32 * c + j]; }

#pragma unroll 1
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int e_base = tile * 128 + w * 32;
    const int e = e_base + j;
    bool active = e < p.E;
    int r = 0, cidx = 0;
    float d0 = 0.f;
    if (active) {
      r = p.erow[e]; cidx = p.ecol[e]; d0 = p.ed0[e];
      if ((unsigned)r >= (unsigned)p.n_nodes || (unsigned)cidx >= (unsigned)p.n_nodes) { active = false; r = 0; cidx = 0; d0 = 0.f; }
    }
    float d = 0.f;
    if (active) {
      const float ddx = p.x[3 * r] - p.x[3 * cidx], ddy = p.x[3 * r + 1] - p.x[3 * cidx + 1], ddz = p.x[3 * r + 2] - p.x[3 * cidx + 2];
      d = ddx * ddx + ddy * ddy + ddz * ddz;
    }
    const bool rl = r < p.n_lig, cl = cidx < p.n_lig;
    const int ty = (rl && cl) ? 1 : ((!rl && !cl) ? 2 : 0);
    if (half == 0) { s_row[j] = active ? r : -1; s_col[j] = cidx; s_d[j] = d; s_d0[j] = d0; s_ty[j] = ty; }

    f32x16 acc[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[c][q] = 0.f;
    const float* Ap = p.dz_in + (size_t)(e < p.E ? e : 0) * H + 4 * half;
    f32x4 ac4 = ldv4(Ap), an4 = ac4;

#pragma unroll 1
    for (int kt = 0; kt < NK; ++kt) {
      st.load(p.Bmat, kt + 1 < NK ? kt + 1 : 0, t);
      const float* bcur = sB + (bslice & 1) * BK * H + (4 * half) * H + j;
#pragma unroll
      for (int g = 0; g < BK / 8; ++g) {
        const int kb = kt * BK + 8 * g;
        if (g + 1 < BK / 8 || kt + 1 < NK) an4 = ldv4(Ap + kb + 8);
        const float a[4] = {ac4.x, ac4.y, ac4.z, ac4.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float* brow = bcur + (8 * g + i) * H;
#pragma unroll
          for (int c = 0; c < CT; ++c) acc[c] = mfma32(a[i], brow[32 * c], acc[c]);
        }
        ac4 = an4;
      }
      st.store(sB + ((bslice + 1) & 1) * BK * H, t);
      ++bslice;
      __syncthreads();
    }

    // ================= epilogue: dz1 = da1 * SiLU'(z1), z1 recomputed in accumulator layout =================
    wave_lds_fence();
    float pg[16], pg0[16];
    // the row-gathered P / Q values of accumulator row q are requested one row AHEAD of their use (round 6: 16 dependent
    // load batches per tile left the waves waiting; same arithmetic, same bits)
    float pv[CT], qv[CT], pvn[CT], qvn[CT];
    auto load_pq = [&](int q, float (&pd)[CT], float (&qd)[CT]) {
      const int m = mfma_row(q, lane);
      const int rowm = s_row[m], colm = s_col[m];
      const float* Pr = p.P + (size_t)(rowm >= 0 ? rowm : 0) * p.ldpq + j;
      const float* Qr = p.Q + (size_t)colm * p.ldpq + j;
#pragma unroll
      for (int c = 0; c < CT; ++c) { pd[c] = Pr[32 * c]; qd[c] = Qr[32 * c]; }
    };
    load_pq(0, pv, qv);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      if (q + 1 < 16) load_pq(q + 1, pvn, qvn);
      const int m = mfma_row(q, lane);
      const int rowm = s_row[m], tym = s_ty[m];
      const float dm = s_d[m], d0m = s_d0[m];
      const bool valid = rowm >= 0;
      const float* tb = sV + (2 + tym) * H + j;
      float gdp = 0.f, gd0p = 0.f;
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        const float z1 = fmaf(d0m, wd0v[c], fmaf(dm, wdv[c], pv[c] + qv[c])) + tb[32 * c];
        const float sg = sigmoidf_fast(z1);
        const float dz = valid ? acc[c][q] * dsilu_from(z1, sg) : 0.f;
        acc[c][q] = dz;
        pwd[c] = fmaf(dm, dz, pwd[c]);
        pwd0[c] = fmaf(d0m, dz, pwd0[c]);
        pt0[c] += tym == 0 ? dz : 0.f;
        pt1[c] += tym == 1 ? dz : 0.f;
        pt2[c] += tym == 2 ? dz : 0.f;
        gdp = fmaf(dz, wdv[c], gdp);
        gd0p = fmaf(dz, wd0v[c], gd0p);
      }
      pg[q] = gdp; pg0[q] = gd0p;
#pragma unroll
      for (int c = 0; c < CT; ++c) { pv[c] = pvn[c]; qv[c] = qvn[c]; }
      __builtin_amdgcn_sched_barrier(0);
    }
    const float gdt = reduce16_half_wave(pg, j);
    const float gd0t = reduce16_half_wave(pg0, j);
    {
      const int em = e_base + mfma_row(j >> 1, lane);
      if (!(j & 1) && em < p.E) { p.gd[em] = gdt; p.gd0[em] = gd0t; }
    }
#pragma unroll
    for (int c = 0; c < CT; ++c) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int em = e_base + mfma_row(q, lane);
        if (em < p.E) p.dz_out[(size_t)em * H + 32 * c + j] = acc[c][q];
      }
    }
    wave_lds_fence();
  }
  __syncthreads();
  float* sp = sB + (size_t)((w * 2 + half) * kPartB) * H;
#pragma unroll
  for (int c = 0; c < CT; ++c) {
    sp[32 * c + j] = pwd[c];
    sp[H + 32 * c + j] = pwd0[c];
    sp[2 * H + 32 * c + j] = pt0[c];
    sp[3 * H + 32 * c + j] = pt1[c];
    sp[4 * H + 32 * c + j] = pt2[c];
  }
  __syncthreads();
  for (int i = t; i < kPartB * H; i += kThreads) {
    float v = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) v += sB[(size_t)q * kPartB * H + i];
    p.part[(size_t)blockIdx.x * kPartAll * H + i] = v;                           // rows 0 .. 4 of the workgroup's [8][H] slot
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// rev[e] = index of the edge (col_e, row_e); -1 when absent (never for a radius graph) or for an inactive entry
__global__ void edge_rev_kernel(const int* erow, const int* ecol, const int* row_ptr, const int* deg, int E, int N, int* rev) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const int r = erow[e], c = ecol[e];
  int out = -1;
  if ((unsigned)r < (unsigned)N && (unsigned)c < (unsigned)N) {
    int lo = row_ptr[c], hi = lo + deg[c] - 1;
    while (lo <= hi) {
      const int mid = (lo + hi) >> 1;
      const int v = ecol[mid];
      if (v == r) { out = mid; break; }
      if (v < r) lo = mid + 1; else hi = mid - 1;
    }
  }
  rev[e] = out;
}

// dP[i] = sum_{e in row i, e < e_lim} dz[e],  dQ[i] = sum_{e in row i, rev(e) in [0, e_lim)} dz[rev(e)]; one wave per row
__global__ __launch_bounds__(kThreads) void rows_gather_kernel(const float* dz, int H, const int* row_ptr, const int* deg,
                                                               const int* rev, int e_lim, int n_rows, float* dP, float* dQ,
                                                               int ldo) {
  const int row = (blockIdx.x * kThreads + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (row >= n_rows) return;
  const int s = row_ptr[row], dg = deg[row];
  for (int k = 4 * lane; k < H; k += 256) {
    float4 sp = make_float4(0.f, 0.f, 0.f, 0.f), sq = sp;
    for (int e = s; e < s + dg; ++e) {
      if (e < e_lim) {
        const float4 v = ld4(dz + (size_t)e * H + k);
        sp.x += v.x; sp.y += v.y; sp.z += v.z; sp.w += v.w;
      }
      const int re = rev[e];
      if (re >= 0 && re < e_lim) {
        const float4 v = ld4(dz + (size_t)re * H + k);
        sq.x += v.x; sq.y += v.y; sq.z += v.z; sq.w += v.w;
      }
    }
    *reinterpret_cast<float4*>(dP + (size_t)row * ldo + k) = sp;
    *reinterpret_cast<float4*>(dQ + (size_t)row * ldo + k) = sq;
  }
}

// d_x[i] (+)= sum over row i's edges e = (i, j) of  [gxr[e] + 2 gd[e] (x_i - x_j)]  (e < e_lim)
//                                               + [gxc[re] + 2 gd[re] (x_i - x_j)]  (re = rev(e) < e_lim)
// (|x_i - x_j|^2 depends on both endpoints; an edge's column-node gradient reaches x_i through its reverse edge)
__global__ __launch_bounds__(kThreads) void edge_to_node3_kernel(const float* gd, const float* gxr, const float* gxc,
                                                                 const float* x, const int* ecol, const int* row_ptr,
                                                                 const int* deg, const int* rev, int e_lim, int n_rows,
                                                                 float* dx, int accumulate) {
  // one wave per node: lane l takes the row's edges l, l + 64, ...; the 64 partial sums meet in a fixed butterfly
  const int i = (blockIdx.x * kThreads + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (i >= n_rows) return;
  const int s = row_ptr[i], dg = deg[i];
  const float xi0 = x[3 * i], xi1 = x[3 * i + 1], xi2 = x[3 * i + 2];
  float a0 = 0.f, a1 = 0.f, a2 = 0.f;
  for (int e = s + lane; e < s + dg; e += 64) {
    const int jn = ecol[e];
    const float f0 = xi0 - x[3 * jn], f1 = xi1 - x[3 * jn + 1], f2 = xi2 - x[3 * jn + 2];
    if (e < e_lim) {
      const float g = gd ? 2.0f * gd[e] : 0.f;
      a0 += g * f0; a1 += g * f1; a2 += g * f2;
      if (gxr) { a0 += gxr[3 * (size_t)e]; a1 += gxr[3 * (size_t)e + 1]; a2 += gxr[3 * (size_t)e + 2]; }
    }
    const int re = rev[e];
    if (re >= 0 && re < e_lim) {
      const float g = gd ? 2.0f * gd[re] : 0.f;
      a0 += g * f0; a1 += g * f1; a2 += g * f2;
      if (gxc) { a0 += gxc[3 * (size_t)re]; a1 += gxc[3 * (size_t)re + 1]; a2 += gxc[3 * (size_t)re + 2]; }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    a0 += __shfl_xor(a0, o, 64); a1 += __shfl_xor(a1, o, 64); a2 += __shfl_xor(a2, o, 64);
  }
  if (lane == 0) {
    if (accumulate) { dx[3 * i] += a0; dx[3 * i + 1] += a1; dx[3 * i + 2] += a2; }
    else { dx[3 * i] = a0; dx[3 * i + 1] = a1; dx[3 * i + 2] = a2; }
  }
}

// out[b][0..2] = sum of v[e][0..2] over the edges (e < e_lim) whose row belongs to sample b; one workgroup per sample
__global__ __launch_bounds__(kThreads) void sample_edge_sum3_kernel(const float* v, const int* row_ptr, const int* lig_off,
                                                                    const int* poc_off, int n_lig, int e_lim, float* out) {
  __shared__ float red[3][kThreads];
  const int b = blockIdx.x, t = threadIdx.x;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f;
  for (int seg = 0; seg < 2; ++seg) {
    const int n0 = seg ? n_lig + poc_off[b] : lig_off[b], n1 = seg ? n_lig + poc_off[b + 1] : lig_off[b + 1];
    const int e0 = row_ptr[n0], e1 = min(row_ptr[n1], e_lim);
    for (int e = e0 + t; e < e1; e += kThreads) { a0 += v[3 * (size_t)e]; a1 += v[3 * (size_t)e + 1]; a2 += v[3 * (size_t)e + 2]; }
  }
  red[0][t] = a0; red[1][t] = a1; red[2][t] = a2;
  __syncthreads();
  for (int s = kThreads / 2; s > 0; s >>= 1) {
    if (t < s) { red[0][t] += red[0][t + s]; red[1][t] += red[1][t + s]; red[2][t] += red[2][t + s]; }
    __syncthreads();
  }
  if (t < 3) out[3 * b + t] = red[t][0];
}

// out[g][i] = sum over the parts p in [g gs, min((g + 1) gs, n_part)) of part[p * stride + i], i < width; fixed order
__global__ void partial_reduce_kernel(const float* part, int n_part, size_t stride, int width, int gs, float* out,
                                      size_t out_stride) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= width) return;
  const int g = blockIdx.y;
  const int p0 = g * gs, p1 = min(p0 + gs, n_part);
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  int pi = p0;
  for (; pi + 4 <= p1; pi += 4) {
    a0 += part[(size_t)pi * stride + i];
    a1 += part[(size_t)(pi + 1) * stride + i];
    a2 += part[(size_t)(pi + 2) * stride + i];
    a3 += part[(size_t)(pi + 3) * stride + i];
  }
  for (; pi < p1; ++pi) a0 += part[(size_t)pi * stride + i];
  out[(size_t)g * out_stride + i] = (a0 + a1) + (a2 + a3);
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight gradient:  Cp[z][m][n] = sum over k in chunk z of A[k][m] B[k][n]   (A [K][lda], B [K][ldb]); the chunks are
// added by partial_reduce_kernel.  Workgroup tile 128 x 128, wave tile 64 x 64 (2 x 2 MFMA tiles of 32 x 32).  The operands
// go through LDS in slabs of 16 k rows (16-byte coalesced loads one slab ahead in registers, two LDS buffers, one barrier per
// slab); a wave reads its A / B words with ds_read_b32 (lane = column, the two half-waves one k row apart: the row stride of
// 160 words puts them on disjoint banks).  First version (operands straight from global memory, one dword load per lane
// and MFMA): 43 - 48 TFLOP/s on the edge-level gradients (profiles/r4m_wgrad_sweep.md).
struct WgradArgs { const float* A; int lda; const float* B; int ldb; int K; int M; int N; float* Cp; int kc; };

constexpr int kWgBK = 16, kWgLd = 160;

__device__ __forceinline__ void wgrad_slab_load(const float* X, int ld, int K1, int k, int c0, int C, bool vec, int t,
                                                f32x4 (&r)[2]) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int q = t + kThreads * i, row = q >> 5, col = c0 + 4 * (q & 31);
    const int kk = k + row;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (kk < K1) {
      const float* src = X + (size_t)kk * ld + col;
      if (vec && col + 3 < C) v = ldv4(src);
      else {
        if (col < C) v[0] = src[0];
        if (col + 1 < C) v[1] = src[1];
        if (col + 2 < C) v[2] = src[2];
        if (col + 3 < C) v[3] = src[3];
      }
    }
    r[i] = v;
  }
}

__device__ __forceinline__ void wgrad_slab_store(float* buf, int t, const f32x4 (&r)[2]) {
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int q = t + kThreads * i;
    *reinterpret_cast<f32x4*>(buf + (q >> 5) * kWgLd + 4 * (q & 31)) = r[i];
  }
}

__global__ __launch_bounds__(kThreads) void wgrad_kernel(WgradArgs p) {
  __shared__ __attribute__((aligned(16))) float sA[2][kWgBK * kWgLd];
  __shared__ __attribute__((aligned(16))) float sB[2][kWgBK * kWgLd];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int wm = w >> 1, wn = w & 1, j = lane & 31, kh = lane >> 5;
  const int m0 = blockIdx.x * 128, n0 = blockIdx.y * 128;
  const int k0 = blockIdx.z * p.kc, k1 = min(p.K, k0 + p.kc);
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[a][b][q] = 0.f;
  if (k0 < k1) {
    const bool va = (p.lda & 3) == 0 && (reinterpret_cast<uintptr_t>(p.A) & 15) == 0;
    const bool vb = (p.ldb & 3) == 0 && (reinterpret_cast<uintptr_t>(p.B) & 15) == 0;
    f32x4 ra[2], rb[2];
    wgrad_slab_load(p.A, p.lda, k1, k0, m0, p.M, va, t, ra);
    wgrad_slab_load(p.B, p.ldb, k1, k0, n0, p.N, vb, t, rb);
    wgrad_slab_store(sA[0], t, ra);
    wgrad_slab_store(sB[0], t, rb);
    __syncthreads();
    int cur = 0;
    const int oa = kh * kWgLd + wm * 64 + j, ob = kh * kWgLd + wn * 64 + j;
#pragma unroll 1
    for (int k = k0; k < k1; k += kWgBK) {
      const bool more = k + kWgBK < k1;
      if (more) {
        wgrad_slab_load(p.A, p.lda, k1, k + kWgBK, m0, p.M, va, t, ra);
        wgrad_slab_load(p.B, p.ldb, k1, k + kWgBK, n0, p.N, vb, t, rb);
      }
      const float* a_ = sA[cur] + oa;
      const float* b_ = sB[cur] + ob;
#pragma unroll
      for (int s = 0; s < kWgBK / 2; ++s) {
        const float a0 = a_[2 * s * kWgLd], a1 = a_[2 * s * kWgLd + 32];
        const float b0 = b_[2 * s * kWgLd], b1 = b_[2 * s * kWgLd + 32];
        acc[0][0] = mfma32(a0, b0, acc[0][0]);
        acc[0][1] = mfma32(a0, b1, acc[0][1]);
        acc[1][0] = mfma32(a1, b0, acc[1][0]);
        acc[1][1] = mfma32(a1, b1, acc[1][1]);
      }
      if (more) {
        wgrad_slab_store(sA[cur ^ 1], t, ra);
        wgrad_slab_store(sB[cur ^ 1], t, rb);
      }
      __syncthreads();
      cur ^= 1;
    }
  }
  float* C = p.Cp + (size_t)blockIdx.z * p.M * p.N;
  const int mb = m0 + wm * 64, nb = n0 + wn * 64;
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int m = mb + 32 * a + mfma_row(q, lane), n = nb + 32 * b + j;
        if (m < p.M && n < p.N) C[(size_t)m * p.N + n] = acc[a][b][q];
      }
}

}  // namespace dsbdd
