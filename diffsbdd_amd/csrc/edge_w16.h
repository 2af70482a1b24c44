// Fourth structure of the fused edge-MLP kernels (same math and arguments as
// edge_mlp.h / edge_wave.h): the wave tile is 16 edges x ALL H features on
// v_mfma_f32_16x16x4_f32 instead of 32 edges on v_mfma_f32_32x32x2_f32.
//
// Why: the ablation ladder of edge_wave.h (profiles/README.md) shows a pure MFMA
// loop at 0.340 ms and the complete kernel at 0.430 ms -- everything in between
// (epilogue 0.040, B-operand reads 0.017, in-loop SiLU 0.015, LDS-DMA 0.011,
// barriers, gathers) is time in which one of only TWO waves per SIMD has no MFMA
// to issue.  A 32-edge tile needs 128 accumulator registers, which caps the
// occupancy at 2.  A 16-edge tile needs 64: four waves per SIMD fit in the
// register file, every A element is still evaluated exactly once
// (lane l = edge l & 15, k quarter l >> 4), and a 16-lane DPP row holds a complete
// edge row, so the attention / head reductions are 4-step row butterflies.
//
//   * workgroup = 8 waves = 128 edges x 2 per CU (H <= 128) or 12 waves = 192 edges x 1 per CU
//     (H >= 192) = 4 resp. 3 waves per SIMD; same persistent XCD-contiguous schedule as edge_wave.h;
//   * lane (e = l & 15, kq = l >> 4) takes k in {16g + 4kq .. 16g + 4kq + 3} of every
//     group of 16; MFMA step i pairs it with B rows 16g + 4kq + i;
//   * W2^T slices (32 rows) by global_load_lds, ONE row per DMA instruction so that
//     the LDS rows can be padded to H + 4 floats: the rows of the two k quarters of a
//     32-lane LDS group then sit 16 banks apart (conflict-free ds_read2_b32);
//   * accumulator layout: register r of lane (e, kq) = edge 4kq + r, column 16t + e.
//     The segmented row sums walk the 16 edges in order and hand the running sums
//     from lane group kq to kq + 1 (three baton passes).
//   * MODE_COORD with two MLPs always runs one workgroup per (tile, MLP).
#pragma once
#include "common.h"
#include "edge_mlp.h"
#include "edge_wave.h"

namespace dsbdd {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// waves per workgroup: 64 accumulator registers (H = 256) plus the K-loop state do not fit the
// 128 registers of 4 waves per SIMD without spilling, so the wide models run 3 waves per SIMD
// (170 registers) as ONE 12-wave workgroup per CU (192-edge tiles; a 6-wave workgroup would put
// 2+2+1+1 waves on the SIMDs and two of them do not pack into 3 per SIMD); H <= 128 runs
// 8 waves x 2 workgroups per CU (4 per SIMD)
template <int H> struct W16Waves {
  static constexpr int value = H > 128 ? 12 : 8;
  static constexpr int per_simd = H > 128 ? 3 : 4;
  static constexpr int wg_per_cu = H > 128 ? 1 : 2;
};

template <int H, int MODE>
struct W16Layout {
  static constexpr int BK = 32;
  static constexpr int LDB = H + 4;
  static constexpr int B_BUF = BK * LDB;
  static constexpr int VEC_PER = 7 * H;
  static constexpr int VEC_OFF = 2 * B_BUF;
  static constexpr int SCR_OFF = VEC_OFF + VEC_PER;
  static constexpr int SCR_PER = 16 + 16 * 3;              // per wave: phi[16], trans[16][3]
  static constexpr int WV = W16Waves<H>::value;
  static constexpr int TOTAL = SCR_OFF + WV * SCR_PER;
};

template <int H, int MODE>
__global__ __launch_bounds__(64 * W16Waves<H>::value, W16Waves<H>::per_simd) void edge_w16_kernel(EdgeArgs p) {
  using L = W16Layout<H, MODE>;
  constexpr int BK = L::BK, LDB = L::LDB, WV = L::WV, NT = 64 * WV;
  constexpr int CT = H / 16;            // 16-col MFMA tiles per wave (all features)
  constexpr int NK = H / BK;            // K slices per tile
  constexpr int BMB = 16 * WV;          // edges per workgroup tile

  __shared__ float smem[L::TOTAL];
  float* sB = smem;                         // [2][BK][LDB]
  float* sV = smem + L::VEC_OFF;            // wd, wd0, tab0..2, b2, w-out of this workgroup's MLP
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int kq = lane >> 4, n = lane & 15;
  float* s_phi = smem + L::SCR_OFF + w * L::SCR_PER;   // [16]
  float* s_tr = s_phi + 16;                             // [16][3]

  const bool two = MODE == MODE_COORD && p.n_mlp == 2;
  const int qsel = two ? ((blockIdx.x >> 3) & 1) : 0;     // the MLP this workgroup evaluates
  const EdgeMlpW& mw = p.mlp[qsel];
  for (int i = t; i < H; i += NT) {
    sV[i] = mw.wd[i];
    sV[H + i] = mw.wd0[i];
    sV[2 * H + i] = mw.table[i];
    sV[3 * H + i] = mw.table[H + i];
    sV[4 * H + i] = mw.table[2 * H + i];
    sV[5 * H + i] = mw.b2[i];
    sV[6 * H + i] = (MODE == MODE_GCL) ? (p.attention ? p.att_w[i] : 0.f) : p.w3[i];
  }
  const float att_b = (MODE == MODE_GCL && p.attention) ? p.att_b[0] : 0.f;
  const float inv_norm = 1.0f / p.norm_factor;

  const int E = *p.e_count;
  const int ntiles = (E + BMB - 1) / BMB;
  const int xcd = blockIdx.x & 7;
  const int kx = two ? (blockIdx.x >> 4) : (blockIdx.x >> 3);
  const int gx = two ? (gridDim.x >> 4) : (gridDim.x >> 3);
  const int tq = ntiles / 8, tr = ntiles % 8;
  const int csize = tq + (xcd < tr ? 1 : 0);
  const int cbase = (xcd < tr) ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
  if (kx >= csize) return;

  // W2^T slice -> LDS: wave w brings rows w, w + WV, ...; one (padded) row per DMA instruction
  auto streamB = [&](int ks, int buf) {
    if (lane < H / 4) {
#pragma unroll
      for (int i = 0; i < (BK + WV - 1) / WV; ++i) {
        const int r = w + WV * i;
        if (r >= BK) break;                                            // wave-uniform
        const float* src = mw.W2T + (size_t)(ks * BK + r) * H + lane * 4;
        float* dst = sB + buf * L::B_BUF + r * LDB;                   // wave-uniform
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
      }
    }
  };

  // this lane's edge.  Only what the K loop needs stays in registers (4 waves per SIMD =
  // 128 VGPRs): squared length, d0, the type's table row, the row id; the P/Q rows as
  // 32-bit element offsets from the (scalar) matrix bases.
  int my_r = -1, my_c = 0, ty_off = 2 * H;
  float my_d = 0.f, my_d0 = 0.f;
  unsigned offP = 4 * kq, offQ = 4 * kq;
  int nx_r = -1, nx_c = 0;
  float nx_d0 = 0.f;
  auto fetch_idx = [&](int tile) {
    const int e = tile * BMB + w * 16 + n;
    nx_r = -1; nx_c = 0; nx_d0 = 0.f;
    if (e < E) { nx_r = p.erow[e]; nx_c = p.ecol[e]; nx_d0 = p.ed0[e]; }
  };
  auto commit_edge = [&]() {                               // (nx_r, nx_c, nx_d0) -> current edge
    my_r = nx_r; my_c = nx_c; my_d0 = nx_d0; my_d = 0.f; ty_off = 2 * H;
    offP = (unsigned)(my_r < 0 ? 0 : my_r) * (unsigned)p.ldpq + 4 * kq;
    offQ = (unsigned)my_c * (unsigned)p.ldpq + 4 * kq;
    if (my_r >= 0) {
      const float dx = p.x[3 * my_r] - p.x[3 * my_c], dy = p.x[3 * my_r + 1] - p.x[3 * my_c + 1],
                  dz = p.x[3 * my_r + 2] - p.x[3 * my_c + 2];
      my_d = dx * dx + dy * dy + dz * dz;                  // coord2diff radial, egnn_new.py:298-299
      const bool rl = my_r < p.n_lig, cl = my_c < p.n_lig;
      ty_off = (2 + ((rl && cl) ? 1 : ((!rl && !cl) ? 2 : 0))) * H;   // dynamics.py:119-124
    }
  };

  streamB(0, 0);
  fetch_idx(cbase + kx);
  commit_edge();
  __syncthreads();          // sV + slice 0 visible
  int bslice = 0;

  const float* __restrict__ Pm = mw.P;
  const float* __restrict__ Qm = mw.Q;
  float4 pc = ld4(Pm + offP), qc = ld4(Qm + offQ);

  const int my_tiles = (csize - kx + gx - 1) / gx;
  int li = kx;
#pragma unroll 1
  for (int u = 0; u < my_tiles; ++u) {
    const bool last_unit = u + 1 == my_tiles;

    f32x4 acc[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[c][r] = 0.f;

#pragma unroll 1
    for (int kt = 0; kt < NK; ++kt) {
      const bool more = kt + 1 < NK;
      if (more) streamB(kt + 1, (bslice + 1) & 1);
      else if (!last_unit) streamB(0, (bslice + 1) & 1);      // continuous stream across tiles
      if (!last_unit && kt == 0) fetch_idx(cbase + li + gx);   // next tile's edge ids
      const float* bcur = sB + (bslice & 1) * L::B_BUF + (4 * kq) * LDB + n;
#pragma unroll
      for (int g = 0; g < BK / 16; ++g) {
        const int kb = kt * BK + 16 * g;                   // this lane's k = kb + 4*kq + i
        const float4 wd4 = *reinterpret_cast<const float4*>(sV + kb + 4 * kq);
        const float4 wz4 = *reinterpret_cast<const float4*>(sV + H + kb + 4 * kq);
        const float4 tb4 = *reinterpret_cast<const float4*>(sV + ty_off + kb + 4 * kq);
        float a[4];
        a[0] = silu(pc.x + qc.x + my_d * wd4.x + my_d0 * wz4.x + tb4.x);
        a[1] = silu(pc.y + qc.y + my_d * wd4.y + my_d0 * wz4.y + tb4.y);
        a[2] = silu(pc.z + qc.z + my_d * wd4.z + my_d0 * wz4.z + tb4.z);
        a[3] = silu(pc.w + qc.w + my_d * wd4.w + my_d0 * wz4.w + tb4.w);
        if (g + 1 < BK / 16 || more) {     // next group's P/Q chunk into the registers just consumed:
          pc = ld4(Pm + offP + kb + 16);   // it has this group's 64 MFMAs to land
          qc = ld4(Qm + offQ + kb + 16);
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float* brow = bcur + (16 * g + i) * LDB;
#pragma unroll
          for (int c = 0; c < CT; ++c)
#ifdef DSBDD_DIAG_NOBREAD
            acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], a[(i + c) & 3], acc[c], 0, 0, 0);   // DIAGNOSTIC ONLY
#else
            acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], brow[c * 16], acc[c], 0, 0, 0);
#endif
        }
        __builtin_amdgcn_s_setprio(0);
      }
      ++bslice;
      __syncthreads();
    }

    // ================= wave-private epilogue =================
    // accumulator register r of lane (n, kq) = edge 4*kq + r, feature 16*c + n
#ifdef DSBDD_DIAG_NOEPI
    if (MODE == MODE_GCL) {   // DIAGNOSTIC ONLY
      float tot = 0.f;
#pragma unroll
      for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) tot += acc[c][r];
      if (tot == 12345.678f) p.agg[lane] = tot;
    } else
#endif
    if (MODE == MODE_GCL) {
      float att[4] = {1.f, 1.f, 1.f, 1.f};
      float part[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < CT; ++c) {                       // m = SiLU(acc + b2)   (egnn_new.py:18-19)
        const float bv = sV[5 * H + c * 16 + n], aw = sV[6 * H + c * 16 + n];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          acc[c][r] = silu(acc[c][r] + bv);
          part[r] += acc[c][r] * aw;
        }
      }
      if (p.attention) {   // att = sigmoid(w_a . m + b_a); a 16-lane row holds complete edge rows
#pragma unroll
        for (int o = 1; o < 16; o <<= 1)
#pragma unroll
          for (int r = 0; r < 4; ++r) part[r] += __shfl_xor(part[r], o);
#pragma unroll
        for (int r = 0; r < 4; ++r) att[r] = sigmoidf_fast(part[r] + att_b);
      }
      // segmented sums of m * att (egnn_new.py:40, 52-54) in edge order
      float sum[CT];
#pragma unroll
      for (int c = 0; c < CT; ++c) sum[c] = 0.f;
      int cur = -1;
      auto flush = [&](int owner) {                        // the owning lane group holds the sums
        if (cur >= 0 && kq == owner) {
          float* dst = p.agg + (size_t)cur * H + n;
#pragma unroll
          for (int c = 0; c < CT; ++c) unsafeAtomicAdd(dst + c * 16, sum[c] * inv_norm);
        }
      };
#pragma unroll
      for (int gb = 0; gb < 4; ++gb) {
        if (gb > 0) {                                      // baton: running sums move one lane group up
#pragma unroll
          for (int c = 0; c < CT; ++c) {
            const float other = __shfl_up(sum[c], 16);
            if (kq == gb) sum[c] = other;
          }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int rn = __builtin_amdgcn_readlane(my_r, 4 * gb + i);
          if (rn != cur) {                                 // scalar compare / branch
            flush(gb);
            cur = rn;
#pragma unroll
            for (int c = 0; c < CT; ++c) sum[c] = 0.f;
          }
          if (kq == gb) {
#pragma unroll
            for (int c = 0; c < CT; ++c) sum[c] = fmaf(acc[c][i], att[i], sum[c]);
          }
        }
      }
      flush(3);
    } else {
      // scalar head: phi = w3 . SiLU(acc + b2)   (egnn_new.py:80-92)
      float part[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        const float bv = sV[5 * H + c * 16 + n], wv = sV[6 * H + c * 16 + n];
#pragma unroll
        for (int r = 0; r < 4; ++r) part[r] += silu(acc[c][r] + bv) * wv;
      }
#pragma unroll
      for (int o = 1; o < 16; o <<= 1)
#pragma unroll
        for (int r = 0; r < 4; ++r) part[r] += __shfl_xor(part[r], o);
      if (n == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) s_phi[4 * kq + r] = part[r];
      }
      wave_lds_fence();
      const float ph = s_phi[n];                            // this lane's edge
      wave_lds_fence();

      // this workgroup's term of trans = u*phi + cross*phi_x   (egnn_new.py:100-109, 296-316)
      float tx = 0.f, ty = 0.f, tz = 0.f;
      if (my_r >= 0) {
        const float xr[3] = {p.x[3 * my_r], p.x[3 * my_r + 1], p.x[3 * my_r + 2]};
        const float xc[3] = {p.x[3 * my_c], p.x[3 * my_c + 1], p.x[3 * my_c + 2]};
        const float f = p.use_tanh ? tanhf(ph) * p.coords_range : ph;
        if (qsel == 0) {
          const float dx = xr[0] - xc[0], dy = xr[1] - xc[1], dz = xr[2] - xc[2];
          const float den = sqrtf(my_d + 1e-8f) + p.norm_constant;
          tx = dx / den * f; ty = dy / den * f; tz = dz / den * f;
        } else {
          const int b = p.node_batch[my_r];
          const float m0 = p.mean[3 * b], m1 = p.mean[3 * b + 1], m2 = p.mean[3 * b + 2];
          const float a0 = xr[0] - m0, a1 = xr[1] - m1, a2 = xr[2] - m2;
          const float b0 = xc[0] - m0, b1 = xc[1] - m1, b2 = xc[2] - m2;
          const float c0 = a1 * b2 - a2 * b1, c1 = a2 * b0 - a0 * b2, c2 = a0 * b1 - a1 * b0;
          const float cden = sqrtf(c0 * c0 + c1 * c1 + c2 * c2) + p.norm_constant;
          tx = c0 / cden * f; ty = c1 / cden * f; tz = c2 / cden * f;
        }
      }
      if (kq == 0) { s_tr[3 * n] = tx; s_tr[3 * n + 1] = ty; s_tr[3 * n + 2] = tz; }
      wave_lds_fence();
      if (lane < 3) {
        int cur = -1;
        float sum = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int rn = __builtin_amdgcn_readlane(my_r, e);
          if (rn != cur) {
            if (cur >= 0) unsafeAtomicAdd(&p.xagg[(size_t)cur * 3 + lane], sum / p.norm_factor);
            cur = rn;
            sum = 0.f;
          }
          sum += s_tr[3 * e + lane];
        }
        if (cur >= 0) unsafeAtomicAdd(&p.xagg[(size_t)cur * 3 + lane], sum / p.norm_factor);
      }
      wave_lds_fence();   // scratch is reused by the next tile
    }

    if (!last_unit) {       // next tile: edge state and first P/Q chunk (the other 3 waves of the SIMD cover the latency)
      commit_edge();
      pc = ld4(Pm + offP); qc = ld4(Qm + offQ);
    }
    li += gx;
  }  // tiles
}

}  // namespace dsbdd
