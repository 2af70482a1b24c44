// 16-edge-granule variant of the fused edge-MLP kernels (same math, arguments and aggregation protocol as edge_wave.h /
// edge_mlp.h -- see there for the algorithm and the reference citations, egnn_new.py:31-52,96-122).
//
// Why: a wave of edge_wave.h owns 32 edges x all H features = 1024 v_mfma_f32_32x32x2_f32 per K loop, 27 us of matrix
// time; a launch whose tiles do not fill the SIMDs a whole number of times pays up to one such unit for the remainder
// (the radius-1 message launches: 3.18 -> 4 wave tiles per SIMD; the C-alpha coordinate stage: 274 (tile, MLP) items on
// 256 CUs).  Here a wave owns 16 edges on v_mfma_f32_16x16x4_f32 -- half the unit, the same FLOP per cycle:
//
//   * lane l IS edge (l & 15) of the wave's tile for the k quarter (l >> 4): it loads float4 chunks of its own P / Q rows
//     (k = 16 g + 4 (l >> 4) + i) and evaluates the A operand in registers, exactly as the 32-edge kernel does; MFMA step
//     i of a 16-k group contracts k = 16 g + {0, 4, 8, 12} + i.  The same vector work per MFMA cycle as the 32-edge kernel
//     (4 A values per lane feed 64 MFMAs of 32 cycles instead of 32 MFMAs of 64 cycles).
//   * B operand: W2^T slices of 32 k through a double-buffered LDS stream; lane (n, kq) reads the 16 column tiles of its
//     k row as FOUR ds_read_b128 from a lane-grouped copy  W2TP16[k][16 n + c] = W2T[k][16 c + n]  (pack16_w2t_kernel);
//     the four 16-byte chunks are read in an order rotated by (n >> 2), which spreads a ds_read_b128 lane group over all
//     16 slots of the 256-byte bank row (conflict-free); accumulator a of lane n then holds column tile
//     4 (((a >> 2) + (n >> 2)) & 3) + (a & 3).  64 accumulator registers at H = 256.
//   * epilogue, MODE_GCL: attention dot product = in-lane sum over the 16 column tiles + an all-reduce over the 16 lanes of
//     a row (4 DPP rotations); the segmented row sums of the tile are ONE more MFMA pass:  S[seg][f] = sum_e ind[seg][e]
//     out[e][f]  with the indicator matrix as A operand and the accumulators as B operand (the k slot (m, kq) stands for
//     edge 4 kq + m, which is the edge lane (n, kq) holds in register m) -- 64 extra MFMAs per tile (+6 %), no LDS staging,
//     a fixed summation order; lane (n, kq) then holds segment 4 kq + r in register r and stores it by the aggregation
//     protocol of edge_mlp.h with 16-edge tiles (head slots are indexed by 16-edge tile: the completion kernels take the
//     tile shift as an argument).
//   * epilogue, MODE_COORD: scalar head by the same all-reduce, trans per edge in the lanes of the first quarter, a
//     16-step segmented walk for the three components (as edge_wave.h).  One (tile, MLP) item per workgroup pass.
//
// Workgroup = 4 waves = 64 edges, persistent over items (grid-stride).  The results differ from the 32-edge kernel in
// rounding only (another k grouping inside the fp32 MFMA chains, another summation tree of the row sums): <= 2e-5
// relative; each variant is bitwise reproducible.  Which variant a stage runs is an engine option (granule mask), never
// decided from timing.
#pragma once
#include "common.h"
#include "edge_mlp.h"
#include "edge_wave.h"
#include "node_chain.h"   // mfma16

namespace dsbdd {

// W2TP16[k][16 n + c] = W2T[k][16 c + n]
__global__ void pack16_w2t_kernel(const float* src, float* dst, int H) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= H * H) return;
  const int k = i / H, o = i % H, ct = H / 16;
  const int n = o / ct, c = o % ct;
  dst[i] = src[k * H + c * 16 + n];
}

__device__ __forceinline__ float dpp_ror4(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_ror2(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x122, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_ror1(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121, 0xF, 0xF, true));
}
// sum over the 16 lanes of a row, result in every lane of the row (rotations inside the row: a fixed order)
__device__ __forceinline__ float row16_allreduce(float v) {
  v += dpp_ror8(v);
  v += dpp_ror4(v);
  v += dpp_ror2(v);
  v += dpp_ror1(v);
  return v;
}
__device__ __forceinline__ int dpp_row_shr1_i(int v, int fill) {   // lane n gets lane n - 1 of its row; lane 0: fill
  return __builtin_amdgcn_update_dpp(fill, v, 0x111, 0xF, 0xF, false);
}

template <int H, int MODE>
struct Wave16Layout {
  static constexpr int BK = 32;
  static constexpr int NV = (MODE == MODE_GCL) ? 1 : 2;
  static constexpr int VEC_PER = 7 * H;
  static constexpr int SCR_OFF = NV * VEC_PER;
  static constexpr int SCR_PER = 96;                                  // per wave scratch (floats)
  static constexpr int B_OFF = (SCR_OFF + 4 * SCR_PER + 255) / 256 * 256;
  static constexpr int TOTAL = B_OFF + 2 * BK * H;
};

template <int H, int MODE>
__global__ __launch_bounds__(kThreads, 2) void edge_wave16_kernel(EdgeArgs p) {
  using L = Wave16Layout<H, MODE>;
  constexpr int BK = L::BK, CT = H / 16, NK = H / BK, NQ = H / 4;
  constexpr int BI = BK * NQ / kThreads;        // float4 of a slice per thread (8 at H = 256)
  constexpr int BH = BI / 2;                    // per 16-k group
  static_assert(H % 64 == 0 && H <= 256 && CT % 4 == 0, "hidden_nf must be 64,128,192 or 256");
  static_assert(BI % 2 == 0, "slice split");
  __shared__ __attribute__((aligned(1024))) float smem[L::TOTAL];
  float* sB = smem + L::B_OFF;
  float* sV = smem;
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int n = lane & 15, kq = lane >> 4;
  float* s_f = smem + L::SCR_OFF + w * L::SCR_PER;           // [16] phi / gates, then trans [16][3]
  int* s_i = reinterpret_cast<int*>(s_f + 64);                // [16] segment of every edge, [16] row of every segment
  const int n_mlp = MODE == MODE_COORD ? p.n_mlp : 1;

  for (int q = 0; q < n_mlp; ++q) {
    const EdgeMlpW& mw = p.mlp[q];
    float* v = sV + q * L::VEC_PER;
    for (int i = t; i < H; i += kThreads) {
      v[i] = mw.wd[i];
      v[H + i] = mw.wd0[i];
      v[2 * H + i] = mw.table[i];
      v[3 * H + i] = mw.table[H + i];
      v[4 * H + i] = mw.table[2 * H + i];
      v[5 * H + i] = mw.b2[i];
      v[6 * H + i] = (MODE == MODE_GCL) ? (p.attention ? p.att_w[i] : 0.f) : p.w3[i];
    }
  }
  const float att_b = (MODE == MODE_GCL && p.attention) ? p.att_b[0] : 0.f;
  const float inv_norm = 1.0f / p.norm_factor;
  const int rot = n >> 2;
  auto ctile = [&](int a) { return 4 * (((a >> 2) + rot) & 3) + (a & 3); };     // column tile of accumulator a (CT = 16)
  auto feat = [&](int a) { return 16 * ((CT == 16) ? ctile(a) : a) + n; };
  const int E = min(*p.e_count, p.e_cap);
  const int ntile = (E + 63) / 64;
  const int nitems = ntile * n_mlp;
  if ((int)blockIdx.x >= nitems) return;

  // ---- W2^T slice stream through staging registers (half a slice per 16-k group) ----
  f32x4 stg[BH];
  auto stage_load = [&](int q, int ks, int g) {
    const char* src = reinterpret_cast<const char*>(p.mlp[q].W2TP16 + (size_t)ks * BK * H);
#pragma unroll
    for (int i = 0; i < BH; ++i) stg[i] = *reinterpret_cast<const f32x4*>(src + ((size_t)(g * BH + i) * kThreads + t) * 16);
  };
  auto stage_store = [&](int buf, int g) {
    float* dst = sB + buf * BK * H;
#pragma unroll
    for (int i = 0; i < BH; ++i) *reinterpret_cast<f32x4*>(dst + ((size_t)(g * BH + i) * kThreads + t) * 4) = stg[i];
  };

  // ---- this lane's edge (current item) and the prefetched one ----
  int my_r = -1, my_c = 0, my_ty = 0, my_prev = -1;
  float my_d = 0.f, my_d0 = 0.f, xr[3] = {0.f, 0.f, 0.f}, xc[3] = {0.f, 0.f, 0.f};
  int nx_r = -1, nx_c = 0, nx_prev = -1;
  float nx_d0 = 0.f, nxr[3] = {0.f, 0.f, 0.f}, nxc[3] = {0.f, 0.f, 0.f};
  int vzero;
  asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));
  auto fetch_idx = [&](int tile) {
    const int e0 = tile * 64 + w * 16, e = e0 + n;
    nx_r = -1; nx_c = 0; nx_d0 = 0.f; nx_prev = -1;
    if (e < E) { nx_r = p.erow[e]; nx_c = p.ecol[e]; nx_d0 = p.ed0[e]; }
    if (e0 > 0 && e0 < E) nx_prev = p.erow[e0 - 1 + vzero];
  };
  auto fetch_x = [&]() {
    if ((unsigned)nx_r >= (unsigned)p.n_nodes || (unsigned)nx_c >= (unsigned)p.n_nodes) { nx_r = -1; nx_c = 0; }
    if (nx_r >= 0) {
#pragma unroll
      for (int k = 0; k < 3; ++k) { nxr[k] = p.x[3 * nx_r + k]; nxc[k] = p.x[3 * nx_c + k]; }
    }
  };
  auto commit_edge = [&]() {
    my_r = nx_r; my_c = nx_c; my_d0 = nx_d0; my_d = 0.f; my_ty = 0; my_prev = nx_prev;
#pragma unroll
    for (int k = 0; k < 3; ++k) { xr[k] = nxr[k]; xc[k] = nxc[k]; }
    if (my_r >= 0) {
      const float dx = xr[0] - xc[0], dy = xr[1] - xc[1], dz = xr[2] - xc[2];
      my_d = dx * dx + dy * dy + dz * dz;
      const bool rl = my_r < p.n_lig, cl = my_c < p.n_lig;
      my_ty = (rl && cl) ? 1 : ((!rl && !cl) ? 2 : 0);
    }
  };

  int item = blockIdx.x;
  int tile = item / n_mlp, q = item - tile * n_mlp;
  stage_load(q, 0, 0); stage_store(0, 0);
  stage_load(q, 0, 1); stage_store(0, 1);
  fetch_idx(tile);
  fetch_x();
  commit_edge();
  __syncthreads();
  int bslice = 0;
  // first P / Q chunk of an item: requested before the previous item's epilogue (here: before the first item)
  f32x4 pc_next = ldv4(p.mlp[q].P + (size_t)(my_r < 0 ? 0 : my_r) * p.ldpq + 4 * kq);
  f32x4 qc_next = ldv4(p.mlp[q].Q + (size_t)my_c * p.ldpq + 4 * kq);

#pragma unroll 1
  for (;;) {
    const int next_item = item + gridDim.x;
    const bool has_next = next_item < nitems;
    const int tile_n = has_next ? next_item / n_mlp : 0, q_n = has_next ? next_item - tile_n * n_mlp : q;
    const bool new_edges = has_next && tile_n != tile;
    const float* vq = sV + q * L::VEC_PER;
    const float* Pp = p.mlp[q].P + (size_t)(my_r < 0 ? 0 : my_r) * p.ldpq + 4 * kq;
    const float* Qp = p.mlp[q].Q + (size_t)my_c * p.ldpq + 4 * kq;
    f32x4 pc = pc_next, qc = qc_next, pn = pc, qn4 = qc;

    f32x4 acc[CT];
#pragma unroll
    for (int a = 0; a < CT; ++a) {
      const float bv = vq[5 * H + feat(a)];
      acc[a] = f32x4{bv, bv, bv, bv};
    }
    const f32x2 dd = splat2(my_d), dz = splat2(my_d0);

#pragma unroll 1
    for (int kt = 0; kt < NK; ++kt) {
      const bool more = kt + 1 < NK;
      if (kt == 0 && new_edges) fetch_idx(tile_n);
      if (kt == 1 && new_edges) fetch_x();
      const int sq = more ? q : q_n, sks = more ? kt + 1 : 0;
      const float* bcur = sB + (bslice & 1) * BK * H + (4 * kq) * H + n * CT;
      const float* vk = vq + kt * BK + 4 * kq;
      const float* vt = vk + (2 + my_ty) * H;
#pragma unroll
      for (int g = 0; g < BK / 16; ++g) {
        const int kb = kt * BK + 16 * g;
        if (g > 0) stage_store((bslice + 1) & 1, g - 1);
        stage_load(sq, sks, g);
        if (g + 1 < BK / 16 || more) { pn = ldv4(Pp + kb + 16); qn4 = ldv4(Qp + kb + 16); }
        const f32x4 wd4 = *reinterpret_cast<const f32x4*>(vk + 16 * g);
        const f32x4 wz4 = *reinterpret_cast<const f32x4*>(vk + H + 16 * g);
        const f32x4 tb4 = *reinterpret_cast<const f32x4*>(vt + 16 * g);
        f32x2 alo = pk_fma(dz, wz4.xy, pk_fma(dd, wd4.xy, pc.xy + qc.xy)) + tb4.xy;
        f32x2 ahi = pk_fma(dz, wz4.zw, pk_fma(dd, wd4.zw, pc.zw + qc.zw)) + tb4.zw;
        alo = silu2(alo);
        ahi = silu2(ahi);
        const float av[4] = {alo.x, alo.y, ahi.x, ahi.y};
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float* brow = bcur + (16 * g + i) * H;
          float bv[CT];
#pragma unroll
          for (int c4 = 0; c4 < CT / 4; ++c4) {
            const int ch = (CT == 16) ? ((c4 + rot) & 3) : c4;
            const float4 b4 = *reinterpret_cast<const float4*>(brow + 4 * ch);
            bv[4 * c4] = b4.x; bv[4 * c4 + 1] = b4.y; bv[4 * c4 + 2] = b4.z; bv[4 * c4 + 3] = b4.w;
          }
#pragma unroll
          for (int a = 0; a < CT; ++a) acc[a] = mfma16(av[i], bv[a], acc[a]);
        }
        __builtin_amdgcn_s_setprio(0);
        pc = pn; qc = qn4;
      }
      stage_store((bslice + 1) & 1, BK / 16 - 1);
      ++bslice;
      __syncthreads();
    }

    if (has_next) {                                          // the next item's first P / Q chunk flies during the epilogue
      const int r_n = new_edges ? nx_r : my_r, c_n = new_edges ? nx_c : my_c;
      pc_next = ldv4(p.mlp[q_n].P + (size_t)(r_n < 0 ? 0 : r_n) * p.ldpq + 4 * kq);
      qc_next = ldv4(p.mlp[q_n].Q + (size_t)c_n * p.ldpq + 4 * kq);
    }
    // ================= wave-private epilogue: lane (n, kq), register r <-> edge 4 kq + r, feature feat(a) =================
    const int wt16 = p.wt_base * 2 + tile * 4 + w;          // global 16-edge tile index (wt_base counts 32-edge tiles)
    if (MODE == MODE_GCL) {
#pragma unroll
      for (int a = 0; a < CT; ++a) {
        const f32x2 m01 = silu2(f32x2{acc[a][0], acc[a][1]}), m23 = silu2(f32x2{acc[a][2], acc[a][3]});
        acc[a] = f32x4{m01.x, m01.y, m23.x, m23.y};
      }
      // edges of this lane's registers: active?
      if (kq == 0) s_i[n] = my_r;
      wave_lds_fence();
      const int4 rows4 = *reinterpret_cast<const int4*>(s_i + 4 * kq);
      wave_lds_fence();
      const bool act[4] = {rows4.x >= 0, rows4.y >= 0, rows4.z >= 0, rows4.w >= 0};
      if (p.attention) {
        float part[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int a = 0; a < CT; ++a) {
          const float aw = vq[6 * H + feat(a)];
#pragma unroll
          for (int r = 0; r < 4; ++r) part[r] = fmaf(acc[a][r], aw, part[r]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) part[r] = sigmoidf_fast(row16_allreduce(part[r]) + att_b);
#pragma unroll
        for (int a = 0; a < CT; ++a)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[a][r] *= part[r];
      }
#pragma unroll
      for (int a = 0; a < CT; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[a][r] = act[r] ? acc[a][r] : 0.f;      // entries that name no edge add nothing
      // segments of the tile: runs of equal row ids
      const int prev_row = dpp_row_shr1_i(my_r, -2);
      const bool first = n == 0 || my_r != prev_row;
      const unsigned long long bal = __builtin_amdgcn_ballot_w64(first && kq == 0);
      const unsigned m16 = (unsigned)bal & 0xFFFFu;
      const int seg = __builtin_popcount(m16 & ((2u << n) - 1u)) - 1;          // segment of edge n
      if (kq == 0) {
        s_i[n] = seg;
        if (first) s_i[16 + seg] = my_r;
      }
      wave_lds_fence();
      const int4 sg4 = *reinterpret_cast<const int4*>(s_i + 4 * kq);           // segments of edges 4 kq + m
      const int4 sr4 = *reinterpret_cast<const int4*>(s_i + 16 + 4 * kq);      // rows of segments 4 kq + r
      wave_lds_fence();
      const int nseg = __builtin_popcount(m16);
      const float ind[4] = {sg4.x == n ? 1.f : 0.f, sg4.y == n ? 1.f : 0.f, sg4.z == n ? 1.f : 0.f, sg4.w == n ? 1.f : 0.f};
      const int srow[4] = {sr4.x, sr4.y, sr4.z, sr4.w};
      const int row0 = __builtin_amdgcn_readlane(my_r, 0);
      const bool head0 = row0 >= 0 && row0 == __builtin_amdgcn_readfirstlane(my_prev);
      float* dst[4];
      bool st[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int s = 4 * kq + r;
        st[r] = s < nseg && srow[r] >= 0;
        dst[r] = (s == 0 && head0) ? p.agg_head + (size_t)wt16 * H : p.agg + (size_t)(srow[r] < 0 ? 0 : srow[r]) * H;
      }
#pragma unroll
      for (int a0 = 0; a0 < CT; a0 += 4) {
        f32x4 S[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) S[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int j = 0; j < 4; ++j) S[j] = mfma16(ind[m], acc[a0 + j][m], S[j]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int f = feat(a0 + j);
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (st[r]) dst[r][f] = S[j][r] * inv_norm;
        }
      }
    } else {
      float part[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int a = 0; a < CT; ++a) {
        const float wv = vq[6 * H + feat(a)];
        const f32x2 m01 = silu2(f32x2{acc[a][0], acc[a][1]}), m23 = silu2(f32x2{acc[a][2], acc[a][3]});
        part[0] = fmaf(m01.x, wv, part[0]); part[1] = fmaf(m01.y, wv, part[1]);
        part[2] = fmaf(m23.x, wv, part[2]); part[3] = fmaf(m23.y, wv, part[3]);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) part[r] = row16_allreduce(part[r]);
      if (n == 0) *reinterpret_cast<float4*>(s_f + 4 * kq) = make_float4(part[0], part[1], part[2], part[3]);
      wave_lds_fence();
      const float ph = s_f[n];
      wave_lds_fence();
      float tx = 0.f, ty = 0.f, tz = 0.f;
      if (my_r >= 0 && kq == 0) {
        const float dx = xr[0] - xc[0], dy = xr[1] - xc[1], dzz = xr[2] - xc[2];
        if (q == 0) {
          const float den = sqrtf(my_d + 1e-8f) + p.norm_constant;
          const float T = p.use_tanh ? tanhf(ph) * p.coords_range : ph;
          tx = dx / den * T; ty = dy / den * T; tz = dzz / den * T;
        } else {
          const int b = p.node_batch[my_r];
          const float m0 = p.mean[3 * b], m1 = p.mean[3 * b + 1], m2 = p.mean[3 * b + 2];
          const float a0 = xr[0] - m0, a1 = xr[1] - m1, a2 = xr[2] - m2;
          const float b0 = xc[0] - m0, b1 = xc[1] - m1, b2 = xc[2] - m2;
          const float c0 = a1 * b2 - a2 * b1, c1 = a2 * b0 - a0 * b2, c2 = a0 * b1 - a1 * b0;
          const float cden = sqrtf(c0 * c0 + c1 * c1 + c2 * c2) + p.norm_constant;
          const float T = p.use_tanh ? tanhf(ph) * p.coords_range : ph;
          tx = c0 / cden * T; ty = c1 / cden * T; tz = c2 / cden * T;
        }
      }
      if (kq == 0) { s_f[16 + 3 * n] = tx; s_f[16 + 3 * n + 1] = ty; s_f[16 + 3 * n + 2] = tz; }
      wave_lds_fence();
      float trv[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) trv[e] = s_f[16 + 3 * e + (lane < 3 ? lane : 0)];
      if (lane < 3) {
        float* xa = p.xagg + q * p.xagg_stride;
        float* xh = p.xagg_head + q * p.xhead_stride;
        const int row0 = __builtin_amdgcn_readlane(my_r, 0);
        bool to_head = row0 >= 0 && row0 == __builtin_amdgcn_readfirstlane(my_prev);
        int cur = -1;
        float sum = 0.f;
        auto put = [&]() {
          if (cur >= 0) {
            const float v = sum / p.norm_factor;
            if (to_head) xh[4 * (size_t)wt16 + lane] = v; else xa[(size_t)cur * 3 + lane] = v;
            to_head = false;
          }
        };
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int rn = __builtin_amdgcn_readlane(my_r, e);
          if (rn != cur) { put(); cur = rn; sum = 0.f; }
          sum += trv[e];
        }
        put();
      }
      wave_lds_fence();
    }

    if (!has_next) break;
    if (new_edges) commit_edge();
    item = next_item; tile = tile_n; q = q_n;
  }
}

}  // namespace dsbdd
