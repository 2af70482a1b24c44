// Split-K variant of the fused edge-MLP kernels (same math, arguments and aggregation protocol as edge_wave.h / edge_mlp.h;
// reference: GCL.edge_model egnn_new.py:31-46, EquivariantUpdate.coord_model :96-122) for launches that do not fill the
// chip with 128-edge workgroup tiles -- the latency regime (crossdock_ca_cond x 32: 153 tiles on 256 CUs; the
// free-running full-atom chain: 272 - 509 tiles on 512 resident workgroups).
//
// edge_wave.h: a WAVE owns 32 edges x all H features = 1 024 v_mfma_f32_32x32x2_f32 = 27 us of matrix time whatever the
// launch size.  Splitting the COLUMNS of a tile over the waves was built and measured twice (rounds 2 and 3): every wave
// then still evaluates the whole A operand (SiLU of the first layer, 4 values per 8 k) for a quarter of the MFMAs, and on
// gfx950 vector instructions are never free beside fp32 MFMAs -- no gain.  Here the REDUCTION dimension is split:
//
//   * a WORKGROUP owns 32 edges (one wave tile of the aggregation protocol); wave w takes k in [w H/4, (w + 1) H/4) of the
//     H x H layer for ALL H output columns: 256 MFMAs (7 us) per wave and item at H = 256, the A operand evaluated once per
//     element (lane l is edge l & 31 exactly as in edge_wave.h; half-wave h takes k = 8 g + 4 h + i of every group of 8).
//   * no LDS staging of W2^T and no barrier in the K loop: the four waves read disjoint K ranges, so nothing is shared.
//     Every lane loads its B values straight from L2 as two 16-byte words per k from a packed copy (pack_w2sk_kernel)
//         W2SK[k][j][cl] = W2T[k][32 ((cl + OWN w) mod CT) + j],   w = k / (H/4), OWN = CT / 4,
//     one group (8 k) ahead in a second register set: 16 B/clk/CU of L2 -> L1 traffic at the MFMA rate.  The column
//     tiles are ROTATED per wave, so that accumulator tile cl of wave w holds feature tile (cl + OWN w) mod CT and the
//     tiles a wave finally owns are always its accumulators 0 .. OWN-1: every register index below is a constant.
//   * epilogue: reduce-scatter of the four partial accumulators through LDS in three rounds (round r: wave w hands its
//     partial of the tiles wave w + r owns -- its accumulators OWN r .. OWN r + OWN - 1 -- to that wave; 32 KB per round,
//     two buffers, three barriers), after which wave w holds z2 of 32 edges x its H/4 columns (second-layer bias: the
//     owner's accumulators start from it).  Attention dot / scalar head: per-wave partial sums over its columns
//     (reduce16_half_wave), the four partials added in wave order through 512 bytes of LDS, one sigmoid per lane.
//     Segmented row sums and the stores follow edge_wave.h on the wave's own columns: the same aggregation protocol (head
//     slots per 32-edge tile), so every per-row sum is still a pure function of the sample's own data.
//   * one work item = (32-edge tile, MLP); static round-robin inside each XCD's contiguous range, 2 workgroups per CU.
//
// Results differ from edge_wave.h in the association of the K sum only (four partial sums of H/4 terms instead of one
// chain): both within 1e-4 of the oracle, each bitwise reproducible.  Which stages use it is an engine option
// (DSBDD_OPT_SPLITK, a chain constant like DSBDD_OPT_GRANULE16).
#pragma once
#include "common.h"
#include "edge_mlp.h"
#include "edge_wave.h"

namespace dsbdd {

template <int H>
struct SplitKLayout {
  static constexpr int CT = H / 32;              // 32-column tiles
  static constexpr int OWN = CT / 4;             // tiles a wave owns after the reduce-scatter
  static constexpr int KW = H / 4;               // k per wave
  static constexpr int NG = KW / 8;              // groups of 8 k per wave and item
  static constexpr int VEC_OFF = 0;              // wd, wd0, tab0..2, b2, w-out of the workgroup's MLP
  static constexpr int SCR_OFF = 7 * H;          // per wave: gates[32] / trans[32][3]
  static constexpr int SCR_PER = 96;
  static constexpr int RED_OFF = SCR_OFF + 4 * SCR_PER;            // [4 waves][32 rows] partial dot products
  static constexpr int X_OFF = (RED_OFF + 128 + 255) / 256 * 256;  // exchange buffers [2][4 waves][OWN][4][64 lanes][4]
  static constexpr int X_BUF = 4 * OWN * 16 * 64;
  static constexpr int TOTAL = X_OFF + 2 * X_BUF;
};

// one thread per element: the per-wave rotated, lane-grouped copy of W2T (see the top of the file)
__global__ __launch_bounds__(256) void pack_w2sk_kernel(const float* __restrict__ W2T, float* __restrict__ out, int H) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= H * H) return;
  const int CT = H / 32, OWN = CT / 4;
  const int k = idx / H, r = idx % H, j = r / CT, cl = r % CT;
  const int w = k / (H / 4);
  out[idx] = W2T[(size_t)k * H + 32 * ((cl + OWN * w) % CT) + j];
}

template <int H, int MODE>
__global__ __launch_bounds__(kThreads, 2) void edge_splitk_kernel(EdgeArgs p) {
  using L = SplitKLayout<H>;
  constexpr int CT = L::CT, OWN = L::OWN, KW = L::KW, NG = L::NG, NB = CT / 4;   // NB: 16-byte words of B per k and lane
  static_assert(H % 128 == 0 && H <= 256, "split-K kernels: hidden_nf 128 or 256");
  static_assert(OWN == 2, "the segmented sums exchange column tiles in pairs (H = 256)");
  static_assert(NG >= 4, "metadata prefetch needs four groups per item");

  __shared__ __attribute__((aligned(1024))) float smem[L::TOTAL];
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int half = lane >> 5, j = lane & 31;
  float* sV = smem + L::VEC_OFF;
  float* s_phi = smem + L::SCR_OFF + w * L::SCR_PER;
  float* s_tr = s_phi;
  float* s_red = smem + L::RED_OFF;
  float* s_x = smem + L::X_OFF;
  const bool split = MODE == MODE_COORD && p.n_mlp == 2;
  const int qsel = split ? ((blockIdx.x >> 3) & 1) : 0;        // the MLP this workgroup evaluates
  const EdgeMlpW& mw = p.mlp[qsel];

  // this wave's B rows: k = w KW + 8 g + 4 half + i  ->  Bw + (8 g + i) H; the first group is requested before anything else
  // Buffer loads: a wave-uniform descriptor of this wave's K range + a 32-bit per-lane byte offset + a scalar offset per
  // (group, step).  (Plain global loads made the compiler keep 32 per-lane 64-bit addresses -- 64 VGPRs -- alive.)
  const unsigned long long bw_addr = reinterpret_cast<unsigned long long>(mw.W2SK + (size_t)w * KW * H);
  const unsigned bw_lo = __builtin_amdgcn_readfirstlane((unsigned)bw_addr);
  const unsigned bw_hi = __builtin_amdgcn_readfirstlane((unsigned)(bw_addr >> 32));
  const __amdgpu_buffer_rsrc_t Bw = __builtin_amdgcn_make_buffer_rsrc(
      reinterpret_cast<void*>(((unsigned long long)bw_hi << 32) | bw_lo), 0, KW * H * 4, 0x00020000);
  const int boff = (4 * half * H + j * CT) * 4;
  // ring of four steps: slot i holds the B words of step (g, i) until its MFMAs have issued and is then refilled with step
  // (g + 1, i) -- one group (32 MFMAs) of latency budget on 8 NB registers instead of two full sets
  f32x4 bq[4][NB];
  auto load_b = [&](int g, int i) {
#pragma unroll
    for (int u = 0; u < NB; ++u)
      bq[i][u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(Bw, boff, (8 * g + i) * H * 4 + 16 * u, 0));
  };
#pragma unroll
  for (int i = 0; i < 4; ++i) load_b(0, i);

  for (int i = t; i < H; i += kThreads) {
    sV[i] = mw.wd[i];
    sV[H + i] = mw.wd0[i];
    sV[2 * H + i] = mw.table[i];
    sV[3 * H + i] = mw.table[H + i];
    sV[4 * H + i] = mw.table[2 * H + i];
    sV[5 * H + i] = mw.b2[i];
    sV[6 * H + i] = (MODE == MODE_GCL) ? (p.attention ? p.att_w[i] : 0.f) : p.w3[i];
  }
  const float att_b = (MODE == MODE_GCL && p.attention) ? p.att_b[0] : 0.f;
  const float inv_norm = 1.0f / p.norm_factor;                 // egnn_new.py:328-329 as one multiply per flushed segment
  auto feat = [&](int cl) { return 32 * ((cl + OWN * w) % CT) + j; };   // feature held by accumulator tile cl of this wave

  constexpr int BMW = 32;
  const int E = min(*p.e_count, p.e_cap);
  const int nt_a = (E + BMW - 1) / BMW;
  const int E_b = (MODE == MODE_GCL && p.e_count_b) ? min(*p.e_count_b, p.e_cap_b) : 0;   // second list of the stage
  const int ntiles = nt_a + (E_b + BMW - 1) / BMW;
  const int xcd = blockIdx.x & 7;
  const int kx = split ? (blockIdx.x >> 4) : (blockIdx.x >> 3);
  const int gx = split ? (gridDim.x >> 4) : (gridDim.x >> 3);
  const int tq = ntiles / 8, tr = ntiles % 8;
  const int csize = tq + (xcd < tr ? 1 : 0);
  const int cbase = (xcd < tr) ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
  if (kx >= csize) return;

  // ---- this lane's edge (current item) and the prefetched one (next item) ----------------
  int my_r = -1, my_c = 0, my_ty = 0;
  float my_d = 0.f, my_d0 = 0.f, xr[3] = {0.f, 0.f, 0.f}, xc[3] = {0.f, 0.f, 0.f};
  int nx_r = -1, nx_c = 0;
  int my_prev = -1, nx_prev = -1;      // row of the edge just before this tile (wave-uniform)
  int my_wt = 0, nx_wt = 0;            // global wave-tile index
  bool my_lb = false, nx_lb = false;   // the tile belongs to the stage's second list
  float nx_d0 = 0.f, nxr[3] = {0.f, 0.f, 0.f}, nxc[3] = {0.f, 0.f, 0.f};
  int vzero;
  asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));
  auto fetch_idx = [&](int tile) {
    const bool lb = MODE == MODE_GCL && tile >= nt_a;
    const int tl = lb ? tile - nt_a : tile, El = lb ? E_b : E;
    const int* er = lb ? p.erow_b : p.erow;
    const int* ec = lb ? p.ecol_b : p.ecol;
    const float* ed = lb ? p.ed0_b : p.ed0;
    const int e0 = tl * BMW, e = e0 + j;
    nx_r = -1; nx_c = 0; nx_d0 = 0.f; nx_prev = -1; nx_wt = (lb ? p.wt_base_b : p.wt_base) + tl; nx_lb = lb;
    if (e < El) { nx_r = er[e]; nx_c = ec[e]; nx_d0 = ed[e]; }
    if (e0 > 0 && e0 < El) nx_prev = er[e0 - 1 + vzero];
  };
  auto fetch_x = [&]() {
    // entries that do not name two rows of this call (stale workspace words after an overflowed build) are inactive
    if ((unsigned)nx_r >= (unsigned)p.n_nodes || (unsigned)nx_c >= (unsigned)p.n_nodes) { nx_r = -1; nx_c = 0; }
    if (nx_r >= 0) {
#pragma unroll
      for (int k = 0; k < 3; ++k) { nxr[k] = p.x[3 * nx_r + k]; nxc[k] = p.x[3 * nx_c + k]; }
    }
  };
  auto commit_edge = [&]() {
    my_r = nx_r; my_c = nx_c; my_d0 = nx_d0; my_d = 0.f; my_ty = 0;
    my_prev = nx_prev; my_wt = nx_wt; my_lb = nx_lb;
#pragma unroll
    for (int k = 0; k < 3; ++k) { xr[k] = nxr[k]; xc[k] = nxc[k]; }
    if (my_r >= 0) {
      const float dx = xr[0] - xc[0], dy = xr[1] - xc[1], dz = xr[2] - xc[2];
      my_d = dx * dx + dy * dy + dz * dz;                  // coord2diff radial, egnn_new.py:298-299
      const bool rl = my_r < p.n_lig, cl = my_c < p.n_lig;
      my_ty = (rl && cl) ? 1 : ((!rl && !cl) ? 2 : 0);     // dynamics.py:119-124
    }
  };

  fetch_idx(cbase + kx);
  fetch_x();
  commit_edge();
  __syncthreads();          // vectors visible

  // this lane's k of group g: w KW + 8 g + 4 half + i
  const int koff = w * KW + 4 * half;
  const float* Pp = mw.P + (size_t)(my_r < 0 ? 0 : my_r) * p.ldpq + koff;
  const float* Qp = mw.Q + (size_t)my_c * p.ldpq + koff;
  f32x4 pc = ldv4(Pp), qc = ldv4(Qp);
  const float* vk = sV + koff;

  int li = kx, next_li = 0;
  bool has_next = false;
#pragma unroll 1
  for (;;) {
    // accumulators: the tiles this wave owns start from the second layer's bias, the partials handed away from zero
    // (the handed-away tiles take a zero C operand in their first MFMA: no 96 register clears per item)
    f32x16 acc[CT];
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int cl = 0; cl < OWN; ++cl) {
      const float bv = sV[5 * H + feat(cl)];
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[cl][r] = bv;
    }
    const float* vt = vk + (2 + my_ty) * H;
    const f32x2 dd = splat2(my_d), dz = splat2(my_d0);

#pragma unroll
    for (int g = 0; g < NG; ++g) {
      // next item of this workgroup: its edge with two dependent loads behind the MFMAs of this one
      if (g == 0) {
        next_li = li + gx;
        has_next = next_li < csize;
        if (has_next) fetch_idx(cbase + next_li);
      }
      if (g == 2 && has_next) fetch_x();
      // A operand: SiLU((P + Q) + d wd + d0 wd0 + tab), the arithmetic of edge_wave.h
      const f32x4 wd4 = *reinterpret_cast<const f32x4*>(vk + 8 * g);
      const f32x4 wz4 = *reinterpret_cast<const f32x4*>(vk + H + 8 * g);
      const f32x4 tb4 = *reinterpret_cast<const f32x4*>(vt + 8 * g);
      f32x2 alo = pk_fma(dz, wz4.xy, pk_fma(dd, wd4.xy, pc.xy + qc.xy)) + tb4.xy;
      f32x2 ahi = pk_fma(dz, wz4.zw, pk_fma(dd, wd4.zw, pc.zw + qc.zw)) + tb4.zw;
      alo = silu2(alo);
      ahi = silu2(ahi);
      const float a[4] = {alo.x, alo.y, ahi.x, ahi.y};
      // the next chunk of this lane's P / Q rows (the last group: the first chunk of the next item's rows)
      if (g + 1 < NG) {
        pc = ldv4(Pp + 8 * (g + 1));
        qc = ldv4(Qp + 8 * (g + 1));
      } else if (has_next) {
        Pp = mw.P + (size_t)(nx_r < 0 ? 0 : nx_r) * p.ldpq + koff;
        Qp = mw.Q + (size_t)nx_c * p.ldpq + koff;
        pc = ldv4(Pp);
        qc = ldv4(Qp);
      }
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int cl = 0; cl < CT; ++cl)
          acc[cl] = mfma32(a[i], bq[i][cl >> 2][cl & 3], (g == 0 && i == 0 && cl >= OWN) ? zero16 : acc[cl]);
        load_b((g + 1) % NG, i);        // (the last group requests group 0 again: the next item's, same rows)
      }
      __builtin_amdgcn_s_setprio(0);
    }

    // ================= reduce-scatter of the four partial accumulators =================
    // round r: this wave's partial of the tiles wave (w + r) & 3 owns = its accumulators OWN r .. OWN r + OWN - 1
#pragma unroll
    for (int r = 1; r < 4; ++r) {
      float* xb = s_x + ((r - 1) & 1) * L::X_BUF;
      float* dst = xb + (size_t)w * (OWN * 16 * 64) + lane * 4;
#pragma unroll
      for (int o = 0; o < OWN; ++o)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4)
          *reinterpret_cast<f32x4*>(dst + (o * 4 + r4) * 256) =
              f32x4{acc[OWN * r + o][4 * r4], acc[OWN * r + o][4 * r4 + 1], acc[OWN * r + o][4 * r4 + 2], acc[OWN * r + o][4 * r4 + 3]};
      __syncthreads();
      const float* src = xb + (size_t)((w - r) & 3) * (OWN * 16 * 64) + lane * 4;
#pragma unroll
      for (int o = 0; o < OWN; ++o)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(src + (o * 4 + r4) * 256);
          acc[o][4 * r4] += v.x; acc[o][4 * r4 + 1] += v.y; acc[o][4 * r4 + 2] += v.z; acc[o][4 * r4 + 3] += v.w;
        }
      // (round r + 1 writes the other buffer; round r + 2 re-uses this one behind round r + 1's barrier)
    }

    const bool last_item = !has_next;
    // ================= epilogue on this wave's OWN column tiles =================
    if (MODE == MODE_GCL) {
      // messages m = SiLU(acc)   (egnn_new.py:18-19; the bias is already in the accumulators)
#pragma unroll
      for (int c = 0; c < OWN; ++c)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const f32x2 m2 = silu2(f32x2{acc[c][r], acc[c][r + 1]});
          acc[c][r] = m2.x; acc[c][r + 1] = m2.y;
        }
      if (p.attention) {   // att = sigmoid(w_a . m + b_a): partial over this wave's columns, the four partials in wave order
        f32x2 part2[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) part2[r] = splat2(0.f);
#pragma unroll
        for (int c = 0; c < OWN; ++c) {
          const f32x2 aw = splat2(sV[6 * H + feat(c)]);
#pragma unroll
          for (int r = 0; r < 8; ++r) part2[r] = pk_fma(f32x2{acc[c][2 * r], acc[c][2 * r + 1]}, aw, part2[r]);
        }
        float part[16];
#pragma unroll
        for (int r = 0; r < 8; ++r) { part[2 * r] = part2[r].x; part[2 * r + 1] = part2[r].y; }
        // lane j of a half ends with the dot product of accumulator register j >> 1 of that half
        const float mine = reduce16_half_wave(part, j);
        if ((j & 1) == 0) s_red[w * 32 + 16 * half + (j >> 1)] = mine;
        __syncthreads();
        const float* rp = s_red + 16 * half + (j >> 1);
        const float dot = ((rp[0] + rp[32]) + rp[64]) + rp[96];
        const float gate = sigmoidf_fast(dot + att_b);
        s_phi[16 * half + (j >> 1)] = gate;
        wave_lds_fence();
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const float4 g4 = *reinterpret_cast<const float4*>(s_phi + 16 * half + 4 * q4);
          part[4 * q4] = g4.x; part[4 * q4 + 1] = g4.y; part[4 * q4 + 2] = g4.z; part[4 * q4 + 3] = g4.w;
        }
        wave_lds_fence();   // the words are rewritten by the next item
#pragma unroll
        for (int c = 0; c < OWN; ++c)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[c][r] *= part[r];       // mij * att, egnn_new.py:40
      } else {
        __syncthreads();    // (keeps the barrier count of an item independent of the flag)
      }
      // segmented sums over the tile's 32 rows on this wave's columns (edge_wave.h: accumulator register rr of half h is
      // row 8 (rr >> 2) + 4 h + (rr & 3); every half adds up its rows of the running segment, the halves meet at a flush)
      f32x2 sum2[OWN];
#pragma unroll
      for (int c = 0; c < OWN; ++c) sum2[c] = splat2(0.f);
      int cur = -1;
      const int row0 = __builtin_amdgcn_readlane(my_r, 0);
      bool to_head = row0 >= 0 && row0 == __builtin_amdgcn_readfirstlane(my_prev);
      auto flush = [&]() {
        if (cur >= 0) {
          float* dst = to_head ? (my_lb ? p.agg_head_b : p.agg_head) + (size_t)my_wt * H
                               : (my_lb ? p.agg_b : p.agg) + (size_t)cur * H;
          // half 0 receives and stores own tile 0, half 1 own tile 1
          const float tot = pair_sum_halves(sum2[0].x + sum2[0].y, sum2[1].x + sum2[1].y);
          dst[feat(half)] = tot * inv_norm;
          to_head = false;
        }
#pragma unroll
        for (int c = 0; c < OWN; ++c) sum2[c] = splat2(0.f);
      };
#pragma unroll
      for (int gb = 0; gb < 8; ++gb) {
        const int hh = gb & 1;
#pragma unroll
        for (int ip = 0; ip < 4; ip += 2) {
          const int k = 4 * (gb >> 1) + ip;
          const int rn0 = __builtin_amdgcn_readlane(my_r, 4 * gb + ip);
          const int rn1 = __builtin_amdgcn_readlane(my_r, 4 * gb + ip + 1);
          if (rn0 != cur) {
            flush();
            cur = rn0;
          }
          if (half == hh) {
#pragma unroll
            for (int c = 0; c < OWN; ++c) sum2[c].x += acc[c][k];
          }
          if (rn1 != rn0) {
            flush();
            cur = rn1;
          }
          if (half == hh) {
#pragma unroll
            for (int c = 0; c < OWN; ++c) sum2[c].y += acc[c][k + 1];
          }
        }
      }
      flush();
    } else {
      // scalar head: phi = w3 . SiLU(acc)   (egnn_new.py:80-92), partial over this wave's columns
      f32x2 part2[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) part2[r] = splat2(0.f);
#pragma unroll
      for (int c = 0; c < OWN; ++c) {
        const f32x2 wv = splat2(sV[6 * H + feat(c)]);
#pragma unroll
        for (int r = 0; r < 8; ++r) part2[r] = pk_fma(silu2(f32x2{acc[c][2 * r], acc[c][2 * r + 1]}), wv, part2[r]);
      }
      float part[16];
#pragma unroll
      for (int r = 0; r < 8; ++r) { part[2 * r] = part2[r].x; part[2 * r + 1] = part2[r].y; }
      // lane j of a half ends with the total of accumulator register j >> 1 = edge mfma_row(j >> 1, lane)
      const float mine = reduce16_half_wave(part, j);
      if ((j & 1) == 0) s_red[w * 32 + mfma_row(j >> 1, lane)] = mine;
      __syncthreads();
      if (w == 0) {
        const float ph = ((s_red[j] + s_red[32 + j]) + s_red[64 + j]) + s_red[96 + j];     // this lane's edge
        // trans = u*phi + cross*phi_x   (egnn_new.py:100-109, 296-316); lane = edge; this workgroup adds its MLP's term
        float tx = 0.f, ty = 0.f, tz = 0.f;
        if (my_r >= 0) {
          if (qsel == 0) {
            const float dx = xr[0] - xc[0], dy = xr[1] - xc[1], dzz = xr[2] - xc[2];
            const float den = sqrtf(my_d + 1e-8f) + p.norm_constant;
            const float ux = dx / den, uy = dy / den, uz = dzz / den;
            if (p.use_tanh) {
              const float th = tanhf(ph);
              tx = ux * th * p.coords_range; ty = uy * th * p.coords_range; tz = uz * th * p.coords_range;
            } else {
              tx = ux * ph; ty = uy * ph; tz = uz * ph;
            }
          } else {
            const int b = p.node_batch[my_r];
            const float m0 = p.mean[3 * b], m1 = p.mean[3 * b + 1], m2 = p.mean[3 * b + 2];
            const float a0 = xr[0] - m0, a1 = xr[1] - m1, a2 = xr[2] - m2;
            const float b0 = xc[0] - m0, b1 = xc[1] - m1, b2 = xc[2] - m2;
            const float c0 = a1 * b2 - a2 * b1, c1 = a2 * b0 - a0 * b2, c2 = a0 * b1 - a1 * b0;
            const float cden = sqrtf(c0 * c0 + c1 * c1 + c2 * c2) + p.norm_constant;
            float phx = ph;
            if (p.use_tanh) phx = tanhf(phx) * p.coords_range;
            tx = c0 / cden * phx; ty = c1 / cden * phx; tz = c2 / cden * phx;
          }
        }
        if (half == 0) { s_tr[3 * j] = tx; s_tr[3 * j + 1] = ty; s_tr[3 * j + 2] = tz; }
        wave_lds_fence();
        float trv[32];
#pragma unroll
        for (int e = 0; e < 32; ++e) trv[e] = s_tr[3 * e + (lane < 3 ? lane : 0)];
        if (lane < 3) {
          float* xa = p.xagg + qsel * p.xagg_stride;
          float* xh = p.xagg_head + qsel * p.xhead_stride;
          const int row0 = __builtin_amdgcn_readlane(my_r, 0);
          bool to_head = row0 >= 0 && row0 == __builtin_amdgcn_readfirstlane(my_prev);
          int cur = -1;
          float sum = 0.f;
          auto put = [&]() {
            if (cur >= 0) {
              const float v = sum / p.norm_factor;
              if (to_head) xh[4 * (size_t)my_wt + lane] = v; else xa[(size_t)cur * 3 + lane] = v;
              to_head = false;
            }
          };
#pragma unroll
          for (int e = 0; e < 32; ++e) {
            const int rn = __builtin_amdgcn_readlane(my_r, e);
            if (rn != cur) {
              put();
              cur = rn;
              sum = 0.f;
            }
            sum += trv[e];
          }
          put();
        }
        wave_lds_fence();   // scratch is reused by the next item
      }
    }

    if (last_item) break;
    commit_edge();
    li = next_li;
  }
}

}  // namespace dsbdd
