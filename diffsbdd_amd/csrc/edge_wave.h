// Third structure of the fused edge-MLP kernels (same math and arguments as
// edge_mlp.h -- see there for the algorithm and the reference citations).
//
// What the hardware counters said about the LDS-tiled kernel (rocprofv3 --pmc,
// profiles/): matrix pipe 52 % busy; every wave parked 28 % of its time at
// barriers / waitcnt and issuing ~5 VALU + 2 LDS instructions per MFMA, most of
// it to move the *A operand* (gather -> SiLU -> transposed LDS write -> LDS read)
// and the tile metadata through LDS, with a workgroup barrier every 32 MFMAs.
//
// Here the MFMA A-operand layout is used the other way round:
//
//   * one wave owns 32 edges x ALL H features.  v_mfma_f32_32x32x2_f32 wants lane l
//     to supply A[i = l & 31][k = l >> 5]: so lane l simply *is* edge (l & 31); it
//     keeps that edge's P/Q row pointers, |d|^2, d0 and type in registers, loads
//     float4 chunks of its own P and Q rows straight from L2 and evaluates
//     a = SiLU(P + Q + d*wd + d0*wd0 + tab) in registers.  Half-wave h takes
//     k in {8g + 4h .. 8g + 4h + 3} of every group of 8 (any k pairing is legal as
//     long as B uses the same one).  No LDS, no transpose, no barrier for A.
//   * only W2^T goes through LDS (K slices of 32, double buffered, a continuous
//     stream across tiles): one workgroup barrier per 128 MFMAs per wave.  The slices
//     travel global -> staging registers -> LDS, a quarter per MFMA group (an LDS-DMA
//     stream makes the compiler drain it in front of every LDS read, see below).
//   * the epilogue is wave-private: attention dot as a reduce-scatter over each
//     half-wave (v_permlane16_swap + DPP; a half-wave holds complete rows), one
//     sigmoid per row, accumulators scaled in place; segmented row sums: every half
//     adds up its rows of the running segment (even / odd rows in the two words of a
//     register pair), the halves meet through v_permlane32_swap when a segment ends;
//     one plain store per (row segment, feature) following the aggregation protocol
//     of edge_mlp.h (no atomics).
//   * vector arithmetic is written on two-element vectors (v_pk_*_f32): beside an
//     fp32 MFMA stream no vector instruction is free (tools/mfma_shadow.hip).
//   * a stage that walks two edge lists (block 0 of a framed call) runs them in one
//     launch: the second list's tiles follow the first's (EdgeArgs::*_b).
//   * tiles are assigned round-robin inside each XCD's contiguous tile range.
//
// Structures that were built, measured and dropped -- a per-XCD work queue for the tiles, the W2^T stream by LDS-DMA
// (global_load_lds), scheduling fences / a one-step-ahead activation pipeline in the emulated path, the in-kernel
// timestamp / phase-clock / "no X" diagnostic builds -- live in tools/edge_wave_diag.h (the round-5 state of this file,
// used by the micro-benchmarks' A/B builds); their numbers are in profiles/README.md and DESIGN.md.
//
// Workgroup = 4 waves = 128 edges; LDS = 2 x 32 x H x 4 B (64 KB at H = 256) +
// vectors -> 2 workgroups per CU, which overlap each other's epilogues.
#pragma once
#include "common.h"
#include "edge_mlp.h"
#include "graph.h"

namespace dsbdd {

// EMU = 0: exact fp32 (v_mfma_f32_32x32x2_f32).  EMU = 6 / 9: fp32 EMULATED on the bf16 matrix cores -- both operands of
// the H x H layer split into three bf16 terms (x = hi + mid + lo exactly), 6 (or all 9) partial products per k step on
// v_mfma_f32_32x32x16_bf16 with fp32 accumulators; see "emulated path" below.
template <int H, int MODE, int EMU = 0>
struct WaveLayout {
  static constexpr int BK = EMU ? 16 : 32;
  // emulated path: a K slice holds the three bf16 planes of 16 k x H columns = 96 H bytes = 24 H floats
  static constexpr int B_BUF = EMU ? 24 * H : BK * H;
  static constexpr int NV = (MODE == MODE_GCL) ? 1 : 2;
  static constexpr int VEC_PER = 7 * H;
  // the per-MLP vectors come first: every in-loop read of them is then `base register + 16-bit immediate`
  // (behind the 64 KB of W2^T slices each read needed its own v_add)
  static constexpr int VEC_OFF = 0;
  static constexpr int SCR_OFF = VEC_OFF + NV * VEC_PER;
  static constexpr int SCR_PER = 32 * 3;                   // per wave: trans[32][3]; phi[32] uses the same words earlier
  static constexpr int B_OFF = (SCR_OFF + 4 * SCR_PER + 4 + 255) / 256 * 256;   // [2][BK][H], 1 KB aligned
  static constexpr int TOTAL = B_OFF + 2 * B_BUF;
};

__device__ __forceinline__ void wave_lds_fence() {
  // LDS operations of one wave complete in order; this only stops the compiler
  // from moving the later reads above the earlier writes.
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0)
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}


// ---- cross-lane helpers of the wave-private epilogue (gfx950: v_permlane{16,32}_swap, DPP) -----------------
// Lane exchanges never go through the LDS crossbar (ds_bpermute, what __shfl_xor compiles to): the two swaps move a
// whole 16- / 32-lane row between two registers in one VALU instruction, the rest are DPP operands.
__device__ __forceinline__ float dpp_xor1(float v) {   // quad_perm [1,0,3,2]
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_xor2(float v) {   // quad_perm [2,3,0,1]
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_half_mirror(float v) {   // lane i <-> 7 - i inside every group of 8
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_ror8(float v) {   // lane i <-> i ^ 8 inside every row of 16
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xF, 0xF, true));
}
// p: this lane keeps it when its row (16 lanes) is even; q: kept when odd.  Returns own kept value + the partner row's
// (lane ^ 16) value of the same register.
__device__ __forceinline__ float pair_sum_rows16(float p, float q) {
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(p), __float_as_uint(q), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// the same across the two 32-lane halves: half 0 gets p(own) + p(lane + 32), half 1 gets q(lane - 32) + q(own)
__device__ __forceinline__ float pair_sum_halves(float p, float q) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(p), __float_as_uint(q), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// Sum of 16 per-lane values over the 32 lanes of a half-wave as a reduce-scatter: every step halves the number of
// registers a lane carries (16 + 8 + 4 + 2 + 1 exchanges instead of 5 x 16 for a butterfly on every register).
// On return lane j of either half holds the total of register  rs_index(j) = j >> 1.
__device__ __forceinline__ float reduce16_half_wave(const float (&part)[16], int j) {
  float k8[8], k4[4], k2[2];
#pragma unroll
  for (int i = 0; i < 8; ++i) k8[i] = pair_sum_rows16(part[i], part[i + 8]);          // lanes ^ 16: keep [8 b4, +8)
  const bool b3 = (j >> 3) & 1, b2 = (j >> 2) & 1, b1 = (j >> 1) & 1;
#pragma unroll
  for (int i = 0; i < 4; ++i) {                                                       // lanes ^ 8: keep [.. + 4 b3, +4)
    const float send = b3 ? k8[i] : k8[i + 4], keep = b3 ? k8[i + 4] : k8[i];
    k4[i] = keep + dpp_ror8(send);
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {                                                       // lane i <-> 7 - i: keep [.. + 2 b2, +2)
    const float send = b2 ? k4[i] : k4[i + 2], keep = b2 ? k4[i + 2] : k4[i];
    k2[i] = keep + dpp_half_mirror(send);
  }
  const float send = b1 ? k2[0] : k2[1], keep = b1 ? k2[1] : k2[0];                   // lanes ^ 2: keep [.. + b1]
  const float k1 = keep + dpp_xor2(send);
  return k1 + dpp_xor1(k1);                                                           // lanes ^ 1: both hold the total
}

// ---- emulated path (EMU = 6 / 9): fp32 on the bf16 matrix cores ------------------------------------------------------
// The fp32 MFMA runs at 1/16 of the bf16 MFMA rate on gfx950 (MI355X_MICROARCH.md), and it runs ON the vector ALUs.  The
// H x H layer  z2 = a1 W2^T  is therefore also available as
//     a1 = a_hi + a_mid + a_lo,   W2^T = b_hi + b_mid + b_lo     (bf16 terms, round-to-nearest splits: EXACT, 3 x 8 bits)
//     z2 ~= sum over k of  a_hi b_hi + a_hi b_mid + a_mid b_hi + a_hi b_lo + a_mid b_mid + a_lo b_hi   [+ the 3 terms <= 2^-24 |a||b|]
// every bf16 x bf16 product exact in fp32, accumulated in the fp32 accumulators of v_mfma_f32_32x32x16_bf16: 6 MFMAs of
// 32 cycles per 16 k instead of 8 MFMAs of 64 cycles -- 2.7 x less matrix time, and the vector ALUs are free beside it.
// Error (tools/emu_error_study.py, profiles/r5_emu_error.md): not larger than the exact fp32 chain's own rounding error
// (one rounding per 16-k partial sum instead of one per k).  Lane l of the MFMA supplies A[l & 31][8 (l >> 5) + i] as a
// bf16x8: the lane still IS edge l & 31 and evaluates its 8 activations of the k step from 32-byte chunks of its P / Q
// rows.  B: the three planes of W2^T are split ONCE (pack_w2e_kernel) into the MFMA's own operand layout
//     W2E[k step][column tile c][plane][lane l][i] = plane(W2T[16 ks + 8 (l >> 5) + i][32 c + (l & 31)]),
// a K slice (96 H bytes) is copied to LDS as it lies, and a lane's operand of (c, plane) is ONE ds_read_b128 at
// `lane base + immediate` -- 1 KiB contiguous per wave and read, conflict-free.  Same tile walk, same epilogue, same
// aggregation protocol as the exact path; results differ from it in rounding only (both are <= 1e-4 from the oracle).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float bf16_rne_f(float x) {     // nearest bf16 (ties to even) as a float; finite inputs
  unsigned u = __float_as_uint(x);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return __uint_as_float(u & 0xFFFF0000u);
}

// one thread per (k step, column tile, lane, i): the three bf16 planes of W2T[k][col] in the MFMA B-operand layout
__global__ __launch_bounds__(256) void pack_w2e_kernel(const float* __restrict__ W2T, unsigned short* __restrict__ out, int H) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= H * H) return;
  const int CT = H / 32;
  const int i = idx & 7, l = (idx >> 3) & 63, c = (idx >> 9) % CT, ks = (idx >> 9) / CT;
  const int k = 16 * ks + 8 * (l >> 5) + i, col = 32 * c + (l & 31);
  const float w = W2T[(size_t)k * H + col];
  const float hi = bf16_rne_f(w), r1 = w - hi, mid = bf16_rne_f(r1), lo = bf16_rne_f(r1 - mid);
  const size_t base = (((size_t)(ks * CT + c) * 3) * 64 + l) * 8 + i;
  out[base] = (unsigned short)(__float_as_uint(hi) >> 16);
  out[base + 512] = (unsigned short)(__float_as_uint(mid) >> 16);
  out[base + 1024] = (unsigned short)(__float_as_uint(lo) >> 16);
}

template <int H, int MODE, bool BPERM, int EMU = 0, bool STORE = false>
__global__ __launch_bounds__(kThreads, 2) void edge_wave_kernel(EdgeArgs p) {
  using L = WaveLayout<H, MODE, EMU>;
  static_assert(!STORE || (!BPERM && EMU == 0), "z2 is stored by the plain exact kernels only");
  constexpr int BK = L::BK;
  constexpr int CT = H / 32;            // 32-col MFMA tiles per wave (all features)
  constexpr int NK = H / BK;            // K slices per unit
  constexpr int NQ = H / 4;
  // staging units of a W2^T slice per thread: float4 (exact path; emulated path when the 96 H bytes of a slice split
  // evenly), else float2
  constexpr int UNIT = EMU ? (((6 * H) % kThreads == 0) ? 4 : 2) : 4;      // floats per unit
  constexpr int BI = EMU ? (24 * H) / (kThreads * UNIT) : BK * NQ / kThreads;
  constexpr int NG = EMU ? CT / 2 : BK / 8;                                // groups of a K step (staging cadence)
  constexpr int BMW = 32, BMB = 128;    // edges per wave / per workgroup
  static_assert(H % 64 == 0 && H <= 256, "hidden_nf must be 64,128,192 or 256");
  static_assert(EMU || (BK * NQ) % kThreads == 0, "B slice split");
  static_assert(!EMU || (24 * H) % (kThreads * UNIT) == 0, "emulated B slice split");
  static_assert(EMU == 0 || EMU == 6 || EMU == 9, "partial products of the emulated path");
  static_assert(!EMU || !BPERM, "the emulated path has its own B layout");
  // s_setprio 1 around every MFMA cluster: the two workgroups sharing a CU are in different
  // phases, so favouring the wave that has MFMAs ready keeps the matrix pipe fed (+2.7 %)
  constexpr bool SETPRIO = true;
  // B operand from the lane-grouped W2^T copy (EdgeMlpW::W2TP): lane j finds the values of all its
  // column tiles in CT consecutive words -> one (CT = 4) or two (CT = 8) ds_read_b128 per k step
  // instead of CT/2 ds_read2_b32.  For CT = 8 the two 16-byte halves are read in swapped order by
  // the lanes with bit 3 set, which spreads a ds_read_b128 lane group over all 16 slots of the
  // 256-byte bank row; accumulator tile c of such a lane then holds feature tile c ^ 4.
  static_assert(!BPERM || CT == 8 || CT == 4, "lane-grouped B reads need 4 or 8 column tiles");
  constexpr bool bperm = BPERM;

  __shared__ __attribute__((aligned(1024))) float smem[L::TOTAL];
  float* sB = smem + L::B_OFF;              // [2][BK][H]
  float* sV = smem + L::VEC_OFF;            // per MLP: wd, wd0, tab0..2, b2, w-out
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);     // wave index as a scalar: LDS-DMA bases stay in SGPRs
  const int half = lane >> 5, j = lane & 31;
  float* s_phi = smem + L::SCR_OFF + w * L::SCR_PER;   // [32]
  float* s_tr = s_phi;                                  // [32][3] (phi is consumed before trans is written)
  const bool split = MODE == MODE_COORD && p.pass_split && p.n_mlp == 2;
  const int qsel = split ? ((blockIdx.x >> 3) & 1) : 0;   // the MLP this workgroup evaluates when split
  const int n_pass = (MODE == MODE_GCL || split) ? 1 : p.n_mlp;

  for (int q = 0; q < n_pass; ++q) {
    const EdgeMlpW& mw = p.mlp[qsel + q];
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
  // aggregate / normalization_factor (egnn_new.py:328-329) as one multiply per flushed
  // segment (<= 1 ulp from the reference's division)
  const float inv_norm = 1.0f / p.norm_factor;

  const int swb = (bperm && CT == 8) ? ((j >> 3) & 1) : 0;          // this lane reads its halves swapped
  auto feat = [&](int c) { return ((c ^ (4 * swb)) * 32) + j; };    // feature held by accumulator tile c

  const int E = min(*p.e_count, p.e_cap);
  const int nt_a = (E + BMB - 1) / BMB;
  const int E_b = (MODE == MODE_GCL && p.e_count_b) ? min(*p.e_count_b, p.e_cap_b) : 0;   // second list of the stage
  const int ntiles = nt_a + (E_b + BMB - 1) / BMB;
  const int xcd = blockIdx.x & 7;
  const int kx = split ? (blockIdx.x >> 4) : (blockIdx.x >> 3);
  const int gx = split ? (gridDim.x >> 4) : (gridDim.x >> 3);
  const int tq = ntiles / 8, tr = ntiles % 8;
  const int csize = tq + (xcd < tr ? 1 : 0);
  const int cbase = (xcd < tr) ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
  // Static round-robin over the XCD's tiles: local tiles kx, kx + gx, ... (with two resident workgroups per CU the static
  // order keeps the (kx, kx + n_CU) pairs of a CU balanced; a work queue measured 3 % slower, see the top of the file)
  if (kx >= csize) return;


  // ---- W2^T slice stream through staging registers --------------------------------------------------------
  // The compiler orders every LDS read behind ALL pending global_load_lds (it cannot tell the slice being filled
  // from the slice being read), so a DMA burst costs each wave one exposed L2 round trip per K step.  Plain loads
  // carry no such dependence: quarter g of the next slice is requested at the top of group g and written to LDS one
  // group later (its latency sits behind the 32 MFMAs in between); nobody reads that buffer before the barrier.
  constexpr int SG = EMU ? BI : (BI + 3) / 4;              // staging registers (units) per thread (emulated path: indexed by unit, the two halves of a slice reuse them)
  typedef float stg_t __attribute__((ext_vector_type(UNIT)));
  stg_t stg[SG];
  auto stage_lo = [](int g) { return BI * g / NG; };
  auto stage_load = [&](int q, int ks, int g) {
    const char* src = EMU ? reinterpret_cast<const char*>(p.mlp[qsel + q].W2E) + (size_t)ks * (96 * H)
                          : reinterpret_cast<const char*>((bperm ? p.mlp[qsel + q].W2TP : p.mlp[qsel + q].W2T) + (size_t)ks * BK * H);
    const unsigned toff = (unsigned)t * (4u * UNIT);
#pragma unroll
    for (int i = stage_lo(g); i < stage_lo(g + 1); ++i)
      stg[EMU ? i : i - stage_lo(g)] = *reinterpret_cast<const stg_t*>(src + (size_t)(kThreads * 4 * UNIT * i) + toff);
  };
  auto stage_store = [&](int buf, int g) {
    float* dst = sB + buf * L::B_BUF + t * UNIT;
#pragma unroll
    for (int i = stage_lo(g); i < stage_lo(g + 1); ++i)
      *reinterpret_cast<stg_t*>(dst + kThreads * UNIT * i) = stg[EMU ? i : i - stage_lo(g)];
  };

  // ---- this lane's edge (current unit) and the prefetched one (next tile) ----------------
  int my_r = -1, my_c = 0, my_ty = 0;
  float my_d = 0.f, my_d0 = 0.f, xr[3] = {0.f, 0.f, 0.f}, xc[3] = {0.f, 0.f, 0.f};
  int nx_r = -1, nx_c = 0;
  int my_prev = -1, nx_prev = -1;      // row of the edge just before this wave tile (wave-uniform)
  int my_wt = 0, nx_wt = 0;            // global wave-tile index
  bool my_lb = false, nx_lb = false;   // the tile belongs to the stage's second list
  float nx_d0 = 0.f, nxr[3] = {0.f, 0.f, 0.f}, nxc[3] = {0.f, 0.f, 0.f};
  int vzero;                           // 0 in a vector register the compiler cannot see through
  asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));
  // fetch_idx only REQUESTS the next tile's indices; they are looked at one K step later (fetch_x: range check, then
  // the coordinates), so that no wave waits for a global load at the top of a K step
  auto fetch_idx = [&](int tile) {
    const bool lb = MODE == MODE_GCL && tile >= nt_a;     // a tile of the second list (wave-uniform)
    const int tl = lb ? tile - nt_a : tile, El = lb ? E_b : E;
    const int* er = lb ? p.erow_b : p.erow;
    const int* ec = lb ? p.ecol_b : p.ecol;
    const float* ed = lb ? p.ed0_b : p.ed0;
    const int e0 = tl * BMB + w * BMW, e = e0 + j;
    nx_r = -1; nx_c = 0; nx_d0 = 0.f; nx_prev = -1; nx_wt = (lb ? p.wt_base_b : p.wt_base) + tl * 4 + w; nx_lb = lb;
    if (e < El) { nx_r = er[e]; nx_c = ec[e]; nx_d0 = ed[e]; }
    if (e0 > 0 && e0 < El) nx_prev = er[e0 - 1 + vzero];   // (a per-lane load: nothing waits for it here)
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

  // B values of one MFMA step: column j of every column tile (plain layout: CT 4-byte reads at stride 32) or
  // the lane-grouped copy (one or two 16-byte reads)
  auto read_b = [&](const float* brow, float (&bv)[CT]) {
    if constexpr (BPERM) {
      const float4 lo = *reinterpret_cast<const float4*>(brow);
      bv[0] = lo.x; bv[1] = lo.y; bv[2] = lo.z; bv[3] = lo.w;
      if constexpr (CT == 8) {
        const float4 hi = *reinterpret_cast<const float4*>(brow + 4 - 8 * swb);
        bv[4] = hi.x; bv[5] = hi.y; bv[6] = hi.z; bv[7] = hi.w;
      }
    } else {
#pragma unroll
      for (int c = 0; c < CT; ++c) bv[c] = brow[c * 32];
    }
  };

  // prologue: first W2^T slice, first edge, first P/Q chunk
#pragma unroll
  for (int g = 0; g < NG; ++g) { stage_load(0, 0, g); stage_store(0, g); }
  fetch_idx(cbase + kx);
  fetch_x();
  commit_edge();
  __syncthreads();          // sV + slice 0 visible
  int bslice = 0;           // running slice counter (buffer = bslice & 1)

  // this lane's k of a step: exact path 8 g + 4 half + i (float4 chunks), emulated path 16 kt + 8 half + i (two float4)
  constexpr int KH = EMU ? 8 : 4;
  const float* Pp = p.mlp[qsel].P + (size_t)(my_r < 0 ? 0 : my_r) * p.ldpq + KH * half;
  const float* Qp = p.mlp[qsel].Q + (size_t)my_c * p.ldpq + KH * half;
  f32x4 pc = ldv4(Pp), qc = ldv4(Qp), pn = pc, qn4 = qc;
  f32x4 pc1 = pc, qc1 = qc;                                // emulated path: second half of the 8-float chunk
  if constexpr (EMU != 0) { pc1 = ldv4(Pp + 4); qc1 = ldv4(Qp + 4); }
  float phi0 = 0.f, phi1 = 0.f;

  bf16x8 a_h = {}, a_m = {}, a_l = {};                     // emulated path: the current k step's activations (three bf16 planes)
  (void)a_h; (void)a_m; (void)a_l;
  int li = kx, q = 0, next_li = 0;
  bool has_next = false;
#pragma unroll 1
  for (;;) {
    const float* vq = sV + q * L::VEC_PER;
    const bool tile_ends = q == n_pass - 1;
    const int qn = tile_ends ? 0 : q + 1;                  // MLP pass of the next unit

    // the accumulators start from the second layer's bias (one fmaf chain per output starting at b2: the bias add
    // of the epilogue costs nothing)
    f32x16 acc[CT];
#pragma unroll
    for (int c = 0; c < CT; ++c) {
      const float bv = vq[5 * H + feat(c)];
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[c][r] = bv;
    }

#pragma unroll 1
    for (int kt = 0; kt < NK; ++kt) {
      const bool more = kt + 1 < NK;
      // next tile of this workgroup: its edge with two dependent loads, all behind the MFMAs of the current tile
      const int st = q * NK + kt;
      if (st == 0) {
        next_li = li + gx;
        has_next = next_li < csize;
        if (has_next) fetch_idx(cbase + next_li);
      }
      if (st == 1 && has_next) fetch_x();
      // the next W2^T slice: a continuous stream across units (the last K step of a workgroup re-reads slice 0 of
      // its MLP: never used; unconditional, so that the compiler counts the loads in flight exactly)
      const int sq = more ? q : qn, sks = more ? kt + 1 : 0;
      const f32x2 dd = splat2(my_d), dz = splat2(my_d0);
      if constexpr (EMU != 0) {
        // ---- emulated path: one 16-k step = 8 activations per lane, split into three bf16x8, 6 (9) MFMAs per column tile.
        // Order of a step: the step's activations, the next P / Q chunk requested, then per pair of column tiles the three
        // B planes (lo: 1 product, mid: 2, hi: 3) and their MFMAs; the next W2E slice travels through staging registers in
        // two halves.  (B reads one pair ahead behind scheduling fences and the next step's activations one step ahead were
        // measured neutral to slower, profiles/r5_emu_microbench.md; those builds live in tools/edge_wave_diag.h.)
        constexpr int NG1 = (NG + 1) / 2;
        auto act8 = [&](int ks, bf16x8& o_h, bf16x8& o_m, bf16x8& o_l) {       // activations of k step ks from pc / qc
          const float* vk = vq + ks * 16 + 8 * half;       // this lane's k = 16 ks + 8 half + i
          const float* vt = vk + (2 + my_ty) * H;
          float av[8];
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            const f32x4 pp = hh ? pc1 : pc, qq = hh ? qc1 : qc;
            const f32x4 wd4 = *reinterpret_cast<const f32x4*>(vk + 4 * hh);
            const f32x4 wz4 = *reinterpret_cast<const f32x4*>(vk + H + 4 * hh);
            const f32x4 tb4 = *reinterpret_cast<const f32x4*>(vt + 4 * hh);
            f32x2 alo = pk_fma(dz, wz4.xy, pk_fma(dd, wd4.xy, pp.xy + qq.xy)) + tb4.xy;   // (the exact path's arithmetic)
            f32x2 ahi = pk_fma(dz, wz4.zw, pk_fma(dd, wd4.zw, pp.zw + qq.zw)) + tb4.zw;
            alo = silu2(alo);
            ahi = silu2(ahi);
            av[4 * hh] = alo.x; av[4 * hh + 1] = alo.y; av[4 * hh + 2] = ahi.x; av[4 * hh + 3] = ahi.y;
          }
#pragma unroll
          for (int i = 0; i < 8; ++i) {                    // exact three-way split: v_cvt_pk_bf16_f32 rounds to nearest even
            const __bf16 h1 = (__bf16)av[i];
            const float r1 = av[i] - (float)h1;
            const __bf16 m1 = (__bf16)r1;
            const float r2 = r1 - (float)m1;
            o_h[i] = h1; o_m[i] = m1; o_l[i] = (__bf16)r2;
          }
        };
        auto load_pq = [&](int ks) {                       // this lane's P / Q chunk of k step ks (32 bytes of either row)
          pc = ldv4(Pp + 16 * ks); pc1 = ldv4(Pp + 16 * ks + 4);
          qc = ldv4(Qp + 16 * ks); qc1 = ldv4(Qp + 16 * ks + 4);
        };
        act8(kt, a_h, a_m, a_l);                           // the step's activations in front of its MFMAs
        load_pq(more ? kt + 1 : 0);
#pragma unroll
        for (int g = 0; g < NG1; ++g) stage_load(sq, sks, g);
        const float* bl = sB + (bslice & 1) * L::B_BUF + lane * 4;       // + (c * 3 + plane) * 256 floats
        bf16x8 bh[2], bm[2], blo[2];
        auto rdb = [&](int cp, int plane, bf16x8 (&dst)[2]) {
#pragma unroll
          for (int u = 0; u < 2; ++u)
            dst[u] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(bl + (2 * cp + u) * 768 + plane * 256));
        };
        rdb(0, 2, blo); rdb(0, 1, bm); rdb(0, 0, bh);
#define EMU_MM(a, b) do { _Pragma("unroll") for (int u = 0; u < 2; ++u) \
          acc[2 * cp + u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b[u], acc[2 * cp + u], 0, 0, 0); } while (0)
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int cp = 0; cp < CT / 2; ++cp) {
          const bool nxt = cp + 1 < CT / 2;
          if constexpr (EMU == 9) { EMU_MM(a_l, blo); EMU_MM(a_m, blo); }
          EMU_MM(a_h, blo);
          if (nxt) rdb(cp + 1, 2, blo);
          if constexpr (EMU == 9) EMU_MM(a_l, bm);
          EMU_MM(a_m, bm); EMU_MM(a_h, bm);
          if (nxt) rdb(cp + 1, 1, bm);
          EMU_MM(a_l, bh); EMU_MM(a_m, bh); EMU_MM(a_h, bh);           // (the leading product last)
          if (nxt) rdb(cp + 1, 0, bh);
          if (NG > 1 && cp == CT / 4 - 1) {                // half way: first half of the slice -> LDS, request the second
#pragma unroll
            for (int g = 0; g < NG1; ++g) stage_store((bslice + 1) & 1, g);
#pragma unroll
            for (int g = NG1; g < NG; ++g) stage_load(sq, sks, g);
          }
        }
#undef EMU_MM
        __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int g = (NG > 1 ? NG1 : 0); g < NG; ++g) stage_store((bslice + 1) & 1, g);
      } else {
      const float* bcur = sB + (bslice & 1) * L::B_BUF + (4 * half) * H + (bperm ? j * CT + 4 * swb : j);
      const float* vk = vq + kt * BK + 4 * half;           // this lane's k = kt*BK + 8g + 4*half + i
      const float* vt = vk + (2 + my_ty) * H;
#pragma unroll
      for (int g = 0; g < BK / 8; ++g) {
        const int kb = kt * BK + 8 * g;                    // this lane's k = kb + 4*half + i
        if (g > 0) stage_store((bslice + 1) & 1, g - 1);
        stage_load(sq, sks, g);
        if (g + 1 < BK / 8 || more) {                      // prefetch the next group's P/Q chunk
          pn = ldv4(Pp + kb + 8);
          qn4 = ldv4(Qp + kb + 8);
        }
        // A operand: SiLU((P + Q) + d wd + d0 wd0 + tab), two values per instruction (explicit fma: the same
        // arithmetic in every instantiation of the kernel)
        const f32x4 wd4 = *reinterpret_cast<const f32x4*>(vk + 8 * g);
        const f32x4 wz4 = *reinterpret_cast<const f32x4*>(vk + H + 8 * g);
        const f32x4 tb4 = *reinterpret_cast<const f32x4*>(vt + 8 * g);
        f32x2 alo = pk_fma(dz, wz4.xy, pk_fma(dd, wd4.xy, pc.xy + qc.xy)) + tb4.xy;
        f32x2 ahi = pk_fma(dz, wz4.zw, pk_fma(dd, wd4.zw, pc.zw + qc.zw)) + tb4.zw;
        alo = silu2(alo);
        ahi = silu2(ahi);
        const float a[4] = {alo.x, alo.y, ahi.x, ahi.y};
        if (SETPRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float* brow = bcur + (8 * g + i) * H;
          float bv[CT];
          read_b(brow, bv);
#pragma unroll
          for (int c = 0; c < CT; ++c) acc[c] = mfma32(a[i], bv[c], acc[c]);
        }
        if (SETPRIO) __builtin_amdgcn_s_setprio(0);
        pc = pn; qc = qn4;
      }
      stage_store((bslice + 1) & 1, BK / 8 - 1);
      }   // exact path
      ++bslice;
      __syncthreads();
    }

    const bool last_unit = tile_ends && !has_next;
    // first P/Q chunk of the NEXT unit: in flight during the epilogue
    if (!last_unit) {
      const int r_n = tile_ends ? nx_r : my_r, c_n = tile_ends ? nx_c : my_c;
      Pp = p.mlp[qsel + qn].P + (size_t)(r_n < 0 ? 0 : r_n) * p.ldpq + KH * half;
      Qp = p.mlp[qsel + qn].Q + (size_t)c_n * p.ldpq + KH * half;
      pc = ldv4(Pp); qc = ldv4(Qp);
      if constexpr (EMU != 0) { pc1 = ldv4(Pp + 4); qc1 = ldv4(Qp + 4); }
    }

    // ================= wave-private epilogue =================
    if constexpr (STORE) {
      // training forward: z2 (bias included) of the wave tile's valid slots, 128-byte row segments per half-wave
      const int e0 = (my_wt - p.wt_base) * BMW;
      float* zb = p.z2_out + (MODE == MODE_COORD ? (size_t)(qsel + q) * p.z2_stride : (size_t)0) + (size_t)e0 * H + j;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = 8 * (r >> 2) + 4 * half + (r & 3);
        if (e0 + row < E) {
#pragma unroll
          for (int c = 0; c < CT; ++c) zb[(size_t)row * H + 32 * c] = acc[c][r];
        }
      }
    }
    if (MODE == MODE_GCL) {
      // messages m = SiLU(acc)   (egnn_new.py:18-19; the bias is already in the accumulators), register pairs
#pragma unroll
      for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const f32x2 m2 = silu2(f32x2{acc[c][r], acc[c][r + 1]});
          acc[c][r] = m2.x; acc[c][r + 1] = m2.y;
        }
      if (p.attention) {   // att = sigmoid(w_a . m + b_a); a half-wave holds complete rows
        f32x2 part2[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) part2[r] = splat2(0.f);
#pragma unroll
        for (int c = 0; c < CT; ++c) {
          const f32x2 aw = splat2(vq[6 * H + feat(c)]);
#pragma unroll
          for (int r = 0; r < 8; ++r) part2[r] = pk_fma(f32x2{acc[c][2 * r], acc[c][2 * r + 1]}, aw, part2[r]);
        }
        float part[16];
#pragma unroll
        for (int r = 0; r < 8; ++r) { part[2 * r] = part2[r].x; part[2 * r + 1] = part2[r].y; }
        // reduce-scatter over the half-wave: lane j ends with the dot product of accumulator register j >> 1,
        // takes ONE sigmoid, and the 16 gates of the half come back through 64 bytes of LDS (broadcast reads)
        const float gate = sigmoidf_fast(reduce16_half_wave(part, j) + att_b);
        s_phi[16 * half + (j >> 1)] = gate;
        wave_lds_fence();
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const float4 g4 = *reinterpret_cast<const float4*>(s_phi + 16 * half + 4 * q4);
          part[4 * q4] = g4.x; part[4 * q4 + 1] = g4.y; part[4 * q4 + 2] = g4.z; part[4 * q4 + 3] = g4.w;
        }
        wave_lds_fence();   // the words are rewritten by the next tile
#pragma unroll
        for (int c = 0; c < CT; ++c)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[c][r] *= part[r];       // mij * att, egnn_new.py:40
      }
      // Segmented sums over the tile's 32 rows.  Accumulator register rr of half h is row 8*(rr>>2) + 4*h + (rr&3):
      // the rows alternate between the halves in groups of 4.  Every half adds up ITS rows of the running segment in
      // edge order; when the segment ends (row ids are wave-uniform scalars: a scalar branch) the two halves' partial
      // sums are added (half 0's + half 1's) and each half stores the column tiles it received -- a fixed order that
      // depends on the tile's edges only.
      // aggregation protocol (edge_mlp.h): the first segment of the tile goes to agg_head[tile]
      // when its row continues from the previous wave tile, every other segment is the start of
      // its row and goes to agg[row]; plain stores, each address written by exactly one wave
      static_assert(CT % 2 == 0, "column tiles are exchanged in pairs");
      // Even and odd rows of a half run in separate sums (the two words of a register pair) that meet at the flush:
      // written as sum[c] += acc[c][r] the compiler pairs the adds ACROSS column tiles, whose accumulators are 16
      // registers apart -- two v_mov per packed add.
      f32x2 sum2[CT];
#pragma unroll
      for (int c = 0; c < CT; ++c) sum2[c] = splat2(0.f);
      int cur = -1;
      const int row0 = __builtin_amdgcn_readlane(my_r, 0);
      bool to_head = row0 >= 0 && row0 == __builtin_amdgcn_readfirstlane(my_prev);
      auto flush = [&]() {
        if (cur >= 0) {
          float* dst = to_head ? (my_lb ? p.agg_head_b : p.agg_head) + (size_t)my_wt * H
                               : (my_lb ? p.agg_b : p.agg) + (size_t)cur * H;
#pragma unroll
          for (int c = 0; c < CT / 2; ++c) {
            // half 0: tile c, half 1: tile c + CT/2
            const float tot = pair_sum_halves(sum2[c].x + sum2[c].y, sum2[c + CT / 2].x + sum2[c + CT / 2].y);
            dst[feat(c + half * (CT / 2))] = tot * inv_norm;
          }
          to_head = false;
        }
#pragma unroll
        for (int c = 0; c < CT; ++c) sum2[c] = splat2(0.f);
      };
#pragma unroll
      for (int gb = 0; gb < 8; ++gb) {
        const int hh = gb & 1;
#pragma unroll
        for (int ip = 0; ip < 4; ip += 2) {
          const int k = 4 * (gb >> 1) + ip;
          const int rn0 = __builtin_amdgcn_readlane(my_r, 4 * gb + ip);
          const int rn1 = __builtin_amdgcn_readlane(my_r, 4 * gb + ip + 1);
          if (rn0 != cur) {                                // scalar compare / branch
            flush();
            cur = rn0;
          }
          if (half == hh) {
#pragma unroll
            for (int c = 0; c < CT; ++c) sum2[c].x += acc[c][k];
          }
          if (rn1 != rn0) {
            flush();
            cur = rn1;
          }
          if (half == hh) {
#pragma unroll
            for (int c = 0; c < CT; ++c) sum2[c].y += acc[c][k + 1];
          }
        }
      }
      flush();
    } else {
      // scalar head: phi = w3 . SiLU(acc)   (egnn_new.py:80-92; the bias is already in the accumulators)
      f32x2 part2[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) part2[r] = splat2(0.f);
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        const f32x2 wv = splat2(vq[6 * H + feat(c)]);
#pragma unroll
        for (int r = 0; r < 8; ++r) part2[r] = pk_fma(silu2(f32x2{acc[c][2 * r], acc[c][2 * r + 1]}), wv, part2[r]);
      }
      float part[16];
#pragma unroll
      for (int r = 0; r < 8; ++r) { part[2 * r] = part2[r].x; part[2 * r + 1] = part2[r].y; }
      // lane j of a half ends with the total of accumulator register j >> 1 = edge mfma_row(j >> 1, lane)
      s_phi[mfma_row(j >> 1, lane)] = reduce16_half_wave(part, j);
      wave_lds_fence();
      const float ph = s_phi[j];                            // this lane's edge
      wave_lds_fence();
      if (qsel + q == 0) phi0 = ph; else phi1 = ph;

      if (tile_ends) {
        // trans = u*phi + cross*phi_x   (egnn_new.py:100-109, 296-316); lane = edge
        float tx = 0.f, ty = 0.f, tz = 0.f;
        if (my_r >= 0) {
          const float dx = xr[0] - xc[0], dy = xr[1] - xc[1], dz = xr[2] - xc[2];
          const float den = sqrtf(my_d + 1e-8f) + p.norm_constant;
          const float ux = dx / den, uy = dy / den, uz = dz / den;
          if (split && qsel == 1) {
            // this workgroup only adds the cross-product term
          } else if (p.use_tanh) {
            const float th = tanhf(phi0);
            tx = ux * th * p.coords_range; ty = uy * th * p.coords_range; tz = uz * th * p.coords_range;
          } else {
            tx = ux * phi0; ty = uy * phi0; tz = uz * phi0;
          }
          if (p.n_mlp == 2 && !(split && qsel == 0)) {
            const int b = p.node_batch[my_r];
            const float m0 = p.mean[3 * b], m1 = p.mean[3 * b + 1], m2 = p.mean[3 * b + 2];
            const float a0 = xr[0] - m0, a1 = xr[1] - m1, a2 = xr[2] - m2;
            const float b0 = xc[0] - m0, b1 = xc[1] - m1, b2 = xc[2] - m2;
            const float c0 = a1 * b2 - a2 * b1, c1 = a2 * b0 - a0 * b2, c2 = a0 * b1 - a1 * b0;
            const float cden = sqrtf(c0 * c0 + c1 * c1 + c2 * c2) + p.norm_constant;
            float phx = phi1;
            if (p.use_tanh) phx = tanhf(phx) * p.coords_range;
            tx += c0 / cden * phx; ty += c1 / cden * phx; tz += c2 / cden * phx;
          }
        }
        if (half == 0) { s_tr[3 * j] = tx; s_tr[3 * j + 1] = ty; s_tr[3 * j + 2] = tz; }
        wave_lds_fence();
        // all 32 values of this lane's component first (independent LDS reads, one wait), then the scalar walk over
        // the rows: read inside the walk, every step sat out an LDS round trip
        float trv[32];
#pragma unroll
        for (int e = 0; e < 32; ++e) trv[e] = s_tr[3 * e + (lane < 3 ? lane : 0)];
        if (lane < 3) {
          const int pop = split ? qsel : 0;
          float* xa = p.xagg + pop * p.xagg_stride;
          float* xh = p.xagg_head + pop * p.xhead_stride;
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
        wave_lds_fence();   // scratch is reused by the next tile
      }
    }

    // advance to the next unit
    if (tile_ends) {
      if (last_unit) break;
      commit_edge();
      li = next_li;
      q = 0;
    } else {
      ++q;
    }
  }  // units
}

}  // namespace dsbdd
