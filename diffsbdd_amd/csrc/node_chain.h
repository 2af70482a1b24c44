// Row-owning node-phase kernel: the node MLP of a GCL (egnn_new.py:21-24,48-58) and the per-node first-layer
// projections that follow it (P|Q of the next message stage, Q / P of the coordinate MLPs -- the factorised
// Linear(cat[h_i, h_j, e]), edge_mlp.h) in ONE launch, with the intermediate activations of a row never leaving the CU:
//
//     t1 = SiLU([h | agg] W1 + b1)        K = 2H   input rows streamed global -> LDS (64-k chunks, LDS-DMA)
//     h  = h + t1 W2 + b2                 K = H    t1 lives in LDS
//     C_p = h W_p   (p = 0 .. n_proj-1)   K = H    the new h lives in LDS; problem p covers the rows [0, *count_p)
//
// Why a new kernel (node_linear.h's register-A GEMM measured 0.29 - 0.46 of the fp32 matrix peak on these launches,
// profiles/r3a_microbench.md): three short launches per block with a fixed cost of ~10 us each, 128 x 64 tiles whose
// count never fits the 256 CUs, a workgroup barrier every 16 MFMAs.  Here
//   * a workgroup (8 waves, one per CU) OWNS a range of rows through all stages -- no grid-wide dependency, one launch;
//   * rows are dealt out in 16-row tiles (v_mfma_f32_16x16x4_f32), cost-weighted: a row that also gets the coordinate
//     projections counts more, so that every CU ends at the same time; the split is computed on the device from the
//     row counts (device scalars, no host sync) -- 19.8 k rows on 256 CUs = 77 rows each instead of 2.42 -> 3 tiles;
//   * the 8 waves split the output COLUMNS (16-column tiles); the row panel is the shared operand, in LDS as
//     [k / 4][row][4 floats] so that a lane's ds_read_b128 delivers four k steps and 16 lanes read 256 contiguous bytes;
//   * weights stream from L2 in a lane-major packed layout (pack_b16_kernel): one coalesced 1-KiB global_load_dwordx4
//     per wave and 16 k, prefetched one group ahead; the K loop contains no VALU work at all (tools/mfma_shadow.hip:
//     any VALU between fp32 MFMAs costs its full issue time) and, from stage 2 on, no barrier;
//   * the MFMA runs "transposed" (A operand = weights, B operand = rows): lane l then holds four consecutive output
//     COLUMNS of row (l & 15) -- exactly the 16-byte unit of the LDS panel and of the global row stores.
//
// Every output element is one fmaf chain over k in ascending 16-k groups, independent of the row split, the number of
// workgroups and the batch composition (bitwise contract of DESIGN.md §5).
#pragma once
#include "common.h"

namespace dsbdd {

typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifdef DSBDD_CHAIN_TS
// -DDSBDD_CHAIN_TS builds (tools/microbench.hip only): marks of wave 0 of every workgroup -- slots 0..7 shader clock
// (s_memtime), 16..23 the constant 100 MHz clock, 8..10 the range (first row, rows, projection passes)
__device__ unsigned long long* g_chain_ts = nullptr;
__device__ __forceinline__ void chain_ts(int slot) {
  if (g_chain_ts && threadIdx.x == 0 && slot < 16) {
    g_chain_ts[blockIdx.x * 32 + slot] = __builtin_readcyclecounter();
    g_chain_ts[blockIdx.x * 32 + 16 + slot] = wall_clock64();
  }
}
#define CHAIN_TS(k) chain_ts(k)
#define CHAIN_NOTE(slot, v) do { if (g_chain_ts && threadIdx.x == 0) g_chain_ts[blockIdx.x * 32 + (slot)] = (unsigned long long)(v); } while (0)
#else
#define CHAIN_TS(k) do { } while (0)
#define CHAIN_NOTE(slot, v) do { } while (0)
#endif

constexpr int kChainThreads = 512;     // 8 waves: one workgroup per CU, two waves per SIMD
constexpr int kChainRowsMax = 96;      // rows a workgroup holds in LDS at a time (6 row tiles)
constexpr int kChainChunkK = 64;       // k per streamed input chunk of stage 1
constexpr int kChainMaxProj = 3;

// Lane-major packed weights for the MFMA A operand: for the 16-column tile ct and the 16-k group g,
// dst[((ct * (K/16) + g) * 64 + lane) * 4 + s] = WT[16 g + 4 (lane >> 4) + s][16 ct + (lane & 15)]
// (WT = [in][out] as the weight slots store it, row stride ldw).
__global__ void pack_b16_kernel(const float* WT, int ldw, int K, int N, float* dst) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= K * N) return;
  const int s = idx & 3, lane = (idx >> 2) & 63, rest = idx >> 8;
  const int g = rest % (K / 16), ct = rest / (K / 16);
  dst[idx] = WT[(size_t)(16 * g + 4 * (lane >> 4) + s) * ldw + 16 * ct + (lane & 15)];
}

struct ChainProj {
  const float* Wp;       // packed weights of this problem's column range: [N/16][H/16][64][4]
  float* C; int ldc;     // output rows (physical row ids), this problem's first column
  int N;                 // columns, multiple of 128 (16 per wave and pass)
  const int* count;      // device scalar: the problem covers `*count` rows of the row list ...; nullptr: all that follow
  int first;             // ... starting at logical row `first` (the ghost rows in front of a level list are skipped)
};

struct NodeChainArgs {
  const int* row_idx;    // logical -> physical row; nullptr: identity
  const int* m_count;    // device scalar: number of logical rows (min with M); nullptr: M
  int M;
  int do_mlp;            // 0: projections only (their input h is read from global memory)
  float* h;              // [.][H] node features: input, residual and (do_mlp) output, in place
  const float* agg;      // [.][H] completed message aggregate
  const float* W1p; const float* b1;     // packed [H cols][2H k]
  const float* W2p; const float* b2;     // packed [H cols][H k]
  int n_proj;
  ChainProj proj[kChainMaxProj];
};

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// One pass of a K loop whose row operand sits in LDS panels: acc[rt][t] += W(ct0 + t) . X(rt) over k = 0 .. 16 * NG.
// xs: this lane's base inside the panel buffer ((kq * R16 + i) * 4 floats), panel stride 4 * R16 floats per k / 4;
// wp: this lane's position in the packed stream of column tile ct0 (consecutive tiles: wstride floats apart).
template <int RT, int CT>
__device__ __forceinline__ void chain_mfma_group(f32x4 (&acc)[RT][CT], const float4 (&wv)[CT], const float4 (&xv)[RT]) {
  __builtin_amdgcn_s_setprio(1);
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int rt = 0; rt < RT; ++rt)
#pragma unroll
      for (int t = 0; t < CT; ++t) {
        const float a = s == 0 ? wv[t].x : (s == 1 ? wv[t].y : (s == 2 ? wv[t].z : wv[t].w));
        const float b = s == 0 ? xv[rt].x : (s == 1 ? xv[rt].y : (s == 2 ? xv[rt].z : xv[rt].w));
        acc[rt][t] = mfma16(a, b, acc[rt][t]);
      }
  __builtin_amdgcn_s_setprio(0);
}

// K loop with the row operand in LDS panels and the weights streamed from L2.
// Two register sets alternate, one group ahead (weights: an L2 round trip; rows: LDS latency), the last pair of groups is
// peeled: no conditional load inside the loop, so every s_waitcnt the compiler inserts sits one whole MFMA block behind
// the loads it waits for.  NG must be even.  Measured and not adopted (profiles/README.md): a ring of four sets with the
// weights requested three or four groups ahead (10 - 15 % slower at every row count, also for one row tile -- the loop is
// not waiting for the L2: a 16-row launch runs at the MFMA time of one row tile on one CU), the first group of a loop
// requested before the preceding epilogue, the L2 warmed, a k-group-major weight layout, de-phased waves; round 4: the
// next chunk's first two weight groups requested BEFORE the chunk barrier of stage 1 (172.3 -> 174.3 us at 19.8 k rows,
// profiles/r4d_node_chain_xbar.md).
// CHUNKED (stage 1): the row operand of group g lives in chunk buffer (g / 4) & 1, which the workgroup's LDS-DMA fills;
// `next_chunk(c)` issues the DMA of chunk c + 1 at the start of chunk c and every chunk ends with a workgroup barrier
// (staging registers instead of the DMA, the remedy of the edge kernels, measured slower here: rows are 16-byte pieces of
// 64 different cache lines per instruction).
template <int RT, int CT, bool CHUNKED, class XOf, class Hook>
__device__ __forceinline__ void chain_kloop(f32x4 (&acc)[RT][CT], XOf&& x_of, const float* wp, size_t wstride, int NG,
                                            Hook&& next_chunk) {
  float4 w0[CT], w1[CT], x0[RT], x1[RT];
  auto load = [&](int g, float4 (&wv)[CT], float4 (&xv)[RT]) {
#ifdef DSBDD_DIAG_CHAIN_NOW
    if (g == 0)
#endif
#pragma unroll
    for (int t = 0; t < CT; ++t) wv[t] = ld4(wp + t * wstride + (size_t)g * 256);
    const float* xs = x_of(g);
#ifdef DSBDD_DIAG_CHAIN_NOX
    if (g == 0)
#endif
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) xv[rt] = *reinterpret_cast<const float4*>(xs + 64 * rt);
  };
  if (!CHUNKED) {
    load(0, w0, x0);
#pragma unroll 1
    for (int g = 0; g + 2 < NG; g += 2) {
      load(g + 1, w1, x1);
      chain_mfma_group<RT, CT>(acc, w0, x0);
      load(g + 2, w0, x0);
      chain_mfma_group<RT, CT>(acc, w1, x1);
    }
    load(NG - 1, w1, x1);
    chain_mfma_group<RT, CT>(acc, w0, x0);
    chain_mfma_group<RT, CT>(acc, w1, x1);
  } else {
#pragma unroll 1
    for (int g = 0; g < NG; g += 4) {                      // one chunk = 4 groups
      next_chunk(g >> 2);
      load(g, w0, x0);
      load(g + 1, w1, x1);
      chain_mfma_group<RT, CT>(acc, w0, x0);
      load(g + 2, w0, x0);
      chain_mfma_group<RT, CT>(acc, w1, x1);
      load(g + 3, w1, x1);
      chain_mfma_group<RT, CT>(acc, w0, x0);
      chain_mfma_group<RT, CT>(acc, w1, x1);
      __syncthreads();
    }
  }
}

// All stages for one range of RT row tiles starting at logical row r0.
#ifndef DSBDD_CHAIN_WHOLE
#define DSBDD_CHAIN_WHOLE 1
#endif
#ifndef DSBDD_CHAIN_PRE
#define DSBDD_CHAIN_PRE 1
#endif

template <int H, int RT>
__device__ __forceinline__ void chain_range(const NodeChainArgs& p, const int r0, const int M, float* smem,
                                            const int* pcount) {
  constexpr int R16 = 16 * RT;
  constexpr int kLdsFloats = kChainRowsMax * H + 2 * (kChainChunkK / 4) * kChainRowsMax * 4;
  // few rows: the whole [h | agg] input of the range fits behind the panels -> one DMA burst, one barrier, no chunking
  constexpr bool WHOLE = DSBDD_CHAIN_WHOLE && R16 * 3 * H <= kLdsFloats;
  float* sPanel = smem;                                // [H / 4][R16][4]
  float* sChunk = smem + R16 * H;                      // WHOLE: [2H / 4][R16][4]; else 2 chunk buffers
  constexpr int CTW = H >= 192 ? 2 : 1;              // 16-column tiles per wave in the H-column stages; H / (16 CTW) waves
                                                     // (all 8 for H = 256 / 128, 6 for H = 192, 4 for H = 64) do the MFMAs
  constexpr int NG = H / 16;                         // 16-k groups of an H-deep stage
  constexpr int CHP = kChainChunkK / 4;              // panels per streamed chunk
  static_assert(H % 64 == 0 && H <= 256, "hidden_nf");
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int i = lane & 15, kq = lane >> 4;
  // physical rows this lane serves: as a DMA lane (rows r0 + lane, r0 + 64 + lane) and as an MFMA lane (r0 + 16 rt + i)
  auto phys = [&](int logical) {
    const int c = logical < M ? logical : M - 1;     // rows past the end read a valid row; their outputs are masked
    return p.row_idx ? p.row_idx[c] : c;
  };
  const float* xs = sPanel + (kq * R16 + i) * 4;
  const bool col_active = w * CTW * 16 < H;          // this wave owns columns of the H-column stages

  // ---- global -> LDS panel copy of K columns starting at column k0 of src (row-gathered): wave w takes panels w, w + 8 ..
  auto dma_panels = [&](const float* src, int k0, int n_panels, float* dst) {
    const int ra = phys(r0 + lane), rb = phys(r0 + 64 + lane);
    for (int q = w; q < n_panels; q += 8) {
      if (R16 >= 64 || lane < R16)      // (inactive lanes of an LDS-DMA instruction write nothing)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)ra * H + k0 + 4 * q),
                                         (__attribute__((address_space(3))) void*)(dst + q * R16 * 4), 16, 0, 0);
      if (R16 > 64 && lane < R16 - 64)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)rb * H + k0 + 4 * q),
                                         (__attribute__((address_space(3))) void*)(dst + (q * R16 + 64) * 4), 16, 0, 0);
    }
  };
  // (rows 0 .. 63 of a panel are one wave instruction: lane = row, LDS destination = wave-uniform base + 16 * lane)

  int rowp[RT];
#pragma unroll
  for (int rt = 0; rt < RT; ++rt) rowp[rt] = phys(r0 + 16 * rt + i);

  CHAIN_TS(1);
  CHAIN_NOTE(8, r0); CHAIN_NOTE(9, R16);
  int n_pass = 0;
  (void)n_pass;
  if (p.do_mlp) {
    {
      // ================= stage 1: t1 = SiLU([h | agg] W1 + b1) =================
      f32x4 acc[RT][CTW];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int c = 0; c < CTW; ++c) acc[rt][c] = f32x4{0.f, 0.f, 0.f, 0.f};
      constexpr int NCH = 2 * H / kChainChunkK;
      const float* wp = p.W1p + ((size_t)(w * CTW) * (2 * H / 16) * 64 + lane) * 4;
      const size_t wstride = (size_t)(2 * H / 16) * 256;
      auto chunk_src = [&](int c) { return c * kChainChunkK < H ? p.h : p.agg; };
      auto chunk_k0 = [&](int c) { return c * kChainChunkK < H ? c * kChainChunkK : c * kChainChunkK - H; };
      if constexpr (WHOLE) {
        dma_panels(p.h, 0, H / 4, sChunk);
        dma_panels(p.agg, 0, H / 4, sChunk + (H / 4) * R16 * 4);
        __syncthreads();
        const float* xb = sChunk + (kq * R16 + i) * 4;
        auto x_all = [&](int g) { return xb + (size_t)g * 16 * R16; };
        auto no_hook = [](int) {};
        if (col_active) chain_kloop<RT, CTW, false>(acc, x_all, wp, wstride, 2 * H / 16, no_hook);
        __syncthreads();
      } else {
        constexpr int CB = CHP * R16 * 4;                  // floats per chunk buffer
        dma_panels(chunk_src(0), chunk_k0(0), CHP, sChunk);
        __syncthreads();
        const float* xb = sChunk + (kq * R16 + i) * 4;
        auto x_of = [&](int g) { return xb + ((g >> 2) & 1) * CB + (g & 3) * 16 * R16; };
        auto next_chunk = [&](int c) {
          if (c + 1 < NCH) dma_panels(chunk_src(c + 1), chunk_k0(c + 1), CHP, sChunk + ((c + 1) & 1) * CB);
        };
        if (col_active) {
          chain_kloop<RT, CTW, true>(acc, x_of, wp, wstride, 2 * H / 16, next_chunk);
        } else {                                           // waves without columns still feed the chunks and the barriers
#pragma unroll 1
          for (int c = 0; c < NCH; ++c) {
            next_chunk(c);
            __syncthreads();
          }
        }
      }
      CHAIN_TS(2);                                         // stage-1 K loop done
      // bias + SiLU -> LDS panels (float4 = four consecutive columns of one row)
#pragma unroll
      for (int c = 0; c < (col_active ? CTW : 0); ++c) {
        const int col = 16 * (w * CTW + c) + 4 * kq;
        const float4 bv = ld4(p.b1 + col);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          float4 v;
          v.x = silu(acc[rt][c][0] + bv.x); v.y = silu(acc[rt][c][1] + bv.y);
          v.z = silu(acc[rt][c][2] + bv.z); v.w = silu(acc[rt][c][3] + bv.w);
          *reinterpret_cast<float4*>(sPanel + ((col >> 2) * R16 + 16 * rt + i) * 4) = v;
        }
      }
      __syncthreads();
      CHAIN_TS(3);                                         // epilogue 1 + barrier
      // ================= stage 2: h += t1 W2 + b2 =================
#pragma unroll
      for (int rt = 0; rt < RT; ++rt)
#pragma unroll
        for (int c = 0; c < CTW; ++c) acc[rt][c] = f32x4{0.f, 0.f, 0.f, 0.f};
      auto x_panel = [&](int g) { return xs + (size_t)g * 16 * R16; };
      auto no_hook = [](int) {};
      // residual rows: with few row tiles requested before the K loop and consumed after it (register budget)
      constexpr bool PRE = DSBDD_CHAIN_PRE && RT <= 4;
      float4 hn[RT][CTW];
      auto load_res = [&]() {
#pragma unroll
        for (int c = 0; c < (col_active ? CTW : 0); ++c)
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) hn[rt][c] = ld4(p.h + (size_t)rowp[rt] * H + 16 * (w * CTW + c) + 4 * kq);
      };
      if (PRE) load_res();
      if (col_active)
        chain_kloop<RT, CTW, false>(acc, x_panel, p.W2p + ((size_t)(w * CTW) * NG * 64 + lane) * 4, (size_t)NG * 256, NG, no_hook);
      if (!PRE) load_res();
      CHAIN_TS(4);                                         // stage-2 K loop done
#pragma unroll
      for (int c = 0; c < (col_active ? CTW : 0); ++c) {
        const int col = 16 * (w * CTW + c) + 4 * kq;
        const float4 bv = ld4(p.b2 + col);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
          float4 v = hn[rt][c];
          v.x += acc[rt][c][0] + bv.x; v.y += acc[rt][c][1] + bv.y;
          v.z += acc[rt][c][2] + bv.z; v.w += acc[rt][c][3] + bv.w;
          hn[rt][c] = v;
          if (r0 + 16 * rt + i < M) *reinterpret_cast<float4*>(p.h + (size_t)rowp[rt] * H + col) = v;
        }
      }
      if (p.n_proj > 0) {
        __syncthreads();                                   // every wave has finished reading t1
#pragma unroll
        for (int c = 0; c < (col_active ? CTW : 0); ++c) {
          const int col = 16 * (w * CTW + c) + 4 * kq;
#pragma unroll
          for (int rt = 0; rt < RT; ++rt) *reinterpret_cast<float4*>(sPanel + ((col >> 2) * R16 + 16 * rt + i) * 4) = hn[rt][c];
        }
        __syncthreads();
      }
    }
  } else if (p.n_proj > 0) {
    dma_panels(p.h, 0, H / 4, sPanel);
    __syncthreads();
  }

  CHAIN_TS(5);                                             // epilogue 2 (+ h panels for the projections)
  // ================= projections from the rows' h in LDS =================
  for (int q = 0; q < p.n_proj; ++q) {
    const ChainProj& pj = p.proj[q];
    const int lo = pj.first, cnt = pj.first + pcount[q];   // the problem's logical rows [lo, cnt)
    if (r0 >= cnt || r0 + R16 <= lo) continue;             // uniform per workgroup
    const int rt_n = min(RT, (cnt - r0 + 15) >> 4);        // row tiles of this range the problem covers
    for (int ct0 = 2 * w; ct0 < pj.N / 16; ct0 += 16) {    // passes of 2 column tiles per wave
      ++n_pass;
      f32x4 acc[RT][2];
#pragma unroll
      for (int rt = 0; rt < RT; ++rt) { acc[rt][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[rt][1] = acc[rt][0]; }
      const float* wp = pj.Wp + ((size_t)ct0 * NG * 64 + lane) * 4;
      // (all RT row tiles are multiplied: the tiles past rt_n cost MFMA time only at the single range that straddles
      //  the problem's end; their results are not stored)
      auto x_panel = [&](int g) { return xs + (size_t)g * 16 * R16; };
      auto no_hook = [](int) {};
      chain_kloop<RT, 2, false>(acc, x_panel, wp, (size_t)NG * 256, NG, no_hook);
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const int col = 16 * (ct0 + c) + 4 * kq;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt)
          if (rt < rt_n && r0 + 16 * rt + i < cnt && r0 + 16 * rt + i >= lo)
            *reinterpret_cast<float4*>(pj.C + (size_t)rowp[rt] * pj.ldc + col) =
                make_float4(acc[rt][c][0], acc[rt][c][1], acc[rt][c][2], acc[rt][c][3]);
      }
    }
  }
  CHAIN_TS(6);                                             // projections done
  CHAIN_NOTE(10, n_pass);
  __syncthreads();                                         // the panels are rewritten by the next range
  CHAIN_TS(7);
}

template <int H>
__global__ __launch_bounds__(kChainThreads, 2) void node_chain_kernel(NodeChainArgs p) {
  __shared__ float smem[kChainRowsMax * H + 2 * (kChainChunkK / 4) * kChainRowsMax * 4];
  CHAIN_TS(0);
  const int M = p.m_count ? min(p.M, *p.m_count) : p.M;
  if (M <= 0) return;
  // ---- cost-weighted split of the row list into ranges of 16-row tiles (identical arithmetic in every workgroup) ----
  // cost of a row, in units of H*H/16 MAC: 48 for the node MLP + N_p / H * 16 for every projection that covers it.
  // All of it in 32-bit arithmetic on wave-uniform values (rows x 240 units < 2^31: launch_node_chain rejects more than 8 M
  // rows): round 3's 64-bit version -- bisection of the cumulative cost, 64-bit divisions -- cost every workgroup 7 us
  // before its first load (in-kernel timeline, profiles/r4h_node_chain_timeline.md).
  int cnt[kChainMaxProj], fst[kChainMaxProj];
  unsigned wgt[kChainMaxProj];
  const unsigned w_mlp = p.do_mlp ? 48u : 0u;
  int M_work = p.do_mlp ? M : 0;                           // rows behind the last problem's end have no work
#pragma unroll
  for (int q = 0; q < kChainMaxProj; ++q) {
    cnt[q] = 0; wgt[q] = 0; fst[q] = 0;
    if (q < p.n_proj) {
      fst[q] = min(M, p.proj[q].first);
      cnt[q] = p.proj[q].count ? min(M - fst[q], *p.proj[q].count) : M - fst[q];
      if (cnt[q] < 0) cnt[q] = 0;
      wgt[q] = (unsigned)(p.proj[q].N * 16 / H);
      M_work = max(M_work, fst[q] + cnt[q]);
    }
  }
  const int Mw = M_work;
  if (Mw <= 0) return;
  CHAIN_NOTE(11, (unsigned long long)(Mw > 0 ? wall_clock64() : 0));     // row counts loaded
  // cumulative cost of the logical rows [0, r)
  auto cost_to = [&](int r) {
    unsigned f = w_mlp * (unsigned)min(r, M_work);
#pragma unroll
    for (int q = 0; q < kChainMaxProj; ++q) f += wgt[q] * (unsigned)max(0, min(r - fst[q], cnt[q]));
    return f;
  };
  // F is piecewise linear: its slope only changes where a problem starts or ends (<= 8 breakpoints, sorted here)
  constexpr int NB = 2 * kChainMaxProj + 3;
  int bpv[NB];
  unsigned Fb[NB];
  {
    int nb = 0;
    bpv[nb++] = 0;
#pragma unroll
    for (int q = 0; q < kChainMaxProj; ++q) {
      bpv[nb++] = q < p.n_proj ? min(Mw, fst[q]) : Mw;
      bpv[nb++] = q < p.n_proj ? min(Mw, fst[q] + cnt[q]) : Mw;
    }
    bpv[nb++] = min(Mw, p.do_mlp ? M : Mw);
    bpv[nb++] = Mw;
#pragma unroll
    for (int a2 = 1; a2 < NB; ++a2)                        // insertion sort, fully unrolled (the arrays stay in registers)
#pragma unroll
      for (int b2 = NB - 1; b2 > 0; --b2)
        if (b2 <= a2 && bpv[b2] < bpv[b2 - 1]) { const int tmp = bpv[b2]; bpv[b2] = bpv[b2 - 1]; bpv[b2 - 1] = tmp; }
#pragma unroll
    for (int a2 = 0; a2 < NB; ++a2) Fb[a2] = cost_to(bpv[a2]);
  }
  const unsigned total = Fb[NB - 1];
  if (total == 0) return;
  // the cheapest row of the list = the smallest positive slope of a piece
  unsigned w_min = ~0u;
#pragma unroll
  for (int a2 = 0; a2 < NB - 1; ++a2) {
    const int len = bpv[a2 + 1] - bpv[a2];
    if (len > 0 && Fb[a2 + 1] > Fb[a2]) w_min = min(w_min, (Fb[a2 + 1] - Fb[a2]) / (unsigned)len);
  }
  if (w_min == ~0u || w_min == 0) w_min = 1;
  // number of ranges: a multiple of the grid, large enough that a range of the cheapest rows fits kChainRowsMax (the
  // boundaries are rounded UP to tiles, so a range has floor or ceil of (rows per range / 16) tiles, never more)
  const unsigned per_max = (unsigned)kChainRowsMax * w_min;
  unsigned V = (total + per_max - 1) / per_max;
  V = (V + gridDim.x - 1) / gridDim.x * gridDim.x;
  // first row (multiple of 16) whose cumulative cost reaches y: F inverted piece by piece, one 32-bit division
  auto row_at = [&](unsigned y) {
    if (y == 0) return 0;
    if (y > total) return (Mw + 15) / 16 * 16;
    // the piece [b[a], b[a + 1]) with F(b[a]) < y <= F(b[a + 1]): the first a whose end reaches y (F is non-decreasing,
    // F(b[0]) = 0 < y); pieces of zero length or zero slope can never be it
    int base = bpv[NB - 2], bnext = bpv[NB - 1];
    unsigned fbase = Fb[NB - 2], fnext = Fb[NB - 1];
    bool found = false;
#pragma unroll
    for (int a2 = 0; a2 < NB - 1; ++a2)
      if (!found && Fb[a2 + 1] >= y) { found = true; base = bpv[a2]; bnext = bpv[a2 + 1]; fbase = Fb[a2]; fnext = Fb[a2 + 1]; }
    const int len = bnext - base;
    int r = bnext;
    if (len > 0 && fnext > fbase) {
      const unsigned slope = (fnext - fbase) / (unsigned)len;   // exact: the cost per row is constant inside a piece
      r = base + (int)((y - fbase + slope - 1) / slope);
    }
    return (r + 15) / 16 * 16;
  };
  // Forced cuts (round 4): a range that STRADDLES the end of a projection problem multiplies all its row tiles for a
  // projection that covers some of them -- it was the slowest workgroup of every launch (171 vs 153 us in the timeline).
  // The row list is therefore cut at every problem's start (rounded down to a tile) and end (rounded up), and every
  // segment between two cuts gets its share of the V ranges by cost and splits them with the same inversion of F.
  // Only for launches of >= 3 row tiles per workgroup: below that the ranges are 1 - 2 tiles, nothing straddles for long, and
  // the extra arithmetic (1.5 us) is what shows (measured: 19.8 k rows 165.5 -> 153.9 us, 11.5 k 109.6 -> 111.5, 3.6 k 54.9 -> 57.8).
  constexpr int NC = 2 * kChainMaxProj + 3;
  const int Tt = (Mw + 15) >> 4;
  const bool forced = Tt >= 3 * (int)gridDim.x;
  const unsigned tq = total / V, trem = total % V;         // ceil(total v / V) = tq v + ceil(trem v / V): no 64-bit division
  auto target = [&](unsigned v) { return tq * v + (trem * v + V - 1) / V; };
  int cut[NC];
  unsigned fcut[NC], nrg[NC];                              // F at the cuts; ranges of the segment that starts at cut a
  unsigned Vs = V;
  if (forced) {
  {
    int nc = 0;
    cut[nc++] = 0;
#pragma unroll
    for (int q = 0; q < kChainMaxProj; ++q) {
      cut[nc++] = (q < p.n_proj && cnt[q] > 0) ? min(Tt, fst[q] >> 4) : Tt;
      cut[nc++] = (q < p.n_proj && cnt[q] > 0) ? min(Tt, (fst[q] + cnt[q] + 15) >> 4) : Tt;
    }
    cut[nc++] = p.do_mlp ? min(Tt, (M + 15) >> 4) : Tt;
    cut[nc++] = Tt;
#pragma unroll
    for (int a2 = 1; a2 < NC; ++a2)
#pragma unroll
      for (int b2 = NC - 1; b2 > 0; --b2)
        if (b2 <= a2 && cut[b2] < cut[b2 - 1]) { const int tmp = cut[b2]; cut[b2] = cut[b2 - 1]; cut[b2 - 1] = tmp; }
  }
#pragma unroll
  for (int a2 = 0; a2 < NC; ++a2) fcut[a2] = cost_to(min(Mw, cut[a2] << 4));
  Vs = 0;
  {
    int big = 0; unsigned cbig = 0;
#pragma unroll
    for (int a2 = 0; a2 < NC - 1; ++a2) {
      const unsigned c = fcut[a2 + 1] - fcut[a2];
      const int tiles = cut[a2 + 1] - cut[a2];
      unsigned n = 0;
      if (tiles > 0) {
        n = (unsigned)((float)c * (float)V / (float)total + 0.5f);
        n = max(n, (unsigned)((tiles + kChainRowsMax / 16 - 1) / (kChainRowsMax / 16)));   // every range fits the LDS panels
        n = min(max(n, 1u), (unsigned)tiles);
        if (c > cbig) { cbig = c; big = a2; }
      }
      nrg[a2] = n; Vs += n;
    }
    nrg[NC - 1] = 0;
    // the ranges left over (or borrowed) by the rounding go to (come from) the most expensive segment
#pragma unroll
    for (int a2 = 0; a2 < NC - 1; ++a2)
      if (a2 == big && Vs != V) {
        const int tiles = cut[a2 + 1] - cut[a2];
        const int lo_n = (tiles + kChainRowsMax / 16 - 1) / (kChainRowsMax / 16);
        int n = (int)nrg[a2] + (int)V - (int)Vs;
        n = min(max(n, max(lo_n, 1)), tiles);
        Vs = Vs - nrg[a2] + (unsigned)n;
        nrg[a2] = (unsigned)n;
      }
  }
  }  // forced
  for (unsigned v = blockIdx.x; v < Vs; v += gridDim.x) {
    int r0, r1;
    if (!forced) {
      r0 = v == 0 ? 0 : row_at(target(v));
      r1 = v + 1 == V ? (Mw + 15) / 16 * 16 : row_at(target(v + 1));
    } else {
    // segment and position of range v
    unsigned j = v, ns = 1, f0 = 0, cseg = 0;
    int b0 = 0, b1 = Tt;
    bool found = false;
#pragma unroll
    for (int a2 = 0; a2 < NC - 1; ++a2) {
      if (!found && j < nrg[a2]) { found = true; ns = nrg[a2]; f0 = fcut[a2]; cseg = fcut[a2 + 1] - fcut[a2]; b0 = cut[a2]; b1 = cut[a2 + 1]; }
      if (!found) j -= nrg[a2];
    }
    if (!found) break;
    const unsigned sq = cseg / ns, srem = cseg % ns;       // ceil(cseg j / ns) = sq j + ceil(srem j / ns)
    auto seg_target = [&](unsigned jj) { return f0 + sq * jj + (srem * jj + ns - 1) / ns; };
    r0 = j == 0 ? (b0 << 4) : min(max(row_at(seg_target(j)), b0 << 4), b1 << 4);
    r1 = j + 1 == ns ? (b1 << 4) : min(max(row_at(seg_target(j + 1)), b0 << 4), b1 << 4);
    }
    CHAIN_NOTE(12, (unsigned long long)(r1 >= r0 ? wall_clock64() : 0));   // range known
    int nt = (r1 - r0) / 16;
    int rs = r0;
    while (nt > 0) {                                       // (at most one piece unless the rounding overshoots)
      const int take = nt > kChainRowsMax / 16 ? kChainRowsMax / 16 : nt;
      switch (take) {
        case 1: chain_range<H, 1>(p, rs, Mw, smem, cnt); break;
        case 2: chain_range<H, 2>(p, rs, Mw, smem, cnt); break;
        case 3: chain_range<H, 3>(p, rs, Mw, smem, cnt); break;
        case 4: chain_range<H, 4>(p, rs, Mw, smem, cnt); break;
        case 5: chain_range<H, 5>(p, rs, Mw, smem, cnt); break;
        default: chain_range<H, 6>(p, rs, Mw, smem, cnt); break;
      }
      rs += 16 * take; nt -= take;
    }
  }
}

inline hipError_t launch_node_chain(hipStream_t s, const NodeChainArgs& a, int H, int n_cu) {
  if (a.M <= 0) return hipSuccess;
  if (a.M > (8 << 20)) return hipErrorInvalidValue;        // the split's 32-bit cost arithmetic (rows x 240 units)
  dim3 grid(n_cu), block(kChainThreads);
  switch (H) {
    case 256: hipLaunchKernelGGL((node_chain_kernel<256>), grid, block, 0, s, a); break;
    case 192: hipLaunchKernelGGL((node_chain_kernel<192>), grid, block, 0, s, a); break;
    case 128: hipLaunchKernelGGL((node_chain_kernel<128>), grid, block, 0, s, a); break;
    case 64: hipLaunchKernelGGL((node_chain_kernel<64>), grid, block, 0, s, a); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace dsbdd
