// Dense per-node linear layer on the fp32 matrix cores:
//
//     C[M][N] = act( [A1 | A2][M][K1+K2] @ WT[K1+K2][N] + bias ) (+ R)
//
// Used for every node-level Linear of the denoiser: encoders/decoders
// (dynamics.py:27-49), embedding / embedding_out (egnn_new.py:212-213), the
// node MLP of GCL (egnn_new.py:21-24,56-57, input cat[h, agg] = the two A
// sources) and the per-node first-layer projections of the edge MLPs (the
// factorised form of Linear(cat[h_i, h_j, e])).
//
// Tiling: workgroup = 256 threads = 4 waves as 2(M) x 2(N); block tile
// BM x 128, K step 32; each wave owns (BM/2) x 64 outputs as RT x 2 MFMA
// 32x32 tiles (v_mfma_f32_32x32x2_f32, exact fp32).  Both operands sit in LDS
// k-major (sA[k][m], sB[k][n]) so that the 32 lanes of a half-wave read 32
// consecutive words (conflict-free ds_read_b32); the A tile is transposed on
// its way into LDS with row stride BM+1 (conflict-free ds_write_b32).
#pragma once
#include "common.h"

namespace dsbdd {

struct NodeLinearArgs {
  const float* A1; int lda1; int K1;
  const float* A2; int lda2; int K2;
  const float* WT; int ldw;
  const float* bias;
  const float* R; int ldr;
  float* C; int ldc;
  int M; int N; int act;
  // optional row gather: logical row m -> physical row row_idx[m] of A1/A2/R/C, with
  // the number of logical rows read from device memory (min(M, *m_count)); used for
  // the active-node subset of the coordinate-MLP projections
  const int* row_idx; const int* m_count;
};

template <int BM, bool VEC_A>
__global__ __launch_bounds__(kThreads) void node_linear_kernel(NodeLinearArgs p) {
  constexpr int BN = 128, BK = 32, LDA = BM + 1;
  constexpr int RT = BM / 64;  // 32-row MFMA tiles per wave
  constexpr int AI = BM / 32;  // float4 A loads per thread per K step
  __shared__ float sA[BK * LDA];
  __shared__ float sB[BK * BN];

  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int wm = w >> 1, wn = w & 1;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int K = p.K1 + p.K2;
  const int M = p.m_count ? min(p.M, *p.m_count) : p.M;
  if (m0 >= M) return;   // uniform per workgroup

  f32x16 acc[RT][2];
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float4 ra[AI], rb[4];
  const int a_kq = (t & 7) * 4, a_m = t >> 3;   // A: 8 lanes cover 32 consecutive k of one row
  const int b_n = (t & 31) * 4, b_k = t >> 5;   // B: 32 lanes cover 128 consecutive n of one k

  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const int ml = m0 + a_m + 32 * i, k = k0 + a_kq;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ml < M) {
        const int m = p.row_idx ? p.row_idx[ml] : ml;
        if (VEC_A) {
          if (k < K) {
            const float* src = (k < p.K1) ? p.A1 + (size_t)m * p.lda1 + k
                                          : p.A2 + (size_t)m * p.lda2 + (k - p.K1);
            v = ld4(src);
          }
        } else {
          float e[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const int kk = k + c;
            e[c] = 0.f;
            if (kk < K)
              e[c] = (kk < p.K1) ? p.A1[(size_t)m * p.lda1 + kk]
                                 : p.A2[(size_t)m * p.lda2 + (kk - p.K1)];
          }
          v = make_float4(e[0], e[1], e[2], e[3]);
        }
      }
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k = k0 + b_k + 8 * i, n = n0 + b_n;
      // WT rows are padded to ldw (multiple of 4) -> a float4 never leaves the row
      rb[i] = (k < K && n < p.ldw) ? ld4(p.WT + (size_t)k * p.ldw + n)
                                   : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };

  auto sstore = [&]() {
#pragma unroll
    for (int i = 0; i < AI; ++i) {
      const int m = a_m + 32 * i;
      sA[(a_kq + 0) * LDA + m] = ra[i].x;
      sA[(a_kq + 1) * LDA + m] = ra[i].y;
      sA[(a_kq + 2) * LDA + m] = ra[i].z;
      sA[(a_kq + 3) * LDA + m] = ra[i].w;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
      *reinterpret_cast<float4*>(&sB[(b_k + 8 * i) * BN + b_n]) = rb[i];
  };

  const int nk = (K + BK - 1) / BK;
  gload(0);
  for (int kt = 0; kt < nk; ++kt) {
    sstore();
    __syncthreads();
    if (kt + 1 < nk) gload((kt + 1) * BK);  // in flight during the MFMAs
    // k pairing of one MFMA: (k, k + 4) inside every group of 8 -- the same summation order as
    // node_gemm_kernel below, so that the choice between the two kernels (made from the problem
    // size) never changes a result bit
    const float* pa = sA + 4 * (lane >> 5) * LDA + wm * (BM / 2) + (lane & 31);
    const float* pb = sB + 4 * (lane >> 5) * BN + wn * 64 + (lane & 31);
#pragma unroll
    for (int g = 0; g < BK / 8; ++g) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int kk = 8 * g + q;
        float a[RT], b[2];
#pragma unroll
        for (int i = 0; i < RT; ++i) a[i] = pa[kk * LDA + i * 32];
#pragma unroll
        for (int j = 0; j < 2; ++j) b[j] = pb[kk * BN + j * 32];
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = mfma32(a[i], b[j], acc[i][j]);
      }
    }
    __syncthreads();
  }

  // epilogue: bias, activation, residual, store (32 consecutive columns per half-wave)
#pragma unroll
  for (int i = 0; i < RT; ++i) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn * 64 + j * 32 + (lane & 31);
      if (col >= p.N) continue;
      const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rowl = m0 + wm * (BM / 2) + i * 32 + mfma_row(r, lane);
        if (rowl >= M) continue;
        const int row = p.row_idx ? p.row_idx[rowl] : rowl;
        float v = acc[i][j][r] + bv;
        if (p.act == 1) v = silu(v);
        if (p.R) v += p.R[(size_t)row * p.ldr + col];
        p.C[(size_t)row * p.ldc + col] = v;
      }
    }
  }
}


// ---------------------------------------------------------------------------
// Same GEMM with the structure of edge_wave.h, for the aligned big layers
// (K1, K2 multiples of BK; N multiple of 32*CT; 16-byte aligned rows):
//   * lane l IS row (l & 31) of the wave's 32-row tile and reads float4 chunks of its
//     own row straight from L2 -- the MFMA A operand never touches LDS.  The chunks of
//     a whole K step are requested one K step ahead (register double buffer), so an
//     L1 miss (every row opens a new 128-byte line every 32 k) has 64 MFMAs to land;
//   * the weight slice [BK][32*CT] is streamed by global_load_lds (double buffered, one
//     barrier per K step); workgroup = 4 waves x 32 rows = 128 rows x 32*CT columns;
//   * K step per tile width: node_bk() below (short steps measured best);
//   * up to three independent problems share one launch (blockIdx.z): the small
//     ligand-row / active-subset projections of the coordinate MLPs ride along with
//     the next block's P|Q projection instead of running alone on a fraction of the CUs.
constexpr int kMaxGroup = 3;
struct NodeGroupArgs { NodeLinearArgs p[kMaxGroup]; };

// K step (rows of a weight slice) per tile width.  Measured with the balanced schedule, same box, ligands/s:
// 64 / 64: 34.83, 32 / 32: 35.10, 16 / 32: 35.21, 128 / 64: 32.7 -- short steps (16 MFMAs per wave between
// barriers, 8 KB of LDS per workgroup) beat long ones here: the launches are a few tiles per CU, so what counts is
// how soon a workgroup's first MFMA issues and how many workgroups a CU holds, not the barrier count.
#ifndef DSBDD_NODE_BK2
#define DSBDD_NODE_BK2 16        // 64-column tile
#endif
#ifndef DSBDD_NODE_BK1
#define DSBDD_NODE_BK1 32        // 32-column (half) tile
#endif
constexpr int node_bk(int ct) { return ct == 1 ? DSBDD_NODE_BK1 : (ct == 2 ? DSBDD_NODE_BK2 : 128 / ct); }

// One 128-row x (32*CT)-column output tile, computed by the calling workgroup.
template <int CT>
__device__ __forceinline__ void node_gemm_tile(const NodeLinearArgs& p, const int m0, const int n0, const int M,
                                               float* sB) {
  constexpr int BN = 32 * CT, BK = node_bk(CT), NG = BK / 8;
  constexpr int BI = BK * BN / 4 / kThreads;          // float4 DMA pieces per thread per slice
  constexpr int RQ = BN / 4;                          // float4 per slice row
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int half = lane >> 5, j = lane & 31;
  const int K = p.K1 + p.K2;

  auto streamB = [&](int ks, int buf) {
#ifdef DSBDD_DIAG_NODE_NODMA
    return;   // DIAGNOSTIC ONLY (wrong results): the weight slices are never streamed
#endif
#pragma unroll
    for (int i = 0; i < BI; ++i) {
      const int f = t + kThreads * i;                // float4 index: row f/RQ, column chunk f%RQ
      const float* src = p.WT + (size_t)(ks * BK + f / RQ) * p.ldw + n0 + (f % RQ) * 4;
      float* dst = sB + buf * BK * BN + (w * 64 + kThreads * i) * 4;   // wave-uniform base
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
  };

  const int rowl = m0 + w * 32 + j;
  const int row = rowl < M ? (p.row_idx ? p.row_idx[rowl] : rowl) : 0;
  const float* a1 = p.A1 + (size_t)row * p.lda1 + 4 * half;
  const float* a2 = p.K2 ? p.A2 + (size_t)row * p.lda2 + 4 * half : a1;
  auto loadA = [&](int ks, float4 (&dst)[NG]) {
    const int k0 = ks * BK;                          // a K step never straddles A1 | A2
    const float* src = k0 < p.K1 ? a1 + k0 : a2 + (k0 - p.K1);
#ifdef DSBDD_DIAG_NODE_NOA
#pragma unroll
    for (int g = 0; g < NG; ++g) dst[g] = make_float4(0.1f, 0.2f, 0.3f, (float)ks);   // DIAGNOSTIC ONLY (wrong results)
    (void)src;
#else
#pragma unroll
    for (int g = 0; g < NG; ++g) dst[g] = ld4(src + 8 * g);
#endif
  };

  f32x16 acc[CT];
#pragma unroll
  for (int c = 0; c < CT; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;

  float4 cur[NG], nxt[NG];
  streamB(0, 0);
  loadA(0, cur);
#pragma unroll
  for (int g = 0; g < NG; ++g) nxt[g] = cur[g];
  __syncthreads();
  const int nk = K / BK;
#pragma unroll 1
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 1 < nk) {
      streamB(kt + 1, (kt + 1) & 1);
      loadA(kt + 1, nxt);
    }
    const float* bcur = sB + (kt & 1) * BK * BN + (4 * half) * BN + j;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const float a[4] = {cur[g].x, cur[g].y, cur[g].z, cur[g].w};
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float* brow = bcur + (8 * g + i) * BN;
#pragma unroll
        for (int c = 0; c < CT; ++c) acc[c] = mfma32(a[i], brow[c * 32], acc[c]);
      }
      __builtin_amdgcn_s_setprio(0);
    }
#pragma unroll
    for (int g = 0; g < NG; ++g) cur[g] = nxt[g];
#ifdef DSBDD_DIAG_NODE_NOBARRIER
    __builtin_amdgcn_s_waitcnt(0x0f70);   // DIAGNOSTIC ONLY (racy): vmcnt(0) without the workgroup barrier
#else
    __syncthreads();
#endif
  }

  // epilogue.  C may alias R (the node MLP's residual is updated in place), so the compiler
  // must keep every R load behind the previous C store: gather the row ids and all residual
  // values of a column tile first, then store -- CT memory round trips instead of 16 * CT.
  int ro[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int rl = m0 + w * 32 + mfma_row(r, lane);
    ro[r] = rl < M ? (p.row_idx ? p.row_idx[rl] : rl) : -1;
  }
#pragma unroll
  for (int c = 0; c < CT; ++c) {
    const int col = n0 + c * 32 + j;
    const float bv = p.bias ? p.bias[col] : 0.f;
    f32x16 res;
#pragma unroll
    for (int r = 0; r < 16; ++r) res[r] = (p.R && ro[r] >= 0) ? p.R[(size_t)ro[r] * p.ldr + col] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (ro[r] < 0) continue;
      float v = acc[c][r] + bv;
      if (p.act == 1) v = silu(v);
      p.C[(size_t)ro[r] * p.ldc + col] = v + res[r];
    }
  }
}


// Tile schedule.  The workgroup ids of a problem are one-dimensional: first the FULL tiles (128 rows x 32*CT columns,
// column tile fastest), then the row tiles at the end of the matrix as HALF-width tiles (32*CT/2 columns, two per
// full tile).  The full tiles are the largest multiple of `balance` (two per CU) that fits: a problem of 620 full
// tiles on 256 CUs (2.42 per CU: some CUs would run 3, most 2) becomes 512 full + 216 half tiles = at most 2.5 per CU.  Every output element is computed by exactly one workgroup with the same k order in both tile
// shapes, so the schedule never changes a bit of the result.  (A problem with fewer tiles than CUs runs entirely on
// half tiles: twice as many, half as long workgroups -- what the 32-column heuristic did for small launches.)
template <int CT>
__global__ __launch_bounds__(kThreads, (CT == 4 ? 3 : 4)) void node_gemm_kernel(NodeGroupArgs ga, int balance) {
  constexpr int BN = 32 * CT, BK = node_bk(CT);
  constexpr int LDSF = 2 * BK * BN > 2 * node_bk(CT > 1 ? CT / 2 : 1) * (BN / 2) ? 2 * BK * BN
                                                                                 : 2 * node_bk(CT > 1 ? CT / 2 : 1) * (BN / 2);
  __shared__ float sB[LDSF];                          // room for either tile shape
  const NodeLinearArgs& p = ga.p[blockIdx.z];
  const int M = p.m_count ? min(p.M, *p.m_count) : p.M;
  const int m_tiles = (M + 127) / 128, gy = p.N / BN;
  int n_full = m_tiles;                               // row tiles computed as full-width tiles
  if (CT > 1 && balance > 0) n_full = (m_tiles * gy / balance) * balance / gy;
  const int id = blockIdx.x;
  if (id < n_full * gy) {
    node_gemm_tile<CT>(p, (id / gy) * 128, (id % gy) * BN, M, sB);
  } else if (CT > 1) {
    const int id2 = id - n_full * gy, rt = n_full + id2 / (2 * gy);
    if (rt >= m_tiles) return;                        // uniform per workgroup
    node_gemm_tile<(CT > 1 ? CT / 2 : 1)>(p, rt * 128, (id2 % (2 * gy)) * (BN / 2), M, sB);
  }
}

// ---------------------------------------------------------------------------
// The tiny two-layer MLPs at both ends of the denoiser (atom / residue encoders and decoders, dynamics.py:27-49:
// 10..128 -> 20..40 -> 10..128 features): out = W1 . SiLU(W0 . in + b0) + b1 per node, both layers in one launch on the
// vector ALUs, up to two independent problems (ligand nodes, pocket nodes) per launch.  They are a few MFLOP; as four
// separate GEMM launches they cost four launch latencies.
struct Mlp2Problem {
  const float* in; int ld_in; int K0;      // [rows][K0]
  const float* W0T; int ldw0; const float* b0; int M1;   // [K0][ldw0], hidden width M1
  const float* W1T; int ldw1; const float* b1; int n_out;   // [M1][ldw1]
  float* out; int ld_out; int rows;
};
struct Mlp2Args { Mlp2Problem p[2]; };
constexpr int kMlp2Rows = 32, kMlp2MaxK = 128, kMlp2MaxMid = 40, kMlp2MaxOut = 128;

__global__ __launch_bounds__(kThreads) void mlp2_kernel(Mlp2Args a) {
  __shared__ float sW0[kMlp2MaxK * kMlp2MaxMid];
  __shared__ float sW1[kMlp2MaxMid * kMlp2MaxOut];
  __shared__ float sIn[kMlp2Rows * (kMlp2MaxK + 1)];
  __shared__ float sMid[kMlp2Rows * (kMlp2MaxMid + 1)];
  const Mlp2Problem& p = a.p[blockIdx.y];
  const int t = threadIdx.x, r0 = blockIdx.x * kMlp2Rows;
  if (r0 >= p.rows) return;
  const int nr = min(kMlp2Rows, p.rows - r0);
  for (int i = t; i < p.K0 * p.M1; i += kThreads) sW0[i] = p.W0T[(size_t)(i / p.M1) * p.ldw0 + i % p.M1];
  for (int i = t; i < p.M1 * p.n_out; i += kThreads) sW1[i] = p.W1T[(size_t)(i / p.n_out) * p.ldw1 + i % p.n_out];
  for (int i = t; i < nr * p.K0; i += kThreads)
    sIn[(i / p.K0) * (kMlp2MaxK + 1) + i % p.K0] = p.in[(size_t)(r0 + i / p.K0) * p.ld_in + i % p.K0];
  __syncthreads();
  for (int i = t; i < nr * p.M1; i += kThreads) {
    const int r = i / p.M1, j = i % p.M1;
    float v = p.b0[j];
    const float* x = sIn + r * (kMlp2MaxK + 1);
    for (int k = 0; k < p.K0; ++k) v = fmaf(x[k], sW0[k * p.M1 + j], v);
    sMid[r * (kMlp2MaxMid + 1) + j] = silu(v);
  }
  __syncthreads();
  for (int i = t; i < nr * p.n_out; i += kThreads) {
    const int r = i / p.n_out, c = i % p.n_out;
    float v = p.b1[c];
    const float* m = sMid + r * (kMlp2MaxMid + 1);
    for (int j = 0; j < p.M1; ++j) v = fmaf(m[j], sW1[j * p.n_out + c], v);
    p.out[(size_t)(r0 + r) * p.ld_out + c] = v;
  }
}

inline bool mlp2_fits(const Mlp2Problem& p) {
  return p.K0 <= kMlp2MaxK && p.M1 <= kMlp2MaxMid && p.n_out <= kMlp2MaxOut;
}

inline hipError_t launch_mlp2(hipStream_t s, const Mlp2Problem* p, int n) {
  Mlp2Args a{};
  int rows = 0;
  for (int i = 0; i < n; ++i) { a.p[i] = p[i]; rows = max(rows, p[i].rows); }
  if (rows <= 0) return hipSuccess;
  hipLaunchKernelGGL(mlp2_kernel, dim3((rows + kMlp2Rows - 1) / kMlp2Rows, n), dim3(kThreads), 0, s, a);
  return hipGetLastError();
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

inline bool vec_ok(const NodeLinearArgs& a) {
  return aligned16(a.A1) && (a.lda1 % 4 == 0) && (a.K1 % 4 == 0) &&
         (a.K2 == 0 || (aligned16(a.A2) && (a.lda2 % 4 == 0) && (a.K2 % 4 == 0)));
}

// column-tile count (1, 2 or 4) of the register-A kernel for this problem, 0 = not eligible
inline int gemm_ct(const NodeLinearArgs& a, int forced_ct, bool allow_small = true) {
  if (!vec_ok(a) || a.ldw % 4 != 0 || !aligned16(a.WT) || a.K1 <= 0) return 0;
  auto fits = [&](int ct) {
    const int bk = node_bk(ct), bn = 32 * ct;
    return a.K1 % bk == 0 && a.K2 % bk == 0 && a.N % bn == 0;
  };
  if (forced_ct && fits(forced_ct)) return forced_ct;
  // few rows (e.g. the C-alpha workloads, M ~ 2k): 32-column tiles make twice as many, half as long
  // workgroups -- the launch is bounded by one workgroup's latency, not by throughput
  if (allow_small && (long)((a.M + 127) / 128) * (a.N / 64) < 128 && fits(1)) return 1;
  // 64-column tiles also for the wide projections (N = 512 / 1024): with the balanced tile schedule they measured
  // 34.77 vs 34.54 ligands/s against 128-column tiles
  return fits(2) ? 2 : (fits(4) ? 4 : 0);
}

inline int node_ct_override() {
  static const int v = [] {
    const char* s = getenv("DSBDD_NODE_CT");
    return s ? atoi(s) : 0;
  }();
  return v;
}

// Granularity of the full-tile count in node_gemm_kernel's tile schedule: two workgroups per CU (measured 34.51 /
// 34.56 / 34.60 ligands/s for 0.5 / 1 / 2 per CU; 4 per CU -- nearly everything on half tiles -- 34.07; off 33.37).
// DSBDD_NODE_BALANCE overrides (0: full-width tiles only).
inline int node_balance() {
  static const int v = [] {
    const char* s = getenv("DSBDD_NODE_BALANCE");
    if (s) return atoi(s);
    int dev = 0, cu = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cu <= 0)
      return 0;
    int p = 1;                       // the schedule needs a power of two (it must be a multiple of the column tiles)
    while (2 * p <= cu) p *= 2;
    return 2 * p;
  }();
  return v;
}

// One launch for up to kMaxGroup eligible problems (all must accept the same CT).
// Returns hipErrorInvalidValue when the group cannot share a launch.
inline hipError_t launch_node_group(hipStream_t s, const NodeLinearArgs* a, int n) {
  if (n <= 0 || n > kMaxGroup) return hipErrorInvalidValue;
  // 32-column tiles only when every problem of the group is small
  bool all_small = true;
  for (int i = 0; i < n; ++i) all_small = all_small && gemm_ct(a[i], node_ct_override()) == 1;
  int ct = 4;
  for (int i = 0; i < n; ++i) {
    const int c = gemm_ct(a[i], node_ct_override(), all_small);
    if (c == 0) return hipErrorInvalidValue;
    if (c < ct) ct = c;
  }
  for (int i = 0; i < n; ++i)
    if (gemm_ct(a[i], ct) != ct) return hipErrorInvalidValue;
  NodeGroupArgs ga{};
  int balance = ct > 1 ? node_balance() : 0;
  for (int i = 0; i < n && balance > 0; ++i)          // the half-width tiles are the ct / 2 kernel's: same constraints
    if (gemm_ct(a[i], ct / 2) != ct / 2 || balance % (a[i].N / (32 * ct)) != 0) balance = 0;
  int gx = 0;
  for (int i = 0; i < n; ++i) {
    ga.p[i] = a[i];
    // one-dimensional workgroup ids per problem (node_gemm_kernel): worst case every row tile as half-width tiles
    gx = max(gx, ((a[i].M + 127) / 128) * (a[i].N / (32 * ct)) * (balance > 0 ? 2 : 1));
  }
  dim3 grid(gx, 1, n), block(kThreads);
  if (ct == 4)      hipLaunchKernelGGL((node_gemm_kernel<4>), grid, block, 0, s, ga, balance);
  else if (ct == 2) hipLaunchKernelGGL((node_gemm_kernel<2>), grid, block, 0, s, ga, balance);
  else              hipLaunchKernelGGL((node_gemm_kernel<1>), grid, block, 0, s, ga, balance);
  return hipGetLastError();
}

// Host-side launcher.  Returns hipError_t of the launch.
inline hipError_t launch_node_linear(hipStream_t s, const NodeLinearArgs& a) {
  if (a.M <= 0 || a.N <= 0) return hipSuccess;
  const bool vec = vec_ok(a);
  // aligned big layers: register-A / LDS-DMA kernel
  if ((long)a.M * a.N >= 64 * 1024 && gemm_ct(a, node_ct_override()) != 0) return launch_node_group(s, &a, 1);
  const int ny = (a.N + 127) / 128;
  const long tiles128 = (long)((a.M + 127) / 128) * ny;
  const bool big = tiles128 >= 512 && !a.row_idx;  // enough 128-row tiles to fill 256 CUs twice
  const int bm = big ? 128 : 64;
  dim3 grid((a.M + bm - 1) / bm, ny), block(kThreads);
  if (big) {
    if (vec) hipLaunchKernelGGL((node_linear_kernel<128, true>), grid, block, 0, s, a);
    else     hipLaunchKernelGGL((node_linear_kernel<128, false>), grid, block, 0, s, a);
  } else {
    if (vec) hipLaunchKernelGGL((node_linear_kernel<64, true>), grid, block, 0, s, a);
    else     hipLaunchKernelGGL((node_linear_kernel<64, false>), grid, block, 0, s, a);
  }
  return hipGetLastError();
}

}  // namespace dsbdd
