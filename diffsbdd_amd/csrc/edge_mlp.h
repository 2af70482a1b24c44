// Fused per-edge MLP kernels of one EquivariantBlock (egnn_new.py:163-184).
//
// Every edge MLP of the reference is  Linear(2H+A -> H) . SiLU . Linear(H -> H)
// . SiLU [. Linear(H -> 1)]  applied to cat[h_row, h_col, d_cur, d_0, emb(type)]
// (egnn_new.py:15-19,35,80-92,99,174).  The first Linear factorises exactly into
// per-node projections P = h @ W1[:, :H]^T and Q = h @ W1[:, H:2H]^T (computed
// once per node by node_linear.h), two rank-1 terms d_cur*wd + d_0*wd0 and a
// per-edge-type constant table[type] = b1 + W1[:, 2H+2:] @ emb[type]:
//
//     a1[e] = SiLU(P[row_e] + Q[col_e] + d_cur[e]*wd + d_0[e]*wd0 + table[type_e])
//
// so the per-edge work that is left is ONE H x H matmul per MLP.  That matmul
// runs here on the fp32 matrix cores: a tile of BM edges is gathered (coalesced
// 128-B row segments of P/Q), activated and written k-major into LDS, W2^T is
// streamed from L2 in K slices, and the workgroup's 4 waves (2 along the edge
// dimension x 2 along the feature dimension) accumulate a BM x H tile with
// v_mfma_f32_32x32x2_f32.  The epilogue never leaves the chip:
//
//   MODE_GCL   (GCL.edge_model + aggregation, egnn_new.py:31-52):
//       m = SiLU(. + b2); att = sigmoid(w_a . m + b_a); the BM x H message tile
//       goes to LDS, one thread per feature walks the (row-sorted) edges and
//       does a segmented sum in edge order -- the reference's scatter_add order
//       -- issuing one atomic per (row segment, feature); result / norm -> agg.
//   MODE_COORD (EquivariantUpdate.coord_model, egnn_new.py:96-122): the same main
//       loop once per scalar MLP (coord_mlp, cross_product_mlp), then
//       phi = tanh(w3 . SiLU(. + b2)) * range, trans = u*phi + cross*phi_x from
//       the coordinates, segmented sum per row -> xagg.
//
// Edges are sorted by (row, col) (dynamics.py:185), so a row's edges are
// contiguous: a row segment spans at most two tiles unless its degree exceeds
// BM, and the two partial sums commute -> the aggregation is deterministic.
// The kernels are persistent over tiles (the edge count lives in device memory:
// no host sync, fixed launch geometry for graph capture) with an XCD-aware
// tile order (each XCD walks a contiguous range of tiles = a few samples whose
// Q rows stay in that XCD's L2).
#pragma once
#include "common.h"

namespace dsbdd {

struct EdgeMlpW {
  const float* P;      // [N][ldpq] row projection (no bias)
  const float* Q;      // [N][ldpq] col projection
  const float* wd;     // [H]
  const float* wd0;    // [H]
  const float* table;  // [3][H]
  const float* W2T;    // [H][H]
  const float* b2;     // [H]
  // optional: W2T with the columns of every row regrouped per MFMA lane,
  // W2TP[k][j * (H/32) + c] = W2T[k][c * 32 + j], so that lane j finds the B values of all its
  // column tiles in consecutive words (16-byte LDS reads in edge_wave.h); nullptr = not available
  const float* W2TP;
};

struct EdgeArgs {
  const int* erow; const int* ecol; const float* ed0;  // [E]
  const int* e_count;     // device scalar: number of edges to process
  const float* x;         // [N][3] current coordinates
  int n_lig;              // nodes < n_lig are ligand nodes (edge types, dynamics.py:119-124)
  int ldpq;
  EdgeMlpW mlp[2];
  // MODE_GCL
  const float* att_w; const float* att_b; int attention;
  float* agg;             // [N][H], zero on entry
  // MODE_COORD
  const float* w3; const int* node_batch; const float* mean;  // mean [B][3]
  float norm_constant; float coords_range; int use_tanh; int n_mlp;
  float* xagg;            // [N][3], zero on entry
  float norm_factor;
  // edge_wave_kernel, MODE_COORD with two MLPs: alternate workgroups take the coordinate /
  // cross-product MLP of a tile (twice as many, half as long work items: better balance when
  // the masked edge prefix is only a few tiles per CU); the two terms of trans are linear
  int pass_split;
};

enum { MODE_GCL = 0, MODE_COORD = 1 };

template <int H, int BM, int BK, int MODE>
struct EdgeLayout {
  static constexpr int LDA = BM + 1;
  static constexpr int A_BUF = BK * LDA;           // floats per A buffer
  static constexpr int B_BUF = BK * H;             // floats per B buffer
  static constexpr int MAIN = 2 * (A_BUF + B_BUF);
  static constexpr int LDM = H + 1;
  static constexpr int EPI = (MODE == MODE_GCL) ? BM * LDM : BM * 65;
  static constexpr int REGION = MAIN > EPI ? MAIN : EPI;
  static constexpr int NV = (MODE == MODE_GCL) ? 1 : 2;  // resident MLP vector sets
  // after REGION: per-MLP vectors (wd, wd0, table[3], b2, w-out) then tile metadata
  static constexpr int VEC_OFF = REGION;
  static constexpr int VEC_PER = 7 * H;
  static constexpr int META_OFF = VEC_OFF + NV * VEC_PER;
  static constexpr int META = 15 * BM;            // 2 x (row,col,type,d,d0) | att/phi0, phi1 | trans[3]
  static constexpr int TOTAL = META_OFF + META;
};

template <int H, int BM, int BK, int MODE>
__global__ __launch_bounds__(kThreads) void edge_mlp_kernel(EdgeArgs p) {
  using L = EdgeLayout<H, BM, BK, MODE>;
  constexpr int LDA = L::LDA, LDM = L::LDM;
  constexpr int RT = BM / 64;        // 32-row tiles per wave (wave = BM/2 edges x H/2 features)
  constexpr int CT = H / 64;         // 32-col tiles per wave
  constexpr int KQ = BK / 4;         // float4 per edge per K slice
  constexpr int AI = BM * KQ / kThreads;
  constexpr int NQ = H / 4;          // float4 per W2T row
  constexpr int BI = BK * NQ / kThreads;
  constexpr int NK = H / BK;
  constexpr int TPR = kThreads / BM; // threads per edge in the row reductions
  static_assert(H % 64 == 0 && H <= 256, "hidden_nf must be 64,128,192 or 256");
  static_assert(BM == 64 || BM == 128, "BM");
  static_assert((BM * KQ) % kThreads == 0 && (BK * NQ) % kThreads == 0, "staging split");
  static_assert(H % BK == 0, "BK");

  __shared__ float smem[L::TOTAL];
  float* sA = smem;                         // [2][BK][LDA]
  float* sB = smem + 2 * L::A_BUF;          // [2][BK][H]
  float* sV = smem + L::VEC_OFF;            // per MLP: wd, wd0, tab0, tab1, tab2, b2, wout
  // tile metadata is double-buffered: the next tile's (row, col, type, d, d0) are
  // fetched while the current tile is in its main loop (hides two dependent
  // global-load latencies per tile)
  float* s_meta = smem + L::META_OFF;       // [2][5][BM]
  float* s_s0 = s_meta + 10 * BM;           // GCL: attention; COORD: phi (coord)
  float* s_s1 = s_meta + 11 * BM;           // COORD: phi (cross)
  float* s_tr = s_meta + 12 * BM;           // COORD: trans [BM][3]

  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int wm = w >> 1, wn = w & 1;
  const int n_pass = (MODE == MODE_GCL) ? 1 : p.n_mlp;

  // ---- resident small vectors ------------------------------------------------
  for (int q = 0; q < n_pass; ++q) {
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

  const int E = *p.e_count;
  const int ntiles = (E + BM - 1) / BM;
  // XCD-aware persistent schedule: XCD x (= blockIdx % 8) owns a contiguous tile range
  const int xcd = blockIdx.x & 7, kx = blockIdx.x >> 3, gx = gridDim.x >> 3;
  const int tq = ntiles / 8, tr = ntiles % 8;
  const int csize = tq + (xcd < tr ? 1 : 0);
  const int cbase = (xcd < tr) ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;

  // staging coordinates
  const int a_kq = (t % KQ) * 4, a_m = t / KQ;  // + (kThreads/KQ)*i

  // one thread per edge: fetch (row, col, d0) / coordinates / write LDS
  int nx_r = -1, nx_c = 0;
  float nx_d0 = 0.f, nx_xr[3] = {0.f, 0.f, 0.f}, nx_xc[3] = {0.f, 0.f, 0.f};
  auto meta_fetch_idx = [&](int tile_id) {
    nx_r = -1; nx_c = 0; nx_d0 = 0.f;
    const int e = tile_id * BM + t;
    if (t < BM && e < E) { nx_r = p.erow[e]; nx_c = p.ecol[e]; nx_d0 = p.ed0[e]; }
  };
  auto meta_fetch_x = [&]() {
    if (t < BM && nx_r >= 0) {
#pragma unroll
      for (int k = 0; k < 3; ++k) { nx_xr[k] = p.x[3 * nx_r + k]; nx_xc[k] = p.x[3 * nx_c + k]; }
    }
  };
  auto meta_store = [&](int buf) {
    if (t < BM) {
      float* mb = s_meta + buf * 5 * BM;
      float d = 0.f;
      int ty = 0;
      if (nx_r >= 0) {
        const float dx = nx_xr[0] - nx_xc[0], dy = nx_xr[1] - nx_xc[1], dz = nx_xr[2] - nx_xc[2];
        d = dx * dx + dy * dy + dz * dz;   // coord2diff radial, egnn_new.py:298-299
        const bool rl = nx_r < p.n_lig, cl = nx_c < p.n_lig;
        ty = (rl && cl) ? 1 : ((!rl && !cl) ? 2 : 0);  // dynamics.py:119-124
      }
      reinterpret_cast<int*>(mb)[t] = nx_r;
      reinterpret_cast<int*>(mb)[BM + t] = nx_c;
      reinterpret_cast<int*>(mb)[2 * BM + t] = ty;
      mb[3 * BM + t] = d;
      mb[4 * BM + t] = nx_d0;
    }
  };

  int mbuf = 0;
  if (kx < csize) {   // first tile of this workgroup: synchronous
    meta_fetch_idx(cbase + kx);
    meta_fetch_x();
    meta_store(0);
  }

  for (int li = kx; li < csize; li += gx, mbuf ^= 1) {
    const bool has_next = li + gx < csize;
    const int next_tile = cbase + li + gx;
    const int* s_row = reinterpret_cast<const int*>(s_meta + mbuf * 5 * BM);
    const int* s_col = s_row + BM;
    const int* s_typ = s_row + 2 * BM;
    const float* s_d = s_meta + mbuf * 5 * BM + 3 * BM;
    const float* s_d0 = s_d + BM;
    __syncthreads();  // previous tile's epilogue is done with LDS; metadata / sV visible

    for (int q = 0; q < n_pass; ++q) {
      const EdgeMlpW& mw = p.mlp[q];
      const float* vq = sV + q * L::VEC_PER;

      f32x16 acc[RT][CT];
#pragma unroll
      for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < CT; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

      float4 rp[AI], rq[AI], rb[BI];

      auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < AI; ++i) {
          const int m = a_m + (kThreads / KQ) * i;
          const int r = s_row[m], c = s_col[m];
          const int rr = r < 0 ? 0 : r;
          rp[i] = ld4(mw.P + (size_t)rr * p.ldpq + k0 + a_kq);
          rq[i] = ld4(mw.Q + (size_t)c * p.ldpq + k0 + a_kq);
        }
#pragma unroll
        for (int i = 0; i < BI; ++i) {
          const int idx = t + kThreads * i;       // float4 index inside the [BK][H] slice
          rb[i] = ld4(mw.W2T + (size_t)k0 * H + idx * 4);
        }
      };

      auto sstore = [&](int buf, int k0) {
        float* a = sA + buf * L::A_BUF;
        float* b = sB + buf * L::B_BUF;
#pragma unroll
        for (int i = 0; i < AI; ++i) {
          const int m = a_m + (kThreads / KQ) * i;
          const float d = s_d[m], d0 = s_d0[m];
          const float* tab = vq + (2 + s_typ[m]) * H + k0 + a_kq;
          const float* wd = vq + k0 + a_kq;
          const float* wd0 = vq + H + k0 + a_kq;
          const float pv[4] = {rp[i].x, rp[i].y, rp[i].z, rp[i].w};
          const float qv[4] = {rq[i].x, rq[i].y, rq[i].z, rq[i].w};
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const float pre = pv[c] + qv[c] + d * wd[c] + d0 * wd0[c] + tab[c];
            a[(a_kq + c) * LDA + m] = silu(pre);
          }
        }
#pragma unroll
        for (int i = 0; i < BI; ++i) {
          const int idx = t + kThreads * i;
          *reinterpret_cast<float4*>(b + idx * 4) = rb[i];
        }
      };

      gload(0);
      sstore(0, 0);
      __syncthreads();
#pragma unroll 1
      for (int kt = 0; kt < NK; ++kt) {
        if (q == 0 && has_next) {   // next tile's metadata, spread over the first K steps
          if (kt == 0) meta_fetch_idx(next_tile);
          if (kt == 1) meta_fetch_x();
          if (kt == (NK > 2 ? 2 : NK - 1)) meta_store(mbuf ^ 1);
        }
        if (kt + 1 < NK) gload((kt + 1) * BK);
        const float* pa = sA + (kt & 1) * L::A_BUF + (lane >> 5) * LDA + wm * (BM / 2) + (lane & 31);
        const float* pb = sB + (kt & 1) * L::B_BUF + (lane >> 5) * H + wn * (H / 2) + (lane & 31);
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
          float a[RT], b[CT];
#pragma unroll
          for (int i = 0; i < RT; ++i) a[i] = pa[kk * LDA + i * 32];
#pragma unroll
          for (int j = 0; j < CT; ++j) b[j] = pb[kk * H + j * 32];
#pragma unroll
          for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int j = 0; j < CT; ++j) acc[i][j] = mfma32(a[i], b[j], acc[i][j]);
        }
        if (kt + 1 < NK) sstore((kt + 1) & 1, (kt + 1) * BK);
        __syncthreads();
      }
      // all waves are past their last LDS read of sA/sB here.

      if (MODE == MODE_GCL) {
        // ---- messages -> LDS tile sM[BM][H+1] --------------------------------
        float* sM = smem;
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
          for (int j = 0; j < CT; ++j) {
            const int col = wn * (H / 2) + j * 32 + (lane & 31);
            const float bv = vq[5 * H + col];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int rl = wm * (BM / 2) + i * 32 + mfma_row(r, lane);
              sM[rl * LDM + col] = silu(acc[i][j][r] + bv);   // egnn_new.py:18-19
            }
          }
        __syncthreads();
        // ---- attention gate (egnn_new.py:26-29,38-40): TPR threads per edge ---
        {
          const int el = t / TPR, part = t % TPR;
          float att = 1.f;
          if (p.attention) {
            float dot = 0.f;
            const float* mrow = sM + el * LDM;
            const float* aw = vq + 6 * H;
            constexpr int CH = H / TPR;           // features per thread
#pragma unroll 8
            for (int i = 0; i < CH; ++i) {        // skewed start: conflict-free LDS banks
              const int k = part * CH + (i + part * (32 / TPR)) % CH;
              dot += mrow[k] * aw[k];
            }
#pragma unroll
            for (int o = 1; o < TPR; o <<= 1) dot += __shfl_xor(dot, o);
            att = sigmoidf_fast(dot + att_b);
          }
          if (part == 0) s_s0[el] = att;
        }
        __syncthreads();
        // ---- segmented sum over the row-sorted edges, one thread per feature --
        if (t < H) {   // whole waves only (H is a multiple of 64): the row id is wave-uniform
          int cur = -1;
          float sum = 0.f;
#pragma unroll 1
          for (int e0 = 0; e0 < BM; e0 += 16) {
            int rr[16];
            float v[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {          // 48 independent LDS reads in flight
              rr[j] = s_row[e0 + j];
              v[j] = sM[(e0 + j) * LDM + t] * s_s0[e0 + j];   // mij * att, egnn_new.py:40
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) {
              const int r = __builtin_amdgcn_readfirstlane(rr[j]);   // scalar compare/branch
              if (r != cur) {
                if (cur >= 0) unsafeAtomicAdd(&p.agg[(size_t)cur * H + t], sum / p.norm_factor);
                cur = r;
                sum = 0.f;
              }
              sum += v[j];   // rows of padding edges (r = -1) are summed but never flushed
            }
          }
          if (cur >= 0) unsafeAtomicAdd(&p.agg[(size_t)cur * H + t], sum / p.norm_factor);
        }
      } else {
        // ---- scalar head: phi = w3 . SiLU(acc + b2)  (egnn_new.py:80-92) ------
        float* sR = smem;  // [BM][65] partial dots: 64 column-lanes per edge
#pragma unroll
        for (int i = 0; i < RT; ++i) {
          float part[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) part[r] = 0.f;
#pragma unroll
          for (int j = 0; j < CT; ++j) {
            const int col = wn * (H / 2) + j * 32 + (lane & 31);
            const float bv = vq[5 * H + col], wv = vq[6 * H + col];
#pragma unroll
            for (int r = 0; r < 16; ++r) part[r] += silu(acc[i][j][r] + bv) * wv;
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int rl = wm * (BM / 2) + i * 32 + mfma_row(r, lane);
            sR[rl * 65 + wn * 32 + (lane & 31)] = part[r];
          }
        }
        __syncthreads();
        {
          const int el = t / TPR, part = t % TPR;
          float s = 0.f;
          constexpr int CH = 64 / TPR;
#pragma unroll
          for (int i = 0; i < CH; ++i) s += sR[el * 65 + part * CH + (i + part * (32 / TPR)) % CH];
#pragma unroll
          for (int o = 1; o < TPR; o <<= 1) s += __shfl_xor(s, o);
          if (part == 0) (q == 0 ? s_s0 : s_s1)[el] = s;
        }
        __syncthreads();
      }
    }  // passes

    if (MODE == MODE_COORD) {
      // ---- trans = u*phi + cross*phi_x  (egnn_new.py:100-109, 296-316) --------
      if (t < BM) {
        const int r = s_row[t];
        float tx = 0.f, ty = 0.f, tz = 0.f;
        if (r >= 0) {
          const int c = s_col[t];
          const float xr0 = p.x[3 * r], xr1 = p.x[3 * r + 1], xr2 = p.x[3 * r + 2];
          const float xc0 = p.x[3 * c], xc1 = p.x[3 * c + 1], xc2 = p.x[3 * c + 2];
          const float dx = xr0 - xc0, dy = xr1 - xc1, dz = xr2 - xc2;
          const float radial = dx * dx + dy * dy + dz * dz;
          // coord_diff = diff / (sqrt(radial + 1e-8) + norm_constant), egnn_new.py:300-301
          const float den = sqrtf(radial + 1e-8f) + p.norm_constant;
          const float ux = dx / den, uy = dy / den, uz = dz / den;
          const float phi = s_s0[t];
          if (p.use_tanh) {   // coord_diff * tanh(phi) * coords_range, egnn_new.py:101
            const float th = tanhf(phi);
            tx = ux * th * p.coords_range;
            ty = uy * th * p.coords_range;
            tz = uz * th * p.coords_range;
          } else {
            tx = ux * phi; ty = uy * phi; tz = uz * phi;
          }
          if (p.n_mlp == 2) {  // coord2cross, egnn_new.py:305-316
            const int b = p.node_batch[r];
            const float m0 = p.mean[3 * b], m1 = p.mean[3 * b + 1], m2 = p.mean[3 * b + 2];
            const float a0 = xr0 - m0, a1 = xr1 - m1, a2 = xr2 - m2;
            const float b0 = xc0 - m0, b1 = xc1 - m1, b2 = xc2 - m2;
            const float c0 = a1 * b2 - a2 * b1, c1 = a2 * b0 - a0 * b2, c2 = a0 * b1 - a1 * b0;
            const float cden = sqrtf(c0 * c0 + c1 * c1 + c2 * c2) + p.norm_constant;
            float phx = s_s1[t];
            if (p.use_tanh) phx = tanhf(phx) * p.coords_range;   // egnn_new.py:108
            tx += c0 / cden * phx; ty += c1 / cden * phx; tz += c2 / cden * phx;
          }
        }
        s_tr[3 * t] = tx; s_tr[3 * t + 1] = ty; s_tr[3 * t + 2] = tz;
      }
      __syncthreads();
      if (t < 3) {
        int cur = -1;
        float sum = 0.f;
#pragma unroll 1
        for (int e0 = 0; e0 < BM; e0 += 16) {
          int rr[16];
          float v[16];
#pragma unroll
          for (int j = 0; j < 16; ++j) { rr[j] = s_row[e0 + j]; v[j] = s_tr[3 * (e0 + j) + t]; }
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            const int r = rr[j];
            if (r != cur) {
              if (cur >= 0) unsafeAtomicAdd(&p.xagg[(size_t)cur * 3 + t], sum / p.norm_factor);
              cur = r;
              sum = 0.f;
            }
            sum += v[j];
          }
        }
        if (cur >= 0) unsafeAtomicAdd(&p.xagg[(size_t)cur * 3 + t], sum / p.norm_factor);
      }
    }
  }  // tiles
}

}  // namespace dsbdd
