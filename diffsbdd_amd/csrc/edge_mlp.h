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
// so the per-edge work that is left is ONE H x H matmul per MLP, which runs on the
// fp32 matrix cores (v_mfma_f32_32x32x2_f32) in edge_wave.h.  The epilogue never
// leaves the chip:
//
//   MODE_GCL   (GCL.edge_model + aggregation, egnn_new.py:31-52):
//       m = SiLU(. + b2); att = sigmoid(w_a . m + b_a); segmented sums of m * att over
//       the (row-sorted) edges of a 32-edge wave tile, in edge order -- the reference's
//       scatter_add order -- scaled by 1 / normalization_factor.
//   MODE_COORD (EquivariantUpdate.coord_model, egnn_new.py:96-122): the same main
//       loop once per scalar MLP (coord_mlp, cross_product_mlp), then
//       phi = tanh(w3 . SiLU(. + b2)) * range, trans = u*phi + cross*phi_x from
//       the coordinates, segmented sum per row.
//
// Aggregation protocol (no atomics, no zero-filled accumulators, fixed summation order):
// edges are sorted by (row, col) (dynamics.py:185), so a row's edges are contiguous.
// Wave tile T = edges [32T, 32T + 32).  The partial sum of a row's edges inside a tile goes
//   * to agg[row] (xagg[row])      when the tile holds the row's FIRST edge,
//   * to agg_head[T] (xagg_head[T]) when the row continues from tile T - 1 (only the first
//     segment of a tile can),
// each written by exactly one wave with a plain store.  agg_complete_kernel (below) and
// coord_update_kernel then form  agg[row] + agg_head[T0 + 1] + ... + agg_head[T1]  in tile
// order, T0 / T1 = tiles of the row's first / last edge (from row_ptr and deg).  Together with
// the 32-aligned (sample, node set) segments of the edge list (graph.h) every per-row sum is a
// pure function of that sample's data: results are bitwise reproducible and independent of the
// batch composition / sharding.
//
// The kernels are persistent over 128-edge workgroup tiles (the edge count lives in device
// memory: no host sync, fixed launch geometry for graph capture); each XCD owns a contiguous
// range of tiles (a few samples whose Q rows stay in that XCD's L2), walked round-robin by its
// workgroups.
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
  // optional: the lane-grouped copy of the 16-edge-granule kernel (edge_wave16.h), W2TP16[k][16 n + c] = W2T[k][16 c + n]
  const float* W2TP16;
  // optional: the three bf16 planes of W2T in the operand layout of v_mfma_f32_32x32x16_bf16 (edge_wave.h, emulated
  // path; pack_w2e_kernel): [H/16 k steps][H/32 column tiles][3 planes][64 lanes][8 bf16] = 6 H^2 bytes
  const void* W2E;
  // optional: the per-wave rotated, lane-grouped copy of the split-K kernel (edge_splitk.h; pack_w2sk_kernel), H^2 floats
  const float* W2SK;
};

struct EdgeArgs {
  const int* erow; const int* ecol; const float* ed0;  // [E]
  const int* e_count;     // device scalar: number of edges to process
  int e_cap;              // capacity of erow/ecol/ed0: the kernels never index past it, even when
                          // the radius graph overflowed (status bit 1 is then set by edges_kernel)
  const float* x;         // [N][3] current coordinates
  int n_lig;              // nodes < n_lig are ligand nodes (edge types, dynamics.py:119-124)
  int n_nodes;            // rows of P / Q / x / agg (ghost rows included): list entries outside [0, n_nodes) are treated
                          // as inactive -- after an edge-capacity overflow parts of a list were never written
  int ldpq;
  EdgeMlpW mlp[2];
  // MODE_GCL
  const float* att_w; const float* att_b; int attention;
  float* agg;             // [N][H]: partial sum of the tile holding the row's first edge
  float* agg_head;        // [wave tiles][H]: partial sums of rows continuing from the previous tile
  // MODE_COORD
  const float* w3; const int* node_batch; const float* mean;  // mean [B][3]
  float norm_constant; float coords_range; int use_tanh; int n_mlp;
  float* xagg;            // [n_q][N][3]   (n_q = 2 with pass_split: one copy per MLP population)
  float* xagg_head;       // [n_q][wave tiles][4]
  size_t xagg_stride;     // floats between the two copies of xagg / xagg_head
  size_t xhead_stride;
  float norm_factor;
  int wt_base;            // global index of this launch's first wave tile (the list pointers may start inside the list)
  int* tile_ctr;          // work-queue counters, all zero between launches (graph.h: kTileCtrInts)
  // -DDSBDD_TIMESTAMPS builds only: [64 workgroups][16 marks] of wall_clock64() (100 MHz) for this launch
  unsigned long long* ts;
  // edge_wave_kernel, MODE_COORD with two MLPs: alternate workgroups take the coordinate /
  // cross-product MLP of a tile (twice as many, half as long work items: better balance when
  // the masked edge prefix is only a few tiles per CU); the two terms of trans are linear
  int pass_split;
  // MODE_GCL, optional second edge list of the same stage (e_count_b != nullptr): its 128-edge tiles follow the first
  // list's in the launch's tile space, its messages go to agg_b / agg_head_b.  One launch instead of two when a stage
  // walks two lists (block 0 of a framed call: ligand-endpoint edges + the frame's pocket-pocket edges).
  const int* erow_b; const int* ecol_b; const float* ed0_b; const int* e_count_b; int e_cap_b; int wt_base_b;
  float* agg_b; float* agg_head_b;
  // edge_wave_kernel<.., STORE = true> (training forward, single list from slot 0, no pass split): the second layer's
  // pre-activation z2 = W2 a1 + b2 of every list slot, [e_cap][H] (coordinate stage: one copy per MLP) -- the backward
  // pass reads it instead of recomputing the H x H layer (train.h, edge_bwd_e_kernel / edge_bwd_ec_kernel)
  float* z2_out;
  size_t z2_stride;       // MODE_COORD: floats between the two MLPs' copies of z2
};

enum { MODE_GCL = 0, MODE_COORD = 1 };

// agg[row] <- agg[row] + agg_head[T0 + 1] + ... + agg_head[T1] (tile order); rows without edges
// become 0.  One wave per row, 16-byte lanes; rows that live in a single tile (about half of them
// at degree ~17) are left untouched.  Costs what the zero fill of the atomic version cost.
// `shift`: log2 of the wave-tile size of the edge kernel that wrote the partial sums (5: edge_wave.h, 4: edge_wave16.h).
__global__ __launch_bounds__(kThreads) void agg_complete_kernel(float* agg, const float* agg_head,
                                                                const int* row_ptr, const int* deg,
                                                                int n_rows, int H, int max_tile, int shift) {
  const int row = (blockIdx.x * kThreads + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (row >= n_rows) return;
  const int d = deg[row], s = row_ptr[row];
  // (max_tile: last slot of agg_head -- after an edge-capacity overflow row_ptr describes edges that were never stored)
  const int t0 = s >> shift, t1 = min((s + d - 1) >> shift, max_tile);
  if (d > 0 && t1 == t0) return;
  for (int k = 4 * lane; k < H; k += 256) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (d > 0) {
      v = ld4(agg + (size_t)row * H + k);
      for (int T = t0 + 1; T <= t1; ++T) {
        const float4 h = ld4(agg_head + (size_t)T * H + k);
        v.x += h.x; v.y += h.y; v.z += h.z; v.w += h.w;
      }
    }
    *reinterpret_cast<float4*>(agg + (size_t)row * H + k) = v;
  }
}


// Block 0 of a pocket-conditioned chain ("pocket frame", engine.hip): the aggregate of a node is
//     [ A: sum over its edges with a ligand endpoint ]  +  [ B: sum over its pocket-pocket edges ]
// A comes from the message stage on the ligand-endpoint list (agg / head_a, rows indexed by node),
// B from the message stage on the static pocket-pocket list of the node's TWIN -- itself, or the same
// pocket atom of the sample that represents a group of identical pockets (agg_b rows indexed by the
// twin's node id, row_ptr_b / deg_b by the twin's index in the pocket-pocket problem).  Both parts are
// completed in tile order; ligand nodes have no B part.
__global__ __launch_bounds__(kThreads) void agg_complete2_kernel(
    float* agg, const float* head_a, const int* row_ptr_a, const int* deg_a, const float* agg_b,
    const float* head_b, const int* row_ptr_b, const int* deg_b, const int* twin_local, int twin_base,
    int n_lig, int n_rows, int H, int n_ghost, int max_tile) {
  const int row = (blockIdx.x * kThreads + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (row >= n_rows + n_ghost) return;
  const bool ghost = row >= n_rows;        // rows of the canonical pocket: the B part of pocket atom row - n_rows only
  const int da = ghost ? 0 : deg_a[row], sa = ghost ? 0 : row_ptr_a[row];
  const int a0 = sa >> 5, a1 = min((sa + da - 1) >> 5, max_tile);
  int db = 0, sb = 0, tw = 0;
  if (row >= n_lig) {
    const int tl = ghost ? row - n_rows : twin_local[row - n_lig];
    db = deg_b[tl]; sb = row_ptr_b[tl]; tw = twin_base + tl;
  }
  const int b0 = sb >> 5, b1 = min((sb + db - 1) >> 5, max_tile);
  for (int k = 4 * lane; k < H; k += 256) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (da > 0) {
      v = ld4(agg + (size_t)row * H + k);
      for (int T = a0 + 1; T <= a1; ++T) {
        const float4 h = ld4(head_a + (size_t)T * H + k);
        v.x += h.x; v.y += h.y; v.z += h.z; v.w += h.w;
      }
    }
    if (db > 0) {
      float4 u = ld4(agg_b + (size_t)tw * H + k);
      for (int T = b0 + 1; T <= b1; ++T) {
        const float4 h = ld4(head_b + (size_t)T * H + k);
        u.x += h.x; u.y += h.y; u.z += h.z; u.w += h.w;
      }
      v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
    }
    *reinterpret_cast<float4*>(agg + (size_t)row * H + k) = v;
  }
}

}  // namespace dsbdd
