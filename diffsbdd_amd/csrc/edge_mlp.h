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
  int e_cap;              // capacity of erow/ecol/ed0: the kernels never index past it, even when
                          // the radius graph overflowed (status bit 1 is then set by edges_kernel)
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

}  // namespace dsbdd
