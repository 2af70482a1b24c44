// Graph construction and per-node geometry kernels of EGNNDynamics.forward:
//   - sample offsets from the (sorted) batch masks
//   - radius graph -> edge list sorted by (row, col)      dynamics.py:169-187
//   - input assembly (split xh, time feature)             dynamics.py:89-111
//   - per-sample mean of x for coord2cross                egnn_new.py:307-310
//   - coordinate update, velocity / NaN guard / COM       dynamics.py:136,155-164
// All of this is O(N * n_per_sample) integer/float byte work (HBM/latency
// bound, microseconds); one wave per row, ballot-compaction keeps the
// reference's (row, col) order without a sort.
#pragma once
#include "common.h"

namespace dsbdd {

constexpr int kTileCtrInts = 32;   // 16 queue heads (8 XCDs x 2 MLP populations) + completion count
constexpr int kEdgeAlign = 32;     // edges of one (sample, node set) start at a multiple of a wave tile
constexpr int kLevels = 5;         // hop levels of the level-ordered edge list: 0 ligand, 1..3, 4 = farther

struct Cutoffs {
  int has_l, has_p, has_i;
  float cl, cp, ci;
};

__device__ __forceinline__ int lower_bound_i64(const int64_t* a, int n, int64_t v) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (a[mid] < v) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// node_batch[i] = sample id of node i ([ligand | pocket] numbering);
// lig_off[b] / poc_off[b] = first ligand / pocket node of sample b (b = 0..B).
__global__ void prep_kernel(const int64_t* mask_lig, int n_lig, const int64_t* mask_poc, int n_poc,
                            int B, int* node_batch, int* lig_off, int* poc_off, int* tile_ctr) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (tile_ctr && i < kTileCtrInts) tile_ctr[i] = 0;   // work-queue counters of the edge kernels
  if (i < n_lig) node_batch[i] = (int)mask_lig[i];
  else if (i < n_lig + n_poc) node_batch[i] = (int)mask_poc[i - n_lig];
  if (i <= B) {
    lig_off[i] = lower_bound_i64(mask_lig, n_lig, i);
    poc_off[i] = lower_bound_i64(mask_poc, n_poc, i);
  }
}

struct SegAlign {
  const int* node_batch; const int* lig_off; const int* poc_off;
  int n_lig; int B;
  int* scan_tmp;   // [n + 1] plain exclusive scan
  int* seg_base;   // [2B + 1]
};

__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// Optional second output of the radius graph: the edges that have a LIGAND endpoint (all edges of
// ligand rows + the ligand columns of pocket rows), in the same aligned layout with its own
// row_ptr / deg.  Block 0 of a pocket-conditioned chain evaluates the pocket-pocket messages
// separately (engine.hip, "pocket frame"), so its message stage runs on this list.
struct EdgeList2 {
  int* deg; int* row_ptr; int* erow; int* ecol; float* ed0; int e_cap;
  SegAlign seg;      // scan_tmp / seg_base of this list
};

// One wave per row node.  FILL=false counts neighbours (deg), FILL=true writes
// them at row_ptr[row].  Candidates are visited in index order (ligand nodes of
// the sample, then its pocket nodes) and compacted with ballot/popcount, so the
// output is sorted by (row, col) like torch.where(adj) (dynamics.py:185).
// Self loops are kept (distance 0 <= cutoff; dynamics.py never removes them).
template <bool FILL>
__global__ __launch_bounds__(kThreads) void edges_kernel(
    const float* __restrict__ x, const int* __restrict__ node_batch,
    const int* __restrict__ lig_off, const int* __restrict__ poc_off, int n_lig, int n_nodes,
    Cutoffs cut, int* __restrict__ deg, const int* __restrict__ row_ptr, int* __restrict__ erow,
    int* __restrict__ ecol, float* __restrict__ ed0, int e_cap, int* __restrict__ status,
    int* __restrict__ act_flag, SegAlign seg, int* __restrict__ row_ptr_out, EdgeList2 l2, int id_offset,
    int* __restrict__ lvl = nullptr) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  for (int i = wave; i < n_nodes; i += nwaves) {
    const int b = node_batch[i];
    const bool il = i < n_lig;
    const float xi = x[3 * i], yi = x[3 * i + 1], zi = x[3 * i + 2];
    // aligned layout: first edge of row i = base of its (sample, node set) segment + the edges of the
    // segment's earlier rows (plain scan differences); compact layout: the plain scan itself
    int base = 0;
    int base2 = 0;
    if (FILL) {
      if (seg.seg_base) {
        const int k = il ? b : seg.B + b;
        const int first = il ? lig_off[b] : n_lig + poc_off[b];
        base = seg.seg_base[k] + seg.scan_tmp[i] - seg.scan_tmp[first];
        if (lane == 0) row_ptr_out[i] = base;
      } else {
        base = row_ptr[i];
      }
    }
    if (FILL && l2.deg) {
      const int k = il ? b : l2.seg.B + b;
      const int first = il ? lig_off[b] : n_lig + poc_off[b];
      base2 = l2.seg.seg_base[k] + l2.seg.scan_tmp[i] - l2.seg.scan_tmp[first];
      if (lane == 0) l2.row_ptr[i] = base2;
    }
    int cnt = 0, cnt_lig = 0;
#pragma unroll 1
    for (int seg = 0; seg < 2; ++seg) {
      const int j_begin = seg == 0 ? lig_off[b] : n_lig + poc_off[b];
      const int j_end = seg == 0 ? lig_off[b + 1] : n_lig + poc_off[b + 1];
      const bool jl = seg == 0;
      const int has = (il && jl) ? cut.has_l : ((!il && !jl) ? cut.has_p : cut.has_i);
      const float c = (il && jl) ? cut.cl : ((!il && !jl) ? cut.cp : cut.ci);
      for (int j0 = j_begin; j0 < j_end; j0 += 64) {
        const int j = j0 + lane;
        const bool valid = j < j_end;
        float d2 = 0.f;
        if (valid) {
          const float dx = xi - x[3 * j], dy = yi - x[3 * j + 1], dz = zi - x[3 * j + 2];
          d2 = dx * dx + dy * dy + dz * dz;
        }
        // torch.cdist(...) <= cutoff  (dynamics.py:174-181), on the exact distance
        const bool pass = valid && (!has || __fsqrt_rn(d2) <= c);
        const unsigned long long m = __ballot(pass);
        if (FILL && pass) {
          const int pos = base + cnt + __popcll(m & ((1ull << lane) - 1ull));
          if (pos >= 0 && pos < e_cap) {   // (pos < 0 only with unsorted masks, which the host rejects)
            erow[pos] = i + id_offset; ecol[pos] = j + id_offset; ed0[pos] = d2;
          }
          if (l2.deg && (il || jl)) {     // ligand columns come first in a row: same running position
            const int pos2 = base2 + cnt + __popcll(m & ((1ull << lane) - 1ull));
            if (pos2 >= 0 && pos2 < l2.e_cap) { l2.erow[pos2] = i; l2.ecol[pos2] = j; l2.ed0[pos2] = d2; }
          }
        }
        cnt += __popcll(m);
      }
      if (seg == 0) cnt_lig = cnt;
    }
    if (!FILL && lane == 0) {
      deg[i] = cnt;
      if (l2.deg) l2.deg[i] = il ? cnt : cnt_lig;
      // "active" for the coordinate MLPs in pocket-conditioning mode: ligand nodes and
      // pocket nodes that are a column of some ligand-row edge (graph is symmetric)
      if (act_flag) act_flag[i] = (il || cnt_lig > 0) ? 1 : 0;
      // hop level of the node (levels_kernel below): 0 ligand, 1 = pocket node with a ligand neighbour
      if (lvl) lvl[i] = il ? 0 : (cnt_lig > 0 ? 1 : kLevels - 1);
    }
    if (FILL && lane == 0 && base + cnt > e_cap) atomicOr(status, 2);
    if (FILL && seg.seg_base) {
      // the last row of a (sample, node set) segment fills the segment up to the next wave-tile
      // boundary with inactive entries (row = -1), see scan_kernel
      const int seg_last = (il ? lig_off[b + 1] : n_lig + poc_off[b + 1]) - 1;
      if (i == seg_last) {
        const int from = base + cnt, to = (from + kEdgeAlign - 1) & ~(kEdgeAlign - 1);
        const int pos = from + lane;
        if (pos >= 0 && pos < to && pos < e_cap) { erow[pos] = -1; ecol[pos] = 0; ed0[pos] = 0.f; }
        if (l2.deg) {
          const int from2 = base2 + (il ? cnt : cnt_lig), to2 = (from2 + kEdgeAlign - 1) & ~(kEdgeAlign - 1);
          const int p2 = from2 + lane;
          if (p2 >= 0 && p2 < to2 && p2 < l2.e_cap) { l2.erow[p2] = -1; l2.ecol[p2] = 0; l2.ed0[p2] = 0.f; }
        }
      }
    }
  }
}

// Exclusive scan of deg[0..n) -> row_ptr[0..n], row_ptr[n] = total.  One
// workgroup of 1024 threads (n is tens of thousands) walks tiles of 4096
// elements: coalesced 16-byte loads, 4-element serial prefix per lane, wave
// scan by lane shuffles, 16 wave totals through LDS, running carry in a register.
//
// With `seg` set, the edges of every (sample, node set) segment -- the rows of the ligand
// nodes of sample b, then the rows of its pocket nodes; 2B segments in node order -- start at
// a multiple of kEdgeAlign: row_ptr[i] = seg_base[segment of i] + (edges of the earlier rows
// of that segment) -- evaluated and stored by edges_kernel<true> from the plain scan and seg_base
// computed here --, row_ptr[n] = padded total, and the gaps are filled with inactive entries by
// edges_kernel<true>.  A wave tile of the edge kernels (32 consecutive edges) then never mixes
// samples and holds the same edges whatever else is in the batch, which makes every per-row sum
// independent of the batch composition (bitwise identical results for any sharding).
// A row's edge count stays in deg[]; row i owns [row_ptr[i], row_ptr[i] + deg[i]).
__device__ __forceinline__ int block_scan_1024(int mine, int* s_wave, int* s_total) {
  // exclusive prefix of `mine` over the 1024 threads of the workgroup; *s_total = sum
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  int incl = mine;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int u = __shfl_up(incl, o);
    if (lane >= o) incl += u;
  }
  if (lane == 63) s_wave[w] = incl;
  __syncthreads();
  if (w == 0) {
    int tot = lane < 16 ? s_wave[lane] : 0;
    int inc = tot;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
      const int u = __shfl_up(inc, o);
      if (lane >= o) inc += u;
    }
    if (lane < 16) s_wave[lane] = inc - tot;
    if (lane == 15) *s_total = inc;
  }
  __syncthreads();
  const int r = s_wave[w] + incl - mine;
  return r;
}

__device__ void scan_one(const int* deg, int* row_ptr, int n, SegAlign seg, int* s_wave, int* s_carry_p);

__global__ __launch_bounds__(1024) void scan_kernel(const int* deg, int* row_ptr, int n, SegAlign seg,
                                                    const int* deg_b, int* row_ptr_b, SegAlign seg_b) {
  __shared__ int s_wave[16];
  __shared__ int s_carry;
  scan_one(deg, row_ptr, n, seg, s_wave, &s_carry);
  if (deg_b) {                     // second list of the same launch (EdgeList2)
    __syncthreads();
    scan_one(deg_b, row_ptr_b, n, seg_b, s_wave, &s_carry);
  }
}

__device__ void scan_one(const int* deg, int* row_ptr, int n, SegAlign seg, int* s_wave, int* s_carry_p) {
  int& s_carry = *s_carry_p;
  const int t = threadIdx.x;
  int* out = seg.seg_base ? seg.scan_tmp : row_ptr;
  const bool vec = ((reinterpret_cast<uintptr_t>(deg) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
  int carry = 0;
  for (int base = 0; base < n; base += 4096) {
    const int i0 = base + 4 * t;
    int v[4] = {0, 0, 0, 0};
    if (vec && i0 + 3 < n) {
      const int4 q = *reinterpret_cast<const int4*>(deg + i0);
      v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) if (i0 + k < n) v[k] = deg[i0 + k];
    }
    const int mine = v[0] + v[1] + v[2] + v[3];
    int run = carry + block_scan_1024(mine, s_wave, s_carry_p);
    const int tile_total = s_carry;
    if (vec && i0 + 3 < n) {
      int4 o4;
      o4.x = run; o4.y = run + v[0]; o4.z = o4.y + v[1]; o4.w = o4.z + v[2];
      *reinterpret_cast<int4*>(out + i0) = o4;
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) if (i0 + k < n) { out[i0 + k] = run; run += v[k]; }
    }
    carry += tile_total;
    __syncthreads();                                   // s_wave / s_carry are reused by the next tile
  }
  if (t == 0) out[n] = carry;
  if (!seg.seg_base) return;
  __syncthreads();                                     // the plain scan is visible to the whole workgroup
  // padded length of every segment -> exclusive scan -> seg_base
  const int S = 2 * seg.B;
  auto seg_begin = [&](int k) { return k < seg.B ? seg.lig_off[k] : seg.n_lig + seg.poc_off[k - seg.B]; };
  auto seg_end = [&](int k) { return k < seg.B ? seg.lig_off[k + 1] : seg.n_lig + seg.poc_off[k - seg.B + 1]; };
  carry = 0;
  for (int base = 0; base < S; base += 1024) {
    const int k = base + t;
    int len = 0;
    if (k < S) len = (max(out[seg_end(k)] - out[seg_begin(k)], 0) + kEdgeAlign - 1) & ~(kEdgeAlign - 1);
    const int ex = carry + block_scan_1024(len, s_wave, s_carry_p);
    if (k < S) seg.seg_base[k] = ex;
    carry += s_carry;
    __syncthreads();
  }
  if (t == 0) {
    seg.seg_base[S] = carry;
    row_ptr[n] = carry;           // padded total; row_ptr[0 .. n) is written by edges_kernel<true>
  }
}

// ---------------------------------------------------------------------------
// Level-ordered edge list (pocket-conditioning mode, ligand output only).
//
// The caller of a pocket-conditioned denoiser step reads the LIGAND rows of the output only
// (conditional_model.py:268-272 discards the pocket part), and the pocket coordinates are never updated.
// Walking the network backwards, the last message stage therefore only has to produce h for the nodes the
// last coordinate stage reads -- the ligand nodes and their neighbours (hop <= 1) --, the stage before it
// for hop <= 2, and so on: stage g of G message stages computes the rows with hop <= G - g, and reads its
// neighbours' h at hop <= G - g + 1.  Everything else is dead code for this call and is not evaluated.
// (Exactly the same numbers come out for the ligand: no live value depends on a skipped one.)
//
// hop(i) = graph distance from node i to the nearest ligand node of its sample (0 for ligand nodes), through
// the edges of this call; level(i) = min(hop, kLevels - 1).  The edge list is re-ordered by
// (level, node set, sample, row): the rows a stage needs are then a PREFIX of the list (lvl_end[r], a device
// scalar, like the update_coords_mask prefix), and the same nodes a prefix of lvl_list (lvl_cnt[r]) for the
// node-level GEMMs.  Every (level, sample) segment starts at a wave-tile boundary and a row's position in
// its segment depends on its own sample only, so results stay independent of the batch composition.
//
//   levels_kernel   one workgroup per sample: hop levels 2, 3 by relaxation over the natural-order list
//                   (level 1 comes from the count pass), then rows / edges per (level, sample)
//   level_scan_kernel   one workgroup: exclusive scans over the kLevels * B segments
//   level_place_kernel  one workgroup per sample, wave L places level L: new row_ptr, lvl_list, pads
//   level_copy_kernel   one thread per natural-list slot: move the edge to its new position
// mean[b] = mean of x over ALL nodes of sample b = blockIdx.x (egnn_new.py:307-310); one fixed reduction tree, shared
// by sample_mean_kernel and levels_kernel (round 5: block 0's mean of a pruned call rides in the levels launch)
__device__ __forceinline__ void sample_mean_body(const float* x, const int* lig_off, const int* poc_off, int n_lig,
                                                 float* mean, float (*red)[kThreads]) {
  const int b = blockIdx.x, t = threadIdx.x;
  const int l0 = lig_off[b], l1 = lig_off[b + 1], p0 = n_lig + poc_off[b], p1 = n_lig + poc_off[b + 1];
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  for (int i = l0 + t; i < l1; i += kThreads) { s0 += x[3 * i]; s1 += x[3 * i + 1]; s2 += x[3 * i + 2]; }
  for (int i = p0 + t; i < p1; i += kThreads) { s0 += x[3 * i]; s1 += x[3 * i + 1]; s2 += x[3 * i + 2]; }
  red[0][t] = s0; red[1][t] = s1; red[2][t] = s2;
  __syncthreads();
  for (int o = kThreads / 2; o > 0; o >>= 1) {
    if (t < o) { red[0][t] += red[0][t + o]; red[1][t] += red[1][t + o]; red[2][t] += red[2][t + o]; }
    __syncthreads();
  }
  if (t < 3) {
    int cnt = (l1 - l0) + (p1 - p0);
    if (cnt == 0) cnt = 1;
    mean[3 * b + t] = red[t][0] / (float)cnt;
  }
  __syncthreads();
}

struct LevelArgs {
  const int* node_batch; const int* lig_off; const int* poc_off; int n_lig; int B;
  int* lvl;                 // [N]
  const int* deg;           // [N]
  const int* row_ptr_nat;   // [N + 1] natural-order list
  const int* erow_nat; const int* ecol_nat; const float* ed0_nat;
  int* seg_rows;            // [kLevels * B] rows of segment (L, b);   level 0 = the ligand rows of sample b
  int* seg_edges;           // [kLevels * B] edges (not padded)
  int* node_base;           // [kLevels * B + 1] exclusive scan of seg_rows
  int* edge_base;           // [kLevels * B + 1] exclusive scan of the padded seg_edges
  int* lvl_cnt;             // [2][kLevels] nodes with level <= r: entries of lvl_list / without the leading ghost rows
  int* lvl_end;             // [2][kLevels] end of the edges of the rows with level <= r (padded): list position / minus edge_off
  int* lvl_list;            // [node_off + N] ghost rows, then the nodes ordered by (level, node id)
  int* row_ptr;             // [N] level-ordered list (the total is lvl_end[kLevels - 1])
  int* erow; int* ecol; float* ed0; int e_cap;
  // running sums over calls (bench / tests): [0..5) nodes with level <= r, [5..10) list slots, [10..15) edges, [15] calls
  unsigned long long* stats;
  // "ghost" rows of the canonical pocket (engine.hip, forward cone): node_off entries at the front of lvl_list and
  // edge_off slots (a multiple of kEdgeAlign) at the front of the edge list are theirs, written once per chain
  int node_off; int edge_off;
  int e_cap_nat;            // capacity of the natural-order list: never indexed past it, even when the radius graph
                            // overflowed (status bit 1 is then set by edges_kernel and the call's result is discarded)
  const float* mean_x; float* mean_out;   // optional: levels_kernel also writes the per-sample mean of mean_x (sample_mean_body)
};

__global__ __launch_bounds__(kThreads) void levels_kernel(LevelArgs a) {
  __shared__ int s_rows[kLevels], s_edges[kLevels];
  __shared__ float s_red[3][kThreads];
  if (a.mean_out) sample_mean_body(a.mean_x, a.lig_off, a.poc_off, a.n_lig, a.mean_out, s_red);
  const int b = blockIdx.x, t = threadIdx.x;
  const int p0 = a.n_lig + a.poc_off[b], p1 = a.n_lig + a.poc_off[b + 1];
  const int l0 = a.lig_off[b], l1 = a.lig_off[b + 1];
  if (t < kLevels) { s_rows[t] = 0; s_edges[t] = 0; }
  for (int k = 2; k < kLevels - 1; ++k) {
    __syncthreads();                       // level k-1 is final (global writes of this workgroup are visible to it)
    for (int i = p0 + t; i < p1; i += kThreads) {
      if (a.lvl[i] != kLevels - 1) continue;
      const int s = a.row_ptr_nat[i], d = a.deg[i];
      bool hit = false;
      for (int e = max(s, 0); e < min(s + d, a.e_cap_nat) && !hit; ++e) {
        const int j = a.ecol_nat[e];
        hit = j >= a.n_lig && a.lvl[j] == k - 1;     // (a ligand neighbour would have made it level 1)
      }
      if (hit) a.lvl[i] = k;               // readers of this pass test == k-1: no race
    }
  }
  __syncthreads();
  for (int i = p0 + t; i < p1; i += kThreads) {
    const int L = a.lvl[i];
    atomicAdd(&s_rows[L], 1);              // integer sums: order does not matter
    atomicAdd(&s_edges[L], a.deg[i]);
  }
  int le = 0;
  for (int i = l0 + t; i < l1; i += kThreads) le += a.deg[i];
  if (le) atomicAdd(&s_edges[0], le);
  __syncthreads();
  if (t < kLevels) {
    a.seg_rows[t * a.B + b] = t == 0 ? (l1 - l0) : s_rows[t];
    a.seg_edges[t * a.B + b] = s_edges[t];
  }
}

__global__ __launch_bounds__(1024) void level_scan_kernel(LevelArgs a, int n_nodes) {
  __shared__ int s_wave[16];
  __shared__ int s_tot;
  const int t = threadIdx.x, S = kLevels * a.B;
  int carry_n = 0, carry_e = 0;
  for (int base = 0; base < S; base += 1024) {
    const int k = base + t;
    const int r = k < S ? a.seg_rows[k] : 0;
    const int e = k < S ? ((a.seg_edges[k] + kEdgeAlign - 1) & ~(kEdgeAlign - 1)) : 0;
    const int xn = carry_n + block_scan_1024(r, s_wave, &s_tot);
    carry_n += s_tot;
    __syncthreads();
    const int xe = carry_e + block_scan_1024(e, s_wave, &s_tot);
    carry_e += s_tot;
    __syncthreads();
    if (k < S) { a.node_base[k] = a.node_off + xn; a.edge_base[k] = a.edge_off + xe; }
  }
  if (t == 0) { a.node_base[S] = a.node_off + carry_n; a.edge_base[S] = a.edge_off + carry_e; }
  __syncthreads();
  if (t < kLevels) {                        // cumulative ends of the levels
    const int cn = t == kLevels - 1 ? a.node_off + carry_n : a.node_base[(t + 1) * a.B];
    const int ce = t == kLevels - 1 ? a.edge_off + carry_e : a.edge_base[(t + 1) * a.B];
    a.lvl_cnt[t] = cn; a.lvl_cnt[kLevels + t] = cn - a.node_off;
    a.lvl_end[t] = ce; a.lvl_end[kLevels + t] = ce - a.edge_off;
    if (a.stats) {
      int own = 0;                          // edges (without padding) of level t, then of the rows with level <= t
      for (int k = t * a.B; k < (t + 1) * a.B; ++k) own += a.seg_edges[k];
      s_wave[t] = own;
      __threadfence_block();
      __builtin_amdgcn_wave_barrier();
      unsigned long long ed = 0;
      for (int k = 0; k <= t; ++k) ed += (unsigned long long)s_wave[k];   // (same wave: LDS writes are in order)
      a.stats[t] += (unsigned long long)(cn - a.node_off);
      a.stats[kLevels + t] += (unsigned long long)(ce - a.edge_off);
      a.stats[2 * kLevels + t] += ed;
      if (t == 0) a.stats[3 * kLevels] += 1ull;
    }
  }
}

// exclusive position of segment k = (L, b) among the kLevels * B segments: nodes, padded edges (wave sums)
__device__ __forceinline__ void level_bases(const LevelArgs& a, int k, int lane, int& nb, int& eb) {
  int sn = 0, se = 0;
  for (int k0 = 0; k0 < k; k0 += 64) {
    const int kk = k0 + lane;
    if (kk < k) { sn += a.seg_rows[kk]; se += (a.seg_edges[kk] + kEdgeAlign - 1) & ~(kEdgeAlign - 1); }
  }
  nb = a.node_off + wave_sum_i(sn);
  eb = a.edge_off + wave_sum_i(se);
}

// fold_scan != 0 (round 5): no level_scan_kernel launch in front -- every wave finds the bases of its own segment with
// two wave sums over the (level, sample) table, and workgroup 0 writes the per-level ends / counts / statistics
__global__ __launch_bounds__(kThreads) void level_place_kernel(LevelArgs a, int fold_scan = 0) {
  static_assert(kThreads / 64 >= kLevels - 1, "one wave per pocket level");
  const int b = blockIdx.x, lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (fold_scan && b == 0 && w == kThreads / 64 - 1) {
    // cumulative ends of the levels (what level_scan_kernel wrote): level t ends where segment (t + 1, 0) begins
    unsigned long long ed_run = 0;
    for (int t = 0; t < kLevels; ++t) {
      int cn, ce;
      level_bases(a, (t + 1) * a.B, lane, cn, ce);
      int own = 0;
      for (int k0 = t * a.B; k0 < (t + 1) * a.B; k0 += 64) { const int kk = k0 + lane; if (kk < (t + 1) * a.B) own += a.seg_edges[kk]; }
      ed_run += (unsigned long long)wave_sum_i(own);
      if (lane == 0) {
        a.lvl_cnt[t] = cn; a.lvl_cnt[kLevels + t] = cn - a.node_off;
        a.lvl_end[t] = ce; a.lvl_end[kLevels + t] = ce - a.edge_off;
        if (a.stats) {
          a.stats[t] += (unsigned long long)(cn - a.node_off);
          a.stats[kLevels + t] += (unsigned long long)(ce - a.edge_off);
          a.stats[2 * kLevels + t] += ed_run;
          if (t == 0) a.stats[3 * kLevels] += 1ull;
        }
      }
    }
  }
  auto pads = [&](int from) {               // fill the segment up to the next wave-tile boundary
    const int to = (from + kEdgeAlign - 1) & ~(kEdgeAlign - 1), pos = from + lane;
    if (pos < to && pos < a.e_cap) { a.erow[pos] = -1; a.ecol[pos] = 0; a.ed0[pos] = 0.f; }
  };
  if (w == 0) {                             // ligand rows keep their natural order
    const int l0 = a.lig_off[b], l1 = a.lig_off[b + 1];
    int nb, eb;
    if (fold_scan) level_bases(a, b, lane, nb, eb); else { nb = a.node_base[b]; eb = a.edge_base[b]; }
    const int s0 = l0 < l1 ? a.row_ptr_nat[l0] : 0;
    for (int i = l0 + lane; i < l1; i += 64) {
      a.lvl_list[nb + (i - l0)] = i;
      a.row_ptr[i] = eb + (a.row_ptr_nat[i] - s0);
    }
    pads(eb + a.seg_edges[b]);
  }
  const int L = w + 1;                      // pocket level of this wave
  if (L >= kLevels) return;
  const int p0 = a.n_lig + a.poc_off[b], p1 = a.n_lig + a.poc_off[b + 1];
  int n_run, e_run;
  if (fold_scan) level_bases(a, L * a.B + b, lane, n_run, e_run);
  else { n_run = a.node_base[L * a.B + b]; e_run = a.edge_base[L * a.B + b]; }
  for (int i0 = p0; i0 < p1; i0 += 64) {
    const int i = i0 + lane;
    const bool mine = i < p1 && a.lvl[i] == L;
    const int d = mine ? a.deg[i] : 0;
    const unsigned long long m = __ballot(mine);
    int incl = d;                            // inclusive wave scan of the degrees of this level's rows
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int u = __shfl_up(incl, o);
      if (lane >= o) incl += u;
    }
    if (mine) {
      a.lvl_list[n_run + __popcll(m & ((1ull << lane) - 1ull))] = i;
      a.row_ptr[i] = e_run + incl - d;
    }
    n_run += __popcll(m);
    e_run += __shfl(incl, 63);
  }
  pads(e_run);
}

// Ghost rows of the canonical pocket: the static pocket-pocket list of the frame's representative (node ids
// id_src + i) becomes the front segment of the level-ordered list with node ids id_dst + i; their degrees,
// list positions and places at the front of lvl_list are static for the chain.
__global__ void ghost_setup_kernel(const int* erow3, const int* ecol3, const float* ed03, const int* row_ptr3,
                                   const int* deg3, int n3, int id_src, int id_dst, int* erow, int* ecol,
                                   float* ed0, int e_cap, int* deg, int* row_ptr, int* lvl_list,
                                   const float* xframe, float* x) {
  const int total = min(row_ptr3[n3], e_cap);
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < total; p += gridDim.x * blockDim.x) {
    const int r = erow3[p];
    erow[p] = r < 0 ? -1 : r - id_src + id_dst;
    ecol[p] = r < 0 ? 0 : ecol3[p] - id_src + id_dst;
    ed0[p] = ed03[p];
  }
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n3; i += gridDim.x * blockDim.x) {
    deg[id_dst + i] = deg3[i];
    row_ptr[id_dst + i] = row_ptr3[i];
    lvl_list[i] = id_dst + i;
    x[3 * (id_dst + i)] = xframe[3 * i]; x[3 * (id_dst + i) + 1] = xframe[3 * i + 1];
    x[3 * (id_dst + i) + 2] = xframe[3 * i + 2];
  }
}

// dst[i][:] = src[rows[i]][:]  (one wave per row, 16-byte lanes): embedded features of the frame's pockets -> ghost rows
__global__ __launch_bounds__(kThreads) void gather_rows_kernel(float* dst, const float* src, const int* rows, int n,
                                                               int H) {
  const int i = (blockIdx.x * kThreads + threadIdx.x) >> 6, lane = threadIdx.x & 63;
  if (i >= n) return;
  const float* a = src + (size_t)rows[i] * H;
  float* b = dst + (size_t)i * H;
  for (int k = 4 * lane; k < H; k += 256) *reinterpret_cast<float4*>(b + k) = ld4(a + k);
}

// h[i] <- h[ghost twin of i] for the pocket rows with lo < level <= hi: rows the next message stage reads but the
// ligand could not have influenced yet -- their value is the canonical pocket's (engine.hip, forward cone).
__global__ __launch_bounds__(kThreads) void canon_fill_kernel(float* h, const int* lvl, const int* twin_local,
                                                              int n_lig, int n_nodes, int ghost_base, int lo,
                                                              int hi, int H, float* pq = nullptr, int ldpq = 0) {
  const int i = n_lig + ((blockIdx.x * kThreads + threadIdx.x) >> 6), lane = threadIdx.x & 63;
  if (i >= n_nodes) return;
  const int L = lvl[i];
  if (L <= lo || L > hi) return;
  const int g = ghost_base + twin_local[i - n_lig];
  const float* src = h + (size_t)g * H;
  float* dst = h + (size_t)i * H;
  for (int k = 4 * lane; k < H; k += 256) *reinterpret_cast<float4*>(dst + k) = ld4(src + k);
  // pq: a projection of h that was computed for the ghost rows in the same launch as their h (the next message stage's
  // P|Q): a row that takes the canonical h takes the canonical projection too -- the same bits a GEMM over it would give
  if (pq)
    for (int k = 4 * lane; k < ldpq; k += 256)
      *reinterpret_cast<float4*>(pq + (size_t)i * ldpq + k) = ld4(pq + (size_t)g * ldpq + k);
}

__global__ void level_copy_kernel(LevelArgs a, int n_nodes) {
  const int total = min(a.row_ptr_nat[n_nodes], a.e_cap_nat);
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const int i = a.erow_nat[e];
    if (i < 0 || i >= n_nodes) continue;    // padding of the natural-order list
    const int pos = a.row_ptr[i] + (e - a.row_ptr_nat[i]);
    if (pos >= 0 && pos < a.e_cap) { a.erow[pos] = i; a.ecol[pos] = a.ecol_nat[e]; a.ed0[pos] = a.ed0_nat[e]; }
  }
}

// Zero fill as an ordinary kernel on the caller's stream.  hipMemsetAsync is avoided inside
// dsbdd_dynamics_forward: its blit submissions were observed to lose their ordering against launches
// of a captured graph on the same stream (profiles/README.md, "eager calls between graph replays").
__global__ void zero_kernel(uint4* p, size_t n16, unsigned char* tail, int n_tail) {
  const uint4 z = make_uint4(0u, 0u, 0u, 0u);
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x)
    p[i] = z;
  if (blockIdx.x == 0 && (int)threadIdx.x < n_tail) tail[threadIdx.x] = 0;
}

__global__ void copy16_kernel(uint4* dst, const uint4* src, size_t n16) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i < n16) dst[i] = src[i];
}

// ptr must be 16-byte aligned (every workspace buffer is 256-byte aligned)
inline hipError_t zero_async(void* ptr, size_t bytes, hipStream_t s) {
  if (bytes == 0) return hipSuccess;
  const size_t n16 = bytes / 16;
  const int n_tail = (int)(bytes % 16);
  size_t blocks = (n16 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(zero_kernel, dim3((unsigned)blocks), dim3(256), 0, s, static_cast<uint4*>(ptr), n16,
                     static_cast<unsigned char*>(ptr) + n16 * 16, n_tail);
  return hipGetLastError();
}

// dst[k][j * CT + c] = src[k][c * 32 + j] for an [H][H] matrix, CT = H / 32 (see EdgeMlpW::W2TP).
__global__ void permute_w2t_kernel(const float* src, float* dst, int H) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= H * H) return;
  const int k = i / H, o = i % H, ct = H / 32;
  const int j = o / ct, c = o % ct;
  dst[i] = src[k * H + c * 32 + j];
}

// Teacher-forced edge list: copy rows/cols, compute d0, build row_ptr counts.
__global__ void ext_edges_kernel(const int* row, const int* col, int E, const float* x, int* erow,
                                 int* ecol, float* ed0, int* deg) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= E) return;
  const int r = row[e], c = col[e];
  erow[e] = r; ecol[e] = c;
  const float dx = x[3 * r] - x[3 * c], dy = x[3 * r + 1] - x[3 * c + 1], dz = x[3 * r + 2] - x[3 * c + 2];
  ed0[e] = dx * dx + dy * dy + dz * dz;
  atomicAdd(&deg[r], 1);
}

// Active-node flags for a teacher-forced edge list: ligand nodes, plus every column
// of an edge whose row is a ligand node.
__global__ void ext_flags_init_kernel(int* flag, int n_lig, int n_nodes) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_nodes) flag[i] = i < n_lig ? 1 : 0;
}
__global__ void ext_flags_kernel(const int* row, const int* col, int E, int n_lig, int* flag) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e < E && row[e] < n_lig) flag[col[e]] = 1;   // benign race: all writers store 1
}
// list[ptr[i]] = i for flagged nodes (ptr = exclusive scan of flag): sorted node list.
__global__ void compact_kernel(const int* flag, const int* ptr, int* list, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && flag[i]) list[ptr[i]] = i;
}

// x[N][3] <- coordinates of both node sets; h0[:, J] <- t[sample]; pad columns
// of h0 (J+1..JP-1) <- 0.  dynamics.py:89-111.
__global__ void assemble_kernel(const float* xh_lig, int dl, const float* xh_poc, int dp, int n_lig,
                                int n_nodes, const int* node_batch, const float* t, int t_count,
                                float* x, float* x_in, float* h0, int J, int JP) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_nodes) return;
  const float* src = i < n_lig ? xh_lig + (size_t)i * dl : xh_poc + (size_t)(i - n_lig) * dp;
  const float a = src[0], b = src[1], c = src[2];
  x[3 * i] = a; x[3 * i + 1] = b; x[3 * i + 2] = c;
  x_in[3 * i] = a; x_in[3 * i + 1] = b; x_in[3 * i + 2] = c;
  h0[(size_t)i * JP + J] = t[t_count == 1 ? 0 : node_batch[i]];
  for (int k = J + 1; k < JP; ++k) h0[(size_t)i * JP + k] = 0.f;
}

// prep_kernel + assemble_kernel in one launch (thread i: node i and sample i); the time feature is looked up through
// the masks themselves, so no thread depends on another one's node_batch entry.
__global__ void prep_assemble_kernel(const int64_t* mask_lig, int n_lig, const int64_t* mask_poc, int n_poc, int B,
                                     int* node_batch, int* lig_off, int* poc_off, int* tile_ctr, const float* xh_lig,
                                     int dl, const float* xh_poc, int dp, const float* t, int t_count, float* x,
                                     float* x_in, float* h0, int J, int JP) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (tile_ctr && i < kTileCtrInts) tile_ctr[i] = 0;
  if (i <= B) {
    lig_off[i] = lower_bound_i64(mask_lig, n_lig, i);
    poc_off[i] = lower_bound_i64(mask_poc, n_poc, i);
  }
  if (i >= n_lig + n_poc) return;
  const int b = (int)(i < n_lig ? mask_lig[i] : mask_poc[i - n_lig]);
  node_batch[i] = b;
  const float* src = i < n_lig ? xh_lig + (size_t)i * dl : xh_poc + (size_t)(i - n_lig) * dp;
  const float a0 = src[0], a1 = src[1], a2 = src[2];
  x[3 * i] = a0; x[3 * i + 1] = a1; x[3 * i + 2] = a2;
  x_in[3 * i] = a0; x_in[3 * i + 1] = a1; x_in[3 * i + 2] = a2;
  h0[(size_t)i * JP + J] = t[t_count == 1 ? 0 : b];
  for (int k = J + 1; k < JP; ++k) h0[(size_t)i * JP + k] = 0.f;
}

// mean[b] = mean of x over ALL nodes of sample b (egnn_new.py:307-310); one
// workgroup per sample.
__global__ __launch_bounds__(kThreads) void sample_mean_kernel(const float* x, const int* lig_off,
                                                               const int* poc_off, int n_lig,
                                                               float* mean) {
  __shared__ float red[3][kThreads];
  sample_mean_body(x, lig_off, poc_off, n_lig, mean, red);
}

// x[i] += (sum over the row's edges of trans) for the nodes whose coordinates are updated
// (update_coords_mask, egnn_new.py:118-121).  The coordinate edge kernel leaves, per MLP
// population q, the partial sum of the wave tile that holds the row's first edge in xagg[q][i]
// and the partial sums of the following tiles of a row that spans several in xhead[q][tile];
// they are added here in tile order (fixed summation order, no atomics).
__global__ void coord_update_kernel(float* x, const float* xagg, const float* xhead, int n_q,
                                    size_t xagg_stride, size_t xhead_stride, const int* row_ptr,
                                    const int* deg, int n_upd3, int max_tile, int shift) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n_upd3) return;
  const int i = idx / 3, c = idx - 3 * i;
  const int d = deg[i];
  if (d == 0) return;
  const int s = row_ptr[i], t0 = s >> shift, t1 = min((s + d - 1) >> shift, max_tile);   // (max_tile, shift: see agg_complete_kernel)
  float tot = 0.f;
  for (int q = 0; q < n_q; ++q) {
    float a = xagg[q * xagg_stride + idx];
    for (int T = t0 + 1; T <= t1; ++T) a += xhead[q * xhead_stride + 4 * (size_t)T + c];
    tot += a;
  }
  x[idx] += tot;
}

// The same update with one workgroup per sample, followed by the per-sample mean of the NEW coordinates (the next
// block's coord2cross reference point, sample_mean_kernel): one launch instead of two between two blocks.
// The mean is reduced exactly like sample_mean_kernel does (same bits).
__global__ __launch_bounds__(kThreads) void coord_update_mean_kernel(
    float* x, const float* xagg, const float* xhead, int n_q, size_t xagg_stride, size_t xhead_stride,
    const int* row_ptr, const int* deg, int n_upd, const int* lig_off, const int* poc_off, int n_lig, float* mean,
    int max_tile, int shift) {
  __shared__ float red[3][kThreads];
  const int b = blockIdx.x, t = threadIdx.x;
  const int l0 = lig_off[b], l1 = lig_off[b + 1], p0 = n_lig + poc_off[b], p1 = n_lig + poc_off[b + 1];
  for (int seg = 0; seg < 2; ++seg) {
    const int a = seg ? p0 : l0, e = min(seg ? p1 : l1, n_upd);
    for (int k = 3 * a + t; k < 3 * e; k += kThreads) {
      const int i = k / 3, c = k - 3 * i;
      const int d = deg[i];
      if (d == 0) continue;
      const int s = row_ptr[i], t0 = s >> shift, t1 = min((s + d - 1) >> shift, max_tile);
      float tot = 0.f;
      for (int q = 0; q < n_q; ++q) {
        float v = xagg[q * xagg_stride + k];
        for (int T = t0 + 1; T <= t1; ++T) v += xhead[q * xhead_stride + 4 * (size_t)T + c];
        tot += v;
      }
      x[k] += tot;
    }
  }
  if (!mean) return;
  __syncthreads();
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  for (int i = l0 + t; i < l1; i += kThreads) { s0 += x[3 * i]; s1 += x[3 * i + 1]; s2 += x[3 * i + 2]; }
  for (int i = p0 + t; i < p1; i += kThreads) { s0 += x[3 * i]; s1 += x[3 * i + 1]; s2 += x[3 * i + 2]; }
  red[0][t] = s0; red[1][t] = s1; red[2][t] = s2;
  __syncthreads();
  for (int o = kThreads / 2; o > 0; o >>= 1) {
    if (t < o) { red[0][t] += red[0][t + o]; red[1][t] += red[1][t + o]; red[2][t] += red[2][t + o]; }
    __syncthreads();
  }
  if (t < 3) {
    int cnt = (l1 - l0) + (p1 - p0);
    if (cnt == 0) cnt = 1;
    mean[3 * b + t] = red[t][0] / (float)cnt;
  }
}

// vel = x_final - x_in (dynamics.py:136), NaN guard (dynamics.py:155-159: flag
// instead of a host sync), optional per-sample mean removal in joint mode
// (dynamics.py:161-164), scatter into eps[:, 0:3].  One workgroup per sample.
__global__ __launch_bounds__(kThreads) void finalize_kernel(
    const float* x, const float* x_in, const int* lig_off, const int* poc_off, int n_lig,
    int remove_mean, float* eps_lig, int dl, float* eps_poc, int dp, int* status) {
  __shared__ float red[3][kThreads];
  const int b = blockIdx.x, t = threadIdx.x;
  const int l0 = lig_off[b], l1 = lig_off[b + 1], p0 = n_lig + poc_off[b], p1 = n_lig + poc_off[b + 1];
  float m0 = 0.f, m1 = 0.f, m2 = 0.f;
  bool nan = false;
  if (remove_mean) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f;
    for (int seg = 0; seg < 2; ++seg) {
      const int a = seg ? p0 : l0, e = seg ? p1 : l1;
      for (int i = a + t; i < e; i += kThreads) {
        s0 += x[3 * i] - x_in[3 * i]; s1 += x[3 * i + 1] - x_in[3 * i + 1]; s2 += x[3 * i + 2] - x_in[3 * i + 2];
      }
    }
    red[0][t] = s0; red[1][t] = s1; red[2][t] = s2;
    __syncthreads();
    for (int o = kThreads / 2; o > 0; o >>= 1) {
      if (t < o) { red[0][t] += red[0][t + o]; red[1][t] += red[1][t + o]; red[2][t] += red[2][t + o]; }
      __syncthreads();
    }
    int cnt = (l1 - l0) + (p1 - p0);
    if (cnt == 0) cnt = 1;
    m0 = red[0][0] / (float)cnt; m1 = red[1][0] / (float)cnt; m2 = red[2][0] / (float)cnt;
  }
  for (int seg = 0; seg < 2; ++seg) {
    const int a = seg ? p0 : l0, e = seg ? p1 : l1;
    for (int i = a + t; i < e; i += kThreads) {
      const float v0 = x[3 * i] - x_in[3 * i], v1 = x[3 * i + 1] - x_in[3 * i + 1],
                  v2 = x[3 * i + 2] - x_in[3 * i + 2];
      nan = nan || (v0 != v0) || (v1 != v1) || (v2 != v2);
      float* dst = nullptr;
      if (seg == 0) dst = eps_lig + (size_t)i * dl;
      else if (eps_poc) dst = eps_poc + (size_t)(i - n_lig) * dp;
      if (dst) { dst[0] = v0 - m0; dst[1] = v1 - m1; dst[2] = v2 - m2; }
    }
  }
  if (nan) atomicOr(status, 1);
}

}  // namespace dsbdd
