// The training step of EGNNDynamics as ONE launch sequence per direction (round 6).
//
// Reference: lightning_modules.py:337-363 (training_step) -> conditional_model.py:202-330 / en_diffusion.py:336-469
// (ddpm.forward) -> dynamics.py:87-167 (EGNNDynamics.forward) -> egnn_new.py:225-244, 163-184, 31-58, 96-122 under autograd.
//
// Rounds 4 - 5 composed the differentiable forward from ~60 torch.autograd.Function nodes (train_hip.py): every node
// called one or two HIP kernels through the C-ABI and left the residual adds, SiLUs, concatenations, weight transposes,
// first-layer re-layouts and gradient accumulations to ~470 small aten launches per step (profiles/r6_train_kernel_stats.md).
// Here the whole network is walked in C++: dsbdd_train_net_forward / dsbdd_train_net_backward enqueue every launch of a
// direction on the caller's stream -- no host synchronisation, no tensor library in between -- and PyTorch sees ONE autograd
// node (train_net.py).  What the sequence consists of:
//   * one batched re-layout launch per step for ALL weights (tn_pack_kernel over a descriptor table: the transposes the
//     GEMM kernels want as B operand, the first edge-MLP layer split into per-node projections / distance columns) + one
//     for the edge-type tables; the packed copies live in a caller-provided persistent buffer,
//   * the existing kernels of the edge stages (edge_wave.h forward; train.h kernel A / wgrad / kernel B / gathers) and of the
//     node level (node_linear.h GEMMs with fused bias / residual; ordered split-K weight gradients; ordered column sums),
//   * a handful of elementwise kernels (SiLU and its derivative on the saved pre-activations, input split, output
//     assembly, per-sample mean backward),
//   * the parameter gradients written in nn.Linear layout [out][in] straight into the caller's gradient tensors; what the
//     factorised first layer produces in pieces (d W_pq, d w_d, d w_d0, d tab) is assembled by one batched launch at the
//     end of the backward pass (tn_unpack_kernel).
// Every sum has a fixed order (no atomics): gradients stay bitwise reproducible.  Activations are kept in a caller-provided
// workspace between the two calls (per block: h, the projections, the aggregate, the node MLP's pre-activation and
// activation, the coordinates); nothing of size [E][H] is kept (the edge activations are recomputed, train.h).
//
// Included by engine.hip behind the dsbdd_train_* building blocks it calls.
#pragma once

namespace dsbdd {

struct TnPackDesc {       // dT[c][r] = src[r][c] (ldT floats per row, padding columns zeroed), dP[r][c] = src[r][c]
  const float* src; int ld_src; int rows; int cols;
  float* dT; int ldT; int padT;     // padT: columns rows .. padT-1 of every dT row are cleared
  float* dP; int ldP;
};
struct TnTabDesc {        // tab[ty][o] = b1[o] + sum_e emb[ty][e] W1[o][col0 + e]
  const float* W1; int ld; int col0; const float* b1; const float* emb; int enf; float* tab; int H;
};
struct TnUnpackDesc {     // one edge MLP's first layer: d W1 [H][ld], d b1 [H], d emb partial [3][enf]
  float* dW1; int ld; float* db1; float* demb_part;
  const float* dWpq; int dq_off;      // d W_pq rows [0, H) = P part, rows [dq_off, dq_off + H) = Q part; [.][H]
  const float* d_vec;                 // [8][H]: d_wd, d_wd0, d_tab[0..2], ...
  const float* W1; const float* emb; int enf; int H;
};
struct TnCopyDesc { float* dst; const float* a; const float* b; int n; };    // dst[i] = a[i] (+ b[i])
// The gradient-assembly tables travel BY VALUE in the kernel arguments (<= 4 KB): they depend on the per-call workspace,
// and a per-step host-to-device copy of a table would need either pinned memory or a synchronisation to be safe.
constexpr int kTnUnpackPerLaunch = 16, kTnCopyPerLaunch = 64;
struct TnUnpackTable { TnUnpackDesc d[kTnUnpackPerLaunch]; };
struct TnCopyTable { TnCopyDesc d[kTnCopyPerLaunch]; };

__global__ void tn_pack_kernel(const TnPackDesc* descs) {
  const TnPackDesc d = descs[blockIdx.y];
  const int total = d.rows * d.cols;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int r = i / d.cols, c = i % d.cols;
    const float v = d.src[(size_t)r * d.ld_src + c];
    if (d.dT) d.dT[(size_t)c * d.ldT + r] = v;
    if (d.dP) d.dP[(size_t)r * d.ldP + c] = v;
  }
  if (d.dT && d.padT > d.rows) {
    const int pw = d.padT - d.rows, tot = pw * d.cols;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += gridDim.x * blockDim.x)
      d.dT[(size_t)(i / pw) * d.ldT + d.rows + i % pw] = 0.f;
  }
  if (d.dP && d.ldP > d.cols) {
    const int pw = d.ldP - d.cols, tot = pw * d.rows;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += gridDim.x * blockDim.x)
      d.dP[(size_t)(i / pw) * d.ldP + d.cols + i % pw] = 0.f;
  }
}

__global__ void tn_tab_kernel(const TnTabDesc* descs) {
  const TnTabDesc d = descs[blockIdx.x];
  for (int i = threadIdx.x; i < 3 * d.H; i += blockDim.x) {
    const int ty = i / d.H, o = i % d.H;
    float v = d.b1[o];
    for (int e = 0; e < d.enf; ++e) v = fmaf(d.emb[ty * d.enf + e], d.W1[(size_t)o * d.ld + d.col0 + e], v);
    d.tab[i] = v;
  }
}

__global__ void tn_unpack_kernel(const TnUnpackTable tab) {
  const TnUnpackDesc& d = tab.d[blockIdx.y];
  const int H = d.H, ld = d.ld, total = H * ld;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    const int o = i / ld, c = i % ld;
    float v;
    if (c < H) v = d.dWpq[(size_t)o * H + c];
    else if (c < 2 * H) v = d.dWpq[(size_t)(d.dq_off + o) * H + (c - H)];
    else if (c == 2 * H) v = d.d_vec[o];
    else if (c == 2 * H + 1) v = d.d_vec[H + o];
    else {   // d W1[o][2H + 2 + e] = sum_ty d_tab[ty][o] emb[ty][e]
      const int e = c - 2 * H - 2;
      v = 0.f;
      for (int ty = 0; ty < 3; ++ty) v = fmaf(d.d_vec[(2 + ty) * H + o], d.emb[ty * d.enf + e], v);
    }
    d.dW1[i] = v;
  }
  if (blockIdx.x == 0) {
    for (int o = threadIdx.x; o < H; o += blockDim.x)
      d.db1[o] = (d.d_vec[2 * H + o] + d.d_vec[3 * H + o]) + d.d_vec[4 * H + o];
    if (d.demb_part)      // d emb[ty][e] = sum_o d_tab[ty][o] W1[o][2H + 2 + e], one thread per entry, fixed order
      for (int i = threadIdx.x; i < 3 * d.enf; i += blockDim.x) {
        const int ty = i / d.enf, e = i % d.enf;
        float v = 0.f;
        for (int o = 0; o < H; ++o) v = fmaf(d.d_vec[(2 + ty) * H + o], d.W1[(size_t)o * ld + 2 * H + 2 + e], v);
        d.demb_part[i] = v;
      }
  }
}

__global__ void tn_copy_kernel(const TnCopyTable tab) {
  const TnCopyDesc& d = tab.d[blockIdx.y];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < d.n; i += gridDim.x * blockDim.x)
    d.dst[i] = d.b ? d.a[i] + d.b[i] : d.a[i];
}

// out[i] = sum over k < n_part of part[k * stride + i] in order (the edge-type embedding's gradient over the MLPs)
__global__ void tn_sum_parts_kernel(const float* part, int n_part, int stride, int n, float* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float v = 0.f;
  for (int k = 0; k < n_part; ++k) v += part[(size_t)k * stride + i];
  out[i] = v;
}

// dynamics.py:89-93,100: x = cat(ligand, pocket coordinates), the feature parts as contiguous matrices
__global__ void tn_split_inputs_kernel(const float* xh_l, int dl, const float* xh_p, int dp, int n_l, int n_p, float* x0,
                                       float* hf_l, float* hf_p) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_l + n_p) return;
  const bool lig = i < n_l;
  const float* src = lig ? xh_l + (size_t)i * dl : xh_p + (size_t)(i - n_l) * dp;
  const int nf = (lig ? dl : dp) - 3;
  float* hf = lig ? hf_l + (size_t)i * nf : hf_p + (size_t)(i - n_l) * nf;
  x0[3 * i] = src[0]; x0[3 * i + 1] = src[1]; x0[3 * i + 2] = src[2];
  for (int k = 0; k < nf; ++k) hf[k] = src[3 + k];
}

// dynamics.py:104-111: h = cat[h, t[mask]]; the padding columns of the row are cleared
__global__ void tn_time_col_kernel(float* h0, int JP, int J, const float* t, int t_count, const int* node_batch, int N) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  h0[(size_t)i * JP + J] = t[t_count == 1 ? 0 : node_batch[i]];
  for (int k = J + 1; k < JP; ++k) h0[(size_t)i * JP + k] = 0.f;
}

__global__ void tn_silu_kernel(const float* z, float* a, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[i] = silu(z[i]);
}
__global__ void tn_silu_bwd_kernel(const float* da, const float* z, float* dz, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { const float zz = z[i]; dz[i] = da[i] * dsilu_from(zz, sigmoidf_fast(zz)); }
}
__global__ void tn_add_kernel(float* dst, const float* src, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] += src[i];
}
__global__ void tn_sub_kernel(float* dst, const float* src, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] -= src[i];
}
__global__ void tn_cat_kernel(const float* h, const float* agg, float* out, int N, int H) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)N * 2 * H) return;
  const int r = (int)(i / (2 * H)), c = (int)(i % (2 * H));
  out[i] = c < H ? h[(size_t)r * H + c] : agg[(size_t)r * H + c - H];
}
// vel = x_final - x_in (dynamics.py:136); NaN -> 0 in training, status bit 1 otherwise (:155-159)
__global__ void tn_vel_kernel(const float* x_fin, const float* x0, float* vel, int n3, int zero_nan, int* status) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n3) return;
  float v = x_fin[i] - x0[i];
  if (v != v) { if (zero_nan) v = 0.f; else atomicOr(status, 1); }
  vel[i] = v;
}
// eps = cat[vel (- per-sample mean in joint mode, dynamics.py:161-164), decoded features]
__global__ void tn_out_kernel(const float* vel, const float* meanv, const int* node_batch, const float* eh_l, int a,
                              const float* eh_p, int r, int n_l, int n_p, float* eps_l, float* eps_p) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_l + n_p) return;
  const bool lig = i < n_l;
  const int nf = lig ? a : r;
  float* dst = lig ? eps_l + (size_t)i * (3 + a) : eps_p + (size_t)(i - n_l) * (3 + r);
  const float* eh = lig ? eh_l + (size_t)i * a : eh_p + (size_t)(i - n_l) * r;
  const int b = node_batch[i];
  for (int k = 0; k < 3; ++k) dst[k] = vel[3 * i + k] - (meanv ? meanv[3 * b + k] : 0.f);
  for (int k = 0; k < nf; ++k) dst[3 + k] = eh[k];
}
// the reverse: d_vel (joint: minus its per-sample mean, applied by the caller through meanv) and the feature gradients
__global__ void tn_split_grads_kernel(const float* d_l, int a, const float* d_p, int r, int n_l, int n_p, float* d_vel,
                                      float* deh_l, float* deh_p) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_l + n_p) return;
  const bool lig = i < n_l;
  const int nf = lig ? a : r;
  const float* src = lig ? d_l + (size_t)i * (3 + a) : d_p + (size_t)(i - n_l) * (3 + r);
  float* dh = lig ? deh_l + (size_t)i * a : deh_p + (size_t)(i - n_l) * r;
  for (int k = 0; k < 3; ++k) d_vel[3 * i + k] = src[k];
  for (int k = 0; k < nf; ++k) dh[k] = src[3 + k];
}
// x[i] -= m[batch(i)]   (the mean-removal of a vector field and its transpose are the same map)
__global__ void tn_sub_mean_kernel(float* x, const float* m, const int* node_batch, int N) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 3 * N) return;
  x[i] -= m[3 * node_batch[i / 3] + i % 3];
}
// SampleMean backward: d_x[i] += d_mean[b] / (number of nodes of sample b)
__global__ void tn_mean_bwd_kernel(float* d_x, const float* d_mean, const int* node_batch, const int* lig_off,
                                   const int* poc_off, int N) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 3 * N) return;
  const int b = node_batch[i / 3];
  const int cnt = (lig_off[b + 1] - lig_off[b]) + (poc_off[b + 1] - poc_off[b]);
  d_x[i] += d_mean[3 * b + i % 3] / (float)(cnt > 0 ? cnt : 1);
}
// d_hout's time column (dropped by dynamics.py:147-149) and padding carry no gradient
__global__ void tn_clear_cols_kernel(float* m, int ld, int c0, int N) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  for (int k = c0; k < ld; ++k) m[(size_t)i * ld + k] = 0.f;
}
// d_xh = cat[d_x, d_hf]
__global__ void tn_join_grads_kernel(const float* d_x, const float* dhf_l, int a, const float* dhf_p, int r, int n_l,
                                     int n_p, float* d_l, float* d_p) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_l + n_p) return;
  const bool lig = i < n_l;
  const int nf = lig ? a : r;
  float* dst = lig ? (d_l ? d_l + (size_t)i * (3 + a) : nullptr) : (d_p ? d_p + (size_t)(i - n_l) * (3 + r) : nullptr);
  if (!dst) return;
  const float* dh = lig ? dhf_l + (size_t)i * a : dhf_p + (size_t)(i - n_l) * r;
  for (int k = 0; k < 3; ++k) dst[k] = d_x[3 * i + k];
  for (int k = 0; k < nf; ++k) dst[3 + k] = dh[k];
}

}  // namespace dsbdd

using namespace dsbdd;

// ---------------------------------------------------------------------------------------------------------------------
// host side
static inline int tn_pad4(int v) { return (v + 3) & ~3; }
// The forward pass keeps z2 [E][H] of every edge MLP for the backward pass (4 E H bytes each: 93 MB at the reference
// batch) instead of recomputing the H x H layer there; DSBDD_TRAIN_STORE_Z2=0 (read when the handle is created): recompute
// (kernel A, rounds 4-5).
static bool tn_store_z2_env() { const char* v = getenv("DSBDD_TRAIN_STORE_Z2"); return !(v && atoi(v) == 0); }

// parameter tensors in the order of EGNNDynamics' own construction (diffsbdd_amd/synthetic.dynamics_param_shapes; the
// aliased cross_product_mlp.4.weight is NOT listed: it IS coord_mlp.4.weight, egnn_new.py:78,85,91)
struct TnParamIndex {
  int ae0, ae2, ad0, ad2, re0, re2, rd0, rd2;   // weight index; bias = +1
  int emb_tab;                                  // edge_embedding.weight or -1
  int emb, emb_out;                             // weight; bias = +1
  int blk0, per_blk, per_sub;                   // block b, sublayer s: base = blk0 + b per_blk + s per_sub
  int n;
  // offsets inside a sublayer: e1 w,b; e2 w,b; n1 w,b; n2 w,b; [att w,b]
  // offsets of the equivariant part (behind S sublayers): c1 w,b; c2 w,b; c3 w; [x1 w,b; x2 w,b]
  int S, att, n_mlp;
  int sub(int b, int s) const { return blk0 + b * per_blk + s * per_sub; }
  int eq(int b) const { return blk0 + b * per_blk + S * per_sub; }
};
static TnParamIndex tn_index(const dsbdd_config& c) {
  TnParamIndex p{};
  int i = 0;
  p.ae0 = i; i += 2; p.ae2 = i; i += 2; p.ad0 = i; i += 2; p.ad2 = i; i += 2;
  p.re0 = i; i += 2; p.re2 = i; i += 2; p.rd0 = i; i += 2; p.rd2 = i; i += 2;
  p.emb_tab = c.edge_embedding_dim > 0 ? i++ : -1;
  p.emb = i; i += 2; p.emb_out = i; i += 2;
  p.S = c.inv_sublayers; p.att = c.attention ? 1 : 0; p.n_mlp = c.reflection_equivariant ? 1 : 2;
  p.per_sub = 8 + 2 * p.att;
  p.per_blk = p.S * p.per_sub + 5 + (p.n_mlp == 2 ? 4 : 0);
  p.blk0 = i;
  p.n = i + c.n_layers * p.per_blk;
  return p;
}

static const int kTnSideDefault = SIDE_CO | SIDE_NODE_WG | SIDE_COORD_WG;     // DSBDD_TRAIN_STREAMS (bit mask of SIDE_*, engine.hip) overrides
struct dsbdd_train_net {
  dsbdd_config cfg;
  TnParamIndex ix;
  // cached descriptor tables (device copies live at the head of the caller's pack buffer)
  std::vector<const float*> last_params;
  void* last_pack = nullptr;
  int n_pack = 0, n_tab = 0;
  // side streams of the backward (TrainSide, engine.hip), created on the first backward call; DSBDD_TRAIN_STREAMS=0: none
  TrainSide side;
  bool side_ready = false;
  int side_mask = 0;
  bool store_z2 = true;
};

// sizes of one call
struct TnDims {
  int a, r, J, D, JP, H, L, S, G, M, A, enf;
  int64_t n_l, n_p, N, B, E;
};
static TnDims tn_dims(const dsbdd_config& c, const dsbdd_train_graph* g) {
  TnDims d{};
  d.a = c.atom_nf; d.r = c.residue_nf; d.J = c.joint_nf; d.D = d.J + 1; d.JP = tn_pad4(d.D); d.H = c.hidden_nf;
  d.L = c.n_layers; d.S = c.inv_sublayers; d.G = d.L * d.S; d.M = c.reflection_equivariant ? 1 : 2;
  d.enf = c.edge_embedding_dim > 0 ? c.edge_embedding_dim : 0; d.A = 2 + d.enf;
  d.n_l = g->n_lig; d.N = g->n_nodes; d.n_p = d.N - d.n_l; d.B = g->batch; d.E = g->n_edges;
  return d;
}

// ---- the persistent pack buffer: descriptor tables + every re-laid-out weight -------------------------------------
struct TnLin { float* WT; int ldT; float* Wp; int ldP; };    // WT [in][ldT] (forward B operand), Wp [out][ldP] (dX B operand)
struct TnEdgeMlp { float *wd, *wd0, *tab, *W2T; };
struct TnPack {
  TnPackDesc* d_pack; TnTabDesc* d_tab;      // device descriptor tables
  TnLin ae0, ae2, ad0, ad2, re0, re2, rd0, rd2, emb, emb_out;
  std::vector<float*> WpqT, Wpq;             // per message stage g: [H][2H], [2H][H]
  std::vector<TnEdgeMlp> gcl;                // per g
  std::vector<TnLin> n1, n2;                 // node MLP layers per g
  std::vector<float*> W4T, W4;               // per block: [H][2H M], [2H M][H]
  std::vector<TnEdgeMlp> eqm;                // per block x MLP
  size_t bytes;
};
static const int kTnMaxDesc = 1024;
static TnPack tn_carve_pack(char* base, const dsbdd_config& c) {
  TnPack p{};
  const int a = c.atom_nf, r = c.residue_nf, J = c.joint_nf, D = J + 1, H = c.hidden_nf, L = c.n_layers, S = c.inv_sublayers;
  const int M = c.reflection_equivariant ? 1 : 2;
  size_t off = 0;
  auto takeb = [&](size_t bytes) { char* q = base ? base + off : nullptr; off += al256(bytes); return q; };
  auto take = [&](size_t floats) { return reinterpret_cast<float*>(takeb(floats * 4)); };
  p.d_pack = reinterpret_cast<TnPackDesc*>(takeb(sizeof(TnPackDesc) * kTnMaxDesc));
  p.d_tab = reinterpret_cast<TnTabDesc*>(takeb(sizeof(TnTabDesc) * 256));
  auto lin = [&](int out, int in) { TnLin l; l.ldT = tn_pad4(out); l.WT = take((size_t)in * l.ldT); l.ldP = tn_pad4(in); l.Wp = take((size_t)out * l.ldP); return l; };
  p.ae0 = lin(2 * a, a); p.ae2 = lin(J, 2 * a); p.ad0 = lin(2 * a, J); p.ad2 = lin(a, 2 * a);
  p.re0 = lin(2 * r, r); p.re2 = lin(J, 2 * r); p.rd0 = lin(2 * r, J); p.rd2 = lin(r, 2 * r);
  p.emb = lin(H, D); p.emb_out = lin(D, H);
  for (int g = 0; g < L * S; ++g) {
    p.WpqT.push_back(take((size_t)H * 2 * H)); p.Wpq.push_back(take((size_t)2 * H * H));
    p.gcl.push_back(TnEdgeMlp{take(H), take(H), take(3 * (size_t)H), take((size_t)H * H)});
    p.n1.push_back(lin(H, 2 * H)); p.n2.push_back(lin(H, H));
  }
  for (int b = 0; b < L; ++b) {
    p.W4T.push_back(take((size_t)H * 2 * H * M)); p.W4.push_back(take((size_t)2 * H * M * H));
    for (int q = 0; q < M; ++q) p.eqm.push_back(TnEdgeMlp{take(H), take(H), take(3 * (size_t)H), take((size_t)H * H)});
  }
  p.bytes = off;
  return p;
}

// ---- the per-call workspace: graph-sized activations and gradients --------------------------------------------------
struct TnWs {
  float *x0, *hf_l, *hf_p, *ze_l, *ae_l, *ze_p, *ae_p, *h0, *hout, *zd_l, *ad_l, *zd_p, *ad_p, *eh_l, *eh_p, *vel, *meanv;
  std::vector<float*> h, x, mean, pq, agg, z, act, pq4;       // h [G + 1], x [L + 1], mean [L], pq/agg/z/act [G], pq4 [L]
  std::vector<float*> z2, z2c;                                // z2 [G] x [E][H], z2c [L] x [M][E][H]: the edge MLPs' second-layer pre-activations
  // backward
  float *d_vel, *deh_l, *deh_p, *d_h[2], *d_x[2], *d_xg, *d_pq4, *d_pq, *da, *dz, *d_agg, *xcat, *d_hout, *d_h0, *d_hf_l, *d_hf_p,
      *d_small, *gd0, *gd0_tot, *d_mean, *colscr, *dWpq, *d_vec, *demb_part, *wg;
  size_t wg_floats;
  char* scratch; size_t scratch_bytes;
  size_t bytes;
};
static TnWs tn_carve_ws(char* base, const dsbdd_config& c, const TnDims& d, bool store_z2) {
  TnWs w{};
  size_t off = 0;
  auto takeb = [&](size_t bytes) { char* q = base ? base + off : nullptr; off += al256(bytes); return q; };
  auto take = [&](size_t floats) { return reinterpret_cast<float*>(takeb((floats > 0 ? floats : 1) * 4)); };
  const size_t N = (size_t)d.N, H = d.H, nl = (size_t)d.n_l, np = (size_t)d.n_p, E = (size_t)(d.E > 0 ? d.E : 1);
  const int mx2 = 2 * (d.a > d.r ? d.a : d.r);
  w.x0 = take(3 * N); w.hf_l = take(nl * d.a); w.hf_p = take(np * d.r);
  w.ze_l = take(nl * 2 * d.a); w.ae_l = take(nl * 2 * d.a); w.ze_p = take(np * 2 * d.r); w.ae_p = take(np * 2 * d.r);
  w.h0 = take(N * d.JP); w.hout = take(N * d.JP);
  w.zd_l = take(nl * 2 * d.a); w.ad_l = take(nl * 2 * d.a); w.zd_p = take(np * 2 * d.r); w.ad_p = take(np * 2 * d.r);
  w.eh_l = take(nl * d.a); w.eh_p = take(np * d.r); w.vel = take(3 * N); w.meanv = take(3 * (size_t)d.B);
  for (int g = 0; g <= d.G; ++g) w.h.push_back(take(N * H));
  for (int b = 0; b <= d.L; ++b) w.x.push_back(take(3 * N));
  for (int b = 0; b < d.L; ++b) { w.mean.push_back(take(3 * (size_t)d.B)); w.pq4.push_back(take(N * 2 * H * d.M)); }
  for (int g = 0; g < d.G; ++g) { w.pq.push_back(take(N * 2 * H)); w.agg.push_back(take(N * H)); w.z.push_back(take(N * H)); w.act.push_back(take(N * H)); }
  for (int g = 0; g < d.G; ++g) w.z2.push_back(store_z2 ? take(E * H) : nullptr);
  for (int b = 0; b < d.L; ++b) w.z2c.push_back(store_z2 ? take(E * H * d.M) : nullptr);
  w.d_vel = take(3 * N); w.deh_l = take(nl * d.a); w.deh_p = take(np * d.r);
  w.d_h[0] = take(N * H); w.d_h[1] = take(N * H); w.d_x[0] = take(3 * N); w.d_x[1] = take(3 * N); w.d_xg = take(3 * N);
  w.d_pq4 = take(N * 2 * H * d.M); w.d_pq = take(N * 2 * H); w.da = take(N * H); w.dz = take(N * H); w.d_agg = take(N * H);
  w.xcat = take(N * 2 * H); w.d_hout = take(N * d.JP); w.d_h0 = take(N * d.JP);
  w.d_hf_l = take(nl * d.a); w.d_hf_p = take(np * d.r); w.d_small = take((nl > np ? nl : np) * mx2 * 2);
  w.gd0 = take(2 * E); w.gd0_tot = take(E); w.d_mean = take(3 * (size_t)d.B);
  w.colscr = take(((N + 31) / 32 + 1) * (size_t)(2 * H > (size_t)d.JP ? 2 * H : d.JP));
  w.dWpq = take((size_t)(d.G * 2 + d.L * 2 * d.M) * H * H);
  w.d_vec = take((size_t)(d.G + d.L * d.M) * 8 * H);
  w.demb_part = take((size_t)(d.G + d.L * d.M) * 3 * (d.enf > 0 ? d.enf : 1));
  {   // the largest split-K plan over the node-level weight gradients of this network (K <= N rows)
    const int shapes[][2] = {{d.H, d.H}, {d.H, 2 * d.H}, {2 * d.H, d.H}, {2 * d.H * d.M, d.H}, {d.D, d.H}, {d.H, d.D}, {d.J, mx2},
                             {mx2, d.J}, {mx2, mx2}};
    w.wg_floats = 0;
    for (const auto& sh : shapes) { const size_t f = wgrad_floats_upto((int64_t)N, sh[0], sh[1]); if (f > w.wg_floats) w.wg_floats = f; }
  }
  w.wg = take(w.wg_floats);
  // (two scratch sets when a coordinate stage has two edge MLPs: their backward chains run side by side)
  w.scratch_bytes = carve_train(nullptr, d.H, d.N, d.E).bytes * (d.M == 2 ? 2 : 1);
  w.scratch = takeb(w.scratch_bytes);
  (void)c;
  w.bytes = off;
  return w;
}

static inline unsigned tn_blocks(size_t n, int per = 256) { return (unsigned)((n + per - 1) / per); }

// y = x W^T (+ b) (+ R) through node_linear: x [M][lda] (K columns), WT [K][ldT]
static int tn_lin(hipStream_t s, const float* x, int lda, int K, const float* x2, int lda2, int K2, const float* WT, int ldT,
                  const float* bias, const float* R, int ldr, float* y, int ldc, int64_t M, int Nout) {
  if (M <= 0) return DSBDD_OK;
  HIP_TRY(nl(s, x, lda, K, x2, lda2, K2, WT, ldT, bias, R, ldr, y, ldc, M, Nout, 0));
  return DSBDD_OK;
}
static int tn_wgrad(hipStream_t s, const float* dy, int ldy, const float* x, int ldx, int64_t K, int M, int Nn, float* dW,
                    const TnWs& w) {
  if (K <= 0) { HIP_TRY(hipMemsetAsync(dW, 0, (size_t)M * Nn * 4, s)); return DSBDD_OK; }
  return wgrad_impl(s, dy, ldy, x, ldx, K, M, Nn, dW, w.wg, w.wg_floats);
}
static int tn_colsum(hipStream_t s, const float* dy, int ldy, int64_t M, int Nn, float* db, const TnWs& w) {
  if (M <= 0) { HIP_TRY(hipMemsetAsync(db, 0, (size_t)Nn * 4, s)); return DSBDD_OK; }
  HIP_TRY(reduce_parts(s, dy, (int)M, (size_t)ldy, Nn, db, w.colscr));
  return DSBDD_OK;
}

// descriptor tables of the weight re-layout; uploaded when the parameter / pack pointers changed
static int tn_prepare_pack(dsbdd_train_net* net, hipStream_t s, const float* const* P, char* pack_base, const TnPack& pk) {
  const dsbdd_config& c = net->cfg;
  const TnParamIndex& ix = net->ix;
  const int a = c.atom_nf, r = c.residue_nf, J = c.joint_nf, D = J + 1, H = c.hidden_nf, L = c.n_layers, S = c.inv_sublayers;
  const int M = ix.n_mlp, enf = c.edge_embedding_dim > 0 ? c.edge_embedding_dim : 0, A = 2 + enf, ld1 = 2 * H + A;
  bool same = net->last_pack == pack_base && (int)net->last_params.size() == ix.n;
  for (int i = 0; same && i < ix.n; ++i) same = net->last_params[i] == P[i];
  if (!same) {
    std::vector<TnPackDesc> pd;
    std::vector<TnTabDesc> td;
    auto lin = [&](int wi, const TnLin& l, int out, int in) {
      pd.push_back(TnPackDesc{P[wi], in, out, in, l.WT, l.ldT, l.ldT, l.Wp, l.ldP});
    };
    lin(ix.ae0, pk.ae0, 2 * a, a); lin(ix.ae2, pk.ae2, J, 2 * a); lin(ix.ad0, pk.ad0, 2 * a, J); lin(ix.ad2, pk.ad2, a, 2 * a);
    lin(ix.re0, pk.re0, 2 * r, r); lin(ix.re2, pk.re2, J, 2 * r); lin(ix.rd0, pk.rd0, 2 * r, J); lin(ix.rd2, pk.rd2, r, 2 * r);
    lin(ix.emb, pk.emb, H, D); lin(ix.emb_out, pk.emb_out, D, H);
    const float* emb = ix.emb_tab >= 0 ? P[ix.emb_tab] : nullptr;
    // first layer of an edge MLP W1 [H][2H + A]: the two [H][H] blocks as columns [c0, c0 + H) / [c0 + H, c0 + 2H) of
    // WT [H][ldT] (rows = k) and as rows [r0, r0 + H) / [r0 + H, r0 + 2H) of Wp [.][H]; the distance columns; the table
    auto first = [&](int wi, float* WT, int ldT, int c0, float* Wp, int r0, const TnEdgeMlp& m) {
      const float* W1 = P[wi];
      pd.push_back(TnPackDesc{W1, ld1, H, H, WT + c0, ldT, 0, Wp + (size_t)r0 * H, H});
      pd.push_back(TnPackDesc{W1 + H, ld1, H, H, WT + c0 + H, ldT, 0, Wp + (size_t)(r0 + H) * H, H});
      pd.push_back(TnPackDesc{W1 + 2 * H, ld1, H, 1, m.wd, H, 0, nullptr, 0});
      pd.push_back(TnPackDesc{W1 + 2 * H + 1, ld1, H, 1, m.wd0, H, 0, nullptr, 0});
      td.push_back(TnTabDesc{W1, ld1, 2 * H + 2, P[wi + 1], emb, enf, m.tab, H});
    };
    for (int b = 0; b < L; ++b) {
      for (int sl = 0; sl < S; ++sl) {
        const int g = b * S + sl, base = ix.sub(b, sl);
        first(base, pk.WpqT[g], 2 * H, 0, pk.Wpq[g], 0, pk.gcl[g]);
        pd.push_back(TnPackDesc{P[base + 2], H, H, H, pk.gcl[g].W2T, H, 0, nullptr, 0});
        pd.push_back(TnPackDesc{P[base + 4], 2 * H, H, 2 * H, pk.n1[g].WT, pk.n1[g].ldT, pk.n1[g].ldT, nullptr, 0});
        pd.push_back(TnPackDesc{P[base + 6], H, H, H, pk.n2[g].WT, pk.n2[g].ldT, pk.n2[g].ldT, nullptr, 0});
      }
      const int e = ix.eq(b);
      first(e, pk.W4T[b], 2 * H * M, 0, pk.W4[b], 0, pk.eqm[b * M]);
      pd.push_back(TnPackDesc{P[e + 2], H, H, H, pk.eqm[b * M].W2T, H, 0, nullptr, 0});
      if (M == 2) {
        first(e + 5, pk.W4T[b], 2 * H * M, 2 * H, pk.W4[b], 2 * H, pk.eqm[b * M + 1]);
        pd.push_back(TnPackDesc{P[e + 7], H, H, H, pk.eqm[b * M + 1].W2T, H, 0, nullptr, 0});
      }
    }
    if ((int)pd.size() > kTnMaxDesc || td.size() > 256) return fail(DSBDD_ERR_CAPACITY, "too many layers for the descriptor tables");
    HIP_TRY(hipMemcpyAsync(pk.d_pack, pd.data(), pd.size() * sizeof(TnPackDesc), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(pk.d_tab, td.data(), td.size() * sizeof(TnTabDesc), hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));        // (the host vectors go out of scope; once per set of pointers)
    net->n_pack = (int)pd.size(); net->n_tab = (int)td.size();
    net->last_params.assign(P, P + ix.n); net->last_pack = pack_base;
  }
  hipLaunchKernelGGL(tn_pack_kernel, dim3(32, net->n_pack), dim3(256), 0, s, (const TnPackDesc*)pk.d_pack);
  HIP_TRY(hipGetLastError());
  hipLaunchKernelGGL(tn_tab_kernel, dim3(net->n_tab), dim3(256), 0, s, (const TnTabDesc*)pk.d_tab);
  HIP_TRY(hipGetLastError());
  return DSBDD_OK;
}

static dsbdd_train_mlp tn_mlp(const float* pq, int col_p, int col_q, int ld, const TnEdgeMlp& m, const float* W2, const float* b2,
                              const float* head, const float* head_b) {
  dsbdd_train_mlp t{};
  t.P = pq + col_p; t.Q = pq + col_q; t.ldpq = ld; t.wd = m.wd; t.wd0 = m.wd0; t.tab = m.tab; t.W2 = W2; t.W2T = m.W2T; t.b2 = b2;
  t.head = head; t.head_b = head_b;
  return t;
}

extern "C" {

int dsbdd_train_net_create(const dsbdd_config* cfg, dsbdd_train_net** out) {
  if (!cfg || !out) return fail(DSBDD_ERR_ARG, "null argument");
  if (!train_h_ok(cfg->hidden_nf) || cfg->n_layers < 1 || cfg->inv_sublayers < 1 || cfg->atom_nf < 1 || cfg->residue_nf < 1)
    return fail(DSBDD_ERR_ARG, "unsupported configuration");
  auto* n = new dsbdd_train_net();
  n->cfg = *cfg; n->ix = tn_index(*cfg);
  { const char* v = getenv("DSBDD_TRAIN_STREAMS"); n->side_mask = v ? atoi(v) & 15 : kTnSideDefault; }
  n->store_z2 = tn_store_z2_env();
  *out = n;
  return DSBDD_OK;
}
void dsbdd_train_net_destroy(dsbdd_train_net* n) {
  if (n && n->side_ready) n->side.destroy();
  delete n;
}
int dsbdd_train_net_param_count(const dsbdd_train_net* n) { return n ? n->ix.n : 0; }
size_t dsbdd_train_net_pack_bytes(const dsbdd_train_net* n) { return n ? tn_carve_pack(nullptr, n->cfg).bytes : 0; }
size_t dsbdd_train_net_workspace_bytes(const dsbdd_train_net* n, const dsbdd_train_graph* g) {
  if (!n || !graph_ok(g)) return 0;
  return tn_carve_ws(nullptr, n->cfg, tn_dims(n->cfg, g), n->store_z2).bytes;
}

int dsbdd_train_net_forward(dsbdd_train_net* net, void* stream, const dsbdd_train_graph* g, const float* const* params,
                            void* pack, size_t pack_bytes, void* ws, size_t ws_bytes, const float* xh_lig,
                            const float* xh_pocket, const float* t, int64_t t_count, int32_t zero_nan, float* eps_lig,
                            float* eps_pocket, int32_t* status) {
  StreamDevice stream_device_(stream);
  if (!net || !graph_ok(g) || !params || !pack || !ws || !t || !status || t_count < 1 || (g->n_lig > 0 && (!xh_lig || !eps_lig)) ||
      (g->n_nodes > g->n_lig && (!xh_pocket || !eps_pocket)))
    return fail(DSBDD_ERR_ARG, "bad argument");
  const dsbdd_config& c = net->cfg;
  const TnParamIndex& ix = net->ix;
  for (int i = 0; i < ix.n; ++i) if (!params[i]) return fail(DSBDD_ERR_ARG, "null parameter " + std::to_string(i));
  const TnDims d = tn_dims(c, g);
  const TnPack pk = tn_carve_pack(static_cast<char*>(pack), c);
  if (pk.bytes > pack_bytes) return fail(DSBDD_ERR_CAPACITY, "pack buffer too small (dsbdd_train_net_pack_bytes)");
  const TnWs w = tn_carve_ws(static_cast<char*>(ws), c, d, net->store_z2);
  if (w.bytes > ws_bytes) return fail(DSBDD_ERR_CAPACITY, "workspace too small (dsbdd_train_net_workspace_bytes)");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const float* const* P = params;
  const int H = d.H, a = d.a, r = d.r, J = d.J, JP = d.JP, N = (int)d.N, nl_ = (int)d.n_l, np_ = (int)d.n_p, M = d.M;
  { const int rc = tn_prepare_pack(net, s, P, static_cast<char*>(pack), pk); if (rc != DSBDD_OK) return rc; }
  // dynamics.py:89-111: split, encoders, time feature
  hipLaunchKernelGGL(tn_split_inputs_kernel, dim3(tn_blocks(N)), dim3(256), 0, s, xh_lig, 3 + a, xh_pocket, 3 + r, nl_, np_, w.x0,
                     w.hf_l, w.hf_p);
  HIP_TRY(hipGetLastError());
  auto mlp2 = [&](const float* x, int lda, int in, int mid, int out, const TnLin& l0, const float* b0, const TnLin& l1,
                  const float* b1, float* z, float* act, float* y, int ldc, int64_t rows) -> int {
    if (rows <= 0) return DSBDD_OK;
    int rc = tn_lin(s, x, lda, in, nullptr, 0, 0, l0.WT, l0.ldT, b0, nullptr, 0, z, mid, rows, mid); if (rc) return rc;
    hipLaunchKernelGGL(tn_silu_kernel, dim3(tn_blocks((size_t)rows * mid)), dim3(256), 0, s, (const float*)z, act, (size_t)rows * mid);
    HIP_TRY(hipGetLastError());
    return tn_lin(s, act, mid, mid, nullptr, 0, 0, l1.WT, l1.ldT, b1, nullptr, 0, y, ldc, rows, out);
  };
  { int rc = mlp2(w.hf_l, a, a, 2 * a, J, pk.ae0, P[ix.ae0 + 1], pk.ae2, P[ix.ae2 + 1], w.ze_l, w.ae_l, w.h0, JP, nl_); if (rc) return rc; }
  { int rc = mlp2(w.hf_p, r, r, 2 * r, J, pk.re0, P[ix.re0 + 1], pk.re2, P[ix.re2 + 1], w.ze_p, w.ae_p, w.h0 + (size_t)nl_ * JP, JP, np_); if (rc) return rc; }
  hipLaunchKernelGGL(tn_time_col_kernel, dim3(tn_blocks(N)), dim3(256), 0, s, w.h0, JP, J, t, (int)t_count, g->node_batch, N);
  HIP_TRY(hipGetLastError());
  // egnn_new.py:233: embedding
  { int rc = tn_lin(s, w.h0, JP, d.D, nullptr, 0, 0, pk.emb.WT, pk.emb.ldT, P[ix.emb + 1], nullptr, 0, w.h[0], H, N, H); if (rc) return rc; }
  HIP_TRY(hipMemcpyAsync(w.x[0], w.x0, (size_t)N * 12, hipMemcpyDeviceToDevice, s));
  const int64_t n_upd = c.update_pocket_coords ? d.N : d.n_l;
  for (int b = 0; b < d.L; ++b) {
    if (M == 2) { const int rc = dsbdd_train_sample_mean(stream, w.x[b], g, w.mean[b]); if (rc) return rc; }
    for (int sl = 0; sl < d.S; ++sl) {
      const int gi = b * d.S + sl, base = ix.sub(b, sl);
      const float* hin = w.h[gi];
      { int rc = tn_lin(s, hin, H, H, nullptr, 0, 0, pk.WpqT[gi], 2 * H, nullptr, nullptr, 0, w.pq[gi], 2 * H, N, 2 * H); if (rc) return rc; }
      const dsbdd_train_mlp m = tn_mlp(w.pq[gi], 0, H, 2 * H, pk.gcl[gi], P[base + 2], P[base + 3],
                                       ix.att ? P[base + 8] : nullptr, ix.att ? P[base + 9] : nullptr);
      { const int rc = gcl_forward_impl(stream, H, g, &m, w.x[b], c.normalization_factor, w.agg[gi], w.scratch, w.scratch_bytes, w.z2[gi]); if (rc) return rc; }
      // node MLP (egnn_new.py:53-58): h + W2 SiLU(W1 [h | agg] + b1) + b2
      { int rc = tn_lin(s, hin, H, H, w.agg[gi], H, H, pk.n1[gi].WT, pk.n1[gi].ldT, P[base + 5], nullptr, 0, w.z[gi], H, N, H); if (rc) return rc; }
      hipLaunchKernelGGL(tn_silu_kernel, dim3(tn_blocks((size_t)N * H)), dim3(256), 0, s, (const float*)w.z[gi], w.act[gi], (size_t)N * H);
      HIP_TRY(hipGetLastError());
      { int rc = tn_lin(s, w.act[gi], H, H, nullptr, 0, 0, pk.n2[gi].WT, pk.n2[gi].ldT, P[base + 7], hin, H, w.h[gi + 1], H, N, H); if (rc) return rc; }
    }
    // coordinate update (egnn_new.py:96-122)
    const int e = ix.eq(b);
    const float* hb = w.h[(b + 1) * d.S];
    { int rc = tn_lin(s, hb, H, H, nullptr, 0, 0, pk.W4T[b], 2 * H * M, nullptr, nullptr, 0, w.pq4[b], 2 * H * M, N, 2 * H * M); if (rc) return rc; }
    dsbdd_train_mlp mm[2];
    mm[0] = tn_mlp(w.pq4[b], 0, H, 2 * H * M, pk.eqm[b * M], P[e + 2], P[e + 3], P[e + 4], nullptr);
    if (M == 2) mm[1] = tn_mlp(w.pq4[b], 2 * H, 3 * H, 2 * H * M, pk.eqm[b * M + 1], P[e + 7], P[e + 8], P[e + 4], nullptr);
    { const int rc = coord_forward_impl(stream, H, g, mm, M, w.x[b], M == 2 ? w.mean[b] : nullptr, n_upd, c.norm_constant,
                                       c.coords_range, c.use_tanh, c.normalization_factor, w.x[b + 1], w.scratch, w.scratch_bytes,
                                       w.z2c[b], (size_t)(d.E > 0 ? d.E : 1) * H); if (rc) return rc; }
  }
  // egnn_new.py:241-243, dynamics.py:136-167
  { int rc = tn_lin(s, w.h[d.G], H, H, nullptr, 0, 0, pk.emb_out.WT, pk.emb_out.ldT, P[ix.emb_out + 1], nullptr, 0, w.hout, JP, N, d.D); if (rc) return rc; }
  { int rc = mlp2(w.hout, JP, J, 2 * a, a, pk.ad0, P[ix.ad0 + 1], pk.ad2, P[ix.ad2 + 1], w.zd_l, w.ad_l, w.eh_l, a, nl_); if (rc) return rc; }
  { int rc = mlp2(w.hout + (size_t)nl_ * JP, JP, J, 2 * r, r, pk.rd0, P[ix.rd0 + 1], pk.rd2, P[ix.rd2 + 1], w.zd_p, w.ad_p, w.eh_p, r, np_); if (rc) return rc; }
  hipLaunchKernelGGL(tn_vel_kernel, dim3(tn_blocks(3 * (size_t)N)), dim3(256), 0, s, (const float*)w.x[d.L], (const float*)w.x0, w.vel,
                     3 * N, (int)zero_nan, status);
  HIP_TRY(hipGetLastError());
  if (c.update_pocket_coords) { const int rc = dsbdd_train_sample_mean(stream, w.vel, g, w.meanv); if (rc) return rc; }
  hipLaunchKernelGGL(tn_out_kernel, dim3(tn_blocks(N)), dim3(256), 0, s, (const float*)w.vel,
                     c.update_pocket_coords ? (const float*)w.meanv : (const float*)nullptr, g->node_batch, (const float*)w.eh_l, a,
                     (const float*)w.eh_p, r, nl_, np_, eps_lig, eps_pocket);
  HIP_TRY(hipGetLastError());
  return DSBDD_OK;
}

int dsbdd_train_net_backward(dsbdd_train_net* net, void* stream, const dsbdd_train_graph* g, const float* const* params,
                             float* const* grads, void* pack, size_t pack_bytes, void* ws, size_t ws_bytes,
                             int64_t e_upd, const float* d_eps_lig, const float* d_eps_pocket, float* d_xh_lig,
                             float* d_xh_pocket) {
  StreamDevice stream_device_(stream);
  if (!net || !graph_ok(g) || !g->rev || !params || !grads || !pack || !ws) return fail(DSBDD_ERR_ARG, "bad argument");
  const dsbdd_config& c = net->cfg;
  const TnParamIndex& ix = net->ix;
  for (int i = 0; i < ix.n; ++i) if (!params[i] || !grads[i]) return fail(DSBDD_ERR_ARG, "null parameter / gradient " + std::to_string(i));
  const TnDims d = tn_dims(c, g);
  if ((d.n_l > 0 && !d_eps_lig) || (d.n_p > 0 && !d_eps_pocket)) return fail(DSBDD_ERR_ARG, "null output gradient");
  const TnPack pk = tn_carve_pack(static_cast<char*>(pack), c);
  const TnWs w = tn_carve_ws(static_cast<char*>(ws), c, d, net->store_z2);
  if (pk.bytes > pack_bytes || w.bytes > ws_bytes) return fail(DSBDD_ERR_CAPACITY, "buffer too small");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (net->side_ready) {          // (the StreamDevice guard above made the stream's device current)
    int dev = -1;
    HIP_TRY(hipGetDevice(&dev));
    if (dev != net->side.device) { net->side.destroy(); net->side_ready = false; }
  }
  if (net->side_mask && !net->side_ready) {
    HIP_TRY(net->side.create());
    net->side.mask = net->side_mask;
    net->side_ready = true;
  }
  TrainSide* sd = net->side_mask ? &net->side : nullptr;
  hipStream_t sw = sd && (sd->mask & SIDE_NODE_WG) ? sd->wg : s;      // the stream of the block loop's weight gradients (w.wg / w.colscr are its scratch there)
  auto fork_w = [&]() -> int { if (sw != s) HIP_TRY(sd->link(s, sd->wg)); return DSBDD_OK; };
  // the bias gradients (ordered column sums, w.colscr) of the sublayers go to the OTHER side stream, idle outside the
  // coordinate stages: the weight-gradient chain of a sublayer (3 GEMMs + their reductions + 2 two-stage column sums) was
  // longer than the main chain's node work beside it, and the main chain waited for it in front of the W2 gradient
  hipStream_t sb = sw != s ? sd->co : s;
  auto fork_b = [&]() -> int { if (sb != s) HIP_TRY(sd->link(s, sd->co)); return DSBDD_OK; };
  const float* const* P = params;
  float* const* G = grads;
  const int H = d.H, a = d.a, r = d.r, J = d.J, JP = d.JP, N = (int)d.N, nl_ = (int)d.n_l, np_ = (int)d.n_p, M = d.M;
  const bool want_in = d_xh_lig || d_xh_pocket;
  const int64_t E = d.E;
  const int64_t n_upd = c.update_pocket_coords ? d.N : d.n_l;
  if (c.update_pocket_coords) e_upd = E;         // (else: row_ptr[n_lig], the edge prefix of the ligand rows, a host value)
  if (e_upd < 0 || e_upd > E) return fail(DSBDD_ERR_ARG, "e_upd out of range");
  std::vector<TnUnpackDesc> ud;      // gradient-assembly tables, launched at the end (by value, see TnUnpackTable)
  std::vector<TnCopyDesc> cd;
  const float* emb = ix.emb_tab >= 0 ? P[ix.emb_tab] : nullptr;
  const int ld1 = 2 * H + d.A;
  int mlp_no = 0;                        // running index of the edge MLPs (d_vec / dWpq / demb_part slots)
  auto vec_of = [&](int k) { return w.d_vec + (size_t)k * 8 * H; };

  // dynamics.py:136-167 backward: outputs -> d_vel, decoder inputs
  hipLaunchKernelGGL(tn_split_grads_kernel, dim3(tn_blocks(N)), dim3(256), 0, s, d_eps_lig, a, d_eps_pocket, r, nl_, np_, w.d_vel,
                     w.deh_l, w.deh_p);
  HIP_TRY(hipGetLastError());
  if (c.update_pocket_coords) {          // vel - mean(vel): the same projection on the gradient
    { const int rc = dsbdd_train_sample_mean(stream, w.d_vel, g, w.meanv); if (rc) return rc; }
    hipLaunchKernelGGL(tn_sub_mean_kernel, dim3(tn_blocks(3 * (size_t)N)), dim3(256), 0, s, w.d_vel, (const float*)w.meanv, g->node_batch, N);
    HIP_TRY(hipGetLastError());
  }
  // decoders: y = W1 SiLU(W0 x + b0) + b1 on hout[:, :J]
  auto mlp2_bwd = [&](const float* dy, int lddy, int out, int mid, int in, const float* x, int ldx, const float* z, const float* act,
                      int wi0, const TnLin& l0, int wi1, const TnLin& l1, float* dx, int lddx, int64_t rows, bool want_dx) -> int {
    float* dact = w.d_small;
    float* dzz = w.d_small + (size_t)(rows > 0 ? rows : 0) * mid;
    int rc;
    rc = tn_wgrad(s, dy, lddy, act, mid, rows, out, mid, G[wi1], w); if (rc) return rc;
    rc = tn_colsum(s, dy, lddy, rows, out, G[wi1 + 1], w); if (rc) return rc;
    if (rows > 0) {
      rc = tn_lin(s, dy, lddy, out, nullptr, 0, 0, l1.Wp, l1.ldP, nullptr, nullptr, 0, dact, mid, rows, mid); if (rc) return rc;
      hipLaunchKernelGGL(tn_silu_bwd_kernel, dim3(tn_blocks((size_t)rows * mid)), dim3(256), 0, s, (const float*)dact, z, dzz, (size_t)rows * mid);
      HIP_TRY(hipGetLastError());
    }
    rc = tn_wgrad(s, dzz, mid, x, ldx, rows, mid, in, G[wi0], w); if (rc) return rc;
    rc = tn_colsum(s, dzz, mid, rows, mid, G[wi0 + 1], w); if (rc) return rc;
    if (want_dx && rows > 0) { rc = tn_lin(s, dzz, mid, mid, nullptr, 0, 0, l0.Wp, l0.ldP, nullptr, nullptr, 0, dx, lddx, rows, in); if (rc) return rc; }
    return DSBDD_OK;
  };
  { int rc = mlp2_bwd(w.deh_l, a, a, 2 * a, J, w.hout, JP, w.zd_l, w.ad_l, ix.ad0, pk.ad0, ix.ad2, pk.ad2, w.d_hout, JP, nl_, true); if (rc) return rc; }
  { int rc = mlp2_bwd(w.deh_p, r, r, 2 * r, J, w.hout + (size_t)nl_ * JP, JP, w.zd_p, w.ad_p, ix.rd0, pk.rd0, ix.rd2, pk.rd2,
                      w.d_hout + (size_t)nl_ * JP, JP, np_, true); if (rc) return rc; }
  hipLaunchKernelGGL(tn_clear_cols_kernel, dim3(tn_blocks(N)), dim3(256), 0, s, w.d_hout, JP, J, N);
  HIP_TRY(hipGetLastError());
  // embedding_out
  int cur = 0;
  { int rc = tn_wgrad(s, w.d_hout, JP, w.h[d.G], H, N, d.D, H, G[ix.emb_out], w); if (rc) return rc; }
  { int rc = tn_colsum(s, w.d_hout, JP, N, d.D, G[ix.emb_out + 1], w); if (rc) return rc; }
  { int rc = tn_lin(s, w.d_hout, JP, d.D, nullptr, 0, 0, pk.emb_out.Wp, pk.emb_out.ldP, nullptr, nullptr, 0, w.d_h[cur], H, N, H); if (rc) return rc; }
  // d_x of the block outputs: x_L enters vel only
  int cx = 0;
  HIP_TRY(hipMemcpyAsync(w.d_x[cx], w.d_vel, (size_t)N * 12, hipMemcpyDeviceToDevice, s));
  if (want_in) HIP_TRY(hipMemsetAsync(w.gd0_tot, 0, (size_t)(E > 0 ? E : 1) * 4, s));
  auto add_gd0 = [&](const float* gd) -> int {
    if (!want_in || E <= 0) return DSBDD_OK;
    hipLaunchKernelGGL(tn_add_kernel, dim3(tn_blocks((size_t)E)), dim3(256), 0, s, w.gd0_tot, gd, (size_t)E);
    HIP_TRY(hipGetLastError());
    return DSBDD_OK;
  };

  for (int b = d.L - 1; b >= 0; --b) {
    const int e = ix.eq(b);
    const float* hb = w.h[(b + 1) * d.S];
    // ---- coordinate update backward: d_x[cx] = gradient w.r.t. x_{b+1}
    dsbdd_train_mlp mm[2];
    mm[0] = tn_mlp(w.pq4[b], 0, H, 2 * H * M, pk.eqm[b * M], P[e + 2], P[e + 3], P[e + 4], nullptr);
    if (M == 2) mm[1] = tn_mlp(w.pq4[b], 2 * H, 3 * H, 2 * H * M, pk.eqm[b * M + 1], P[e + 7], P[e + 8], P[e + 4], nullptr);
    dsbdd_train_mlp_grad og[2];
    const int k0 = mlp_no;
    float* dW4 = w.dWpq + (size_t)(d.G * 2 + b * 2 * M) * H * H;       // [2H M][H]
    for (int q = 0; q < M; ++q) {
      og[q].dP = w.d_pq4 + 2 * H * q; og[q].dQ = w.d_pq4 + 2 * H * q + H; og[q].ldo = 2 * H * M;
      og[q].d_vec = vec_of(k0 + q); og[q].d_W2 = G[q == 0 ? e + 2 : e + 7]; og[q].gd0 = w.gd0 + (size_t)q * (E > 0 ? E : 1);
    }
    float* dxo = w.d_x[cx];
    float* dxi = w.d_x[cx ^ 1];
    if (e_upd > 0 && n_upd > 0) {
      if (want_in) HIP_TRY(hipMemsetAsync(w.gd0, 0, (size_t)2 * (E > 0 ? E : 1) * 4, s));
      { const int rc = coord_backward_impl(stream, H, g, mm, M, w.x[b], M == 2 ? w.mean[b] : nullptr, n_upd, e_upd, c.norm_constant,
                                          c.coords_range, c.use_tanh, c.normalization_factor, dxo, og, dxi,
                                          M == 2 ? w.d_mean : nullptr, w.scratch, w.scratch_bytes, sd, w.z2c[b],
                                          (size_t)(E > 0 ? E : 1) * H); if (rc) return rc; }
      for (int q = 0; q < M; ++q) { const int rc = add_gd0(og[q].gd0); if (rc) return rc; }
      // identity path x -> x_out
      hipLaunchKernelGGL(tn_add_kernel, dim3(tn_blocks(3 * (size_t)N)), dim3(256), 0, s, dxi, (const float*)dxo, 3 * (size_t)N);
      HIP_TRY(hipGetLastError());
      if (M == 2) {
        hipLaunchKernelGGL(tn_mean_bwd_kernel, dim3(tn_blocks(3 * (size_t)N)), dim3(256), 0, s, dxi, (const float*)w.d_mean, g->node_batch,
                           g->lig_off, g->poc_off, N);
        HIP_TRY(hipGetLastError());
      }
      // d_h += d_pq4 W4; d W4 = d_pq4^T h
      { int rc = tn_lin(s, w.d_pq4, 2 * H * M, 2 * H * M, nullptr, 0, 0, pk.W4[b], H, nullptr, w.d_h[cur], H, w.d_h[cur ^ 1], H, N, H); if (rc) return rc; }
      cur ^= 1;
      { int rc = fork_w(); if (rc) return rc; }
      { int rc = tn_wgrad(sw, w.d_pq4, 2 * H * M, hb, H, N, 2 * H * M, H, dW4, w); if (rc) return rc; }
    } else {
      HIP_TRY(hipMemcpyAsync(dxi, dxo, (size_t)N * 12, hipMemcpyDeviceToDevice, s));
      HIP_TRY(hipMemsetAsync(dW4, 0, (size_t)2 * H * M * H * 4, s));
      for (int q = 0; q < M; ++q) {
        HIP_TRY(hipMemsetAsync(vec_of(k0 + q), 0, (size_t)8 * H * 4, s));
        HIP_TRY(hipMemsetAsync(og[q].d_W2, 0, (size_t)H * H * 4, s));
      }
    }
    cx ^= 1;
    for (int q = 0; q < M; ++q) {
      const int wi = q == 0 ? e : e + 5;
      ud.push_back(TnUnpackDesc{G[wi], ld1, G[wi + 1], emb ? w.demb_part + (size_t)(k0 + q) * 3 * d.enf : nullptr, dW4 + (size_t)q * 2 * H * H, H,
                                vec_of(k0 + q), P[wi], emb, d.enf, H});
      cd.push_back(TnCopyDesc{G[wi + 3], vec_of(k0 + q) + 5 * H, nullptr, H});                     // d b2
    }
    cd.push_back(TnCopyDesc{G[e + 4], vec_of(k0) + 6 * H, M == 2 ? vec_of(k0 + 1) + 6 * H : nullptr, H});   // d w3 (shared head)
    mlp_no += M;
    // ---- sublayers, last first: d_h[cur] = gradient w.r.t. the sublayer's output h
    for (int sl = d.S - 1; sl >= 0; --sl) {
      const int gi = b * d.S + sl, base = ix.sub(b, sl);
      const float* hin = w.h[gi];
      float* dout = w.d_h[cur];
      // node MLP backward
      // (weight gradients on the side stream; the main stream joins it in front of the message stage's W2 gradient
      //  (mlp_backward) -- which is before anything these launches read is overwritten: dout by the message stage's last
      //  tn_lin, dz / xcat by the next sublayer, d_pq / d_pq4 by the node gathers that follow the join)
      { int rc = fork_w(); if (rc) return rc; }
      { int rc = tn_wgrad(sw, dout, H, w.act[gi], H, N, H, H, G[base + 6], w); if (rc) return rc; }
      { int rc = fork_b(); if (rc) return rc; }
      { int rc = tn_colsum(sb, dout, H, N, H, G[base + 7], w); if (rc) return rc; }
      { int rc = tn_lin(s, dout, H, H, nullptr, 0, 0, P[base + 6], H, nullptr, nullptr, 0, w.da, H, N, H); if (rc) return rc; }
      hipLaunchKernelGGL(tn_silu_bwd_kernel, dim3(tn_blocks((size_t)N * H)), dim3(256), 0, s, (const float*)w.da, (const float*)w.z[gi], w.dz, (size_t)N * H);
      HIP_TRY(hipGetLastError());
      hipLaunchKernelGGL(tn_cat_kernel, dim3(tn_blocks((size_t)N * 2 * H)), dim3(256), 0, s, hin, (const float*)w.agg[gi], w.xcat, N, H);
      HIP_TRY(hipGetLastError());
      { int rc = fork_w(); if (rc) return rc; }
      { int rc = tn_wgrad(sw, w.dz, H, w.xcat, 2 * H, N, H, 2 * H, G[base + 4], w); if (rc) return rc; }
      { int rc = fork_b(); if (rc) return rc; }
      { int rc = tn_colsum(sb, w.dz, H, N, H, G[base + 5], w); if (rc) return rc; }
      // d_hin = dz W1[:, :H] + d_out (residual);  d_agg = dz W1[:, H:]
      { int rc = tn_lin(s, w.dz, H, H, nullptr, 0, 0, P[base + 4], 2 * H, nullptr, dout, H, w.d_h[cur ^ 1], H, N, H); if (rc) return rc; }
      { int rc = tn_lin(s, w.dz, H, H, nullptr, 0, 0, P[base + 4] + H, 2 * H, nullptr, nullptr, 0, w.d_agg, H, N, H); if (rc) return rc; }
      cur ^= 1;
      // message stage backward
      const dsbdd_train_mlp m = tn_mlp(w.pq[gi], 0, H, 2 * H, pk.gcl[gi], P[base + 2], P[base + 3],
                                       ix.att ? P[base + 8] : nullptr, ix.att ? P[base + 9] : nullptr);
      dsbdd_train_mlp_grad o{};
      const int k = mlp_no++;
      float* dWpq = w.dWpq + (size_t)gi * 2 * H * H;
      o.dP = w.d_pq; o.dQ = w.d_pq + H; o.ldo = 2 * H; o.d_vec = vec_of(k); o.d_W2 = G[base + 2]; o.gd0 = w.gd0;
      if (want_in) HIP_TRY(hipMemsetAsync(w.gd0, 0, (size_t)(E > 0 ? E : 1) * 4, s));
      { const int rc = gcl_backward_impl(stream, H, g, &m, w.x[b], c.normalization_factor, w.d_agg, &o, w.d_xg, w.scratch, w.scratch_bytes, sd, w.z2[gi]); if (rc) return rc; }
      { const int rc = add_gd0(w.gd0); if (rc) return rc; }
      hipLaunchKernelGGL(tn_add_kernel, dim3(tn_blocks(3 * (size_t)N)), dim3(256), 0, s, w.d_x[cx], (const float*)w.d_xg, 3 * (size_t)N);
      HIP_TRY(hipGetLastError());
      { int rc = tn_lin(s, w.d_pq, 2 * H, 2 * H, nullptr, 0, 0, pk.Wpq[gi], H, nullptr, w.d_h[cur], H, w.d_h[cur ^ 1], H, N, H); if (rc) return rc; }
      cur ^= 1;
      { int rc = fork_w(); if (rc) return rc; }
      { int rc = tn_wgrad(sw, w.d_pq, 2 * H, hin, H, N, 2 * H, H, dWpq, w); if (rc) return rc; }
      ud.push_back(TnUnpackDesc{G[base], ld1, G[base + 1], emb ? w.demb_part + (size_t)k * 3 * d.enf : nullptr, dWpq, H, vec_of(k), P[base], emb,
                                d.enf, H});
      cd.push_back(TnCopyDesc{G[base + 3], vec_of(k) + 5 * H, nullptr, H});                        // d b2
      if (ix.att) {
        cd.push_back(TnCopyDesc{G[base + 8], vec_of(k) + 6 * H, nullptr, H});
        cd.push_back(TnCopyDesc{G[base + 9], vec_of(k) + 7 * H, nullptr, 1});
      }
    }
  }
  if (sd) { HIP_TRY(sd->link(sd->wg, s)); HIP_TRY(sd->link(sd->co, s)); }      // every weight / bias gradient of the blocks is complete from here on
  // embedding
  { int rc = tn_wgrad(s, w.d_h[cur], H, w.h0, JP, N, H, d.D, G[ix.emb], w); if (rc) return rc; }
  { int rc = tn_colsum(s, w.d_h[cur], H, N, H, G[ix.emb + 1], w); if (rc) return rc; }
  { int rc = tn_lin(s, w.d_h[cur], H, H, nullptr, 0, 0, pk.emb.Wp, pk.emb.ldP, nullptr, nullptr, 0, w.d_h0, JP, N, d.D); if (rc) return rc; }
  // encoders
  { int rc = mlp2_bwd(w.d_h0, JP, J, 2 * a, a, w.hf_l, a, w.ze_l, w.ae_l, ix.ae0, pk.ae0, ix.ae2, pk.ae2, w.d_hf_l, a, nl_, want_in); if (rc) return rc; }
  { int rc = mlp2_bwd(w.d_h0 + (size_t)nl_ * JP, JP, J, 2 * r, r, w.hf_p, r, w.ze_p, w.ae_p, ix.re0, pk.re0, ix.re2, pk.re2, w.d_hf_p, r, np_, want_in); if (rc) return rc; }
  // the assembled first layers, the copied vectors, the edge-type embedding
  for (size_t o = 0; o < ud.size(); o += kTnUnpackPerLaunch) {
    TnUnpackTable tab{};
    const int n = (int)std::min<size_t>(kTnUnpackPerLaunch, ud.size() - o);
    for (int i = 0; i < n; ++i) tab.d[i] = ud[o + i];
    hipLaunchKernelGGL(tn_unpack_kernel, dim3(64, (unsigned)n), dim3(256), 0, s, tab);
    HIP_TRY(hipGetLastError());
  }
  for (size_t o = 0; o < cd.size(); o += kTnCopyPerLaunch) {
    TnCopyTable tab{};
    const int n = (int)std::min<size_t>(kTnCopyPerLaunch, cd.size() - o);
    for (int i = 0; i < n; ++i) tab.d[i] = cd[o + i];
    hipLaunchKernelGGL(tn_copy_kernel, dim3(1, (unsigned)n), dim3(256), 0, s, tab);
    HIP_TRY(hipGetLastError());
  }
  if (emb) {
    hipLaunchKernelGGL(tn_sum_parts_kernel, dim3(1), dim3(256), 0, s, (const float*)w.demb_part, mlp_no, 3 * d.enf, 3 * d.enf, G[ix.emb_tab]);
    HIP_TRY(hipGetLastError());
  }
  if (want_in) {
    // x enters through the first block (d_x[cx]), the velocity (- d_vel) and the input distances d0 of every edge MLP
    float* dx = w.d_x[cx];
    if (E > 0) {
      { const int rc = dsbdd_train_radial_backward(stream, g, w.x0, w.gd0_tot, w.d_xg); if (rc) return rc; }
      hipLaunchKernelGGL(tn_add_kernel, dim3(tn_blocks(3 * (size_t)N)), dim3(256), 0, s, dx, (const float*)w.d_xg, 3 * (size_t)N);
      HIP_TRY(hipGetLastError());
    }
    hipLaunchKernelGGL(tn_sub_kernel, dim3(tn_blocks(3 * (size_t)N)), dim3(256), 0, s, dx, (const float*)w.d_vel, 3 * (size_t)N);
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(tn_join_grads_kernel, dim3(tn_blocks(N)), dim3(256), 0, s, (const float*)dx, (const float*)w.d_hf_l, a,
                       (const float*)w.d_hf_p, r, nl_, np_, d_xh_lig, d_xh_pocket);
    HIP_TRY(hipGetLastError());
  }
  (void)J;
  return DSBDD_OK;
}

}  // extern "C"
