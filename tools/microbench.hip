// Standalone timing harness for the hot kernels (no torch: a run costs seconds on the GPU box).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DVARIANT flags] tools/microbench.hip -o tools/bin/mb_<tag>
//   tools/bin/mb_<tag> [B] [reps]
//
// Builds a synthetic batch with the geometry of the benchmark workload (B samples x (23 ligand + 286 pocket)
// nodes, 5 A radius graph on uniformly random points at the density of a protein pocket, ligand-ligand complete,
// (sample, node set) segments 32-aligned with inactive padding entries, rows sorted) and times
//   * edge_wave_kernel<256, MODE_GCL>   on the whole list, on the ligand-endpoint prefix and on a 28 % prefix
//   * edge_wave_kernel<256, MODE_COORD> on the ligand-row prefix (two MLPs, one workgroup per (tile, MLP))
//   * the node GEMMs of one block (layer 1, layer 2, grouped projections) at 19.8 k / 11.5 k / 3.6 k rows
// with HIP events over `reps` back-to-back launches, and prints a checksum of every output so that variants of a
// kernel (compile-time flags) can be compared for equality.  Test / measurement infrastructure, not product code.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../diffsbdd_amd/csrc/common.h"
#include "../diffsbdd_amd/csrc/edge_mlp.h"
#include "edge_wave_diag.h"   // the round-5 state of csrc/edge_wave.h with its A/B and diagnostic build switches
#include "../diffsbdd_amd/csrc/graph.h"
#include "../diffsbdd_amd/csrc/node_linear.h"
#include "../diffsbdd_amd/csrc/node_chain.h"

using namespace dsbdd;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <class T>
static T* dev(const std::vector<T>& v) {
  T* p = nullptr;
  CK(hipMalloc(&p, std::max<size_t>(v.size(), 1) * sizeof(T)));
  if (!v.empty()) CK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return p;
}
template <class T>
static T* dev_zero(size_t n) {
  T* p = nullptr;
  CK(hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(T)));
  CK(hipMemset(p, 0, std::max<size_t>(n, 1) * sizeof(T)));
  return p;
}
static double checksum(const float* d, size_t n) {
  std::vector<float> h(n);
  CK(hipMemcpy(h.data(), d, n * sizeof(float), hipMemcpyDeviceToHost));
  double s = 0.0;
  for (size_t i = 0; i < n; ++i) s += (double)h[i] * (double)((i % 977) + 1);
  return s;
}
static std::vector<float> rnd(std::mt19937& g, size_t n, float scale) {
  std::uniform_real_distribution<float> u(-scale, scale);
  std::vector<float> v(n);
  for (auto& x : v) x = u(g);
  return v;
}

template <class F>
static float time_us(F&& launch, int reps) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) launch();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(a, 0));
  for (int i = 0; i < reps; ++i) launch();
  CK(hipEventRecord(b, 0));
  CK(hipEventSynchronize(b));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, a, b));
  CK(hipGetLastError());
  return ms * 1e3f / reps;
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 64;
  const int reps = argc > 2 ? atoi(argv[2]) : 20;
  constexpr int H = 256;
  const int nl = 23, np = 286;
  const int n_lig = B * nl, n_poc = B * np, N = n_lig + n_poc;
  std::mt19937 g(1234);

  // ---- geometry: ligand in the middle of a box of pocket atoms -------------------------------------------
  std::vector<float> x(3 * (size_t)N);
  std::uniform_real_distribution<float> box(0.f, 17.5f), mid(6.5f, 11.0f);
  for (int b = 0; b < B; ++b) {
    for (int i = 0; i < nl; ++i) for (int k = 0; k < 3; ++k) x[3 * (size_t)(b * nl + i) + k] = mid(g) + 40.f * b;
    for (int i = 0; i < np; ++i) for (int k = 0; k < 3; ++k) x[3 * (size_t)(n_lig + b * np + i) + k] = box(g) + 40.f * b;
  }
  auto d2 = [&](int i, int j) {
    float s = 0.f;
    for (int k = 0; k < 3; ++k) { const float d = x[3 * (size_t)i + k] - x[3 * (size_t)j + k]; s += d * d; }
    return s;
  };
  std::vector<int> erow, ecol, node_batch(N);
  std::vector<float> ed0;
  std::vector<int> row_ptr(N + 1, 0), deg(N, 0);
  auto pad = [&]() { while (erow.size() % 32) { erow.push_back(-1); ecol.push_back(0); ed0.push_back(0.f); } };
  auto add_rows = [&](int first, int count, int b, bool lig_rows) {
    for (int i = first; i < first + count; ++i) {
      node_batch[i] = b;
      row_ptr[i] = (int)erow.size();
      for (int j = b * nl; j < (b + 1) * nl; ++j)
        if (lig_rows || d2(i, j) <= 25.f) { erow.push_back(i); ecol.push_back(j); ed0.push_back(d2(i, j)); }
      for (int j = n_lig + b * np; j < n_lig + (b + 1) * np; ++j)
        if (d2(i, j) <= 25.f) { erow.push_back(i); ecol.push_back(j); ed0.push_back(d2(i, j)); }
      deg[i] = (int)erow.size() - row_ptr[i];
    }
    pad();
  };
  for (int b = 0; b < B; ++b) add_rows(b * nl, nl, b, true);
  const int E_lig = (int)erow.size();                    // ligand-row prefix (update_coords_mask)
  for (int b = 0; b < B; ++b) add_rows(n_lig + b * np, np, b, false);
  const int E = (int)erow.size();
  row_ptr[N] = E;
  long real = 0;
  for (int r : erow) real += r >= 0;
  printf("# B=%d N=%d E=%d slots (%ld edges, %.1f per node), ligand-row prefix %d slots, tiles(128)=%d\n", B, N, E, real,
         (double)real / N, E_lig, (E + 127) / 128);

  int *d_erow = dev(erow), *d_ecol = dev(ecol), *d_nb = dev(node_batch);
  float* d_ed0 = dev(ed0);
  float* d_x = dev(x);
  std::vector<int> counts = {E, E_lig, (int)(0.28 * E) / 32 * 32, (int)(0.69 * E) / 32 * 32};
  int* d_counts = dev(counts);
  int* d_tile_ctr = dev_zero<int>(kTileCtrInts);
  int* d_rowptr = dev(row_ptr);
  int* d_deg = dev(deg);

  // ---- weights / projections ------------------------------------------------------------------------------
  const float ws = 1.f / 16.f;
  float* d_pq = dev(rnd(g, (size_t)N * 4 * H, 1.0f));     // up to 4H columns (coordinate stage layout)
  auto mk = [&](size_t n, float s) { return dev(rnd(g, n, s)); };
  struct Mlp { float *wd, *wd0, *tab, *w2t, *b2, *w2tp; } m[2];
  for (int q = 0; q < 2; ++q) {
    m[q] = {mk(H, 0.05f), mk(H, 0.05f), mk(3 * H, 0.3f), mk((size_t)H * H, ws), mk(H, 0.1f), dev_zero<float>((size_t)H * H)};
    hipLaunchKernelGGL(permute_w2t_kernel, dim3((H * H + 255) / 256), dim3(256), 0, 0, (const float*)m[q].w2t, m[q].w2tp, H);
  }
  float *d_attw = mk(H, ws), *d_attb = mk(1, 0.1f), *d_w3 = mk(H, ws);
  float* d_agg = dev_zero<float>((size_t)N * H);
  float* d_head = dev_zero<float>((size_t)(E / 32 + 2) * H);
  float* d_xagg = dev_zero<float>((size_t)2 * N * 3);
  float* d_xhead = dev_zero<float>((size_t)2 * (E / 32 + 2) * 4);
  float* d_mean = dev(std::vector<float>(3 * (size_t)B, 8.75f));
  CK(hipDeviceSynchronize());

  auto edge_args = [&](int mode, int count_idx) {
    EdgeArgs a{};
    a.erow = d_erow; a.ecol = d_ecol; a.ed0 = d_ed0; a.e_count = d_counts + count_idx; a.e_cap = E;
    a.x = d_x; a.n_lig = n_lig; a.n_nodes = N; a.tile_ctr = d_tile_ctr; a.norm_factor = 100.f; a.wt_base = 0;
    if (mode == MODE_GCL) {
      a.ldpq = 2 * H;
      a.mlp[0] = EdgeMlpW{d_pq, d_pq + H, m[0].wd, m[0].wd0, m[0].tab, m[0].w2t, m[0].b2, m[0].w2tp};
      a.mlp[1] = a.mlp[0];
      a.att_w = d_attw; a.att_b = d_attb; a.attention = 1; a.agg = d_agg; a.agg_head = d_head;
    } else {
      a.ldpq = 4 * H;
      a.mlp[0] = EdgeMlpW{d_pq + 2 * H, d_pq, m[0].wd, m[0].wd0, m[0].tab, m[0].w2t, m[0].b2, m[0].w2tp};
      a.mlp[1] = EdgeMlpW{d_pq + 3 * H, d_pq + H, m[1].wd, m[1].wd0, m[1].tab, m[1].w2t, m[1].b2, m[1].w2tp};
      a.w3 = d_w3; a.node_batch = d_nb; a.mean = d_mean; a.norm_constant = 1.f; a.coords_range = 15.f;
      a.use_tanh = 1; a.n_mlp = 2; a.xagg = d_xagg; a.xagg_head = d_xhead; a.xagg_stride = (size_t)N * 3;
      a.xhead_stride = (size_t)(E / 32 + 2) * 4; a.pass_split = 1;
    }
    return a;
  };
  int n_cu = 256;
  { hipDeviceProp_t p; if (hipGetDeviceProperties(&p, 0) == hipSuccess && p.multiProcessorCount > 0) n_cu = p.multiProcessorCount; }
  auto grid_of = [&](int mode, int edges) {
    long tiles = (edges + 127) / 128, gmax = 2L * n_cu;
    long gg = mode == MODE_COORD ? 2 * tiles : tiles;
    if (gg > gmax) gg = gmax;
    const int q8 = mode == MODE_COORD ? 16 : 8;
    return (int)std::max<long>((gg + q8 - 1) / q8 * q8, q8);
  };
  printf("| kernel | edges / rows | us / launch | TFLOP/s | frac of 157.3 | checksum |\n|---|---|---|---|---|---|\n");
#ifndef MB_SKIP_EDGE
  const char* names[4] = {"GCL whole list", "GCL ligand-row prefix", "GCL 28 % prefix", "GCL 69 % prefix"};
  for (int ci : {0, 3, 2, 1}) {
    EdgeArgs a = edge_args(MODE_GCL, ci);
    const int grid = grid_of(MODE_GCL, counts[ci]);
    CK(hipMemset(d_agg, 0, (size_t)N * H * 4)); CK(hipMemset(d_head, 0, (size_t)(E / 32 + 2) * H * 4));
    const float us = time_us([&] { hipLaunchKernelGGL((edge_wave_kernel<H, MODE_GCL, true>), dim3(grid), dim3(kThreads), 0, 0, a); }, reps);
    const double fl = 2.0 * counts[ci] * ((double)H * H + 4.0 * H);
    printf("| %s (grid %d) | %d | %.1f | %.1f | %.3f | %.6e |\n", names[ci], grid, counts[ci], us, fl / us / 1e6, fl / us / 1e6 / 157.3,
           checksum(d_agg, (size_t)N * H) + checksum(d_head, (size_t)(counts[ci] / 32) * H));
  }
  {
    EdgeArgs a = edge_args(MODE_COORD, 1);
    const int grid = grid_of(MODE_COORD, E_lig);
    const float us = time_us([&] { hipLaunchKernelGGL((edge_wave_kernel<H, MODE_COORD, true>), dim3(grid), dim3(kThreads), 0, 0, a); }, reps);
    const double fl = 2.0 * 2.0 * E_lig * ((double)H * H + 3.0 * H);
    printf("| COORD ligand-row prefix, 2 MLPs (grid %d) | %d | %.1f | %.1f | %.3f | %.6e |\n", grid, E_lig, us, fl / us / 1e6,
           fl / us / 1e6 / 157.3, checksum(d_xagg, (size_t)2 * N * 3) + checksum(d_xhead, (size_t)2 * (E / 32 + 2) * 4));
  }
#endif
#ifndef MB_SKIP_NODE
  // ---- node GEMMs of one block ----------------------------------------------------------------------------
  float* d_h = dev(rnd(g, (size_t)N * H, 1.0f));
  float* d_aggc = dev(rnd(g, (size_t)N * H, 1.0f));
  float* d_t1 = dev_zero<float>((size_t)N * H);
  float* d_hn = dev_zero<float>((size_t)N * H);
  float* d_pqg = dev_zero<float>((size_t)N * 2 * H);
  float* d_pqc = dev_zero<float>((size_t)N * 4 * H);
  float *W1 = mk((size_t)2 * H * H, ws), *b1 = mk(H, 0.1f), *W2 = mk((size_t)H * H, ws * 0.02f), *b2n = mk(H, 0.002f);
  float *Wpq = mk((size_t)H * 2 * H, ws), *Wc = mk((size_t)H * 4 * H, ws);
  std::vector<int> perm(N);
  for (int i = 0; i < N; ++i) perm[i] = i;
  std::shuffle(perm.begin() + n_lig, perm.end(), g);      // gathered row lists: ligand rows first, then pocket rows in level order
  int* d_rows = dev(perm);
  std::vector<int> mcounts = {N, 11545, 3639, n_lig, 16};
  int* d_mcounts = dev(mcounts);
  // packed weights of the row-owning chain kernel (node_chain.h)
  auto pack = [&](const float* WT, int ldw, int K, int Ncols) {
    float* d = dev_zero<float>((size_t)K * Ncols + 1024);
    hipLaunchKernelGGL(pack_b16_kernel, dim3((K * Ncols + 255) / 256), dim3(256), 0, 0, WT, ldw, K, Ncols, d);
    return d;
  };
  float *W1p = pack(W1, H, 2 * H, H), *W2p = pack(W2, H, H, H), *Wpqp = pack(Wpq, 2 * H, H, 2 * H), *Wcp = pack(Wc, 4 * H, H, 4 * H);
  float* d_h0 = dev(rnd(g, (size_t)N * H, 1.0f));
  float* d_hc = dev_zero<float>((size_t)N * H);
  float* d_pqg2 = dev_zero<float>((size_t)N * 2 * H);
  float* d_pqc2 = dev_zero<float>((size_t)N * 4 * H);
  auto maxdiff = [&](const float* a, const float* b, size_t n) {
    std::vector<float> ha(n), hb(n);
    CK(hipMemcpy(ha.data(), a, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), b, n * 4, hipMemcpyDeviceToHost));
    double m = 0, ref = 0;
    for (size_t k = 0; k < n; ++k) { m = std::max(m, (double)std::fabs(ha[k] - hb[k])); ref = std::max(ref, (double)std::fabs(ha[k])); }
    if (!(m == m)) m = 1e30;
    return m / std::max(ref, 1e-30);
  };
  for (int mi = 0; mi < 3; ++mi) {
    const int M = mcounts[mi];
    const int* ridx = mi == 0 ? nullptr : d_rows;
    const int* mc = mi == 0 ? nullptr : d_mcounts + mi;
    NodeLinearArgs n1{d_hn, H, H, d_aggc, H, H, W1, H, b1, nullptr, 0, d_t1, H, N, H, 1, ridx, mc};
    NodeLinearArgs n2{d_t1, H, H, nullptr, 0, 0, W2, H, b2n, d_hn, H, d_hn, H, N, H, 0, ridx, mc};
    NodeLinearArgs grp[3] = {
        {d_hn, H, H, nullptr, 0, 0, Wc, 4 * H, nullptr, nullptr, 0, d_pqc, 4 * H, N, 2 * H, 0, d_rows, d_mcounts + 2},
        {d_hn, H, H, nullptr, 0, 0, Wc + 2 * H, 4 * H, nullptr, nullptr, 0, d_pqc + 2 * H, 4 * H, n_lig, 2 * H, 0, nullptr, nullptr},
        {d_hn, H, H, nullptr, 0, 0, Wpq, 2 * H, nullptr, nullptr, 0, d_pqg, 2 * H, N, 2 * H, 0, ridx, mc}};
    NodeChainArgs ca{};
    ca.row_idx = d_rows; ca.m_count = d_mcounts + mi; ca.M = N; ca.do_mlp = 1; ca.h = d_hc; ca.agg = d_aggc;
    ca.W1p = W1p; ca.b1 = b1; ca.W2p = W2p; ca.b2 = b2n; ca.n_proj = 3;
    ca.proj[0] = ChainProj{Wcp, d_pqc2, 4 * H, 2 * H, d_mcounts + 2, 0};                                          // Qc|Qx: active rows
    ca.proj[1] = ChainProj{Wcp + (size_t)(2 * H / 16) * (H / 16) * 256, d_pqc2 + 2 * H, 4 * H, 2 * H, d_mcounts + 3, 0};   // Pc|Px: ligand rows
    ca.proj[2] = ChainProj{Wpqp, d_pqg2, 2 * H, 2 * H, nullptr, 0};                                                // next P|Q: all rows
    // ---- one clean run of either path from the same input: results must agree (fp32 summation order differs) ----
    CK(hipMemcpy(d_hn, d_h0, (size_t)N * H * 4, hipMemcpyDeviceToDevice)); CK(hipMemcpy(d_hc, d_h0, (size_t)N * H * 4, hipMemcpyDeviceToDevice));
    CK(hipMemset(d_pqg, 0, (size_t)N * 2 * H * 4)); CK(hipMemset(d_pqc, 0, (size_t)N * 4 * H * 4));
    CK(hipMemset(d_pqg2, 0, (size_t)N * 2 * H * 4)); CK(hipMemset(d_pqc2, 0, (size_t)N * 4 * H * 4));
    (void)launch_node_linear(0, n1); (void)launch_node_linear(0, n2); (void)launch_node_group(0, grp, 3);
    CK(launch_node_chain(0, ca, H, n_cu));
    CK(hipDeviceSynchronize());
    printf("| chain vs three launches, max |diff| / max |ref| (M = %d): h %.2e, next P|Q %.2e, coordinate projections %.2e | | | | | |\n", M,
           maxdiff(d_hn, d_hc, (size_t)N * H), maxdiff(d_pqg, d_pqg2, (size_t)N * 2 * H), maxdiff(d_pqc, d_pqc2, (size_t)N * 4 * H));
    const float u1 = time_us([&] { (void)launch_node_linear(0, n1); }, reps);
    const float u2 = time_us([&] { (void)launch_node_linear(0, n2); }, reps);
    const float u3 = time_us([&] { (void)launch_node_group(0, grp, 3); }, reps);
    const float uc = time_us([&] { (void)launch_node_chain(0, ca, H, n_cu); }, reps);
    NodeChainArgs cm = ca; cm.n_proj = 0;
    const float um = time_us([&] { (void)launch_node_chain(0, cm, H, n_cu); }, reps);
    NodeChainArgs cp = ca; cp.do_mlp = 0;
    const float up = time_us([&] { (void)launch_node_chain(0, cp, H, n_cu); }, reps);
    const double f1 = 2.0 * M * 2.0 * H * H, f2 = 2.0 * M * (double)H * H,
                 f3 = 2.0 * ((double)M * H * 2 * H + 3639.0 * H * 2 * H + (double)n_lig * H * 2 * H);
    printf("| node MLP layer 1 (K=512,N=256) | %d | %.1f | %.1f | %.3f | |\n", M, u1, f1 / u1 / 1e6, f1 / u1 / 1e6 / 157.3);
    printf("| node MLP layer 2 (K=256,N=256,+res) | %d | %.1f | %.1f | %.3f | |\n", M, u2, f2 / u2 / 1e6, f2 / u2 / 1e6 / 157.3);
    printf("| grouped projections (Qc|Qx act, Pc|Px lig, next P|Q) | %d | %.1f | %.1f | %.3f | |\n", M, u3, f3 / u3 / 1e6, f3 / u3 / 1e6 / 157.3);
    printf("| = node phase of one block, three launches | %d | %.1f | %.1f | %.3f | |\n", M, u1 + u2 + u3, (f1 + f2 + f3) / (u1 + u2 + u3) / 1e6,
           (f1 + f2 + f3) / (u1 + u2 + u3) / 1e6 / 157.3);
    printf("| **node_chain: MLP + 3 projections, one launch** | %d | %.1f | %.1f | %.3f | |\n", M, uc, (f1 + f2 + f3) / uc / 1e6, (f1 + f2 + f3) / uc / 1e6 / 157.3);
    printf("| node_chain: MLP only | %d | %.1f | %.1f | %.3f | |\n", M, um, (f1 + f2) / um / 1e6, (f1 + f2) / um / 1e6 / 157.3);
    printf("| node_chain: projections only (h from global) | %d | %.1f | %.1f | %.3f | |\n", M, up, f3 / up / 1e6, f3 / up / 1e6 / 157.3);
#ifdef DSBDD_CHAIN_TS
    {   // in-kernel timeline of wave 0 of every workgroup (shader clock), one launch
      unsigned long long* d_ts = dev_zero<unsigned long long>((size_t)n_cu * 32);
      CK(hipMemcpyToSymbol(HIP_SYMBOL(g_chain_ts), &d_ts, sizeof(d_ts)));
      CK(launch_node_chain(0, ca, H, n_cu));
      CK(hipDeviceSynchronize());
      std::vector<unsigned long long> ts((size_t)n_cu * 32);
      CK(hipMemcpy(ts.data(), d_ts, ts.size() * 8, hipMemcpyDeviceToHost));
      unsigned long long* nul = nullptr;
      CK(hipMemcpyToSymbol(HIP_SYMBOL(g_chain_ts), &nul, sizeof(nul)));
      const char* nm[8] = {"entry", "split + panel DMA issue", "stage-1 K loop (chunked, 8 barriers)", "epilogue 1 (bias, SiLU -> LDS) + barrier",
                           "stage-2 K loop", "epilogue 2 (residual, store, h -> LDS, 2 barriers)", "projections (all passes)", "final barrier"};
      double sum[8] = {0}; int cnt = 0; double tot = 0;
      for (int b = 0; b < n_cu; ++b) {
        const unsigned long long* t = ts.data() + (size_t)b * 32;
        if (!t[0] || !t[7]) continue;
        for (int k = 1; k < 8; ++k) sum[k] += (double)(t[k] - t[k - 1]);
        tot += (double)(t[7] - t[0]); ++cnt;
      }
      printf("| node_chain timeline (M = %d, wave 0, mean over %d workgroups, shader cycles; LAST range of each workgroup) | | | | | |\n", M, cnt);
      for (int k = 1; k < 8; ++k) printf("| ... %s | | %.0f cycles | %.1f %% | | |\n", nm[k], sum[k] / cnt, 100.0 * sum[k] / tot);
      double wall = 0;
      for (int b = 0; b < n_cu; ++b) { const unsigned long long* t = ts.data() + (size_t)b * 32; if (t[0] && t[7]) wall += (double)(t[16 + 7] - t[16]); }
      {
        unsigned long long t0 = ~0ull, t1 = 0; double smin = 1e30, smax = 0, lat_max = 0; int bmax = -1, bmin = -1;
        for (int b = 0; b < n_cu; ++b) { const unsigned long long* t = ts.data() + (size_t)b * 32; if (t[0] && t[7]) { t0 = std::min(t0, t[16]); t1 = std::max(t1, t[16 + 7]); } }
        for (int b = 0; b < n_cu; ++b) {
          const unsigned long long* t = ts.data() + (size_t)b * 32;
          if (!t[0] || !t[7]) continue;
          const double sp = (double)(t[16 + 7] - t[16]) / 100.0;
          if (sp > smax) { smax = sp; bmax = b; }
          if (sp < smin) { smin = sp; bmin = b; }
          lat_max = std::max(lat_max, (double)(t[16] - t0) / 100.0);
        }
        printf("| ... first entry -> last end %.1f us; workgroup span min %.1f (wg %d) / max %.1f us (wg %d); latest entry +%.1f us | | | | | |\n",
               (double)(t1 - t0) / 100.0, smin, bmin, smax, bmax, lat_max);
        for (int b : {bmin, bmax, 0, 20, 40, 60, 80, 120, 200, 255}) {
          const unsigned long long* t = ts.data() + (size_t)b * 32;
          printf("| ... wg %d r0 %llu rows %llu passes %llu phases (us):", b, t[8], t[9], t[10]);
          for (int k = 1; k < 8; ++k) printf(" %.1f", (double)(t[16 + k] - t[16 + k - 1]) / 100.0);
          printf("; entry -> counts loaded %.1f us, -> range known %.1f us, -> chain_range %.1f us | | | | | |\n", (double)(t[11] - t[16]) / 100.0,
                 (double)(t[12] - t[16]) / 100.0, (double)(t[17] - t[16]) / 100.0);
        }
      }
      printf("| ... entry -> end | | %.0f cycles = %.1f us by the 100 MHz wall clock -> shader clock %.2f GHz | | | |\n", tot / cnt,
             wall / cnt / 100.0, (tot / cnt) / (wall / cnt * 10.0));
    }
#endif
    if (mi == 0) {   // the launch's fixed cost: 16 rows (one row tile on one workgroup, every other workgroup only computes the split)
      NodeChainArgs ct = cm; ct.m_count = d_mcounts + 4;
      const float ut = time_us([&] { (void)launch_node_chain(0, ct, H, n_cu); }, reps);
      NodeChainArgs c3 = ca; c3.m_count = d_mcounts + 4;
      const float ut3 = time_us([&] { (void)launch_node_chain(0, c3, H, n_cu); }, reps);
      printf("| node_chain: 16 rows (fixed cost), MLP only / MLP + 3 projections | 16 | %.1f / %.1f | | | |\n", ut, ut3);
    }
  }
#endif
  return 0;
}
