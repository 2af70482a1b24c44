// Kernel-level comparison of the default (edge_wave.h: a wave owns 32 edges x all H columns) and the split-K
// (edge_splitk.h: a workgroup owns 32 edges, wave w a quarter of the reduction dimension) fused edge kernels on
// prefixes of one synthetic edge list (the geometry of tools/microbench.hip: B samples x (23 ligand + 286 pocket) nodes,
// or the C-alpha geometry with `ca`): time per launch at several edge counts, and the completed aggregates of the two
// variants against each other (they differ in rounding only).  Test / measurement infrastructure, not product code.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench_sk.hip -o tools/bin/mbsk && tools/bin/mbsk [B] [reps] [ca]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "../diffsbdd_amd/csrc/common.h"
#include "../diffsbdd_amd/csrc/edge_mlp.h"
#include "../diffsbdd_amd/csrc/edge_wave.h"
#include "../diffsbdd_amd/csrc/edge_wave16.h"
#include "../diffsbdd_amd/csrc/edge_splitk.h"
#include "../diffsbdd_amd/csrc/graph.h"

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
static double maxdiff(const float* a, const float* b, size_t n) {
  std::vector<float> ha(n), hb(n);
  CK(hipMemcpy(ha.data(), a, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hb.data(), b, n * 4, hipMemcpyDeviceToHost));
  double m = 0, ref = 0;
  for (size_t k = 0; k < n; ++k) { m = std::max(m, (double)std::fabs(ha[k] - hb[k])); ref = std::max(ref, (double)std::fabs(ha[k])); }
  if (!(m == m)) m = 1e30;
  return m / std::max(ref, 1e-30);
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 64;
  const int reps = argc > 2 ? atoi(argv[2]) : 20;
  const bool ca = argc > 3 && !strcmp(argv[3], "ca");
  constexpr int H = 256;
  const int nl = 23, np = ca ? 36 : 286;
  const float boxw = ca ? 24.f : 17.5f, cut2 = ca ? 100.f : 25.f;      // C-alpha: ~ the density of the 3rfm CA pocket in model units x 2
  const int n_lig = B * nl, n_poc = B * np, N = n_lig + n_poc;
  std::mt19937 g(1234);
  std::vector<float> x(3 * (size_t)N);
  std::uniform_real_distribution<float> box(0.f, boxw), mid(0.37f * boxw, 0.63f * boxw);
  for (int b = 0; b < B; ++b) {
    for (int i = 0; i < nl; ++i) for (int k = 0; k < 3; ++k) x[3 * (size_t)(b * nl + i) + k] = mid(g) + 60.f * b;
    for (int i = 0; i < np; ++i) for (int k = 0; k < 3; ++k) x[3 * (size_t)(n_lig + b * np + i) + k] = box(g) + 60.f * b;
  }
  auto d2 = [&](int i, int j) {
    float s = 0.f;
    for (int k = 0; k < 3; ++k) { const float d = x[3 * (size_t)i + k] - x[3 * (size_t)j + k]; s += d * d; }
    return s;
  };
  std::vector<int> erow, ecol, node_batch(N), row_ptr(N + 1, 0), deg(N, 0);
  std::vector<float> ed0;
  auto pad = [&]() { while (erow.size() % 32) { erow.push_back(-1); ecol.push_back(0); ed0.push_back(0.f); } };
  auto add_rows = [&](int first, int count, int b, bool lig_rows) {
    for (int i = first; i < first + count; ++i) {
      node_batch[i] = b;
      row_ptr[i] = (int)erow.size();
      for (int j = b * nl; j < (b + 1) * nl; ++j)
        if (lig_rows || d2(i, j) <= cut2) { erow.push_back(i); ecol.push_back(j); ed0.push_back(d2(i, j)); }
      for (int j = n_lig + b * np; j < n_lig + (b + 1) * np; ++j)
        if (d2(i, j) <= cut2) { erow.push_back(i); ecol.push_back(j); ed0.push_back(d2(i, j)); }
      deg[i] = (int)erow.size() - row_ptr[i];
    }
    pad();
  };
  for (int b = 0; b < B; ++b) add_rows(b * nl, nl, b, true);
  const int E_lig = (int)erow.size();
  for (int b = 0; b < B; ++b) add_rows(n_lig + b * np, np, b, false);
  const int E = (int)erow.size();
  row_ptr[N] = E;
  printf("# B=%d N=%d E=%d slots, ligand-row prefix %d slots (%s geometry)\n", B, N, E, E_lig, ca ? "C-alpha" : "full-atom");

  int *d_erow = dev(erow), *d_ecol = dev(ecol), *d_nb = dev(node_batch), *d_rowptr = dev(row_ptr), *d_deg = dev(deg);
  float *d_ed0 = dev(ed0), *d_x = dev(x);
  std::vector<int> counts = {E, E_lig, (int)(0.28 * E) / 32 * 32, (int)(0.69 * E) / 32 * 32, (int)(0.12 * E) / 32 * 32, (int)(0.5 * E) / 32 * 32};
  int* d_counts = dev(counts);
  int* d_tile_ctr = dev_zero<int>(kTileCtrInts);
  const float ws = 1.f / 16.f;
  float* d_pq = dev(rnd(g, (size_t)N * 4 * H, 1.0f));
  auto mk = [&](size_t n, float s) { return dev(rnd(g, n, s)); };
  struct Mlp { float *wd, *wd0, *tab, *w2t, *b2, *w2tp, *w2tp16, *w2sk; } m[2];
  for (int q = 0; q < 2; ++q) {
    m[q] = {mk(H, 0.05f), mk(H, 0.05f), mk(3 * H, 0.3f), mk((size_t)H * H, ws), mk(H, 0.1f), dev_zero<float>((size_t)H * H),
            dev_zero<float>((size_t)H * H), dev_zero<float>((size_t)H * H)};
    hipLaunchKernelGGL(pack_w2sk_kernel, dim3((H * H + 255) / 256), dim3(256), 0, 0, (const float*)m[q].w2t, m[q].w2sk, H);
    hipLaunchKernelGGL(permute_w2t_kernel, dim3((H * H + 255) / 256), dim3(256), 0, 0, (const float*)m[q].w2t, m[q].w2tp, H);
    hipLaunchKernelGGL(pack16_w2t_kernel, dim3((H * H + 255) / 256), dim3(256), 0, 0, (const float*)m[q].w2t, m[q].w2tp16, H);
  }
  float *d_attw = mk(H, ws), *d_attb = mk(1, 0.1f), *d_w3 = mk(H, ws);
  const size_t slots16 = (size_t)(E / 16 + 4);
  float *d_agg[2] = {dev_zero<float>((size_t)N * H), dev_zero<float>((size_t)N * H)};
  float* d_head = dev_zero<float>(slots16 * H);
  float* d_xagg = dev_zero<float>((size_t)2 * N * 3);
  float* d_xhead = dev_zero<float>(2 * slots16 * 4);
  float* d_xo[2] = {dev_zero<float>((size_t)N * 3), dev_zero<float>((size_t)N * 3)};
  float* d_mean = dev(std::vector<float>(3 * (size_t)B, 0.5f * boxw));
  CK(hipDeviceSynchronize());
  int n_cu = 256;
  { hipDeviceProp_t p; if (hipGetDeviceProperties(&p, 0) == hipSuccess && p.multiProcessorCount > 0) n_cu = p.multiProcessorCount; }

  auto edge_args = [&](int mode, int ci, float* agg) {
    EdgeArgs a{};
    a.erow = d_erow; a.ecol = d_ecol; a.ed0 = d_ed0; a.e_count = d_counts + ci; a.e_cap = E;
    a.x = d_x; a.n_lig = n_lig; a.n_nodes = N; a.tile_ctr = d_tile_ctr; a.norm_factor = 100.f; a.wt_base = 0;
    if (mode == MODE_GCL) {
      a.ldpq = 2 * H;
      a.mlp[0] = EdgeMlpW{d_pq, d_pq + H, m[0].wd, m[0].wd0, m[0].tab, m[0].w2t, m[0].b2, m[0].w2tp, m[0].w2tp16, nullptr, m[0].w2sk};
      a.mlp[1] = a.mlp[0];
      a.att_w = d_attw; a.att_b = d_attb; a.attention = 1; a.agg = agg; a.agg_head = d_head;
    } else {
      a.ldpq = 4 * H;
      a.mlp[0] = EdgeMlpW{d_pq + 2 * H, d_pq, m[0].wd, m[0].wd0, m[0].tab, m[0].w2t, m[0].b2, m[0].w2tp, m[0].w2tp16, nullptr, m[0].w2sk};
      a.mlp[1] = EdgeMlpW{d_pq + 3 * H, d_pq + H, m[1].wd, m[1].wd0, m[1].tab, m[1].w2t, m[1].b2, m[1].w2tp, m[1].w2tp16, nullptr, m[1].w2sk};
      a.w3 = d_w3; a.node_batch = d_nb; a.mean = d_mean; a.norm_constant = 1.f; a.coords_range = 15.f;
      a.use_tanh = 1; a.n_mlp = 2; a.xagg = d_xagg; a.xagg_head = d_xhead; a.xagg_stride = (size_t)N * 3;
      a.xhead_stride = slots16 * 4; a.pass_split = 1;
    }
    return a;
  };
  auto grid32 = [&](int mode, int edges) {
    long tiles = (edges + 127) / 128, gmax = 2L * n_cu;
    long gg = mode == MODE_COORD ? 2 * tiles : tiles;
    if (gg > gmax) gg = gmax;
    const int q8 = mode == MODE_COORD ? 16 : 8;
    return (int)std::max<long>((gg + q8 - 1) / q8 * q8, q8);
  };
  auto grid16 = [&](int mode, int edges, int per_cu) {       // split-K: one item per (32-edge tile, MLP)
    long items = (long)((edges + 31) / 32) * (mode == MODE_COORD ? 2 : 1), gmax = (long)per_cu * n_cu;
    long gg = std::min(items, gmax);
    const int q8 = mode == MODE_COORD ? 16 : 8;
    return (int)std::max<long>((gg + q8 - 1) / q8 * q8, q8);
  };
  // rows whose edges lie in the prefix `cnt`: the completion kernels only look at rows [0, n_rows)
  auto rows_in = [&](int cnt) { int r = 0; while (r < N && row_ptr[r] + deg[r] <= cnt) ++r; return r; };

  printf("| stage | edges | edge_wave us (grid) | frac | split-K us (grid 2/CU) | frac | split-K us (grid 1/CU) | max rel diff split-K vs edge_wave |\n|---|---|---|---|---|---|---|---|\n");
  for (int ci : {0, 3, 5, 2, 4, 1}) {
    const int cnt = counts[ci], nr = rows_in(cnt);
    const double fl = 2.0 * cnt * ((double)H * H + 4.0 * H);
    EdgeArgs a32 = edge_args(MODE_GCL, ci, d_agg[0]), a16 = edge_args(MODE_GCL, ci, d_agg[1]);
    const int g32 = grid32(MODE_GCL, cnt), g16 = grid16(MODE_GCL, cnt, 2), g16b = grid16(MODE_GCL, cnt, 1);
    const float u32 = time_us([&] { hipLaunchKernelGGL((edge_wave_kernel<H, MODE_GCL, true>), dim3(g32), dim3(kThreads), 0, 0, a32); }, reps);
    hipLaunchKernelGGL(agg_complete_kernel, dim3((nr + 3) / 4), dim3(kThreads), 0, 0, d_agg[0], (const float*)d_head, (const int*)d_rowptr,
                       (const int*)d_deg, nr, H, (int)slots16 - 1, 5);
    CK(hipDeviceSynchronize());
    const float u16 = time_us([&] { hipLaunchKernelGGL((edge_splitk_kernel<H, MODE_GCL>), dim3(g16), dim3(kThreads), 0, 0, a16); }, reps);
    const float u16b = time_us([&] { hipLaunchKernelGGL((edge_splitk_kernel<H, MODE_GCL>), dim3(g16b), dim3(kThreads), 0, 0, a16); }, reps);
    hipLaunchKernelGGL(agg_complete_kernel, dim3((nr + 3) / 4), dim3(kThreads), 0, 0, d_agg[1], (const float*)d_head, (const int*)d_rowptr,
                       (const int*)d_deg, nr, H, (int)slots16 - 1, 5);
    CK(hipDeviceSynchronize());
    printf("| GCL | %d | %.1f (%d) | %.3f | %.1f (%d) | %.3f | %.1f | %.2e |\n", cnt, u32, g32, fl / u32 / 1e6 / 157.3, u16, g16,
           fl / u16 / 1e6 / 157.3, u16b, maxdiff(d_agg[0], d_agg[1], (size_t)nr * H));
  }
  for (int ci : {1}) {
    const int cnt = counts[ci], nr = rows_in(cnt);
    const double fl = 2.0 * 2.0 * cnt * ((double)H * H + 3.0 * H);
    EdgeArgs a32 = edge_args(MODE_COORD, ci, nullptr), a16 = a32;
    const int g32 = grid32(MODE_COORD, cnt), g16 = grid16(MODE_COORD, cnt, 2), g16b = grid16(MODE_COORD, cnt, 1);
    const float u32 = time_us([&] { hipLaunchKernelGGL((edge_wave_kernel<H, MODE_COORD, true>), dim3(g32), dim3(kThreads), 0, 0, a32); }, reps);
    CK(hipMemset(d_xo[0], 0, (size_t)N * 12));
    hipLaunchKernelGGL(coord_update_kernel, dim3((3 * nr + 255) / 256), dim3(256), 0, 0, d_xo[0], (const float*)d_xagg, (const float*)d_xhead, 2,
                       a32.xagg_stride, a32.xhead_stride, (const int*)d_rowptr, (const int*)d_deg, 3 * nr, (int)slots16 - 1, 5);
    CK(hipDeviceSynchronize());
    const float u16 = time_us([&] { hipLaunchKernelGGL((edge_splitk_kernel<H, MODE_COORD>), dim3(g16), dim3(kThreads), 0, 0, a16); }, reps);
    const float u16b = time_us([&] { hipLaunchKernelGGL((edge_splitk_kernel<H, MODE_COORD>), dim3(g16b), dim3(kThreads), 0, 0, a16); }, reps);
    CK(hipMemset(d_xo[1], 0, (size_t)N * 12));
    hipLaunchKernelGGL(coord_update_kernel, dim3((3 * nr + 255) / 256), dim3(256), 0, 0, d_xo[1], (const float*)d_xagg, (const float*)d_xhead, 2,
                       a16.xagg_stride, a16.xhead_stride, (const int*)d_rowptr, (const int*)d_deg, 3 * nr, (int)slots16 - 1, 5);
    CK(hipDeviceSynchronize());
    printf("| COORD, 2 MLPs | %d | %.1f (%d) | %.3f | %.1f (%d) | %.3f | %.1f | %.2e |\n", cnt, u32, g32, fl / u32 / 1e6 / 157.3, u16, g16,
           fl / u16 / 1e6 / 157.3, u16b, maxdiff(d_xo[0], d_xo[1], (size_t)nr * 3));
  }
  return 0;
}
