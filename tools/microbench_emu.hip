// Emulated-fp32 edge kernels (edge_wave.h, EMU = 6 / 9: three-way bf16 split on v_mfma_f32_32x32x16_bf16) against the exact
// fp32 kernels: timing on the benchmark geometry AND the error of either against a float64 evaluation on the host
// (VERDICT r4 item 1, step (i) on the real kernels).  Derived from tools/microbench.hip (same synthetic batch).
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
  const bool with_ref = !(argc > 3 && atoi(argv[3]) == 0);   // third argument 0: timing only
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
  std::vector<float> h_pq = rnd(g, (size_t)N * 4 * H, 1.0f);
  float* d_pq = dev(h_pq);     // up to 4H columns (coordinate stage layout)
  auto mk = [&](size_t n, float s) { return dev(rnd(g, n, s)); };
  struct Mlp { float *wd, *wd0, *tab, *w2t, *b2, *w2tp; unsigned short* w2e; } m[2];
  std::vector<float> h_wd[2], h_wd0[2], h_tab[2], h_w2t[2], h_b2[2];
  for (int q = 0; q < 2; ++q) {
    h_wd[q] = rnd(g, H, 0.05f); h_wd0[q] = rnd(g, H, 0.05f); h_tab[q] = rnd(g, 3 * H, 0.3f); h_w2t[q] = rnd(g, (size_t)H * H, ws); h_b2[q] = rnd(g, H, 0.1f);
    m[q] = {dev(h_wd[q]), dev(h_wd0[q]), dev(h_tab[q]), dev(h_w2t[q]), dev(h_b2[q]), dev_zero<float>((size_t)H * H), dev_zero<unsigned short>((size_t)3 * H * H + 4096)};
    hipLaunchKernelGGL(permute_w2t_kernel, dim3((H * H + 255) / 256), dim3(256), 0, 0, (const float*)m[q].w2t, m[q].w2tp, H);
    hipLaunchKernelGGL(pack_w2e_kernel, dim3((H * H + 255) / 256), dim3(256), 0, 0, (const float*)m[q].w2t, m[q].w2e, H);
  }
  std::vector<float> h_attw = rnd(g, H, ws), h_attb = rnd(g, 1, 0.1f), h_w3 = rnd(g, H, ws);
  float *d_attw = dev(h_attw), *d_attb = dev(h_attb), *d_w3 = dev(h_w3);
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
      a.mlp[0] = EdgeMlpW{d_pq, d_pq + H, m[0].wd, m[0].wd0, m[0].tab, m[0].w2t, m[0].b2, m[0].w2tp, nullptr, m[0].w2e};
      a.mlp[1] = a.mlp[0];
      a.att_w = d_attw; a.att_b = d_attb; a.attention = 1; a.agg = d_agg; a.agg_head = d_head;
    } else {
      a.ldpq = 4 * H;
      a.mlp[0] = EdgeMlpW{d_pq + 2 * H, d_pq, m[0].wd, m[0].wd0, m[0].tab, m[0].w2t, m[0].b2, m[0].w2tp, nullptr, m[0].w2e};
      a.mlp[1] = EdgeMlpW{d_pq + 3 * H, d_pq + H, m[1].wd, m[1].wd0, m[1].tab, m[1].w2t, m[1].b2, m[1].w2tp, nullptr, m[1].w2e};
      a.w3 = d_w3; a.node_batch = d_nb; a.mean = d_mean; a.norm_constant = 1.f; a.coords_range = 15.f;
      a.use_tanh = 1; a.n_mlp = 2; a.xagg = d_xagg; a.xagg_head = d_xhead; a.xagg_stride = (size_t)N * 3;
      a.xhead_stride = (size_t)(E / 32 + 2) * 4; a.pass_split = 1;
    }
    return a;
  };
  int n_cu = 256;
  { hipDeviceProp_t p; if (hipGetDeviceProperties(&p, 0) == hipSuccess && p.multiProcessorCount > 0) n_cu = p.multiProcessorCount; }
  auto grid_of = [&](int mode, int edges) {
    long tiles = (edges + 127) / 128, gmax = (argc > 4 && atoi(argv[4]) > 0) ? atoi(argv[4]) : 2L * n_cu;   // fourth argument: cap on the grid (256 = one workgroup per CU)
    long gg = mode == MODE_COORD ? 2 * tiles : tiles;
    if (gg > gmax) gg = gmax;
    const int q8 = mode == MODE_COORD ? 16 : 8;
    return (int)std::max<long>((gg + q8 - 1) / q8 * q8, q8);
  };

  // ---- error against float64 on the host: the GCL stage's aggregate of sample 0's rows ------------------------
  auto launch = [&](int emu, int mode, const EdgeArgs& a, int grid) {
    if (mode == MODE_GCL) {
      if (emu == 0) hipLaunchKernelGGL((edge_wave_kernel<H, MODE_GCL, true>), dim3(grid), dim3(kThreads), 0, 0, a);
      else if (emu == 6) hipLaunchKernelGGL((edge_wave_kernel<H, MODE_GCL, false, 6>), dim3(grid), dim3(kThreads), 0, 0, a);
      else hipLaunchKernelGGL((edge_wave_kernel<H, MODE_GCL, false, 9>), dim3(grid), dim3(kThreads), 0, 0, a);
    } else {
      if (emu == 0) hipLaunchKernelGGL((edge_wave_kernel<H, MODE_COORD, true>), dim3(grid), dim3(kThreads), 0, 0, a);
      else if (emu == 6) hipLaunchKernelGGL((edge_wave_kernel<H, MODE_COORD, false, 6>), dim3(grid), dim3(kThreads), 0, 0, a);
      else hipLaunchKernelGGL((edge_wave_kernel<H, MODE_COORD, false, 9>), dim3(grid), dim3(kThreads), 0, 0, a);
    }
  };
  std::vector<int> rows_chk;
  for (int i = 0; i < nl; ++i) rows_chk.push_back(i);
  for (int i = 0; i < np; ++i) rows_chk.push_back(n_lig + i);
  std::vector<double> ref((size_t)rows_chk.size() * H, 0.0);
  if (with_ref) {
    const std::vector<float>&wd = h_wd[0], &wd0 = h_wd0[0], &tab = h_tab[0], &w2t = h_w2t[0], &b2 = h_b2[0];
    std::vector<double> a1(H), z(H);
    for (size_t ri = 0; ri < rows_chk.size(); ++ri) {
      const int i = rows_chk[ri];
      for (int e = row_ptr[i]; e < row_ptr[i] + deg[i]; ++e) {
        const int jn = ecol[e];
        // |d|^2 exactly as the kernel forms it (fp32 differences, fp32 sum): the inputs of the layer are the same fp32 numbers
        const float dx = x[3 * (size_t)i] - x[3 * (size_t)jn], dy = x[3 * (size_t)i + 1] - x[3 * (size_t)jn + 1], dzz = x[3 * (size_t)i + 2] - x[3 * (size_t)jn + 2];
        const float dd = dx * dx + dy * dy + dzz * dzz;
        const bool rl = i < n_lig, cl = jn < n_lig;
        const int ty = (rl && cl) ? 1 : ((!rl && !cl) ? 2 : 0);
        for (int k = 0; k < H; ++k) {
          const double pre = (double)h_pq[(size_t)i * 2 * H + k] + (double)h_pq[(size_t)jn * 2 * H + H + k] + (double)dd * wd[k] + (double)ed0[e] * wd0[k] + tab[ty * H + k];
          a1[k] = pre / (1.0 + std::exp(-pre));
        }
        double dot = h_attb[0];
        for (int f = 0; f < H; ++f) {
          double s = b2[f];
          for (int k = 0; k < H; ++k) s += a1[k] * (double)w2t[(size_t)k * H + f];
          z[f] = s / (1.0 + std::exp(-s));
          dot += z[f] * h_attw[f];
        }
        const double att = 1.0 / (1.0 + std::exp(-dot));
        for (int f = 0; f < H; ++f) ref[ri * H + f] += z[f] * att / 100.0;
      }
    }
  }
  if (with_ref) printf("\n## error of the GCL stage's aggregate against float64 (sample 0: %zu rows x %d features)\n\n", rows_chk.size(), H);
  if (with_ref) printf("| kernel | max abs err | max err / max |agg| | rms err / max |agg| | vs exact fp32 |\n|---|---|---|---|---|\n");
  double base_max = 0;
  std::vector<float> agg_of[3];
  const int emus[3] = {0, 6, 9};
  for (int v = 0; v < 3; ++v) {
    EdgeArgs a = edge_args(MODE_GCL, 0);
    CK(hipMemset(d_agg, 0, (size_t)N * H * 4)); CK(hipMemset(d_head, 0, (size_t)(E / 32 + 2) * H * 4));
    launch(emus[v], MODE_GCL, a, grid_of(MODE_GCL, E));
    hipLaunchKernelGGL(agg_complete_kernel, dim3((N + 3) / 4), dim3(kThreads), 0, 0, d_agg, (const float*)d_head, (const int*)d_rowptr,
                       (const int*)d_deg, N, H, E / 32 + 1, 5);
    CK(hipDeviceSynchronize());
    agg_of[v].resize((size_t)N * H);
    CK(hipMemcpy(agg_of[v].data(), d_agg, (size_t)N * H * 4, hipMemcpyDeviceToHost));
    double mx = 0, ss = 0, am = 0;
    for (size_t ri = 0; ri < rows_chk.size(); ++ri)
      for (int f = 0; f < H; ++f) {
        const double r = ref[ri * H + f], d = (double)agg_of[v][(size_t)rows_chk[ri] * H + f] - r;
        mx = std::max(mx, std::fabs(d)); ss += d * d; am = std::max(am, std::fabs(r));
      }
    if (v == 0) base_max = mx;
    if (with_ref) printf("| %s | %.3e | %.3e | %.3e | %.2f x |\n", v == 0 ? "exact fp32 (v_mfma_f32_32x32x2_f32)" : (v == 1 ? "emulated, 6 products" : "emulated, 9 products"),
           mx, mx / am, std::sqrt(ss / (rows_chk.size() * H)) / am, mx / base_max);
  }
  {
    double d6 = 0, d9 = 0, am = 0;
    for (size_t k = 0; k < (size_t)N * H; ++k) {
      d6 = std::max(d6, (double)std::fabs(agg_of[1][k] - agg_of[0][k])); d9 = std::max(d9, (double)std::fabs(agg_of[2][k] - agg_of[0][k]));
      am = std::max(am, (double)std::fabs(agg_of[0][k]));
    }
    printf("\nwhole batch (%d rows): max |emulated - exact| / max |agg| = %.3e (6 products), %.3e (9 products)\n", N, d6 / am, d9 / am);
  }

#ifdef DSBDD_DIAG_PHASES
  {   // in-situ phase clocks of the emulated GCL kernel over the whole list (hipcc -DDSBDD_DIAG_PHASES)
    EdgeArgs a = edge_args(MODE_GCL, 0);
    const int grid = grid_of(MODE_GCL, counts[0]);
    unsigned long long* d_ts = dev_zero<unsigned long long>((size_t)grid * 4 * 8);
    a.ts = d_ts;
    launch(6, MODE_GCL, a, grid);
    CK(hipDeviceSynchronize());
    CK(hipMemset(d_ts, 0, (size_t)grid * 4 * 8 * 8));
    const float us = time_us([&] { launch(6, MODE_GCL, a, grid); }, 1);
    std::vector<unsigned long long> ts((size_t)grid * 4 * 8);
    CK(hipMemcpy(ts.data(), d_ts, ts.size() * 8, hipMemcpyDeviceToHost));
    const char* pn[5] = {"outside the K loop (prologue, accumulator init, epilogue)", "activations, P / Q request, first staging loads, first B reads",
                         "MFMA phase (48 MFMAs, B reads, staging in the middle)", "trailing staging stores", "barrier"};
    double tot[5] = {0, 0, 0, 0, 0}, all = 0;
    for (size_t wv = 0; wv < (size_t)grid * 4; ++wv) for (int i = 0; i < 5; ++i) { tot[i] += (double)ts[wv * 8 + i] / 4.0; all += (double)ts[wv * 8 + i] / 4.0; }
    const double ksteps = (double)((counts[0] + 127) / 128) * (H / 16);          // (every launch overwrites its clocks)
    printf("\n## phase clocks, emulated GCL kernel, whole list, grid %d (%.1f us with the clocks in)\n\n| phase | share of the wave's time | shader cycles per K step |\n|---|---|---|\n", grid, us);
    for (int i = 0; i < 5; ++i) printf("| %s | %.3f | %.0f |\n", pn[i], tot[i] / all, tot[i] / ksteps);
    printf("| all | 1 | %.0f |\n", all / ksteps);
  }
#endif

  // ---- timing -----------------------------------------------------------------------------------------------------
  printf("\n## timing (B = %d)\n\n| kernel | edges | us / launch | TFLOP/s (algorithmic fp32 FLOPs) | frac of 157.3 | speed-up vs exact |\n|---|---|---|---|---|---|\n", B);
  const char* names[4] = {"GCL whole list", "GCL ligand-row prefix", "GCL 28 % prefix", "GCL 69 % prefix"};
  for (int ci : {0, 3, 2, 1}) {
    float base = 0;
    for (int v = 0; v < 3; ++v) {
      EdgeArgs a = edge_args(MODE_GCL, ci);
      const int grid = grid_of(MODE_GCL, counts[ci]);
      const float us = time_us([&] { launch(emus[v], MODE_GCL, a, grid); }, reps);
      if (v == 0) base = us;
      const double fl = 2.0 * counts[ci] * ((double)H * H + 4.0 * H);
      printf("| %s, %s | %d | %.1f | %.1f | %.3f | %.2f |\n", names[ci], v == 0 ? "exact" : (v == 1 ? "emu-6" : "emu-9"), counts[ci], us, fl / us / 1e6,
             fl / us / 1e6 / 157.3, base / us);
    }
  }
  {
    std::vector<float> xa[3];
    float base = 0;
    for (int v = 0; v < 3; ++v) {
      EdgeArgs a = edge_args(MODE_COORD, 1);
      const int grid = grid_of(MODE_COORD, E_lig);
      CK(hipMemset(d_xagg, 0, (size_t)2 * N * 3 * 4));
      const float us = time_us([&] { launch(emus[v], MODE_COORD, a, grid); }, reps);
      if (v == 0) base = us;
      xa[v].resize((size_t)2 * N * 3);
      CK(hipMemcpy(xa[v].data(), d_xagg, xa[v].size() * 4, hipMemcpyDeviceToHost));
      double dm = 0, am = 0;
      for (size_t k = 0; k < xa[v].size(); ++k) { dm = std::max(dm, (double)std::fabs(xa[v][k] - xa[0][k])); am = std::max(am, (double)std::fabs(xa[0][k])); }
      const double fl = 2.0 * 2.0 * E_lig * ((double)H * H + 3.0 * H);
      printf("| COORD ligand-row prefix, 2 MLPs, %s (max |diff to exact| / max = %.2e) | %d | %.1f | %.1f | %.3f | %.2f |\n",
             v == 0 ? "exact" : (v == 1 ? "emu-6" : "emu-9"), dm / am, E_lig, us, fl / us / 1e6, fl / us / 1e6 / 157.3, base / us);
    }
  }
  return 0;
}
