// Design input for the latency regime (C-alpha config: ~50 launches of 5 - 50 us per EGNN call): what does a kernel
// boundary cost on this GPU, in a stream and inside a captured graph, and what does a grid-wide barrier inside one
// persistent kernel cost?
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/launch_floor.hip -o tools/bin/launch_floor && tools/bin/launch_floor
//
// (1) N dependent, trivially small kernels (every thread adds 1 to its word) at several grid sizes: us per kernel in
//     stream order and as one hipGraph replay;
// (2) the same N steps inside ONE kernel of 256 / 512 workgroups with a grid barrier between steps -- a flat atomic
//     counter and an XCD-hierarchical one (arrive on a per-XCD counter, the last arrival of an XCD arrives on the global
//     one); every spin is bounded (a barrier that does not complete within ~50 ms sets an error flag and the kernel
//     leaves), so a scheduling surprise cannot hang the box.
// Test / measurement infrastructure, not product code.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void step_kernel(float* buf, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) buf[i] += 1.f;
}

// ctr[0]: global arrivals, ctr[1 .. 8]: per-XCD arrivals, ctr[16]: error flag.  Generation-counting barrier: the
// target of barrier number g is g * (number of arrivals per generation); counters only grow (zeroed by the host).
template <bool HIER>
__device__ __forceinline__ bool grid_barrier(unsigned* ctr, unsigned gen, unsigned n_wg) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    __threadfence();
    if (HIER) {
      const unsigned xcd = blockIdx.x & 7, per = n_wg >> 3;
      if (atomicAdd(&ctr[1 + xcd], 1u) == gen * per + per - 1) atomicAdd(&ctr[0], 1u);
      const unsigned target = (gen + 1) * 8;
      long spins = 0;
      while (__hip_atomic_load(&ctr[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(4);         // back off: the pollers share one L2 line with the arriving atomics
        if (++spins > 400000) { ctr[16] = 1; ok = false; break; }
      }
    } else {
      atomicAdd(&ctr[0], 1u);
      const unsigned target = (gen + 1) * n_wg;
      long spins = 0;
      while (__hip_atomic_load(&ctr[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(4);
        if (++spins > 400000) { ctr[16] = 1; ok = false; break; }
      }
    }
    __threadfence();
  }
  __syncthreads();
  return ok;
}

template <bool HIER>
__global__ __launch_bounds__(256) void persistent_kernel(float* buf, int n, int steps, unsigned* ctr) {
  const unsigned n_wg = gridDim.x;
  for (int s = 0; s < steps; ++s) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += n_wg * blockDim.x) buf[i] += 1.f;
    if (!grid_barrier<HIER>(ctr, (unsigned)s, n_wg)) return;
    if (__hip_atomic_load(&ctr[16], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
  }
}

int main() {
  const int n = 1 << 20, steps = 200;
  float* buf;
  unsigned* ctr;
  CK(hipMalloc(&buf, n * 4));
  CK(hipMemset(buf, 0, n * 4));
  CK(hipMalloc(&ctr, 64 * 4));
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  printf("| what | grid | us per step |\n|---|---|---|\n");
  for (int grid : {64, 512, 4096}) {
    const int cover = grid * 256;
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(step_kernel, dim3(grid), dim3(256), 0, st, buf, cover);
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(a, st));
    for (int i = 0; i < steps; ++i) hipLaunchKernelGGL(step_kernel, dim3(grid), dim3(256), 0, st, buf, cover);
    CK(hipEventRecord(b, st));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    printf("| %d dependent kernels, stream order | %d | %.2f |\n", steps, grid, ms * 1e3 / steps);
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < steps; ++i) hipLaunchKernelGGL(step_kernel, dim3(grid), dim3(256), 0, st, buf, cover);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(a, st));
    CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(b, st));
    CK(hipEventSynchronize(b));
    CK(hipEventElapsedTime(&ms, a, b));
    printf("| the same as one hipGraph replay | %d | %.2f |\n", grid, ms * 1e3 / steps);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  }
  for (int hier = 0; hier < 2; ++hier)
    for (int grid : {256, 512}) {
      const int cover = grid * 256;
      float ms = 0;
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipMemsetAsync(ctr, 0, 64 * 4, st));
        CK(hipEventRecord(a, st));
        if (hier) hipLaunchKernelGGL((persistent_kernel<true>), dim3(grid), dim3(256), 0, st, buf, cover, steps, ctr);
        else hipLaunchKernelGGL((persistent_kernel<false>), dim3(grid), dim3(256), 0, st, buf, cover, steps, ctr);
        CK(hipEventRecord(b, st));
        CK(hipEventSynchronize(b));
        CK(hipEventElapsedTime(&ms, a, b));
      }
      unsigned h[17];
      CK(hipMemcpy(h, ctr, 17 * 4, hipMemcpyDeviceToHost));
      printf("| one kernel, %d steps, %s grid barrier%s | %d | %.2f |\n", steps, hier ? "XCD-hierarchical" : "flat",
             h[16] ? " (BARRIER TIMED OUT)" : "", grid, ms * 1e3 / steps);
    }
  std::vector<float> hb(8);
  CK(hipMemcpy(hb.data(), buf, 32, hipMemcpyDeviceToHost));
  printf("\n(buf[0] = %.0f)\n", hb[0]);
  return 0;
}
