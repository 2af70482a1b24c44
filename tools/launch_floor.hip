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

// Counters (unsigned words, zeroed by the host; every word on its own 128-byte line): ctr[0] top-level arrivals,
// ctr[32 * (1 + x)] arrivals of group x, ctr[32 * (9 + x)] generation word of group x, ctr[32 * 17] error flag.
// Groups = blockIdx.x & 7: the observed workgroup -> XCD mapping (speed only; the counts are exact for ANY placement).
// HIER = the XCD-hierarchical barrier of the guide (MI355X_MICROARCH.md "barrier-xcd"): every workgroup arrives on its
// group's counter (lane-0 agent release first); the group's LAST arriver -- the leader -- arrives on the top counter,
// polls it (relaxed loads, s_sleep), takes one agent acquire and publishes the group's generation word; every other
// workgroup polls only ITS group's generation word and takes one agent acquire.  8 pollers on the top line, n_wg / 8
// on each group line, instead of n_wg pollers on one line (what round 3's version did: 18 - 40 us per barrier).
// Generation counting: counters only grow, the target of barrier number g is (g + 1) * arrivals per generation.
constexpr int kLine = 32;
__device__ __forceinline__ bool poll_ge(unsigned* w, unsigned target, unsigned* err) {
  long spins = 0;
  while (__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
    __builtin_amdgcn_s_sleep(2);
    if (++spins > 400000) { __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return false; }
  }
  return true;
}

template <bool HIER>
__device__ __forceinline__ bool grid_barrier(unsigned* ctr, unsigned gen, unsigned n_wg) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    unsigned* err = ctr + kLine * 17;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (HIER) {
      const unsigned x = blockIdx.x & 7, per = n_wg >> 3;
      unsigned* grp = ctr + kLine * (1 + x);
      unsigned* genw = ctr + kLine * (9 + x);
      const unsigned ticket = __hip_atomic_fetch_add(grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (ticket == gen * per + per - 1) {                       // the group's leader
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = poll_ge(ctr, (gen + 1) * 8, err);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        __hip_atomic_store(genw, gen + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
        ok = poll_ge(genw, gen + 1, err);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      }
    } else {
      __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      ok = poll_ge(ctr, (gen + 1) * n_wg, err);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
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
    if (__hip_atomic_load(&ctr[kLine * 17], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
  }
}

int main() {
  const int n = 1 << 20, steps = 200;
  float* buf;
  unsigned* ctr;
  CK(hipMalloc(&buf, n * 4));
  CK(hipMemset(buf, 0, n * 4));
  CK(hipMalloc(&ctr, 32 * 18 * 4));
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
        CK(hipMemsetAsync(ctr, 0, 32 * 18 * 4, st));
        CK(hipEventRecord(a, st));
        if (hier) hipLaunchKernelGGL((persistent_kernel<true>), dim3(grid), dim3(256), 0, st, buf, cover, steps, ctr);
        else hipLaunchKernelGGL((persistent_kernel<false>), dim3(grid), dim3(256), 0, st, buf, cover, steps, ctr);
        CK(hipEventRecord(b, st));
        CK(hipEventSynchronize(b));
        CK(hipEventElapsedTime(&ms, a, b));
      }
      unsigned h[17] = {0};
      CK(hipMemcpy(&h[16], ctr + 32 * 17, 4, hipMemcpyDeviceToHost));
      printf("| one kernel, %d steps, %s grid barrier%s | %d | %.2f |\n", steps, hier ? "XCD-hierarchical" : "flat",
             h[16] ? " (BARRIER TIMED OUT)" : "", grid, ms * 1e3 / steps);
    }
  std::vector<float> hb(8);
  CK(hipMemcpy(hb.data(), buf, 32, hipMemcpyDeviceToHost));
  printf("\n(buf[0] = %.0f)\n", hb[0]);
  return 0;
}
