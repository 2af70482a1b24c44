// Distance-based bond assignment for a batch of generated molecules
// (reference: analysis/molecule_builder.py:30-55 get_bond_order_batch and the
// (X, A, E) construction of make_mol_edm, :101-118).  One wave per molecule walks
// the strictly lower triangle; a few hundred pairs per molecule -> latency bound,
// it exists so that the post-processing of a sampling batch stays on the device
// and needs one small copy (B * n_max^2 bytes) instead of B OpenBabel round trips.
#pragma once
#include "common.h"

namespace dsbdd {

struct BondArgs {
  const float* x;          // [N][3] Angstrom
  const int* atom_type;    // [N]
  const int* mol_off;      // [B+1] first atom of every molecule
  const float* b1; const float* b2; const float* b3;   // [A][A] pm
  float m1, m2, m3;        // margins, pm
  int n_types; int n_max;
  signed char* order;      // [B][n_max][n_max], zero-initialised by the caller
};

__global__ __launch_bounds__(64) void bond_orders_kernel(BondArgs p) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int lo = p.mol_off[b], n = min(p.mol_off[b + 1] - lo, p.n_max);
  const int pairs = n * (n - 1) / 2;
  signed char* out = p.order + (size_t)b * p.n_max * p.n_max;
  for (int q = lane; q < pairs; q += 64) {
    // q -> (i, j), i > j, row-major over the lower triangle
    int i = (int)((1.0f + sqrtf(1.0f + 8.0f * (float)q)) * 0.5f);
    while (i * (i - 1) / 2 > q) --i;
    while ((i + 1) * i / 2 <= q) ++i;
    const int j = q - i * (i - 1) / 2;
    const float dx = p.x[3 * (lo + i)] - p.x[3 * (lo + j)];
    const float dy = p.x[3 * (lo + i) + 1] - p.x[3 * (lo + j) + 1];
    const float dz = p.x[3 * (lo + i) + 2] - p.x[3 * (lo + j) + 2];
    // explicit rounding steps: bit-identical to the float32 oracle (no fma contraction)
    const float s = __fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz));
    const float d = __fmul_rn(100.0f, __fsqrt_rn(s));
    const int ti = p.atom_type[lo + i], tj = p.atom_type[lo + j];
    const int k = ti * p.n_types + tj;
    int o = 0;
    if (d < __fadd_rn(p.b1[k], p.m1)) o = 1;
    if (d < __fadd_rn(p.b2[k], p.m2)) o = 2;
    if (d < __fadd_rn(p.b3[k], p.m3)) o = 3;
    out[i * p.n_max + j] = (signed char)o;
  }
}

}  // namespace dsbdd
