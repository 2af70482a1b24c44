// Wave-specialised, software-pipelined version of the fused edge-MLP kernels of
// edge_mlp.h (same math, same arguments, same results up to fp32 summation
// order; see that file for the algorithm and the reference citations).
//
// Why: in the homogeneous kernel every wave alternates between "stage the next
// K slice" (gather P/Q rows, SiLU, LDS writes), "MFMA" and a long epilogue
// (attention dot, segmented row sums), so the matrix pipe idles ~50 % of the
// time (rocprof: 52 % of the fp32 MFMA peak).  Here a workgroup has
//
//   waves 0-3  CONSUMERS : ds_read + v_mfma_f32_32x32x2_f32 only; at the end of a
//                          tile they dump SiLU(acc + b2) into the LDS message tile
//   waves 4-7  PRODUCERS : (a) gather + SiLU + LDS write of the NEXT K slice,
//                          (b) metadata (row, col, |d|^2, type) of the NEXT tile,
//                          (c) the whole epilogue of the PREVIOUS tile (attention
//                              gate, segmented sums in edge order, atomics)
//
// One consumer and one producer wave share each SIMD: the MFMA pipe and the VALU
// pipe run concurrently, so (a)-(c) are off the critical path.  All 8 waves run
// the same barrier sequence: NK + 1 intervals per "unit" (= tile x MLP pass):
//
//   interval s < NK : consumers  MFMA on slice s (stage buffer s & 1)
//                     producers  write slice s+1 (loads issued in interval s-1),
//                                issue loads of slice s+2, epilogue chunk s of the
//                                previous unit, metadata step s of the next tile
//   interval s = NK : consumers  acc -> LDS (message tile / partial head dots)
//                     producers  write slice 0 of the next unit, issue slice 1
//
// Tile = 64 edges x H features, K slice = 16 (LDS: 2 stage buffers 41 KB +
// message tile 66 KB + vectors/metadata: 1 workgroup of 512 threads per CU).
#pragma once
#include "common.h"
#include "edge_mlp.h"

namespace dsbdd {

template <int H, int MODE>
struct PipeLayout {
  static constexpr int BM = 64, BK = 16;
  static constexpr int LDA = BM + 1;
  static constexpr int A_BUF = BK * LDA;
  static constexpr int B_BUF = BK * H;
  static constexpr int STAGE = 2 * (A_BUF + B_BUF);
  static constexpr int LDM = H + 1;
  static constexpr int EPI = (MODE == MODE_GCL) ? BM * LDM : BM * 65;
  static constexpr int NV = (MODE == MODE_GCL) ? 1 : 2;
  static constexpr int VEC_PER = 7 * H;
  static constexpr int EPI_OFF = STAGE;
  static constexpr int VEC_OFF = EPI_OFF + EPI;
  static constexpr int META_OFF = VEC_OFF + NV * VEC_PER;
  static constexpr int META = 3 * 5 * BM;          // 3 tiles in flight x (row, col, type, d, d0)
  static constexpr int SC_OFF = META_OFF + META;   // att/phi0 [BM], phi1 [BM], trans [BM][3]
  static constexpr int TOTAL = SC_OFF + 5 * BM;
};

template <int H, int MODE>
__global__ __launch_bounds__(512) void edge_pipe_kernel(EdgeArgs p) {
  using L = PipeLayout<H, MODE>;
  constexpr int BM = L::BM, BK = L::BK, LDA = L::LDA, LDM = L::LDM;
  constexpr int CT = H / 64;           // 32-col MFMA tiles per consumer wave (wave = 32 edges x H/2)
  constexpr int NQ = H / 4;            // float4 per W2T row
  constexpr int BI = BK * NQ / 256;    // float4 of the B slice per producer thread
  constexpr int NK = H / BK;           // K slices per unit
  constexpr int TPR = 256 / BM;        // producer threads per edge in the row reductions (4)
  constexpr int RPI0 = (BM + NK - 2) / (NK - 1);
  constexpr int RPI = ((RPI0 < 8 ? 8 : RPI0) + 7) / 8 * 8;   // rows of the segmented sum per interval
  constexpr int SEG_INTERVALS = (BM + RPI - 1) / RPI;         // intervals 1 .. SEG_INTERVALS
  static_assert(H % 64 == 0 && H <= 256, "hidden_nf must be 64,128,192 or 256");
  static_assert((BK * NQ) % 256 == 0, "B slice split");
  static_assert(SEG_INTERVALS <= NK - 1, "epilogue must finish before the tile is overwritten");
  static_assert(NK >= 4 && NK % 2 == 0, "metadata pipeline needs 3 intervals; 2 stage buffers need an even NK");
  // interval in which the loads of the NEXT unit's slice 0 are issued: after the
  // next tile's metadata is in LDS (stored in interval 2, visible from interval 3)
  constexpr int S_NEXT = (NK - 2 >= 3) ? NK - 2 : NK - 1;

  __shared__ float smem[L::TOTAL];
  float* sA = smem;
  float* sB = smem + 2 * L::A_BUF;
  float* sE = smem + L::EPI_OFF;            // message tile [BM][H+1] / partial dots [BM][65]
  float* sV = smem + L::VEC_OFF;
  float* s_meta = smem + L::META_OFF;       // [3][5][BM]
  float* s_s0 = smem + L::SC_OFF;           // attention / phi (coord)
  float* s_s1 = s_s0 + BM;                  // phi (cross)
  float* s_tr = s_s0 + 2 * BM;              // trans [BM][3]

  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const bool producer = w >= 4;
  const int pt = t - 256;                   // producer thread id (valid if producer)
  const int cw = w & 3, wm = cw >> 1, wn = cw & 1;   // consumer wave position (2 x 2)
  const int n_pass = (MODE == MODE_GCL) ? 1 : p.n_mlp;

  for (int q = 0; q < n_pass; ++q) {
    const EdgeMlpW& mw = p.mlp[q];
    float* v = sV + q * L::VEC_PER;
    for (int i = t; i < H; i += 512) {
      v[i] = mw.wd[i];
      v[H + i] = mw.wd0[i];
      v[2 * H + i] = mw.table[i];
      v[3 * H + i] = mw.table[H + i];
      v[4 * H + i] = mw.table[2 * H + i];
      v[5 * H + i] = mw.b2[i];
      v[6 * H + i] = (MODE == MODE_GCL) ? (p.attention ? p.att_w[i] : 0.f) : p.w3[i];
    }
  }
  const float att_b = (MODE == MODE_GCL && p.attention) ? p.att_b[0] : 0.f;

  const int E = *p.e_count;
  const int ntiles = (E + BM - 1) / BM;
  const int xcd = blockIdx.x & 7, kx = blockIdx.x >> 3, gx = gridDim.x >> 3;
  const int tq = ntiles / 8, tr = ntiles % 8;
  const int csize = tq + (xcd < tr ? 1 : 0);
  const int cbase = (xcd < tr) ? xcd * (tq + 1) : tr * (tq + 1) + (xcd - tr) * tq;
  const int my_tiles = (csize > kx) ? (csize - kx + gx - 1) / gx : 0;   // tiles cbase + kx + j*gx
  const int U = my_tiles * n_pass;                                      // units
  if (U == 0) return;
  auto tile_of = [&](int j) { return cbase + kx + j * gx; };

  // ---- producer state -----------------------------------------------------------
  int nx_r = -1, nx_c = 0;
  float nx_d0 = 0.f, nx_xr[3] = {0.f, 0.f, 0.f}, nx_xc[3] = {0.f, 0.f, 0.f};
  auto meta_fetch_idx = [&](int tile_id) {
    nx_r = -1; nx_c = 0; nx_d0 = 0.f;
    const int e = tile_id * BM + pt;
    if (pt < BM && e < E) { nx_r = p.erow[e]; nx_c = p.ecol[e]; nx_d0 = p.ed0[e]; }
  };
  auto meta_fetch_x = [&]() {
    if (pt < BM && nx_r >= 0) {
#pragma unroll
      for (int k = 0; k < 3; ++k) { nx_xr[k] = p.x[3 * nx_r + k]; nx_xc[k] = p.x[3 * nx_c + k]; }
    }
  };
  auto meta_store = [&](int buf) {
    if (pt < BM) {
      float* mb = s_meta + buf * 5 * BM;
      float d = 0.f;
      int ty = 0;
      if (nx_r >= 0) {
        const float dx = nx_xr[0] - nx_xc[0], dy = nx_xr[1] - nx_xc[1], dz = nx_xr[2] - nx_xc[2];
        d = dx * dx + dy * dy + dz * dz;
        const bool rl = nx_r < p.n_lig, cl = nx_c < p.n_lig;
        ty = (rl && cl) ? 1 : ((!rl && !cl) ? 2 : 0);
      }
      reinterpret_cast<int*>(mb)[pt] = nx_r;
      reinterpret_cast<int*>(mb)[BM + pt] = nx_c;
      reinterpret_cast<int*>(mb)[2 * BM + pt] = ty;
      mb[3 * BM + pt] = d;
      mb[4 * BM + pt] = nx_d0;
    }
  };

  const int a_kq = (pt & 3) * 4, a_m = (pt >> 2) & (BM - 1);   // A: 4 lanes x float4 = 16 k of one edge
  float4 rp = make_float4(0.f, 0.f, 0.f, 0.f), rq = rp, rb[BI];
#pragma unroll
  for (int i = 0; i < BI; ++i) rb[i] = rp;
  // loads of K slice `ks` of unit `u` (tile metadata buffer mb, MLP pass q)
  auto gload = [&](int mbuf, int q, int ks) {
    const EdgeMlpW& mw = p.mlp[q];
    const int* m_row = reinterpret_cast<const int*>(s_meta + mbuf * 5 * BM);
    const int r = m_row[a_m], c = m_row[BM + a_m];
    const int rr = r < 0 ? 0 : r;
    rp = ld4(mw.P + (size_t)rr * p.ldpq + ks * BK + a_kq);
    rq = ld4(mw.Q + (size_t)c * p.ldpq + ks * BK + a_kq);
#pragma unroll
    for (int i = 0; i < BI; ++i) rb[i] = ld4(mw.W2T + (size_t)ks * BK * H + (pt + 256 * i) * 4);
  };
  auto sstore = [&](int mbuf, int q, int ks) {
    const float* vq = sV + q * L::VEC_PER;
    const float* mb = s_meta + mbuf * 5 * BM;
    float* a = sA + (ks & 1) * L::A_BUF;
    float* b = sB + (ks & 1) * L::B_BUF;
    const float d = mb[3 * BM + a_m], d0 = mb[4 * BM + a_m];
    const int ty = reinterpret_cast<const int*>(mb)[2 * BM + a_m];
    const float* tab = vq + (2 + ty) * H + ks * BK + a_kq;
    const float* wd = vq + ks * BK + a_kq;
    const float* wd0 = vq + H + ks * BK + a_kq;
    const float pv[4] = {rp.x, rp.y, rp.z, rp.w};
    const float qv[4] = {rq.x, rq.y, rq.z, rq.w};
#pragma unroll
    for (int c = 0; c < 4; ++c)
      a[(a_kq + c) * LDA + a_m] = silu(pv[c] + qv[c] + d * wd[c] + d0 * wd0[c] + tab[c]);
#pragma unroll
    for (int i = 0; i < BI; ++i) *reinterpret_cast<float4*>(b + (pt + 256 * i) * 4) = rb[i];
  };

  // segmented-sum state of the epilogue (GCL: one feature per producer thread)
  int seg_cur = -1;
  float seg_sum = 0.f;

  // ---- prologue: metadata of the first tile, slice 0 of unit 0 ----------------------
  if (producer) {
    meta_fetch_idx(tile_of(0));
    meta_fetch_x();
    meta_store(0);
  }
  __syncthreads();          // sV + metadata visible
  if (producer) {
    gload(0, 0, 0);
    sstore(0, 0, 0);
    gload(0, 0, 1);         // in flight: slice 1
  }

  f32x16 acc[CT];

  // u == U is the drain iteration (epilogue of the last unit only)
  for (int u = 0; u <= U; ++u) {
    const int j = u / n_pass, q = u - j * n_pass;          // local tile index, MLP pass
    const int mbuf = j % 3;
    const bool live = u < U;
    const int un = u + 1, jn = un / n_pass, qn = un - jn * n_pass;   // next unit
    const bool next_live = un < U;
    const bool next_new_tile = next_live && jn != j;
    const int up = u - 1, jp = up / n_pass, qp = up - jp * n_pass;   // previous unit (u >= 1)
    const int pbuf = ((jp % 3) + 3) % 3;
    const int s_end = live ? NK : (MODE == MODE_GCL ? SEG_INTERVALS : 2);   // drain: epilogue intervals only

    if (!producer && live) {
#pragma unroll
      for (int c = 0; c < CT; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    }

#pragma unroll 1
    for (int s = 0; s <= s_end; ++s) {
      __syncthreads();
      if (!producer) {
        // =================== CONSUMERS ===================
        if (live) {
          if (s < NK) {
            const float* pa = sA + (s & 1) * L::A_BUF + (lane >> 5) * LDA + wm * 32 + (lane & 31);
            const float* pb = sB + (s & 1) * L::B_BUF + (lane >> 5) * H + wn * (H / 2) + (lane & 31);
#pragma unroll
            for (int kk = 0; kk < BK; kk += 2) {
              const float a = pa[kk * LDA];
              float b[CT];
#pragma unroll
              for (int c = 0; c < CT; ++c) b[c] = pb[kk * H + c * 32];
#pragma unroll
              for (int c = 0; c < CT; ++c) acc[c] = mfma32(a, b[c], acc[c]);
            }
          } else if (s == NK) {
            const float* vq = sV + q * L::VEC_PER;
            if (MODE == MODE_GCL) {
              // messages: SiLU(acc + b2) -> sE[edge][feature]   (egnn_new.py:18-19)
#pragma unroll
              for (int c = 0; c < CT; ++c) {
                const int col = wn * (H / 2) + c * 32 + (lane & 31);
                const float bv = vq[5 * H + col];
#pragma unroll
                for (int r = 0; r < 16; ++r)
                  sE[(wm * 32 + mfma_row(r, lane)) * LDM + col] = silu(acc[c][r] + bv);
              }
            } else {
              // scalar head partial dots: sum_col SiLU(acc + b2) * w3   (egnn_new.py:80-92)
              float part[16];
#pragma unroll
              for (int r = 0; r < 16; ++r) part[r] = 0.f;
#pragma unroll
              for (int c = 0; c < CT; ++c) {
                const int col = wn * (H / 2) + c * 32 + (lane & 31);
                const float bv = vq[5 * H + col], wv = vq[6 * H + col];
#pragma unroll
                for (int r = 0; r < 16; ++r) part[r] += silu(acc[c][r] + bv) * wv;
              }
#pragma unroll
              for (int r = 0; r < 16; ++r)
                sE[(wm * 32 + mfma_row(r, lane)) * 65 + wn * 32 + (lane & 31)] = part[r];
            }
          }
        }
      } else {
        // =================== PRODUCERS ===================
        // ---- (a) staging -------------------------------------------------------------
        if (live) {
          if (s + 1 < NK) {
            sstore(mbuf, q, s + 1);                       // loads were issued one interval ago
            if (s + 2 < NK) gload(mbuf, q, s + 2);
          }
          if (s == S_NEXT && next_live) {                 // staging registers are free here
            gload(next_new_tile ? (mbuf + 1) % 3 : mbuf, qn, 0);
          } else if (s == NK && next_live) {
            const int nb = next_new_tile ? (mbuf + 1) % 3 : mbuf;
            sstore(nb, qn, 0);
            gload(nb, qn, 1);
          }
        }
        // ---- (b) metadata of the next tile ---------------------------------------------
        if (live && next_new_tile) {
          if (s == 0) meta_fetch_idx(tile_of(jn));
          if (s == 1) meta_fetch_x();
          if (s == 2) meta_store((mbuf + 1) % 3);
        }
        // ---- (c) epilogue of the previous unit -----------------------------------------
        if (u >= 1) {
          const int* p_row = reinterpret_cast<const int*>(s_meta + pbuf * 5 * BM);
          const float* vqp = sV + qp * L::VEC_PER;
          if (MODE == MODE_GCL) {
            if (s == 0) {
              // attention gate (egnn_new.py:26-29,38-40): 4 threads per edge
              const int el = pt / TPR, part = pt % TPR;
              float att = 1.f;
              if (p.attention) {
                float dot = 0.f;
                const float* mrow = sE + el * LDM;
                const float* aw = vqp + 6 * H;
                constexpr int CH = H / TPR;
#pragma unroll 8
                for (int i = 0; i < CH; ++i) {
                  const int k = part * CH + (i + part * (32 / TPR)) % CH;
                  dot += mrow[k] * aw[k];
                }
#pragma unroll
                for (int o = 1; o < TPR; o <<= 1) dot += __shfl_xor(dot, o);
                att = sigmoidf_fast(dot + att_b);
              }
              if (part == 0) s_s0[el] = att;
              seg_cur = -1;
              seg_sum = 0.f;
            } else if (s <= SEG_INTERVALS && pt < H) {
              // segmented sum in edge order, RPI edges per interval, one feature per thread
              const int e_begin = (s - 1) * RPI;
#pragma unroll 1
              for (int e0 = e_begin; e0 < e_begin + RPI && e0 < BM; e0 += 8) {
                int rr[8];
                float v[8];
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                  rr[jj] = p_row[e0 + jj];
                  v[jj] = sE[(e0 + jj) * LDM + pt] * s_s0[e0 + jj];   // mij * att, egnn_new.py:40
                }
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) {
                  const int r = __builtin_amdgcn_readfirstlane(rr[jj]);
                  if (r != seg_cur) {
                    if (seg_cur >= 0)
                      unsafeAtomicAdd(&p.agg[(size_t)seg_cur * H + pt], seg_sum / p.norm_factor);
                    seg_cur = r;
                    seg_sum = 0.f;
                  }
                  seg_sum += v[jj];
                }
              }
              if (s == SEG_INTERVALS && seg_cur >= 0) {
                unsafeAtomicAdd(&p.agg[(size_t)seg_cur * H + pt], seg_sum / p.norm_factor);
                seg_cur = -1;
              }
            }
          } else {
            if (s == 0) {
              const int el = pt / TPR, part = pt % TPR;
              float sum = 0.f;
              constexpr int CH = 64 / TPR;
#pragma unroll
              for (int i = 0; i < CH; ++i) sum += sE[el * 65 + part * CH + (i + part * (32 / TPR)) % CH];
#pragma unroll
              for (int o = 1; o < TPR; o <<= 1) sum += __shfl_xor(sum, o);
              if (part == 0) (qp == 0 ? s_s0 : s_s1)[el] = sum;
            } else if (qp == n_pass - 1) {
              if (s == 1 && pt < BM) {
                // trans = u*phi + cross*phi_x   (egnn_new.py:100-109, 296-316)
                const int r = p_row[pt];
                float tx = 0.f, ty = 0.f, tz = 0.f;
                if (r >= 0) {
                  const int c = p_row[BM + pt];
                  const float xr0 = p.x[3 * r], xr1 = p.x[3 * r + 1], xr2 = p.x[3 * r + 2];
                  const float xc0 = p.x[3 * c], xc1 = p.x[3 * c + 1], xc2 = p.x[3 * c + 2];
                  const float dx = xr0 - xc0, dy = xr1 - xc1, dz = xr2 - xc2;
                  const float radial = dx * dx + dy * dy + dz * dz;
                  const float den = sqrtf(radial + 1e-8f) + p.norm_constant;
                  const float ux = dx / den, uy = dy / den, uz = dz / den;
                  const float phi = s_s0[pt];
                  if (p.use_tanh) {
                    const float th = tanhf(phi);
                    tx = ux * th * p.coords_range; ty = uy * th * p.coords_range; tz = uz * th * p.coords_range;
                  } else {
                    tx = ux * phi; ty = uy * phi; tz = uz * phi;
                  }
                  if (p.n_mlp == 2) {
                    const int b = p.node_batch[r];
                    const float m0 = p.mean[3 * b], m1 = p.mean[3 * b + 1], m2 = p.mean[3 * b + 2];
                    const float a0 = xr0 - m0, a1 = xr1 - m1, a2 = xr2 - m2;
                    const float b0 = xc0 - m0, b1 = xc1 - m1, b2 = xc2 - m2;
                    const float c0 = a1 * b2 - a2 * b1, c1 = a2 * b0 - a0 * b2, c2 = a0 * b1 - a1 * b0;
                    const float cden = sqrtf(c0 * c0 + c1 * c1 + c2 * c2) + p.norm_constant;
                    float phx = s_s1[pt];
                    if (p.use_tanh) phx = tanhf(phx) * p.coords_range;
                    tx += c0 / cden * phx; ty += c1 / cden * phx; tz += c2 / cden * phx;
                  }
                }
                s_tr[3 * pt] = tx; s_tr[3 * pt + 1] = ty; s_tr[3 * pt + 2] = tz;
              } else if (s == 2 && pt < 3) {
                int cur = -1;
                float sum = 0.f;
#pragma unroll 1
                for (int e0 = 0; e0 < BM; e0 += 16) {
                  int rr[16];
                  float v[16];
#pragma unroll
                  for (int jj = 0; jj < 16; ++jj) { rr[jj] = p_row[e0 + jj]; v[jj] = s_tr[3 * (e0 + jj) + pt]; }
#pragma unroll
                  for (int jj = 0; jj < 16; ++jj) {
                    const int r = rr[jj];
                    if (r != cur) {
                      if (cur >= 0) unsafeAtomicAdd(&p.xagg[(size_t)cur * 3 + pt], sum / p.norm_factor);
                      cur = r;
                      sum = 0.f;
                    }
                    sum += v[jj];
                  }
                }
                if (cur >= 0) unsafeAtomicAdd(&p.xagg[(size_t)cur * 3 + pt], sum / p.norm_factor);
              }
            }
          }
        }
      }
    }  // intervals
  }    // units
}

}  // namespace dsbdd
