"""CPU emulation of what libdiffsbdd_hip.so computes, written against the PACKED
weight slots and following diffsbdd_amd/csrc/engine.hip launch by launch.

Purpose (host-logic tests, no GPU): prove that the weight packing
(diffsbdd_amd/engine.pack_weights), the slot order, the paddings, the exact
first-layer factorisation, the edge-prefix trick for update_coords_mask and the
orchestration order reproduce the oracle -- so that a GPU mismatch can only be
a kernel bug, not an algebra/plumbing bug.  Also emulates the MFMA 32x32x2
lane mapping used by the kernels' LDS indexing.
"""
import numpy as np
import torch
import torch.nn.functional as F

from diffsbdd_amd import _lib


def pad4(v):
    return (v + 3) & ~3


def node_linear(A1, K1, A2, K2, WT, bias, R, N, act):
    """C = act([A1[:, :K1] | A2[:, :K2]] @ WT[:K1+K2, :N] + bias) + R"""
    A = A1[:, :K1] if A2 is None else torch.cat([A1[:, :K1], A2[:, :K2]], 1)
    C = A @ WT[:K1 + (K2 if A2 is not None else 0), :N]
    if bias is not None:
        C = C + bias[:N]
    if act:
        C = F.silu(C)
    if R is not None:
        C = C + R[:, :N]
    return C


def edge_mlp_first(P, Q, row, col, d, d0, typ, wd, wd0, tab):
    pre = P[row] + Q[col] + d[:, None] * wd[None] + d0[:, None] * wd0[None] + tab[typ]
    return F.silu(pre)


def build_edges(x, mask_l, mask_p, cfg):
    """graph.h edges_kernel semantics: exact distance, sqrt(d2) <= cutoff."""
    nl = len(mask_l)
    m = torch.cat([mask_l, mask_p])
    d = x[:, None, :] - x[None, :, :]
    d2 = (d * d).sum(-1)
    dist = torch.sqrt(d2)
    N = len(m)
    is_l = torch.arange(N) < nl
    ll = is_l[:, None] & is_l[None, :]
    pp = (~is_l[:, None]) & (~is_l[None, :])
    ok = torch.ones(N, N, dtype=torch.bool)
    if cfg.has_cutoff_ligand:
        ok[ll] = (dist <= cfg.cutoff_ligand)[ll]
    if cfg.has_cutoff_pocket:
        ok[pp] = (dist <= cfg.cutoff_pocket)[pp]
    if cfg.has_cutoff_interaction:
        lp = ~(ll | pp)
        ok[lp] = (dist <= cfg.cutoff_interaction)[lp]
    adj = ok & (m[:, None] == m[None, :])
    row, col = torch.where(adj)
    return row, col


def forward(cfg, slots, xh_lig, xh_pocket, t, mask_l, mask_p, edges=None, trace=None):
    """Mirror of dsbdd_dynamics_forward."""
    c = cfg
    H, J = c.hidden_nf, c.joint_nf
    JP = pad4(J + 1)
    a, r = c.atom_nf, c.residue_nf
    nl, npk = xh_lig.shape[0], xh_pocket.shape[0]
    N = nl + npk
    n_mlp = 1 if c.reflection_equivariant else 2
    G = {n: i for i, n in enumerate(_lib.G_NAMES)}
    per = c.inv_sublayers * len(_lib.GCL_NAMES) + len(_lib.EQ_NAMES)

    def gcl(b, s, name):
        return slots[len(_lib.G_NAMES) + b * per + s * len(_lib.GCL_NAMES) + _lib.GCL_NAMES.index(name)]

    def eq(b, name):
        return slots[len(_lib.G_NAMES) + b * per + c.inv_sublayers * len(_lib.GCL_NAMES)
                     + _lib.EQ_NAMES.index(name)]

    node_batch = torch.cat([mask_l, mask_p])
    B = int(node_batch.max()) + 1
    x = torch.cat([xh_lig[:, :3], xh_pocket[:, :3]]).clone()
    x_in = x.clone()
    tt = t.reshape(-1)
    h0 = torch.zeros(N, JP)
    h0[:, J] = tt[0] if tt.numel() == 1 else tt[node_batch]
    # encoders
    e1 = node_linear(xh_lig[:, 3:], a, None, 0, slots[G["ATOM_ENC_W0T"]], slots[G["ATOM_ENC_B0"]], None, 2 * a, 1)
    h0[:nl, :J] = node_linear(e1, 2 * a, None, 0, slots[G["ATOM_ENC_W1T"]], slots[G["ATOM_ENC_B1"]], None, J, 0)
    e2 = node_linear(xh_pocket[:, 3:], r, None, 0, slots[G["RES_ENC_W0T"]], slots[G["RES_ENC_B0"]], None, 2 * r, 1)
    h0[nl:, :J] = node_linear(e2, 2 * r, None, 0, slots[G["RES_ENC_W1T"]], slots[G["RES_ENC_B1"]], None, J, 0)
    # edges
    if edges is None:
        row, col = build_edges(x, mask_l, mask_p, c)
    else:
        row, col = edges[0].long(), edges[1].long()
    d0 = ((x[row] - x[col]) ** 2).sum(1)
    rl, cl = row < nl, col < nl
    typ = torch.zeros(len(row), dtype=torch.long)
    typ[rl & cl] = 1
    typ[(~rl) & (~cl)] = 2
    # embedding
    h = node_linear(h0, JP, None, 0, slots[G["EMB_WT"]], slots[G["EMB_B"]], None, H, 0)
    n_upd = N if c.update_pocket_coords else nl
    e_upd = int((row < n_upd).sum())       # = row_ptr[n_upd]: a prefix of the row-sorted list
    assert bool((row[:e_upd] < n_upd).all())
    for blk in range(c.n_layers):
        if n_mlp == 2:
            mean = torch.zeros(B, 3).index_add_(0, node_batch, x)
            cnt = torch.zeros(B).index_add_(0, node_batch, torch.ones(N)).clamp(min=1)
            mean = mean / cnt[:, None]
        d = ((x[row] - x[col]) ** 2).sum(1)
        for sub in range(c.inv_sublayers):
            pq = node_linear(h, H, None, 0, gcl(blk, sub, "E1_WT"), None, None, 2 * H, 0)
            a1 = edge_mlp_first(pq[:, :H], pq[:, H:2 * H], row, col, d, d0, typ,
                                gcl(blk, sub, "E1_WD"), gcl(blk, sub, "E1_WD0"), gcl(blk, sub, "E1_TAB"))
            m = F.silu(a1 @ gcl(blk, sub, "E2_WT")[:, :H] + gcl(blk, sub, "E2_B"))
            if c.attention:
                att = torch.sigmoid(m @ gcl(blk, sub, "ATT_W") + gcl(blk, sub, "ATT_B"))
                m = m * att[:, None]
            agg = torch.zeros(N, H).index_add_(0, row, m) / c.normalization_factor
            t1 = node_linear(h, H, agg, H, gcl(blk, sub, "N1_WT"), gcl(blk, sub, "N1_B"), None, H, 1)
            h = node_linear(t1, H, None, 0, gcl(blk, sub, "N2_WT"), gcl(blk, sub, "N2_B"), h, H, 0)
        PQ = (2 if c.reflection_equivariant else 4) * H
        pq = node_linear(h, H, None, 0, eq(blk, "C1_WT"), None, None, PQ, 0)
        ru, cu, du, d0u, tu = row[:e_upd], col[:e_upd], d[:e_upd], d0[:e_upd], typ[:e_upd]
        w3 = eq(blk, "W3")
        QW = n_mlp * H   # column order [Q_coord | Q_cross | P_coord | P_cross]
        if not c.update_pocket_coords:
            # engine computes the Q part only for active nodes and the P part only for ligand rows
            active = torch.zeros(N, dtype=torch.bool)
            active[:nl] = True
            active[cu] = True
            pq = pq.clone()
            pq[~active, :QW] = float("nan")
            pq[nl:, QW:] = float("nan")
        a1 = edge_mlp_first(pq[:, QW:QW + H], pq[:, :H], ru, cu, du, d0u, tu, eq(blk, "C_WD"), eq(blk, "C_WD0"),
                            eq(blk, "C_TAB"))
        phi = F.silu(a1 @ eq(blk, "C_W2T")[:, :H] + eq(blk, "C_B2")) @ w3
        diff = x[ru] - x[cu]
        u = diff / (torch.sqrt((diff * diff).sum(1, keepdim=True) + 1e-8) + c.norm_constant)
        if c.use_tanh:
            trans = u * torch.tanh(phi)[:, None] * c.coords_range
        else:
            trans = u * phi[:, None]
        if n_mlp == 2:
            a1 = edge_mlp_first(pq[:, QW + H:QW + 2 * H], pq[:, H:2 * H], ru, cu, du, d0u, tu, eq(blk, "X_WD"),
                                eq(blk, "X_WD0"), eq(blk, "X_TAB"))
            phx = F.silu(a1 @ eq(blk, "X_W2T")[:, :H] + eq(blk, "X_B2")) @ w3
            if c.use_tanh:
                phx = torch.tanh(phx) * c.coords_range
            aa = x[ru] - mean[node_batch[ru]]
            bb = x[cu] - mean[node_batch[cu]]
            cr = torch.linalg.cross(aa, bb, dim=1)
            cr = cr / (torch.linalg.norm(cr, dim=1, keepdim=True) + c.norm_constant)
            trans = trans + cr * phx[:, None]
        xagg = torch.zeros(N, 3).index_add_(0, ru, trans) / c.normalization_factor
        x = x.clone()
        x[:n_upd] += xagg[:n_upd]
        if trace is not None:
            trace.append((h.clone(), x.clone()))
    hout = node_linear(h, H, None, 0, slots[G["EMBOUT_WT"]], slots[G["EMBOUT_B"]], None, JP, 0)
    d1 = node_linear(hout[:nl], J, None, 0, slots[G["ATOM_DEC_W0T"]], slots[G["ATOM_DEC_B0"]], None, 2 * a, 1)
    eh_l = node_linear(d1, 2 * a, None, 0, slots[G["ATOM_DEC_W1T"]], slots[G["ATOM_DEC_B1"]], None, a, 0)
    d2 = node_linear(hout[nl:], J, None, 0, slots[G["RES_DEC_W0T"]], slots[G["RES_DEC_B0"]], None, 2 * r, 1)
    eh_p = node_linear(d2, 2 * r, None, 0, slots[G["RES_DEC_W1T"]], slots[G["RES_DEC_B1"]], None, r, 0)
    vel = x - x_in
    if c.update_pocket_coords:
        mean = torch.zeros(B, 3).index_add_(0, node_batch, vel)
        cnt = torch.zeros(B).index_add_(0, node_batch, torch.ones(N)).clamp(min=1)
        vel = vel - (mean / cnt[:, None])[node_batch]
    return torch.cat([vel[:nl], eh_l], 1), torch.cat([vel[nl:], eh_p], 1), torch.stack([row, col])


# ---------------------------------------------------------------------------
# MFMA lane-mapping emulation (csrc/common.h mfma32 / mfma_row and the k-major
# LDS indexing of node_linear.h / edge_mlp.h)
# ---------------------------------------------------------------------------
def mfma_32x32x2(a_lane, b_lane, acc):
    """a_lane, b_lane: [64] operands; acc: [64][16].  Lane l holds A[i=l&31][k=l>>5],
    B[k=l>>5][j=l&31]; D reg r of lane l = D[(r&3)+8*(r>>2)+4*(l>>5)][l&31]."""
    A = np.zeros((32, 2), np.float64)
    Bm = np.zeros((2, 32), np.float64)
    for l in range(64):
        A[l & 31, l >> 5] = a_lane[l]
        Bm[l >> 5, l & 31] = b_lane[l]
    D = A @ Bm
    out = acc.copy()
    for l in range(64):
        for r in range(16):
            out[l, r] += D[(r & 3) + 8 * (r >> 2) + 4 * (l >> 5), l & 31]
    return out


def emulate_tile_gemm(Atile, Btile, BM, BN_half_tiles, wave_rows, lda):
    """Emulates the kernels' main loop for ONE workgroup tile with 4 waves
    (2 x 2): A tile [BM][K] and B tile [K][BN] are first laid out k-major in
    'LDS' exactly as the kernels do (sA[k*LDA + m], sB[k*BN + n]); every wave
    reads its operands with the kernels' pointer arithmetic.  Returns C [BM][BN]
    reassembled from the accumulator registers via the epilogue's index math."""
    K = Atile.shape[1]
    BN = Btile.shape[1]
    CT = BN_half_tiles           # 32-col tiles per wave
    RT = wave_rows // 32
    sA = np.zeros(K * lda)
    sB = np.zeros(K * BN)
    for k in range(K):
        sA[k * lda:k * lda + BM] = Atile[:, k]
        sB[k * BN:(k + 1) * BN] = Btile[k]
    C = np.full((BM, BN), np.nan)
    for w in range(4):
        wm, wn = w >> 1, w & 1
        acc = [[np.zeros((64, 16)) for _ in range(CT)] for _ in range(RT)]
        lanes = np.arange(64)
        pa = (lanes >> 5) * lda + wm * wave_rows + (lanes & 31)
        pb = (lanes >> 5) * BN + wn * (BN // 2) + (lanes & 31)
        for kk in range(0, K, 2):
            for i in range(RT):
                a = sA[pa + kk * lda + i * 32]
                for j in range(CT):
                    b = sB[pb + kk * BN + j * 32]
                    acc[i][j] = mfma_32x32x2(a, b, acc[i][j])
        for i in range(RT):
            for j in range(CT):
                for l in range(64):
                    col = wn * (BN // 2) + j * 32 + (l & 31)
                    for r in range(16):
                        rowl = wm * wave_rows + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)
                        assert np.isnan(C[rowl, col])
                        C[rowl, col] = acc[i][j][l, r]
    return C


# ---------------------------------------------------------------------------
# aligned edge layout + atomic-free aggregation protocol (csrc/graph.h scan_kernel /
# edges_kernel<true>, csrc/edge_mlp.h): a host model of the index arithmetic
# ---------------------------------------------------------------------------
EDGE_ALIGN = 32


def aligned_layout(deg, node_batch, n_lig, B):
    """row_ptr of the padded layout: the edges of every (sample, node set) segment -- ligand rows of
    sample b, then (after all ligand segments) pocket rows of sample b -- start at a multiple of 32.
    deg [N] edges per row; node numbering [ligand | pocket], samples contiguous in each part.
    Returns (row_ptr [N+1] with row_ptr[N] = padded total, pad_mask [total] True where a slot is padding)."""
    deg = np.asarray(deg, np.int64)
    nb = np.asarray(node_batch, np.int64)
    N = len(deg)
    scan = np.concatenate([[0], np.cumsum(deg)])
    seg_of = np.where(np.arange(N) < n_lig, nb, B + nb)                      # segment id of every node
    seg_len = np.zeros(2 * B, np.int64)
    np.add.at(seg_len, seg_of, deg)
    padded = (seg_len + EDGE_ALIGN - 1) // EDGE_ALIGN * EDGE_ALIGN
    seg_base = np.concatenate([[0], np.cumsum(padded)])
    first = {}
    for i in range(N):
        first.setdefault(int(seg_of[i]), i)
    row_ptr = np.zeros(N + 1, np.int64)
    for i in range(N):
        k = int(seg_of[i])
        row_ptr[i] = seg_base[k] + scan[i] - scan[first[k]]
    row_ptr[N] = seg_base[-1]
    pad = np.ones(int(seg_base[-1]), bool)
    for i in range(N):
        pad[row_ptr[i]:row_ptr[i] + deg[i]] = False
    return row_ptr, pad


def tile_protocol_aggregate(values, erow, row_ptr, deg):
    """The kernels' aggregation: per 32-slot wave tile, segmented sums in slot order; the segment that
    holds a row's first edge goes to agg[row], a first segment that continues the previous tile's row
    goes to head[tile]; completion adds head[T0+1 .. T1] in tile order.  values [slots, F] float32
    (padding slots have erow = -1).  Returns agg [N, F] float32."""
    values = np.asarray(values, np.float32)
    n_slots, F = values.shape
    N = len(deg)
    n_tiles = (n_slots + 31) // 32
    agg = np.zeros((N, F), np.float32)
    head = np.zeros((n_tiles + 1, F), np.float32)
    for T in range(n_tiles):
        lo, hi = 32 * T, min(32 * T + 32, n_slots)
        prev = erow[lo - 1] if lo > 0 else -1
        cur, acc, first_seg = -1, None, True

        def flush():
            nonlocal first_seg
            if cur >= 0:
                if first_seg and cur == prev:
                    head[T] = acc
                else:
                    agg[cur] = acc
                first_seg = False
        for s in range(lo, hi):
            r = int(erow[s])
            if r != cur:
                flush()
                cur, acc = r, np.zeros(F, np.float32)
            if r >= 0:
                acc = (acc + values[s]).astype(np.float32)
        flush()
    out = np.zeros((N, F), np.float32)
    for i in range(N):
        if deg[i] == 0:
            continue
        t0, t1 = row_ptr[i] // 32, (row_ptr[i] + deg[i] - 1) // 32
        v = agg[i].copy()
        for T in range(t0 + 1, t1 + 1):
            v = (v + head[T]).astype(np.float32)
        out[i] = v
    return out


# ---------------------------------------------------------------------------
# Host models of two device-side schedules (index arithmetic only)
# ---------------------------------------------------------------------------
def node_gemm_schedule(M, N, ct, balance, grid_x=None):
    """Which output block every workgroup id of node_gemm_kernel<ct> computes (csrc/node_linear.h, "Tile schedule"):
    list of (row0, col0, n_cols) per id, None for ids that exit.  Rows come in tiles of 128, full tiles are
    32*ct columns wide, the row tiles after the first n_full are computed as two half-width tiles each."""
    bn = 32 * ct
    m_tiles, gy = (M + 127) // 128, N // bn
    n_full = m_tiles
    if ct > 1 and balance > 0:
        n_full = (m_tiles * gy // balance) * balance // gy
    if grid_x is None:
        grid_x = m_tiles * gy * (2 if (ct > 1 and balance > 0) else 1)
    out = []
    for wid in range(grid_x):
        if wid < n_full * gy:
            out.append(((wid // gy) * 128, (wid % gy) * bn, bn))
        elif ct > 1:
            id2 = wid - n_full * gy
            rt = n_full + id2 // (2 * gy)
            out.append(None if rt >= m_tiles else (rt * 128, (id2 % (2 * gy)) * (bn // 2), bn // 2))
        else:
            out.append(None)
    return out, n_full


def stage_plan(n_stages, cone, levels=5):
    """Radius and ghost flag of every message stage of a ligand-output-only call (csrc/engine.hip forward_impl):
    backward cone G - g, with the forward cone min(g + 1, G - g); radii are capped at levels - 1 ("every row"); the
    canonical pocket is evaluated while a later stage still reads rows the previous one did not compute."""
    lv = levels - 1
    radius = [min((min(n_stages - g, g + 1) if cone else n_stages - g), lv) for g in range(n_stages)]
    last = -1
    if cone:
        for g in range(n_stages - 1):
            if min(radius[g + 1] + 1, lv) > radius[g]:
                last = g
    return radius, [int(g <= last) for g in range(n_stages)]


def node_chain_ranges(M, grid, do_mlp, projs, H=256, rows_max=96):
    """Host model of node_chain_kernel's cost-weighted row split (csrc/node_chain.h).  projs: [(N columns, count or
    None, first)].  Returns the list of (r0, tiles) pieces every workgroup walks, as [[(r0, tiles), ...] per workgroup],
    plus (Mw, total cost, V)."""
    w_mlp = 48 if do_mlp else 0
    fst = [min(M, f) for _, _, f in projs]
    cnt = [max(0, min(M - f, c) if c is not None else M - f) for (_, c, _), f in zip(projs, fst)]
    wgt = [n * 16 // H for n, _, _ in projs]
    Mw = M if do_mlp else 0
    for f, c in zip(fst, cnt):
        Mw = max(Mw, f + c)
    if Mw <= 0:
        return [[] for _ in range(grid)], (0, 0, 0)

    def cost_to(r):
        return w_mlp * min(r, Mw) + sum(w * max(0, min(r - f, c)) for w, f, c in zip(wgt, fst, cnt))

    total = cost_to(Mw)
    if total <= 0:
        return [[] for _ in range(grid)], (Mw, 0, 0)
    w_min = cost_to(1)
    for f, c in zip(fst, cnt):
        for r in (f, f + c):
            if r < Mw:
                w_min = min(w_min, cost_to(r + 1) - cost_to(r))
    w_min = max(w_min, 1)
    per_max = rows_max * w_min
    V = -(-total // per_max)
    V = -(-V // grid) * grid

    def row_at(y):
        lo, hi = 0, (Mw + 15) // 16
        while lo < hi:
            mid = (lo + hi) >> 1
            if cost_to(mid * 16) < y:
                lo = mid + 1
            else:
                hi = mid
        return lo * 16

    out = [[] for _ in range(grid)]
    for wg in range(grid):
        v = wg
        while v < V:
            r0 = 0 if v == 0 else row_at((total * v + V - 1) // V)
            r1 = (Mw + 15) // 16 * 16 if v + 1 == V else row_at((total * (v + 1) + V - 1) // V)
            nt, rs = (r1 - r0) // 16, r0
            while nt > 0:
                take = min(nt, rows_max // 16)
                out[wg].append((rs, take))
                rs += 16 * take
                nt -= take
            v += grid
    return out, (Mw, total, V)


def edge_tile_walk(E_a, E_b, grid, split=False):
    """Host model of edge_wave_kernel's static tile assignment (csrc/edge_wave.h): the launch's tile space is the first
    list's 128-edge tiles followed by the second list's; every XCD (blockIdx & 7) owns a contiguous range, walked
    round-robin by its workgroups.  Returns, per workgroup, the list of (list id, local tile, first edge, edges) it
    processes (split: the coordinate stage's one-workgroup-per-(tile, MLP) form; then also the MLP index)."""
    nt_a, nt_b = -(-E_a // 128), -(-E_b // 128)
    ntiles = nt_a + nt_b
    out = []
    for b in range(grid):
        xcd = b & 7
        kx = (b >> 4) if split else (b >> 3)
        gx = (grid >> 4) if split else (grid >> 3)
        qsel = ((b >> 3) & 1) if split else 0
        tq, tr = divmod(ntiles, 8)
        csize = tq + (1 if xcd < tr else 0)
        cbase = xcd * (tq + 1) if xcd < tr else tr * (tq + 1) + (xcd - tr) * tq
        items = []
        li = kx
        while li < csize:
            tile = cbase + li
            lb = tile >= nt_a
            tl = tile - nt_a if lb else tile
            El = E_b if lb else E_a
            e0 = tl * 128
            items.append((int(lb), tl, e0, max(0, min(128, El - e0)), qsel))
            li += gx
        out.append(items)
    return out


def slice_stage_schedule(H, BK=32, threads=256):
    """Host model of the W2^T slice stream of csrc/edge_wave.h: a slice is BK x H floats = BI float4 per thread; MFMA group
    g of a K step requests the float4 indices [BI g / 4, BI (g + 1) / 4) and writes them to LDS one group later (the last
    quarter in front of the barrier).  Returns [(g_load, g_store, float4 index)] for one thread."""
    BI = BK * (H // 4) // threads
    lo = lambda g: BI * g // 4
    ops = []
    for g in range(4):
        for i in range(lo(g), lo(g + 1)):
            ops.append((g, g + 1, i))        # stored at the top of group g + 1; g + 1 == 4: in front of the barrier
    return BI, (BI + 3) // 4, ops
