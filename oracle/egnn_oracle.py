"""ORACLE (test infrastructure, NOT product code).

CPU restatement (torch, fp32 or fp64) of the reference's EGNN denoiser
`EGNNDynamics.forward` -- the literal reference graph (concatenate, full
Linear, scatter in edge order), *not* the factorised algorithm the HIP kernels
use, so that it is an independent check of the kernels' algebra.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may
import this module.  Parity pinning: validated against the real reference
(imported through oracle/ref_shim.py in the build container) by
tests/test_oracle_vs_reference.py, and against the committed golden vectors in
tests/golden/*.npz (generated from the real reference by
tests/golden/make_golden.py) by tests/test_oracle_golden.py.

All functions take the model as a flat `state_dict`-style mapping
{name: tensor} with the reference's key names and a `cfg`
dict of the EGNNDynamics constructor kwargs.  Citations are
/root/reference/<path>:<line>.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


# ---------------------------------------------------------------------------
# small helpers
# ---------------------------------------------------------------------------
def _linear(sd, prefix, x):
    """nn.Linear with weight [out,in] (+ optional bias)."""
    w = sd[prefix + ".weight"]
    b = sd.get(prefix + ".bias")
    return F.linear(x, w, b)


def _mlp2(sd, prefix, x):
    """Sequential(Linear, SiLU, Linear): dynamics.py:27-49."""
    return _linear(sd, prefix + ".2", F.silu(_linear(sd, prefix + ".0", x)))


def segment_sum(data, seg, num_segments):
    """egnn_new.py:319-329 -- scatter_add_ over dim 0, *sequential in edge
    order* on CPU."""
    out = data.new_zeros((num_segments, data.shape[1]))
    out.index_add_(0, seg, data)
    return out


def segment_mean(data, seg, num_segments):
    """egnn_new.py:331-335 (aggregation 'mean': divide by count, 0 -> 1)."""
    s = segment_sum(data, seg, num_segments)
    cnt = segment_sum(torch.ones_like(data), seg, num_segments)
    cnt[cnt == 0] = 1
    return s / cnt


# ---------------------------------------------------------------------------
# geometry: egnn_new.py:296-316
# ---------------------------------------------------------------------------
def coord2diff(x, row, col, norm_constant=1.0):
    """egnn_new.py:296-302.  radial = |xi-xj|^2 ; u = (xi-xj)/(sqrt(radial+1e-8)+nc)."""
    diff = x[row] - x[col]
    radial = (diff * diff).sum(1, keepdim=True)
    norm = torch.sqrt(radial + 1e-8)
    return radial, diff / (norm + norm_constant)


def coord2cross(x, row, col, batch_mask, norm_constant=1.0):
    """egnn_new.py:305-316.  Cross product of the two end points relative to
    the per-sample mean over ALL nodes of the sample."""
    nseg = int(batch_mask.max()) + 1
    mean = segment_mean(x, batch_mask, nseg)
    a = x[row] - mean[batch_mask[row]]
    b = x[col] - mean[batch_mask[col]]
    cr = torch.linalg.cross(a, b, dim=1)
    nrm = torch.linalg.norm(cr, dim=1, keepdim=True)
    return cr / (nrm + norm_constant)


# ---------------------------------------------------------------------------
# edge construction: dynamics.py:169-187
# ---------------------------------------------------------------------------
def pair_dist(a, b, exact=True):
    """Distance matrix.  exact=False reproduces torch.cdist's default (the
    matmul formulation for > 25 points, error up to ~1e-2 for un-centred
    coordinates); exact=True is sqrt(sum (a-b)^2)."""
    if not exact:
        return torch.cdist(a, b)
    d = a[:, None, :] - b[None, :, :]
    return torch.sqrt((d * d).sum(-1))


def get_edges_blockwise(mask_l, mask_p, x_l, x_p, cut_l, cut_p, cut_i):
    """dynamics.py:169-187 evaluated sample by sample with exact distances.

    The reference builds one dense [N, N] adjacency whose same-sample test
    (dynamics.py:170-172) makes it block diagonal; here only the diagonal blocks
    are formed (O(sum n_b^2) memory instead of O(N^2): 1.5 GB -> 25 MB at the
    benchmark batch).  Same edge set and the same (row, col) order as
    get_edges(..., exact=True); node numbering = [ligand nodes | pocket nodes]."""
    nl = len(mask_l)
    B = int(max(mask_l.max() if len(mask_l) else 0, mask_p.max() if len(mask_p) else 0)) + 1
    rows, cols = [], []
    for b in range(B):
        il = torch.nonzero(mask_l == b).view(-1)
        ip = torch.nonzero(mask_p == b).view(-1)
        xl, xp = x_l[il], x_p[ip]
        a_l = torch.ones(len(il), len(il), dtype=torch.bool)
        a_p = torch.ones(len(ip), len(ip), dtype=torch.bool)
        a_c = torch.ones(len(il), len(ip), dtype=torch.bool)
        if cut_l is not None:
            a_l &= pair_dist(xl, xl) <= cut_l
        if cut_p is not None:
            a_p &= pair_dist(xp, xp) <= cut_p
        if cut_i is not None:
            a_c &= pair_dist(xl, xp) <= cut_i
        adj = torch.cat((torch.cat((a_l, a_c), 1), torch.cat((a_c.T, a_p), 1)), 0)
        ids = torch.cat((il, ip + nl))
        r, c = torch.where(adj)
        rows.append(ids[r]); cols.append(ids[c])
    row, col = torch.cat(rows), torch.cat(cols)
    order = torch.argsort(row * (nl + len(mask_p)) + col)
    return torch.stack((row[order], col[order]), 0)


def get_edges(mask_l, mask_p, x_l, x_p, cut_l, cut_p, cut_i, exact=True):
    """dynamics.py:169-187.  Returns int64 [2,E], sorted by (row, col), self
    loops kept; node numbering = [ligand nodes | pocket nodes]."""
    if exact and len(mask_l) + len(mask_p) > 4096:
        return get_edges_blockwise(mask_l, mask_p, x_l, x_p, cut_l, cut_p, cut_i)
    adj_l = mask_l[:, None] == mask_l[None, :]
    adj_p = mask_p[:, None] == mask_p[None, :]
    adj_c = mask_l[:, None] == mask_p[None, :]
    if cut_l is not None:
        adj_l = adj_l & (pair_dist(x_l, x_l, exact) <= cut_l)
    if cut_p is not None:
        adj_p = adj_p & (pair_dist(x_p, x_p, exact) <= cut_p)
    if cut_i is not None:
        adj_c = adj_c & (pair_dist(x_l, x_p, exact) <= cut_i)
    adj = torch.cat((torch.cat((adj_l, adj_c), 1), torch.cat((adj_c.T, adj_p), 1)), 0)
    return torch.stack(torch.where(adj), 0)


def edge_ambiguity_band(mask_l, mask_p, x_l, x_p, cut_l, cut_p, cut_i, tol):
    """Pairs whose distance is within `tol` of the applicable cutoff: the edge
    builder under test may legitimately differ from the reference on these
    (torch.cdist's matmul formulation is not exact)."""
    x = torch.cat((x_l, x_p)).double()
    m = torch.cat((mask_l, mask_p))
    nl = len(mask_l)
    d = pair_dist(x, x)
    cut = torch.full_like(d, float("inf"))
    if cut_l is not None:
        cut[:nl, :nl] = cut_l
    if cut_p is not None:
        cut[nl:, nl:] = cut_p
    if cut_i is not None:
        cut[:nl, nl:] = cut_i
        cut[nl:, :nl] = cut_i
    same = m[:, None] == m[None, :]
    return same & ((d - cut).abs() <= tol)


# ---------------------------------------------------------------------------
# EGNN layers
# ---------------------------------------------------------------------------
def gcl(sd, p, h, row, col, edge_attr, cfg):
    """GCL.forward, egnn_new.py:31-66 (edge_model -> node_model)."""
    inp = torch.cat([h[row], h[col], edge_attr], 1)                     # :35
    m = F.silu(_linear(sd, p + ".edge_mlp.2",
                       F.silu(_linear(sd, p + ".edge_mlp.0", inp))))    # :15-19,36
    if cfg["attention"]:
        att = torch.sigmoid(_linear(sd, p + ".att_mlp.0", m))           # :26-29,39
        out = m * att                                                   # :40
    else:
        out = m
    agg = segment_sum(out, row, h.shape[0])                             # :50
    if cfg["aggregation_method"] == "sum":
        agg = agg / cfg["normalization_factor"]                         # :328-329
    else:
        cnt = segment_sum(torch.ones_like(out), row, h.shape[0])
        cnt[cnt == 0] = 1
        agg = agg / cnt
    nin = torch.cat([h, agg], 1)                                        # :56
    return h + _linear(sd, p + ".node_mlp.2",
                       F.silu(_linear(sd, p + ".node_mlp.0", nin)))     # :21-24,57


def equivariant_update(sd, p, h, x, row, col, coord_diff, coord_cross, edge_attr,
                       update_coords_mask, cfg):
    """EquivariantUpdate.coord_model, egnn_new.py:96-122.  NB coords_range that
    reaches this layer is the *un-divided* value (15.0): egnn_new.py:218 passes
    `coords_range`, not `coords_range_layer`."""
    crange = float(cfg.get("coords_range", 15.0))
    inp = torch.cat([h[row], h[col], edge_attr], 1)                     # :99

    def scalar_mlp(q):
        t = F.silu(_linear(sd, q + ".0", inp))
        t = F.silu(_linear(sd, q + ".2", t))
        return F.linear(t, sd[q + ".4.weight"])                         # :78 (no bias)

    phi = scalar_mlp(p + ".coord_mlp")
    if cfg["tanh"]:
        trans = coord_diff * torch.tanh(phi) * crange                   # :101
    else:
        trans = coord_diff * phi                                        # :103
    if not cfg["reflection_equivariant"]:
        phx = scalar_mlp(p + ".cross_product_mlp")                      # :106
        if cfg["tanh"]:
            phx = torch.tanh(phx) * crange                              # :108
        trans = trans + coord_cross * phx                               # :109
    agg = segment_sum(trans, row, x.shape[0])                           # :114
    if cfg["aggregation_method"] == "sum":
        agg = agg / cfg["normalization_factor"]
    else:
        cnt = segment_sum(torch.ones_like(trans), row, x.shape[0])
        cnt[cnt == 0] = 1
        agg = agg / cnt
    if update_coords_mask is not None:
        agg = update_coords_mask * agg                                  # :118-119
    return x + agg                                                      # :121


def egnn_forward(sd, h, x, row, col, update_coords_mask, batch_mask, edge_attr, cfg,
                 trace=None):
    """EGNN.forward egnn_new.py:225-244 + EquivariantBlock.forward :163-184."""
    p = "egnn"
    nc = float(cfg["norm_constant"])
    d0, _ = coord2diff(x, row, col)                                     # :228
    edge_feat = d0 if edge_attr is None else torch.cat([d0, edge_attr], 1)  # :231-232
    h = _linear(sd, p + ".embedding", h)                                # :233
    for i in range(cfg["n_layers"]):
        bp = f"{p}.e_block_{i}"
        dist, cdiff = coord2diff(x, row, col, nc)                       # :166
        ccross = None if cfg["reflection_equivariant"] else \
            coord2cross(x, row, col, batch_mask, nc)                    # :167-171
        ea = torch.cat([dist, edge_feat], 1)                            # :174
        for s in range(cfg["inv_sublayers"]):
            h = gcl(sd, f"{bp}.gcl_{s}", h, row, col, ea, cfg)          # :175-177
        x = equivariant_update(sd, bp + ".gcl_equiv", h, x, row, col, cdiff, ccross,
                               ea, update_coords_mask, cfg)             # :178-179
        if trace is not None:
            trace.append((h.clone(), x.clone()))
    h = _linear(sd, p + ".embedding_out", h)                            # :241
    return h, x


def remove_mean_batch(x, idx):
    """en_diffusion.py:918-922."""
    nseg = int(idx.max()) + 1
    return x - segment_mean(x, idx, nseg)[idx]


def dynamics_forward(sd, cfg, xh_atoms, xh_residues, t, mask_atoms, mask_residues,
                     edges=None, exact_dist=True, trace=None):
    """EGNNDynamics.forward, dynamics.py:87-167 (mode 'egnn_dynamics',
    condition_time=True).  `edges` ([2,E] int64) teacher-forces the edge list;
    None builds it as dynamics.py:169-187 does.  Returns (eps_atoms,
    eps_residues, edges)."""
    nd = cfg.get("n_dims", 3)
    x_a, h_a = xh_atoms[:, :nd], xh_atoms[:, nd:]
    x_r, h_r = xh_residues[:, :nd], xh_residues[:, nd:]
    h_a = _mlp2(sd, "atom_encoder", h_a)                                # :96
    h_r = _mlp2(sd, "residue_encoder", h_r)                             # :97
    x = torch.cat((x_a, x_r), 0)                                        # :100
    h = torch.cat((h_a, h_r), 0)
    mask = torch.cat([mask_atoms, mask_residues])
    nl = len(mask_atoms)
    if t.numel() == 1:                                                  # :105-107
        h_time = torch.full_like(h[:, :1], float(t.reshape(-1)[0]))
    else:
        h_time = t[mask]                                                # :110
    h = torch.cat([h, h_time], 1)

    if edges is None:
        edges = get_edges(mask_atoms, mask_residues, x_a, x_r,
                          cfg["edge_cutoff_ligand"], cfg["edge_cutoff_pocket"],
                          cfg["edge_cutoff_interaction"], exact=exact_dist)
    row, col = edges[0], edges[1]
    assert torch.all(mask[row] == mask[col])                            # :115

    if cfg.get("edge_embedding_dim"):                                   # :118-127
        et = torch.zeros(edges.shape[1], dtype=torch.long)
        et[(row < nl) & (col < nl)] = 1
        et[(row >= nl) & (col >= nl)] = 2
        edge_types = sd["edge_embedding.weight"][et]
    else:
        edge_types = None

    if cfg["update_pocket_coords"]:
        ucm = None
    else:                                                               # :130-132
        ucm = torch.cat((torch.ones(nl, 1, dtype=x.dtype),
                         torch.zeros(len(mask_residues), 1, dtype=x.dtype)))
    h_fin, x_fin = egnn_forward(sd, h, x, row, col, ucm, mask, edge_types, cfg, trace)
    vel = x_fin - x                                                     # :136
    h_fin = h_fin[:, :-1]                                               # :149
    h_fa = _mlp2(sd, "atom_decoder", h_fin[:nl])                        # :152
    h_fr = _mlp2(sd, "residue_decoder", h_fin[nl:])                     # :153
    if torch.any(torch.isnan(vel)):                                     # :155-159
        raise ValueError("NaN detected in EGNN output")
    if cfg["update_pocket_coords"]:
        vel = remove_mean_batch(vel, mask)                              # :161-164
    return (torch.cat([vel[:nl], h_fa], -1),
            torch.cat([vel[nl:], h_fr], -1), edges)                     # :166-167


# ---------------------------------------------------------------------------
# work / traffic model used by bench.py
# ---------------------------------------------------------------------------
def flop_model(cfg, N, N_l, N_p, E, E_u, a, r):
    """Returns (F_ref, F_min) in MAC per dynamics call, SURVEY.md §8(d)."""
    H, J, L = cfg["hidden_nf"], cfg["joint_nf"], cfg["n_layers"]
    A = 2 + (cfg.get("edge_embedding_dim") or 0)
    node_terms = 2 * N * (J + 1) * H + N_l * 2 * (2 * a * a + 2 * a * J) \
        + N_p * 2 * (2 * r * r + 2 * r * J)
    f_ref = L * (E * 3 * ((2 * H + A) * H + H * H + H) + N * 3 * H * H) + node_terms
    f_min = L * (E * (H * H + (A + 2) * H) + E_u * 2 * (H * H + (A + 1) * H)
                 + N * (2 * H * H + 3 * H * H) + N * 4 * H * H) + node_terms
    return f_ref, f_min
