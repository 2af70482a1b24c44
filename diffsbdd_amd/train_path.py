"""Differentiable evaluation of `EGNNDynamics` for the TRAINING step ("must stay an nn.Module
... and be differentiable ... training calls it under autograd", conditional_model.py:253, en_diffusion.py:378,
lightning_modules.py:337-363).

The sampling hot path -- every call under `torch.no_grad()` / in eval mode -- runs on the fused gfx950 kernels, which
have no backward pass.  When the module is in training mode AND autograd is recording, `EGNNDynamics.forward` comes
here instead: the same function (dynamics.py:87-167, egnn_new.py:7-122,163-184,225-244,296-335) written with
differentiable tensor operations on the GPU, on the module's own parameters, so that `loss.backward()` fills their
`.grad` and an optimiser can step them.  What it shares with the kernels:
  * the radius graph comes from the HIP builder (`dsbdd_build_edges` through `get_edges`): the same edge set, built on
    the device without the reference's O(N^2) distance matrices (graph construction carries no gradient);
  * the first layer of every edge MLP in its factorised form, Linear(cat[h_i, h_j, e]) = P[i] + Q[j] + e W_e^T
    (csrc/edge_mlp.h): two node-level GEMMs and a gather instead of an [E, 2H + A] concatenation -- one third of the
    activation memory the literal graph keeps for the backward pass.
GEMMs are plain library calls (rocBLAS / hipBLASLt through torch); there is no CPU fallback: parameters and inputs must
be on a GPU.  Gradients are checked against the oracle under autograd (tests/test_gpu_parity.py).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import _lib


def _seg_sum(src, index, n):
    out = torch.zeros((n,) + src.shape[1:], dtype=src.dtype, device=src.device)
    return out.index_add_(0, index, src)


def _first_layer(lin, H, h, row, col, edge_attr):
    """Linear(cat[h[row], h[col], edge_attr]) without the concatenation (egnn_new.py:35,99)."""
    w = lin.weight
    p = F.linear(h, w[:, :H])
    q = F.linear(h, w[:, H:2 * H])
    return p[row] + q[col] + F.linear(edge_attr, w[:, 2 * H:], lin.bias)


def _edge_mlp_tail(seq, z):
    """The layers of an edge MLP after its first Linear (children 1 ..)."""
    for k in range(1, len(seq)):          # (by index: children() lists a re-used activation module only once)
        z = seq[k](z)
    return z


def coord2diff(x, row, col, norm_constant):
    """egnn_new.py:296-302."""
    diff = x[row] - x[col]
    radial = (diff ** 2).sum(1, keepdim=True)
    norm = torch.sqrt(radial + 1e-8)
    return radial, diff / (norm + norm_constant)


def coord2cross(x, row, col, batch_mask, n_batch, norm_constant):
    """egnn_new.py:305-316: cross product of the endpoints relative to the sample's mean position."""
    cnt = _seg_sum(torch.ones_like(x[:, :1]), batch_mask, n_batch).clamp(min=1)
    mean = _seg_sum(x, batch_mask, n_batch) / cnt
    a = x[row] - mean[batch_mask[row]]
    b = x[col] - mean[batch_mask[col]]
    cross = torch.cross(a, b, dim=1)
    norm = torch.linalg.norm(cross, dim=1, keepdim=True)
    return cross / (norm + norm_constant)


def dynamics_forward_autograd(m, xh_atoms, xh_residues, t, mask_atoms, mask_residues):
    """`EGNNDynamics.forward` (dynamics.py:87-167) under autograd.  `m`: the EGNNDynamics module."""
    hp = m._hp
    dev = m.egnn.embedding.weight.device
    if dev.type != "cuda":
        raise _lib.HipLibraryError("the training path runs on the GPU only (parameters are on %s); there is no CPU "
                                   "fallback" % dev)
    H, nd = hp["hidden_nf"], m.n_dims
    norm_c, norm_f = float(hp["norm_constant"]), float(hp["normalization_factor"])
    xh_atoms = xh_atoms.to(dev, torch.float32)
    xh_residues = xh_residues.to(dev, torch.float32)
    mask_atoms = mask_atoms.to(dev, torch.int64)
    mask_residues = mask_residues.to(dev, torch.int64)
    n_l = xh_atoms.shape[0]
    x = torch.cat((xh_atoms[:, :nd], xh_residues[:, :nd]), 0)
    h = torch.cat((m.atom_encoder(xh_atoms[:, nd:]), m.residue_encoder(xh_residues[:, nd:])), 0)   # dynamics.py:96-97
    mask = torch.cat((mask_atoms, mask_residues))
    n_batch = int(mask.max().item()) + 1 if mask.numel() else 1
    t = t.to(dev, torch.float32)
    h_time = t.reshape(-1)[:1].expand(h.shape[0], 1) if t.numel() == 1 else t.reshape(-1, 1)[mask]  # :104-111
    h = torch.cat((h, h_time), 1)
    with torch.no_grad():
        edges = m.get_edges(mask_atoms, mask_residues, x[:n_l].detach(), x[n_l:].detach())          # :114, HIP builder
    row, col = edges[0], edges[1]
    if m.edge_embedding is not None:                                                               # :116-127
        rl, cl = row < n_l, col < n_l
        etype = torch.zeros(row.shape[0], dtype=torch.int64, device=dev)
        etype[rl & cl] = 1
        etype[~rl & ~cl] = 2
        edge_extra = m.edge_embedding(etype)
    else:
        edge_extra = None
    upd = None
    if not m.update_pocket_coords:                                                                 # :130-132
        upd = torch.cat((torch.ones(n_l, 1, device=dev), torch.zeros(x.shape[0] - n_l, 1, device=dev)), 0)

    # ---- EGNN (egnn_new.py:225-244) ----
    n = x.shape[0]
    d0, _ = coord2diff(x, row, col, norm_c)
    h = m.egnn.embedding(h)
    x_cur = x
    for i in range(hp["n_layers"]):
        blk = getattr(m.egnn, f"e_block_{i}")
        # EquivariantBlock.forward, egnn_new.py:163-184
        d, u = coord2diff(x_cur, row, col, norm_c)
        cross = None if hp["reflection_equivariant"] else coord2cross(x_cur, row, col, mask, n_batch, norm_c)
        parts = [d, d0] + ([edge_extra] if edge_extra is not None else [])
        ea = torch.cat(parts, 1)
        for s in range(hp["inv_sublayers"]):
            gcl = getattr(blk, f"gcl_{s}")
            mij = _edge_mlp_tail(gcl.edge_mlp, _first_layer(gcl.edge_mlp[0], H, h, row, col, ea))   # :31-46
            if hp["attention"]:
                mij = mij * gcl.att_mlp(mij)
            agg = _seg_sum(mij, row, n) / norm_f                                                   # :48-52,319-335
            h = h + gcl.node_mlp(torch.cat((h, agg), 1))                                           # :53-58
        eq = blk.gcl_equiv                                                                         # :96-122
        phi = _edge_mlp_tail(eq.coord_mlp, _first_layer(eq.coord_mlp[0], H, h, row, col, ea))
        trans = u * torch.tanh(phi) * hp["coords_range"] if hp["tanh"] else u * phi
        if cross is not None:
            phi_x = _edge_mlp_tail(eq.cross_product_mlp, _first_layer(eq.cross_product_mlp[0], H, h, row, col, ea))
            if hp["tanh"]:
                phi_x = torch.tanh(phi_x) * hp["coords_range"]
            trans = trans + cross * phi_x
        aggx = _seg_sum(trans, row, n) / norm_f
        if upd is not None:
            aggx = aggx * upd
        x_cur = x_cur + aggx
    h = m.egnn.embedding_out(h)

    vel = x_cur - x                                                                                # dynamics.py:136
    h = h[:, :-1]                                                                                  # drop the time column
    h_atoms = m.atom_decoder(h[:n_l])
    h_res = m.residue_decoder(h[n_l:])
    if torch.isnan(vel).any():                                                                     # :155-159
        if m.training:
            vel = torch.where(torch.isnan(vel), torch.zeros_like(vel), vel)
        else:
            raise ValueError("NaN detected in EGNN output")
    if m.update_pocket_coords:                                                                     # :161-164
        cnt = _seg_sum(torch.ones_like(vel[:, :1]), mask, n_batch).clamp(min=1)
        vel = vel - (_seg_sum(vel, mask, n_batch) / cnt)[mask]
    return torch.cat((vel[:n_l], h_atoms), 1), torch.cat((vel[n_l:], h_res), 1)
