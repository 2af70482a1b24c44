"""Training step of `EGNNDynamics` on the hand-written gfx950 kernels.

`loss.backward()` of the reference's training step (lightning_modules.py:337-363 -> conditional_model.py:202-330 /
en_diffusion.py:336-469 -> dynamics.py:87-167 -> egnn_new.py:31-58,96-122) needs the denoiser to be differentiable.  Here
`EGNNDynamics.forward` in training mode is a composition of `torch.autograd.Function` objects whose forward AND backward
are HIP kernels of libdiffsbdd_hip.so (csrc/train.h, csrc/edge_wave.h, csrc/node_linear.h):

  * `EdgeGCL`   -- GCL.edge_model + attention + aggregation (egnn_new.py:31-52): forward = the fused message kernel of the
                   sampling path; backward = kernels A / wgrad / B / gather of csrc/train.h.  No [E, H] tensor is kept
                   between forward and backward: the edge activations are recomputed from the per-node projections.
  * `EdgeCoord` -- EquivariantUpdate.coord_model (egnn_new.py:96-122) incl. coord2diff / coord2cross and the masked
                   update of x; backward also returns the coordinate gradient through the geometry.
  * `HipLinear` -- every node-level Linear: forward `dsbdd_node_linear`, backward `dsbdd_node_linear` (dX),
                   `dsbdd_train_wgrad` (dW = dY^T X, ordered split-K) and `dsbdd_train_colsum` (db).
  * `EdgeRadial`, `SampleMean` -- the squared input distances and the per-sample mean position as differentiable inputs.

PyTorch is plumbing here: residual adds, SiLU on [N, H] node tensors, concatenations, weight re-layouts and the
autograd tape.  Every sum over edges / nodes in the backward kernels has a fixed order: gradients are bitwise reproducible
(no atomics).  The A/B switch to the eager torch path of round 3 (train_path.py) is `DSBDD_TRAIN=torch`.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn.functional as F

from . import _lib
from .engine import edge_capacity, make_config


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream(dev):
    return torch.cuda.current_stream(dev).cuda_stream


class TrainGraph:
    """Radius graph of one training batch (dynamics.py:169-187) + what the backward kernels need on top of it."""

    def __init__(self, module, mask_atoms, mask_residues, x, batch=None):
        eng_cfg = make_config(**module._hp)  # dsbdd_config (cut-offs); no inference engine is built for a training step
        lib = _lib.load()
        dev = x.device
        self.lib, self.dev = lib, dev
        ml = mask_atoms.to(device=dev, dtype=torch.int64).contiguous()
        mp = mask_residues.to(device=dev, dtype=torch.int64).contiguous()
        n_l, n_p = ml.numel(), mp.numel()
        N = n_l + n_p
        if batch is None:       # (a training step passes it: one t per sample -- two host syncs less)
            batch = int(max(int(ml.max()) if n_l else 0, int(mp.max()) if n_p else 0)) + 1
        cap = max(edge_capacity(ml, mp, batch), 1)
        i32 = dict(dtype=torch.int32, device=dev)
        self.node_batch = torch.empty(N, **i32)
        self.lig_off, self.poc_off = torch.empty(batch + 1, **i32), torch.empty(batch + 1, **i32)
        self.deg, self.row_ptr = torch.empty(N, **i32), torch.empty(N + 1, **i32)
        erow, ecol = torch.empty(cap, **i32), torch.empty(cap, **i32)
        ed0 = torch.empty(cap, dtype=torch.float32, device=dev)
        status = torch.zeros(1, **i32)
        xc = x.detach().to(torch.float32).contiguous()
        _lib.check(lib.dsbdd_build_edges(_stream(dev), xc.data_ptr(), ml.data_ptr(), mp.data_ptr(), n_l, n_p, batch,
                                         C.byref(eng_cfg), self.node_batch.data_ptr(), self.lig_off.data_ptr(),
                                         self.poc_off.data_ptr(), self.deg.data_ptr(), self.row_ptr.data_ptr(),
                                         erow.data_ptr(), ecol.data_ptr(), ed0.data_ptr(), cap, status.data_ptr()),
                   "dsbdd_build_edges")
        # one host sync per training forward: the edge count and the edge prefix of the ligand rows
        rp = self.row_ptr[n_l::N - n_l].tolist() if N > n_l else [int(self.row_ptr[N].item())] * 2
        E = int(rp[-1])
        self.E, self.e_lig = E, int(rp[0])
        self.erow, self.ecol, self.ed0 = erow[:max(E, 1)], ecol[:max(E, 1)], ed0[:max(E, 1)]
        self.n_lig, self.N, self.batch = n_l, N, batch
        self.rev = torch.empty(max(E, 1), **i32)
        self._cnt = None
        self.c = _lib.TrainGraph(erow=self.erow.data_ptr(), ecol=self.ecol.data_ptr(), ed0=self.ed0.data_ptr(),
                                 row_ptr=self.row_ptr.data_ptr(), deg=self.deg.data_ptr(), rev=self.rev.data_ptr(),
                                 node_batch=self.node_batch.data_ptr(), lig_off=self.lig_off.data_ptr(),
                                 poc_off=self.poc_off.data_ptr(), n_lig=n_l, n_nodes=N, n_edges=E, batch=batch)
        _lib.check(lib.dsbdd_train_edge_rev(_stream(dev), C.byref(self.c), self.rev.data_ptr()), "dsbdd_train_edge_rev")
        self._scratch = {}

    @property
    def cnt(self):
        """nodes per sample (>= 1), for the per-stage Functions' mean backward; the network path does not need it"""
        if self._cnt is None:
            self._cnt = torch.bincount(self.node_batch.long(), minlength=self.batch).clamp(min=1).to(torch.float32)
        return self._cnt

    def scratch(self, H):
        if H not in self._scratch:
            nbytes = self.lib.dsbdd_train_scratch_bytes(H, self.N, self.E)
            self._scratch[H] = torch.empty(nbytes, dtype=torch.uint8, device=self.dev)
        return self._scratch[H]

    def edges(self):
        return torch.stack((self.erow[:self.E].long(), self.ecol[:self.E].long()), 0)


def _pad_cols(mat):
    """[K][N] -> contiguous [K][round_up(N, 4)] (the B operand of dsbdd_node_linear needs ld % 4 == 0)."""
    K, N = mat.shape
    if N % 4 == 0:
        return mat.contiguous()
    out = torch.zeros(K, (N + 3) // 4 * 4, dtype=mat.dtype, device=mat.device)
    out[:, :N] = mat
    return out


def _node_linear(x, wt, bias, n_out):
    """x [M][K] @ wt [K][ld >= n_out] (+ bias) -> [M][n_out] on dsbdd_node_linear."""
    lib = _lib.load()
    M, K = x.shape
    out = torch.empty(M, n_out, dtype=torch.float32, device=x.device)
    if M == 0:
        return out
    _lib.check(lib.dsbdd_node_linear(_stream(x.device), x.data_ptr(), x.stride(0), K, None, 0, 0, wt.data_ptr(),
                                     wt.stride(0), _ptr(bias), None, 0, out.data_ptr(), n_out, M, n_out, 0),
               "dsbdd_node_linear")
    return out


class HipLinear(torch.autograd.Function):
    """y = x W^T + b with W in nn.Linear layout [out][in]."""

    @staticmethod
    def forward(ctx, x, W, b):
        x = x.contiguous()
        W = W.contiguous()
        ctx.save_for_backward(x, W)
        ctx.has_bias = b is not None
        return _node_linear(x, _pad_cols(W.t()), b.contiguous() if b is not None else None, W.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x, W = ctx.saved_tensors
        lib = _lib.load()
        dy = dy.contiguous()
        M, n_out = dy.shape
        n_in = W.shape[1]
        dev = dy.device
        dx = dW = db = None
        if ctx.needs_input_grad[0]:
            dx = _node_linear(dy, _pad_cols(W), None, n_in)
        if ctx.needs_input_grad[1]:
            dW = torch.empty(n_out, n_in, dtype=torch.float32, device=dev)
            if M == 0:
                dW.zero_()
            else:
                nb = lib.dsbdd_train_wgrad_scratch_bytes(M, n_out, n_in)
                scr = torch.empty(nb, dtype=torch.uint8, device=dev)
                _lib.check(lib.dsbdd_train_wgrad(_stream(dev), dy.data_ptr(), n_out, x.data_ptr(), n_in, M, n_out, n_in,
                                                 dW.data_ptr(), scr.data_ptr(), nb), "dsbdd_train_wgrad")
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = torch.empty(n_out, dtype=torch.float32, device=dev)
            if M == 0:
                db.zero_()
            else:
                nb = 4 * ((M + 31) // 32) * n_out
                scr = torch.empty(nb, dtype=torch.uint8, device=dev)
                _lib.check(lib.dsbdd_train_colsum(_stream(dev), dy.data_ptr(), n_out, M, n_out, db.data_ptr(),
                                                  scr.data_ptr(), nb), "dsbdd_train_colsum")
        return dx, dW, db


class EdgeRadial(torch.autograd.Function):
    """ed0[e] = |x_row - x_col|^2 of the call's input coordinates (egnn_new.py:233, 296-299): the values are the edge
    builder's, the backward is the HIP scatter kernel."""

    @staticmethod
    def forward(ctx, x, g):
        ctx.g = g
        ctx.save_for_backward(x)
        return g.ed0[:g.E].clone()

    @staticmethod
    def backward(ctx, gd):
        (x,) = ctx.saved_tensors
        g = ctx.g
        dx = torch.empty_like(x)
        gdc = gd.contiguous()
        if g.E == 0:
            return dx.zero_(), None
        _lib.check(g.lib.dsbdd_train_radial_backward(_stream(x.device), C.byref(g.c), x.data_ptr(), gdc.data_ptr(),
                                                     dx.data_ptr()), "dsbdd_train_radial_backward")
        return dx, None


class SampleMean(torch.autograd.Function):
    """Mean position of all nodes of every sample (coord2cross, egnn_new.py:307-310)."""

    @staticmethod
    def forward(ctx, x, g):
        ctx.g = g
        mean = torch.empty(g.batch, 3, dtype=torch.float32, device=x.device)
        _lib.check(g.lib.dsbdd_train_sample_mean(_stream(x.device), x.data_ptr(), C.byref(g.c), mean.data_ptr()),
                   "dsbdd_train_sample_mean")
        return mean

    @staticmethod
    def backward(ctx, dmean):
        g = ctx.g
        nb = g.node_batch.long()
        return (dmean / g.cnt[:, None])[nb], None


def _mlp_struct(P, Q, ld, wd, wd0, tab, W2, W2T, b2, head, head_b):
    return _lib.TrainMlp(P=P.data_ptr(), Q=Q.data_ptr(), ldpq=ld, wd=wd.data_ptr(), wd0=wd0.data_ptr(),
                         tab=tab.data_ptr(), W2=W2.data_ptr(), W2T=W2T.data_ptr(), b2=b2.data_ptr(),
                         head=head.data_ptr() if head is not None else None,
                         head_b=head_b.data_ptr() if head_b is not None else None)


class EdgeGCL(torch.autograd.Function):
    """agg = segment_sum(m * att) / nf over the rows (egnn_new.py:31-52) from the first-layer projections pq = [P | Q]."""

    @staticmethod
    def forward(ctx, pq, x, ed0, wd, wd0, tab, W2, b2, att_w, att_b, g, norm_factor):
        H = W2.shape[0]
        pq, x = pq.contiguous(), x.contiguous()
        ed0c = g.ed0                                     # (same values as `ed0`; the graph's buffer has the list's length)
        wd, wd0, tab, W2, b2 = wd.contiguous(), wd0.contiguous(), tab.contiguous(), W2.contiguous(), b2.contiguous()
        W2T = W2.t().contiguous()
        aw = att_w.contiguous().reshape(-1) if att_w is not None else None
        ab = att_b.contiguous().reshape(-1) if att_b is not None else None
        agg = torch.empty(g.N, H, dtype=torch.float32, device=x.device)
        scr = g.scratch(H)
        m = _mlp_struct(pq, pq[:, H:], 2 * H, wd, wd0, tab, W2, W2T, b2, aw, ab)
        _lib.check(g.lib.dsbdd_train_gcl_forward(_stream(x.device), H, C.byref(g.c), C.byref(m), x.data_ptr(),
                                                 float(norm_factor), agg.data_ptr(), scr.data_ptr(), scr.numel()),
                   "dsbdd_train_gcl_forward")
        ctx.g, ctx.norm_factor, ctx.H = g, float(norm_factor), H
        ctx.has_att = aw is not None
        ctx.save_for_backward(pq, x, wd, wd0, tab, W2, W2T, b2, aw if aw is not None else b2, ab if ab is not None else b2)
        del ed0c
        return agg

    @staticmethod
    def backward(ctx, d_agg):
        pq, x, wd, wd0, tab, W2, W2T, b2, aw, ab = ctx.saved_tensors
        g, H = ctx.g, ctx.H
        dev = x.device
        if not ctx.has_att:
            aw = ab = None
        d_agg = d_agg.contiguous()
        d_pq = torch.empty(g.N, 2 * H, dtype=torch.float32, device=dev)
        d_vec = torch.empty(8, H, dtype=torch.float32, device=dev)
        d_W2 = torch.empty(H, H, dtype=torch.float32, device=dev)
        gd0 = torch.zeros(max(g.E, 1), dtype=torch.float32, device=dev)
        d_x = torch.empty(g.N, 3, dtype=torch.float32, device=dev)
        scr = g.scratch(H)
        m = _mlp_struct(pq, pq[:, H:], 2 * H, wd, wd0, tab, W2, W2T, b2, aw, ab)
        out = _lib.TrainMlpGrad(dP=d_pq.data_ptr(), dQ=d_pq[:, H:].data_ptr(), ldo=2 * H, d_vec=d_vec.data_ptr(),
                                d_W2=d_W2.data_ptr(), gd0=gd0.data_ptr())
        _lib.check(g.lib.dsbdd_train_gcl_backward(_stream(dev), H, C.byref(g.c), C.byref(m), x.data_ptr(),
                                                  ctx.norm_factor, d_agg.data_ptr(), C.byref(out), d_x.data_ptr(),
                                                  scr.data_ptr(), scr.numel()), "dsbdd_train_gcl_backward")
        d_aw = d_vec[6].reshape(1, H) if ctx.has_att else None
        d_ab = d_vec[7, :1].clone() if ctx.has_att else None
        return (d_pq, d_x, gd0[:g.E], d_vec[0], d_vec[1], d_vec[2:5], d_W2, d_vec[5], d_aw, d_ab, None, None)


class EdgeCoord(torch.autograd.Function):
    """x_out = x + mask * segment_sum(u T(phi) + cross T(phi_x)) / nf (egnn_new.py:96-122, 296-316).
    pq = [P_coord | Q_coord (| P_cross | Q_cross)]; the MLP parameter groups follow; w3 = the shared output layer."""

    @staticmethod
    def forward(ctx, pq, x, mean, ed0, wd_c, wd0_c, tab_c, W2_c, b2_c, wd_x, wd0_x, tab_x, W2_x, b2_x, w3, g, cfg):
        H = W2_c.shape[0]
        n_mlp = 1 if W2_x is None else 2
        pq, x = pq.contiguous(), x.contiguous()
        ld = 2 * H * n_mlp
        w3v = w3.contiguous().reshape(-1)
        groups = []
        for q, (wd, wd0, tab, W2, b2) in enumerate(((wd_c, wd0_c, tab_c, W2_c, b2_c), (wd_x, wd0_x, tab_x, W2_x, b2_x))):
            if q >= n_mlp:
                break
            W2 = W2.contiguous()
            groups.append((wd.contiguous(), wd0.contiguous(), tab.contiguous(), W2, W2.t().contiguous(), b2.contiguous()))
        mean_c = mean.contiguous() if mean is not None else None
        arr = (_lib.TrainMlp * n_mlp)()
        for q, (wd, wd0, tab, W2, W2T, b2) in enumerate(groups):
            arr[q] = _mlp_struct(pq[:, 2 * H * q:], pq[:, 2 * H * q + H:], ld, wd, wd0, tab, W2, W2T, b2, w3v, None)
        n_upd = g.N if cfg["update_pocket_coords"] else g.n_lig
        x_out = torch.empty_like(x)
        scr = g.scratch(H)
        _lib.check(g.lib.dsbdd_train_coord_forward(_stream(x.device), H, C.byref(g.c), arr, n_mlp, x.data_ptr(),
                                                   _ptr(mean_c), n_upd, float(cfg["norm_constant"]),
                                                   float(cfg["coords_range"]), int(bool(cfg["tanh"])),
                                                   float(cfg["normalization_factor"]), x_out.data_ptr(), scr.data_ptr(),
                                                   scr.numel()), "dsbdd_train_coord_forward")
        ctx.g, ctx.cfg, ctx.H, ctx.n_mlp, ctx.n_upd = g, cfg, H, n_mlp, n_upd
        flat = [t for grp in groups for t in grp]
        ctx.save_for_backward(pq, x, mean_c if mean_c is not None else x, w3v, *flat)
        return x_out

    @staticmethod
    def backward(ctx, d_xout):
        pq, x, mean_c, w3v, *flat = ctx.saved_tensors
        g, cfg, H, n_mlp, n_upd = ctx.g, ctx.cfg, ctx.H, ctx.n_mlp, ctx.n_upd
        dev = x.device
        ld = 2 * H * n_mlp
        d_xout = d_xout.contiguous()
        e_upd = g.E if n_upd == g.N else g.e_lig
        arr = (_lib.TrainMlp * n_mlp)()
        outs = (_lib.TrainMlpGrad * n_mlp)()
        d_pq = torch.empty(g.N, ld, dtype=torch.float32, device=dev)
        d_vec = torch.empty(n_mlp, 8, H, dtype=torch.float32, device=dev)
        d_W2 = torch.empty(n_mlp, H, H, dtype=torch.float32, device=dev)
        gd0 = torch.zeros(n_mlp, max(g.E, 1), dtype=torch.float32, device=dev)
        d_x = torch.zeros(g.N, 3, dtype=torch.float32, device=dev)
        d_mean = torch.zeros(g.batch, 3, dtype=torch.float32, device=dev) if n_mlp == 2 else None
        for q in range(n_mlp):
            wd, wd0, tab, W2, W2T, b2 = flat[6 * q:6 * q + 6]
            arr[q] = _mlp_struct(pq[:, 2 * H * q:], pq[:, 2 * H * q + H:], ld, wd, wd0, tab, W2, W2T, b2, w3v, None)
            outs[q] = _lib.TrainMlpGrad(dP=d_pq[:, 2 * H * q:].data_ptr(), dQ=d_pq[:, 2 * H * q + H:].data_ptr(), ldo=ld,
                                        d_vec=d_vec[q].data_ptr(), d_W2=d_W2[q].data_ptr(), gd0=gd0[q].data_ptr())
        scr = g.scratch(H)
        if e_upd > 0 and n_upd > 0:
            _lib.check(g.lib.dsbdd_train_coord_backward(
                _stream(dev), H, C.byref(g.c), arr, n_mlp, x.data_ptr(), _ptr(mean_c if n_mlp == 2 else None), n_upd, e_upd,
                float(cfg["norm_constant"]), float(cfg["coords_range"]), int(bool(cfg["tanh"])),
                float(cfg["normalization_factor"]), d_xout.data_ptr(), outs, d_x.data_ptr(), _ptr(d_mean), scr.data_ptr(),
                scr.numel()), "dsbdd_train_coord_backward")
        else:
            d_pq.zero_(); d_vec.zero_(); d_W2.zero_()
        d_x = d_x + d_xout                                   # the identity path x -> x_out
        d_w3 = d_vec[:, 6].sum(0).reshape(1, H)
        gd0_tot = gd0.sum(0)[:g.E]
        gq = []
        for q in range(2):
            if q < n_mlp:
                gq += [d_vec[q, 0], d_vec[q, 1], d_vec[q, 2:5], d_W2[q], d_vec[q, 5]]
            else:
                gq += [None] * 5
        return (d_pq, d_x, d_mean, gd0_tot, *gq, d_w3, None, None)


class EdgeFirstLayer(torch.autograd.Function):
    """First Linear of an edge MLP (egnn_new.py:35,99) in the layouts the edge kernels read:
    weight [H][2H + 2 (+ emb)], bias [H], edge-embedding table [3][emb] (or None) ->
    W_pq [2H][H] (the per-node projections P | Q), wd, wd0 [H] (the two distance columns), tab [3][H] (bias + embedded
    edge type).  One autograd node instead of six slicing / concatenation nodes per MLP: their backward was a zero fill
    and a copy of the whole weight each (a third of the step's small launches)."""

    @staticmethod
    def forward(ctx, w, bias, emb_w):
        H = w.shape[0]
        ctx.H = H
        ctx.save_for_backward(w, emb_w if emb_w is not None else bias)
        ctx.has_emb = emb_w is not None
        w_pq = torch.cat((w[:, :H], w[:, H:2 * H]), 0)
        wd, wd0 = w[:, 2 * H].contiguous(), w[:, 2 * H + 1].contiguous()
        if emb_w is not None:
            tab = torch.addmm(bias[None, :].expand(emb_w.shape[0], H), emb_w, w[:, 2 * H + 2:].t())
        else:
            tab = bias[None, :].expand(3, H).contiguous()
        return w_pq, wd, wd0, tab

    @staticmethod
    def backward(ctx, d_wpq, d_wd, d_wd0, d_tab):
        w, emb_w = ctx.saved_tensors
        H = ctx.H
        z = lambda *shape: torch.zeros(*shape, dtype=w.dtype, device=w.device)
        # the weight gradient as ONE concatenation (five slice assignments were five launches per MLP and step)
        parts = [d_wpq[:H] if d_wpq is not None else z(H, H), d_wpq[H:] if d_wpq is not None else z(H, H),
                 (d_wd if d_wd is not None else z(H))[:, None], (d_wd0 if d_wd0 is not None else z(H))[:, None]]
        d_bias = d_emb = None
        if d_tab is not None:
            d_bias = d_tab.sum(0)
            if ctx.has_emb:
                parts.append(d_tab.t() @ emb_w)
                d_emb = d_tab @ w[:, 2 * H + 2:]
        else:
            if ctx.has_emb:
                parts.append(z(H, w.shape[1] - 2 * H - 2))
            if ctx.needs_input_grad[1]:
                d_bias = z(H)
        return torch.cat(parts, 1), d_bias, d_emb


def _wgrad(dy, x):
    """dW [n_out][n_in] = dy^T x on dsbdd_train_wgrad (ordered split-K: bitwise reproducible)."""
    lib = _lib.load()
    M, n_out = dy.shape
    n_in = x.shape[1]
    dW = torch.empty(n_out, n_in, dtype=torch.float32, device=dy.device)
    if M == 0:
        return dW.zero_()
    nb = lib.dsbdd_train_wgrad_scratch_bytes(M, n_out, n_in)
    scr = torch.empty(nb, dtype=torch.uint8, device=dy.device)
    _lib.check(lib.dsbdd_train_wgrad(_stream(dy.device), dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0), M, n_out,
                                     n_in, dW.data_ptr(), scr.data_ptr(), nb), "dsbdd_train_wgrad")
    return dW


def _colsum(dy):
    lib = _lib.load()
    M, n_out = dy.shape
    db = torch.empty(n_out, dtype=torch.float32, device=dy.device)
    if M == 0:
        return db.zero_()
    nb = 4 * ((M + 31) // 32) * n_out
    scr = torch.empty(nb, dtype=torch.uint8, device=dy.device)
    _lib.check(lib.dsbdd_train_colsum(_stream(dy.device), dy.data_ptr(), dy.stride(0), M, n_out, db.data_ptr(),
                                      scr.data_ptr(), nb), "dsbdd_train_colsum")
    return db


class NodeMLP(torch.autograd.Function):
    """GCL.node_model (egnn_new.py:53-58): h + Linear2(SiLU(Linear1([h | agg]))) as one autograd node: the concatenation is
    the two-operand form of dsbdd_node_linear, the residual its R operand; backward = two dX GEMMs, two ordered
    weight-gradient GEMMs, two column sums and aten's fused SiLU backward."""

    @staticmethod
    def forward(ctx, h, agg, W1, b1, W2, b2):
        lib = _lib.load()
        h, agg = h.contiguous(), agg.contiguous()
        W1, W2 = W1.contiguous(), W2.contiguous()
        M, H = h.shape
        Ka = agg.shape[1]
        n_hid, n_out = W1.shape[0], W2.shape[0]
        dev = h.device
        z = torch.empty(M, n_hid, dtype=torch.float32, device=dev)
        out = torch.empty(M, n_out, dtype=torch.float32, device=dev)
        if M:
            w1t = _pad_cols(W1.t())
            _lib.check(lib.dsbdd_node_linear(_stream(dev), h.data_ptr(), h.stride(0), H, agg.data_ptr(), agg.stride(0), Ka,
                                             w1t.data_ptr(), w1t.stride(0), _ptr(b1.contiguous()), None, 0, z.data_ptr(),
                                             n_hid, M, n_hid, 0), "dsbdd_node_linear")
            a = F.silu(z)
            w2t = _pad_cols(W2.t())
            _lib.check(lib.dsbdd_node_linear(_stream(dev), a.data_ptr(), a.stride(0), n_hid, None, 0, 0, w2t.data_ptr(),
                                             w2t.stride(0), _ptr(b2.contiguous()), h.data_ptr(), h.stride(0),
                                             out.data_ptr(), n_out, M, n_out, 0), "dsbdd_node_linear")
        else:
            a = z
        ctx.save_for_backward(h, agg, z, a, W1, W2)
        return out

    @staticmethod
    def backward(ctx, d_out):
        h, agg, z, a, W1, W2 = ctx.saved_tensors
        d_out = d_out.contiguous()
        H = h.shape[1]
        da = _node_linear(d_out, _pad_cols(W2), None, W2.shape[1])
        dW2, db2 = _wgrad(d_out, a), _colsum(d_out)
        dz = torch.ops.aten.silu_backward(da, z)
        d_cat = _node_linear(dz, _pad_cols(W1), None, W1.shape[1])
        xcat = torch.cat((h, agg), 1)
        dW1, db1 = _wgrad(dz, xcat), _colsum(dz)
        return d_out + d_cat[:, :H], d_cat[:, H:], dW1, db1, dW2, db2


def _lin(layer, x):
    return HipLinear.apply(x, layer.weight, layer.bias)


def _mlp2(seq, x):
    """Sequential(Linear, SiLU, Linear): encoders / decoders (dynamics.py:27-49)."""
    return _lin(seq[2], F.silu(_lin(seq[0], x)))


def _edge_params(first, H, emb):
    """First layer of an edge MLP (egnn_new.py:35,99) -> (W_pq [2H][H], wd, wd0, tab [3][H])."""
    return EdgeFirstLayer.apply(first.weight, first.bias, emb.weight if emb is not None else None)


def dynamics_forward_hip(m, xh_atoms, xh_residues, t, mask_atoms, mask_residues):
    """`EGNNDynamics.forward` (dynamics.py:87-167) under autograd, on the HIP kernels.  `m`: the EGNNDynamics module."""
    hp = m._hp
    dev = m.egnn.embedding.weight.device
    if dev.type != "cuda":
        raise _lib.HipLibraryError("the training path runs on the GPU only (parameters are on %s); there is no CPU "
                                   "fallback" % dev)
    H, nd = hp["hidden_nf"], m.n_dims
    xh_atoms = xh_atoms.to(dev, torch.float32)
    xh_residues = xh_residues.to(dev, torch.float32)
    mask_atoms = mask_atoms.to(dev, torch.int64)
    mask_residues = mask_residues.to(dev, torch.int64)
    n_l = xh_atoms.shape[0]
    x = torch.cat((xh_atoms[:, :nd], xh_residues[:, :nd]), 0).contiguous()
    g = TrainGraph(m, mask_atoms, mask_residues, x, batch=int(t.numel()) if t.numel() > 1 else None)
    h = torch.cat((_mlp2(m.atom_encoder, xh_atoms[:, nd:]), _mlp2(m.residue_encoder, xh_residues[:, nd:])), 0)   # :96-97
    mask = torch.cat((mask_atoms, mask_residues))
    t = t.to(dev, torch.float32)
    h_time = t.reshape(-1)[:1].expand(h.shape[0], 1) if t.numel() == 1 else t.reshape(-1, 1)[mask]   # :104-111
    h = torch.cat((h, h_time), 1)
    ed0 = EdgeRadial.apply(x, g) if x.requires_grad else g.ed0[:g.E]
    emb = m.edge_embedding
    nf = float(hp["normalization_factor"])

    # ---- EGNN (egnn_new.py:225-244) ----
    h = _lin(m.egnn.embedding, h)
    x_cur = x
    for i in range(hp["n_layers"]):
        blk = getattr(m.egnn, f"e_block_{i}")
        mean = None if hp["reflection_equivariant"] else SampleMean.apply(x_cur, g)
        for s in range(hp["inv_sublayers"]):
            gcl = getattr(blk, f"gcl_{s}")
            w_pq, wd, wd0, tab = _edge_params(gcl.edge_mlp[0], H, emb)
            pq = HipLinear.apply(h, w_pq, None)
            if hp["attention"]:
                aw, ab = gcl.att_mlp[0].weight, gcl.att_mlp[0].bias
            else:
                aw = ab = None
            agg = EdgeGCL.apply(pq, x_cur, ed0, wd, wd0, tab, gcl.edge_mlp[2].weight, gcl.edge_mlp[2].bias, aw, ab, g, nf)
            h = NodeMLP.apply(h, agg, gcl.node_mlp[0].weight, gcl.node_mlp[0].bias, gcl.node_mlp[2].weight,
                              gcl.node_mlp[2].bias)                                                    # :53-58
        eq = blk.gcl_equiv                                                                             # :96-122
        wc_pq, wd_c, wd0_c, tab_c = _edge_params(eq.coord_mlp[0], H, emb)
        if eq.cross_product_mlp is not None:
            wx_pq, wd_x, wd0_x, tab_x = _edge_params(eq.cross_product_mlp[0], H, emb)
            pq4 = HipLinear.apply(h, torch.cat((wc_pq, wx_pq), 0), None)
            x_cur = EdgeCoord.apply(pq4, x_cur, mean, ed0, wd_c, wd0_c, tab_c, eq.coord_mlp[2].weight, eq.coord_mlp[2].bias,
                                    wd_x, wd0_x, tab_x, eq.cross_product_mlp[2].weight, eq.cross_product_mlp[2].bias,
                                    eq.coord_mlp[4].weight, g, hp)
        else:
            pq2 = HipLinear.apply(h, wc_pq, None)
            x_cur = EdgeCoord.apply(pq2, x_cur, None, ed0, wd_c, wd0_c, tab_c, eq.coord_mlp[2].weight, eq.coord_mlp[2].bias,
                                    None, None, None, None, None, eq.coord_mlp[4].weight, g, hp)
    h = _lin(m.egnn.embedding_out, h)

    vel = x_cur - x                                                                                # dynamics.py:136
    h = h[:, :-1]                                                                                  # drop the time column
    h_atoms = _mlp2(m.atom_decoder, h[:n_l])
    h_res = _mlp2(m.residue_decoder, h[n_l:])
    if m.training:                                                                                 # :155-159
        vel = torch.where(torch.isnan(vel), torch.zeros_like(vel), vel)     # (unconditionally: no host sync per step)
    elif torch.isnan(vel).any():
        raise ValueError("NaN detected in EGNN output")
    if m.update_pocket_coords:                                                                     # :161-164
        vel = vel - SampleMean.apply(vel.contiguous(), g)[g.node_batch.long()]
    return torch.cat((vel[:n_l], h_atoms), 1), torch.cat((vel[n_l:], h_res), 1)
