"""The training step of `EGNNDynamics` as ONE autograd node over ONE launch sequence per direction (round 6).

`loss.backward()` of the reference's training step (lightning_modules.py:337-363 -> conditional_model.py:202-330 /
en_diffusion.py:336-469 -> dynamics.py:87-167) needs `EGNNDynamics.forward` under autograd.  Rounds 4 - 5 composed it
from ~60 `torch.autograd.Function` nodes (train_hip.py) with ~470 small aten launches per step between the HIP kernels;
here `dsbdd_train_net_forward` / `dsbdd_train_net_backward` (csrc/train_net.h) walk the whole network in C++ and PyTorch
keeps exactly one node on its tape: `EGNNTrainFunction`.  PyTorch is plumbing: it owns the parameter / gradient
tensors, the per-call workspace and the stream.

`DSBDD_TRAIN=functions` selects the per-stage Functions of train_hip.py (A/B, and the only path for gradients of
gradients, which neither implements), `DSBDD_TRAIN=torch` round 3's eager path.
"""
from __future__ import annotations

import ctypes as C
import weakref

import torch

from . import _lib
from .engine import make_config
from .train_hip import TrainGraph, _stream


def param_names(hp):
    """The parameter order of include/diffsbdd_hip.h `dsbdd_train_net_*` (names as in EGNNDynamics.state_dict())."""
    names = []

    def lin(n, bias=True):
        names.append(n + ".weight")
        if bias:
            names.append(n + ".bias")
    for n in ("atom_encoder.0", "atom_encoder.2", "atom_decoder.0", "atom_decoder.2",
              "residue_encoder.0", "residue_encoder.2", "residue_decoder.0", "residue_decoder.2"):
        lin(n)
    if hp["edge_embedding_dim"]:
        names.append("edge_embedding.weight")
    lin("egnn.embedding")
    lin("egnn.embedding_out")
    for i in range(hp["n_layers"]):
        for s in range(hp["inv_sublayers"]):
            p = f"egnn.e_block_{i}.gcl_{s}"
            lin(p + ".edge_mlp.0"); lin(p + ".edge_mlp.2"); lin(p + ".node_mlp.0"); lin(p + ".node_mlp.2")
            if hp["attention"]:
                lin(p + ".att_mlp.0")
        p = f"egnn.e_block_{i}.gcl_equiv"
        lin(p + ".coord_mlp.0"); lin(p + ".coord_mlp.2"); lin(p + ".coord_mlp.4", bias=False)
        if not hp["reflection_equivariant"]:
            lin(p + ".cross_product_mlp.0"); lin(p + ".cross_product_mlp.2")
    return names


class _Net:
    """Per-module handle: the C-side descriptor cache and the persistent buffer of re-laid-out weights."""

    def __init__(self, module):
        self.lib = _lib.load()
        cfg = make_config(**module._hp)
        h = C.c_void_p()
        _lib.check(self.lib.dsbdd_train_net_create(C.byref(cfg), C.byref(h)), "dsbdd_train_net_create")
        self.handle = h
        self.names = param_names(module._hp)
        assert self.lib.dsbdd_train_net_param_count(h) == len(self.names)
        self.pack_bytes = int(self.lib.dsbdd_train_net_pack_bytes(h))
        self.pack = None
        self.device = None

    def pack_for(self, dev):
        if self.pack is None or self.device != dev:
            self.pack = torch.empty(self.pack_bytes, dtype=torch.uint8, device=dev)
            self.device = dev
        return self.pack

    def __del__(self):
        try:
            self.lib.dsbdd_train_net_destroy(self.handle)
        except Exception:
            pass


_NETS = weakref.WeakKeyDictionary()     # module -> _Net; kept OUT of the module (deepcopy / pickling of the module stay plain)


def _net_of(module):
    net = _NETS.get(module)
    if net is None:
        net = _NETS[module] = _Net(module)
    return net


def _ptr_table(tensors):
    arr = (C.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = t.data_ptr()
    return arr


class EGNNTrainFunction(torch.autograd.Function):
    """(xh_atoms, xh_residues, *parameters) -> (eps_atoms, eps_residues); forward and backward are one C call each."""

    @staticmethod
    def forward(ctx, module, t, mask_atoms, mask_residues, zero_nan, xh_atoms, xh_residues, *params):
        net = _net_of(module)
        lib = net.lib
        dev = params[0].device
        xl = xh_atoms.detach().to(dev, torch.float32).contiguous()
        xp = xh_residues.detach().to(dev, torch.float32).contiguous()
        nd = module.n_dims
        x = torch.cat((xl[:, :nd], xp[:, :nd]), 0).contiguous()
        tt = t.detach().to(dev, torch.float32).reshape(-1).contiguous()
        g = TrainGraph(module, mask_atoms, mask_residues, x, batch=int(tt.numel()) if tt.numel() > 1 else None)
        ws_bytes = int(lib.dsbdd_train_net_workspace_bytes(net.handle, C.byref(g.c)))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        pack = net.pack_for(dev)
        ps = [p.detach().contiguous() for p in params]
        if any(p.dtype != torch.float32 for p in ps):
            raise _lib.HipLibraryError("the training path computes in float32: parameters must be float32")
        eps_l = torch.empty_like(xl)
        eps_p = torch.empty_like(xp)
        status = torch.zeros(1, dtype=torch.int32, device=dev)
        _lib.check(lib.dsbdd_train_net_forward(
            net.handle, _stream(dev), C.byref(g.c), _ptr_table(ps), pack.data_ptr(), pack.numel(), ws.data_ptr(), ws.numel(),
            xl.data_ptr(), xp.data_ptr(), tt.data_ptr(), tt.numel(), int(bool(zero_nan)), eps_l.data_ptr(), eps_p.data_ptr(),
            status.data_ptr()), "dsbdd_train_net_forward")
        ctx.net, ctx.g, ctx.ws, ctx.pack, ctx.ps = net, g, ws, pack, ps
        ctx.status = status
        ctx.in_grad = (xh_atoms.requires_grad, xh_residues.requires_grad)
        ctx.shapes = (xl.shape, xp.shape)
        ctx.mark_non_differentiable(status)
        ctx.set_materialize_grads(False)       # an unused output arrives as None (see backward), not as a zero tensor
        return eps_l, eps_p, status

    @staticmethod
    def backward(ctx, d_l, d_p, _d_status):
        net, g, ws, pack, ps = ctx.net, ctx.g, ctx.ws, ctx.pack, ctx.ps
        if ws is None:
            raise RuntimeError("EGNNTrainFunction keeps its activations for ONE backward pass (retain_graph / a second "
                               "backward through the same forward is not supported; DSBDD_TRAIN=functions is the per-stage path)")
        lib = net.lib
        dev = ps[0].device
        f32 = dict(dtype=torch.float32, device=dev)
        ctx.none_l, ctx.none_p = d_l is None, d_p is None
        d_l = torch.zeros(ctx.shapes[0], **f32) if d_l is None else d_l.to(**f32).contiguous()
        d_p = torch.zeros(ctx.shapes[1], **f32) if d_p is None else d_p.to(**f32).contiguous()
        grads = [torch.empty_like(p) for p in ps]
        want_l, want_p = ctx.in_grad
        dx_l = torch.empty(ctx.shapes[0], **f32) if (want_l or want_p) else None
        dx_p = torch.empty(ctx.shapes[1], **f32) if (want_l or want_p) else None
        _lib.check(lib.dsbdd_train_net_backward(
            net.handle, _stream(dev), C.byref(g.c), _ptr_table(ps), _ptr_table(grads), pack.data_ptr(), pack.numel(),
            ws.data_ptr(), ws.numel(), int(g.e_lig), d_l.data_ptr(), d_p.data_ptr(),
            dx_l.data_ptr() if dx_l is not None else None, dx_p.data_ptr() if dx_p is not None else None),
            "dsbdd_train_net_backward")
        ctx.ws = None          # the activations are consumed
        # a decoder whose output the loss never read has NO gradient (autograd's None, as for the reference's modules:
        # the pocket-conditioned loss ignores eps_pocket, so residue_decoder stays untouched by the optimiser), not zeros
        for unused, prefix in ((ctx.none_l, "atom_decoder."), (ctx.none_p, "residue_decoder.")):
            if unused:
                for i, n in enumerate(net.names):
                    if n.startswith(prefix):
                        grads[i] = None
        return (None, None, None, None, None, dx_l if want_l else None, dx_p if want_p else None, *grads)


def dynamics_forward_net(m, xh_atoms, xh_residues, t, mask_atoms, mask_residues):
    """`EGNNDynamics.forward` (dynamics.py:87-167) under autograd as one node.  `m`: the EGNNDynamics module."""
    dev = m.egnn.embedding.weight.device
    if dev.type != "cuda":
        raise _lib.HipLibraryError("the training path runs on the GPU only (parameters are on %s); there is no CPU "
                                   "fallback" % dev)
    net = _net_of(m)
    named = dict(m.named_parameters())
    params = [named[n] for n in net.names]
    eps_l, eps_p, status = EGNNTrainFunction.apply(m, t, mask_atoms.to(dev), mask_residues.to(dev), m.training,
                                                   xh_atoms.to(dev), xh_residues.to(dev), *params)
    if not m.training and int(status.item()) & _lib.STATUS_NAN:                       # dynamics.py:155-159
        raise ValueError("NaN detected in EGNN output")
    return eps_l, eps_p
