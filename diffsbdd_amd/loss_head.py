"""The loss terms of the pocket-conditioned training step around the network call on three HIP launches (round 6).

`ConditionalDDPM.forward` in training mode (conditional_model.py:202-330 of the reference) is ~ 250 small torch launches
around the EGNN call: normalisation, centre-of-mass projections, z_t, KL of the prior, likelihood constants, the
categorical L0 term, the squared-error terms, their autograd backward.  `csrc/loss_head.h` evaluates the same formulas
in `dsbdd_loss_cond_pre` (everything that does not read the network's output, one workgroup per sample),
`dsbdd_loss_cond_post` and `dsbdd_loss_cond_post_backward`; the only torch launches left are the two random draws
(t and eps keep their generators: `_draw_t_int`, the keyed `_randn`) and the means of the two logged quantities.

Used when the model is in training mode under autograd with a predefined noise schedule on a GPU (`fused_ok`);
`DSBDD_LOSS=torch` keeps the torch terms (the A/B switch of tests/test_gpu_train.py, and what evaluation, learned
schedules and the joint model use)."""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _lib


def _stream(dev):
    return torch.cuda.current_stream(dev).cuda_stream


def fused_ok(ddpm, dev):
    from .en_diffusion import PredefinedNoiseSchedule
    return (os.environ.get("DSBDD_LOSS", "hip") != "torch" and dev.type == "cuda" and ddpm.training and torch.is_grad_enabled()
            and isinstance(ddpm.gamma, PredefinedNoiseSchedule) and ddpm.n_dims == 3 and ddpm.norm_biases[1] is not None)


class _Post(torch.autograd.Function):
    """(net) -> (error_t, loss_0_x, xh_hat, info_x, info_h); backward: d net."""

    @staticmethod
    def forward(ctx, net, cfg, eps, z_t, lig_h, lm, ps):
        lib = _lib.load()
        dev = net.device
        net_c = net.detach().contiguous()
        B = cfg.batch
        out = torch.empty(lib.dsbdd_loss_out_rows(), B, dtype=torch.float32, device=dev)
        xh_hat = torch.empty_like(net_c)
        _lib.check(lib.dsbdd_loss_cond_post(_stream(dev), C.byref(cfg), net_c.data_ptr(), eps.data_ptr(), z_t.data_ptr(),
                                            lig_h.data_ptr(), lm.data_ptr(), ps.data_ptr(), xh_hat.data_ptr(), out.data_ptr()),
                   "dsbdd_loss_cond_post")
        ctx.cfg, ctx.saved = cfg, (net_c, eps, lig_h, lm, ps)
        ctx.set_materialize_grads(False)
        err, l0x, info_x, info_h = out.unbind(0)
        ctx.mark_non_differentiable(info_x, info_h)
        return err, l0x, xh_hat, info_x, info_h

    @staticmethod
    def backward(ctx, g_err, g_l0x, g_hat, _gx, _gh):
        lib = _lib.load()
        net_c, eps, lig_h, lm, ps = ctx.saved
        dev = net_c.device
        f32 = lambda g: None if g is None else g.to(torch.float32).contiguous()
        g_err, g_l0x, g_hat = f32(g_err), f32(g_l0x), f32(g_hat)
        d_net = torch.empty_like(net_c)
        p = lambda g: g.data_ptr() if g is not None else None
        _lib.check(lib.dsbdd_loss_cond_post_backward(_stream(dev), C.byref(ctx.cfg), net_c.data_ptr(), eps.data_ptr(),
                                                     lig_h.data_ptr(), lm.data_ptr(), ps.data_ptr(), p(g_err), p(g_l0x), p(g_hat),
                                                     d_net.data_ptr()), "dsbdd_loss_cond_post_backward")
        return d_net, None, None, None, None, None, None


def conditional_forward(ddpm, ligand, pocket, return_info=False):
    """`ConditionalDDPM.forward` in training mode; same 12-tuple (+ info) as the torch terms."""
    lib = _lib.load()
    dev = ddpm._hip_device(None)
    ligand, pocket = ddpm._to_device(ligand, dev), ddpm._to_device(pocket, dev)
    f32 = dict(dtype=torch.float32, device=dev)
    lx = ligand['x'].to(**f32).contiguous()
    lh = ligand['one_hot'].to(**f32).contiguous()
    px = pocket['x'].to(**f32).contiguous()
    ph = pocket['one_hot'].to(**f32).contiguous()
    lm = ligand['mask'].to(device=dev, dtype=torch.int64).contiguous()
    pm = pocket['mask'].to(device=dev, dtype=torch.int64).contiguous()
    B = ligand['size'].size(0)
    a, r = lh.shape[1], ph.shape[1]
    # log p(n_lig | n_pocket) as a device table when the size distribution is the package's DistributionNodes; any other
    # object keeps its own log_prob_n1_given_n2 call below
    table_of = getattr(ddpm.size_distribution, "_table", None)
    tab = table_of(0, dev) if table_of is not None else None
    cfg = _lib.LossCfg(batch=B, n_lig=lx.shape[0], n_pocket=px.shape[0], atom_nf=a, residue_nf=r, timesteps=ddpm.T,
                       remove_com=int(bool(ddpm._remove_com)), vnode_idx=-1 if ddpm.vnode_idx is None else int(ddpm.vnode_idx),
                       norm_value_x=float(ddpm.norm_values[0]), norm_value_h=float(ddpm.norm_values[1]),
                       norm_bias_h=float(ddpm.norm_biases[1]), n1_tab=tab.shape[0] if tab is not None else 0,
                       n2_tab=tab.shape[1] if tab is not None else 0)
    # the two random draws keep their generators and their order (t first, then eps: conditional_model.py:217, :238)
    t_int = ddpm._draw_t_int(B, dev)
    eps = ddpm._randn(lm, 3 + a, B)
    gamma_table = ddpm.gamma.gamma.detach().to(**f32).contiguous()
    ps = torch.empty(lib.dsbdd_loss_rows(), B, **f32)
    z_t = torch.empty(lx.shape[0], 3 + a, **f32)
    xh_pocket = torch.empty(px.shape[0], 3 + r, **f32)
    lxn, lhn, pxn, phn = torch.empty_like(lx), torch.empty_like(lh), torch.empty_like(px), torch.empty_like(ph)
    _lib.check(lib.dsbdd_loss_cond_pre(_stream(dev), C.byref(cfg), lx.data_ptr(), lh.data_ptr(), lm.data_ptr(), px.data_ptr(),
                                       ph.data_ptr(), pm.data_ptr(), eps.data_ptr(), t_int.data_ptr(), gamma_table.data_ptr(),
                                       tab.data_ptr() if tab is not None else None, z_t.data_ptr(), xh_pocket.data_ptr(),
                                       ps.data_ptr(), lxn.data_ptr(), lhn.data_ptr(), pxn.data_ptr(), phn.data_ptr()),
               "dsbdd_loss_cond_pre")
    # normalize() works in place on the dictionaries (en_diffusion.py:880-895): the caller finds the normalised batch there
    ligand['x'], ligand['one_hot'], pocket['x'], pocket['one_hot'] = lxn, lhn, pxn, phn
    (t, _g_t, _g_s, _al, _si, snr_w, neg_log_c, kl_prior, l0_h, log_pN, delta_log_px, _tz) = ps.unbind(0)
    net, _ = ddpm.dynamics(z_t, xh_pocket, t.unsqueeze(1), lm, pm)
    error_t, l0_x, xh_hat, info_x, info_h = _Post.apply(net, cfg, eps, z_t, lh, lm, ps)
    if tab is None:
        log_pN = ddpm.log_pN(ligand['size'], pocket['size'])
    zero = torch.tensor(0.0)
    terms = (delta_log_px, error_t, zero, snr_w, l0_x, zero.clone(), l0_h, neg_log_c, kl_prior, log_pN, t_int.squeeze(), xh_hat)
    if not return_info:
        return terms
    return (*terms, {'eps_hat_lig_x': info_x.mean(), 'eps_hat_lig_h': info_h.mean()})
