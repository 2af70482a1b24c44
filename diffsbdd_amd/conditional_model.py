"""`ConditionalDDPM` / `SimpleConditionalDDPM` -- drop-in for the sampling API of
/root/reference/equivariant_diffusion/conditional_model.py:12-746 (pocket fixed,
ligand diffused): `sample_given_pocket`, `inpaint`, `diversify`,
`sample_p_zs_given_zt`, `sample_p_zt_given_zs`, `sample_p_xh_given_z0`.

The reverse loop (conditional_model.py:518-526) is the benchmark's hot path:
per step one `EGNNDynamics.forward_async` (gfx950 kernels), one noise draw and
one fused update kernel (`dsbdd_cond_reverse_update`: posterior mean, noise,
ligand-COM removal from ligand and pocket), all enqueued on one stream with no
host synchronisation.
"""
from __future__ import annotations

import torch
import ctypes as C

import torch.nn.functional as F

from . import _lib
from .en_diffusion import EnVariationalDiffusion, _chain, num_nodes_to_batch_mask, seg_mean, seg_sum

__all__ = ["ConditionalDDPM", "SimpleConditionalDDPM"]


class ConditionalDDPM(EnVariationalDiffusion):
    """Conditional diffusion module."""

    _remove_com = 1   # SimpleConditionalDDPM switches the COM projection off

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        assert not self.dynamics.update_pocket_coords

    # ---- COM handling (conditional_model.py:688-696) ------------------------------
    @classmethod
    def remove_mean_batch(cls, x_lig, x_pocket, lig_indices, pocket_indices):
        """Subtract the LIGAND centre of mass from ligand and pocket."""
        n = int(max(lig_indices.max(), pocket_indices.max())) + 1
        mean = seg_mean(x_lig, lig_indices, n)
        return x_lig - mean[lig_indices], x_pocket - mean[pocket_indices]

    def _remove_lig_com(self, x_lig, x_pocket, lig_mask, pocket_mask, batch):
        if not self._remove_com:
            return x_lig, x_pocket
        mean = seg_mean(x_lig, lig_mask, batch)
        return x_lig - mean[lig_mask], x_pocket - mean[pocket_mask]

    # ---- Gaussian draws around a mean (conditional_model.py:140-160) ------------------
    def sample_normal(self, *args):
        raise NotImplementedError("Has been replaced by sample_normal_zero_com()")

    def sample_normal_zero_com(self, mu_lig, xh0_pocket, sigma, lig_mask, pocket_mask, fix_noise=False):
        if fix_noise:
            raise NotImplementedError("fix_noise option isn't implemented yet")
        batch = sigma.shape[0]
        nd = self.n_dims
        eps = self._randn(lig_mask, nd + self.atom_nf, batch)
        out = mu_lig + sigma[lig_mask] * eps
        xl, xp = self._remove_lig_com(out[:, :nd], xh0_pocket[:, :nd], lig_mask, pocket_mask, batch)
        return (torch.cat([xl, out[:, nd:]], dim=1).contiguous(),
                torch.cat([xp, xh0_pocket[:, nd:]], dim=1).contiguous())

    def _cond_gauss_(self, z_lig, xh_pocket, lig_mask, pocket_mask, batch, a, sigma, noise=None):
        """In place (csrc/ddpm.h cond_affine_noise_kernel): z_lig <- a z_lig + sigma eps, then the ligand
        COM is removed from ligand and pocket.  The sampling loops use this instead of the tensor
        formulas below: one launch, fixed summation order."""
        if lig_mask.numel() == 0:
            return                                  # no ligand atoms: nothing to draw, no centre of mass to remove
        if noise is None:
            noise = self._randn(lig_mask, self.n_dims + self.atom_nf, batch)
        _lib.check(_lib.load().dsbdd_cond_affine_noise(
            self._cs(z_lig), z_lig.data_ptr(), xh_pocket.data_ptr(), noise.data_ptr(), lig_mask.data_ptr(),
            pocket_mask.data_ptr(), lig_mask.numel(), pocket_mask.numel(), batch, self.atom_nf, self.residue_nf,
            float(a), float(sigma), self._remove_com), "dsbdd_cond_affine_noise")

    def noised_representation(self, xh_lig, xh0_pocket, lig_mask, pocket_mask, gamma_t):
        batch = gamma_t.shape[0]
        nd = self.n_dims
        alpha_t, sigma_t = self.alpha(gamma_t, xh_lig), self.sigma(gamma_t, xh_lig)
        eps = self._randn(lig_mask, nd + self.atom_nf, batch)
        z = alpha_t[lig_mask] * xh_lig + sigma_t[lig_mask] * eps
        xl, xp = self._remove_lig_com(z[:, :nd], xh0_pocket[:, :nd], lig_mask, pocket_mask, batch)
        return (torch.cat([xl, z[:, nd:]], dim=1).contiguous(),
                torch.cat([xp, xh0_pocket[:, nd:]], dim=1).contiguous(), eps)

    def sample_combined_position_feature_noise(self, lig_indices, xh0_pocket, pocket_indices):
        raise NotImplementedError("Use sample_normal_zero_com() instead.")

    def sample(self, *args):
        raise NotImplementedError("Conditional model does not support sampling without given pocket.")

    # ---- one reverse step (conditional_model.py:432-464) ------------------------------------
    fused_step = True    # keyed noise: one launch per reverse step (csrc/ddpm.h cond_step_keyed_kernel); False: the separate
                         # randn / update / repaint launches (bitwise the same results: tests/test_gpu_parity.py)

    def _cond_step(self, s, co, z_lig, xh_pocket, lig_mask, pocket_mask, batch, status, repaint=None, t_next=None):
        """In place: z_lig (level s+1 -> s) and the pocket translation.  `repaint` = (zk_tmp, xh0_lig, com_pocket_0,
        fixed_f, resample): the RePaint iteration behind the step (conditional_model.py:600-660).  `t_next`: the time of
        the chain's next denoiser call (written by the fused kernel, so that the call needs no fill launch)."""
        eps, _, _ = self._dyn(z_lig, xh_pocket, co.t_value[s + 1], lig_mask, pocket_mask, batch, status,
                              False)
        lib = _lib.load()
        if self.fused_step and self.noise_source is None and lig_mask.numel() > 0:
            if self._seed is None:
                self._seed = int(torch.randint(0, 2 ** 62, (1,)).item())
            ids = getattr(self, "_sample_ids", None)
            if ids is not None:
                if ids.numel() != batch:
                    raise ValueError(f"seed(sample_ids=...) has {ids.numel()} entries for a batch of {batch}")
                if ids.device != lig_mask.device:
                    ids = self._sample_ids = ids.to(lig_mask.device).contiguous()
            zk, xh0, com0, fixed_f, resample = repaint if repaint is not None else (None, None, None, None, False)
            mode = 0 if repaint is None else (2 if resample else 1)
            t_word = getattr(self, "_last_t", None) if t_next is not None else None
            _lib.check(lib.dsbdd_cond_step_keyed(
                self._cs(z_lig), z_lig.data_ptr(), xh_pocket.data_ptr(), eps.data_ptr(),
                zk.data_ptr() if zk is not None else None, xh0.data_ptr() if xh0 is not None else None,
                com0.data_ptr() if com0 is not None else None, fixed_f.data_ptr() if fixed_f is not None else None,
                lig_mask.data_ptr(), pocket_mask.data_ptr(), lig_mask.numel(), pocket_mask.numel(), batch, self.atom_nf,
                self.residue_nf, float(co.alpha_ts[s]), float(co.c_eps[s]), float(co.sigma[s]), mode,
                float(co.alpha[s]), float(co.sigma_t[s]), float(co.sigma_ts[s]), self._remove_com,
                C.c_uint64(self._seed & (2 ** 64 - 1)), C.c_uint64(self._draw), self._sample_offset,
                ids.data_ptr() if ids is not None else None,
                t_word.data_ptr() if t_word is not None else None, float(t_next) if t_next is not None else 0.0),
                "dsbdd_cond_step_keyed")
            self._draw += 1 + (0 if repaint is None else (2 if resample else 1))
            if t_word is not None:
                self._t_prefilled = (t_word.data_ptr(), float(t_next))
            return True
        noise = self._randn(lig_mask, self.n_dims + self.atom_nf, batch)
        _lib.check(lib.dsbdd_cond_reverse_update(
            torch.cuda.current_stream(z_lig.device).cuda_stream, z_lig.data_ptr(), xh_pocket.data_ptr(),
            eps.data_ptr(), noise.data_ptr(), lig_mask.data_ptr(), pocket_mask.data_ptr(),
            lig_mask.numel(), pocket_mask.numel(), batch, self.atom_nf, self.residue_nf,
            float(co.alpha_ts[s]), float(co.c_eps[s]), float(co.sigma[s]), self._remove_com),
            "dsbdd_cond_reverse_update")
        return False

    def _step_impl(self, s_int, co, z_l, z_p, lig_mask, pocket_mask, batch, status):
        self._cond_step(s_int, co, z_l, z_p, lig_mask, pocket_mask, batch, status)

    def sample_p_zt_given_zs(self, zs_lig, xh0_pocket, ligand_mask, pocket_mask, gamma_t, gamma_s,
                             fix_noise=False):
        _, sigma_ts, alpha_ts = self.sigma_and_alpha_t_given_s(gamma_t, gamma_s, zs_lig)
        mu = alpha_ts[ligand_mask] * zs_lig
        return self.sample_normal_zero_com(mu, xh0_pocket, sigma_ts, ligand_mask, pocket_mask, fix_noise)

    # ---- p(x, h | z_0) (conditional_model.py:112-135) ---------------------------------------------
    def sample_p_xh_given_z0(self, z0_lig, xh0_pocket, lig_mask, pocket_mask, batch_size, fix_noise=False,
                             _in_chain=False):
        if not _in_chain:
            lig_mask, pocket_mask = self._begin_chain(lig_mask, pocket_mask, batch_size)
        dev = z0_lig.device
        nd = self.n_dims
        t0 = torch.zeros((batch_size, 1), device=dev)
        gamma_0 = self.gamma(t0)
        sigma_x = self.SNR(-0.5 * gamma_0)
        status = torch.zeros(1, dtype=torch.int32, device=dev)
        try:
            net, _, _ = self._dyn(z0_lig.contiguous(), xh0_pocket.contiguous(), 0.0, lig_mask, pocket_mask,
                                  batch_size, status, False)
            self._check_status(status)
        finally:
            if not _in_chain:
                self._end_chain()
        if fix_noise:
            raise NotImplementedError("fix_noise option isn't implemented yet")
        xh_lig = self.compute_x_pred(net, z0_lig, gamma_0, lig_mask).contiguous()
        xh_pocket = xh0_pocket.clone().contiguous()
        # sample_normal_zero_com(mu, ., sigma_x) in one kernel; one t (one sigma_x) for the whole batch
        self._cond_gauss_(xh_lig, xh_pocket, lig_mask, pocket_mask, batch_size, 1.0, float(sigma_x.reshape(-1)[0]))
        x_lig, h_lig = self.unnormalize(xh_lig[:, :nd], z0_lig[:, nd:])
        x_pocket, h_pocket = self.unnormalize(xh_pocket[:, :nd], xh_pocket[:, nd:])
        h_lig = F.one_hot(torch.argmax(h_lig, dim=1), self.atom_nf)
        return x_lig, h_lig, x_pocket, h_pocket

    # ---- sampling (conditional_model.py:478-555) -----------------------------------------------------
    def _prepare_pocket(self, pocket, dev):
        for k in ('x', 'one_hot', 'size', 'mask'):
            pocket[k] = pocket[k].to(dev)
        pocket['mask'] = pocket['mask'].to(torch.int64).contiguous()
        return pocket

    @torch.no_grad()
    @_chain
    def sample_given_pocket(self, pocket, num_nodes_lig, return_frames=1, timesteps=None):
        timesteps = self.T if timesteps is None else timesteps
        assert 0 < return_frames <= timesteps
        assert timesteps % return_frames == 0
        dev = self._hip_device(None)
        pocket = self._prepare_pocket(pocket, dev)
        n = len(pocket['size'])
        nd = self.n_dims
        _, pocket = self.normalize(pocket=pocket)
        pm = pocket['mask']
        xh0_pocket = torch.cat([pocket['x'], pocket['one_hot']], dim=1)
        lig_mask = num_nodes_to_batch_mask(n, num_nodes_lig, dev).contiguous()
        lig_mask, pm = self._begin_chain(lig_mask, pm, n, pocket=pocket)

        # z_T ~ N(pocket COM, I), then ligand-COM-free (conditional_model.py:501-508)
        mu_x = self._seg_mean3(pocket['x'], pm, n)
        z_lig = torch.cat((mu_x, torch.zeros((n, self.atom_nf), device=dev)), dim=1)[lig_mask].contiguous()
        xh_pocket = xh0_pocket.clone().contiguous()
        self._cond_gauss_(z_lig, xh_pocket, lig_mask, pm, n, 1.0, 1.0)

        out_lig = torch.zeros((return_frames,) + z_lig.size(), device=dev)
        out_pocket = torch.zeros((return_frames,) + xh_pocket.size(), device=dev)
        status = torch.zeros(1, dtype=torch.int32, device=dev)
        co = self._coefs(timesteps)
        for s in reversed(range(0, timesteps)):
            self._cond_step(s, co, z_lig, xh_pocket, lig_mask, pm, n, status, t_next=co.t_value[s])
            if (s * return_frames) % timesteps == 0:
                idx = (s * return_frames) // timesteps
                out_lig[idx], out_pocket[idx] = self.unnormalize_z(z_lig, xh_pocket)
        self._check_status(status)
        if self._remove_com:
            self.assert_mean_zero_with_mask(z_lig[:, :nd], lig_mask)

        x_lig, h_lig, x_pocket, h_pocket = self.sample_p_xh_given_z0(z_lig, xh_pocket, lig_mask, pm, n,
                                                                     _in_chain=True)
        if self._remove_com:
            self.assert_mean_zero_with_mask(x_lig, lig_mask)
        if return_frames == 1:                                              # :541-547
            max_cog = seg_sum(x_lig, lig_mask, n).abs().max().item()
            if max_cog > 5e-2:
                print(f'Warning CoG drift with error {max_cog:.3f}. Projecting the positions down.')
                x_lig, x_pocket = self.remove_mean_batch(x_lig, x_pocket, lig_mask, pm)
        out_lig[0] = torch.cat([x_lig, h_lig], dim=1)
        out_pocket[0] = torch.cat([x_pocket, h_pocket], dim=1)
        return out_lig.squeeze(0), out_pocket.squeeze(0), lig_mask, pm

    def _inpaint_iteration(self, s, co, z_lig, xh_pocket, zk_tmp, xh0_lig, com_pocket_0, fixed_f, lm, pm, n, status,
                           resample):
        """One (s, u) iteration of the RePaint loop (conditional_model.py:600-660), in place on z_lig / xh_pocket
        (stable pointers: the engine replays its graph): the unknown part takes one reverse step (which moves the
        pocket with the ligand COM), then ONE kernel noises the known part to level s around the moved pocket,
        aligns the COM of the fixed atoms, blends, and (between resamplings) applies q(z_t | z_s)."""
        if self._cond_step(s, co, z_lig, xh_pocket, lm, pm, n, status,
                           repaint=(zk_tmp, xh0_lig, com_pocket_0, fixed_f, resample),
                           t_next=co.t_value[s + 1] if resample else co.t_value[s]):
            return                                          # (one fused launch did the step and the iteration)
        dl = self.n_dims + self.atom_nf
        n1 = self._randn(lm, dl, n)
        n2 = self._randn(lm, dl, n) if resample else None
        _lib.check(_lib.load().dsbdd_cond_repaint_update(
            self._cs(z_lig), z_lig.data_ptr(), xh_pocket.data_ptr(), zk_tmp.data_ptr(), xh0_lig.data_ptr(),
            com_pocket_0.data_ptr(), fixed_f.data_ptr(), n1.data_ptr(), n2.data_ptr() if resample else None,
            lm.data_ptr(), pm.data_ptr(), lm.numel(), pm.numel(), n, self.atom_nf, self.residue_nf,
            float(co.alpha[s]), float(co.sigma_t[s]), float(co.alpha_ts[s]), float(co.sigma_ts[s]),
            int(resample), self._remove_com), "dsbdd_cond_repaint_update")

    # ---- RePaint-style inpainting (conditional_model.py:557-686) ---------------------------------------
    @torch.no_grad()
    @_chain
    def inpaint(self, ligand, pocket, lig_fixed, resamplings=1, return_frames=1, timesteps=None,
                center='ligand'):
        timesteps = self.T if timesteps is None else timesteps
        assert 0 < return_frames <= timesteps
        assert timesteps % return_frames == 0
        if len(lig_fixed.size()) == 1:
            lig_fixed = lig_fixed.unsqueeze(1)
        dev = self._hip_device(None)
        pocket = self._prepare_pocket(pocket, dev)
        ligand = self._prepare_pocket(ligand, dev)
        fixed_f = lig_fixed.to(dev).float().reshape(-1).contiguous()
        n = len(ligand['size'])
        ligand, pocket = self.normalize(ligand, pocket)
        lm, pm = self._begin_chain(ligand['mask'], pocket['mask'], n, pocket=pocket)

        xh0_pocket = torch.cat([pocket['x'], pocket['one_hot']], dim=1)
        com_pocket_0 = self._seg_mean3(pocket['x'], pm, n)
        xh0_lig = torch.cat([ligand['x'], ligand['one_hot']], dim=1).contiguous()
        if center == 'ligand':      # COM of the fixed atoms (row ids found once: one host sync per chain)
            sel = fixed_f != 0
            mean_known = self._seg_mean3(ligand['x'][sel], lm[sel].contiguous(), n)
        elif center == 'pocket':
            mean_known = com_pocket_0
        else:
            raise NotImplementedError(f"Centering option {center} not implemented")
        z_lig = torch.cat((mean_known, torch.zeros((n, self.atom_nf), device=dev)), dim=1)[lm].contiguous()
        xh_pocket = xh0_pocket.clone().contiguous()
        self._cond_gauss_(z_lig, xh_pocket, lm, pm, n, 1.0, 1.0)
        zk_tmp = torch.empty_like(z_lig)                                   # kernel scratch

        out_lig = torch.zeros((return_frames,) + z_lig.size(), device=dev)
        out_pocket = torch.zeros((return_frames,) + xh_pocket.size(), device=dev)
        status = torch.zeros(1, dtype=torch.int32, device=dev)
        co = self._coefs(timesteps)
        for s in reversed(range(0, timesteps)):
            for u in range(resamplings):
                self._inpaint_iteration(s, co, z_lig, xh_pocket, zk_tmp, xh0_lig, com_pocket_0, fixed_f, lm, pm, n,
                                        status, resample=u < resamplings - 1)
                if u == resamplings - 1 and (s * return_frames) % timesteps == 0:
                    idx = (s * return_frames) // timesteps
                    out_lig[idx], out_pocket[idx] = self.unnormalize_z(z_lig, xh_pocket)
        self._check_status(status)
        x_lig, h_lig, x_pocket, h_pocket = self.sample_p_xh_given_z0(z_lig, xh_pocket, lm, pm, n, _in_chain=True)
        out_lig[0] = torch.cat([x_lig, h_lig], dim=1)
        out_pocket[0] = torch.cat([x_pocket, h_pocket], dim=1)
        return out_lig.squeeze(0), out_pocket.squeeze(0), lm, pm

    # ---- diversification (conditional_model.py:332-409) ---------------------------------------------------
    def partially_noised_ligand(self, ligand, pocket, noising_steps):
        n = ligand['size'].size(0)
        nd = self.n_dims
        dev = ligand['x'].device
        t = torch.ones((n, 1), device=dev).float() * noising_steps / self.T
        gamma_t = self.inflate_batch_array(self.gamma(t), ligand['x'])
        xh0_lig = torch.cat([ligand['x'], ligand['one_hot']], dim=1)
        xh0_pocket = torch.cat([pocket['x'], pocket['one_hot']], dim=1)
        xl, xp = self._remove_lig_com(xh0_lig[:, :nd], xh0_pocket[:, :nd], ligand['mask'], pocket['mask'], n)
        xh0_lig = torch.cat([xl, xh0_lig[:, nd:]], dim=1)
        xh0_pocket = torch.cat([xp, xh0_pocket[:, nd:]], dim=1)
        return self.noised_representation(xh0_lig, xh0_pocket, ligand['mask'], pocket['mask'], gamma_t)

    @torch.no_grad()
    @_chain
    def diversify(self, ligand, pocket, noising_steps):
        dev = self._hip_device(None)
        pocket = self._prepare_pocket(pocket, dev)
        ligand = self._prepare_pocket(ligand, dev)
        ligand, pocket = self.normalize(ligand, pocket)
        n = len(pocket['size'])
        lm, pm = self._begin_chain(ligand['mask'], pocket['mask'], n, pocket=pocket)
        ligand['mask'], pocket['mask'] = lm, pm
        # partially_noised_ligand (conditional_model.py:332-362) with the fused kernels: centre at the
        # ligand COM (a = 1, sigma = 0), then q(z_t | x) at t = noising_steps / T
        z_lig = torch.cat([ligand['x'], ligand['one_hot']], dim=1).contiguous()
        xh_pocket = torch.cat([pocket['x'], pocket['one_hot']], dim=1).contiguous()
        self._cond_gauss_(z_lig, xh_pocket, lm, pm, n, 1.0, 0.0, noise=z_lig)
        g_t = self.gamma.gamma[int(noising_steps)].detach().cpu()     # gamma(noising_steps / T)
        self._cond_gauss_(z_lig, xh_pocket, lm, pm, n, float(torch.sqrt(torch.sigmoid(-g_t))),
                          float(torch.sqrt(torch.sigmoid(g_t))))
        status = torch.zeros(1, dtype=torch.int32, device=dev)
        co = self._coefs(self.T)
        for s in reversed(range(0, noising_steps)):
            self._cond_step(s, co, z_lig, xh_pocket, lm, pm, n, status, t_next=co.t_value[s])
        self._check_status(status)
        x_lig, h_lig, x_pocket, h_pocket = self.sample_p_xh_given_z0(z_lig, xh_pocket, lm, pm, n, _in_chain=True)
        if self._remove_com:
            self.assert_mean_zero_with_mask(x_lig, lm)
        return torch.cat([x_lig, h_lig], dim=1), torch.cat([x_pocket, h_pocket], dim=1), lm, pm

    # ---- loss terms (conditional_model.py:36-110, 202-330); see EnVariationalDiffusion.forward ---------------
    def kl_prior(self, xh_lig, mask_lig, num_nodes):
        """KL(q(z_T | x) || N(0, 1)) over the ligand, conditional_model.py:36-74."""
        nd = self.n_dims
        ones = torch.ones((len(num_nodes), 1), device=xh_lig.device)
        gamma_T = self.gamma(ones)
        mu = self.alpha(gamma_T, xh_lig)[mask_lig] * xh_lig
        sigma_T = self.sigma(gamma_T, xh_lig).squeeze()
        one = torch.ones_like(sigma_T)
        nb = len(num_nodes)
        kl_h = self.gaussian_KL(self.sum_except_batch(mu[:, nd:] ** 2, mask_lig, nb), sigma_T, one, d=1)
        kl_x = self.gaussian_KL(self.sum_except_batch(mu[:, :nd] ** 2, mask_lig, nb), sigma_T, one,
                                self.subspace_dimensionality(num_nodes))
        return kl_x + kl_h

    def forward(self, ligand, pocket, return_info=False):
        """The reference's loss terms for the pocket-conditioned model (conditional_model.py:202-330):
        same 12-tuple as the joint model with error_t_pocket = loss_0_x_pocket = 0."""
        from . import loss_head
        if loss_head.fused_ok(self, self._hip_device(None)):
            # training mode under autograd, predefined schedule: the same terms on three HIP launches around the network
            # call (csrc/loss_head.h); DSBDD_LOSS=torch keeps the torch terms below
            return loss_head.conditional_forward(self, ligand, pocket, return_info)
        with self._loss_context():
            dev = self._hip_device(None)
            ligand, pocket = self._to_device(ligand, dev), self._to_device(pocket, dev)
            ligand, pocket = self.normalize(ligand, pocket)
            lm, pm = ligand['mask'], pocket['mask']
            n = ligand['size'].size(0)
            nd = self.n_dims
            delta_log_px = self.delta_log_px(ligand['size'])
            t_int = self._draw_t_int(n, dev)
            s_int = t_int - 1
            t_is_zero = (t_int == 0).float()
            s, t = s_int / self.T, t_int / self.T
            gamma_s = self.inflate_batch_array(self.gamma(s), ligand['x'])
            gamma_t = self.inflate_batch_array(self.gamma(t), ligand['x'])
            xh0_lig = torch.cat([ligand['x'], ligand['one_hot']], dim=1)
            xh0_pocket = torch.cat([pocket['x'], pocket['one_hot']], dim=1)
            xl, xp = self._remove_lig_com(xh0_lig[:, :nd], xh0_pocket[:, :nd], lm, pm, n)         # :232-236
            xh0_lig = torch.cat([xl, xh0_lig[:, nd:]], dim=1)
            xh0_pocket = torch.cat([xp, xh0_pocket[:, nd:]], dim=1)
            z_t, xh_pocket, eps_t = self.noised_representation(xh0_lig, xh0_pocket, lm, pm, gamma_t)
            # the terms that do not read the network's output come first: the host issues them while the GPU still works
            # on the previous step, and nothing but the error terms stands between the network call and backward()
            # (same values: none of them draws random numbers).  No term below synchronises with the device.
            SNR_weight = (1 - self.SNR(gamma_s - gamma_t)).squeeze(1)
            neg_log_constants = -self.log_constants_p_x_given_z0(n_nodes=ligand['size'], device=dev)
            kl_prior = self.kl_prior(xh0_lig, lm, ligand['size'])
            log_pN = self.log_pN(ligand['size'], pocket['size'])
            vmask = vmask2 = None
            if self.vnode_idx is not None:
                vmask = ligand['one_hot'][:, self.vnode_idx].bool()
                vmask2 = torch.zeros(vmask.shape[0], nd + ligand['one_hot'].shape[1], dtype=torch.bool, device=dev)
                vmask2[:, :nd] = vmask.unsqueeze(1)
            l0_h_train = -self._log_ph_given_z0(ligand['one_hot'], z_t[:, nd:], lm, gamma_t) if self.training else None
            net, _ = self.dynamics(z_t, xh_pocket, t, lm, pm)
            xh_lig_hat = self.xh_given_zt_and_epsilon(z_t, net, gamma_t, lm)
            squared_error = (eps_t - net) ** 2
            if vmask is not None:                # coordinates of virtual atoms do not contribute (:253-255)
                squared_error = squared_error.masked_fill(vmask2, 0)      # (no boolean-index assignment: that is a nonzero() + sync)
            error_t_lig = self.sum_except_batch(squared_error, lm, n)

            def loss0(z0, e0, n0, g, l0_h=None):
                sq = (e0[:, :nd] - n0[:, :nd]) ** 2
                if vmask is not None:
                    sq = sq.masked_fill(vmask.unsqueeze(1), 0)
                if l0_h is None:
                    l0_h = -self._log_ph_given_z0(ligand['one_hot'], z0[:, nd:], lm, g)
                return 0.5 * self.sum_except_batch(sq, lm, n), l0_h

            if self.training:
                tz = t_is_zero.squeeze()
                l0_x, l0_h = loss0(z_t, eps_t, net, gamma_t, l0_h_train)
                l0_x, l0_h = l0_x * tz, l0_h * tz
                error_t_lig = error_t_lig * (1 - tz)
            else:                                   # separate pass at t = 0 (:285-302)
                t_zeros = torch.zeros_like(s)
                gamma_0 = self.inflate_batch_array(self.gamma(t_zeros), ligand['x'])
                z_0, xh_pocket0, eps_0 = self.noised_representation(xh0_lig, xh0_pocket, lm, pm, gamma_0)
                net_0, _ = self.dynamics(z_0, xh_pocket0, t_zeros, lm, pm)
                l0_x, l0_h = loss0(z_0, eps_0, net_0, gamma_0)
            info = {
                'eps_hat_lig_x': seg_mean(net[:, :nd].abs().mean(1), lm, n).mean(),
                'eps_hat_lig_h': seg_mean(net[:, nd:].abs().mean(1), lm, n).mean(),
            }
            zero = torch.tensor(0.0)
            loss_terms = (delta_log_px, error_t_lig, zero, SNR_weight, l0_x, zero.clone(), l0_h,
                          neg_log_constants, kl_prior, log_pN, t_int.squeeze(), xh_lig_hat)
        return (*loss_terms, info) if return_info else loss_terms

    def log_pN(self, N_lig, N_pocket):
        return self.size_distribution.log_prob_n1_given_n2(N_lig, N_pocket)


class SimpleConditionalDDPM(ConditionalDDPM):
    """The same model without the subspace trick (conditional_model.py:702-746):
    no COM projection; the pocket is centred once before sampling."""

    _remove_com = 0

    def subspace_dimensionality(self, input_size):
        return input_size * self.n_dims

    @classmethod
    def remove_mean_batch(cls, x_lig, x_pocket, lig_indices, pocket_indices):
        return x_lig, x_pocket

    @staticmethod
    def assert_mean_zero_with_mask(x, node_mask, eps=1e-10):
        return

    @torch.no_grad()
    def sample_given_pocket(self, pocket, num_nodes_lig, return_frames=1, timesteps=None):
        n = len(pocket['size'])
        com = seg_mean(pocket['x'], pocket['mask'], n)
        pocket['x'] = pocket['x'] - com[pocket['mask']]
        return super().sample_given_pocket(pocket, num_nodes_lig, return_frames, timesteps)
