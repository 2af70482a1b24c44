"""ORACLE (test infrastructure, NOT product code).

CPU restatement of the reference's DDPM sampling path around the EGNN
denoiser: the predefined noise schedule, the per-step posterior update, the
final decode, and the sampling / RePaint loops, for both the conditional model
(`ConditionalDDPM`, conditional_model.py) and the joint model
(`EnVariationalDiffusion`, en_diffusion.py).

Noise is *injected*: every function that draws Gaussian noise takes a
`noise(shape) -> tensor` callable, so that the oracle, the real reference (via
a patched `sample_gaussian` / `torch.randn`) and the HIP path consume identical
numbers.  Citations are /root/reference/<path>:<line>.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from . import egnn_oracle as eo


# ---------------------------------------------------------------------------
# noise schedule: en_diffusion.py:1125-1190
# ---------------------------------------------------------------------------
def clip_noise_schedule(alphas2, clip_value=0.001):
    """en_diffusion.py:1125-1138."""
    alphas2 = np.concatenate([np.ones(1), alphas2], axis=0)
    step = np.clip(alphas2[1:] / alphas2[:-1], a_min=clip_value, a_max=1.0)
    return np.cumprod(step, axis=0)


def polynomial_schedule(timesteps, s=1e-4, power=3.0):
    """en_diffusion.py:1141-1155."""
    steps = timesteps + 1
    x = np.linspace(0, steps, steps)
    alphas2 = (1 - np.power(x / steps, power)) ** 2
    alphas2 = clip_noise_schedule(alphas2, clip_value=0.001)
    precision = 1 - 2 * s
    return precision * alphas2 + s


def cosine_beta_schedule(timesteps, s=0.008, raise_to_power=1.0):
    """en_diffusion.py:1105-1122."""
    steps = timesteps + 2
    x = np.linspace(0, steps, steps)
    ac = np.cos(((x / steps) + s) / (1 + s) * np.pi * 0.5) ** 2
    ac = ac / ac[0]
    betas = np.clip(1 - (ac[1:] / ac[:-1]), a_min=0, a_max=0.999)
    ac = np.cumprod(1.0 - betas, axis=0)
    if raise_to_power != 1:
        ac = np.power(ac, raise_to_power)
    return ac


def gamma_table(noise_schedule, timesteps, precision):
    """PredefinedNoiseSchedule.__init__, en_diffusion.py:1163-1186.
    Returns the fp32 table gamma[0..T]."""
    if noise_schedule == "cosine":
        alphas2 = cosine_beta_schedule(timesteps)
    elif "polynomial" in noise_schedule:
        power = float(noise_schedule.split("_")[1])
        alphas2 = polynomial_schedule(timesteps, s=precision, power=power)
    else:
        raise ValueError(noise_schedule)
    sigmas2 = 1 - alphas2
    return torch.from_numpy(-(np.log(alphas2) - np.log(sigmas2))).float()


def gamma_at(table, t, timesteps):
    """PredefinedNoiseSchedule.forward, en_diffusion.py:1188-1190."""
    return table[torch.round(t * timesteps).long()]


def sigma(g):
    """en_diffusion.py:865-868."""
    return torch.sqrt(torch.sigmoid(g))


def alpha(g):
    """en_diffusion.py:870-873."""
    return torch.sqrt(torch.sigmoid(-g))


def sigma_and_alpha_t_given_s(g_t, g_s):
    """en_diffusion.py:83-107."""
    sigma2 = -torch.expm1(F.softplus(g_s) - F.softplus(g_t))
    log_a2 = F.logsigmoid(-g_t) - F.logsigmoid(-g_s)
    return sigma2, torch.sqrt(sigma2), torch.exp(0.5 * log_a2)


def repaint_schedule(resamplings, jump_length, timesteps):
    """en_diffusion.py:653-674."""
    sched, cur = [], 0
    while cur < timesteps:
        if cur + jump_length < timesteps:
            if sched:
                sched[-1] += jump_length
                sched.extend([jump_length] * (resamplings - 1))
            else:
                sched.extend([jump_length] * resamplings)
            cur += jump_length
        else:
            res = timesteps - cur
            if sched:
                sched[-1] += res
            else:
                sched.append(res)
            cur += res
    return list(reversed(sched))


# ---------------------------------------------------------------------------
# a tiny model container
# ---------------------------------------------------------------------------
class OracleModel:
    """Weights + hyper-parameters of one DDPM (either flavour).

    sd   : state_dict of the *dynamics* (keys without the `dynamics.` prefix)
    cfg  : EGNNDynamics kwargs (egnn_oracle.dynamics_forward)
    conditional : True -> ConditionalDDPM semantics, False -> joint
    """

    def __init__(self, sd, cfg, atom_nf, residue_nf, timesteps, noise_schedule,
                 noise_precision, norm_values=(1.0, 1.0), norm_biases=(None, 0.0),
                 conditional=True, n_dims=3, simple=False):
        self.sd, self.cfg = sd, cfg
        self.atom_nf, self.residue_nf, self.n_dims = atom_nf, residue_nf, n_dims
        self.T = timesteps
        self.gamma = gamma_table(noise_schedule, timesteps, noise_precision)
        self.norm_values, self.norm_biases = norm_values, norm_biases
        self.conditional = conditional
        self.simple = simple  # SimpleConditionalDDPM: conditional_model.py:702-746
        self.n_dynamics_calls = 0
        self.edge_hook = None  # optional: fn(call_index) -> edges to teacher-force
        self.exact_dist = True  # False: torch.cdist like the reference (dynamics.py:174-181)

    def g(self, t):
        return gamma_at(self.gamma, t, self.T)

    def dynamics(self, z_lig, z_pocket, t, lig_mask, pocket_mask):
        edges = self.edge_hook(self.n_dynamics_calls) if self.edge_hook else None
        self.n_dynamics_calls += 1
        e_l, e_p, _ = eo.dynamics_forward(self.sd, self.cfg, z_lig, z_pocket, t,
                                          lig_mask, pocket_mask, edges=edges, exact_dist=self.exact_dist)
        return e_l, e_p


def _seg_mean(x, idx, n):
    return eo.segment_mean(x, idx, n)


# ---------------------------------------------------------------------------
# conditional model: conditional_model.py
# ---------------------------------------------------------------------------
def cond_remove_mean(m, x_lig, x_pocket, lig_mask, pocket_mask, n):
    """ConditionalDDPM.remove_mean_batch, conditional_model.py:688-696:
    subtract the LIGAND centre of mass from ligand and pocket.
    (SimpleConditionalDDPM: identity, :717-721.)"""
    if m.simple:
        return x_lig, x_pocket
    mean = _seg_mean(x_lig, lig_mask, n)
    return x_lig - mean[lig_mask], x_pocket - mean[pocket_mask]


def cond_sample_normal_zero_com(m, mu_lig, xh0_pocket, sig, lig_mask, pocket_mask, noise, n):
    """conditional_model.py:140-160."""
    nd = m.n_dims
    eps = noise((len(lig_mask), nd + m.atom_nf))                         # :147-149
    out = mu_lig + sig[lig_mask] * eps                                   # :151
    xh_pocket = xh0_pocket.clone()
    xl, xp = cond_remove_mean(m, out[:, :nd], xh0_pocket[:, :nd], lig_mask, pocket_mask, n)
    out = torch.cat([xl, out[:, nd:]], 1)
    xh_pocket = torch.cat([xp, xh_pocket[:, nd:]], 1)
    return out, xh_pocket


def cond_step_coeffs(m, s, t):
    """The per-step scalars of conditional_model.py:435-442,456."""
    g_s, g_t = m.g(s), m.g(t)
    s2, s_ts, a_ts = sigma_and_alpha_t_given_s(g_t, g_s)
    sig_s, sig_t = sigma(g_s), sigma(g_t)
    return s2, s_ts, a_ts, sig_s, sig_t


def cond_sample_p_zs_given_zt(m, s, t, zt_lig, xh0_pocket, lig_mask, pocket_mask, noise):
    """conditional_model.py:432-464."""
    n = s.shape[0]
    s2, s_ts, a_ts, sig_s, sig_t = cond_step_coeffs(m, s, t)
    eps, _ = m.dynamics(zt_lig, xh0_pocket, t, lig_mask, pocket_mask)    # :445
    mu = zt_lig / a_ts[lig_mask] - (s2 / a_ts / sig_t)[lig_mask] * eps   # :451-453
    sg = s_ts * sig_s / sig_t                                            # :456
    return cond_sample_normal_zero_com(m, mu, xh0_pocket, sg, lig_mask, pocket_mask, noise, n)


def cond_sample_p_xh_given_z0(m, z0_lig, xh0_pocket, lig_mask, pocket_mask, n, noise):
    """conditional_model.py:112-135 (+ compute_x_pred en_diffusion.py:157-169)."""
    nd = m.n_dims
    t0 = torch.zeros((n, 1))
    g0 = m.g(t0)
    sigma_x = torch.exp(0.5 * g0)                                        # SNR(-0.5 g0) :118
    net, _ = m.dynamics(z0_lig, xh0_pocket, t0, lig_mask, pocket_mask)
    mu = 1.0 / alpha(g0)[lig_mask] * (z0_lig - sigma(g0)[lig_mask] * net)  # :166
    xh_lig, xh_pocket = cond_sample_normal_zero_com(
        m, mu, xh0_pocket, sigma_x, lig_mask, pocket_mask, noise, n)
    x_lig = xh_lig[:, :nd] * m.norm_values[0]                            # unnormalize :897-901
    h_lig = z0_lig[:, nd:] * m.norm_values[1] + m.norm_biases[1]
    x_p = xh_pocket[:, :nd] * m.norm_values[0]
    h_p = xh_pocket[:, nd:] * m.norm_values[1] + m.norm_biases[1]
    h_lig = F.one_hot(torch.argmax(h_lig, 1), m.atom_nf)                 # :132
    return x_lig, h_lig, x_p, h_p


def normalize(m, ligand=None, pocket=None):
    """en_diffusion.py:880-895 (returns new dicts; the reference mutates)."""
    def nz(d):
        if d is None:
            return None
        d = dict(d)
        d["x"] = d["x"] / m.norm_values[0]
        d["one_hot"] = (d["one_hot"].float() - m.norm_biases[1]) / m.norm_values[1]
        return d
    return nz(ligand), nz(pocket)


def cond_sample_given_pocket(m, pocket, num_nodes_lig, noise, timesteps=None, trace=None):
    """ConditionalDDPM.sample_given_pocket, conditional_model.py:478-555
    (return_frames=1).  `trace`, if a list, receives (z_t_lig, xh_pocket_t,
    z_s_lig, xh_pocket_s) per reverse step for teacher-forced comparisons."""
    nd = m.n_dims
    timesteps = m.T if timesteps is None else timesteps
    n = len(pocket["size"])
    pocket = dict(pocket)
    if m.simple:                                                         # :741-743
        com = _seg_mean(pocket["x"], pocket["mask"], n)
        pocket["x"] = pocket["x"] - com[pocket["mask"]]
    _, pocket = normalize(m, None, pocket)                               # :492
    xh0_pocket = torch.cat([pocket["x"], pocket["one_hot"]], 1)
    lig_mask = torch.repeat_interleave(torch.arange(n), num_nodes_lig)   # utils.py:146-154
    mu_x = _seg_mean(pocket["x"], pocket["mask"], n)                     # :502
    mu = torch.cat((mu_x, torch.zeros((n, m.atom_nf))), 1)[lig_mask]
    sig = torch.ones((n, 1))
    z_lig, xh_pocket = cond_sample_normal_zero_com(
        m, mu, xh0_pocket, sig, lig_mask, pocket["mask"], noise, n)      # :507
    for s in reversed(range(timesteps)):                                 # :518-526
        s_arr = torch.full((n, 1), float(s)) / timesteps
        t_arr = torch.full((n, 1), float(s + 1)) / timesteps
        z_prev, p_prev = z_lig, xh_pocket
        z_lig, xh_pocket = cond_sample_p_zs_given_zt(
            m, s_arr, t_arr, z_lig, xh_pocket, lig_mask, pocket["mask"], noise)
        if trace is not None:
            trace.append((z_prev, p_prev, z_lig, xh_pocket))
    x_lig, h_lig, x_p, h_p = cond_sample_p_xh_given_z0(
        m, z_lig, xh_pocket, lig_mask, pocket["mask"], n, noise)         # :535
    max_cog = eo.segment_sum(x_lig, lig_mask, n).abs().max().item()      # :542
    if max_cog > 5e-2:
        x_lig, x_p = cond_remove_mean(m, x_lig, x_p, lig_mask, pocket["mask"], n)
    return (torch.cat([x_lig, h_lig], 1), torch.cat([x_p, h_p], 1),
            lig_mask, pocket["mask"])


def cond_noised_representation(m, xh_lig, xh0_pocket, lig_mask, pocket_mask, g_t, noise, n):
    """conditional_model.py:162-183."""
    nd = m.n_dims
    eps = noise((len(lig_mask), nd + m.atom_nf))
    z = alpha(g_t)[lig_mask] * xh_lig + sigma(g_t)[lig_mask] * eps
    xl, xp = cond_remove_mean(m, z[:, :nd], xh0_pocket[:, :nd], lig_mask, pocket_mask, n)
    return torch.cat([xl, z[:, nd:]], 1), torch.cat([xp, xh0_pocket[:, nd:]], 1), eps


def cond_sample_p_zt_given_zs(m, zs_lig, xh0_pocket, lig_mask, pocket_mask, g_t, g_s, noise, n):
    """conditional_model.py:420-430."""
    _, s_ts, a_ts = sigma_and_alpha_t_given_s(g_t, g_s)
    mu = a_ts[lig_mask] * zs_lig
    return cond_sample_normal_zero_com(m, mu, xh0_pocket, s_ts, lig_mask, pocket_mask, noise, n)


def cond_inpaint_iteration(m, s, timesteps, z_lig, xh_pocket, x0_lig, h0_lig, com_pocket_0, lig_fixed, lm, pm,
                           noise, resample=False):
    """One (s, u) iteration of ConditionalDDPM.inpaint's loop body, conditional_model.py:600-660: reverse step
    of the unknown part, the known part noised to level s around the moved pocket, COM alignment over the fixed
    atoms, blend, and (between resamplings) q(z_t | z_s).  x0_lig / h0_lig: the normalised known ligand."""
    nd = m.n_dims
    n = com_pocket_0.shape[0]
    fixed = lig_fixed.bool().view(-1)
    lf = lig_fixed.view(-1, 1).to(z_lig.dtype)
    s_arr = torch.full((n, 1), float(s)) / timesteps
    t_arr = torch.full((n, 1), float(s + 1)) / timesteps
    g_t, g_s = m.g(t_arr), m.g(s_arr)
    z_unknown, xh_pocket = cond_sample_p_zs_given_zt(m, s_arr, t_arr, z_lig, xh_pocket, lm, pm, noise)
    com_pocket = _seg_mean(xh_pocket[:, :nd], pm, n)
    xh_ligand = torch.cat([x0_lig + (com_pocket - com_pocket_0)[lm], h0_lig], 1)
    z_known, xh_pocket, _ = cond_noised_representation(m, xh_ligand, xh_pocket, lm, pm, g_s, noise, n)
    com_noised = _seg_mean(z_known[fixed][:, :nd], lm[fixed], n)
    com_denoised = _seg_mean(z_unknown[fixed][:, :nd], lm[fixed], n)
    dx = com_denoised - com_noised
    z_known = torch.cat([z_known[:, :nd] + dx[lm], z_known[:, nd:]], 1)
    xh_pocket = torch.cat([xh_pocket[:, :nd] + dx[pm], xh_pocket[:, nd:]], 1)
    z_lig = z_known * lf + z_unknown * (1 - lf)
    if resample:
        z_lig, xh_pocket = cond_sample_p_zt_given_zs(m, z_lig, xh_pocket, lm, pm, g_t, g_s, noise, n)
    return z_lig, xh_pocket


def cond_inpaint(m, ligand, pocket, lig_fixed, noise, resamplings=1, timesteps=None,
                 center="ligand"):
    """ConditionalDDPM.inpaint, conditional_model.py:557-686 (return_frames=1)."""
    nd = m.n_dims
    timesteps = m.T if timesteps is None else timesteps
    if lig_fixed.dim() == 1:
        lig_fixed = lig_fixed.unsqueeze(1)
    n = len(ligand["size"])
    ligand, pocket = normalize(m, ligand, pocket)
    lm, pm = ligand["mask"], pocket["mask"]
    fixed = lig_fixed.bool().view(-1)
    xh0_pocket = torch.cat([pocket["x"], pocket["one_hot"]], 1)
    com_pocket_0 = _seg_mean(pocket["x"], pm, n)
    xh_ligand = torch.cat([ligand["x"], ligand["one_hot"]], 1).clone()
    if center == "ligand":
        mean_known = _seg_mean(ligand["x"][fixed], lm[fixed], n)
    elif center == "pocket":
        mean_known = _seg_mean(pocket["x"], pm, n)
    else:
        raise NotImplementedError(center)
    mu = torch.cat((mean_known, torch.zeros((n, m.atom_nf))), 1)[lm]
    z_lig, xh_pocket = cond_sample_normal_zero_com(
        m, mu, xh0_pocket, torch.ones((n, 1)), lm, pm, noise, n)
    for s in reversed(range(timesteps)):
        for u in range(resamplings):
            z_lig, xh_pocket = cond_inpaint_iteration(
                m, s, timesteps, z_lig, xh_pocket, ligand["x"], xh_ligand[:, nd:], com_pocket_0, lig_fixed, lm, pm,
                noise, resample=u < resamplings - 1)
    x_lig, h_lig, x_p, h_p = cond_sample_p_xh_given_z0(m, z_lig, xh_pocket, lm, pm, n, noise)
    return torch.cat([x_lig, h_lig], 1), torch.cat([x_p, h_p], 1), lm, pm


def cond_diversify(m, ligand, pocket, noising_steps, noise):
    """ConditionalDDPM.diversify + partially_noised_ligand,
    conditional_model.py:332-409."""
    nd = m.n_dims
    ligand, pocket = normalize(m, ligand, pocket)
    n = len(pocket["size"])
    lm, pm = ligand["mask"], pocket["mask"]
    t = torch.ones((n, 1)) * noising_steps / m.T
    g_t = m.g(t)
    xh0_lig = torch.cat([ligand["x"], ligand["one_hot"]], 1)
    xh0_pocket = torch.cat([pocket["x"], pocket["one_hot"]], 1)
    xl, xp = cond_remove_mean(m, xh0_lig[:, :nd], xh0_pocket[:, :nd], lm, pm, n)
    xh0_lig = torch.cat([xl, xh0_lig[:, nd:]], 1)
    xh0_pocket = torch.cat([xp, xh0_pocket[:, nd:]], 1)
    z_lig, xh_pocket, _ = cond_noised_representation(m, xh0_lig, xh0_pocket, lm, pm, g_t, noise, n)
    for s in reversed(range(noising_steps)):
        s_arr = torch.full((n, 1), float(s)) / m.T
        t_arr = torch.full((n, 1), float(s + 1)) / m.T
        z_lig, xh_pocket = cond_sample_p_zs_given_zt(m, s_arr, t_arr, z_lig, xh_pocket, lm, pm, noise)
    x_lig, h_lig, x_p, h_p = cond_sample_p_xh_given_z0(m, z_lig, xh_pocket, lm, pm, n, noise)
    return torch.cat([x_lig, h_lig], 1), torch.cat([x_p, h_p], 1), lm, pm


# ---------------------------------------------------------------------------
# joint model: en_diffusion.py
# ---------------------------------------------------------------------------
def joint_noise(m, lig_mask, pocket_mask, noise):
    """sample_combined_position_feature_noise, en_diffusion.py:559-578:
    draw order = x for all nodes (COM-projected), then h_lig, then h_pocket."""
    nl = len(lig_mask)
    zx = noise((nl + len(pocket_mask), m.n_dims))
    zx = eo.remove_mean_batch(zx, torch.cat((lig_mask, pocket_mask)))    # :932-942
    zh_l = noise((nl, m.atom_nf))
    zh_p = noise((len(pocket_mask), m.residue_nf))
    return torch.cat([zx[:nl], zh_l], 1), torch.cat([zx[nl:], zh_p], 1)


def _joint_remove_mean(m, z_lig, z_pocket, lig_mask, pocket_mask):
    nd, nl = m.n_dims, len(lig_mask)
    zx = eo.remove_mean_batch(torch.cat((z_lig[:, :nd], z_pocket[:, :nd]), 0),
                              torch.cat((lig_mask, pocket_mask)))
    return torch.cat((zx[:nl], z_lig[:, nd:]), 1), torch.cat((zx[nl:], z_pocket[:, nd:]), 1)


def joint_sample_p_zs_given_zt(m, s, t, zt_lig, zt_pocket, lig_mask, pocket_mask, noise):
    """en_diffusion.py:503-557."""
    g_s, g_t = m.g(s), m.g(t)
    s2, s_ts, a_ts = sigma_and_alpha_t_given_s(g_t, g_s)
    sig_s, sig_t = sigma(g_s), sigma(g_t)
    e_l, e_p = m.dynamics(zt_lig, zt_pocket, t, lig_mask, pocket_mask)
    c = s2 / a_ts / sig_t
    mu_l = zt_lig / a_ts[lig_mask] - c[lig_mask] * e_l                   # :532-534
    mu_p = zt_pocket / a_ts[pocket_mask] - c[pocket_mask] * e_p          # :535-537
    sg = s_ts * sig_s / sig_t                                            # :540
    n_l, n_p = joint_noise(m, lig_mask, pocket_mask, noise)              # sample_normal :290-300
    zs_l = mu_l + sg[lig_mask] * n_l
    zs_p = mu_p + sg[pocket_mask] * n_p
    return _joint_remove_mean(m, zs_l, zs_p, lig_mask, pocket_mask)      # :547-556


def joint_sample_p_xh_given_z0(m, z0_lig, z0_pocket, lig_mask, pocket_mask, n, noise):
    """en_diffusion.py:263-288."""
    nd = m.n_dims
    t0 = torch.zeros((n, 1))
    g0 = m.g(t0)
    sigma_x = torch.exp(0.5 * g0)
    e_l, e_p = m.dynamics(z0_lig, z0_pocket, t0, lig_mask, pocket_mask)
    mu_l = 1.0 / alpha(g0)[lig_mask] * (z0_lig - sigma(g0)[lig_mask] * e_l)
    mu_p = 1.0 / alpha(g0)[pocket_mask] * (z0_pocket - sigma(g0)[pocket_mask] * e_p)
    n_l, n_p = joint_noise(m, lig_mask, pocket_mask, noise)
    xh_l = mu_l + sigma_x[lig_mask] * n_l
    xh_p = mu_p + sigma_x[pocket_mask] * n_p
    x_l = xh_l[:, :nd] * m.norm_values[0]
    h_l = z0_lig[:, nd:] * m.norm_values[1] + m.norm_biases[1]
    x_p = xh_p[:, :nd] * m.norm_values[0]
    h_p = z0_pocket[:, nd:] * m.norm_values[1] + m.norm_biases[1]
    return (x_l, F.one_hot(torch.argmax(h_l, 1), m.atom_nf),
            x_p, F.one_hot(torch.argmax(h_p, 1), m.residue_nf))


def _joint_final(m, z_lig, z_pocket, lig_mask, pocket_mask, n, noise):
    x_l, h_l, x_p, h_p = joint_sample_p_xh_given_z0(m, z_lig, z_pocket, lig_mask, pocket_mask, n, noise)
    comb = torch.cat((lig_mask, pocket_mask))
    x = torch.cat((x_l, x_p))
    if eo.segment_sum(x, comb, n).abs().max().item() > 5e-2:              # :637-644
        x = eo.remove_mean_batch(x, comb)
        x_l, x_p = x[:len(x_l)], x[len(x_l):]
    return torch.cat([x_l, h_l], 1), torch.cat([x_p, h_p], 1), lig_mask, pocket_mask


def joint_sample(m, n, num_nodes_lig, num_nodes_pocket, noise, timesteps=None):
    """EnVariationalDiffusion.sample, en_diffusion.py:580-651 (return_frames=1)."""
    timesteps = m.T if timesteps is None else timesteps
    lig_mask = torch.repeat_interleave(torch.arange(n), num_nodes_lig)
    pocket_mask = torch.repeat_interleave(torch.arange(n), num_nodes_pocket)
    z_l, z_p = joint_noise(m, lig_mask, pocket_mask, noise)
    for s in reversed(range(timesteps)):
        s_arr = torch.full((n, 1), float(s)) / timesteps
        t_arr = torch.full((n, 1), float(s + 1)) / timesteps
        z_l, z_p = joint_sample_p_zs_given_zt(m, s_arr, t_arr, z_l, z_p, lig_mask, pocket_mask, noise)
    return _joint_final(m, z_l, z_p, lig_mask, pocket_mask, n, noise)


def joint_noised_representation(m, xh_lig, xh_pocket, lig_mask, pocket_mask, g_t, noise):
    """en_diffusion.py:302-317."""
    n_l, n_p = joint_noise(m, lig_mask, pocket_mask, noise)
    z_l = alpha(g_t)[lig_mask] * xh_lig + sigma(g_t)[lig_mask] * n_l
    z_p = alpha(g_t)[pocket_mask] * xh_pocket + sigma(g_t)[pocket_mask] * n_p
    return z_l, z_p


def joint_sample_p_zt_given_zs(m, zs_lig, zs_pocket, lig_mask, pocket_mask, g_t, g_s, noise):
    """en_diffusion.py:479-501."""
    _, s_ts, a_ts = sigma_and_alpha_t_given_s(g_t, g_s)
    n_l, n_p = joint_noise(m, lig_mask, pocket_mask, noise)
    zt_l = a_ts[lig_mask] * zs_lig + s_ts[lig_mask] * n_l
    zt_p = a_ts[pocket_mask] * zs_pocket + s_ts[pocket_mask] * n_p
    return _joint_remove_mean(m, zt_l, zt_p, lig_mask, pocket_mask)


def joint_inpaint_iteration(m, s, timesteps, z_l, z_p, xh0_l, xh0_p, lig_fixed, pocket_fixed, lm, pm, noise,
                            jump_to=None):
    """One iteration of EnVariationalDiffusion.inpaint's loop body, en_diffusion.py:742-809: the known part noised
    to level s (:745), one reverse step of the whole state (:749), COM alignment over the fixed nodes (:752-772),
    blend (:775-778) and -- at the end of a resampling segment (`jump_to` = s + jump_length) -- the jump back
    q(z_t | z_s) (:793-809).  xh0_*: the normalised input centred at the COM of the known nodes."""
    nd = m.n_dims
    n = int(max(lm.max(), pm.max())) + 1
    lfb, pfb = lig_fixed.bool().view(-1), pocket_fixed.bool().view(-1)
    lf, pf = lig_fixed.view(-1, 1).to(z_l.dtype), pocket_fixed.view(-1, 1).to(z_l.dtype)
    s_arr = torch.full((n, 1), float(s)) / timesteps
    t_arr = torch.full((n, 1), float(s + 1)) / timesteps
    g_s = m.g(s_arr)
    zk_l, zk_p = joint_noised_representation(m, xh0_l, xh0_p, lm, pm, g_s, noise)  # :745
    zu_l, zu_p = joint_sample_p_zs_given_zt(m, s_arr, t_arr, z_l, z_p, lm, pm, noise)  # :749
    idx = torch.cat((lm[lfb], pm[pfb]))
    com_n = _seg_mean(torch.cat((zk_l[:, :nd][lfb], zk_p[:, :nd][pfb])), idx, n)
    com_d = _seg_mean(torch.cat((zu_l[:, :nd][lfb], zu_p[:, :nd][pfb])), idx, n)
    dx = com_d - com_n
    zk_l = torch.cat([zk_l[:, :nd] + dx[lm], zk_l[:, nd:]], 1)
    zk_p = torch.cat([zk_p[:, :nd] + dx[pm], zk_p[:, nd:]], 1)
    z_l = zk_l * lf + zu_l * (1 - lf)                            # :775-778
    z_p = zk_p * pf + zu_p * (1 - pf)
    if jump_to is not None:                                      # :793-809
        t_arr2 = torch.full((n, 1), float(jump_to)) / timesteps
        z_l, z_p = joint_sample_p_zt_given_zs(m, z_l, z_p, lm, pm, m.g(t_arr2), g_s, noise)
    return z_l, z_p


def joint_inpaint(m, ligand, pocket, lig_fixed, pocket_fixed, noise, resamplings=1,
                  jump_length=1, timesteps=None):
    """EnVariationalDiffusion.inpaint, en_diffusion.py:676-837 (return_frames=1)."""
    nd = m.n_dims
    timesteps = m.T if timesteps is None else timesteps
    if lig_fixed.dim() == 1:
        lig_fixed = lig_fixed.unsqueeze(1)
    if pocket_fixed.dim() == 1:
        pocket_fixed = pocket_fixed.unsqueeze(1)
    ligand, pocket = normalize(m, ligand, pocket)
    n = len(ligand["size"])
    lm, pm = ligand["mask"], pocket["mask"]
    lfb, pfb = lig_fixed.bool().view(-1), pocket_fixed.bool().view(-1)
    xh0_l = torch.cat([ligand["x"], ligand["one_hot"]], 1)
    xh0_p = torch.cat([pocket["x"], pocket["one_hot"]], 1)
    mean_known = _seg_mean(torch.cat((ligand["x"][lfb], pocket["x"][pfb])),
                           torch.cat((lm[lfb], pm[pfb])), n)             # :706-712
    xh0_l = torch.cat([xh0_l[:, :nd] - mean_known[lm], xh0_l[:, nd:]], 1)
    xh0_p = torch.cat([xh0_p[:, :nd] - mean_known[pm], xh0_p[:, nd:]], 1)
    z_l, z_p = joint_noise(m, lm, pm, noise)                             # :719
    sched = repaint_schedule(resamplings, jump_length, timesteps)
    s = timesteps - 1
    for i, n_steps in enumerate(sched):
        for j in range(n_steps):
            jump = j == n_steps - 1 and i < len(sched) - 1
            z_l, z_p = joint_inpaint_iteration(m, s, timesteps, z_l, z_p, xh0_l, xh0_p, lig_fixed, pocket_fixed,
                                               lm, pm, noise, jump_to=s + jump_length if jump else None)
            if jump:
                s = s + jump_length
            s -= 1
    return _joint_final(m, z_l, z_p, lm, pm, n, noise)


# ---------------------------------------------------------------------------
# noise sources
# ---------------------------------------------------------------------------
class NoiseTape:
    """Deterministic noise source: a seeded CPU generator, optionally recording
    every draw so a second consumer can replay the identical sequence."""

    def __init__(self, seed=1234, dtype=torch.float32):
        self.gen = torch.Generator().manual_seed(seed)
        self.dtype = dtype
        self.draws = []

    def __call__(self, shape):
        x = torch.randn(tuple(shape), generator=self.gen, dtype=torch.float32).to(self.dtype)
        self.draws.append(x)
        return x


class NoiseReplay:
    def __init__(self, draws):
        self.draws, self.i = list(draws), 0

    def __call__(self, shape):
        x = self.draws[self.i]
        self.i += 1
        assert tuple(x.shape) == tuple(shape), (x.shape, shape)
        return x


# ---------------------------------------------------------------------------
# training / validation loss terms: conditional_model.py:202-330,
# en_diffusion.py:336-469 with the random draws injected (t_int and the Gaussian noise)
# ---------------------------------------------------------------------------
def _sum_except_batch(x, idx, n):
    """en_diffusion.py:944-946."""
    return eo.segment_sum(x.sum(-1, keepdim=True), idx, n).squeeze(1)


def _gaussian_kl(mu_norm2, q_sigma, p_sigma, d):
    """en_diffusion.py:838-853."""
    return d * torch.log(p_sigma / q_sigma) + 0.5 * (d * q_sigma ** 2 + mu_norm2) / (p_sigma ** 2) - 0.5 * d


def _cdf(x):
    return 0.5 * (1.0 + torch.erf(x / np.sqrt(2)))


def _subspace_dim(m, n_nodes):
    """en_diffusion.py:914-917; SimpleConditionalDDPM: conditional_model.py:713-715."""
    return n_nodes * m.n_dims if m.simple else (n_nodes - 1) * m.n_dims


def _log_ph_given_z0(m, one_hot, z_h, mask, g0, n, epsilon=1e-10):
    """The categorical part of log_pxh_given_z0_without_constants (en_diffusion.py:214-260,
    conditional_model.py:76-110): integral of N(z_h, sigma_0) over [0.5, 1.5] per class, normalised."""
    nv, nb = m.norm_values[1], m.norm_biases[1]
    sigma_0_cat = sigma(g0) * nv
    onehot = one_hot * nv + nb
    centered = (z_h * nv + nb) - 1
    logp = torch.log(_cdf((centered + 0.5) / sigma_0_cat[mask]) - _cdf((centered - 0.5) / sigma_0_cat[mask]) + epsilon)
    logp = logp - torch.logsumexp(logp, dim=1, keepdim=True)
    return _sum_except_batch(logp * onehot, mask, n)


def loss_terms(m, ligand, pocket, t_int, noise, training, size_log_prob=None):
    """`forward(ligand, pocket)` of ConditionalDDPM (conditional_model.py:202-330) or
    EnVariationalDiffusion (en_diffusion.py:336-469) -> the reference's 12-tuple
    (delta_log_px, error_t_lig, error_t_pocket, SNR_weight, loss_0_x_ligand, loss_0_x_pocket,
     loss_0_h, neg_log_constants, kl_prior, log_pN, t_int, xh_lig_hat).
    `ligand` / `pocket`: un-normalised dicts (not modified); t_int [B,1] float (the reference draws
    torch.randint(lowest_t, T+1)); noise: callable(shape)."""
    nd, T = m.n_dims, m.T
    ligand = {k: v.clone() for k, v in ligand.items()}
    pocket = {k: v.clone() for k, v in pocket.items()}
    ligand, pocket = normalize(m, ligand, pocket)
    lm, pm = ligand["mask"], pocket["mask"]
    n = len(ligand["size"])
    cond = m.conditional
    n_nodes = ligand["size"] if cond else ligand["size"] + pocket["size"]
    delta_log_px = -_subspace_dim(m, n_nodes) * np.log(m.norm_values[0])          # :198-200 / :332-334
    s_int = t_int - 1
    t_is_zero = (t_int == 0).float()
    s, t = s_int / T, t_int / T
    g_s, g_t = m.g(s), m.g(t)
    xh_l = torch.cat([ligand["x"], ligand["one_hot"]], 1)
    xh_p = torch.cat([pocket["x"], pocket["one_hot"]], 1)

    def noised(g):
        if cond:
            z, xp, eps = cond_noised_representation(m, xh_l, xh_p, lm, pm, g, noise, n)
            return z, xp, eps, None
        n_l, n_p = joint_noise(m, lm, pm, noise)                                     # en_diffusion.py:302-317
        return (alpha(g)[lm] * xh_l + sigma(g)[lm] * n_l, alpha(g)[pm] * xh_p + sigma(g)[pm] * n_p, n_l, n_p)

    if cond:                                                                         # :232-236
        xl, xp = cond_remove_mean(m, xh_l[:, :nd], xh_p[:, :nd], lm, pm, n)
        xh_l = torch.cat([xl, xh_l[:, nd:]], 1)
        xh_p = torch.cat([xp, xh_p[:, nd:]], 1)
    z_l, z_p, eps_l, eps_p = noised(g_t)
    net_l, net_p = m.dynamics(z_l, z_p, t, lm, pm)
    a_t, s_t = alpha(g_t), sigma(g_t)
    xh_lig_hat = z_l / a_t[lm] - net_l * s_t[lm] / a_t[lm]                           # xh_given_zt_and_epsilon
    error_t_lig = _sum_except_batch((eps_l - net_l) ** 2, lm, n)
    error_t_pocket = torch.tensor(0.0) if cond else _sum_except_batch((eps_p - net_p) ** 2, pm, n)
    snr_weight = (1 - torch.exp(-(g_s - g_t))).squeeze(1)
    g0 = m.g(torch.zeros((n, 1)))
    neg_log_constants = -(_subspace_dim(m, n_nodes) * (-0.5 * g0.view(n) - 0.5 * np.log(2 * np.pi)))
    # KL(q(z_T | x) || N(0, 1))  (conditional_model.py:36-74, en_diffusion.py:109-155)
    g_T = m.g(torch.ones((n, 1)))
    a_T, s_T = alpha(g_T), sigma(g_T).squeeze()
    mu_l = a_T[lm] * xh_l
    mu2_h = _sum_except_batch(mu_l[:, nd:] ** 2, lm, n)
    mu2_x = _sum_except_batch(mu_l[:, :nd] ** 2, lm, n)
    if not cond:
        mu_p = a_T[pm] * xh_p
        mu2_h = mu2_h + _sum_except_batch(mu_p[:, nd:] ** 2, pm, n)
        mu2_x = mu2_x + _sum_except_batch(mu_p[:, :nd] ** 2, pm, n)
    ones = torch.ones_like(s_T)
    kl_prior = _gaussian_kl(mu2_x, s_T, ones, _subspace_dim(m, n_nodes)) + _gaussian_kl(mu2_h, s_T, ones, 1)

    def loss0(z_l0, z_p0, e_l0, e_p0, n_l0, n_p0, g):
        lx_l = 0.5 * _sum_except_batch((e_l0[:, :nd] - n_l0[:, :nd]) ** 2, lm, n)
        lh = -_log_ph_given_z0(m, ligand["one_hot"], z_l0[:, nd:], lm, g, n)
        if cond:
            return lx_l, torch.tensor(0.0), lh
        lx_p = 0.5 * _sum_except_batch((e_p0[:, :nd] - n_p0[:, :nd]) ** 2, pm, n)
        lh = lh - _log_ph_given_z0(m, pocket["one_hot"], z_p0[:, nd:], pm, g, n)
        return lx_l, lx_p, lh

    if training:
        l0_xl, l0_xp, l0_h = loss0(z_l, z_p, eps_l, eps_p, net_l, net_p, g_t)
        tz = t_is_zero.squeeze()
        l0_xl, l0_h = l0_xl * tz, l0_h * tz
        l0_xp = l0_xp if cond else l0_xp * tz
        error_t_lig = error_t_lig * (1 - tz)
        error_t_pocket = error_t_pocket if cond else error_t_pocket * (1 - tz)
    else:
        z_l0, z_p0, e_l0, e_p0 = noised(g0)
        n_l0, n_p0 = m.dynamics(z_l0, z_p0, torch.zeros_like(s), lm, pm)
        l0_xl, l0_xp, l0_h = loss0(z_l0, z_p0, e_l0, e_p0, n_l0, n_p0, g0)
    log_pN = size_log_prob(ligand["size"], pocket["size"]) if size_log_prob is not None else None
    return (delta_log_px, error_t_lig, error_t_pocket, snr_weight, l0_xl, l0_xp, l0_h, neg_log_constants,
            kl_prior, log_pN, t_int.squeeze(), xh_lig_hat)
