"""`EnVariationalDiffusion` -- drop-in for the *sampling* API of
/root/reference/equivariant_diffusion/en_diffusion.py:13-955 (joint ligand+pocket
diffusion): `sample`, `inpaint`, `sample_p_zs_given_zt`, `sample_p_zt_given_zs`,
`sample_p_xh_given_z0`, the predefined noise schedule and the helpers callers
use (`normalize`, `unnormalize`, `size_distribution`, `norm_values`, `T`, ...).

Per reverse step the work is: one `EGNNDynamics.forward_async` (HIP kernels)
+ one fused posterior-update kernel (`dsbdd_joint_reverse_update`) + the noise
draw; nothing in the loop synchronises with the host (the reference has >= 10
syncs per step).  The COM / NaN invariants the reference asserts
with `.item()` every step are checked once at the end of the chain from device
flags.

Out of scope: the training loss (`forward`, `kl_prior*`,
`log_pxh_given_z0_without_constants`, the learned `GammaNetwork`).  They raise
NotImplementedError here.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict

import numpy as np
import os

import torch
import torch.nn.functional as F
from torch import nn

from . import _lib

__all__ = ["EnVariationalDiffusion", "DistributionNodes", "PredefinedNoiseSchedule",
           "num_nodes_to_batch_mask"]


# ---------------------------------------------------------------------------
# small host-side helpers (torch as plumbing; none of this is per-step hot)
# ---------------------------------------------------------------------------
def num_nodes_to_batch_mask(n_samples, num_nodes, device):
    """utils.num_nodes_to_batch_mask (/root/reference/utils.py:146-154)."""
    assert isinstance(num_nodes, int) or len(num_nodes) == n_samples
    if isinstance(num_nodes, torch.Tensor):
        num_nodes = num_nodes.to(device)
    return torch.repeat_interleave(torch.arange(n_samples, device=device), num_nodes)


def seg_sum(src, index, n):
    out = torch.zeros((n,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    return out.index_add_(0, index, src)


def seg_mean(src, index, n):
    """torch_scatter.scatter_mean semantics (count clamped to >= 1)."""
    s = seg_sum(src, index, n)
    cnt = torch.zeros(n, dtype=src.dtype, device=src.device)
    cnt.index_add_(0, index, torch.ones(index.shape[0], dtype=src.dtype, device=src.device))
    return s / cnt.clamp(min=1).view((-1,) + (1,) * (s.dim() - 1))


# ---------------------------------------------------------------------------
# noise schedule (en_diffusion.py:1105-1190)
# ---------------------------------------------------------------------------
def _clip_noise_schedule(alphas2, clip_value=0.001):
    alphas2 = np.concatenate([np.ones(1), alphas2], axis=0)
    steps = np.clip(alphas2[1:] / alphas2[:-1], a_min=clip_value, a_max=1.0)
    return np.cumprod(steps, axis=0)


def _polynomial_alphas2(timesteps, s, power):
    n = timesteps + 1
    x = np.linspace(0, n, n)
    a2 = _clip_noise_schedule((1 - np.power(x / n, power)) ** 2, clip_value=0.001)
    return (1 - 2 * s) * a2 + s


def _cosine_alphas2(timesteps, s=0.008):
    n = timesteps + 2
    x = np.linspace(0, n, n)
    ac = np.cos(((x / n) + s) / (1 + s) * np.pi * 0.5) ** 2
    ac = ac / ac[0]
    betas = np.clip(1 - (ac[1:] / ac[:-1]), a_min=0, a_max=0.999)
    return np.cumprod(1.0 - betas, axis=0)


class PredefinedNoiseSchedule(nn.Module):
    """Lookup table gamma[0..T] = -(log alpha^2 - log sigma^2), built in fp64
    numpy and stored as an fp32 Parameter named `gamma` (state_dict key
    `gamma.gamma`), en_diffusion.py:1158-1190."""

    def __init__(self, noise_schedule, timesteps, precision):
        super().__init__()
        self.timesteps = timesteps
        if noise_schedule == 'cosine':
            alphas2 = _cosine_alphas2(timesteps)
        elif 'polynomial' in noise_schedule:
            parts = noise_schedule.split('_')
            assert len(parts) == 2
            alphas2 = _polynomial_alphas2(timesteps, s=precision, power=float(parts[1]))
        else:
            raise ValueError(noise_schedule)
        sigmas2 = 1 - alphas2
        table = -(np.log(alphas2) - np.log(sigmas2))
        self.gamma = nn.Parameter(torch.from_numpy(table).float(), requires_grad=False)

    def forward(self, t):
        return self.gamma[torch.round(t * self.timesteps).long()]


class _PositiveLinear(nn.Module):
    """Linear layer whose effective weight is softplus(weight) > 0 (en_diffusion.py:1030-1061)."""

    def __init__(self, in_features, out_features, weight_init_offset=-2):
        super().__init__()
        self.weight = nn.Parameter(torch.empty((out_features, in_features)))
        self.bias = nn.Parameter(torch.empty(out_features))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        with torch.no_grad():
            self.weight.add_(weight_init_offset)
        bound = 1 / math.sqrt(in_features)
        nn.init.uniform_(self.bias, -bound, bound)

    def forward(self, x):
        return F.linear(x, F.softplus(self.weight), self.bias)


class GammaNetwork(nn.Module):
    """Learned monotone noise schedule of the VDM paper (`noise_schedule='learned'`, en_diffusion.py:1064-1102):
    gamma(t) = gamma_0 + (gamma_1 - gamma_0) * (g(t) - g(0)) / (g(1) - g(0)), g(t) = l1(t) + l3(sigmoid(l2(l1(t))))
    with positive weights.  Plain PyTorch (a 1 -> 1024 -> 1 network evaluated on T + 1 scalars: not on the hot path);
    same parameter names as the reference, so a checkpoint trained with a learned schedule loads.  The samplers read
    the schedule as a table `gamma[0..T]` like the predefined schedules: `.gamma` evaluates the network at t = k / T."""

    def __init__(self, timesteps):
        super().__init__()
        self.timesteps = timesteps
        self.l1 = _PositiveLinear(1, 1)
        self.l2 = _PositiveLinear(1, 1024)
        self.l3 = _PositiveLinear(1024, 1)
        self.gamma_0 = nn.Parameter(torch.tensor([-5.]))
        self.gamma_1 = nn.Parameter(torch.tensor([10.]))
        self._table = None

    def gamma_tilde(self, t):
        l1_t = self.l1(t)
        return l1_t + self.l3(torch.sigmoid(self.l2(l1_t)))

    def forward(self, t):
        g0, g1, gt = self.gamma_tilde(torch.zeros_like(t)), self.gamma_tilde(torch.ones_like(t)), self.gamma_tilde(t)
        return self.gamma_0 + (self.gamma_1 - self.gamma_0) * (gt - g0) / (g1 - g0)

    @property
    def gamma(self):
        """[T + 1] table of the current parameters (re-evaluated when a parameter changed)."""
        key = tuple((p.data_ptr(), p._version) for p in self.parameters())
        if self._table is None or self._table[0] != key:
            with torch.no_grad():
                t = torch.arange(self.timesteps + 1, device=self.gamma_0.device, dtype=torch.float32) / self.timesteps
                self._table = (key, self.forward(t.view(-1, 1)).view(-1).contiguous())
        return self._table[1]

    def table_key(self):
        """Changes whenever a parameter changed (consumers that cache values derived from `.gamma`)."""
        _ = self.gamma
        return self._table[0]


class DistributionNodes:
    """Joint categorical over (n_ligand_nodes, n_pocket_nodes) from a 2-D
    histogram; host side, once per batch (en_diffusion.py:958-1028)."""

    def __init__(self, histogram):
        hist = torch.as_tensor(np.asarray(histogram)).float() + 1e-3
        self.prob = hist / hist.sum()
        n1, n2 = self.prob.shape
        self.idx_to_n_nodes = torch.stack(torch.meshgrid(
            torch.arange(n1), torch.arange(n2), indexing='ij'), -1).view(-1, 2)
        self.n_nodes_to_idx = {(int(a), int(b)): i for i, (a, b) in enumerate(self.idx_to_n_nodes.tolist())}
        self.m = torch.distributions.Categorical(self.prob.view(-1), validate_args=True)
        self.n1_given_n2 = [torch.distributions.Categorical(self.prob[:, j], validate_args=True)
                            for j in range(n2)]
        self.n2_given_n1 = [torch.distributions.Categorical(self.prob[i, :], validate_args=True)
                            for i in range(n1)]
        self._tables = {}

    def sample(self, n_samples=1):
        idx = self.m.sample((n_samples,))
        nl, npk = self.idx_to_n_nodes[idx].T
        return nl, npk

    def sample_conditional(self, n1=None, n2=None):
        assert (n1 is None) ^ (n2 is None), "Exactly one input argument must be None"
        dists = self.n1_given_n2 if n2 is not None else self.n2_given_n1
        c = n2 if n2 is not None else n1
        return torch.tensor([dists[int(i)].sample() for i in c.tolist()], device=c.device)

    def log_prob(self, batch_n_nodes_1, batch_n_nodes_2):
        assert batch_n_nodes_1.dim() == 1 and batch_n_nodes_2.dim() == 1
        key = (2, str(batch_n_nodes_1.device))
        if key not in self._tables:          # the joint categorical's logits as an [n1][n2] table on the device
            self._tables[key] = self.m.logits.view(self.prob.shape).to(batch_n_nodes_1.device)
        return self._tables[key][batch_n_nodes_1.long(), batch_n_nodes_2.long()]

    def _table(self, which, device):
        """[n1][n2] log-probabilities of the conditional categoricals -- the `logits` the per-sample
        `Categorical.log_prob` calls gather from (same numbers), on `device`, built once: one gather per batch instead
        of a host loop over the samples with two device-to-host copies."""
        key = (which, str(device))
        if key not in self._tables:
            if which == 0:     # log p(n1 | n2)
                t = torch.stack([d.logits for d in self.n1_given_n2], dim=1)
            else:              # log p(n2 | n1)
                t = torch.stack([d.logits for d in self.n2_given_n1], dim=0)
            self._tables[key] = t.to(device)
        return self._tables[key]

    def log_prob_n1_given_n2(self, n1, n2):
        assert n1.dim() == 1 and n2.dim() == 1
        return self._table(0, n1.device)[n1.long(), n2.long()]

    def log_prob_n2_given_n1(self, n2, n1):
        assert n1.dim() == 1 and n2.dim() == 1
        return self._table(1, n1.device)[n1.long(), n2.long()]


# ---------------------------------------------------------------------------
class StepCoefficients:
    """Per-step scalars of the reverse process, computed once per chain on the
    host with the same fp32 torch formulas the reference evaluates per step on
    [B,1] tensors (en_diffusion.py:83-107,865-873; conditional_model.py:435-456):

        alpha_ts[s], c_eps[s] = sigma2_ts / alpha_ts / sigma_t, sigma[s] = sigma_ts sigma_s / sigma_t
        and t_value[s] = (s+1)/timesteps as the reference's int-tensor division gives it.
    """

    def __init__(self, gamma_table, T, timesteps):
        g = gamma_table.detach().float().cpu()
        idx = torch.arange(0, timesteps + 1, dtype=torch.int64)
        tv = idx / timesteps                                    # float32, as s_array / timesteps
        gam = g[torch.round(tv * T).long()]
        g_s, g_t = gam[:-1], gam[1:]
        sigma2 = -torch.expm1(F.softplus(g_s) - F.softplus(g_t))
        alpha_ts = torch.exp(0.5 * (F.logsigmoid(-g_t) - F.logsigmoid(-g_s)))
        sigma_ts = torch.sqrt(sigma2)
        sig_s, sig_t = torch.sqrt(torch.sigmoid(g_s)), torch.sqrt(torch.sigmoid(g_t))
        self.t_value = tv                                       # [timesteps+1]
        self.gamma = gam
        self.alpha_ts = alpha_ts
        self.sigma_ts = sigma_ts
        self.c_eps = sigma2 / alpha_ts / sig_t
        self.sigma = sigma_ts * sig_s / sig_t
        self.alpha = torch.sqrt(torch.sigmoid(-gam))            # alpha_t per index
        self.sigma_t = torch.sqrt(torch.sigmoid(gam))

    def pair(self, s, t):
        """(alpha_{t|s}, sigma_{t|s}) between two level indices s < t (en_diffusion.py:83-107)."""
        g_s, g_t = self.gamma[s], self.gamma[t]
        sigma2 = -torch.expm1(F.softplus(g_s) - F.softplus(g_t))
        alpha_ts = torch.exp(0.5 * (F.logsigmoid(-g_t) - F.logsigmoid(-g_s)))
        return float(alpha_ts), float(torch.sqrt(sigma2))


def _chain(fn):
    """Sampling entry point: whatever happens, the chain state (engine pocket frame, edge bound, prefilled time word) of
    this call is released."""
    import functools

    @functools.wraps(fn)
    def wrapped(self, *args, **kwargs):
        try:
            return fn(self, *args, **kwargs)
        finally:
            self._end_chain()
    return wrapped


class EnVariationalDiffusion(nn.Module):
    """The E(n) diffusion module (joint ligand + pocket)."""

    def __init__(
            self,
            dynamics: nn.Module, atom_nf: int, residue_nf: int,
            n_dims: int, size_histogram: Dict,
            timesteps: int = 1000, parametrization='eps',
            noise_schedule='learned', noise_precision=1e-4,
            loss_type='vlb', norm_values=(1., 1.), norm_biases=(None, 0.),
            virtual_node_idx=None):
        super().__init__()
        assert loss_type in {'vlb', 'l2'}
        assert parametrization == 'eps'
        self.loss_type = loss_type
        if noise_schedule == 'learned':
            assert loss_type == 'vlb', 'A noise schedule can only be learned with a vlb objective.'
            self.gamma = GammaNetwork(timesteps)
        else:
            self.gamma = PredefinedNoiseSchedule(noise_schedule, timesteps=timesteps,
                                                 precision=noise_precision)
        self.dynamics = dynamics
        self.atom_nf = atom_nf
        self.residue_nf = residue_nf
        self.n_dims = n_dims
        self.num_classes = self.atom_nf
        self.T = timesteps
        self.parametrization = parametrization
        self.norm_values = norm_values
        self.norm_biases = norm_biases
        self.register_buffer('buffer', torch.zeros(1))
        self.size_distribution = DistributionNodes(size_histogram)
        self.vnode_idx = virtual_node_idx
        if noise_schedule != 'learned':       # en_diffusion.py:65: the check reads the table of a FIXED schedule only
            self.check_issues_norm_values()
        # noise: None -> sharding-invariant keyed Philox on the GPU; a callable
        # (shape) -> tensor injects external noise (parity tests)
        self.noise_source = None
        self._seed = None           # None: taken from torch's global RNG at the first draw of every sampling call
        self._user_seeded = False   # seed() was called: the key and the draw counter persist across calls
        self._sample_offset = 0
        self._sample_ids = None
        self._draw = 0
        self._coef_cache = {}
        self._chain = None          # (edge bound,) of the running chain

    # ---- noise -----------------------------------------------------------------
    def set_noise_source(self, fn):
        self.noise_source = fn

    def seed(self, seed, sample_offset=0, sample_ids=None):
        """Seed the keyed generator.  `sample_offset` = global index of this
        shard's first sample, so a chain's noise is independent of sharding;
        `sample_ids` (int64 [batch]) gives every sample of the batch an explicit
        global id instead (batches packed from several pockets, testset.py).
        Without a call to seed() the key is drawn from torch's global generator at
        the start of EVERY sampling call (`_begin_chain`), so `torch.manual_seed` /
        `pl.seed_everything` before a call control its samples, as they do for the
        reference (which draws everything from the global generator)."""
        self._seed, self._sample_offset, self._draw = int(seed), int(sample_offset), 0
        self._user_seeded = True
        self._sample_ids = None if sample_ids is None else torch.as_tensor(sample_ids, dtype=torch.int64)

    def _randn(self, mask, n_cols, batch, stream_id=0):
        n = mask.numel()
        if self.noise_source is not None:
            return self.noise_source((n, n_cols)).to(device=mask.device, dtype=torch.float32).contiguous()
        out = torch.empty((n, n_cols), dtype=torch.float32, device=mask.device)
        if self._seed is None:
            self._seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        ids = getattr(self, "_sample_ids", None)
        if ids is not None:
            if ids.numel() != batch:
                raise ValueError(f"seed(sample_ids=...) has {ids.numel()} entries for a batch of {batch}")
            if ids.device != mask.device:
                ids = self._sample_ids = ids.to(mask.device).contiguous()
        lib = _lib.load()
        _lib.check(lib.dsbdd_randn_keyed(
            torch.cuda.current_stream(mask.device).cuda_stream, out.data_ptr(), mask.data_ptr(), n,
            n_cols, batch, self._sample_offset, ids.data_ptr() if ids is not None else None,
            C.c_uint64(self._seed & (2 ** 64 - 1)), C.c_uint64(self._draw), stream_id), "dsbdd_randn_keyed")
        self._draw += 1
        return out

    @staticmethod
    def sample_gaussian(size, device):
        return torch.randn(size, device=device)

    # ---- fused per-sample kernels (csrc/ddpm.h): fixed reduction order, in place ----------------
    @staticmethod
    def _cs(t):
        return torch.cuda.current_stream(t.device).cuda_stream

    def _seg_mean3(self, x, mask, batch):
        """Per-sample mean of x[:, :3] (scatter_mean semantics) with a fixed summation order
        (torch's index_add_ uses float atomics: neither reproducible nor sharding-invariant)."""
        x = x.to(torch.float32).contiguous()
        if not x.is_cuda:
            raise _lib.HipLibraryError("_seg_mean3 runs on the HIP kernels: device tensors only (no CPU fallback)")
        mask = mask.to(device=x.device, dtype=torch.int64).contiguous()     # the kernel reads raw int64 device memory
        out = torch.zeros((batch, 3), dtype=torch.float32, device=x.device)
        if x.shape[0] == 0:
            return out
        _lib.check(_lib.load().dsbdd_segment_mean3(self._cs(x), x.data_ptr(), x.shape[1], mask.data_ptr(),
                                                   x.shape[0], batch, out.data_ptr()), "dsbdd_segment_mean3")
        return out

    def _joint_noise_raw(self, lig_mask, pocket_mask, batch):
        """Gaussian draws for both node sets, [n][3 + nf] each, x part NOT yet COM-centred (the
        kernels centre it).  Injected noise keeps the reference's draw order (x of all nodes, h_lig,
        h_pocket: en_diffusion.py:559-578); the keyed generator draws one block per node set."""
        nl = lig_mask.numel()
        if self.noise_source is not None:
            comb = torch.cat((lig_mask, pocket_mask))
            zx = self._randn(comb, self.n_dims, batch)
            zh_l = self._randn(lig_mask, self.atom_nf, batch)
            zh_p = self._randn(pocket_mask, self.residue_nf, batch)
            return (torch.cat([zx[:nl], zh_l], dim=1).contiguous(),
                    torch.cat([zx[nl:], zh_p], dim=1).contiguous())
        return (self._randn(lig_mask, self.n_dims + self.atom_nf, batch, stream_id=1),
                self._randn(pocket_mask, self.n_dims + self.residue_nf, batch, stream_id=2))

    def _joint_gauss_(self, z_l, z_p, lig_mask, pocket_mask, batch, a, sigma, remove_com):
        """In place: z <- a z + sigma eps with COM-free eps (+ joint COM removal of the result)."""
        n_l, n_p = self._joint_noise_raw(lig_mask, pocket_mask, batch)
        _lib.check(_lib.load().dsbdd_joint_affine_noise(
            self._cs(z_l), z_l.data_ptr(), z_p.data_ptr(), n_l.data_ptr(), n_p.data_ptr(), lig_mask.data_ptr(),
            pocket_mask.data_ptr(), lig_mask.numel(), pocket_mask.numel(), batch, self.atom_nf, self.residue_nf,
            float(a), float(sigma), 1, int(remove_com)), "dsbdd_joint_affine_noise")

    # ---- schedule helpers (same names as the reference) --------------------------
    def check_issues_norm_values(self, num_stdevs=8):
        gamma_0 = self.gamma(torch.zeros((1, 1)))
        sigma_0 = self.sigma(gamma_0, target_tensor=torch.zeros((1, 1))).item()
        norm_value = self.norm_values[1]
        if sigma_0 * num_stdevs > 1. / norm_value:
            raise ValueError(
                f'Value for normalization value {norm_value} probably too large with sigma_0 '
                f'{sigma_0:.5f} and 1 / norm_value = {1. / norm_value}')

    @staticmethod
    def inflate_batch_array(array, target):
        return array.view((array.size(0),) + (1,) * (len(target.size()) - 1))

    def sigma(self, gamma, target_tensor):
        return self.inflate_batch_array(torch.sqrt(torch.sigmoid(gamma)), target_tensor)

    def alpha(self, gamma, target_tensor):
        return self.inflate_batch_array(torch.sqrt(torch.sigmoid(-gamma)), target_tensor)

    @staticmethod
    def SNR(gamma):
        return torch.exp(-gamma)

    def sigma_and_alpha_t_given_s(self, gamma_t, gamma_s, target_tensor):
        sigma2 = self.inflate_batch_array(
            -torch.expm1(F.softplus(gamma_s) - F.softplus(gamma_t)), target_tensor)
        alpha_ts = self.inflate_batch_array(
            torch.exp(0.5 * (F.logsigmoid(-gamma_t) - F.logsigmoid(-gamma_s))), target_tensor)
        return sigma2, torch.sqrt(sigma2), alpha_ts

    def subspace_dimensionality(self, input_size):
        return (input_size - 1) * self.n_dims

    def _coefs(self, timesteps) -> StepCoefficients:
        # a learned schedule rebuilds its table after every parameter update -- possibly at the same address with
        # version 0 -- so the key carries the generation of the table (GammaNetwork.table_key), not only its pointer
        gen = self.gamma.table_key() if hasattr(self.gamma, "table_key") else None
        key = (timesteps, self.gamma.gamma.data_ptr(), self.gamma.gamma._version, gen)
        if key not in self._coef_cache:
            self._coef_cache = {key: StepCoefficients(self.gamma.gamma, self.T, timesteps)}
        return self._coef_cache[key]

    # ---- normalisation (en_diffusion.py:880-912; in place on the dicts) ---------
    def normalize(self, ligand=None, pocket=None):
        for d in (ligand, pocket):
            if d is not None:
                d['x'] = d['x'] / self.norm_values[0]
                d['one_hot'] = (d['one_hot'].float() - self.norm_biases[1]) / self.norm_values[1]
        return ligand, pocket

    def unnormalize(self, x, h_cat):
        return x * self.norm_values[0], h_cat * self.norm_values[1] + self.norm_biases[1]

    def unnormalize_z(self, z_lig, z_pocket):
        nd = self.n_dims
        x_l, h_l = self.unnormalize(z_lig[:, :nd], z_lig[:, nd:])
        x_p, h_p = self.unnormalize(z_pocket[:, :nd], z_pocket[:, nd:])
        return torch.cat([x_l, h_l], dim=1), torch.cat([x_p, h_p], dim=1)

    @staticmethod
    def remove_mean_batch(x, indices, n=None):
        if n is None:
            n = int(indices.max()) + 1 if indices.numel() else 0
        return x - seg_mean(x, indices, n)[indices]

    @staticmethod
    def assert_mean_zero_with_mask(x, node_mask, eps=1e-10):
        largest = x.abs().max().item()
        n = int(node_mask.max()) + 1
        err = seg_sum(x, node_mask, n).abs().max().item()
        rel = err / (largest + eps)
        assert rel < 1e-2, f'Mean is not zero, relative_error {rel}'

    @staticmethod
    def sample_center_gravity_zero_gaussian_batch(size, lig_indices, pocket_indices):
        assert len(size) == 2
        x = torch.randn(size, device=lig_indices.device)
        return EnVariationalDiffusion.remove_mean_batch(x, torch.cat((lig_indices, pocket_indices)))

    # ---- loss terms (en_diffusion.py:109-262, 336-469) --------------------------------------
    # Evaluation (eval mode / no_grad: validation_step, likelihood estimates): the network passes run on the HIP
    # kernels.  Training step (training mode with autograd recording): the network passes go through the autograd
    # Functions of train_hip.py (HIP forward and backward kernels; DSBDD_TRAIN=torch: the eager path of train_path.py) and
    # every loss term carries its graph, so `loss.backward()` reaches the parameters (lightning_modules.py:337-363).
    t_int_source = None          # optional callable(batch) -> [B,1] float tensor (tests); default torch.randint

    def _draw_t_int(self, batch, device):
        lowest_t = 0 if self.training else 1                               # en_diffusion.py:349-352
        if self.t_int_source is not None:
            return self.t_int_source(batch).to(device=device, dtype=torch.float32).view(batch, 1)
        return torch.randint(lowest_t, self.T + 1, size=(batch, 1), device=device).float()

    def _loss_context(self):
        """no_grad for evaluation; the autograd graph is kept for a training step."""
        import contextlib
        return contextlib.nullcontext() if (self.training and torch.is_grad_enabled()) else torch.no_grad()

    @staticmethod
    def gaussian_KL(q_mu_minus_p_mu_squared, q_sigma, p_sigma, d):
        """en_diffusion.py:838-853."""
        return d * torch.log(p_sigma / q_sigma) + \
            0.5 * (d * q_sigma ** 2 + q_mu_minus_p_mu_squared) / (p_sigma ** 2) - 0.5 * d

    def log_constants_p_x_given_z0(self, n_nodes, device):
        """en_diffusion.py:171-184."""
        batch_size = len(n_nodes)
        dof = self.subspace_dimensionality(n_nodes)
        gamma_0 = self.gamma(torch.zeros((batch_size, 1), device=device))
        log_sigma_x = 0.5 * gamma_0.view(batch_size)
        return dof * (-log_sigma_x - 0.5 * np.log(2 * np.pi))

    def _log_ph_given_z0(self, one_hot, z_h, mask, gamma_0, epsilon=1e-10):
        """Categorical part of log_pxh_given_z0_without_constants (en_diffusion.py:214-260)."""
        nv, nb = self.norm_values[1], self.norm_biases[1]
        sigma_0_cat = self.sigma(gamma_0, target_tensor=z_h) * nv
        onehot = one_hot * nv + nb
        centered = (z_h * nv + nb) - 1
        logp = torch.log(self.cdf_standard_gaussian((centered + 0.5) / sigma_0_cat[mask])
                         - self.cdf_standard_gaussian((centered - 0.5) / sigma_0_cat[mask]) + epsilon)
        logp = logp - torch.logsumexp(logp, dim=1, keepdim=True)
        return self.sum_except_batch(logp * onehot, mask, gamma_0.shape[0])

    def kl_prior_with_pocket(self, xh_lig, xh_pocket, mask_lig, mask_pocket, num_nodes):
        """KL(q(z_T | x) || N(0, 1)), en_diffusion.py:109-155."""
        nd = self.n_dims
        ones = torch.ones((len(num_nodes), 1), device=xh_lig.device)
        gamma_T = self.gamma(ones)
        alpha_T = self.alpha(gamma_T, xh_lig)
        sigma_T = self.sigma(gamma_T, xh_lig).squeeze()
        mu_l, mu_p = alpha_T[mask_lig] * xh_lig, alpha_T[mask_pocket] * xh_pocket
        one = torch.ones_like(sigma_T)
        nb = len(num_nodes)
        mu2_h = self.sum_except_batch(mu_l[:, nd:] ** 2, mask_lig, nb) + self.sum_except_batch(mu_p[:, nd:] ** 2, mask_pocket, nb)
        mu2_x = self.sum_except_batch(mu_l[:, :nd] ** 2, mask_lig, nb) + self.sum_except_batch(mu_p[:, :nd] ** 2, mask_pocket, nb)
        return self.gaussian_KL(mu2_x, sigma_T, one, self.subspace_dimensionality(num_nodes)) + \
            self.gaussian_KL(mu2_h, sigma_T, one, d=1)

    def _to_device(self, d, dev):
        for k in ('x', 'one_hot', 'size', 'mask'):
            d[k] = d[k].to(dev)
        d['mask'] = d['mask'].to(torch.int64).contiguous()
        return d

    def forward(self, ligand, pocket, return_info=False):
        """The reference's loss terms (en_diffusion.py:336-469), same 12-tuple (+ info):
        (delta_log_px, error_t_lig, error_t_pocket, SNR_weight, loss_0_x_ligand, loss_0_x_pocket,
         loss_0_h, neg_log_constants, kl_prior, log_pN, t_int, xh_lig_hat)."""
        with self._loss_context():
            dev = self._hip_device(None)
            ligand, pocket = self._to_device(ligand, dev), self._to_device(pocket, dev)
            ligand, pocket = self.normalize(ligand, pocket)
            lm, pm = ligand['mask'], pocket['mask']
            n = ligand['size'].size(0)
            nd = self.n_dims
            n_nodes = ligand['size'] + pocket['size']
            delta_log_px = self.delta_log_px(n_nodes)
            t_int = self._draw_t_int(n, dev)
            s_int = t_int - 1
            t_is_zero = (t_int == 0).float()
            t_is_not_zero = 1 - t_is_zero
            s, t = s_int / self.T, t_int / self.T
            gamma_s = self.inflate_batch_array(self.gamma(s), ligand['x'])
            gamma_t = self.inflate_batch_array(self.gamma(t), ligand['x'])
            xh_lig = torch.cat([ligand['x'], ligand['one_hot']], dim=1)
            xh_pocket = torch.cat([pocket['x'], pocket['one_hot']], dim=1)
            z_l, z_p, eps_l, eps_p = self.noised_representation(xh_lig, xh_pocket, lm, pm, gamma_t)
            net_l, net_p = self.dynamics(z_l.contiguous(), z_p.contiguous(), t, lm, pm)
            xh_lig_hat = self.xh_given_zt_and_epsilon(z_l, net_l, gamma_t, lm)
            error_t_lig = self.sum_except_batch((eps_l - net_l) ** 2, lm, n)
            error_t_pocket = self.sum_except_batch((eps_p - net_p) ** 2, pm, n)
            SNR_weight = (1 - self.SNR(gamma_s - gamma_t)).squeeze(1)
            neg_log_constants = -self.log_constants_p_x_given_z0(n_nodes=n_nodes, device=dev)
            kl_prior = self.kl_prior_with_pocket(xh_lig, xh_pocket, lm, pm, n_nodes)

            def loss0(zl, zp, el, ep, nl, np_, g):
                lx_l = 0.5 * self.sum_except_batch((el[:, :nd] - nl[:, :nd]) ** 2, lm, n)
                lx_p = 0.5 * self.sum_except_batch((ep[:, :nd] - np_[:, :nd]) ** 2, pm, n)
                lh = -(self._log_ph_given_z0(ligand['one_hot'], zl[:, nd:], lm, g)
                       + self._log_ph_given_z0(pocket['one_hot'], zp[:, nd:], pm, g))
                return lx_l, lx_p, lh

            if self.training:
                tz = t_is_zero.squeeze()
                l0_xl, l0_xp, l0_h = loss0(z_l, z_p, eps_l, eps_p, net_l, net_p, gamma_t)
                l0_xl, l0_xp, l0_h = l0_xl * tz, l0_xp * tz, l0_h * tz
                error_t_lig = error_t_lig * t_is_not_zero.squeeze()
                error_t_pocket = error_t_pocket * t_is_not_zero.squeeze()
            else:                                   # separate pass at t = 0 (en_diffusion.py:426-446)
                t_zeros = torch.zeros_like(s)
                gamma_0 = self.inflate_batch_array(self.gamma(t_zeros), ligand['x'])
                z0_l, z0_p, e0_l, e0_p = self.noised_representation(xh_lig, xh_pocket, lm, pm, gamma_0)
                n0_l, n0_p = self.dynamics(z0_l.contiguous(), z0_p.contiguous(), t_zeros, lm, pm)
                l0_xl, l0_xp, l0_h = loss0(z0_l, z0_p, e0_l, e0_p, n0_l, n0_p, gamma_0)
            log_pN = self.log_pN(ligand['size'], pocket['size'])
            info = {
                'eps_hat_lig_x': seg_mean(net_l[:, :nd].abs().mean(1), lm, n).mean(),
                'eps_hat_lig_h': seg_mean(net_l[:, nd:].abs().mean(1), lm, n).mean(),
                'eps_hat_pocket_x': seg_mean(net_p[:, :nd].abs().mean(1), pm, n).mean(),
                'eps_hat_pocket_h': seg_mean(net_p[:, nd:].abs().mean(1), pm, n).mean(),
            }
            loss_terms = (delta_log_px, error_t_lig, error_t_pocket, SNR_weight, l0_xl, l0_xp, l0_h,
                          neg_log_constants, kl_prior, log_pN, t_int.squeeze(), xh_lig_hat)
        return (*loss_terms, info) if return_info else loss_terms

    # ---- dynamics call ---------------------------------------------------------------
    def _check_status(self, status):
        st = int(status.item())
        if st & _lib.STATUS_EDGE_OVERFLOW:
            raise RuntimeError("edge capacity overflow in the EGNN kernels")
        if st & _lib.STATUS_NAN:
            raise ValueError("NaN detected in EGNN output")

    share_identical_pockets = True   # evaluate block 0's pocket-pocket messages once for a batch of identical pockets
    # forward cone of the engine (csrc/engine.hip): 2 = on, 0 = off (1 = the engine's own cost model, for direct C-API
    # callers).  Cone on / off differ in rounding (the canonical pocket is evaluated on the raw pocket coordinates), so the
    # mode is decided HERE, once per chain and from the pocket groups alone (`_cone_for_groups`: on while the distinct
    # pockets are at most 0.2 of the batch (measured break-even at B = 64: 12 groups, profiles/r4n_cone_rule.md) -- one pocket repeated, the reference's generate_ligands / test.py case: on;
    # every pocket different: off), never by an engine heuristic in the middle of a chain.  A driver that wants
    # bit-identical molecules for ANY packing of pockets into batches pins the mode (testset.make_hip_sampler: 2 for
    # the duration of a batch); the environment variable DSBDD_CONE overrides both.
    cone_mode = None

    @staticmethod
    def _cone_for_groups(rep, batch):
        """2 (on) / 0 (off) from representative[b] (None: every sample its own pocket)."""
        n_groups = batch if rep is None else int(torch.unique(rep).numel())
        return 2 if 5 * n_groups <= batch else 0
    # 16-edge-granule edge kernels (csrc/edge_wave16.h, include/diffsbdd_hip.h DSBDD_OPT_GRANULE16): bit mask of the stages
    # that use them; None leaves the engine's setting (default 0, environment DSBDD_GRANULE16) alone.  The variants agree to
    # rounding, so the mask is part of a chain's definition like `cone_mode`: set once, never changed by the engine.
    edge_granule16 = None

    @staticmethod
    def granule16_auto(lig_mask, batch, n_blocks, n_mlp=2, n_cu=256, pocket_mask=None):
        """`edge_granule16 = "auto"`: the coordinate stages go to the 16-edge kernels when halving the work unit saves at
        least a fifth of the stage -- workgroup items of 128 edges (32-edge kernel, one per (tile, MLP)) against items
        of 64 edges at half the time each: ceil(items64 / n_cu) / 2 <= 0.8 ceil(items128 / n_cu).  The ligand-row edge
        count is only known on the device; the rule must hold for both ends of a host-side bracket -- the complete
        ligand graph alone, and with n_pocket / 24 (at most 12) pocket neighbours per ligand atom on top: the 3rfm pockets
        give 0.5 (C-alpha, 36 nodes) and 10 (full-atom, 286 nodes) per atom (one host sync per chain).
        Measured (profiles/r4c_granule16_ab.md): crossdock_ca_cond x 32 (274 -> 548 items) 52.1 -> 54.7 ligands/s; the
        full-atom headline (748 items: 3 rounds either way) is slower on them and keeps the 32-edge kernels.  Message
        stages are left alone: their edge counts are device-side and the two kernels are within 1 % there."""
        nl = torch.bincount(lig_mask, minlength=batch).to(torch.int64)
        lo = int((nl * nl).sum().item())
        hi = lo
        if pocket_mask is not None:
            npk = torch.bincount(pocket_mask, minlength=batch).to(torch.int64)
            hi = int((nl * (nl + torch.clamp((npk + 23) // 24, max=12))).sum().item())

        def pays(e_u):
            items128 = n_mlp * ((e_u + 127) // 128)
            items64 = n_mlp * ((e_u + 63) // 64)
            return -(-items64 // n_cu) * 0.5 <= 0.8 * -(-items128 // n_cu)
        return (((1 << n_blocks) - 1) << 16) if (pays(lo) and pays(hi)) else 0

    # Split-K edge kernels (csrc/edge_splitk.h, include/diffsbdd_hip.h DSBDD_OPT_SPLITK): bit mask of the stages that use them
    # (same layout as the granule mask), "auto" (`splitk_auto`: decided once per chain from the batch's sizes), or None =
    # leave the engine's setting (default 0, environment DSBDD_SPLITK) alone.  The variants agree to rounding, so the mask
    # is part of a chain's definition like `cone_mode`: set once per chain, never changed by the engine.
    edge_splitk = None

    @staticmethod
    def splitk_auto(lig_mask, pocket_mask, batch, n_stages, n_blocks, n_mlp, hidden_nf, pocket_x=None, cutoff_pocket=None,
                    n_cu=256):
        """`edge_splitk = "auto"`: the stages whose LARGEST possible launch is at most 1.5 rounds of the default kernel's
        128-edge workgroup tiles on the chip's CUs go to the split-K kernels (a quarter of the work unit, the same vector
        work per MFMA: csrc/edge_splitk.h).  Measured (profiles/r6b_mbsk.md): below that the default kernel pays a whole
        27-us wave tile per CU whatever the launch holds (20.8 k edges: 43 -> 37 us, 11.6 k: 41 -> 25 us, 5 k: 41 -> 14 us;
        coordinate stage of 16 full-atom samples 72 -> 55 us), above ~2 rounds it is 3 - 10 % ahead (0.65 - 0.74 of the
        peak against 0.60 - 0.68).  Decided ONCE PER CHAIN from host-side bounds, never from a device-side count, so that
        a sample's bits do not depend on the batch it is evaluated in:
            E   <= sum_b  n_l^2 + 2 n_l c_b + PP_b      (message stages; PP_b = pocket-pocket edges of the chain's frame,
                                                         counted exactly on the device when `pocket_x` is given -- the
                                                         pocket is rigid during a chain --, else the complete graph)
            E_u <= sum_b  n_l^2 + n_l c_b               (coordinate stages: ligand rows)
        with c_b = min(ceil(n_p / 24), 12) pocket neighbours per ligand atom (the bracket of `granule16_auto`).
        crossdock_ca_cond x 32: every stage (22 k / 18 k edges); full-atom x 16: the coordinate stages; full-atom x 64:
        none.  hidden_nf 256 only (the kernels' constraint).  One host sync per chain."""
        if hidden_nf != 256:
            return 0
        nl = torch.bincount(lig_mask, minlength=batch).to(torch.int64)
        npk = torch.bincount(pocket_mask, minlength=batch).to(torch.int64)
        c = torch.clamp((npk + 23) // 24, max=12)
        ll, lp = (nl * nl).sum(), (nl * c).sum()
        pp = (npk * npk).sum()
        if pocket_x is not None and cutoff_pocket is not None and int(npk.min()) == int(npk.max()) and int(npk[0]) > 0:
            xb = pocket_x.reshape(batch, int(npk[0]), -1)[:, :, :3].float()
            pp = (torch.cdist(xb, xb, compute_mode="donot_use_mm_for_euclid_dist") <= float(cutoff_pocket)).sum()
        e_hi, e_u_hi = (torch.stack([ll + 2 * lp + pp, ll + lp])).tolist()

        def tiles(e):
            return -(-int(e) // 128)
        limit = 3 * n_cu // 2
        mask = ((1 << n_stages) - 1) if tiles(e_hi) <= limit else 0
        if n_mlp * tiles(e_u_hi) <= limit:
            mask |= ((1 << n_blocks) - 1) << 16
        return mask

    # Arithmetic of the H x H layer of the fused edge kernels (include/diffsbdd_hip.h DSBDD_OPT_EMU; csrc/edge_wave.h,
    # "emulated path"): None leaves the engine's setting alone (default 0 = exact fp32 MFMA; environment DSBDD_EMU),
    # 6 / 9 = fp32 emulated on the bf16 matrix cores (three-way bf16 split of both operands, 6 / 9 partial products, fp32
    # accumulate).  Opt-in; part of a chain's definition like the granule mask (the two paths differ in rounding).
    edge_emulation = None

    def _apply_engine_option(self, which, value, default_env):
        """value None = "not set by this module": if an earlier chain of this module wrote the option, restore the
        engine's default (the environment variable, else 0) -- ADVICE r4: a stale mask must not outlive the attribute."""
        eng = self.dynamics.engine()
        if value is None:
            if which not in getattr(eng, "_options", {}):
                return
            env = os.environ.get(default_env)
            value = int(env, 0) if env not in (None, "") else 0
        eng.set_option(which, value)

    frame_min_pocket_nodes = 128     # pockets smaller than this (C-alpha models) keep the single-list block 0: the
                                     # extra launches of the split cost more than their few pocket-pocket edges

    def _begin_chain(self, lig_mask, pocket_mask, batch, pocket=None):
        """Start of a sampling call: int64 contiguous masks on the device and the edge bound of
        this batch, from the mask CONTENTS (one host sync per chain, which also rejects unsorted
        masks before any kernel runs).  `pocket` (normalised dict, pocket-conditioned models): the
        pocket only translates during the chain, so the engine gets its raw coordinates as a frame
        (csrc/engine.hip, dsbdd_engine_set_pocket_frame) -- and, when all samples carry the same
        pocket (prepare_pocket(repeats=n), lightning_modules.py:738-750), block 0's pocket-pocket
        messages are evaluated for one sample only; results are bit-identical either way."""
        from .engine import edge_capacity
        dev = self._hip_device(None)
        if not getattr(self, "_user_seeded", False):
            self._seed, self._draw = None, 0          # a fresh key from torch's generator for this call
        lm = lig_mask.to(device=dev, dtype=torch.int64).contiguous()
        pm = pocket_mask.to(device=dev, dtype=torch.int64).contiguous()
        cap = edge_capacity(lm, pm, batch)
        self._chain = (cap,)
        self._framed = False
        g16 = None
        if self.edge_granule16 is not None:
            if self.edge_granule16 == "auto":
                hp = self.dynamics._hp
                g16 = self.granule16_auto(lm, batch, hp["n_layers"], 1 if hp["reflection_equivariant"] else 2,
                                          pocket_mask=pm) \
                    if not self.dynamics.update_pocket_coords else 0
            else:
                g16 = int(self.edge_granule16) & 0xFFFFFFFF
            g16 = g16 - (1 << 32) if g16 >= (1 << 31) else g16
        self._apply_engine_option(_lib.OPT_GRANULE16, g16, "DSBDD_GRANULE16")
        sk = None
        if self.edge_splitk is not None:
            if self.edge_splitk == "auto":
                hp = self.dynamics._hp
                sk = 0 if self.dynamics.update_pocket_coords else self.splitk_auto(
                    lm, pm, batch, hp["n_layers"] * hp["inv_sublayers"], hp["n_layers"],
                    1 if hp["reflection_equivariant"] else 2, hp["hidden_nf"],
                    pocket_x=pocket["x"].to(dev) if pocket is not None else None, cutoff_pocket=hp["edge_cutoff_pocket"])
            else:
                sk = int(self.edge_splitk) & 0xFFFFFFFF
            sk = sk - (1 << 32) if sk >= (1 << 31) else sk
        self._apply_engine_option(_lib.OPT_SPLITK, sk, "DSBDD_SPLITK")
        emu = None if self.edge_emulation is None else int(self.edge_emulation)
        if emu not in (None, 0, 6, 9):
            raise ValueError("edge_emulation must be None, 0, 6 or 9")
        self._apply_engine_option(_lib.OPT_EMU, emu, "DSBDD_EMU")
        if pocket is not None and not self.dynamics.update_pocket_coords and pm.numel() > 0 and \
                int(pocket['size'].min()) >= self.frame_min_pocket_nodes:
            # (the size rule looks at every sample's pocket: in practice a property of the model's pocket
            # representation -- full-atom pockets have hundreds of nodes, C-alpha pockets a few dozen)
            sizes = pocket['size'].to(dev).to(torch.int64)
            x = pocket['x'].to(device=dev, dtype=torch.float32)
            rep = None
            if self.share_identical_pockets:
                rep = self._pocket_groups(x, pocket['one_hot'].to(dev), sizes, batch)
            self.dynamics.engine().set_pocket_frame(x, pm, sizes, lm.numel(), batch, cap, representative=rep)
            mode = self.cone_mode if self.cone_mode is not None else self._cone_for_groups(rep, batch)
            env = os.environ.get("DSBDD_CONE")
            self.dynamics.engine().set_option(_lib.OPT_CONE, int(env) if env not in (None, "") else int(mode))
            self._framed = True
        return lm, pm

    @staticmethod
    def _pocket_groups(x, one_hot, sizes, batch):
        """representative[b] = first sample of the batch whose pocket (atom count, coordinates, features) is identical
        to sample b's.  The common case -- one pocket repeated (prepare_pocket(repeats=n)) -- is decided on the device;
        otherwise the pockets are hashed on the host (one copy of the pocket array per chain)."""
        if bool((sizes == sizes[0]).all()) and int(sizes[0]) * batch == x.shape[0]:
            n0 = int(sizes[0])
            if bool((x.view(batch, n0, -1) == x[:n0]).all()) and bool((one_hot.view(batch, n0, -1) == one_hot[:n0]).all()):
                return torch.zeros(batch, dtype=torch.int64)
        import hashlib
        xs, hs = x.cpu().numpy(), one_hot.float().cpu().numpy()
        ends = torch.cumsum(sizes, 0).cpu().tolist()
        first, rep, lo = {}, [], 0
        for b, hi in enumerate(ends):
            key = hashlib.blake2b(xs[lo:hi].tobytes() + hs[lo:hi].tobytes(), digest_size=16).digest()
            rep.append(first.setdefault(key, b))
            lo = hi
        return torch.tensor(rep, dtype=torch.int64)

    def _end_chain(self):
        """End of a sampling call: the pocket frame is released and the module leaves the "in chain" state, so that a
        denoiser call outside a chain validates the packed weights again (`_dyn`: in_chain) and no stale prefilled time
        word survives."""
        if getattr(self, "_framed", False):
            self.dynamics.engine().clear_pocket_frame()
            self._framed = False
        self._chain = None
        self._t_prefilled = None

    def _dyn(self, z_lig, z_pocket, t_value, lig_mask, pocket_mask, batch, status, want_pocket):
        """One denoiser call.  t and the eps outputs live in persistent buffers keyed by the
        state tensors, so that consecutive reverse steps of a chain call the engine with
        identical pointers (the engine replays its captured hipGraph)."""
        key = (z_lig.data_ptr(), z_pocket.data_ptr(), batch, bool(want_pocket), z_lig.shape, z_pocket.shape)
        buf = self._dyn_bufs.get(key) if hasattr(self, "_dyn_bufs") else None
        if buf is None:
            if not hasattr(self, "_dyn_bufs") or len(self._dyn_bufs) > 16:
                self._dyn_bufs = {}
            # one t for the whole batch (dynamics.py:105-107 accepts it; the engine's forward cone requires it)
            buf = (torch.empty((1,), dtype=torch.float32, device=z_lig.device),
                   torch.empty_like(z_lig), torch.empty_like(z_pocket) if want_pocket else None)
            self._dyn_bufs[key] = buf
        t, eps_l, eps_p = buf
        pre = getattr(self, "_t_prefilled", None)
        if pre != (t.data_ptr(), float(t_value)):        # (the previous step's fused kernel may have written it already)
            t.fill_(float(t_value))
        self._t_prefilled = None
        self._last_t = t
        return self.dynamics.forward_async(z_lig, z_pocket, t, lig_mask, pocket_mask, status=status,
                                           want_pocket=want_pocket, batch=batch, eps_lig=eps_l,
                                           eps_pocket=eps_p,
                                           edge_cap=self._chain[0] if self._chain is not None else None,
                                           in_chain=self._chain is not None)

    # ---- joint noise (en_diffusion.py:559-578) ------------------------------------------
    def sample_combined_position_feature_noise(self, lig_indices, pocket_indices):
        batch = int(max(lig_indices.max(), pocket_indices.max())) + 1
        return self._joint_noise(lig_indices, pocket_indices, batch)

    def _joint_noise(self, lig_mask, pocket_mask, batch):
        """COM-free joint noise as tensors (public helpers; the sampling loops centre inside the kernels)."""
        n_l, n_p = self._joint_noise_raw(lig_mask, pocket_mask, batch)
        nd = self.n_dims
        comb = torch.cat((lig_mask, pocket_mask))
        zx = torch.cat((n_l[:, :nd], n_p[:, :nd]))
        zx = zx - seg_mean(zx, comb, batch)[comb]
        nl = lig_mask.numel()
        return torch.cat([zx[:nl], n_l[:, nd:]], dim=1), torch.cat([zx[nl:], n_p[:, nd:]], dim=1)

    # ---- one reverse step, joint model (en_diffusion.py:503-557) ---------------------------
    def _joint_step(self, s, co, z_lig, z_pocket, lig_mask, pocket_mask, batch, status):
        """In place: z_* at level s+1 -> level s."""
        eps_l, eps_p, _ = self._dyn(z_lig, z_pocket, co.t_value[s + 1], lig_mask, pocket_mask, batch,
                                    status, True)
        n_l, n_p = self._joint_noise_raw(lig_mask, pocket_mask, batch)
        lib = _lib.load()
        _lib.check(lib.dsbdd_joint_reverse_update(
            torch.cuda.current_stream(z_lig.device).cuda_stream, z_lig.data_ptr(), z_pocket.data_ptr(),
            eps_l.data_ptr(), eps_p.data_ptr(), n_l.data_ptr(), n_p.data_ptr(), lig_mask.data_ptr(),
            pocket_mask.data_ptr(), lig_mask.numel(), pocket_mask.numel(), batch, self.atom_nf,
            self.residue_nf, float(co.alpha_ts[s]), float(co.c_eps[s]), float(co.sigma[s]), 1),
            "dsbdd_joint_reverse_update")

    @_chain
    def sample_p_zs_given_zt(self, s, t, zt_lig, zt_pocket, ligand_mask, pocket_mask, fix_noise=False):
        """Functional form with the reference's signature: s, t are [B,1]
        tensors (all entries equal, as every caller passes them)."""
        if fix_noise:
            raise NotImplementedError("fix_noise option isn't implemented yet")
        batch = s.shape[0]
        timesteps, s_int = self._infer_step(s, t)
        co = self._coefs(timesteps)
        ligand_mask, pocket_mask = self._begin_chain(ligand_mask, pocket_mask, batch)
        z_l, z_p = zt_lig.clone().contiguous(), zt_pocket.clone().contiguous()
        status = torch.zeros(1, dtype=torch.int32, device=z_l.device)
        self._step_impl(s_int, co, z_l, z_p, ligand_mask, pocket_mask, batch, status)
        self._check_status(status)
        return z_l, z_p

    def _step_impl(self, s_int, co, z_l, z_p, lig_mask, pocket_mask, batch, status):
        self._joint_step(s_int, co, z_l, z_p, lig_mask, pocket_mask, batch, status)

    def _infer_step(self, s, t):
        """Recover (timesteps, s) from the [B,1] float tensors s = s_int/timesteps,
        t = (s_int+1)/timesteps the reference API passes."""
        s0, t0 = float(s.reshape(-1)[0]), float(t.reshape(-1)[0])
        timesteps = int(round(1.0 / (t0 - s0)))
        return timesteps, int(round(s0 * timesteps))

    # ---- q(z_t | z_s) and q(z_t | x) (en_diffusion.py:479-501, 302-317) ----------------------
    def sample_p_zt_given_zs(self, zs_lig, zs_pocket, ligand_mask, pocket_mask, gamma_t, gamma_s,
                             fix_noise=False):
        if fix_noise:
            raise NotImplementedError("fix_noise option isn't implemented yet")
        batch = gamma_t.shape[0]
        _, sigma_ts, alpha_ts = self.sigma_and_alpha_t_given_s(gamma_t, gamma_s, zs_lig)
        n_l, n_p = self._joint_noise(ligand_mask, pocket_mask, batch)
        zt_l = alpha_ts[ligand_mask] * zs_lig + sigma_ts[ligand_mask] * n_l
        zt_p = alpha_ts[pocket_mask] * zs_pocket + sigma_ts[pocket_mask] * n_p
        return self._remove_joint_com(zt_l, zt_p, ligand_mask, pocket_mask, batch)

    def _remove_joint_com(self, z_l, z_p, lig_mask, pocket_mask, batch):
        nd, nl = self.n_dims, lig_mask.numel()
        comb = torch.cat((lig_mask, pocket_mask))
        zx = torch.cat((z_l[:, :nd], z_p[:, :nd]), dim=0)
        zx = zx - seg_mean(zx, comb, batch)[comb]
        return torch.cat((zx[:nl], z_l[:, nd:]), dim=1), torch.cat((zx[nl:], z_p[:, nd:]), dim=1)

    def noised_representation(self, xh_lig, xh_pocket, lig_mask, pocket_mask, gamma_t):
        batch = gamma_t.shape[0]
        alpha_t, sigma_t = self.alpha(gamma_t, xh_lig), self.sigma(gamma_t, xh_lig)
        eps_l, eps_p = self._joint_noise(lig_mask, pocket_mask, batch)
        z_l = alpha_t[lig_mask] * xh_lig + sigma_t[lig_mask] * eps_l
        z_p = alpha_t[pocket_mask] * xh_pocket + sigma_t[pocket_mask] * eps_p
        return z_l, z_p, eps_l, eps_p

    # ---- p(x, h | z_0) (en_diffusion.py:263-288, 157-169) ------------------------------------------
    def compute_x_pred(self, net_out, zt, gamma_t, batch_mask):
        sigma_t = self.sigma(gamma_t, target_tensor=net_out)
        alpha_t = self.alpha(gamma_t, target_tensor=net_out)
        return 1. / alpha_t[batch_mask] * (zt - sigma_t[batch_mask] * net_out)

    def sample_p_xh_given_z0(self, z0_lig, z0_pocket, lig_mask, pocket_mask, batch_size, fix_noise=False,
                             _in_chain=False):
        if fix_noise:
            raise NotImplementedError("fix_noise option isn't implemented yet")
        if not _in_chain:
            lig_mask, pocket_mask = self._begin_chain(lig_mask, pocket_mask, batch_size)
        dev = z0_lig.device
        t0 = torch.zeros((batch_size, 1), device=dev)
        gamma_0 = self.gamma(t0)
        sigma_x = self.SNR(-0.5 * gamma_0)
        status = torch.zeros(1, dtype=torch.int32, device=dev)
        try:
            e_l, e_p, _ = self._dyn(z0_lig.contiguous(), z0_pocket.contiguous(), 0.0, lig_mask, pocket_mask,
                                    batch_size, status, True)
            self._check_status(status)
        finally:
            if not _in_chain:
                self._end_chain()
        xh_l = self.compute_x_pred(e_l, z0_lig, gamma_0, lig_mask).contiguous()
        xh_p = self.compute_x_pred(e_p, z0_pocket, gamma_0, pocket_mask).contiguous()
        # xh = mu + sigma_x * eps, eps COM-free (en_diffusion.py:263-288); one t for the whole batch
        self._joint_gauss_(xh_l, xh_p, lig_mask, pocket_mask, batch_size, 1.0, float(sigma_x.reshape(-1)[0]), False)
        nd = self.n_dims
        x_l, h_l = self.unnormalize(xh_l[:, :nd], z0_lig[:, nd:])
        x_p, h_p = self.unnormalize(xh_p[:, :nd], z0_pocket[:, nd:])
        h_l = F.one_hot(torch.argmax(h_l, dim=1), self.atom_nf)
        h_p = F.one_hot(torch.argmax(h_p, dim=1), self.residue_nf)
        return x_l, h_l, x_p, h_p

    def _finish_joint(self, z_l, z_p, lig_mask, pocket_mask, n, return_frames, out_lig, out_pocket):
        x_l, h_l, x_p, h_p = self.sample_p_xh_given_z0(z_l, z_p, lig_mask, pocket_mask, n, _in_chain=True)
        comb = torch.cat((lig_mask, pocket_mask))
        self.assert_mean_zero_with_mask(torch.cat((x_l, x_p), dim=0), comb)
        if return_frames == 1:                                         # en_diffusion.py:636-644
            x = torch.cat((x_l, x_p))
            max_cog = seg_sum(x, comb, n).abs().max().item()
            if max_cog > 5e-2:
                print(f'Warning CoG drift with error {max_cog:.3f}. Projecting the positions down.')
                x = self.remove_mean_batch(x, comb)
                x_l, x_p = x[:len(x_l)], x[len(x_l):]
        out_lig[0] = torch.cat([x_l, h_l], dim=1)
        out_pocket[0] = torch.cat([x_p, h_p], dim=1)
        return out_lig.squeeze(0), out_pocket.squeeze(0), lig_mask, pocket_mask

    # ---- sampling (en_diffusion.py:580-651) --------------------------------------------------------
    @torch.no_grad()
    @_chain
    def sample(self, n_samples, num_nodes_lig, num_nodes_pocket, return_frames=1, timesteps=None,
               device='cpu'):
        timesteps = self.T if timesteps is None else timesteps
        assert 0 < return_frames <= timesteps
        assert timesteps % return_frames == 0
        device = self._hip_device(device)
        lig_mask = num_nodes_to_batch_mask(n_samples, num_nodes_lig, device).contiguous()
        pocket_mask = num_nodes_to_batch_mask(n_samples, num_nodes_pocket, device).contiguous()
        lig_mask, pocket_mask = self._begin_chain(lig_mask, pocket_mask, n_samples)
        co = self._coefs(timesteps)
        z_l = torch.zeros((lig_mask.numel(), self.n_dims + self.atom_nf), device=device)
        z_p = torch.zeros((pocket_mask.numel(), self.n_dims + self.residue_nf), device=device)
        self._joint_gauss_(z_l, z_p, lig_mask, pocket_mask, n_samples, 0.0, 1.0, False)     # z_T
        out_lig = torch.zeros((return_frames,) + z_l.size(), device=device)
        out_pocket = torch.zeros((return_frames,) + z_p.size(), device=device)
        status = torch.zeros(1, dtype=torch.int32, device=device)
        for s in reversed(range(0, timesteps)):
            self._joint_step(s, co, z_l, z_p, lig_mask, pocket_mask, n_samples, status)
            if (s * return_frames) % timesteps == 0:
                idx = (s * return_frames) // timesteps
                out_lig[idx], out_pocket[idx] = self.unnormalize_z(z_l, z_p)
        self._check_status(status)
        return self._finish_joint(z_l, z_p, lig_mask, pocket_mask, n_samples, return_frames,
                                  out_lig, out_pocket)

    def _hip_device(self, device):
        """The sampling state lives where the dynamics' parameters live."""
        p = next(self.dynamics.parameters())
        if p.device.type != 'cuda':
            raise _lib.HipLibraryError(
                "sampling runs on the HIP kernels only: move the model to a GPU "
                f"(parameters are on {p.device}); there is no CPU fallback")
        return p.device

    # ---- RePaint (en_diffusion.py:653-837) ------------------------------------------------------------
    def get_repaint_schedule(self, resamplings, jump_length, timesteps):
        """Number of denoising steps before each jump back, in execution order."""
        sched, cur = [], 0
        while cur < timesteps:
            step = jump_length if cur + jump_length < timesteps else timesteps - cur
            last = cur + jump_length >= timesteps
            if sched:
                sched[-1] += step
                if not last:
                    sched.extend([jump_length] * (resamplings - 1))
            else:
                sched.extend([step] * (1 if last else resamplings))
            cur += step
        return sched[::-1]

    def _joint_inpaint_iteration(self, s, co, z_l, z_p, zk_l, zk_p, xh0_l, xh0_p, lfix, pfix, lm, pm, n, status,
                                 jump_to=None, frame_out=None):
        """One iteration of the RePaint loop body (en_diffusion.py:742-809), in place on the state buffers z_l / z_p
        (stable pointers: the engine replays its captured graph): noise draws for the known part q(z_s | x) (:742-746),
        one reverse step of the whole state (one EGNN call), then ONE kernel for COM alignment over the fixed nodes,
        the blend and -- at the end of a resampling segment (`jump_to` = s + jump_length) -- the jump back
        q(z_t | z_s) (:793-809).  `frame_out` = (out_lig, out_pocket, idx): the blended state is a frame of the
        chain visualisation; the jump then runs as its own launch behind the copy."""
        lib = _lib.load()

        def repaint(n1, n2, a_ts, s_ts):
            _lib.check(lib.dsbdd_joint_repaint_update(
                self._cs(z_l), z_l.data_ptr(), z_p.data_ptr(), zk_l.data_ptr(), zk_p.data_ptr(),
                xh0_l.data_ptr(), xh0_p.data_ptr(), lfix.data_ptr(), pfix.data_ptr(), n1[0].data_ptr(),
                n1[1].data_ptr(), n2[0].data_ptr() if n2 else None, n2[1].data_ptr() if n2 else None,
                lm.data_ptr(), pm.data_ptr(), lm.numel(), pm.numel(), n, self.atom_nf, self.residue_nf,
                float(co.alpha[s]), float(co.sigma_t[s]), a_ts, s_ts, 1 if n2 else 0),
                "dsbdd_joint_repaint_update")

        n1 = self._joint_noise_raw(lm, pm, n)                   # known part: q(z_s | x)
        self._joint_step(s, co, z_l, z_p, lm, pm, n, status)    # unknown part: one reverse step
        jump = jump_to is not None
        a_ts, s_ts = co.pair(s, jump_to) if jump else (1.0, 0.0)
        if jump and frame_out is None:
            repaint(n1, self._joint_noise_raw(lm, pm, n), a_ts, s_ts)
        else:
            repaint(n1, None, 1.0, 0.0)
            if frame_out is not None:
                out_lig, out_pocket, idx = frame_out
                out_lig[idx], out_pocket[idx] = self.unnormalize_z(z_l, z_p)
            if jump:
                self._joint_gauss_(z_l, z_p, lm, pm, n, a_ts, s_ts, True)

    @torch.no_grad()
    @_chain
    def inpaint(self, ligand, pocket, lig_fixed, pocket_fixed, resamplings=1, jump_length=1,
                return_frames=1, timesteps=None):
        timesteps = self.T if timesteps is None else timesteps
        assert 0 < return_frames <= timesteps
        assert timesteps % return_frames == 0
        assert jump_length == 1 or return_frames == 1, \
            "Chain visualization is only implemented for jump_length=1"
        if len(lig_fixed.size()) == 1:
            lig_fixed = lig_fixed.unsqueeze(1)
        if len(pocket_fixed.size()) == 1:
            pocket_fixed = pocket_fixed.unsqueeze(1)
        ligand, pocket = self.normalize(ligand, pocket)
        dev = self._hip_device(None)
        n = len(ligand['size'])
        nd = self.n_dims
        lm, pm = self._begin_chain(ligand['mask'], pocket['mask'], n)
        lfix = lig_fixed.to(dev).float().reshape(-1).contiguous()
        pfix = pocket_fixed.to(dev).float().reshape(-1).contiguous()
        comb = torch.cat((lm, pm))
        xh0_l = torch.cat([ligand['x'], ligand['one_hot']], dim=1).to(dev).contiguous()
        xh0_p = torch.cat([pocket['x'], pocket['one_hot']], dim=1).to(dev).contiguous()
        # centre the input at the COM of the known nodes (en_diffusion.py:703-709); the row ids of the
        # known nodes are found once (one host sync per chain), in sample order for the fixed-order mean
        kx = torch.cat((xh0_l[:, :nd][lfix != 0], xh0_p[:, :nd][pfix != 0]))
        kid = torch.cat((lm[lfix != 0], pm[pfix != 0]))
        order = torch.argsort(kid, stable=True)
        mean_known = self._seg_mean3(kx[order], kid[order].contiguous(), n)
        xh0_l[:, :nd] = xh0_l[:, :nd] - mean_known[lm]
        xh0_p[:, :nd] = xh0_p[:, :nd] - mean_known[pm]

        co = self._coefs(timesteps)
        z_l, z_p = torch.zeros_like(xh0_l), torch.zeros_like(xh0_p)
        self._joint_gauss_(z_l, z_p, lm, pm, n, 0.0, 1.0, False)                            # z_T
        zk_l, zk_p = torch.empty_like(z_l), torch.empty_like(z_p)                           # kernel scratch
        out_lig = torch.zeros((return_frames,) + z_l.size(), device=dev)
        out_pocket = torch.zeros((return_frames,) + z_p.size(), device=dev)
        status = torch.zeros(1, dtype=torch.int32, device=dev)

        schedule = self.get_repaint_schedule(resamplings, jump_length, timesteps)
        s = timesteps - 1
        for i, n_denoise_steps in enumerate(schedule):
            for j in range(n_denoise_steps):
                jump = j == n_denoise_steps - 1 and i < len(schedule) - 1
                frame = (n_denoise_steps > jump_length or i == len(schedule) - 1) and \
                    (s * return_frames) % timesteps == 0
                frame_out = None
                if frame:
                    idx = (s * return_frames) // timesteps
                    frame_out = (out_lig, out_pocket, idx)
                self._joint_inpaint_iteration(s, co, z_l, z_p, zk_l, zk_p, xh0_l, xh0_p, lfix, pfix, lm, pm, n, status,
                                              jump_to=s + jump_length if jump else None, frame_out=frame_out)
                if jump:
                    s = s + jump_length
                s -= 1
        self._check_status(status)
        self.assert_mean_zero_with_mask(torch.cat((z_l[:, :nd], z_p[:, :nd]), dim=0), comb)
        return self._finish_joint(z_l, z_p, lm, pm, n, return_frames, out_lig, out_pocket)

    # ---- misc API kept for callers -------------------------------------------------------------
    def xh_given_zt_and_epsilon(self, z_t, epsilon, gamma_t, batch_mask):
        alpha_t, sigma_t = self.alpha(gamma_t, z_t), self.sigma(gamma_t, z_t)
        return z_t / alpha_t[batch_mask] - epsilon * sigma_t[batch_mask] / alpha_t[batch_mask]

    def log_pN(self, N_lig, N_pocket):
        return self.size_distribution.log_prob(N_lig, N_pocket)

    def delta_log_px(self, num_nodes):
        return -self.subspace_dimensionality(num_nodes) * np.log(self.norm_values[0])

    @staticmethod
    def sum_except_batch(x, indices, n=None):
        """utils.sum_except_batch.  `n`: the batch size when the caller knows it -- `indices.max()` as a host value is a
        device synchronisation, and after the network call of a training step it stalls the host behind the whole
        EGNN forward (round 6: the loss terms pass it)."""
        if n is None:
            n = int(indices.max()) + 1
        return seg_sum(x.sum(-1), indices, n)

    @staticmethod
    def cdf_standard_gaussian(x):
        return 0.5 * (1. + torch.erf(x / math.sqrt(2)))
