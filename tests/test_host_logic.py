"""CPU tests of the host side: weight packing + launch orchestration (through
the emulator that mirrors engine.hip), state_dict contract, schedule tables,
RePaint schedule, C-ABI symbol export."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

from diffsbdd_amd import _lib
from diffsbdd_amd.dynamics import EGNNDynamics
from diffsbdd_amd.engine import make_config, pack_weights
from diffsbdd_amd.en_diffusion import (DistributionNodes, EnVariationalDiffusion,
                                       PredefinedNoiseSchedule, StepCoefficients)
from diffsbdd_amd.conditional_model import ConditionalDDPM, SimpleConditionalDDPM
from oracle import ddpm_oracle as do
from oracle import egnn_oracle as eo
from oracle import weights as W
from tests import _emulate as em
from tests._golden import Case, DYN_CASES, GOLDEN_DIR

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _hp(cfg):
    return dict(atom_nf=cfg["atom_nf"], residue_nf=cfg["residue_nf"], joint_nf=cfg["joint_nf"],
                hidden_nf=cfg["hidden_nf"], n_layers=cfg["n_layers"], inv_sublayers=cfg["inv_sublayers"],
                attention=cfg["attention"], tanh=cfg["tanh"],
                update_pocket_coords=cfg["update_pocket_coords"],
                reflection_equivariant=cfg["reflection_equivariant"],
                edge_embedding_dim=cfg["edge_embedding_dim"],
                edge_cutoff_ligand=cfg["edge_cutoff_ligand"], edge_cutoff_pocket=cfg["edge_cutoff_pocket"],
                edge_cutoff_interaction=cfg["edge_cutoff_interaction"], norm_constant=cfg["norm_constant"],
                normalization_factor=cfg["normalization_factor"])


@pytest.mark.parametrize("name", DYN_CASES)
def test_packed_weights_and_orchestration_reproduce_golden(name):
    """Emulated engine (packed slots, factorised first layer, edge prefix) ==
    reference golden, teacher-forced edges."""
    c = Case(name)
    cfg = make_config(**_hp(c.cfg))
    slots = pack_weights(c.state_dict(), cfg, "cpu")
    assert len(slots) == len(_lib.G_NAMES) + cfg.n_layers * (
        cfg.inv_sublayers * len(_lib.GCL_NAMES) + len(_lib.EQ_NAMES))
    trace = []
    e_l, e_p, _ = em.forward(cfg, slots, c.t("xh_lig"), c.t("xh_pocket"), c.t("t"), c.t("mask_lig"),
                             c.t("mask_pocket"), edges=c.t("edges", torch.int64), trace=trace)
    assert (e_l - c.t("eps_lig")).abs().max() < 2e-5
    assert (e_p - c.t("eps_pocket")).abs().max() < 2e-5
    for i, (h, x) in enumerate(trace):
        if c.has(f"trace_x_{i}"):
            assert (x - c.t(f"trace_x_{i}")).abs().max() < 2e-5


@pytest.mark.parametrize("name", ["dyn_small_cond", "dyn_small_joint", "dyn_small_variant"])
def test_emulated_edge_builder_semantics(name):
    """graph.h semantics (exact distance, sqrt(d2) <= cutoff, order) vs the
    reference list, up to the cdist ambiguity band."""
    c = Case(name)
    cfg = make_config(**_hp(c.cfg))
    x = torch.cat([c.t("xh_lig")[:, :3], c.t("xh_pocket")[:, :3]])
    row, col = em.build_edges(x, c.t("mask_lig"), c.t("mask_pocket"), cfg)
    ref = c.t("edges", torch.int64)
    n = len(x)
    a = torch.zeros(n, n, dtype=torch.bool); a[ref[0], ref[1]] = True
    b = torch.zeros(n, n, dtype=torch.bool); b[row, col] = True
    band = eo.edge_ambiguity_band(c.t("mask_lig"), c.t("mask_pocket"), x[:len(c.t("mask_lig"))],
                                  x[len(c.t("mask_lig")):], c.cfg["edge_cutoff_ligand"],
                                  c.cfg["edge_cutoff_pocket"], c.cfg["edge_cutoff_interaction"], tol=1e-3)
    assert not ((a ^ b) & ~band).any()


def test_mfma_lane_mapping_and_lds_indexing():
    """The kernels' k-major LDS layout + MFMA 32x32x2 operand/accumulator
    mapping reproduce A @ B (asymmetric operands: catches transposes)."""
    rng = np.random.default_rng(0)
    for BM, BN, wave_rows in ((128, 128, 64), (64, 128, 32), (128, 256, 64), (64, 192, 32), (64, 64, 32)):
        K = 8
        A = rng.standard_normal((BM, K))
        B = rng.standard_normal((K, BN))
        C = em.emulate_tile_gemm(A, B, BM, BN // 64, wave_rows, BM + 1)
        assert not np.isnan(C).any()
        np.testing.assert_allclose(C, A @ B, atol=1e-12)


def test_state_dict_contract():
    """Key names / shapes / order of SURVEY.md §8b incl. the aliased last layer."""
    for arch in ("crossdock_ca_cond", "crossdock_fullatom_cond", "moad_fullatom_joint", "small_variant"):
        cfg, dd = W.arch_cfg(arch)
        model = EGNNDynamics(**cfg)
        got = {k: tuple(v.shape) for k, v in model.state_dict().items()}
        want = W.dynamics_param_shapes(cfg)
        assert list(got.items()) == [(k, tuple(v)) for k, v in want.items()]
        if not cfg["reflection_equivariant"]:
            eqp = model.egnn.e_block_0.gcl_equiv
            assert eqp.coord_mlp[4].weight is eqp.cross_product_mlp[4].weight
        model.load_state_dict(W.random_state_dict(cfg, 0), strict=True)
        cls = ConditionalDDPM if dd["conditional"] else EnVariationalDiffusion
        ddpm = cls(dynamics=model, atom_nf=cfg["atom_nf"], residue_nf=cfg["residue_nf"], n_dims=3,
                   size_histogram=np.ones((6, 30)), timesteps=dd["timesteps"],
                   noise_schedule=dd["noise_schedule"], noise_precision=dd["noise_precision"],
                   loss_type="l2", norm_values=dd["norm_values"])
        keys = list(ddpm.state_dict().keys())
        assert keys[0] == "buffer" and keys[1] == "gamma.gamma"
        assert all(k.startswith("dynamics.") for k in keys[2:])
        assert len(keys) == 2 + len(want)
    n_params = sum(p.numel() for p in EGNNDynamics(**W.arch_cfg("crossdock_fullatom_cond")[0]).parameters())
    assert n_params == 4821003   # SURVEY.md §8a (4,821,504 incl. buffer + gamma table)


def test_unsupported_configurations_fail_loudly():
    cfg, _ = W.arch_cfg("small_cond")
    for bad in (dict(hidden_nf=100), dict(mode="gnn_dynamics"), dict(sin_embedding=True),
                dict(aggregation_method="mean"), dict(condition_time=False)):
        kw = dict(cfg); kw.update(bad)
        with pytest.raises(NotImplementedError):
            EGNNDynamics(**kw)
    model = EGNNDynamics(**cfg)   # CPU module: construction is fine, running is not
    with pytest.raises(_lib.HipLibraryError):
        model(torch.zeros(2, 13), torch.zeros(3, 13), torch.zeros(1, 1), torch.zeros(2, dtype=torch.long),
              torch.zeros(3, dtype=torch.long))
    with pytest.raises(AssertionError):   # conditional_model.py:16-18
        kw = dict(cfg); kw["update_pocket_coords"] = True
        ConditionalDDPM(dynamics=EGNNDynamics(**kw), atom_nf=10, residue_nf=10, n_dims=3,
                        size_histogram=np.ones((4, 8)), timesteps=20, noise_schedule="polynomial_2",
                        noise_precision=5e-4, loss_type="l2", norm_values=(1., 4.))
    with pytest.raises(ValueError):       # en_diffusion.py:68-81
        ConditionalDDPM(dynamics=EGNNDynamics(**cfg), atom_nf=10, residue_nf=10, n_dims=3,
                        size_histogram=np.ones((4, 8)), timesteps=20, noise_schedule="polynomial_2",
                        noise_precision=5e-4, loss_type="l2", norm_values=(1., 40.))


def test_schedule_tables_and_step_coefficients():
    z = np.load(os.path.join(GOLDEN_DIR, "schedule.npz"))
    for tag, (sched, T, prec) in {"poly2_T500_p5e-4": ("polynomial_2", 500, 5e-4),
                                  "poly2_T500_p1e-5": ("polynomial_2", 500, 1e-5),
                                  "cosine_T20_p1e-4": ("cosine", 20, 1e-4),
                                  "cosine_T1000_p1e-4": ("cosine", 1000, 1e-4)}.items():
        g = PredefinedNoiseSchedule(sched, T, prec)
        np.testing.assert_array_equal(g.gamma.numpy(), z["gamma_" + tag])
    # coefficients == what the oracle (== reference formulas) computes per step
    g = PredefinedNoiseSchedule("polynomial_2", 500, 5e-4)
    for timesteps in (500, 50):
        co = StepCoefficients(g.gamma, 500, timesteps)
        m = do.OracleModel({}, {}, 10, 10, 500, "polynomial_2", 5e-4)
        for s in (0, 1, timesteps // 2, timesteps - 1):
            s_arr = torch.full((1, 1), float(s)) / timesteps
            t_arr = torch.full((1, 1), float(s + 1)) / timesteps
            s2, s_ts, a_ts, sig_s, sig_t = do.cond_step_coeffs(m, s_arr, t_arr)
            # torch's vectorised and scalar CPU kernels differ in the last ulp of
            # expm1/softplus/logsigmoid (the reference itself evaluates these on
            # [B,1] tensors, i.e. on either path depending on B): compare to 1e-6
            rel = lambda a, b: abs(float(a) - float(b)) <= 1e-6 * abs(float(b))
            assert rel(co.alpha_ts[s], a_ts)
            assert rel(co.c_eps[s], s2 / a_ts / sig_t)
            assert rel(co.sigma[s], s_ts * sig_s / sig_t)
            assert float(co.t_value[s + 1]) == float(t_arr)


def test_repaint_schedule_matches_reference_golden():
    z = np.load(os.path.join(GOLDEN_DIR, "schedule.npz"))
    cfg, dd = W.arch_cfg("small_joint")
    ddpm = EnVariationalDiffusion(dynamics=EGNNDynamics(**cfg), atom_nf=10, residue_nf=10, n_dims=3,
                                  size_histogram=np.ones((4, 8)), timesteps=20, noise_schedule="polynomial_2",
                                  noise_precision=1e-5, loss_type="l2", norm_values=(5., 5.))
    for rec in json.loads(str(z["repaint_json"])):
        assert ddpm.get_repaint_schedule(rec["resamplings"], rec["jump_length"], rec["timesteps"]) \
            == rec["schedule"], rec
    for r in range(1, 5):
        for j in range(1, 6):
            for T in (1, 2, 7, 10, 23):
                assert ddpm.get_repaint_schedule(r, j, T) == do.repaint_schedule(r, j, T), (r, j, T)


def test_distribution_nodes():
    torch.manual_seed(0)
    hist = np.zeros((5, 7)); hist[2, 3] = 10; hist[4, 3] = 30; hist[1, 6] = 5
    d = DistributionNodes(hist)
    nl, npk = d.sample(200)
    assert set(zip(nl.tolist(), npk.tolist())) <= {(2, 3), (4, 3), (1, 6)}
    n1 = d.sample_conditional(n2=torch.tensor([3, 3, 6, 3]))
    assert n1[2] == 1 and set(n1[[0, 1, 3]].tolist()) <= {2, 4}
    lp = d.log_prob(torch.tensor([4]), torch.tensor([3]))
    assert abs(float(lp) - np.log((30 + 1e-3) / (45 + 35e-3))) < 1e-5


def test_c_abi_exports_every_declared_symbol():
    """include/diffsbdd_hip.h <-> libdiffsbdd_hip.so <-> ctypes binding."""
    hdr = open(os.path.join(ROOT, "include", "diffsbdd_hip.h")).read()
    declared = set(re.findall(r"\b(dsbdd_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    lib = _lib.load()                      # loads without a GPU; no compute calls here
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.dsbdd_abi_version() == _lib.ABI_VERSION == 6
    # enum sizes the binding mirrors
    assert len(_lib.G_NAMES) == 20 and len(_lib.GCL_NAMES) == 12 and len(_lib.EQ_NAMES) == 12
    for names, prefix in ((_lib.G_NAMES, "DSBDD_G_"), (_lib.GCL_NAMES, "DSBDD_GCL_"), (_lib.EQ_NAMES, "DSBDD_EQ_")):
        pos = [hdr.index(prefix + n) for n in names]
        assert pos == sorted(pos), prefix        # same order as the header enums
    # engine bookkeeping entry points work on the host alone
    cfg = make_config(**_hp(W.arch_cfg("crossdock_fullatom_cond")[0]))
    h = ctypes.c_void_p()
    assert lib.dsbdd_engine_create(ctypes.byref(cfg), ctypes.byref(h)) == 0
    assert lib.dsbdd_engine_weight_slots(h) == 20 + 6 * 24
    assert lib.dsbdd_engine_workspace_bytes(h, 1472, 18304, 64, 6_200_000) > 0
    lib.dsbdd_engine_destroy(h)
    bad = make_config(**{**_hp(W.arch_cfg("small_cond")[0]), "hidden_nf": 96})
    assert lib.dsbdd_engine_create(ctypes.byref(bad), ctypes.byref(h)) == _lib.ERR_ARG
    assert b"hidden_nf" in lib.dsbdd_last_error()


def test_c_abi_argument_and_state_errors():
    """Error behaviour of the entry points that can be exercised without a GPU: every
    misuse returns a negative DSBDD_ERR_* code with a message, nothing is launched."""
    lib = _lib.load()
    cfg = make_config(**_hp(W.arch_cfg("small_cond")[0]))
    h = ctypes.c_void_p()
    assert lib.dsbdd_engine_create(None, ctypes.byref(h)) == _lib.ERR_ARG
    assert lib.dsbdd_engine_create(ctypes.byref(cfg), ctypes.byref(h)) == 0
    n_slots = lib.dsbdd_engine_weight_slots(h)
    assert n_slots == 20 + 2 * 24
    # forward before weights / workspace
    one = ctypes.c_void_p(4096)          # any non-null, never dereferenced on these paths
    args = (h, None, one, one, one, 1, one, one, 4, 8, 1, None, None, 0, one, one, one)
    assert lib.dsbdd_dynamics_forward(*args) == _lib.ERR_STATE
    assert b"weights" in lib.dsbdd_last_error()
    assert lib.dsbdd_dynamics_forward(None, *args[1:]) == _lib.ERR_ARG
    # weights: wrong count, null slot, misaligned slot
    arr = (ctypes.c_void_p * n_slots)(*([4096] * n_slots))
    assert lib.dsbdd_engine_set_weights(h, arr, n_slots - 1) == _lib.ERR_ARG
    arr_null = (ctypes.c_void_p * n_slots)(*([4096] * (n_slots - 1) + [None]))
    assert lib.dsbdd_engine_set_weights(h, arr_null, n_slots) == _lib.ERR_ARG
    arr_mis = (ctypes.c_void_p * n_slots)(*([4096] * (n_slots - 1) + [4100]))
    assert lib.dsbdd_engine_set_weights(h, arr_mis, n_slots) == _lib.ERR_ARG
    assert b"aligned" in lib.dsbdd_last_error()
    assert lib.dsbdd_engine_set_weights(h, arr, n_slots) == 0
    assert lib.dsbdd_dynamics_forward(*args) == _lib.ERR_STATE
    assert b"workspace" in lib.dsbdd_last_error()
    # workspace: misaligned, too small, too large for int32 indexing
    need = lib.dsbdd_engine_workspace_bytes(h, 4, 8, 1, 144)
    assert need > 0 and lib.dsbdd_engine_workspace_bytes(h, 4, 8, 0, 144) == 0
    assert lib.dsbdd_engine_bind_workspace(h, ctypes.c_void_p(4096 + 64), need, 4, 8, 1, 144) == _lib.ERR_ARG
    assert lib.dsbdd_engine_bind_workspace(h, ctypes.c_void_p(4096), need - 1, 4, 8, 1, 144) == _lib.ERR_CAPACITY
    assert lib.dsbdd_engine_bind_workspace(h, ctypes.c_void_p(4096), need, 1 << 30, 8, 1, 144) == _lib.ERR_ARG
    assert lib.dsbdd_engine_bind_workspace(h, ctypes.c_void_p(4096), need, 4, 8, 1, 144) == 0
    # call sizes beyond the bound capacity, bad t count
    big = list(args)
    big[8] = 5
    assert lib.dsbdd_dynamics_forward(*big) == _lib.ERR_CAPACITY
    bad_t = list(args)
    bad_t[5] = 3
    assert lib.dsbdd_dynamics_forward(*bad_t) == _lib.ERR_ARG
    lib.dsbdd_engine_destroy(h)
    # stand-alone kernels validate too
    assert lib.dsbdd_node_linear(None, None, 4, 4, None, 0, 0, one, 4, None, None, 0, one, 4, 1, 4, 0) == _lib.ERR_ARG
    assert lib.dsbdd_bond_orders(None, one, one, one, 0, 10, one, one, one, 3.0, 2.0, 1.0, 8, one) == _lib.ERR_ARG


def test_loss_forward_needs_a_gpu_in_both_modes():
    """forward() evaluates the loss terms on the HIP kernels (eval / no_grad) or on the differentiable
    GPU path (training step); a CPU module fails like every other entry point -- there is no CPU fallback."""
    cfg, dd = W.arch_cfg("small_cond")
    ddpm = ConditionalDDPM(dynamics=EGNNDynamics(**cfg), atom_nf=10, residue_nf=10, n_dims=3,
                           size_histogram=np.ones((4, 8)), timesteps=20, noise_schedule="polynomial_2",
                           noise_precision=5e-4, loss_type="l2", norm_values=(1., 4.))
    lig = {"x": torch.zeros(2, 3), "one_hot": torch.zeros(2, 10), "size": torch.tensor([2]), "mask": torch.zeros(2, dtype=torch.long)}
    poc = {"x": torch.zeros(3, 3), "one_hot": torch.zeros(3, 10), "size": torch.tensor([3]), "mask": torch.zeros(3, dtype=torch.long)}
    for training in (True, False):
        ddpm.train(training)
        with pytest.raises(_lib.HipLibraryError):
            ddpm({k: v.clone() for k, v in lig.items()}, {k: v.clone() for k, v in poc.items()})


def test_aligned_edge_layout_and_atomic_free_aggregation_protocol():
    """Host model of csrc/graph.h (segment-aligned row offsets) and csrc/edge_mlp.h (tile partial sums
    + ordered head sums): every row's edges are found at [row_ptr, row_ptr + deg), segments start at
    multiples of 32, the protocol reproduces the plain per-row sum exactly on integer-valued data, and a
    sample's rows get bit-identical sums whatever else is in the batch (the tile cut of a row depends only
    on its own sample)."""
    rng = np.random.default_rng(0)

    def problem(n_lig_per, n_poc_per):
        B = len(n_lig_per)
        nb = np.concatenate([np.repeat(np.arange(B), n_lig_per), np.repeat(np.arange(B), n_poc_per)])
        n_lig = int(np.sum(n_lig_per))
        return B, nb, n_lig

    def degrees(seed, n):
        return np.random.default_rng(seed).integers(0, 90, size=n)          # rows of up to 3 tiles, some empty

    sizes_l, sizes_p = [5, 0, 9, 7], [40, 33, 0, 51]
    B, nb, n_lig = problem(sizes_l, sizes_p)
    # per-node degrees / per-edge values drawn from the node's (sample, local index): independent of the batch
    local = np.concatenate([np.arange(n) for n in sizes_l] + [np.arange(n) for n in sizes_p])
    part = np.concatenate([np.zeros(n_lig, int), np.ones(len(nb) - n_lig, int)])

    def node_deg(b, p, i):
        return int(np.random.default_rng(1000 * b + 500 * p + i).integers(0, 90))

    def node_vals(b, p, i, d):
        return np.random.default_rng(77 + 1000 * b + 500 * p + i).normal(size=(d, 4)).astype(np.float32)

    def run(samples):
        keep = np.isin(nb, samples)
        nbs = np.searchsorted(np.asarray(samples), nb[keep])                 # relabel 0..len-1
        nl = int((keep[:n_lig]).sum())
        idx = np.nonzero(keep)[0]
        deg = np.array([node_deg(nb[i], part[i], local[i]) for i in idx])
        row_ptr, pad = em.aligned_layout(deg, nbs, nl, len(samples))
        assert row_ptr[-1] % 32 == 0 and len(pad) == row_ptr[-1]
        erow = np.full(int(row_ptr[-1]), -1, np.int64)
        vals = np.zeros((int(row_ptr[-1]), 4), np.float32)
        for r, i in enumerate(idx):
            erow[row_ptr[r]:row_ptr[r] + deg[r]] = r
            vals[row_ptr[r]:row_ptr[r] + deg[r]] = node_vals(nb[i], part[i], local[i], deg[r])
        assert np.array_equal(erow < 0, pad)
        # every (sample, node set) segment starts at a tile boundary
        seg_first = {}
        for r, i in enumerate(idx):
            seg_first.setdefault((int(part[i]), int(nbs[r])), int(row_ptr[r]))
        assert all(v % 32 == 0 for v in seg_first.values())
        out = em.tile_protocol_aggregate(vals, erow, row_ptr, deg)
        return idx, deg, vals, erow, out

    idx, deg, vals, erow, out = run([0, 1, 2, 3])
    # exact on integers: protocol == plain scatter add
    ivals = np.round(vals * 8).astype(np.float32)
    rp, _ = em.aligned_layout(deg, np.searchsorted([0, 1, 2, 3], nb), n_lig, 4)
    ref = np.zeros((len(deg), 4), np.float32)
    for s in range(len(erow)):
        if erow[s] >= 0:
            ref[erow[s]] += ivals[s]
    assert np.array_equal(em.tile_protocol_aggregate(ivals, erow, rp, deg), ref)
    # batch invariance, bitwise: samples {0, 2} alone, and sample 3 alone
    for sub in ([0, 2], [3], [1, 3]):
        idx_s, _, _, _, out_s = run(sub)
        pos = {int(i): r for r, i in enumerate(idx)}
        assert np.array_equal(out_s, np.stack([out[pos[int(i)]] for i in idx_s])), sub


def test_pocket_groups_and_frame_layout():
    """Host side of the pocket frame (en_diffusion._pocket_groups, engine.frame_layout): identical pockets are found by
    content, the frame holds the distinct representatives in order, twin / frame_rows index them consistently, and
    inconsistent groupings are rejected before anything reaches the GPU."""
    import torch
    from diffsbdd_amd.en_diffusion import EnVariationalDiffusion as EVD
    from diffsbdd_amd.engine import frame_layout
    g = torch.Generator().manual_seed(0)
    A, B_ = torch.randn(5, 3, generator=g), torch.randn(7, 3, generator=g)
    hA, hB = torch.eye(4)[torch.tensor([0, 1, 2, 3, 0])], torch.eye(4)[torch.tensor([1, 1, 2, 0, 3, 2, 1])]
    # one pocket repeated: decided without hashing -> all zeros
    x = A.repeat(3, 1)
    assert EVD._pocket_groups(x, hA.repeat(3, 1), torch.tensor([5, 5, 5]), 3).tolist() == [0, 0, 0]
    # A B A B' A: B' differs from B in one coordinate; same sizes do not make pockets identical
    Bp = B_.clone()
    Bp[3, 1] += 1e-3
    xs, hs, sizes = torch.cat([A, B_, A, Bp, A]), torch.cat([hA, hB, hA, hB, hA]), torch.tensor([5, 7, 5, 7, 5])
    rep = EVD._pocket_groups(xs, hs, sizes, 5)
    assert rep.tolist() == [0, 1, 0, 3, 0]
    # same coordinates, different atom types -> different pockets
    hA2 = hA.clone()
    hA2[0] = torch.eye(4)[2]
    assert EVD._pocket_groups(torch.cat([A, A]), torch.cat([hA, hA2]), torch.tensor([5, 5]), 2).tolist() == [0, 1]
    mask = torch.repeat_interleave(torch.arange(5), sizes)
    mask3, rows, twin, sz_f = frame_layout(sizes, rep, mask)
    assert sz_f.tolist() == [5, 7, 7] and mask3.tolist() == [0] * 5 + [1] * 7 + [2] * 7
    assert rows.tolist() == list(range(0, 5)) + list(range(5, 12)) + list(range(17, 24))       # samples 0, 1, 3
    assert torch.equal(xs[rows.long()][twin.long()], torch.cat([A, B_, A, Bp, A]))               # twin reproduces every pocket
    assert twin[:5].tolist() == twin[12:17].tolist() == twin[24:].tolist() == [0, 1, 2, 3, 4]
    # every sample its own representative = identity
    m3, r3, t3, _ = frame_layout(sizes, torch.arange(5), mask)
    assert r3.tolist() == t3.tolist() == list(range(29)) and torch.equal(m3, mask)
    for bad in ([0, 1, 1, 3, 0],        # sample 2 (5 atoms) represented by a 7-atom pocket
                [0, 1, 0, 3, 2],        # sample 2 represents sample 4 but not itself
                [0, 1, 0, 3, 9]):       # not a sample of the batch
        with pytest.raises(ValueError):
            frame_layout(sizes, torch.tensor(bad), mask)


def test_node_gemm_tile_schedule_covers_every_output_once():
    """Host model of node_gemm_kernel's workgroup-id -> tile mapping (tests/_emulate.node_gemm_schedule mirrors
    csrc/node_linear.h): every (row tile, 32-column block) of the output is produced by exactly one workgroup for any
    row count (it is read on the device), the full tiles are a multiple of the balance granularity, and the benchmark's
    620-tile problem becomes 512 full + 216 half tiles."""
    for M, N, ct, balance in [(19776, 256, 2, 512), (19776, 512, 2, 512), (19776, 1024, 4, 512), (3640, 256, 2, 512),
                              (1, 256, 2, 512), (128, 64, 2, 512), (19776, 256, 2, 0), (19776, 256, 1, 512),
                              (65536, 512, 2, 512), (11545, 256, 2, 256), (5000, 192, 2, 512)]:
        # the host launches for the row CAPACITY (here: up to 1.5x the rows the device finds)
        cap_tiles = ((int(M * 1.5) + 127) // 128)
        grid = cap_tiles * (N // (32 * ct)) * (2 if (ct > 1 and balance > 0) else 1)
        sched, n_full = em.node_gemm_schedule(M, N, ct, balance, grid)
        m_tiles = (M + 127) // 128
        seen = np.zeros((m_tiles, N // 32), dtype=int)
        for blk in sched:
            if blk is None:
                continue
            r0, c0, w = blk
            assert r0 % 128 == 0 and r0 // 128 < m_tiles and c0 + w <= N
            seen[r0 // 128, c0 // 32:(c0 + w) // 32] += 1
        assert (seen == 1).all(), (M, N, ct, balance)
        gy = N // (32 * ct)
        if ct > 1 and balance > 0:
            assert (n_full * gy) % balance == 0 and (m_tiles - n_full) * gy < balance
        else:
            assert n_full == m_tiles
    sched, n_full = em.node_gemm_schedule(19776, 256, 2, 512)
    real = [b for b in sched if b is not None]
    assert n_full == 128 and sum(1 for b in real if b[2] == 64) == 512 and sum(1 for b in real if b[2] == 32) == 216
    # dispatch order: all full tiles first, column tile fastest
    assert [b[2] for b in real[:512]] == [64] * 512 and [b[1] for b in real[:4]] == [0, 64, 128, 192]


def test_stage_plan_model():
    """Host model of the per-stage plan (tests/_emulate.stage_plan mirrors engine.hip; the GPU tests compare
    dsbdd_engine_last_plan with the same formulas): a row evaluated by stage g + 1 is either computed by stage g or
    canonical, the last stage computes hop <= 1, and without the forward cone the first stages compute everything."""
    assert em.stage_plan(6, True) == ([1, 2, 3, 3, 2, 1], [1, 1, 1, 0, 0, 0])
    assert em.stage_plan(6, False) == ([4, 4, 4, 3, 2, 1], [0] * 6)
    assert em.stage_plan(2, True) == ([1, 1], [1, 0])            # stage 1 reads hop 2, which stage 0 did not compute
    assert em.stage_plan(4, True) == ([1, 2, 2, 1], [1, 1, 0, 0])
    assert em.stage_plan(10, True)[0] == [1, 2, 3, 4, 4, 4, 4, 3, 2, 1]
    for G in range(1, 13):
        for cone in (False, True):
            radius, ghost = em.stage_plan(G, cone)
            assert radius[-1] == 1 and all(1 <= r <= 4 for r in radius)
            for g in range(G - 1):
                # stage g + 1 reads level <= radius[g + 1] + 1: computed by stage g, or taken from the canonical pocket
                need = min(radius[g + 1] + 1, 4)
                assert need <= radius[g] or ghost[g] == 1
                assert radius[g + 1] >= radius[g] - 1          # the backward cone shrinks by one hop per stage
            assert not cone or ghost == sorted(ghost, reverse=True)       # ghosts only in the leading stages


def test_node_chain_row_split_covers_every_tile_once_and_balances_cost():
    """Host model of the row split of csrc/node_chain.h (tests/_emulate.node_chain_ranges mirrors the device code):
    for the benchmark's stages and for random problems every 16-row tile of the work range belongs to exactly one piece,
    no piece exceeds 96 rows, and no workgroup carries more than the mean cost plus one tile of the most expensive rows
    (which is what makes 19.8 k rows a single round on 256 CUs)."""
    rng = np.random.RandomState(0)
    cases = [(19776, 256, True, [(512, 3639, 0), (512, 1472, 0), (512, None, 0)]),        # all rows, full chain
             (18764 + 286, 256, True, [(512, 3639, 286), (512, 1472, 286), (512, None, 0)]),  # ghost rows in front
             (11545, 256, True, [(512, 3639, 0), (512, 1472, 0)]), (3639, 256, True, [(512, 3639, 0), (512, 1472, 0)]),
             (19776, 256, False, [(512, None, 0)]), (1888, 256, True, [(512, None, 0)]), (5, 256, True, []),
             (40000, 256, True, [(512, None, 0)])]
    for _ in range(40):
        M = int(rng.randint(1, 30000))
        projs = [(int(rng.choice([128, 256, 512, 768])), None if rng.rand() < 0.3 else int(rng.randint(0, M + 50)),
                  int(rng.choice([0, 0, rng.randint(0, M + 1)]))) for _ in range(rng.randint(0, 4))]
        cases.append((M, int(rng.choice([8, 104, 256, 304])), bool(rng.rand() < 0.7) or not projs, projs))
    for M, grid, do_mlp, projs in cases:
        pieces, (Mw, total, V) = em.node_chain_ranges(M, grid, do_mlp, projs)
        if Mw == 0 or total == 0:
            assert all(not p for p in pieces)
            continue
        n_tiles = (Mw + 15) // 16
        seen = np.zeros(n_tiles, dtype=int)
        for wg in pieces:
            for r0, take in wg:
                assert 1 <= take <= 6 and r0 % 16 == 0
                seen[r0 // 16:r0 // 16 + take] += 1
        assert (seen == 1).all(), (M, grid, do_mlp, projs)
        assert V % grid == 0

        def cost(r0, take):      # cost of the rows of a piece (same weights as the device code)
            H, w_mlp = 256, 48 if do_mlp else 0
            c = 0
            for r in range(r0, min(r0 + 16 * take, Mw)):
                c += w_mlp + sum((n * 16 // H) for n, cnt, f in projs
                                 if min(M, f) <= r < min(M, f) + max(0, min(M - min(M, f), cnt) if cnt is not None else M - min(M, f)))
            return c
        if M <= 20000 and grid == 256:
            loads = [sum(cost(*p) for p in wg) for wg in pieces]
            w_max = (48 if do_mlp else 0) + sum(n * 16 // 256 for n, _, _ in projs)
            # V / grid ranges per workgroup, each at most total / V plus one tile of the most expensive rows
            assert max(loads) <= total / grid + (V // grid) * 16 * w_max + 1e-9
    # the headline stage: one range per CU, 5 or 6 tiles each
    pieces, (_, _, V) = em.node_chain_ranges(19776, 256, True, [(512, 3639, 0), (512, 1472, 0), (512, None, 0)])
    assert V == 256 and all(len(p) == 1 for p in pieces) and {t for p in pieces for _, t in p} <= {1, 2, 3, 4, 5, 6}


def test_edge_tile_walk_covers_both_lists_once_and_slice_staging_covers_a_slice_once():
    """Host models of two schedules of csrc/edge_wave.h (tests/_emulate mirrors the device arithmetic).
    (1) The static tile walk with a second edge list behind the first (block 0 of a framed call runs both lists in one
    launch): every 128-edge tile of either list is processed by exactly one workgroup -- exactly one (tile, MLP) pair in
    the coordinate stage's split form -- for any edge counts and any grid that is a multiple of 8 (16); the XCD ranges
    differ by at most one tile.
    (2) The W2^T slice stream through staging registers: over the four MFMA groups of a K step a thread requests every
    float4 of its share of the slice exactly once, never holds more than ceil(BI / 4) at a time, and every request is
    written one group later."""
    rng = np.random.RandomState(1)
    cases = [(61440, 4480, 512), (61441, 0, 512), (0, 4480, 512), (354071, 0, 512), (100, 50, 8), (0, 0, 512), (127, 129, 64)]
    cases += [(int(rng.randint(0, 400000)), int(rng.choice([0, rng.randint(0, 20000)])), int(8 * rng.randint(1, 65))) for _ in range(30)]
    for E_a, E_b, grid in cases:
        walk = em.edge_tile_walk(E_a, E_b, grid)
        nt = [-(-E_a // 128), -(-E_b // 128)]
        seen = [np.zeros(nt[0], int), np.zeros(nt[1], int)]
        edges = [0, 0]
        per_xcd = np.zeros(8, int)
        for b, items in enumerate(walk):
            for lb, tl, e0, n, _ in items:
                seen[lb][tl] += 1
                edges[lb] += n
                per_xcd[b & 7] += 1
                assert e0 % 128 == 0 and 0 < n <= 128
        assert (seen[0] == 1).all() and (seen[1] == 1).all(), (E_a, E_b, grid)
        assert edges == [E_a, E_b]
        assert per_xcd.max() - per_xcd.min() <= 1
        if grid % 16 == 0:       # coordinate stage: one workgroup per (tile, MLP)
            pairs = {}
            for items in em.edge_tile_walk(E_a, 0, grid, split=True):
                for lb, tl, e0, n, q in items:
                    pairs[(tl, q)] = pairs.get((tl, q), 0) + 1
            assert sorted(pairs) == [(t, q) for t in range(nt[0]) for q in (0, 1)] and set(pairs.values()) <= {1}
    for H in (64, 128, 192, 256):
        BI, SG, ops = em.slice_stage_schedule(H)
        assert BI == 32 * (H // 4) // 256
        assert sorted(i for _, _, i in ops) == list(range(BI))
        for g in range(4):
            held = [i for gl, gs, i in ops if gl <= g < gs]
            assert len(held) <= SG
        assert all(gs == gl + 1 for gl, gs, _ in ops)


def test_per_chain_engine_choices_are_pure_functions_of_the_batch():
    """Round 4: the forward cone and the 16-edge-granule choice are made in Python, once per chain, from the pocket groups /
    ligand sizes alone (never by an engine heuristic in the middle of a chain)."""
    from diffsbdd_amd.en_diffusion import EnVariationalDiffusion as D
    assert D._cone_for_groups(torch.zeros(64, dtype=torch.int64), 64) == 2          # one pocket repeated
    assert D._cone_for_groups(torch.arange(64), 64) == 0 and D._cone_for_groups(None, 64) == 0
    two = torch.cat([torch.zeros(40, dtype=torch.int64), torch.full((24,), 40)])
    assert D._cone_for_groups(two, 64) == 2                                         # 2 groups of 64: 10 <= 64
    many = torch.cat([torch.zeros(40, dtype=torch.int64), torch.arange(40, 64)])    # 40 x A + 24 singletons: measured slower
    assert D._cone_for_groups(many, 64) == 0 and D._cone_for_groups(torch.arange(64) // 2, 64) == 0
    assert D._cone_for_groups(torch.arange(64) // 6, 64) == 2 and D._cone_for_groups(torch.arange(64) // 4, 64) == 0   # 11 / 16 groups
    # the test-set driver pins one mode for a whole run, from the complete job list: samples per pocket >= 5 -> on
    from diffsbdd_amd import testset as ts
    mk = lambda n: ts.PocketJob("p", [], 100, n)
    assert ts.cone_mode_for_jobs([mk(100)] * 3) == 2 and ts.cone_mode_for_jobs([mk(4)] * 10) == 0
    assert ts.cone_mode_for_jobs([mk(2)] * 9 + [mk(40)]) == 2 and ts.cone_mode_for_jobs([]) == 0
    with pytest.raises(ValueError):
        ts.make_hip_sampler(None, cone_mode=1)
    # coordinate stages on the 16-edge kernels: C-alpha x 32 (274 -> 548 items: a round saved) yes, full-atom x 64 no
    ca = torch.repeat_interleave(torch.arange(32), 23)
    fa = torch.repeat_interleave(torch.arange(64), 23)
    ca_p = torch.repeat_interleave(torch.arange(32), 36)
    fa_p = torch.repeat_interleave(torch.arange(64), 286)
    assert D.granule16_auto(ca, 32, 6, pocket_mask=ca_p) == 0x003F0000
    assert D.granule16_auto(fa, 64, 6, pocket_mask=fa_p) == 0 and D.granule16_auto(fa, 64, 6) == 0
    assert D.granule16_auto(ca[:23 * 8], 8, 6) == 0x003F0000                        # 8 x 23: 68 -> 134 items, half the time
    # round 6, split-K kernels: the stages whose largest possible launch is <= 1.5 rounds of 128-edge tiles; the frame's
    # pocket-pocket edges are counted exactly when the pocket coordinates are known, else bounded by the complete graph
    from diffsbdd_amd import synthetic
    p_ca, p_fa = synthetic.load_pocket("ca", 32, "cpu"), synthetic.load_pocket("fa", 64, "cpu")
    auto = lambda lm, pk, b, **kw: D.splitk_auto(lm, pk["mask"], b, 6, 6, 2, 256, **kw)
    assert auto(ca, p_ca, 32, pocket_x=p_ca["x"], cutoff_pocket=5.0) == 0x003F003F     # C-alpha x 32: every stage
    assert auto(ca, p_ca, 32) == 0x003F0000                                             # pocket unknown: the coordinate stages
    assert auto(fa, p_fa, 64, pocket_x=p_fa["x"], cutoff_pocket=5.0) == 0 and auto(fa, p_fa, 64) == 0
    p16 = synthetic.load_pocket("fa", 16, "cpu")
    assert auto(fa[:23 * 16], p16, 16, pocket_x=p16["x"], cutoff_pocket=5.0) == 0x003F0000   # full-atom x 16: coordinate stages
    assert D.splitk_auto(ca, p_ca["mask"], 32, 6, 6, 2, 192, pocket_x=p_ca["x"], cutoff_pocket=5.0) == 0   # hidden_nf 256 only


def test_training_net_parameter_order_and_sizes():
    """Round 6, the training step as one launch sequence (csrc/train_net.h): the parameter order the C side indexes is the
    module's own construction order minus the aliased cross_product_mlp.4.weight; the size queries are host-only."""
    import ctypes as C
    from diffsbdd_amd import synthetic
    from diffsbdd_amd.dynamics import EGNNDynamics
    from diffsbdd_amd.engine import make_config
    from diffsbdd_amd.train_net import param_names
    lib = _lib.load()
    for arch in ("small_cond", "small_joint", "small_variant", "crossdock_fullatom_cond"):
        cfg, _ = synthetic.arch_cfg(arch)
        m = EGNNDynamics(**cfg, device="cpu")
        names = param_names(m._hp)
        named = dict(m.named_parameters())
        assert set(names) == set(named)                                  # named_parameters() lists the shared tensor once
        assert names == [k for k in synthetic.dynamics_param_shapes(cfg) if not k.endswith("cross_product_mlp.4.weight")]
        h = C.c_void_p()
        c = make_config(**m._hp)
        assert lib.dsbdd_train_net_create(C.byref(c), C.byref(h)) == _lib.OK
        try:
            assert lib.dsbdd_train_net_param_count(h) == len(names)
            assert lib.dsbdd_train_net_pack_bytes(h) > 4 * sum(p.numel() for p in named.values())
            dummy = 4096
            g = _lib.TrainGraph(erow=dummy, ecol=dummy, ed0=dummy, row_ptr=dummy, deg=dummy, rev=dummy, node_batch=dummy,
                                lig_off=dummy, poc_off=dummy, n_lig=23, n_nodes=309, n_edges=5000, batch=1)
            small = lib.dsbdd_train_net_workspace_bytes(h, C.byref(g))
            g.n_edges = 50000
            assert lib.dsbdd_train_net_workspace_bytes(h, C.byref(g)) > small > 0
            g.n_nodes = 0
            assert lib.dsbdd_train_net_workspace_bytes(h, C.byref(g)) == 0                  # not a graph
        finally:
            lib.dsbdd_train_net_destroy(h)
    bad = make_config(**{**m._hp, "hidden_nf": 96})
    h = C.c_void_p()
    assert lib.dsbdd_train_net_create(C.byref(bad), C.byref(h)) == _lib.ERR_ARG


def test_training_c_abi_argument_errors_and_sizes():
    """The dsbdd_train_* entry points reject misuse with DSBDD_ERR_ARG / _CAPACITY before anything is launched (no GPU)."""
    lib = _lib.load()
    assert lib.dsbdd_train_scratch_bytes(96, 100, 1000) == 0                        # unsupported hidden_nf
    n = lib.dsbdd_train_scratch_bytes(256, 4944, 91152)
    assert n > 3 * 91152 * 256 * 4                                                  # dz2, a1, dz1 + partials
    assert lib.dsbdd_train_wgrad_scratch_bytes(0, 4, 4) == 0
    w = lib.dsbdd_train_wgrad_scratch_bytes(91152, 256, 256)
    assert 256 * 256 * 4 <= w <= 200 * 256 * 256 * 4
    # the split-K plan is not monotonic in K (the chunk length is rounded up to 32 rows after the chunk cap: with
    # round 6's cap of 256 workgroups K = 10 000 needs more chunks than K = 16 385 at 256 x 256); the scratch bound must cover
    # every prefix K' <= K -- the coordinate stage's backward runs on an edge prefix
    assert lib.dsbdd_train_wgrad_plan_bytes(10000, 256, 256) > lib.dsbdd_train_wgrad_plan_bytes(16385, 256, 256)
    rng = np.random.default_rng(0)
    for H in (64, 128, 192, 256):
        for E in [int(v) for v in rng.integers(1, 400000, 40)] + [30857, 160000, 91152]:
            bound = lib.dsbdd_train_wgrad_scratch_bytes(E, H, H)
            ks = [int(v) for v in rng.integers(1, E + 1, 60)] + [E, max(1, E // 2), 17228 if E >= 17228 else 1]
            assert all(lib.dsbdd_train_wgrad_plan_bytes(k, H, H) <= bound for k in ks), (H, E)
    one = ctypes.c_void_p(4096)
    assert lib.dsbdd_train_wgrad(None, None, 4, one, 4, 8, 4, 4, one, one, 1 << 20) == _lib.ERR_ARG
    assert lib.dsbdd_train_wgrad(None, one, 4, one, 4, 8, 4, 4, one, one, 16) == _lib.ERR_CAPACITY
    assert lib.dsbdd_train_colsum(None, one, 2, 8, 4, one, one, 1 << 20) == _lib.ERR_ARG      # lda < N
    g = _lib.TrainGraph()                                                            # all-null graph
    assert lib.dsbdd_train_edge_rev(None, ctypes.byref(g), one) == _lib.ERR_ARG
    m = _lib.TrainMlp()
    assert lib.dsbdd_train_gcl_forward(None, 256, ctypes.byref(g), ctypes.byref(m), one, 100.0, one, one, 1 << 20) == _lib.ERR_ARG
    assert b"bad argument" in lib.dsbdd_last_error()


def test_oracle_ref_archive_matches_the_reference_sources():
    """oracle/_ref/reference_path.zip (oracle/make_ref.py; git-ignored, shipped with the push) holds the reference's own
    modules of this path, unmodified: SHA-256 against /root/reference when that exists (build container), and the archive
    imports through oracle/ref_shim.py."""
    import hashlib
    import zipfile
    from oracle import make_ref
    if not os.path.isfile(os.path.join(make_ref.SRC, make_ref.FILES[0])):
        pytest.skip("reference sources not present on this machine")
    assert make_ref.make(verbose=False) and make_ref.available() and make_ref.callers_available()
    with zipfile.ZipFile(make_ref.ARCHIVE) as z:
        man = json.loads(z.read("MANIFEST.json"))
        assert "MIT" in z.read("LICENSE").decode()[:40]           # the reference's licence travels with the copy
        for f in make_ref.FILES + make_ref.CALLER_FILES:
            assert hashlib.sha256(z.read(f)).hexdigest() == man["sha256"][f] == \
                hashlib.sha256(open(os.path.join(make_ref.SRC, f), "rb").read()).hexdigest()


@pytest.mark.parametrize("H,emb", [(64, None), (64, 4), (192, 8)])
def test_edge_first_layer_relayout_on_cpu(H, emb):
    """train_hip.EdgeFirstLayer is plain torch (no kernel): the re-layout of an edge MLP's first Linear (egnn_new.py:35,99)
    and its hand-written backward against autograd over the slicing expression, on the CPU."""
    from diffsbdd_amd.train_hip import EdgeFirstLayer
    g = torch.Generator().manual_seed(H + (emb or 0))
    cols = 2 * H + 2 + (emb or 0)
    w = (torch.randn(H, cols, generator=g, dtype=torch.float64) * 0.1).requires_grad_(True)
    b = torch.randn(H, generator=g, dtype=torch.float64).requires_grad_(True)
    e = torch.randn(3, emb, generator=g, dtype=torch.float64).requires_grad_(True) if emb else None
    outs = EdgeFirstLayer.apply(w, b, e)
    gs = [torch.randn(o.shape, generator=g, dtype=torch.float64) for o in outs]
    torch.autograd.backward(outs, gs)
    w2, b2 = w.detach().clone().requires_grad_(True), b.detach().clone().requires_grad_(True)
    e2 = e.detach().clone().requires_grad_(True) if emb else None
    ref = (torch.cat((w2[:, :H], w2[:, H:2 * H]), 0), w2[:, 2 * H], w2[:, 2 * H + 1],
           (b2[None, :] + e2 @ w2[:, 2 * H + 2:].t()) if emb else b2[None, :].expand(3, H))
    torch.autograd.backward(ref, gs)
    for o, r in zip(outs, ref):
        assert torch.allclose(o, r, atol=1e-12)
    assert torch.allclose(w.grad, w2.grad, atol=1e-12) and torch.allclose(b.grad, b2.grad, atol=1e-12)
    if emb:
        assert torch.allclose(e.grad, e2.grad, atol=1e-12)
    # a missing upstream gradient (an unused output) is a zero gradient, not an error
    outs = EdgeFirstLayer.apply(w, b, e)
    w.grad = None
    outs[0].sum().backward()
    assert torch.equal(w.grad[:, :2 * H], torch.ones(H, 2 * H, dtype=torch.float64)) and (w.grad[:, 2 * H:] == 0).all()


def test_committed_pmc_traffic_was_measured_on_the_committed_kernel_sources():
    """`roofline.traffic` of the benchmark line is read from a committed PMC measurement (counters cannot be collected from
    inside the process); `bench.py` reports it only when the file's `kernel_source_sha16` equals the hash of the edge
    kernel's sources in this tree.  This test keeps the committed pair consistent: editing `csrc/edge_wave.h`,
    `edge_mlp.h` or `common.h` without re-running `tools/pmc_traffic.sh` fails HERE instead of silently turning the
    driver's `traffic` into null."""
    import bench
    from diffsbdd_amd.build import kernel_source_hash
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", bench.PMC_TRAFFIC_FILE)
    assert os.path.isfile(path), path
    rec = json.load(open(path))
    assert rec["kernel_source_sha16"] == kernel_source_hash(), \
        "profiles/%s is stale: re-run tools/pmc_traffic.sh on the GPU box and commit its json" % bench.PMC_TRAFFIC_FILE
    # the figure itself: between the algorithmic bytes of the launch (65.5 MB) and a small multiple of them
    assert 60e6 < rec["traffic_bytes_per_launch"] < 200e6
    assert abs(rec["traffic_bytes_per_launch"] - (2 * rec["FETCH_SIZE_kb"] + rec["WRITE_SIZE_kb"]) * 1024) < 1.0


def test_size_distribution_log_prob_tables_equal_the_categorical_calls():
    """Round 6: log p(n1 | n2), log p(n2 | n1) and the joint log p(n1, n2) of a batch are ONE gather from a cached
    [n1][n2] table of the categoricals' logits (en_diffusion.py:958-1028 evaluates a Categorical per sample on the host) --
    the same numbers, no host loop and no device synchronisation in the training step's loss terms."""
    import numpy as np
    from diffsbdd_amd.en_diffusion import DistributionNodes
    rng = np.random.default_rng(0)
    d = DistributionNodes(rng.integers(0, 50, size=(12, 30)).astype(float))
    n1 = torch.tensor([0, 3, 11, 5, 7, 7])
    n2 = torch.tensor([29, 0, 4, 4, 10, 10])
    ref = torch.stack([d.n1_given_n2[int(c)].log_prob(i) for i, c in zip(n1, n2)])
    assert torch.equal(ref, d.log_prob_n1_given_n2(n1, n2))
    ref = torch.stack([d.n2_given_n1[int(c)].log_prob(i) for i, c in zip(n2, n1)])
    assert torch.equal(ref, d.log_prob_n2_given_n1(n2, n1))
    idx = torch.tensor([d.n_nodes_to_idx[(int(a), int(b))] for a, b in zip(n1, n2)])
    assert torch.equal(d.m.log_prob(idx), d.log_prob(n1, n2))


def test_sum_except_batch_with_a_known_batch_size():
    from diffsbdd_amd.en_diffusion import EnVariationalDiffusion as D
    x = torch.arange(24.0).view(8, 3)
    idx = torch.tensor([0, 0, 1, 1, 1, 3, 3, 3])
    assert torch.equal(D.sum_except_batch(x, idx), D.sum_except_batch(x, idx, 4))
    assert D.sum_except_batch(x, idx, 6).shape == (6,)       # trailing samples without nodes: zeros
    assert torch.equal(D.remove_mean_batch(x, idx), D.remove_mean_batch(x, idx, 4))
