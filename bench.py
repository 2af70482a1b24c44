#!/usr/bin/env python
"""Benchmark of the DDPM sampling hot path on MI355X (contract: see the task
description / DESIGN.md 6 Measurement).

    python bench.py --gpus N --steps K --warmup W

One "step" = one complete sampling chain over one batch of synthetic pockets on
BASELINE.json configs[2] (crossdock_fullatom_cond, 64 pockets per GPU, 23 ligand
atoms each, T = 500 reverse steps + the final decode = 501 EGNN evaluations).
Inputs are resident in HBM before the timed region; weights are seeded random (no
checkpoint is reachable), pockets are the 3rfm full-atom pocket fixture repeated.

Which call is timed: `value` is `ConditionalDDPM.inpaint` with every ligand atom
known ("anchored" states, `--states anchored`, the default): the same 501 EGNN calls
and fused updates as sampling, on the state distribution a TRAINED model's chain
has (ligand inside the pocket) -- a free-running chain on random weights drifts out
of the pocket and its calls get cheaper.  The metric's literal call,
`ConditionalDDPM.sample_given_pocket` free-running, is timed in the same run as
`other_states`, with its own kernel timing and roofline block.

For N > 1 either the driver launches one process per GPU with torch.distributed.run,
or a plain `python bench.py --gpus N` starts the same job itself (self_launch);
every rank runs its own 64 pockets (weak scaling, no data-path collective) and
the finished ligands are gathered once per chain over RCCL.  value = ligands of
all ranks / max-over-ranks wall time.

The JSON line also carries
  roofline        : the dominant kernel (fused GCL edge stage, edge_wave_kernel<H, MODE_GCL>,
                    csrc/edge_wave.h) timed live with HIP events on its own stream;
                    algorithmic FLOPs (E_r (H^2 + (A+2) H) MAC per launch, E_r =
                    edges of the rows the launch evaluates) / duration vs the 157.3 TFLOP/s
                    fp32 matrix peak; the same over the whole call
  cpu_baseline    : the reference's own modules (kind "reference", oracle/_ref, when the build
                    shipped them) or the CPU oracle (kind "port") timed on this host's cores on
                    a bounded sample of the same workload
  other_workloads : one chain each of configs[1] (C-alpha x 32), configs[4] on one GPU (joint
                    RePaint x 64) and the heterogeneous-pocket batch, so that the driver's
                    record holds them
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from diffsbdd_amd import sharding, synthetic  # noqa: E402
from diffsbdd_amd.pocket import prepare_pocket  # noqa: E402

FP32_MATRIX_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 / fp32 vector peak
BF16_MATRIX_PEAK_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16 MFMA (v_mfma_f32_32x32x16_bf16), no sparsity
HBM_PEAK_GBPS = 8000.0
PMC_TRAFFIC_FILE = "r6_pmc_traffic.json"
METRIC = "sampled ligands/sec (500-step DDPM, fullatom_cond) at 1/2/4/8 MI355X"

WORKLOADS = {
    # name: (arch, pocket key, default batch per GPU)
    "crossdock_fullatom_cond": ("crossdock_fullatom_cond", "fa", 64),
    "crossdock_ca_cond": ("crossdock_ca_cond", "ca", 32),
    # joint model: RePaint inpainting with the pocket fixed, resamplings=2, jump_length=1
    # (what generate_ligands does for the joint model; 999 + 1 EGNN calls at T=500)
    "moad_fullatom_joint": ("moad_fullatom_joint", "fa", 64),
}


load_pocket, anchor_ligand = synthetic.load_pocket, synthetic.anchor_ligand


def emu_dtype(k):
    return f"f32-emulated (3 x bf16 split of both operands, fp32 accumulate, {k} partial products)"


def build_model(arch, device):
    from diffsbdd_amd.conditional_model import ConditionalDDPM
    from diffsbdd_amd.dynamics import EGNNDynamics
    from diffsbdd_amd.en_diffusion import EnVariationalDiffusion
    cfg, dd = synthetic.arch_cfg(arch)
    dyn = EGNNDynamics(**cfg, device=device)
    dyn.load_state_dict(synthetic.random_state_dict(cfg, seed=0))
    cls = ConditionalDDPM if dd["conditional"] else EnVariationalDiffusion
    model = cls(dynamics=dyn, atom_nf=cfg["atom_nf"], residue_nf=cfg["residue_nf"], n_dims=3,
                            size_histogram=np.ones((40, 400)), timesteps=dd["timesteps"],
                            noise_schedule=dd["noise_schedule"], noise_precision=dd["noise_precision"],
                            loss_type="l2", norm_values=dd["norm_values"]).to(device)
    model.eval()
    return cfg, dd, model


def _cpu_problem(arch, key, b_cpu, n_lig):
    """The CPU legs' inputs: the pocket batch (normalised) and the ligand's z_T mean, as sample_given_pocket builds them."""
    from oracle import ddpm_oracle as do
    cfg, dd = synthetic.arch_cfg(arch)
    sd = synthetic.random_state_dict(cfg, seed=0)
    m = do.OracleModel(sd, cfg, cfg["atom_nf"], cfg["residue_nf"], dd["timesteps"], dd["noise_schedule"],
                       dd["noise_precision"], norm_values=dd["norm_values"], conditional=True)
    m.exact_dist = False      # torch.cdist for the radius graph, as the reference does (dynamics.py:174-181)
    pocket = load_pocket(key, b_cpu, "cpu")
    _, pocket = do.normalize(m, None, pocket)
    xh_pocket = torch.cat([pocket["x"], pocket["one_hot"]], 1)
    lig_mask = torch.repeat_interleave(torch.arange(b_cpu), n_lig)
    mu = torch.cat((do._seg_mean(pocket["x"], pocket["mask"], b_cpu), torch.zeros(b_cpu, cfg["atom_nf"])), 1)[lig_mask]
    return cfg, dd, sd, m, pocket, xh_pocket, lig_mask, mu


def cpu_port(arch, key, b_cpu, n_lig, n_calls, steps=2, max_threads=32):
    """The oracle (CPU port of the reference path) on this host's cores:
    `steps` consecutive reverse steps after one warm-up at batch b_cpu,
    extrapolated to a full chain (>= 98.6 % of a chain is the per-step dynamics
    call)."""
    from oracle import ddpm_oracle as do
    # torch's intra-op pool stops scaling long before a 256-core host is full (measured on
    # the GPU box: 256 threads -> 75 s/step at batch 16, i.e. 30x SLOWER than 8 threads on an
    # 8-core machine); use at most `max_threads` and report the number actually used.
    cores = min(os.cpu_count() or 1, max_threads)
    torch.set_num_threads(cores)
    cfg, dd, sd, m, pocket, xh_pocket, lig_mask, mu = _cpu_problem(arch, key, b_cpu, n_lig)
    tape = do.NoiseTape(1234)
    z, xp = do.cond_sample_normal_zero_com(m, mu, xh_pocket, torch.ones(b_cpu, 1), lig_mask, pocket["mask"], tape, b_cpu)
    T = dd["timesteps"]
    times = []
    with torch.no_grad():
        for i, s in enumerate(range(T - 1, T - 2 - steps, -1)):
            s_arr = torch.full((b_cpu, 1), float(s)) / T
            t_arr = torch.full((b_cpu, 1), float(s + 1)) / T
            t0 = time.perf_counter()
            z, xp = do.cond_sample_p_zs_given_zt(m, s_arr, t_arr, z, xp, lig_mask, pocket["mask"], tape)
            dt = time.perf_counter() - t0
            if i > 0:
                times.append(dt)
    t_step = float(np.mean(times))
    return {"value": b_cpu / (t_step * n_calls), "unit": "ligands/s", "cores": cores, "kind": "port",
            "sample": f"{steps} reverse steps (EGNN call + posterior update) of {arch} at batch {b_cpu} after 1 "
                      f"warm-up step, {t_step:.3f} s/step, extrapolated to {n_calls} EGNN calls per chain",
            "torch_threads": torch.get_num_threads()}


def cpu_reference(arch, key, b_cpu, n_lig, n_calls, steps=2, max_threads=32):
    """THE REFERENCE ITSELF on this host's cores: its own `ConditionalDDPM.sample_p_zs_given_zt` -> `EGNNDynamics.forward`
    (conditional_model.py:432-464, dynamics.py:87-167) imported from oracle/_ref/reference_path.zip (oracle/make_ref.py:
    the unmodified modules, shipped with the push; third-party stubs from oracle/ref_shim.py), same weights, same
    pocket batch, same protocol as cpu_port.  None when the archive was not shipped."""
    from oracle import make_ref
    if not make_ref.available():
        return None
    os.environ["DIFFSBDD_REFERENCE"] = make_ref.ARCHIVE
    import contextlib
    import io
    from oracle import ref_shim
    if ref_shim.REF_ROOT != make_ref.ARCHIVE:
        ref_shim.REF_ROOT = make_ref.ARCHIVE
    dyn_mod, en_mod, cond_mod, _ = ref_shim.import_reference()
    cores = min(os.cpu_count() or 1, max_threads)
    torch.set_num_threads(cores)
    cfg, dd, sd, _, pocket, xh_pocket, lig_mask, mu = _cpu_problem(arch, key, b_cpu, n_lig)
    with contextlib.redirect_stdout(io.StringIO()):
        d = dyn_mod.EGNNDynamics(**cfg).eval()
        d.load_state_dict(sd)
        model = cond_mod.ConditionalDDPM(dynamics=d, atom_nf=cfg["atom_nf"], residue_nf=cfg["residue_nf"], n_dims=3,
                                         size_histogram=np.ones((40, 400)), timesteps=dd["timesteps"],
                                         noise_schedule=dd["noise_schedule"], noise_precision=dd["noise_precision"],
                                         loss_type="l2", norm_values=dd["norm_values"]).eval()
    torch.manual_seed(1234)
    T = dd["timesteps"]
    times = []
    with torch.no_grad():
        z, xp = model.sample_normal_zero_com(mu, xh_pocket, torch.ones(b_cpu, 1), lig_mask, pocket["mask"])
        for i, s in enumerate(range(T - 1, T - 2 - steps, -1)):
            s_arr = torch.full((b_cpu, 1), float(s)) / T
            t_arr = torch.full((b_cpu, 1), float(s + 1)) / T
            t0 = time.perf_counter()
            z, xp = model.sample_p_zs_given_zt(s_arr, t_arr, z, xp, lig_mask, pocket["mask"])
            dt = time.perf_counter() - t0
            if i > 0:
                times.append(dt)
    t_step = float(np.mean(times))
    return {"value": b_cpu / (t_step * n_calls), "unit": "ligands/s", "cores": cores, "kind": "reference",
            "sample": f"{steps} reverse steps of the reference's own ConditionalDDPM.sample_p_zs_given_zt (oracle/_ref: its "
                      f"unmodified modules) of {arch} at batch {b_cpu} after 1 warm-up step, {t_step:.3f} s/step, "
                      f"extrapolated to {n_calls} EGNN calls per chain",
            "torch_threads": torch.get_num_threads()}


def cpu_baseline(arch, key, b_cpu, n_lig, n_calls, steps=2, max_threads=32, port_batch=16, port_steps=2):
    """kind "reference" when oracle/_ref was shipped with the push: the reference itself at batch `b_cpu` x `steps` reverse
    steps (SURVEY.md 8d's protocol: the benchmark batch, k = 5 steps after one warm-up); the port's figure then rides along
    as `port`, from a shorter sample (batch `port_batch` x `port_steps`), as a cross-check.  Without the archive: the port at
    the full sample."""
    from oracle import make_ref
    have_ref = make_ref.available()
    port = cpu_port(arch, key, port_batch if have_ref else b_cpu, n_lig, n_calls,
                    steps=port_steps if have_ref else steps, max_threads=max_threads)
    try:
        ref = cpu_reference(arch, key, b_cpu, n_lig, n_calls, steps=steps, max_threads=max_threads)
    except Exception as exc:      # a broken archive must not cost the benchmark line
        ref = None
        port["reference_error"] = repr(exc)[:200]
    if ref is None:
        return port
    ref["port"] = {k: port[k] for k in ("value", "sample")}
    return ref


def cpu_config0(max_threads=32):
    """BASELINE.json configs[0] end to end on the host, NOT extrapolated: crossdock_ca_cond, one
    pocket (3rfm), n_samples = 4, 50 DDPM steps -- the oracle's `sample_given_pocket` from z_T to the molecules."""
    from oracle import ddpm_oracle as do
    torch.set_num_threads(min(os.cpu_count() or 1, max_threads))
    cfg, dd = synthetic.arch_cfg("crossdock_ca_cond")
    sd = synthetic.random_state_dict(cfg, seed=0)
    m = do.OracleModel(sd, cfg, cfg["atom_nf"], cfg["residue_nf"], dd["timesteps"], dd["noise_schedule"],
                       dd["noise_precision"], norm_values=dd["norm_values"], conditional=True)
    m.exact_dist = False
    pocket = load_pocket("ca", 4, "cpu")
    t0 = time.perf_counter()
    with torch.no_grad():
        do.cond_sample_given_pocket(m, pocket, torch.full((4,), 23), do.NoiseTape(1), timesteps=50)
    dt = time.perf_counter() - t0
    return {"value": 4 / dt, "unit": "ligands/s", "seconds": dt, "kind": "port",
            "sample": "crossdock_ca_cond, 1 pocket (3rfm) x 4 samples x 23 ligand atoms, 50 DDPM steps + final decode = 51 "
                      "EGNN calls, whole chain timed (not extrapolated)", "torch_threads": torch.get_num_threads()}


def call_flops(cfg, lv, plan, N, E, joint):
    """Algorithmic FLOP of one EGNNDynamics.forward as the engine evaluates it (F_min restricted to
    the rows every stage computes): lv = mean node / edge counts per hop level (engine.level_stats), plan = the
    stages' radii.  Without level statistics (joint model, calls that return the pocket part): every row in every stage."""
    H, L, S = cfg["hidden_nf"], cfg["n_layers"], cfg["inv_sublayers"]
    A = 2 + (cfg.get("edge_embedding_dim") or 0)
    n_mlp = 1 if cfg["reflection_equivariant"] else 2
    G = L * S
    radii = plan[0] if plan and plan[0] else [4] * G
    if lv is None:
        nodes, edges = [N] * 5, [E] * 5
        e_upd, n_act, n_lig = (E, N, N) if joint else (E, N, N)
    else:
        nodes, edges = lv["nodes"], lv["edges"]
        e_upd, n_act, n_lig = edges[0], nodes[1], nodes[0]
    mac = 0.0
    for g in range(G):
        r = min(radii[g], 4)
        mac += edges[r] * (H * H + (A + 2) * H)                    # message stage
        mac += nodes[r] * 3 * H * H                                # node MLP
        nxt = min(min(radii[g + 1], 4) + 1, 4) if g + 1 < G else None
        if nxt is not None:
            mac += nodes[nxt] * 2 * H * H                          # next stage's P|Q
    mac += nodes[min(min(radii[0], 4) + 1, 4)] * 2 * H * H         # block 0's P|Q
    mac += L * (e_upd * n_mlp * (H * H + (A + 1) * H) + (n_act + n_lig) * n_mlp * H * H)   # coordinate stages
    mac += 2 * N * (cfg["joint_nf"] + 1) * H                       # embedding in / out
    return 2.0 * mac


def ref_flops(cfg, N, E, n_lig_nodes):
    """FLOP of one EGNNDynamics.forward as the REFERENCE evaluates it (the literal graph, SURVEY.md 8d F_ref: every edge
    MLP on every edge with its full first layer, every row in every stage) -- what the CPU baseline executes per call;
    BASELINE.md 3.5 asks for 2 F_ref / t beside the F_min-based figure."""
    H, L, S, J = cfg["hidden_nf"], cfg["n_layers"], cfg["inv_sublayers"], cfg["joint_nf"]
    A = 2 + (cfg.get("edge_embedding_dim") or 0)
    n_mlp = 1 if cfg["reflection_equivariant"] else 2
    a, r = cfg["atom_nf"], cfg["residue_nf"]
    mac = L * (E * (S + n_mlp) * ((2 * H + A) * H + H * H + H) + S * N * 3 * H * H) + 2 * N * (J + 1) * H + \
        n_lig_nodes * 2 * (2 * a * a + 2 * a * J) + (N - n_lig_nodes) * 2 * (2 * r * r + 2 * r * J)
    return 2.0 * mac


def secondary_workloads(device, n_lig_atoms, steps=3, emulated_legs=True):
    """`steps` timed chains (after one warm-up chain) of the other single-GPU configurations, so that the driver's record of
    the default run holds them: BASELINE.json configs[1] (crossdock_ca_cond x 32), configs[4] on ONE GPU
    (moad_fullatom_joint x 64, RePaint resamplings = 2: 1000 EGNN calls) and the heterogeneous-pocket variant of
    configs[2] (3rfm / 5ndu alternating, every sample under its own rotation)."""
    out = []
    for workload, pockets in (("crossdock_ca_cond", "same"), ("crossdock_fullatom_cond", "mixed"),
                              ("moad_fullatom_joint", "same")):
        arch, key, B = WORKLOADS[workload]
        cfg, dd, model = build_model(arch, device)
        model.edge_granule16 = "auto"     # opt-in: coordinate stages on the 16-edge kernels where a round is saved (C-alpha)
        model.edge_splitk = "auto"        # opt-in, per chain from host-side size bounds (EnVariationalDiffusion.splitk_auto): the
                                          # split-K kernels for stages of at most 1.5 rounds of tiles; takes precedence over granule16
        T = dd["timesteps"]
        joint = not dd["conditional"]
        n_calls = (sum(model.get_repaint_schedule(2, 1, T)) + 1) if joint else T + 1
        n_lig = torch.full((B,), n_lig_atoms, dtype=torch.int64)
        if pockets == "mixed":
            pocket0, anchor = synthetic.mixed_pockets(key, B, n_lig_atoms, cfg["atom_nf"], device)
        else:
            pocket0 = load_pocket(key, B, device)
            anchor = None if joint else anchor_ligand(B, n_lig_atoms, cfg["atom_nf"], device)

        def chain(seed):
            model.seed(seed)
            pocket = {k: v.clone() for k, v in pocket0.items()}
            if joint:
                lmask = torch.repeat_interleave(torch.arange(B, device=device), n_lig_atoms)
                ligand = {"x": torch.zeros(B * n_lig_atoms, 3, device=device),
                          "one_hot": torch.zeros(B * n_lig_atoms, cfg["atom_nf"], device=device),
                          "size": n_lig.to(device), "mask": lmask}
                return model.inpaint(ligand, pocket, torch.zeros(B * n_lig_atoms, device=device),
                                     torch.ones(pocket["x"].shape[0], device=device), resamplings=2, jump_length=1,
                                     timesteps=T)
            ligand = {k: v.clone() for k, v in anchor.items()}
            return model.inpaint(ligand, pocket, torch.ones(B * n_lig_atoms, device=device), resamplings=1, timesteps=T)

        chain(100)
        torch.cuda.synchronize(device)
        eng = model.dynamics.engine()
        lv0 = eng.level_stats(raw=True)
        t0 = time.perf_counter()
        for k in range(steps):
            out_l = chain(200 + k)[0]
        torch.cuda.synchronize(device)
        dt = (time.perf_counter() - t0) / steps
        assert torch.isfinite(out_l).all()
        N = B * n_lig_atoms + pocket0["x"].shape[0]
        lv, plan = eng.level_stats(since=lv0), eng.last_plan()
        e_last = eng.edge_count(N)
        call = call_flops(cfg, lv, plan, N, e_last, joint)
        whole = call * n_calls / dt / 1e12 if call else None
        out.append({"workload": workload, "pockets": pockets, "batch": B, "states": None if joint else "anchored",
                    "egnn_calls_per_chain": n_calls, "value": B / dt, "unit": "ligands/s", "ms_per_step": dt * 1e3,
                    "steps": steps, "warmup": 1, "whole_call_frac": (whole / FP32_MATRIX_PEAK_TFLOPS) if whole else None,
                    "reference_graph_tflops": ref_flops(cfg, N, e_last, B * n_lig_atoms) * n_calls / dt / 1e12,
                    "stage_radii": plan[0], "stage_ghost": plan[1],
                    "edge_granule16": "auto (EnVariationalDiffusion.granule16_auto): mask 0x%08x" %
                                      (eng._options.get(2, 0) & 0xFFFFFFFF),
                    "edge_splitk": "auto (EnVariationalDiffusion.splitk_auto): mask 0x%08x" % (eng._options.get(4, 0) & 0xFFFFFFFF)})
        # the same workload with the edge kernels' H x H layer emulated on the bf16 matrix cores (opt-in path, DSBDD_OPT_EMU = 6;
        # the granule / split-K masks are ignored by the engine then: every edge stage runs the emulated 32-edge kernel)
        if emulated_legs:
            try:
                model.edge_granule16 = model.edge_splitk = None
                model.edge_emulation = 6
                chain(300)
                torch.cuda.synchronize(device)
                t0 = time.perf_counter()
                for k in range(steps):
                    out_e = chain(400 + k)[0]
                torch.cuda.synchronize(device)
                dte = (time.perf_counter() - t0) / steps
                assert torch.isfinite(out_e).all()
                out[-1]["emulated"] = {"value": B / dte, "unit": "ligands/s", "ms_per_step": dte * 1e3, "steps": steps, "warmup": 1,
                                       "dtype": emu_dtype(6), "vs_exact_value": (B / dte) / out[-1]["value"]}
            except Exception as exc:      # an emulated leg must not cost the benchmark line
                out[-1]["emulated"] = {"value": None, "error": repr(exc)[:200]}
        del model, eng
        torch.cuda.empty_cache()
    return out


def training_leg(device, n_lig_atoms, steps=10, warmup=3, workload="crossdock_fullatom_cond", B=16):
    """SURVEY.md 8f-3, so that the driver's record holds it: the reference's training step (lightning_modules.py:337-363:
    `ddpm(ligand, pocket)` -> l2 objective -> `backward()` -> AdamW(amsgrad), :183-185) at the reference's batch size
    (configs/crossdock_fullatom_cond.yml:13: 16; crossdock_ca_cond.yml:13: 96), loss terms, forward AND backward on the HIP
    kernels (diffsbdd_amd/loss_head.py, train_net.py over csrc/loss_head.h, train_net.h, train.h), free-running (no
    synchronisation inside a step beyond the radius graph's edge count), batches resident.
    Roofline: forward + input-gradient + weight-gradient FLOPs of what the step evaluates (every row in every stage;
    coordinate MLPs on the ligand-row edges) against the fp32 matrix peak.  (Since round 6 the forward pass keeps the edge
    MLPs' second-layer pre-activations; the recompute FLOPs stated separately are only spent with DSBDD_TRAIN_STORE_Z2=0.)"""
    arch, key, _ = WORKLOADS[workload]
    cfg, dd, model = build_model(arch, device)
    model.train(True)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4, amsgrad=True, weight_decay=1e-12)

    def loss_of(terms):      # the network-dependent terms of the 12-tuple, reduced like lightning_modules.py:262-275
        return sum(torch.as_tensor(terms[i]).float().mean() for i in (1, 2, 4, 5, 6))

    batches = [(load_pocket(key, B, device), anchor_ligand(B, n_lig_atoms, cfg["atom_nf"], device)) for _ in range(warmup + steps)]
    for pocket, ligand in batches[:warmup]:
        opt.zero_grad(set_to_none=True)
        loss_of(model(ligand, pocket)).backward()
        opt.step()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for pocket, ligand in batches[warmup:]:
        opt.zero_grad(set_to_none=True)
        loss = loss_of(model(ligand, pocket))
        loss.backward()
        opt.step()
    torch.cuda.synchronize(device)
    dt = (time.perf_counter() - t0) / steps
    assert torch.isfinite(loss)
    pocket, ligand = load_pocket(key, B, device), anchor_ligand(B, n_lig_atoms, cfg["atom_nf"], device)
    with torch.no_grad():
        e = model.dynamics.get_edges(ligand["mask"], pocket["mask"], ligand["x"], pocket["x"])
    n_l = int(ligand["mask"].numel())
    N, E, E_u = n_l + int(pocket["mask"].numel()), int(e.shape[1]), int((e[0] < n_l).sum())
    H, L, S = cfg["hidden_nf"], cfg["n_layers"], cfg["inv_sublayers"]
    A = 2 + (cfg.get("edge_embedding_dim") or 0)
    n_mlp = 1 if cfg["reflection_equivariant"] else 2
    fwd_mac = L * (S * (E * (H * H + (A + 2) * H) + N * (2 * H * H + 3 * H * H)) +
                   n_mlp * (E_u * (H * H + (A + 1) * H) + N * 2 * H * H)) + 2 * N * (cfg["joint_nf"] + 1) * H
    flops = 3 * 2.0 * fwd_mac                                   # forward, input gradients, weight gradients
    recompute = 2.0 * L * (S * E + n_mlp * E_u) * H * H           # the second-layer pre-activations once more (kernel A)
    tfl = flops / dt / 1e12
    del model, opt
    torch.cuda.empty_cache()
    return {"workload": "training step: " + workload, "pockets": "same", "batch": B, "value": B / dt, "unit": "complexes/s",
            "ms_per_step": dt * 1e3, "steps": steps, "warmup": warmup, "dtype": "f32", "nodes": N, "edges": E,
            "edges_coordinate_stage": E_u, "optimizer": "AdamW(amsgrad)", "call": "ddpm(ligand, pocket) -> l2 terms -> "
            "backward() -> opt.step() (lightning_modules.py:337-363,183-185), loss terms, EGNN forward and backward on HIP kernels",
            "roofline": {"bound": "mfma", "achieved": tfl, "peak": FP32_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": tfl / FP32_MATRIX_PEAK_TFLOPS, "algorithmic_flops_per_step": flops,
                         "counted": "forward + dgrad + wgrad of every Linear (3 x 2 x forward MAC)",
                         "recompute_flops_per_step_not_counted_only_with_DSBDD_TRAIN_STORE_Z2_0": recompute}}


def self_launch(n, attempts=3):
    """Re-exec this script under torch.distributed.run with n ranks on this node
    (rendezvous on 127.0.0.1 and a free port); returns the job's exit code.  The port is probed and released before
    the launcher binds it, so another process may take it in between: a rendezvous that fails with EADDRINUSE is
    retried on a fresh port."""
    import socket
    import subprocess
    import tempfile
    rc = 1
    for _ in range(attempts):
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get(
            "HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        with tempfile.TemporaryFile(mode="w+") as err:
            rc = subprocess.run(cmd, env=env, stderr=err).returncode
            err.seek(0)
            text = err.read()
        sys.stderr.write(text)
        low = text.lower()
        if rc == 0 or not ("address already in use" in low or "eaddrinuse" in low):
            break
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2, help="timed sampling chains")
    ap.add_argument("--warmup", type=int, default=1, help="untimed warm-up chains")
    ap.add_argument("--workload", default="crossdock_fullatom_cond", choices=sorted(WORKLOADS))
    ap.add_argument("--batch", type=int, default=None, help="pockets per GPU")
    ap.add_argument("--timesteps", type=int, default=None, help="DDPM steps (default: the config's 500)")
    ap.add_argument("--n-lig", type=int, default=23)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=None, help="batch of the CPU baseline (default: the benchmark batch)")
    ap.add_argument("--cpu-steps", type=int, default=5, help="timed reverse steps of the CPU baseline (SURVEY.md 8d: k = 5)")
    ap.add_argument("--cpu-threads", type=int, default=32)
    ap.add_argument("--time-every", type=int, default=8,
                    help="HIP-event timing of the dominant kernel on every k-th EGNN call (1 = every call)")
    ap.add_argument("--no-kernel-timing", action="store_true",
                    help="skip the HIP-event timing of the dominant kernel (roofline = null); the engine "
                         "then replays its captured hipGraph instead of launching eagerly")
    ap.add_argument("--pockets", default="same", choices=["same", "mixed", "grouped"],
                    help="'same' (default, BASELINE configs[2]): one pocket repeated over the batch, what "
                         "prepare_pocket(repeats=n) hands the samplers; 'mixed' (SURVEY.md 8d, heterogeneous variant): "
                         "3rfm and 5ndu alternating, every sample under its own random rigid rotation (seed 0) -- "
                         "no two pockets of the batch are identical; 'grouped': the first 5/8 of the batch one pocket, "
                         "the rest distinct (a test-set batch of one large job plus refills)")
    ap.add_argument("--n-same", type=int, default=None,
                    help="--pockets grouped: number of samples that share the first pocket (default 5/8 of the batch)")
    ap.add_argument("--states", default="anchored", choices=["anchored", "free"],
                    help="pocket-conditioned workloads: 'anchored' (default, the headline) keeps every step's ligand "
                         "state on the forward process of a pose inside the pocket -- ConditionalDDPM.inpaint with "
                         "every ligand atom known --, which is the state distribution a trained model's chain has; "
                         "'free' is sample_given_pocket free-running on the random weights, whose ligand drifts out "
                         "of the pocket (fewer ligand-pocket contacts, cheaper calls).  The other one is reported "
                         "as a secondary figure.")
    ap.add_argument("--no-other-leg", action="store_true", help="skip the secondary figure (the other state model)")
    ap.add_argument("--other-steps", type=int, default=5, help="timed chains of the secondary figure")
    ap.add_argument("--secondary-steps", type=int, default=3, help="timed chains of every `other_workloads` leg")
    ap.add_argument("--no-other-workloads", action="store_true",
                    help="skip the one-chain legs of the other single-GPU configurations (configs[1], configs[4] on one "
                         "GPU, heterogeneous pockets) that ride along with the default run")
    ap.add_argument("--no-config0", action="store_true",
                    help="skip the end-to-end CPU run of BASELINE configs[0] (C-alpha, 4 samples, 50 steps)")
    ap.add_argument("--granule16", default=None,
                    help="16-edge-granule edge kernels: 'auto' or a stage bit mask (DSBDD_OPT_GRANULE16); default: off")
    ap.add_argument("--splitk", default=None,
                    help="split-K edge kernels (csrc/edge_splitk.h): 'auto' (per chain, host-side size bounds) or a stage bit "
                         "mask (DSBDD_OPT_SPLITK); default: off")
    ap.add_argument("--emulation", type=int, default=0, choices=[0, 6, 9],
                    help="arithmetic of the main legs' H x H edge layers: 0 (default) exact fp32 MFMA; 6 / 9: fp32 emulated on "
                         "the bf16 matrix cores (DSBDD_OPT_EMU); `dtype` of the line then says so")
    ap.add_argument("--no-emulated-leg", action="store_true",
                    help="skip the separate leg that runs the headline chain with the emulated-fp32 edge kernels")
    ap.add_argument("--emulated-steps", type=int, default=3, help="timed chains of the emulated leg")
    ap.add_argument("--dump-ligands", default=None,
                    help="rank 0 writes the gathered ligands of the LAST timed chain (all_lig [rows, 3 + atom_nf], all_mask = "
                         "global sample ids) to this .npz (tests: the gathered result of W ranks == W sequential world-1 runs)")
    ap.add_argument("--backend", default=None, help="torch.distributed backend (default nccl = RCCL)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="testing only: all ranks use cuda:0 (needs --backend gloo)")
    args = ap.parse_args()

    if (args.granule16 is not None or args.splitk is not None) and args.emulation:
        raise SystemExit("--granule16 / --splitk and --emulation exclude each other: those kernels have no emulated form (the engine "
                         "ignores their masks with DSBDD_OPT_EMU), the line would describe a configuration that did not run")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start one rank per GPU ourselves (the same launch the
        # driver uses) and hand the result through; rank 0 of the child job prints the JSON line
        sys.exit(self_launch(args.gpus))
    rank, local_rank, world = sharding.init_distributed(args.backend)
    if args.share_gpu:
        local_rank = 0
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} != WORLD_SIZE {world}")
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)

    arch, key, default_batch = WORKLOADS[args.workload]
    B = args.batch or default_batch
    cfg, dd, model = build_model(arch, device)
    if args.granule16 is not None:
        model.edge_granule16 = "auto" if args.granule16 == "auto" else int(args.granule16, 0)
    if args.splitk is not None:
        model.edge_splitk = "auto" if args.splitk == "auto" else int(args.splitk, 0)
    if args.emulation:
        model.edge_emulation = args.emulation
    T = args.timesteps or dd["timesteps"]
    joint = not dd["conditional"]
    n_calls = (sum(model.get_repaint_schedule(2, 1, T)) + 1) if joint else T + 1
    pocket0 = load_pocket(key, B, device)
    n_lig = torch.full((B,), args.n_lig, dtype=torch.int64)
    lo = rank * B                                   # weak scaling: every rank owns B global samples
    eng = model.dynamics.engine()
    replicas, n_streams = None, 1
    anchor = None if joint else anchor_ligand(B, args.n_lig, cfg["atom_nf"], device)
    if args.pockets in ("mixed", "grouped"):      # grouped: 5/8 of the batch one pocket, the rest singletons (40 + 24 at B = 64)
        pocket0, anchor_m = synthetic.mixed_pockets(key, B, args.n_lig, cfg["atom_nf"], device,
                                                    n_same=((args.n_same or 5 * B // 8) if args.pockets == "grouped" else 0))
        anchor = None if joint else anchor_m

    def chain(seed, states=None):
        states = states or args.states
        model.seed(seed, sample_offset=lo)
        pocket = {k: v.clone() for k, v in pocket0.items()}
        if not joint and states == "anchored" and replicas is None:
            ligand = {k: v.clone() for k, v in anchor.items()}
            out_l, out_p, lm, pm = model.inpaint(ligand, pocket, torch.ones(B * args.n_lig, device=device),
                                                 resamplings=1, timesteps=T)
        elif joint:
            lmask = torch.repeat_interleave(torch.arange(B, device=device), args.n_lig)
            ligand = {"x": torch.zeros(B * args.n_lig, 3, device=device),
                      "one_hot": torch.zeros(B * args.n_lig, cfg["atom_nf"], device=device),
                      "size": n_lig.to(device), "mask": lmask}
            out_l, out_p, lm, pm = model.inpaint(ligand, pocket, torch.zeros(B * args.n_lig, device=device),
                                                 torch.ones(pocket["x"].shape[0], device=device),
                                                 resamplings=2, jump_length=1, timesteps=T)
        else:
            out_l, out_p, lm, pm = model.sample_given_pocket(pocket, n_lig, timesteps=T)
        return sharding.gather_ligands(out_l, lm, lo)

    def barrier():
        # RCCL's barrier is a collective on a device: name it (one rank per GPU), gloo needs none
        if torch.distributed.get_backend() == "nccl":
            torch.distributed.barrier(device_ids=[local_rank])
        else:
            torch.distributed.barrier()

    def sync():
        torch.cuda.synchronize(device)
        if world > 1:
            barrier()
            torch.cuda.synchronize(device)

    red_dev = device if (world == 1 or torch.distributed.get_backend() == "nccl") else torch.device("cpu")
    for w in range(args.warmup):
        chain(100 + w)
    sync()
    per_call = cfg["n_layers"] * cfg["inv_sublayers"]
    if not args.no_kernel_timing:
        # the GCL launches of every 8th EGNN call are bracketed by HIP events (those calls run eagerly,
        # the other 7 replay the engine's captured graph, as in production)
        eng.profile(args.time_every, max_launches=(args.steps * n_calls // args.time_every + 2) * per_call)
    lv0 = eng.level_stats(raw=True)
    t0 = time.perf_counter()
    for k in range(args.steps):
        all_lig, all_mask = chain(200 + k)
    sync()
    elapsed = time.perf_counter() - t0
    kern_ms, kern_n = (eng.profile_read() if not args.no_kernel_timing else (0.0, 0))
    eng.profile(False, 0)
    lv_main = eng.level_stats(since=lv0)
    e_main = eng.edge_count(B * args.n_lig + pocket0["x"].shape[0])
    timed_level = eng.last_plan()[2]
    plan_main = eng.last_plan()
    # secondary figure: the other state model of the pocket-conditioned chain -- with --states anchored (default) this is
    # the metric's LITERAL call, sample_given_pocket free-running -- one warm-up, --other-steps timed chains, its own
    # kernel timing, level statistics and roofline block
    other = None
    if not joint and world == 1 and not args.no_other_leg:
        o_states = "free" if args.states == "anchored" else "anchored"
        chain(300, o_states)
        sync()
        if not args.no_kernel_timing:
            eng.profile(args.time_every, max_launches=(args.other_steps * n_calls // args.time_every + 2) * per_call)
        lv1 = eng.level_stats(raw=True)
        t1 = time.perf_counter()
        for k in range(args.other_steps):
            chain(301 + k, o_states)
        sync()
        el_o = time.perf_counter() - t1
        k_ms, k_n = (eng.profile_read() if not args.no_kernel_timing else (0.0, 0))
        eng.profile(False, 0)
        raw_o = (k_ms, k_n, eng.level_stats(since=lv1), eng.last_plan(), el_o, eng.edge_count(B * args.n_lig + pocket0["x"].shape[0]))
        # the same leg once more with EVERY edge stage on the split-K kernels (explicit mask, a chain constant; the per-chain
        # "auto" rule keeps full-atom x 64 on the default kernels because host-side bounds cannot tell this chain's short
        # launches -- the ligand of a random-weight chain drifts out of the pocket -- from a trained model's long ones)
        sk_var = None
        if args.splitk is None and not args.emulation and cfg["hidden_nf"] == 256:
            model.edge_splitk = -1
            chain(320, o_states)
            sync()
            t1s = time.perf_counter()
            for k in range(args.other_steps):
                chain(321 + k, o_states)
            sync()
            el_s = time.perf_counter() - t1s
            model.edge_splitk = None
            sk_var = {"value": B * args.other_steps / el_s, "unit": "ligands/s", "ms_per_step": el_s / args.other_steps * 1e3,
                      "steps": args.other_steps, "engine_option": "DSBDD_OPT_SPLITK = 0xFFFFFFFF (every edge stage on csrc/edge_splitk.h; "
                      "explicit, not the default)"}
        other = {"states": o_states, "splitk_all": sk_var,
                 "call": ("ConditionalDDPM.sample_given_pocket (conditional_model.py:478-555)"
                                              if o_states == "free" else "ConditionalDDPM.inpaint, all atoms known"),
                 "value": B * args.other_steps / el_o, "unit": "ligands/s", "ms_per_step": el_o / args.other_steps * 1e3,
                 "steps": args.other_steps, "live_levels": raw_o[2], "_raw": raw_o}
    # separate leg: the headline chain with the edge kernels' H x H layer EMULATED on the bf16 matrix cores
    # (csrc/edge_wave.h "emulated path"; same inputs, same states, same seeds as the main leg's protocol)
    emulated = None
    if world == 1 and not args.no_emulated_leg and not args.emulation and args.workload == "crossdock_fullatom_cond":
        model.edge_emulation = 6
        chain(400)
        sync()
        if not args.no_kernel_timing:
            eng.profile(args.time_every, max_launches=(args.emulated_steps * n_calls // args.time_every + 2) * per_call)
        lv2 = eng.level_stats(raw=True)
        t2 = time.perf_counter()
        for k in range(args.emulated_steps):
            chain(401 + k)
        sync()
        el_e = time.perf_counter() - t2
        k_ms, k_n = (eng.profile_read() if not args.no_kernel_timing else (0.0, 0))
        eng.profile(False, 0)
        emulated = {"dtype": emu_dtype(6), "states": args.states, "value": B * args.emulated_steps / el_e, "unit": "ligands/s",
                    "ms_per_step": el_e / args.emulated_steps * 1e3, "steps": args.emulated_steps, "warmup": 1,
                    "engine_option": "DSBDD_OPT_EMU = %d" % eng.lib.dsbdd_engine_get_option(eng.handle, 3),
                    "_raw": (k_ms, k_n, eng.level_stats(since=lv2), eng.last_plan(), el_e,
                             eng.edge_count(B * args.n_lig + pocket0["x"].shape[0]))}
        model.edge_emulation = None
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tt.item())

    n_ligands_total = B * world * args.steps
    assert all_lig.shape[0] == B * world * args.n_lig and torch.isfinite(all_lig).all()
    # global index of every rank's first sample, as the ranks used it (gathered: the record shows the sharding that ran)
    offsets = [lo]
    if world > 1:
        ot = torch.tensor([lo], dtype=torch.int64, device=red_dev)
        buf = [torch.zeros_like(ot) for _ in range(world)]
        torch.distributed.all_gather(buf, ot)
        offsets = [int(b.item()) for b in buf]
    if rank == 0 and args.dump_ligands:
        np.savez(args.dump_ligands, all_lig=all_lig.cpu().numpy(), all_mask=all_mask.cpu().numpy())

    if rank == 0:
        N = B * args.n_lig + pocket0["x"].shape[0]
        E = e_main                        # edges of the main leg's last call
        H = cfg["hidden_nf"]
        A = 2 + (cfg.get("edge_embedding_dim") or 0)

        def roofline_of(kern_ms, kern_n, lv, plan, elapsed_s, n_chains, e_last, with_traffic, emu=0):
            """The roofline block of one leg: the dominant kernel's timed launches and the whole call.  emu = k: the launches
            ran the emulated path -- `achieved` / `peak` / `frac` are then the bf16 matrix FLOPs the kernel EXECUTES
            (k products x 2 E H^2) against the dense bf16 peak; the algorithmic fp32 figure rides along."""
            t_level = plan[2]
            # the timed launches are those of the largest radius of the call's plan (csrc/engine.hip): the rows of
            # level <= t_level, a prefix of the edge list whose mean length the engine accumulated
            E_timed = e_last
            if lv is not None and t_level < 4:
                E_timed = lv["edges"][t_level]
            flops_per_launch = 2.0 * E_timed * (H * H + (A + 2) * H)
            avg_ms = kern_ms / max(kern_n, 1)
            achieved = flops_per_launch / (avg_ms * 1e-3) / 1e12 if kern_n else None
            # algorithmic HBM bytes of the same launch: P|Q read once per node, W2^T, edge list, agg written
            bytes_per_launch = 4.0 * (N * 2 * H + H * H + 3 * E_timed + 3 * N + N * H)
            traffic, traffic_src = None, None
            tpath = os.path.join(ROOT, "profiles", PMC_TRAFFIC_FILE)
            if with_traffic and args.workload == "crossdock_fullatom_cond" and B == 64 and args.states == "anchored" and \
                    args.pockets == "same" and not args.emulation and os.path.isfile(tpath):
                # PMC counters cannot be read from inside this process; the figure is the one measured
                # with `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` on this same workload (tools/pmc_traffic.sh) --
                # valid only for the kernel it was measured on: the file carries the hash of the kernel's sources
                from diffsbdd_amd.build import kernel_source_hash
                tj = json.load(open(tpath))
                now = kernel_source_hash()
                if tj.get("kernel_source_sha16") == now:
                    traffic = tj["traffic_bytes_per_launch"]
                    traffic_src = (f"profiles/{PMC_TRAFFIC_FILE} (rocprofv3 --pmc, gfx950-corrected), measured on kernel "
                                   f"sources sha256/16 {now}")
                else:
                    traffic_src = (f"stale: profiles/{PMC_TRAFFIC_FILE} was measured on kernel sources "
                                   f"{tj.get('kernel_source_sha16')}, this build is {now} (re-run tools/pmc_traffic.sh)")
            # algorithmic work of a WHOLE call from what the stages evaluated (mean over the timed chains): SURVEY.md
            # 8d's F_min restricted to the rows / edges of every stage's radius
            call = call_flops(cfg, lv, plan, N, e_last, joint)
            whole = call * n_calls * n_chains / elapsed_s / 1e12 if call else None
            head = {"bound": "mfma", "kernel": "edge_wave_kernel<H, MODE_GCL> (fused GCL edge stage, csrc/edge_wave.h)",
                    "achieved": achieved, "peak": FP32_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
                    "frac": (achieved / FP32_MATRIX_PEAK_TFLOPS) if achieved else None}
            if emu:
                ex = emu * 2.0 * E_timed * H * H / (avg_ms * 1e-3) / 1e12 if kern_n else None
                head = {"bound": "mfma", "kernel": f"edge_wave_kernel<H, MODE_GCL, EMU = {emu}> (csrc/edge_wave.h, emulated path)",
                        "achieved": ex, "peak": BF16_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s (bf16 MFMA FLOPs executed: "
                        f"{emu} partial products x 2 E H^2)", "frac": (ex / BF16_MATRIX_PEAK_TFLOPS) if ex else None,
                        "algorithmic_fp32_tflops": achieved,
                        "algorithmic_fp32_vs_exact_peak": (achieved / FP32_MATRIX_PEAK_TFLOPS) if achieved else None}
            return {
                **head,
                "traffic": traffic, "traffic_unit": "HBM bytes per launch", "traffic_source": traffic_src,
                "avg_launch_ms": avg_ms, "timed_launches": kern_n, "edges_per_launch": E_timed,
                "algorithmic_flops_per_launch": flops_per_launch,
                # timed = the message-stage launches that run over the WHOLE edge list (same work every launch).
                # Pocket-conditioned chains: block 0 is split by the pocket frame and the last stages run on prefixes
                # of the level-ordered list (csrc/graph.h), so fewer than n_layers launches per call qualify.
                "timed_launch_kind": ("full edge list" if lv is None or t_level >= 4
                                      else f"rows of hop level <= {t_level} (the call's largest message-stage launches)"),
                # share of the leg's wall time spent in the timed launches (timed on every k-th call only)
                "kernel_share_of_wall": (kern_ms * args.time_every / (elapsed_s * 1e3)) if kern_n else None,
                # the same roofline over the WHOLE call: algorithmic FLOP of everything a call evaluates / wall time
                "whole_call_tflops": whole, "whole_call_frac": (whole / FP32_MATRIX_PEAK_TFLOPS) if whole else None,
                # the same wall time priced with the work of the REFERENCE's literal graph (what the CPU
                # baseline executes per call) -- not a hardware rate, may exceed the peak: the exact algebraic savings
                "reference_graph_flops_per_call": ref_flops(cfg, N, e_last, B * args.n_lig),
                "reference_graph_tflops": ref_flops(cfg, N, e_last, B * args.n_lig) * n_calls * n_chains / elapsed_s / 1e12,
                "algorithmic_flops_per_call": call, "stage_radii": plan[0], "stage_ghost": plan[1],
                # mean over the calls of the chain: nodes / edge-list slots with hop level <= r (r = 0: ligand rows,
                # r = 4: everything); message stage g of G evaluates level <= G - g
                "live_levels": lv,
                "hbm_algorithmic_gbps": bytes_per_launch / (avg_ms * 1e-3) / 1e9 if kern_n else None,
                "hbm_frac_of_8TBps": (bytes_per_launch / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS) if kern_n else None,
            }

        roofline = roofline_of(kern_ms, kern_n, lv_main, plan_main, elapsed, args.steps, e_main, True, emu=args.emulation)
        if other is not None:
            k_ms, k_n, lv_o, plan_o, el_o, e_o = other.pop("_raw")
            other["roofline"] = roofline_of(k_ms, k_n, lv_o, plan_o, el_o, args.other_steps, e_o, False, emu=args.emulation)
        if emulated is not None:
            k_ms, k_n, lv_e, plan_e, el_e, e_e = emulated.pop("_raw")
            emulated["roofline"] = roofline_of(k_ms, k_n, lv_e, plan_e, el_e, args.emulated_steps, e_e, False, emu=6)
            emulated["vs_exact_value"] = emulated["value"] / (n_ligands_total / elapsed)
            emulated["gate"] = ("opt-in (ddpm.edge_emulation = 6 / DSBDD_OPT_EMU); every -m gpu parity test holds at 1e-4 with it on, "
                                "error vs float64 <= 2 x the exact path's: tests/test_gpu_emu.py, profiles/r5_emu_*")
        other_workloads = None
        if world == 1 and not args.no_other_workloads and args.workload == "crossdock_fullatom_cond" and \
                args.pockets == "same" and args.timesteps is None and args.batch is None:
            other_workloads = secondary_workloads(device, args.n_lig, steps=args.secondary_steps, emulated_legs=not args.no_emulated_leg)
            for wl_, b_ in (("crossdock_fullatom_cond", 16), ("crossdock_ca_cond", 96)):
                try:
                    other_workloads.append(training_leg(device, args.n_lig, workload=wl_, B=b_))
                except Exception as exc:      # the training legs must not cost the benchmark line
                    other_workloads.append({"workload": "training step: " + wl_, "pockets": "same", "value": None,
                                            "error": repr(exc)[:300]})
        cpu = None
        if world == 1 and not args.no_cpu_baseline and not joint:
            cpu = cpu_baseline(arch, key, args.cpu_batch or B, args.n_lig, n_calls, steps=args.cpu_steps,
                               max_threads=args.cpu_threads)
            if not args.no_config0:
                cpu["config0"] = cpu_config0(max_threads=args.cpu_threads)
        value = n_ligands_total / elapsed
        pocket_desc = (f"3rfm {key} pocket, {pocket0['x'].shape[0] // B} nodes" if args.pockets == "same" else
                       f"3rfm / 5ndu {key} pockets alternating, each under its own rigid rotation: no two identical"
                       if args.pockets == "mixed" else
                       f"{args.n_same or 5 * B // 8} x one 3rfm {key} pocket + {B - (args.n_same or 5 * B // 8)} distinct 3rfm / 5ndu pockets")
        line = {
            "metric": METRIC, "value": value, "unit": "ligands/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": emu_dtype(args.emulation) if args.emulation else "f32",
            "data": "synthetic",
            # (states / pockets / batch first: the driver's record keeps the first 120 characters)
            "config": {"workload": f"{args.workload} states={'n/a' if joint else args.states}"
                                   f"{'' if joint or args.states != 'anchored' else ' (inpaint, all atoms known)'} "
                                   f"pockets={args.pockets} batch={B}/GPU T={T}: ({pocket_desc}) x {args.n_lig} ligand atoms, "
                                   f"T={T} reverse steps" + (" (RePaint, resamplings=2)" if joint else "") +
                                   f" + final decode = {n_calls} EGNN calls per chain" +
                                   ("" if joint else (", ligand states anchored to the forward process of a pose in "
                                                      "the pocket (inpaint, all atoms known)" if args.states == "anchored"
                                                      else ", free-running on random weights")),
                       "states": None if joint else args.states, "pockets": args.pockets,
                       "batch_per_gpu": B, "global_batch": B * world, "timesteps": T,
                       "nodes_per_gpu": N, "edges_per_call": E, "parallelism": f"dp{world} (pocket sharding)",
                       "weights": "seeded random (diffsbdd_amd/synthetic.py, seed 0)"},
            "roofline": roofline, "cpu_baseline": cpu, "other_states": other, "emulated": emulated,
            "other_workloads": other_workloads,
            "speedup_vs_cpu_baseline": (value / world / cpu["value"]) if cpu else None,
            "hipgraph": dict(zip(("replays", "captures", "eager_calls"), eng.graph_stats())),
            "rccl_ranks": torch.distributed.get_world_size() if world > 1 else 1,
            "rank_sample_offsets": offsets,
            "backend": torch.distributed.get_backend() if world > 1 else None,
            "host_cores": os.cpu_count(),
        }

        def r3(v):
            return None if v is None else float(f"{v:.4g}")

        def wl(name, pockets=None):
            for w_ in other_workloads or []:
                if w_.get("workload") == name and (pockets is None or w_.get("pockets") == pockets):
                    return w_
            return None
        # LAST key, compact (< 1200 characters): the line's secondary figures where the driver's record (which keeps the
        # tail of the output) can see them
        ca, mixed_w, joint_w = wl("crossdock_ca_cond"), wl("crossdock_fullatom_cond", "mixed"), wl("moad_fullatom_joint")
        train_w = wl("training step: crossdock_fullatom_cond")
        summ = {"value": r3(value), "dom_frac": r3(roofline["frac"]), "whole_frac": r3(roofline["whole_call_frac"]),
                "x_cpu": r3(line["speedup_vs_cpu_baseline"])}
        if other is not None:
            summ[other["states"]] = {"value": r3(other["value"]), "dom_frac": r3(other["roofline"]["frac"]),
                                     "whole_frac": r3(other["roofline"]["whole_call_frac"])}
            if other.get("splitk_all"):
                summ[other["states"]]["splitk_all"] = r3(other["splitk_all"]["value"])
        if emulated is not None:
            summ["emulated"] = {"value": r3(emulated["value"]), "bf16_frac": r3(emulated["roofline"]["frac"]), "dtype": "f32-emulated(bf16x3,6)"}
        for k_, w_ in (("ca32", ca), ("mixed", mixed_w), ("joint", joint_w)):
            if w_ is not None:
                summ[k_] = {"value": r3(w_.get("value")), "whole_frac": r3(w_.get("whole_call_frac"))}
                if w_.get("emulated"):
                    summ[k_]["emulated"] = r3(w_["emulated"].get("value"))
        if train_w is not None and train_w.get("value") is not None:
            summ["train"] = {"ms": r3(train_w["ms_per_step"]), "frac": r3(train_w["roofline"]["frac"])}
        train_ca = wl("training step: crossdock_ca_cond")
        if train_ca is not None and train_ca.get("value") is not None:
            summ["train_ca96"] = {"ms": r3(train_ca["ms_per_step"]), "frac": r3(train_ca["roofline"]["frac"])}
        line["summary"] = summ
        assert len(json.dumps(summ)) < 1200
        print(json.dumps(line), flush=True)
    if world > 1:
        barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
