"""GPU parity at the BASELINE.json batch sizes (-m gpu): the exact problems
bench.py times (3rfm pocket repeated, 23 ligand atoms per sample, seeded
weights) against the CPU oracle evaluated on the spot (the literal reference
graph, ~5-15 s per call on 32 host threads).

These are the launches whose persistent grids are saturated: E > 512 * 128
edges, so every workgroup of the edge kernels walks several tiles (next-tile
prefetch, commit_edge, the continuous W2^T stream across units, the (tile,
MLP) work items of the coordinate stage).  Tolerance: 1e-4 absolute per
timestep / per block (BASELINE.json north_star), stated next to each assert.
"""
import os

import numpy as np
import pytest
import torch

from oracle import ddpm_oracle as do
from oracle import egnn_oracle as eo
from oracle import weights as W
from tests._golden import GOLDEN_DIR

pytestmark = pytest.mark.gpu
TOL = 1e-4
RESIDENT_TILES = 512            # 2 workgroups x 256 CUs, 128 edges per workgroup tile


def dev():
    return torch.device("cuda:0")


class oracle_threads:
    """torch's intra-op pool stops scaling far below the 256 cores of the GPU host
    (bench.py: 256 threads are 30x slower than 8); the oracle runs on <= 32."""

    def __enter__(self):
        self.old = torch.get_num_threads()
        torch.set_num_threads(min(os.cpu_count() or 1, 32))

    def __exit__(self, *a):
        torch.set_num_threads(self.old)


def excess(a, b, atol=TOL, rtol=1e-5):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return ((a - b).abs() - (atol + rtol * b.abs())).max().item()


def bench_problem(arch, B, n_lig=23, seed=0, t_value=0.6):
    """The bench.py workload as plain tensors: normalised pocket, a ligand state
    z_t around the pocket centre, one t for the batch."""
    from diffsbdd_amd.pocket import prepare_pocket
    cfg, dd = W.arch_cfg(arch)
    z = np.load(os.path.join(GOLDEN_DIR, "pocket_3rfm.npz"))
    key = "ca" if cfg["residue_nf"] == 20 else "fa"
    pocket = prepare_pocket(z[key + "_x"], z[key + "_types"], cfg["residue_nf"], repeats=B)
    nx, nh = dd["norm_values"]
    xp = pocket["x"].float() / nx
    hp = pocket["one_hot"].float() / nh
    mp = pocket["mask"].long()
    g = torch.Generator().manual_seed(seed)
    ml = torch.repeat_interleave(torch.arange(B), n_lig)
    com = eo.segment_mean(xp, mp, B)
    # per-sample rigid shifts so that the samples are not copies of each other
    xl = com[ml] + torch.randn(B * n_lig, 3, generator=g) * (1.0 / nx) * 1.5
    hl = torch.randn(B * n_lig, cfg["atom_nf"], generator=g) * 0.5
    if dd["conditional"]:
        lig_com = eo.segment_mean(xl, ml, B)
        xl, xp = xl - lig_com[ml], xp - lig_com[mp]
    else:   # joint model: the pocket is a noised state as well, COM of the complex removed
        xp = xp + torch.randn(xp.shape, generator=g) * (0.3 / nx)
        hp = hp + torch.randn(hp.shape, generator=g) * 0.3
        allx = torch.cat([xl, xp])
        m = eo.segment_mean(allx, torch.cat([ml, mp]), B)
        xl, xp = xl - m[ml], xp - m[mp]
    t = torch.full((B, 1), t_value)
    return cfg, dd, torch.cat([xl, hl], 1), torch.cat([xp, hp], 1), t, ml, mp


def make_dynamics(cfg, sd):
    from diffsbdd_amd.dynamics import EGNNDynamics
    m = EGNNDynamics(**cfg, device=dev())
    m.load_state_dict(sd)
    return m.eval()      # inference: the HIP kernels (training mode + autograd = the differentiable path, train_path.py)


def edge_flips(ours, ref, n):
    """Number of (row, col) pairs present in exactly one of the two lists."""
    a = ours[0] * n + ours[1]
    b = ref[0] * n + ref[1]
    return int(np.setxor1d(a.numpy(), b.numpy()).size)


@pytest.mark.parametrize("arch,B,saturated", [
    ("crossdock_fullatom_cond", 64, True),     # BASELINE configs[2]: the bench line
    ("moad_fullatom_joint", 64, True),         # configs[4]: H = 192, edge-type table, all rows updated
    ("crossdock_ca_cond", 32, False),          # configs[1]: the latency regime
])
def test_bench_problem_forward_vs_oracle(arch, B, saturated):
    """One EGNNDynamics.forward of the benchmark problem: eps and every block's
    (h, x) against the oracle, through (i) the public call that builds the radius
    graph on the device and (ii) the teacher-forced edge list."""
    cfg, dd, xl, xp, t, ml, mp = bench_problem(arch, B)
    sd = W.random_state_dict(cfg, 0)
    N = len(ml) + len(mp)
    m = make_dynamics(cfg, sd)
    d = dev()
    # (i) public API, device-built edges (second and third call: captured graph / replay)
    args_d = [v.to(d) for v in (xl, xp, t, ml, mp)]
    for _ in range(3):
        f_l, f_p = m(*args_d)
    er, ec = m.engine().last_edges(N)
    ours = torch.stack([er, ec])
    E = ours.shape[1]
    if saturated:
        assert E > RESIDENT_TILES * 128, f"E = {E}: the persistent grid would not be saturated"
    with oracle_threads():
        ref_edges = eo.get_edges_blockwise(ml, mp, xl[:, :3], xp[:, :3], cfg["edge_cutoff_ligand"],
                                           cfg["edge_cutoff_pocket"], cfg["edge_cutoff_interaction"])
        flips = edge_flips(ours, ref_edges, N)
        print(f"[{arch} B={B}] N={N} E={E} tiles={-(-E // 128)} edge flips vs exact CPU builder: {flips}")
        # an exact tie at the cutoff may round differently (fp32 summation order of dx^2+dy^2+dz^2)
        assert flips <= max(4, E // 50000), flips
        trace = []
        o_l, o_p, _ = eo.dynamics_forward(sd, cfg, xl, xp, t, ml, mp, edges=ours, trace=trace)
    assert excess(f_l, o_l) <= 0 and excess(f_p, o_p) <= 0, (excess(f_l, o_l), excess(f_p, o_p))   # 1e-4
    # (ii) teacher-forced edges + per-block trace
    th, tx = m.engine().set_trace(N)
    e_l, e_p, status = m.forward_async(xl, xp, t, ml, mp, edges=ours)
    torch.cuda.synchronize()
    m.engine().clear_trace()
    assert int(status.item()) == 0
    worst_x = worst_h = 0.0
    for i, (h, x) in enumerate(trace):
        ex = (tx[i].cpu() - x).abs().max().item()
        eh = (th[i].cpu() - h).abs().max().item() / max(1.0, h.abs().max().item())
        worst_x, worst_h = max(worst_x, ex), max(worst_h, eh)
        assert ex < TOL and eh < TOL, (arch, i, ex, eh)                                              # 1e-4
    print(f"[{arch} B={B}] worst per-block error: x {worst_x:.2e}, h (relative to max|h|) {worst_h:.2e}; "
          f"eps {max((e_l.cpu() - o_l).abs().max().item(), (e_p.cpu() - o_p).abs().max().item()):.2e}")
    assert excess(e_l, o_l) <= 0 and excess(e_p, o_p) <= 0
    # the two call paths agree to summation order
    assert (e_l - f_l).abs().max().item() < 1e-5


def _make_ddpm(arch, sd):
    from diffsbdd_amd.conditional_model import ConditionalDDPM
    from diffsbdd_amd.en_diffusion import EnVariationalDiffusion
    cfg, dd = W.arch_cfg(arch)
    cls = ConditionalDDPM if dd["conditional"] else EnVariationalDiffusion
    return cls(dynamics=make_dynamics(cfg, sd), atom_nf=cfg["atom_nf"], residue_nf=cfg["residue_nf"], n_dims=3,
               size_histogram=np.ones((4, 8)), timesteps=dd["timesteps"], noise_schedule=dd["noise_schedule"],
               noise_precision=dd["noise_precision"], loss_type="l2", norm_values=dd["norm_values"]).to(dev())


@pytest.mark.parametrize("arch,B,n_steps", [("crossdock_fullatom_cond", 64, 3), ("moad_fullatom_joint", 64, 2),
                                            ("crossdock_ca_cond", 32, 3)])       # configs[2], [4], [1]
def test_bench_problem_reverse_steps_teacher_forced(arch, B, n_steps):
    """sample_p_zs_given_zt (conditional_model.py:432-464 / en_diffusion.py:503-557) at the
    benchmark batch: the oracle's z_t goes into both sides every step, the same injected
    noise, the device-built edge list handed to the oracle; z_s within 1e-4 per timestep."""
    cfg, dd, xl, xp, _, ml, mp = bench_problem(arch, B, seed=1)
    sd = W.random_state_dict(cfg, 0)
    N = len(ml) + len(mp)
    model = _make_ddpm(arch, sd)
    om = do.OracleModel(sd, cfg, cfg["atom_nf"], cfg["residue_nf"], dd["timesteps"], dd["noise_schedule"],
                        dd["noise_precision"], norm_values=dd["norm_values"], conditional=dd["conditional"])
    d = dev()
    T = dd["timesteps"]
    z_l, z_p = xl, xp
    eng = model.dynamics.engine()
    worst = -1.0
    for k, s_int in enumerate(range(T - 200, T - 200 - n_steps, -1)):
        s = torch.full((B, 1), float(s_int)) / T
        t = torch.full((B, 1), float(s_int + 1)) / T
        tape = do.NoiseTape(77 + k)
        # HIP side first (its radius graph is then teacher-forced into the oracle)
        pre = do.NoiseTape(77 + k)
        shapes = [(len(ml), 3 + cfg["atom_nf"])] if dd["conditional"] else \
            [(N, 3), (len(ml), cfg["atom_nf"]), (len(mp), cfg["residue_nf"])]
        draws = [pre(sh) for sh in shapes]
        model.set_noise_source(do.NoiseReplay(draws))
        h_l, h_p = model.sample_p_zs_given_zt(s.to(d), t.to(d), z_l.to(d), z_p.to(d), ml.to(d), mp.to(d))
        er, ec = eng.last_edges(N)
        if k == 0 and "ca_" not in arch:          # the C-alpha problem is the latency regime (153 tiles)
            assert er.numel() > RESIDENT_TILES * 128
        om.edge_hook = lambda i, e=torch.stack([er, ec]): e
        with oracle_threads():
            if dd["conditional"]:
                o_l, o_p = do.cond_sample_p_zs_given_zt(om, s, t, z_l, z_p, ml, mp, tape)
            else:
                o_l, o_p = do.joint_sample_p_zs_given_zt(om, s, t, z_l, z_p, ml, mp, tape)
        worst = max(worst, excess(h_l, o_l), excess(h_p, o_p))
        z_l, z_p = o_l, o_p          # teacher forcing: the oracle's state feeds the next step
    assert worst <= 0, worst         # 1e-4 per timestep (+ rtol 1e-5 on large joint states)


def test_pocket_frame_block0_split_and_shared_pockets():
    """Pocket frame (csrc/engine.hip, dsbdd_engine_set_pocket_frame): block 0 evaluates the pocket-pocket
    messages on the chain's raw pocket coordinates, separately from the edges with a ligand endpoint, and for
    a batch of identical pockets only once.  (i) parity with the oracle at 1e-4 while the pocket is translated
    per sample (what the reverse steps do), (ii) the shared and the per-sample evaluation agree bit for bit,
    (iii) so do a batch and its halves (frames of their own)."""
    from diffsbdd_amd.engine import edge_capacity
    B = 8
    cfg, dd, xl, xp, t, ml, mp = bench_problem("crossdock_fullatom_cond", B)
    sd = W.random_state_dict(cfg, 0)
    d = dev()
    N = len(ml) + len(mp)
    n0 = len(mp) // B
    # raw pocket = sample 0's coordinates for every sample; the chain state is a per-sample translation of it
    raw = xp[:n0, :3].repeat(B, 1)
    g = torch.Generator().manual_seed(3)
    shift = torch.randn(B, 3, generator=g)
    xp_t = torch.cat([raw + shift[mp], xp[:, 3:]], 1)
    xl_t = torch.cat([xl[:, :3] + shift[ml], xl[:, 3:]], 1)
    sizes = torch.full((B,), n0)

    def run(shared, sl=slice(None), sp=slice(None), batch=B):
        m = make_dynamics(cfg, sd)
        eng = m.engine()
        a = [v.to(d) for v in (xl_t[sl], xp_t[sp], t[:batch], ml[sl] - ml[sl][0], mp[sp] - mp[sp][0])]
        cap = edge_capacity(a[3], a[4], batch)
        eng.set_pocket_frame(raw[sp].to(d), a[4], sizes[:batch].to(d), a[0].shape[0], batch, cap, shared)
        outs = [m.forward_async(*a, batch=batch, edge_cap=cap) for _ in range(3)]      # eager, capture, replay
        torch.cuda.synchronize()
        assert all(int(o[2].item()) == 0 for o in outs)
        assert torch.equal(outs[0][0], outs[2][0]) and torch.equal(outs[0][1], outs[2][1])
        er, ec = eng.last_edges(a[0].shape[0] + a[1].shape[0])
        eng.clear_pocket_frame()
        return outs[0][0], outs[0][1], torch.stack([er, ec])

    s_l, s_p, edges = run(True)
    u_l, u_p, _ = run(False)
    assert torch.equal(s_l, u_l) and torch.equal(s_p, u_p)                      # (ii)
    with oracle_threads():
        o_l, o_p, _ = eo.dynamics_forward(sd, cfg, xl_t, xp_t, t, ml, mp, edges=edges)
    assert excess(s_l, o_l) <= 0 and excess(s_p, o_p) <= 0                      # (i) 1e-4
    nl = len(ml) // B
    h_l, h_p, _ = run(True, slice(4 * nl, None), slice(4 * n0, None), 4)        # samples 4..7 as their own batch
    assert torch.equal(h_l, s_l[4 * nl:]) and torch.equal(h_p, s_p[4 * n0:])    # (iii)
    one_l, one_p, _ = run(False, slice(7 * nl, None), slice(7 * n0, None), 1)   # a single sample, not shared
    assert torch.equal(one_l, s_l[7 * nl:]) and torch.equal(one_p, s_p[7 * n0:])
    # without a frame (plain forward) the association of block 0 differs: same numbers to rounding
    m = make_dynamics(cfg, sd)
    p_l, p_p = m(*[v.to(d) for v in (xl_t, xp_t, t, ml, mp)])
    assert (p_l - s_l).abs().max().item() < 1e-5 and (p_p - s_p).abs().max().item() < 1e-5


def _hop_levels(row, col, n_lig, n_nodes, n_levels=5):
    """Reference for csrc/graph.h levels_kernel: graph distance to the nearest ligand node, capped."""
    lvl = np.full(n_nodes, n_levels - 1, dtype=np.int64)
    lvl[:n_lig] = 0
    for k in range(1, n_levels - 1):
        src = lvl[col] == k - 1
        hit = np.zeros(n_nodes, dtype=bool)
        hit[row[src]] = True
        lvl[hit & (lvl == n_levels - 1)] = k
    return lvl


@pytest.mark.parametrize("arch,B,frame", [
    ("crossdock_fullatom_cond", 8, False),
    ("crossdock_fullatom_cond", 8, True),
    ("crossdock_ca_cond", 8, False),
    ("crossdock_fullatom_cond", 8, "shared"),   # identical pockets: forward cone on top (ghost rows of the canonical pocket)
    ("small_cond", 6, True),              # two blocks: block 0 is a pruned stage AND split by the pocket frame
    ("small_cond", 6, "shared"),
    ("small_variant", 6, False),          # two sublayers per block, ligand cutoff, E(3) variant
    ("crossdock_fullatom_cond", 8, "shared8"),  # the same with the persistent edge grid capped at 8 workgroups
                                                # (DSBDD_EDGE_MAX_WG): ghost rows x several tiles per workgroup
])
def test_ligand_only_call_evaluates_live_rows(arch, B, frame, monkeypatch):
    """A pocket-conditioned call that returns the ligand part only (what ConditionalDDPM's chains ask for,
    conditional_model.py:268-272) evaluates, per message stage, the rows the ligand output depends on -- prefixes
    of the level-ordered edge list (csrc/graph.h).  (i) the level structures against a numpy BFS over the
    natural-order list, (ii) ligand eps against the all-rows call (different summation tiles: 2e-5) and the
    oracle (1e-4), (iii) eager / captured / replayed calls and a batch vs its second half: bit-identical."""
    from diffsbdd_amd.engine import edge_capacity
    cfg, dd, xl, xp, t, ml, mp = bench_problem(arch, B)
    sd = W.random_state_dict(cfg, 0)
    d = dev()
    nl, n0 = len(ml) // B, len(mp) // B
    sizes = torch.full((B,), n0)

    if frame == "shared8":
        monkeypatch.setenv("DSBDD_EDGE_MAX_WG", "8")      # read when the engine is created (make_dynamics below)
        frame = "shared"
    shared = frame == "shared"
    from diffsbdd_amd import _lib
    if shared:      # identical pockets: every sample's pocket is EXACTLY sample 0's plus a translation (what a chain has)
        raw = xp[:n0, :3].repeat(B, 1)
        delta = (xp[:, :3] - raw).view(B, n0, 3)[:, 0]
        xp = torch.cat([raw + delta[mp], xp[:, 3:]], 1)
    else:
        raw = xp[:, :3]

    def run(want_pocket, lo=0, cone=None):
        # the mode a chain pins per batch (EnVariationalDiffusion._cone_for_groups): on for identical pockets, off otherwise
        # -- never the engine's own size-dependent rule, which may differ between a batch and its half
        cone = (2 if shared else 0) if cone is None else int(cone)
        sl, sp, batch = slice(lo * nl, None), slice(lo * n0, None), B - lo
        m = make_dynamics(cfg, sd)
        eng = m.engine()
        a = [v.to(d) for v in (xl[sl], xp[sp], t[:1] if shared else t[:batch], ml[sl] - lo, mp[sp] - lo)]
        cap = edge_capacity(a[3], a[4], batch)
        if frame:
            eng.set_pocket_frame(raw[sp].to(d), a[4], sizes[:batch].to(d), a[0].shape[0], batch, cap, shared)
        eng.set_option(_lib.OPT_CONE, cone)
        outs = [m.forward_async(*a, batch=batch, edge_cap=cap, want_pocket=want_pocket) for _ in range(3)]
        torch.cuda.synchronize()
        assert all(int(o[2].item()) == 0 for o in outs)
        assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][0], outs[2][0])      # (iii)
        n = a[0].shape[0] + a[1].shape[0]
        er, ec = eng.last_edges(n)
        lv = eng.last_levels(n) if not want_pocket else None
        plan = eng.last_plan()
        if frame:
            eng.clear_pocket_frame()
        return outs[0][0], torch.stack([er, ec]), lv, plan

    full, edges, _, plan = run(True)
    assert set(plan[0]) == {4} and not any(plan[1])                 # every row in every stage
    lig, edges2, lv, plan = run(False)
    G = cfg["n_layers"] * cfg["inv_sublayers"]
    if shared:      # forward cone: stage g computes level <= min(g + 1, G - g), ghosts while the radii ascend
        assert plan[0] == [min(g + 1, G - g, 4) for g in range(G)]
        assert plan[1] == [int(g + 1 < G and min(min(g + 2, G - g - 1) + 1, 4) > min(g + 1, G - g, 4)) for g in range(G)]
        nocone, _, _, plan2 = run(False, cone=False)
        assert plan2[0] == [min(G - g, 4) for g in range(G)] and not any(plan2[1])
        assert (nocone - lig).abs().max().item() < 2e-5
    else:
        assert plan[0] == [min(G - g, 4) for g in range(G)] and not any(plan[1])
    assert torch.equal(edges, edges2)
    N = len(ml) + len(mp)
    row, col = edges[0].numpy(), edges[1].numpy()
    # (i) levels, their order and the re-ordered list
    want = _hop_levels(row, col, len(ml), N)
    assert np.array_equal(lv["level"], want)
    order = np.lexsort((np.arange(N), want))
    assert np.array_equal(lv["order"], order)
    assert np.array_equal(lv["count"], np.cumsum(np.bincount(want, minlength=5)))
    deg = np.bincount(row, minlength=N)
    assert np.array_equal(lv["deg"], deg)
    nat_ptr = np.concatenate([[0], np.cumsum(deg)])            # natural list without its padding
    ghost_slots = 0
    if frame:       # the pocket-pocket lists of the frame's pockets (one, or every sample's) sit in front of the list
        for b in range(1 if shared else B):
            lo_b = len(ml) + b * n0
            ghost_slots += (int(((row >= lo_b) & (col >= len(ml)) & (row < lo_b + n0)).sum()) + 31) // 32 * 32
    assert lv["ghost_slots"] == ghost_slots and lv["ghost_nodes"] == (0 if not frame else (n0 if shared else B * n0))
    pos = ghost_slots
    batch_of = np.concatenate([ml.numpy(), mp.numpy()])
    prev_key = None
    level_last = {}
    for i in order:
        key = (want[i], batch_of[i])
        if key != prev_key:
            pos = (pos + 31) // 32 * 32                        # every (level, sample) segment starts a wave tile
            prev_key = key
        assert lv["row_ptr"][i] == pos
        s = slice(pos, pos + deg[i])
        assert (lv["row"][s] == i).all()
        assert np.array_equal(lv["col"][s], col[nat_ptr[i]:nat_ptr[i + 1]])
        pos += deg[i]
        level_last[want[i]] = pos
    running = ghost_slots
    for r in range(5):                                         # edge prefix of the rows with level <= r
        if r in level_last:
            running = (level_last[r] + 31) // 32 * 32
        assert lv["end"][r] == running
    assert lv["end"][4] == running
    pad = np.ones(len(lv["row"]), dtype=bool)
    pad[:ghost_slots] = False
    for i in range(N):
        pad[lv["row_ptr"][i]:lv["row_ptr"][i] + deg[i]] = False
    assert (lv["row"][pad] == -1).all()
    # (ii) same ligand output
    assert (lig - full).abs().max().item() < 2e-5
    with oracle_threads():
        o_l, _, _ = eo.dynamics_forward(sd, cfg, xl, xp, t, ml, mp, edges=edges)
    assert excess(lig, o_l) <= 0
    # (iii) the second half of the batch on its own
    half, _, _, _ = run(False, lo=B // 2)
    assert torch.equal(half, lig[B // 2 * nl:])


def test_full_atom_chains_with_identical_pockets_vs_oracle():
    """Free-running chains on a full-atom pocket repeated over the batch -- the configuration in which the
    chain hands the engine a pocket frame and the engine runs the forward cone on a canonical pocket
    (csrc/engine.hip): ConditionalDDPM.sample_given_pocket and .inpaint (every ligand atom known: bench.py's
    anchored states) against the oracle with the same noise tape, 4 steps, 1e-3 on coordinates (the tolerance
    of the other free-running tests), identical atom types."""
    from diffsbdd_amd.pocket import prepare_pocket
    arch, B, T = "crossdock_fullatom_cond", 3, 4
    cfg, dd = W.arch_cfg(arch)
    sd = W.random_state_dict(cfg, 0)
    z = np.load(os.path.join(GOLDEN_DIR, "pocket_3rfm.npz"))
    pocket = prepare_pocket(z["fa_x"], z["fa_types"], cfg["residue_nf"], repeats=B)
    n_lig = torch.full((B,), 14)
    om = do.OracleModel(sd, cfg, cfg["atom_nf"], cfg["residue_nf"], dd["timesteps"], dd["noise_schedule"],
                        dd["noise_precision"], norm_values=dd["norm_values"], conditional=True)
    model = _make_ddpm(arch, sd)
    model.cone_mode = 2          # three samples are below the rule's break-even (one group per >= 5 samples): pin the cone on
    # sample_given_pocket
    tape = do.NoiseTape(5)
    with oracle_threads():
        o_l, o_p, _, _ = do.cond_sample_given_pocket(om, {k: v.clone() for k, v in pocket.items()}, n_lig, tape, timesteps=T)
    model.set_noise_source(do.NoiseReplay(tape.draws))
    h_l, h_p, _, _ = model.sample_given_pocket({k: v.clone() for k, v in pocket.items()}, n_lig, timesteps=T)
    radius, ghost, _ = model.dynamics.engine().last_plan()
    assert radius == [1, 2, 3, 3, 2, 1] and ghost == [1, 1, 1, 0, 0, 0]          # the cone was on
    assert (h_l.cpu()[:, :3] - o_l[:, :3]).abs().max().item() < 1e-3
    assert torch.equal(h_l.cpu()[:, 3:].long(), o_l[:, 3:].long())
    assert (h_p.cpu() - o_p).abs().max().item() < 1e-3
    # inpaint with the 3rfm ligand as the known part (all atoms)
    g = torch.Generator().manual_seed(2)
    ligand = {"x": torch.from_numpy(z["ligand_x"]).float().repeat(B, 1),
              "one_hot": torch.nn.functional.one_hot(torch.randint(0, cfg["atom_nf"], (B * 14,), generator=g),
                                                     cfg["atom_nf"]).float(),
              "size": n_lig, "mask": torch.repeat_interleave(torch.arange(B), 14)}
    fixed = torch.ones(B * 14)
    tape = do.NoiseTape(6)
    with oracle_threads():
        o_l, o_p, _, _ = do.cond_inpaint(om, {k: v.clone() for k, v in ligand.items()},
                                         {k: v.clone() for k, v in pocket.items()}, fixed, tape, resamplings=1, timesteps=T)
    model.set_noise_source(do.NoiseReplay(tape.draws))
    h_l, h_p, _, _ = model.inpaint({k: v.clone() for k, v in ligand.items()}, {k: v.clone() for k, v in pocket.items()},
                                   fixed, resamplings=1, timesteps=T)
    assert model.dynamics.engine().last_plan()[0] == [1, 2, 3, 3, 2, 1]
    assert (h_l.cpu()[:, :3] - o_l[:, :3]).abs().max().item() < 1e-3
    assert torch.equal(h_l.cpu()[:, 3:].long(), o_l[:, 3:].long())
    assert (h_p.cpu() - o_p).abs().max().item() < 1e-3


def test_ligand_only_call_with_ragged_and_empty_samples():
    """Edge cases of the row elimination with a pocket frame: a batch of DIFFERENT full-atom pockets (286 / 201 / 150
    atoms; samples 0 and 3 share one: two samples, one representative), ligands of different sizes including a sample
    WITHOUT ligand atoms (all of its pocket rows are unreachable: level 4), and single-sample batches.
    Ligand eps against the oracle (1e-4) and the all-rows call (2e-5); levels against the BFS."""
    from diffsbdd_amd import _lib
    from diffsbdd_amd.engine import edge_capacity
    from diffsbdd_amd.pocket import prepare_pocket
    cfg, dd = W.arch_cfg("crossdock_fullatom_cond")
    sd = W.random_state_dict(cfg, 0)
    z = np.load(os.path.join(GOLDEN_DIR, "pocket_3rfm.npz"))
    d = dev()
    g = torch.Generator().manual_seed(11)
    P = torch.from_numpy(z["fa_x"]).float()
    types = torch.from_numpy(z["fa_types"]).long()
    center = P.mean(0)
    order = (P - center).norm(dim=1).argsort()            # pockets 2 and 3: the atoms closest to the centre
    sizes_p = [286, 201, 150, 286]
    sizes_l = [23, 9, 0, 14]
    xs, raws, hs, mp, ml, xl = [], [], [], [], [], []
    for b, (n_p, n_l) in enumerate(zip(sizes_p, sizes_l)):
        idx = order[:n_p].sort().values
        shift = torch.randn(3, generator=g) * 3
        xs.append(P[idx] - center + shift)
        raws.append(P[idx] - center)
        hs.append(torch.nn.functional.one_hot(types[idx], cfg["residue_nf"]).float() / dd["norm_values"][1])
        mp.append(torch.full((n_p,), b))
        ml.append(torch.full((n_l,), b))
        xl.append(shift + torch.randn(n_l, 3, generator=g) * 1.5)
    xp_all = torch.cat([torch.cat(xs), torch.cat(hs)], 1)
    mp_all, ml_all = torch.cat(mp), torch.cat(ml)
    xl_all = torch.cat([torch.cat(xl), torch.randn(len(ml_all), cfg["atom_nf"], generator=g) * 0.5], 1)
    B = len(sizes_p)
    t = torch.full((1,), 0.4)

    raw_all = torch.cat(raws)

    def run(sel, want_pocket, shared=False, rep=None, cone=2):
        keep_l = torch.isin(ml_all, torch.tensor(sel))
        keep_p = torch.isin(mp_all, torch.tensor(sel))
        remap = torch.full((B,), -1, dtype=torch.long)
        remap[torch.tensor(sel)] = torch.arange(len(sel))
        a_cpu = (xl_all[keep_l], xp_all[keep_p], t, remap[ml_all[keep_l]], remap[mp_all[keep_p]])
        a = [v.to(d) for v in a_cpu]
        m = make_dynamics(cfg, sd)
        eng = m.engine()
        cap = edge_capacity(a[3], a[4], len(sel))
        szs = torch.tensor([sizes_p[s] for s in sel])
        eng.set_option(_lib.OPT_CONE, cone)      # 2: on whatever the engine's cost model says (1 = by the cost model)
        if rep is None:      # every sample its own representative, frame = the current coordinates
            eng.set_pocket_frame(a[1][:, :3].contiguous(), a[4], szs.to(d), a[0].shape[0], len(sel), cap, shared)
        else:                # groups of identical pockets: the frame is the raw (untranslated) pocket
            eng.set_pocket_frame(raw_all[keep_p].to(d), a[4], szs.to(d), a[0].shape[0], len(sel), cap,
                                 representative=rep)
        outs = [m.forward_async(*a, batch=len(sel), edge_cap=cap, want_pocket=want_pocket) for _ in range(3)]
        torch.cuda.synchronize()
        assert all(int(o[2].item()) == 0 for o in outs) and torch.equal(outs[0][0], outs[2][0])
        n = a[0].shape[0] + a[1].shape[0]
        er, ec = eng.last_edges(n)
        lv = None if want_pocket else eng.last_levels(n)
        plan = eng.last_plan()
        eng.clear_pocket_frame()
        return outs[0][0], torch.stack([er, ec]), lv, plan, a_cpu

    full, edges, _, _, a_cpu = run([0, 1, 2, 3], True)
    lig, _, lv, plan, _ = run([0, 1, 2, 3], False)
    assert plan[0] == [1, 2, 3, 3, 2, 1] and plan[1] == [1, 1, 1, 0, 0, 0]   # every pocket its own representative
    # left to the engine's cost model the cone stays off here (as many ghost rows as pocket rows: the canonical network
    # would cost more than the skipped rows save); same ligand output to rounding
    auto, _, _, plan_a, _ = run([0, 1, 2, 3], False, cone=1)
    assert plan_a[0] == [4, 4, 4, 3, 2, 1] and not any(plan_a[1])
    assert (auto - lig).abs().max().item() < 2e-5
    n_l = len(ml_all)
    want = _hop_levels(edges[0].numpy(), edges[1].numpy(), n_l, n_l + len(mp_all))
    assert np.array_equal(lv["level"], want)
    rows_2 = n_l + sum(sizes_p[:2])
    assert (want[rows_2:rows_2 + sizes_p[2]] == 4).all()               # the sample without a ligand
    assert (lig - full).abs().max().item() < 2e-5
    with oracle_threads():
        o_l, _, _ = eo.dynamics_forward(sd, cfg, *a_cpu[:2], t, *a_cpu[3:], edges=edges)
    assert excess(lig, o_l) <= 0
    # samples 0 and 3 carry the same pocket: one representative for both (raw frame coordinates)
    grp, _, lvg, plang, _ = run([0, 1, 2, 3], False, rep=[0, 1, 2, 0])
    assert plang[0] == [1, 2, 3, 3, 2, 1] and lvg["ghost_nodes"] == 286 + 201 + 150 and lv["ghost_nodes"] == sum(sizes_p)
    assert (grp - lig).abs().max().item() < 2e-5
    # samples 0 and 3 on their own: the same bits as inside the mixed batch (a sample's result depends on its own
    # rows and on its representative's canonical pocket only)
    for s, lo, hi in ((0, 0, 23), (3, 32, 46)):
        one, e1, _, plan1, a1 = run([s], False)
        assert plan1[0] == [1, 2, 3, 3, 2, 1] and plan1[1] == [1, 1, 1, 0, 0, 0]
        assert torch.equal(one, lig[lo:hi])
        with oracle_threads():
            o1, _, _ = eo.dynamics_forward(sd, cfg, *a1[:2], t, *a1[3:], edges=e1)
        assert excess(one, o1) <= 0


def test_frame_survives_unrelated_calls_on_the_same_engine():
    """The frame's ghost rows live behind the real nodes in the engine's node arrays.  A call the frame does not apply
    to (here: a larger batch, eager and through its own replayed graph) writes over them; the next framed call --
    a graph replay -- must see them restored (dsbdd_dynamics_forward re-writes them from the pristine frame data)."""
    from diffsbdd_amd.engine import edge_capacity
    cfg, dd, xl, xp, t, ml, mp = bench_problem("crossdock_fullatom_cond", 4)
    sd = W.random_state_dict(cfg, 0)
    d = dev()
    nl, n0 = len(ml) // 4, len(mp) // 4
    m = make_dynamics(cfg, sd)
    eng = m.engine()
    big = [v.to(d) for v in (xl, xp, t[:1], ml, mp)]
    cap_big = edge_capacity(big[3], big[4], 4)
    small = [v.to(d) for v in (xl[:2 * nl], xp[:2 * n0], t[:1], ml[:2 * nl], mp[:2 * n0])]
    eng.ensure_workspace(len(ml), len(mp), 4, cap_big)          # one workspace for both problem sizes
    cap = edge_capacity(small[3], small[4], 2)
    from diffsbdd_amd import _lib
    eng.set_option(_lib.OPT_CONE, 2)                            # (two samples, two representatives: force the cone)
    eng.set_pocket_frame(small[1][:, :3].contiguous(), small[4], torch.full((2,), n0).to(d), 2 * nl, 2, cap, False)
    first = [m.forward_async(*small, batch=2, edge_cap=cap_big, want_pocket=False)[0].clone() for _ in range(3)]
    assert eng.last_plan()[0] == [1, 2, 3, 3, 2, 1]
    other = [m.forward_async(*big, batch=4, edge_cap=cap_big, want_pocket=False)[0].clone() for _ in range(3)]
    assert eng.last_plan()[0] == [4, 4, 4, 3, 2, 1]            # the frame does not apply to this size
    again = [m.forward_async(*small, batch=2, edge_cap=cap_big, want_pocket=False)[0].clone() for _ in range(2)]
    torch.cuda.synchronize()
    assert torch.equal(first[0], first[2]) and torch.equal(other[0], other[2])
    assert torch.equal(again[0], first[0]) and torch.equal(again[1], first[0])
    eng.clear_pocket_frame()


@pytest.mark.parametrize("B,n_lig,mode,n_steps,granule", [
    (64, 23, "inpaint", 3, "32"),      # bench.py's headline plan at its size
    (64, 23, "sample", 2, "32"),       # the secondary (free-running) leg's step at the same size
    (3, 14, "inpaint2", 4, "32"),      # small batch, resamplings = 2: the q(z_t | z_s) jump inside the fused kernel
    (64, 23, "inpaint", 2, "16"),      # the headline plan with every eligible stage on the 16-edge-granule kernels
    (3, 14, "inpaint2", 4, "16"),
    (64, 23, "inpaint", 2, "sk"),      # ... and with every stage (block 0's two-list launch included) on the split-K kernels
    (64, 23, "sample", 2, "sk"),
    (3, 14, "inpaint2", 4, "sk"),
])
def test_bench_plan_steps_teacher_forced_vs_oracle(B, n_lig, mode, n_steps, granule, monkeypatch):
    """The EXACT engine plan bench.py times, against the oracle at the size it is timed at: B identical 3rfm
    full-atom pockets (prepare_pocket(repeats=B) -> one representative: shared pocket frame + forward cone),
    the anchored ligand pose, ligand output only, one t for the batch -- message-stage radii [1,2,3,3,2,1] with the
    canonical pocket on ghost rows for the first three, and at B = 64 E > 512 * 128 (several tiles per workgroup).
    Teacher-forced iterations of ConditionalDDPM.inpaint's loop body (`_cond_step` + `dsbdd_cond_repaint_update`,
    conditional_model.py:432-464,600-660; mode 'sample': the reverse step of sample_given_pocket alone): the
    oracle's state goes into both sides every step (copied into the SAME device tensors, so the second call
    captures the hipGraph and the later ones replay it), the same injected noise, the device-built radius graph
    handed to the oracle; z_lig and the moved pocket within 1e-4 per step."""
    from diffsbdd_amd import synthetic
    if granule == "16":
        monkeypatch.setenv("DSBDD_GRANULE16", "0xFFFFFFFF")     # (block 0's two-list launch of a framed call stays on 32)
    if granule == "sk":
        monkeypatch.setenv("DSBDD_SPLITK", "0xFFFFFFFF")        # (csrc/edge_splitk.h)
    arch = "crossdock_fullatom_cond"
    cfg, dd = W.arch_cfg(arch)
    sd = W.random_state_dict(cfg, 0)
    T = dd["timesteps"]
    d = dev()
    dl = 3 + cfg["atom_nf"]
    model = _make_ddpm(arch, sd)
    model.cone_mode = 2          # the plan of the benchmark (B = 3 is below the per-chain rule's break-even of 5 samples per pocket)
    om = do.OracleModel(sd, cfg, cfg["atom_nf"], cfg["residue_nf"], T, dd["noise_schedule"],
                        dd["noise_precision"], norm_values=dd["norm_values"], conditional=True)
    # the benchmark's inputs (bench.py: load_pocket + anchor_ligand), once per side
    o_lig, o_poc = do.normalize(om, synthetic.anchor_ligand(B, n_lig, cfg["atom_nf"], "cpu"),
                                synthetic.load_pocket("fa", B, "cpu"))
    lm_c, pm_c = o_lig["mask"], o_poc["mask"]
    ligand, pocket = model.normalize(synthetic.anchor_ligand(B, n_lig, cfg["atom_nf"], d),
                                     synthetic.load_pocket("fa", B, d))
    N = len(lm_c) + len(pm_c)
    try:
        lm, pm = model._begin_chain(ligand["mask"], pocket["mask"], B, pocket=pocket)      # sets the shared frame
        assert model._framed
        fixed_c = torch.ones(B * n_lig)
        fixed_f = fixed_c.to(d)
        com0 = model._seg_mean3(pocket["x"], pm, B)
        com0_c = do._seg_mean(o_poc["x"], pm_c, B)
        xh0_lig = torch.cat([ligand["x"], ligand["one_hot"]], 1).contiguous()
        s0 = 120                                                       # alpha ~ 0.93: the pose sits inside the pocket
        g_t = om.g(torch.full((B, 1), float(s0 + 1)) / T)
        z_o, xp_o, _ = do.cond_noised_representation(
            om, torch.cat([o_lig["x"], o_lig["one_hot"]], 1), torch.cat([o_poc["x"], o_poc["one_hot"]], 1),
            lm_c, pm_c, g_t, do.NoiseTape(5), B)
        z_d, xp_d = z_o.to(d).contiguous(), xp_o.to(d).contiguous()
        zk_tmp = torch.empty_like(z_d)
        status = torch.zeros(1, dtype=torch.int32, device=d)
        co = model._coefs(T)
        eng = model.dynamics.engine()
        worst = -1.0
        for k in range(n_steps):
            s = s0 - k
            resample = mode == "inpaint2" and k % 2 == 0
            z_d.copy_(z_o); xp_d.copy_(xp_o)                           # teacher forcing, stable pointers
            pre = do.NoiseTape(40 + k)
            model.set_noise_source(do.NoiseReplay([pre((B * n_lig, dl)) for _ in range(3)]))
            if mode == "sample":
                model._cond_step(s, co, z_d, xp_d, lm, pm, B, status)
            else:
                model._inpaint_iteration(s, co, z_d, xp_d, zk_tmp, xh0_lig, com0, fixed_f, lm, pm, B, status, resample)
            torch.cuda.synchronize()
            assert int(status.item()) == 0
            radius, ghost, _ = eng.last_plan()
            assert radius == [1, 2, 3, 3, 2, 1] and ghost == [1, 1, 1, 0, 0, 0], (radius, ghost)
            er, ec = eng.last_edges(N)
            if B == 64:
                assert er.numel() > RESIDENT_TILES * 128, er.numel()
            om.edge_hook = lambda i, e=torch.stack([er, ec]): e
            with oracle_threads():
                if mode == "sample":
                    sa, ta = torch.full((B, 1), float(s)) / T, torch.full((B, 1), float(s + 1)) / T
                    z_o, xp_o = do.cond_sample_p_zs_given_zt(om, sa, ta, z_o, xp_o, lm_c, pm_c, do.NoiseTape(40 + k))
                else:
                    z_o, xp_o = do.cond_inpaint_iteration(om, s, T, z_o, xp_o, o_lig["x"], o_lig["one_hot"], com0_c,
                                                          fixed_c, lm_c, pm_c, do.NoiseTape(40 + k), resample=resample)
            e_l, e_p = excess(z_d, z_o), excess(xp_d, xp_o)
            print(f"[bench plan B={B} {mode}, step s={s}] E={er.numel()} excess over 1e-4: lig {e_l:.2e} pocket "
                  f"{e_p:.2e}; max |diff| lig {(z_d.cpu() - z_o).abs().max().item():.2e}")
            worst = max(worst, e_l, e_p)
        replays, captures, eager = eng.graph_stats()
        if n_steps >= 3:
            assert captures >= 1 and replays >= 1, (replays, captures, eager)    # the replayed graph was compared too
        assert worst <= 0, worst                                       # 1e-4 per timestep
    finally:
        model.set_noise_source(None)
        model._end_chain()


@pytest.mark.parametrize("arch,B,frame,granule", [
    ("small_cond", 6, "shared", "32"),               # H = 64: the chain kernel does not apply (three-launch path on both sides)
    ("small_variant", 6, False, "32"),               # H = 128, two sublayers (the next sublayer's P|Q rides the chain), E(3)
    ("crossdock_fullatom_cond", 8, "shared", "32"),  # H = 256: ghost rows in front of the level list, cone radii 1,2,3,3,2,1
    ("crossdock_fullatom_cond", 8, False, "32"),     # backward cone only
    ("crossdock_ca_cond", 8, False, "32"),
    ("small_variant", 6, False, "16"),               # the same calls with the edge stages on the 16-edge-granule kernels
    ("crossdock_fullatom_cond", 8, "shared", "16"),
    ("crossdock_ca_cond", 8, False, "16"),
    ("crossdock_fullatom_cond", 8, "shared", "sk"),  # ... and on the split-K kernels
    ("crossdock_ca_cond", 8, False, "sk"),
])
@pytest.mark.parametrize("want_pocket", [False, True])
def test_node_chain_kernel_vs_three_launches_and_oracle(arch, B, frame, granule, want_pocket, monkeypatch):
    """The row-owning node-phase kernel (csrc/node_chain.h: node MLP + the projections of the new h in one launch,
    16-row tiles dealt out by cost) against the three-launch node phase (csrc/node_linear.h) on the same call --
    different MFMA shapes, i.e. a different summation order inside every 16-k group: 2e-5 -- and against the oracle
    (1e-4).  DSBDD_NODE_CHAIN_MIN_ROWS=0 forces the kernel onto problems far below its default row threshold, so that
    every row-tile count (1 .. 6), ragged ends, ghost-row offsets and the projection-only launches are exercised."""
    from diffsbdd_amd.engine import edge_capacity
    if granule == "16":
        monkeypatch.setenv("DSBDD_GRANULE16", "0xFFFFFFFF")
    if granule == "sk":
        monkeypatch.setenv("DSBDD_SPLITK", "0xFFFFFFFF")
    cfg, dd, xl, xp, t, ml, mp = bench_problem(arch, B)
    sd = W.random_state_dict(cfg, 0)
    d = dev()
    n0 = len(mp) // B
    sizes = torch.full((B,), n0)
    shared = frame == "shared"
    if shared:
        raw = xp[:n0, :3].repeat(B, 1)
        delta = (xp[:, :3] - raw).view(B, n0, 3)[:, 0]
        xp = torch.cat([raw + delta[mp], xp[:, 3:]], 1)
    else:
        raw = xp[:, :3]

    def run(chain):
        if chain:
            monkeypatch.setenv("DSBDD_NODE_CHAIN_MIN_ROWS", "0")
            monkeypatch.delenv("DSBDD_NODE_CHAIN", raising=False)
        else:
            monkeypatch.setenv("DSBDD_NODE_CHAIN", "0")
        m = make_dynamics(cfg, sd)
        eng = m.engine()
        a = [v.to(d) for v in (xl, xp, t[:1] if shared else t, ml, mp)]
        cap = edge_capacity(a[3], a[4], B)
        if frame:
            eng.set_pocket_frame(raw.to(d), a[4], sizes.to(d), a[0].shape[0], B, cap, shared)
        outs = [m.forward_async(*a, batch=B, edge_cap=cap, want_pocket=want_pocket) for _ in range(3)]
        torch.cuda.synchronize()
        assert all(int(o[2].item()) == 0 for o in outs)
        assert torch.equal(outs[0][0], outs[2][0])                       # eager = replayed graph
        er, ec = eng.last_edges(a[0].shape[0] + a[1].shape[0])
        if frame:
            eng.clear_pocket_frame()
        return outs[0][0], outs[0][1], torch.stack([er, ec])

    c_l, c_p, edges = run(True)
    o_l, o_p, _ = run(False)
    assert (c_l - o_l).abs().max().item() < 2e-5
    if want_pocket:
        assert (c_p - o_p).abs().max().item() < 2e-5
    with oracle_threads():
        r_l, r_p, _ = eo.dynamics_forward(sd, cfg, xl, xp, t, ml, mp, edges=edges)
    assert excess(c_l, r_l) <= 0
    if want_pocket:
        assert excess(c_p, r_p) <= 0
