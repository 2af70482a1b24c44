"""GPU parity at the BASELINE.json batch sizes, part 2 (-m gpu): the cases the round-5 review found missing at the
sizes bench.py times --

  (a) the reverse step at the ENDS of the schedule (s = T - 1, 1, 0: the first step out of z_T and the last two before
      the decode, where alpha_{t|s}, sigma_{t|s} and c_eps take their extreme values) and the final decode
      `sample_p_xh_given_z0` (conditional_model.py:112-135, en_diffusion.py:263-288: argmax one-hots + x) at B = 64 / 32;
  (b) the heterogeneous batches of bench.py's `--pockets mixed` / `grouped` legs (synthetic.mixed_pockets: 3rfm / 5ndu
      alternating, each under its own rotation; 40 copies of one pocket + 24 singletons) at B = 64, through the chain
      entry the legs use (`_begin_chain` with the pocket dict: pocket frame, groups, the chain-pinned cone mode);
  (c) one iteration of the JOINT model's RePaint loop including the jump back q(z_t | z_s) between resamplings
      (en_diffusion.py:742-809) at B = 64.

Same protocol as tests/test_gpu_fullsize.py: the oracle's state goes into both sides (teacher forcing), the same
injected noise, the device-built radius graph handed to the oracle, 1e-4 absolute (+ 1e-5 relative on large states)
per timestep.
"""
import numpy as np
import pytest
import torch

from oracle import ddpm_oracle as do
from oracle import weights as W
from tests.test_gpu_fullsize import (RESIDENT_TILES, _make_ddpm, bench_problem, dev, excess, oracle_threads)

pytestmark = pytest.mark.gpu


def _oracle_model(arch, sd):
    cfg, dd = W.arch_cfg(arch)
    return do.OracleModel(sd, cfg, cfg["atom_nf"], cfg["residue_nf"], dd["timesteps"], dd["noise_schedule"],
                          dd["noise_precision"], norm_values=dd["norm_values"], conditional=dd["conditional"])


def _shapes(cfg, dd, ml, mp):
    if dd["conditional"]:
        return [(len(ml), 3 + cfg["atom_nf"])]
    return [(len(ml) + len(mp), 3), (len(ml), cfg["atom_nf"]), (len(mp), cfg["residue_nf"])]


@pytest.mark.parametrize("arch,B", [("crossdock_fullatom_cond", 64), ("moad_fullatom_joint", 64), ("crossdock_ca_cond", 32)])
def test_reverse_steps_at_the_schedule_ends_and_final_decode(arch, B):
    """sample_p_zs_given_zt at s = T - 1, 1, 0 and sample_p_xh_given_z0 behind it, teacher-forced at the benchmark batch."""
    cfg, dd, xl, xp, _, ml, mp = bench_problem(arch, B, seed=2)
    sd = W.random_state_dict(cfg, 0)
    N = len(ml) + len(mp)
    model = _make_ddpm(arch, sd)
    om = _oracle_model(arch, sd)
    d = dev()
    T = dd["timesteps"]
    eng = model.dynamics.engine()
    z_l, z_p = xl, xp
    worst = -1.0
    for k, s_int in enumerate((T - 1, 1, 0)):
        s = torch.full((B, 1), float(s_int)) / T
        t = torch.full((B, 1), float(s_int + 1)) / T
        pre = do.NoiseTape(91 + k)
        model.set_noise_source(do.NoiseReplay([pre(sh) for sh in _shapes(cfg, dd, ml, mp)]))
        h_l, h_p = model.sample_p_zs_given_zt(s.to(d), t.to(d), z_l.to(d), z_p.to(d), ml.to(d), mp.to(d))
        er, ec = eng.last_edges(N)
        om.edge_hook = lambda i, e=torch.stack([er, ec]): e
        with oracle_threads():
            fn = do.cond_sample_p_zs_given_zt if dd["conditional"] else do.joint_sample_p_zs_given_zt
            o_l, o_p = fn(om, s, t, z_l, z_p, ml, mp, do.NoiseTape(91 + k))
        e_l, e_p = excess(h_l, o_l), excess(h_p, o_p)
        print(f"[{arch} B={B}] step s={s_int}: excess over 1e-4 lig {e_l:.2e} pocket {e_p:.2e}")
        worst = max(worst, e_l, e_p)
        z_l, z_p = o_l, o_p
    assert worst <= 0, worst                                            # 1e-4 per timestep
    # final decode on the oracle's z_0
    pre = do.NoiseTape(95)
    model.set_noise_source(do.NoiseReplay([pre(sh) for sh in _shapes(cfg, dd, ml, mp)]))
    x_l, h_l, x_p, h_p = model.sample_p_xh_given_z0(z_l.to(d), z_p.to(d), ml.to(d), mp.to(d), B)
    er, ec = eng.last_edges(N)
    om.edge_hook = lambda i, e=torch.stack([er, ec]): e
    with oracle_threads():
        fn = do.cond_sample_p_xh_given_z0 if dd["conditional"] else do.joint_sample_p_xh_given_z0
        ox_l, oh_l, ox_p, oh_p = fn(om, z_l, z_p, ml, mp, B, do.NoiseTape(95))
    model.set_noise_source(None)
    assert excess(x_l, ox_l) <= 0 and excess(x_p, ox_p) <= 0, (excess(x_l, ox_l), excess(x_p, ox_p))   # 1e-4 on un-normalised x
    assert torch.equal(h_l.cpu().long(), oh_l.long())                   # identical atom types
    if not dd["conditional"]:
        assert torch.equal(h_p.cpu().long(), oh_p.long())
    else:       # the pocket features come back un-normalised (conditional_model.py:128-130)
        assert excess(h_p, oh_p) <= 0


@pytest.mark.parametrize("n_same", [0, 40])
def test_heterogeneous_pocket_batches_teacher_forced_vs_oracle(n_same):
    """bench.py's `--pockets mixed` (n_same = 0: 64 distinct pockets) and `grouped` (40 copies of one pocket + 24
    singletons) batches at B = 64: the frame / group / cone decisions `_begin_chain` takes for them (every pocket its own
    representative resp. 25 groups; the per-chain rule leaves the forward cone off: radii [4,4,4,3,2,1]), then one
    teacher-forced iteration of the anchored loop body and one free reverse step against the oracle."""
    from diffsbdd_amd import synthetic
    arch, B, n_lig = "crossdock_fullatom_cond", 64, 23
    cfg, dd = W.arch_cfg(arch)
    sd = W.random_state_dict(cfg, 0)
    T = dd["timesteps"]
    d = dev()
    dl = 3 + cfg["atom_nf"]
    model = _make_ddpm(arch, sd)
    om = _oracle_model(arch, sd)
    p_c, l_c = synthetic.mixed_pockets("fa", B, n_lig, cfg["atom_nf"], "cpu", n_same=n_same)
    p_d, l_d = synthetic.mixed_pockets("fa", B, n_lig, cfg["atom_nf"], d, n_same=n_same)
    o_lig, o_poc = do.normalize(om, l_c, p_c)
    lm_c, pm_c = o_lig["mask"], o_poc["mask"]
    ligand, pocket = model.normalize(l_d, p_d)
    N = len(lm_c) + len(pm_c)
    try:
        lm, pm = model._begin_chain(ligand["mask"], pocket["mask"], B, pocket=pocket)
        assert model._framed
        fixed_c = torch.ones(B * n_lig)
        fixed_f = fixed_c.to(d)
        com0 = model._seg_mean3(pocket["x"], pm, B)
        com0_c = do._seg_mean(o_poc["x"], pm_c, B)
        xh0_lig = torch.cat([ligand["x"], ligand["one_hot"]], 1).contiguous()
        s0 = 120
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
        for k, mode in enumerate(("inpaint", "sample")):
            s = s0 - k
            z_d.copy_(z_o); xp_d.copy_(xp_o)
            pre = do.NoiseTape(60 + k)
            model.set_noise_source(do.NoiseReplay([pre((B * n_lig, dl)) for _ in range(3)]))
            if mode == "sample":
                model._cond_step(s, co, z_d, xp_d, lm, pm, B, status)
            else:
                model._inpaint_iteration(s, co, z_d, xp_d, zk_tmp, xh0_lig, com0, fixed_f, lm, pm, B, status, False)
            torch.cuda.synchronize()
            assert int(status.item()) == 0
            radius, ghost, _ = eng.last_plan()
            assert radius == [4, 4, 4, 3, 2, 1] and not any(ghost), (radius, ghost)     # cone off: > 12 distinct pockets
            er, ec = eng.last_edges(N)
            assert er.numel() > RESIDENT_TILES * 128
            om.edge_hook = lambda i, e=torch.stack([er, ec]): e
            with oracle_threads():
                if mode == "sample":
                    sa, ta = torch.full((B, 1), float(s)) / T, torch.full((B, 1), float(s + 1)) / T
                    z_o, xp_o = do.cond_sample_p_zs_given_zt(om, sa, ta, z_o, xp_o, lm_c, pm_c, do.NoiseTape(60 + k))
                else:
                    z_o, xp_o = do.cond_inpaint_iteration(om, s, T, z_o, xp_o, o_lig["x"], o_lig["one_hot"], com0_c,
                                                          fixed_c, lm_c, pm_c, do.NoiseTape(60 + k), resample=False)
            e_l, e_p = excess(z_d, z_o), excess(xp_d, xp_o)
            print(f"[mixed pockets n_same={n_same} {mode}, s={s}] E={er.numel()} excess over 1e-4: lig {e_l:.2e} pocket {e_p:.2e}")
            worst = max(worst, e_l, e_p)
        assert worst <= 0, worst
    finally:
        model.set_noise_source(None)
        model._end_chain()


def test_joint_repaint_iteration_with_jump_teacher_forced_vs_oracle():
    """EnVariationalDiffusion.inpaint's loop body (en_diffusion.py:742-809) at the size bench.py's joint leg runs it
    (moad_fullatom_joint x 64, ligand unknown / pocket fixed): iteration 0 ends a resampling segment and jumps back
    with q(z_t | z_s) inside the fused kernel, iteration 1 is a plain one; both against the oracle's
    `joint_inpaint_iteration` (which `joint_inpaint` -- pinned by the reference goldens -- is built from)."""
    arch, B = "moad_fullatom_joint", 64
    cfg, dd, xl, xp, _, ml, mp = bench_problem(arch, B, seed=3)
    sd = W.random_state_dict(cfg, 0)
    T = dd["timesteps"]
    d = dev()
    N = len(ml) + len(mp)
    model = _make_ddpm(arch, sd)
    om = _oracle_model(arch, sd)
    g = torch.Generator().manual_seed(9)
    # the known part: the (normalised, centred) input the chain holds on to -- here the benchmark state itself
    xh0_l, xh0_p = xl.clone(), xp.clone()
    lfix_c, pfix_c = torch.zeros(len(ml)), torch.ones(len(mp))
    z_o_l = xl + 0.2 * torch.randn(xl.shape, generator=g)
    z_o_p = xp + 0.2 * torch.randn(xp.shape, generator=g)
    try:
        lm, pm = model._begin_chain(ml, mp, B)
        z_l, z_p = z_o_l.to(d).contiguous(), z_o_p.to(d).contiguous()
        zk_l, zk_p = torch.empty_like(z_l), torch.empty_like(z_p)
        x0l, x0p = xh0_l.to(d).contiguous(), xh0_p.to(d).contiguous()
        lfix, pfix = lfix_c.to(d), pfix_c.to(d)
        status = torch.zeros(1, dtype=torch.int32, device=d)
        co = model._coefs(T)
        eng = model.dynamics.engine()
        worst = -1.0
        s = 200
        for k, jump in enumerate((True, False)):
            z_l.copy_(z_o_l); z_p.copy_(z_o_p)
            pre = do.NoiseTape(30 + k)
            draws = []
            for _ in range(3 if jump else 2):                          # known part, reverse step, (jump)
                draws += [pre(sh) for sh in _shapes(cfg, dd, ml, mp)]
            model.set_noise_source(do.NoiseReplay(draws))
            model._joint_inpaint_iteration(s, co, z_l, z_p, zk_l, zk_p, x0l, x0p, lfix, pfix, lm, pm, B, status,
                                           jump_to=s + 1 if jump else None)
            torch.cuda.synchronize()
            assert int(status.item()) == 0
            er, ec = eng.last_edges(N)
            if k == 0:
                assert er.numel() > RESIDENT_TILES * 128 // 2, er.numel()
            om.edge_hook = lambda i, e=torch.stack([er, ec]): e
            with oracle_threads():
                z_o_l, z_o_p = do.joint_inpaint_iteration(om, s, T, z_o_l, z_o_p, xh0_l, xh0_p, lfix_c, pfix_c, ml, mp,
                                                          do.NoiseTape(30 + k), jump_to=s + 1 if jump else None)
            e_l, e_p = excess(z_l, z_o_l), excess(z_p, z_o_p)
            print(f"[joint RePaint B={B}, s={s}, jump={jump}] E={er.numel()} excess over 1e-4: lig {e_l:.2e} pocket {e_p:.2e}")
            worst = max(worst, e_l, e_p)
            s = s if jump else s - 1
        assert worst <= 0, worst
    finally:
        model.set_noise_source(None)
        model._end_chain()


def test_splitk_auto_chain_on_the_calpha_config_vs_oracle_chain():
    """`ddpm.edge_splitk = "auto"` (EnVariationalDiffusion.splitk_auto) through the public sampling call: on
    crossdock_ca_cond every stage goes to the split-K kernels (csrc/edge_splitk.h), the mask reaches the engine as a
    chain constant, and a short free-running `sample_given_pocket` equals the oracle's chain on the same noise tape
    (1e-3 on coordinates, identical atom types: the tolerance of the other free-running chains); the same chain with the
    default kernels agrees with it to rounding, and with `edge_splitk = None` the engine's mask is back to 0."""
    from diffsbdd_amd import _lib, synthetic
    arch, B, T = "crossdock_ca_cond", 8, 4
    cfg, dd = W.arch_cfg(arch)
    sd = W.random_state_dict(cfg, 0)
    pocket = synthetic.load_pocket("ca", B, "cpu")
    n_lig = torch.full((B,), 23)
    om = _oracle_model(arch, sd)
    tape = do.NoiseTape(12)
    with oracle_threads():
        o_l, o_p, _, _ = do.cond_sample_given_pocket(om, {k: v.clone() for k, v in pocket.items()}, n_lig, tape, timesteps=T)
    outs = {}
    for mode in ("auto", None):
        model = _make_ddpm(arch, sd)
        model.edge_splitk = mode
        model.set_noise_source(do.NoiseReplay(tape.draws))
        h_l, h_p, _, _ = model.sample_given_pocket({k: v.clone() for k, v in pocket.items()}, n_lig, timesteps=T)
        mask = model.dynamics.engine().get_option(_lib.OPT_SPLITK) & 0xFFFFFFFF
        assert mask == (0x003F003F if mode == "auto" else 0), hex(mask)
        assert (h_l.cpu()[:, :3] - o_l[:, :3]).abs().max().item() < 1e-3
        assert torch.equal(h_l.cpu()[:, 3:].long(), o_l[:, 3:].long())
        outs[mode] = h_l.cpu()
    assert (outs["auto"] - outs[None]).abs().max().item() < 1e-4
    import os
    if os.environ.get("DSBDD_EMU", "0") in ("", "0"):                     # (the emulated path ignores the mask by design)
        assert not torch.equal(outs["auto"], outs[None])                  # the variant really ran
