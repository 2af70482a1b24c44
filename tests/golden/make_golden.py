#!/usr/bin/env python
"""Generate the committed golden vectors from the REAL reference.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

It imports the reference's `equivariant_diffusion` modules through
oracle/ref_shim.py (third-party stubs only), loads the seeded
synthetic weights of oracle/weights.py into the reference's own modules, runs
them on CPU (fp32) and stores inputs/outputs as small .npz files next to this
script.  /root/reference does not exist on the GPU box; the tests read only the
.npz files.

What is stored
  pocket_<id>.npz     pocket fixtures derived from example/<id>.pdb + ligand sdf
  schedule.npz        gamma tables and RePaint schedules (known answers)
  dyn_<case>.npz      one EGNNDynamics.forward call: inputs, the reference's edge
                      list, eps outputs (+ per-block h/x for the small archs)
  ddpm_<case>.npz     sampling-loop traces with injected noise: per reverse step
                      (z_t, pocket_t, z_s, pocket_s), the noise tape, final output
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle.ref_shim import import_reference, REF_ROOT  # noqa: E402
from oracle import weights as W  # noqa: E402
from oracle.ddpm_oracle import NoiseTape  # noqa: E402
import importlib.util  # noqa: E402

_spec = importlib.util.spec_from_file_location(
    "_pocket", os.path.join(ROOT, "diffsbdd_amd", "pocket.py"))
pocket_mod = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(pocket_mod)

dyn_mod, en_mod, cond_mod, egnn_mod = import_reference()
torch.set_num_threads(8)


def save(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = v
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"  wrote {name}.npz  ({os.path.getsize(path) / 1024:.1f} KiB)")


# ---------------------------------------------------------------------------
def make_pockets():
    fixtures = {}
    for pid, sdf in (("3rfm", "3rfm_B_CFF.sdf"), ("5ndu", "5ndu_C_8V2.sdf")):
        res = pocket_mod.read_pdb_residues(os.path.join(REF_ROOT, "example", pid + ".pdb"))
        lig = pocket_mod.read_sdf_coords(os.path.join(REF_ROOT, "example", sdf))
        sel = pocket_mod.pocket_residues_from_ligand(res, lig, 8.0)
        ca_x, ca_t, _ = pocket_mod.featurize_pocket(sel, "CA")
        fa_x, fa_t, _ = pocket_mod.featurize_pocket(sel, "full-atom")
        save("pocket_" + pid, ca_x=ca_x, ca_types=ca_t, fa_x=fa_x, fa_types=fa_t,
             ligand_x=lig, n_residues=np.int64(len(sel)))
        fixtures[pid] = dict(ca=(ca_x, ca_t, 20), fa=(fa_x, fa_t, 10))
    return fixtures


def make_schedule():
    out = {}
    for tag, (sched, T, prec) in {
        "poly2_T500_p5e-4": ("polynomial_2", 500, 5e-4),
        "poly2_T500_p1e-5": ("polynomial_2", 500, 1e-5),
        "poly2_T20_p5e-4": ("polynomial_2", 20, 5e-4),
        "cosine_T20_p1e-4": ("cosine", 20, 1e-4),
        "cosine_T1000_p1e-4": ("cosine", 1000, 1e-4),
    }.items():
        g = en_mod.PredefinedNoiseSchedule(sched, timesteps=T, precision=prec)
        out["gamma_" + tag] = g.gamma.detach().numpy()
    dummy = en_mod.EnVariationalDiffusion.__new__(en_mod.EnVariationalDiffusion)
    reps = []
    for (r, j, T) in [(1, 1, 500), (2, 1, 500), (10, 1, 500), (10, 10, 500), (3, 2, 7),
                      (2, 5, 20), (1, 1, 1), (4, 3, 10)]:
        s = en_mod.EnVariationalDiffusion.get_repaint_schedule(dummy, r, j, T)
        reps.append(dict(resamplings=r, jump_length=j, timesteps=T, schedule=s))
    out["repaint_json"] = np.array(json.dumps(reps))
    save("schedule", **out)


# ---------------------------------------------------------------------------
def build_ref_dynamics(cfg, sd):
    kw = {k: v for k, v in cfg.items()}
    d = dyn_mod.EGNNDynamics(**kw).eval()
    d.load_state_dict(sd)
    return d


def synth_inputs(arch, cfg, ddpm, pockets, B, n_lig, seed, t_val=None, hetero=False):
    """A plausible mid-chain state: pocket = fixture (normalised, centred on the
    ligand COM as the conditional sampler keeps it), ligand = Gaussian blob."""
    g = torch.Generator().manual_seed(seed)
    a, r = cfg["atom_nf"], cfg["residue_nf"]
    nv = ddpm["norm_values"]
    xs, hs, mp = [], [], []
    for b in range(B):
        pid = "3rfm" if (not hetero or b % 2 == 0) else "5ndu"
        key = "ca" if r == 20 else "fa"
        px, pt, _ = pockets[pid][key]
        px = torch.from_numpy(px).clone()
        if hetero:  # random rigid rotation per sample
            q, _ = torch.linalg.qr(torch.randn(3, 3, generator=g))
            px = px @ q.T
        px = (px - px.mean(0, keepdim=True)) / nv[0]
        xs.append(px)
        hs.append(torch.nn.functional.one_hot(torch.from_numpy(pt), r).float() / nv[1])
        mp.append(torch.full((len(px),), b, dtype=torch.int64))
    x_p, h_p, mask_p = torch.cat(xs), torch.cat(hs), torch.cat(mp)
    if isinstance(n_lig, int):
        n_lig = [n_lig] * B
    mask_l = torch.repeat_interleave(torch.arange(B), torch.tensor(n_lig))
    spread = 2.5 / nv[0]
    x_l = torch.randn(len(mask_l), 3, generator=g) * spread
    # conditional sampler invariant: ligand COM = 0, pocket shifted accordingly
    if ddpm["conditional"]:
        for b in range(B):
            sel = mask_l == b
            com = x_l[sel].mean(0, keepdim=True)
            x_l[sel] -= com
    else:
        # joint: COM of ligand+pocket = 0 per sample
        for b in range(B):
            sl, sp = mask_l == b, mask_p == b
            com = torch.cat((x_l[sl], x_p[sp])).mean(0, keepdim=True)
            x_l[sl] -= com
            x_p[sp] -= com
        h_p = h_p + 0.3 * torch.randn(h_p.shape, generator=g)
        x_p = x_p + 0.05 * torch.randn(x_p.shape, generator=g)
    h_l = torch.randn(len(mask_l), a, generator=g) * 0.7
    if t_val is None:
        t = torch.rand(B, 1, generator=g)
    else:
        t = torch.full((B, 1), float(t_val))
    return (torch.cat([x_l, h_l], 1).float(), torch.cat([x_p, h_p], 1).float(), t,
            mask_l, mask_p)


def make_dyn_case(name, arch, pockets, B, n_lig, seed, store_weights=False, hetero=False,
                  t_val=None, store_trace=False):
    print(f"[dyn] {name}")
    cfg, ddpm = W.arch_cfg(arch)
    sd = W.random_state_dict(cfg, seed=seed)
    d = build_ref_dynamics(cfg, sd)
    xh_l, xh_p, t, ml, mp = synth_inputs(arch, cfg, ddpm, pockets, B, n_lig, seed + 100,
                                         t_val=t_val, hetero=hetero)
    trace = []
    hooks = []
    for i in range(cfg["n_layers"]):
        blk = d.egnn._modules[f"e_block_{i}"]
        hooks.append(blk.register_forward_hook(
            lambda m, inp, out: trace.append((out[0].detach().clone(), out[1].detach().clone()))))
    with torch.no_grad():
        eps_l, eps_p = d(xh_l, xh_p, t, ml, mp)
        edges = d.get_edges(ml, mp, xh_l[:, :3], xh_p[:, :3])
    for h in hooks:
        h.remove()
    arrs = dict(cfg_json=np.array(json.dumps(cfg)), ddpm_json=np.array(json.dumps(ddpm)),
                arch=np.array(arch), seed=np.int64(seed),
                checksum=np.array(W.state_dict_checksum(sd)),
                xh_lig=xh_l, xh_pocket=xh_p, t=t, mask_lig=ml, mask_pocket=mp,
                edges=edges.to(torch.int32), eps_lig=eps_l, eps_pocket=eps_p)
    if store_weights:
        for k, v in sd.items():
            arrs["w:" + k] = v
    if store_trace:
        for i, (h, x) in enumerate(trace):
            arrs[f"trace_h_{i}"] = h
            arrs[f"trace_x_{i}"] = x
    else:
        # keep the x trajectory (tiny) for the full-size archs too
        for i, (h, x) in enumerate(trace):
            arrs[f"trace_x_{i}"] = x
    save(name, **arrs)
    print(f"    N_l={len(ml)} N_p={len(mp)} E={edges.shape[1]} |eps_x|max={eps_l[:, :3].abs().max():.3f}"
          f" |eps_h|max={eps_l[:, 3:].abs().max():.3f}")


# ---------------------------------------------------------------------------
class patched_randn:
    """Route every torch.randn call of the reference through a NoiseTape."""

    def __init__(self, tape):
        self.tape = tape

    def __enter__(self):
        self.orig = torch.randn
        tape, orig = self.tape, self.orig

        def fake(*size, device=None, **kw):
            if "generator" in kw:  # the tape's own draw
                return orig(*size, **kw)
            if len(size) == 1 and not isinstance(size[0], int):
                size = tuple(size[0])
            return tape(size)
        torch.randn = fake
        return self

    def __exit__(self, *a):
        torch.randn = self.orig


def build_ref_ddpm(arch, seed):
    cfg, dd = W.arch_cfg(arch)
    sd = W.random_state_dict(cfg, seed=seed)
    d = build_ref_dynamics(cfg, sd)
    cls = cond_mod.ConditionalDDPM if dd["conditional"] else en_mod.EnVariationalDiffusion
    import io
    import contextlib
    with contextlib.redirect_stdout(io.StringIO()):
        model = cls(dynamics=d, atom_nf=cfg["atom_nf"], residue_nf=cfg["residue_nf"], n_dims=3,
                    size_histogram=np.ones((12, 60)), timesteps=dd["timesteps"],
                    noise_schedule=dd["noise_schedule"], noise_precision=dd["noise_precision"],
                    loss_type="l2", norm_values=dd["norm_values"]).eval()
    return cfg, dd, sd, model


def small_pocket(pockets, key, B, n_keep, rotate_seed=None):
    """A truncated 3rfm pocket (first n_keep nodes) repeated B times, in Angstrom."""
    px, pt, nt = pockets["3rfm"][key]
    px, pt = px[:n_keep], pt[:n_keep]
    return pocket_mod.prepare_pocket(px, pt, nt, repeats=B)


def wrap_trace(model, trace):
    orig = model.sample_p_zs_given_zt

    def wrapped(s, t, z_l, z_p, lm, pm, fix_noise=False):
        out = orig(s, t, z_l, z_p, lm, pm, fix_noise)
        trace.append((s.clone(), t.clone(), z_l.clone(), z_p.clone(), out[0].clone(), out[1].clone()))
        return out
    model.sample_p_zs_given_zt = wrapped


def make_ddpm_cond(name, arch, pockets, seed, B=3, n_keep=40, timesteps=None):
    print(f"[ddpm] {name}")
    cfg, dd, sd, model = build_ref_ddpm(arch, seed)
    key = "ca" if cfg["residue_nf"] == 20 else "fa"
    pocket = small_pocket(pockets, key, B, n_keep)
    pocket_in = {k: v.clone() for k, v in pocket.items()}
    n_lig = torch.tensor([5, 8, 6][:B])
    tape = NoiseTape(seed + 7)
    trace = []
    wrap_trace(model, trace)
    with patched_randn(tape), torch.no_grad():
        out_l, out_p, lm, pm = model.sample_given_pocket(pocket, n_lig, timesteps=timesteps)
    arrs = dict(cfg_json=np.array(json.dumps(cfg)), ddpm_json=np.array(json.dumps(dd)),
                arch=np.array(arch), seed=np.int64(seed),
                checksum=np.array(W.state_dict_checksum(sd)),
                pocket_x=pocket_in["x"], pocket_one_hot=pocket_in["one_hot"],
                pocket_size=pocket_in["size"], pocket_mask=pocket_in["mask"],
                num_nodes_lig=n_lig, out_lig=out_l, out_pocket=out_p, lig_mask=lm,
                n_draws=np.int64(len(tape.draws)), n_steps=np.int64(len(trace)),
                timesteps=np.int64(model.T if timesteps is None else timesteps))
    for i, dr in enumerate(tape.draws):
        arrs[f"noise_{i}"] = dr
    for i, (s, t, zt, pt, zs, ps) in enumerate(trace):
        arrs[f"step{i}_s"] = s
        arrs[f"step{i}_t"] = t
        arrs[f"step{i}_zt"] = zt
        arrs[f"step{i}_pt"] = pt
        arrs[f"step{i}_zs"] = zs
        arrs[f"step{i}_ps"] = ps
    save(name, **arrs)


def make_ddpm_cond_inpaint(name, arch, pockets, seed, B=2, n_keep=30, timesteps=6,
                           resamplings=2):
    print(f"[ddpm] {name}")
    cfg, dd, sd, model = build_ref_ddpm(arch, seed)
    key = "ca" if cfg["residue_nf"] == 20 else "fa"
    pocket = small_pocket(pockets, key, B, n_keep)
    g = torch.Generator().manual_seed(seed + 3)
    n_lig = torch.tensor([6, 7][:B])
    lm = torch.repeat_interleave(torch.arange(B), n_lig)
    com = torch.stack([pocket["x"][pocket["mask"] == b].mean(0) for b in range(B)])
    ligand = {"x": com[lm] + torch.randn(len(lm), 3, generator=g) * 1.5,
              "one_hot": torch.nn.functional.one_hot(
                  torch.randint(0, cfg["atom_nf"], (len(lm),), generator=g), cfg["atom_nf"]),
              "size": n_lig, "mask": lm}
    lig_fixed = torch.zeros(len(lm))
    lig_fixed[[0, 1, 2, 6, 7]] = 1
    pocket_in = {k: v.clone() for k, v in pocket.items()}
    ligand_in = {k: v.clone() for k, v in ligand.items()}
    tape = NoiseTape(seed + 9)
    with patched_randn(tape), torch.no_grad():
        out_l, out_p, _, _ = model.inpaint(ligand, pocket, lig_fixed, resamplings=resamplings,
                                           timesteps=timesteps)
    # diversify on the same inputs
    pocket2 = {k: v.clone() for k, v in pocket_in.items()}
    ligand2 = {k: v.clone() for k, v in ligand_in.items()}
    tape2 = NoiseTape(seed + 11)
    with patched_randn(tape2), torch.no_grad():
        div_l, div_p, _, _ = model.diversify(ligand2, pocket2, noising_steps=4)
    arrs = dict(cfg_json=np.array(json.dumps(cfg)), ddpm_json=np.array(json.dumps(dd)),
                arch=np.array(arch), seed=np.int64(seed),
                checksum=np.array(W.state_dict_checksum(sd)),
                pocket_x=pocket_in["x"], pocket_one_hot=pocket_in["one_hot"],
                pocket_size=pocket_in["size"], pocket_mask=pocket_in["mask"],
                ligand_x=ligand_in["x"], ligand_one_hot=ligand_in["one_hot"],
                ligand_size=ligand_in["size"], ligand_mask=ligand_in["mask"],
                lig_fixed=lig_fixed, resamplings=np.int64(resamplings),
                timesteps=np.int64(timesteps), out_lig=out_l, out_pocket=out_p,
                n_draws=np.int64(len(tape.draws)),
                div_lig=div_l, div_pocket=div_p, div_steps=np.int64(4),
                n_draws_div=np.int64(len(tape2.draws)))
    for i, dr in enumerate(tape.draws):
        arrs[f"noise_{i}"] = dr
    for i, dr in enumerate(tape2.draws):
        arrs[f"divnoise_{i}"] = dr
    save(name, **arrs)


def make_ddpm_joint(name, arch, pockets, seed, B=2, timesteps=5):
    print(f"[ddpm] {name}")
    cfg, dd, sd, model = build_ref_ddpm(arch, seed)
    n_lig = torch.tensor([5, 7][:B])
    n_poc = torch.tensor([20, 24][:B])
    tape = NoiseTape(seed + 13)
    trace = []
    wrap_trace(model, trace)
    with patched_randn(tape), torch.no_grad():
        out_l, out_p, lm, pm = model.sample(B, n_lig, n_poc, timesteps=timesteps)
    arrs = dict(cfg_json=np.array(json.dumps(cfg)), ddpm_json=np.array(json.dumps(dd)),
                arch=np.array(arch), seed=np.int64(seed),
                checksum=np.array(W.state_dict_checksum(sd)),
                num_nodes_lig=n_lig, num_nodes_pocket=n_poc, timesteps=np.int64(timesteps),
                out_lig=out_l, out_pocket=out_p, lig_mask=lm, pocket_mask=pm,
                n_draws=np.int64(len(tape.draws)), n_steps=np.int64(len(trace)))
    for i, dr in enumerate(tape.draws):
        arrs[f"noise_{i}"] = dr
    for i, (s, t, zt, pt, zs, ps) in enumerate(trace):
        arrs[f"step{i}_s"] = s
        arrs[f"step{i}_t"] = t
        arrs[f"step{i}_zt"] = zt
        arrs[f"step{i}_pt"] = pt
        arrs[f"step{i}_zs"] = zs
        arrs[f"step{i}_ps"] = ps

    # RePaint inpainting with the joint model (pocket fixed, ligand free):
    # what generate_ligands does for the joint model (lightning_modules.py:814-834)
    model2 = build_ref_ddpm(arch, seed)[3]
    key = "fa"
    pocket = small_pocket(pockets, key, B, 24)
    lmask = torch.repeat_interleave(torch.arange(B), n_lig)
    ligand = {"x": torch.zeros(len(lmask), 3), "one_hot": torch.zeros(len(lmask), cfg["atom_nf"]),
              "size": n_lig, "mask": lmask}
    pocket_in = {k: v.clone() for k, v in pocket.items()}
    lig_fixed = torch.zeros(len(lmask))
    pocket_fixed = torch.ones(len(pocket["mask"]))
    tape3 = NoiseTape(seed + 17)
    with patched_randn(tape3), torch.no_grad():
        inp_l, inp_p, _, _ = model2.inpaint(ligand, pocket, lig_fixed, pocket_fixed,
                                            resamplings=2, jump_length=1, timesteps=4)
    arrs.update(inp_pocket_x=pocket_in["x"], inp_pocket_one_hot=pocket_in["one_hot"],
                inp_pocket_size=pocket_in["size"], inp_pocket_mask=pocket_in["mask"],
                inp_out_lig=inp_l, inp_out_pocket=inp_p, inp_timesteps=np.int64(4),
                inp_resamplings=np.int64(2), n_draws_inp=np.int64(len(tape3.draws)))
    for i, dr in enumerate(tape3.draws):
        arrs[f"inpnoise_{i}"] = dr
    save(name, **arrs)


def main():
    pockets = make_pockets()
    make_schedule()
    # one dynamics call per architecture
    make_dyn_case("dyn_small_cond", "small_cond", pockets, B=3, n_lig=[5, 9, 7], seed=1,
                  store_weights=True, store_trace=True, hetero=True)
    make_dyn_case("dyn_small_joint", "small_joint", pockets, B=2, n_lig=[6, 8], seed=2,
                  store_trace=True, hetero=True)
    make_dyn_case("dyn_small_variant", "small_variant", pockets, B=2, n_lig=[7, 4], seed=3,
                  store_trace=True, hetero=True)
    make_dyn_case("dyn_ca_cond", "crossdock_ca_cond", pockets, B=2, n_lig=23, seed=0, hetero=True)
    make_dyn_case("dyn_fullatom_cond", "crossdock_fullatom_cond", pockets, B=2, n_lig=23,
                  seed=0, hetero=True)
    make_dyn_case("dyn_fullatom_joint", "moad_fullatom_joint", pockets, B=2, n_lig=23, seed=0,
                  hetero=True)
    # sampling loops (small archs, injected noise)
    make_ddpm_cond("ddpm_small_cond", "small_cond", pockets, seed=4, timesteps=None)
    make_ddpm_cond("ddpm_small_variant", "small_variant", pockets, seed=5, B=2, n_keep=30,
                   timesteps=10)
    make_ddpm_cond_inpaint("ddpm_small_cond_inpaint", "small_cond", pockets, seed=6)
    make_ddpm_joint("ddpm_small_joint", "small_joint", pockets, seed=8)


if __name__ == "__main__":
    main()
