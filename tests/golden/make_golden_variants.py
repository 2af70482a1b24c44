#!/usr/bin/env python
"""Golden vectors for the API variants of the sampling loops that the main goldens do not touch, from the REAL
reference (run in the build container only:  python tests/golden/make_golden_variants.py):

  frames    ConditionalDDPM.sample_given_pocket(return_frames = timesteps)        conditional_model.py:478-555
            EnVariationalDiffusion.sample(return_frames = timesteps)              en_diffusion.py:580-651
  simple    SimpleConditionalDDPM.sample_given_pocket (no COM projection)         conditional_model.py:702-746
  pocketc   ConditionalDDPM.inpaint(center='pocket', return_frames = timesteps)   conditional_model.py:557-686
  jump      EnVariationalDiffusion.inpaint(resamplings=2, jump_length=2) with some ligand atoms fixed and the
            pocket free apart from the first sample's                                en_diffusion.py:676-837
All with an injected noise tape (every torch.randn of the reference goes through it)."""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)

import make_golden as mg  # noqa: E402
from oracle.ddpm_oracle import NoiseTape  # noqa: E402


def base(cfg, dd, sd, seed):
    return dict(cfg_json=np.array(json.dumps(cfg)), ddpm_json=np.array(json.dumps(dd)), seed=np.int64(seed),
                checksum=np.array(mg.W.state_dict_checksum(sd)))


def put(arrs, prefix, d):
    for k, v in d.items():
        arrs[prefix + k] = v


def tape_to(arrs, prefix, tape):
    arrs["n_" + prefix] = np.int64(len(tape.draws))
    for i, d in enumerate(tape.draws):
        arrs[f"{prefix}_{i}"] = d


def main():
    pockets = mg.make_pockets()
    # ---- frames + pocket centering + simple model: small_cond weights ----------------------------------------
    cfg, dd, sd, model = mg.build_ref_ddpm("small_cond", 31)
    arrs = base(cfg, dd, sd, 31)
    B, T = 2, 4
    pocket = mg.small_pocket(pockets, "fa", B, 30)
    put(arrs, "pocket_", {k: v.clone() for k, v in pocket.items()})
    n_lig = torch.tensor([5, 7])
    arrs["num_nodes_lig"] = n_lig
    tape = NoiseTape(41)
    with mg.patched_randn(tape), torch.no_grad():
        fl, fp, _, _ = model.sample_given_pocket({k: v.clone() for k, v in pocket.items()}, n_lig,
                                                 return_frames=T, timesteps=T)
    arrs.update(frames_lig=fl, frames_pocket=fp, timesteps=np.int64(T))
    tape_to(arrs, "fnoise", tape)
    # inpaint centred at the pocket, all frames
    g = torch.Generator().manual_seed(5)
    lm = torch.repeat_interleave(torch.arange(B), n_lig)
    com = torch.stack([pocket["x"][pocket["mask"] == b].mean(0) for b in range(B)])
    ligand = {"x": com[lm] + torch.randn(len(lm), 3, generator=g),
              "one_hot": torch.nn.functional.one_hot(torch.randint(0, cfg["atom_nf"], (len(lm),), generator=g),
                                                     cfg["atom_nf"]), "size": n_lig, "mask": lm}
    put(arrs, "ligand_", {k: v.clone() for k, v in ligand.items()})
    lig_fixed = torch.zeros(len(lm)); lig_fixed[[0, 2, 5, 6, 11]] = 1
    arrs["lig_fixed"] = lig_fixed
    tape = NoiseTape(43)
    with mg.patched_randn(tape), torch.no_grad():
        pl, pp, _, _ = model.inpaint({k: v.clone() for k, v in ligand.items()}, {k: v.clone() for k, v in pocket.items()},
                                     lig_fixed, resamplings=2, return_frames=T, timesteps=T, center="pocket")
    arrs.update(pocketc_lig=pl, pocketc_pocket=pp)
    tape_to(arrs, "pnoise", tape)
    # SimpleConditionalDDPM with the same weights
    dyn = mg.build_ref_dynamics(cfg, sd)
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        simple = mg.cond_mod.SimpleConditionalDDPM(
            dynamics=dyn, atom_nf=cfg["atom_nf"], residue_nf=cfg["residue_nf"], n_dims=3,
            size_histogram=np.ones((12, 60)), timesteps=dd["timesteps"], noise_schedule=dd["noise_schedule"],
            noise_precision=dd["noise_precision"], loss_type="l2", norm_values=dd["norm_values"]).eval()
    tape = NoiseTape(45)
    with mg.patched_randn(tape), torch.no_grad():
        sl, sp, _, _ = simple.sample_given_pocket({k: v.clone() for k, v in pocket.items()}, n_lig, timesteps=T)
    arrs.update(simple_lig=sl, simple_pocket=sp)
    tape_to(arrs, "snoise", tape)
    mg.save("ddpm_variants_cond", **arrs)

    # ---- joint: frames of sample(); RePaint with jump_length = 2 and a mixed fixed set --------------------------
    cfg, dd, sd, model = mg.build_ref_ddpm("small_joint", 33)
    arrs = base(cfg, dd, sd, 33)
    n_lig, n_poc = torch.tensor([5, 6]), torch.tensor([18, 21])
    arrs.update(num_nodes_lig=n_lig, num_nodes_pocket=n_poc)
    tape = NoiseTape(47)
    with mg.patched_randn(tape), torch.no_grad():
        fl, fp, _, _ = model.sample(2, n_lig, n_poc, return_frames=4, timesteps=4)
    arrs.update(frames_lig=fl, frames_pocket=fp, timesteps=np.int64(4))
    tape_to(arrs, "fnoise", tape)
    pocket = mg.small_pocket(pockets, "fa", 2, 20)
    lm = torch.repeat_interleave(torch.arange(2), n_lig)
    g = torch.Generator().manual_seed(6)
    com = torch.stack([pocket["x"][pocket["mask"] == b].mean(0) for b in range(2)])
    ligand = {"x": com[lm] + torch.randn(len(lm), 3, generator=g),
              "one_hot": torch.nn.functional.one_hot(torch.randint(0, cfg["atom_nf"], (len(lm),), generator=g),
                                                     cfg["atom_nf"]), "size": n_lig, "mask": lm}
    put(arrs, "ligand_", {k: v.clone() for k, v in ligand.items()})
    put(arrs, "inp_pocket_", {k: v.clone() for k, v in pocket.items()})
    lig_fixed = torch.zeros(len(lm)); lig_fixed[[1, 2, 7]] = 1
    pocket_fixed = (pocket["mask"] == 0).float()
    arrs.update(lig_fixed=lig_fixed, pocket_fixed=pocket_fixed)
    tape = NoiseTape(49)
    with mg.patched_randn(tape), torch.no_grad():
        jl, jp, _, _ = model.inpaint({k: v.clone() for k, v in ligand.items()}, {k: v.clone() for k, v in pocket.items()},
                                     lig_fixed, pocket_fixed, resamplings=2, jump_length=2, timesteps=6)
    arrs.update(jump_lig=jl, jump_pocket=jp, jump_timesteps=np.int64(6))
    tape_to(arrs, "jnoise", tape)
    mg.save("ddpm_variants_joint", **arrs)


if __name__ == "__main__":
    main()
