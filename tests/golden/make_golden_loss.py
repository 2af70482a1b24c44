#!/usr/bin/env python
"""Golden vectors of the training / validation loss terms from the REAL reference:
`ConditionalDDPM.forward` (conditional_model.py:202-330) and `EnVariationalDiffusion.forward`
(en_diffusion.py:336-469), eval mode (two network passes) and training mode (masked L_0), with the
random draws pinned: torch.randint returns a fixed t_int, torch.randn goes through a NoiseTape.
Run in the build container only:   python tests/golden/make_golden_loss.py"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)

import make_golden as mg  # noqa: E402  (imports the reference through oracle/ref_shim.py)
from oracle.ddpm_oracle import NoiseTape  # noqa: E402


class patched_randint:
    def __init__(self, value):
        self.value = value

    def __enter__(self):
        self.orig = torch.randint
        torch.randint = lambda *a, **k: self.value.clone().long()
        return self

    def __exit__(self, *a):
        torch.randint = self.orig


def make_case(name, arch, seed, t_values, training):
    cfg, dd, sd, model = mg.build_ref_ddpm(arch, seed)
    model.train(training)
    pockets = mg.make_pockets()
    key = "ca" if cfg["residue_nf"] == 20 else "fa"
    B = len(t_values)
    pocket = mg.small_pocket(pockets, key, B, 40)
    g = torch.Generator().manual_seed(seed)
    n_lig = torch.tensor([5, 8, 6, 7][:B])
    lm = torch.repeat_interleave(torch.arange(B), n_lig)
    com = torch.stack([pocket["x"][pocket["mask"] == b].mean(0) for b in range(B)])
    types = torch.randint(0, cfg["atom_nf"], (len(lm),), generator=g)
    ligand = {"x": com[lm] + torch.randn(len(lm), 3, generator=g) * 1.5,
              "one_hot": torch.nn.functional.one_hot(types, cfg["atom_nf"]).float(), "size": n_lig, "mask": lm}
    pocket = {k: (v.float() if k in ("x", "one_hot") else v) for k, v in pocket.items()}
    lig_in = {k: v.clone() for k, v in ligand.items()}
    poc_in = {k: v.clone() for k, v in pocket.items()}
    t_int = torch.tensor(t_values, dtype=torch.float32).view(B, 1)
    tape = NoiseTape(seed + 11)
    with mg.patched_randn(tape), patched_randint(t_int), torch.no_grad():
        out = model(ligand, pocket, return_info=True)
    terms, info = out[:12], out[12]
    arrs = dict(cfg_json=np.array(json.dumps(cfg)), ddpm_json=np.array(json.dumps(dd)), seed=seed,
                checksum=np.array(mg.W.state_dict_checksum(sd)), training=int(training), t_int=t_int,
                n_draws=len(tape.draws))
    for k, v in lig_in.items():
        arrs["ligand_" + k] = v
    for k, v in poc_in.items():
        arrs["pocket_" + k] = v
    for i, d in enumerate(tape.draws):
        arrs[f"noise_{i}"] = d
    names = ("delta_log_px", "error_t_lig", "error_t_pocket", "SNR_weight", "loss_0_x_ligand", "loss_0_x_pocket",
             "loss_0_h", "neg_log_constants", "kl_prior", "log_pN", "t_int_out", "xh_lig_hat")
    for nme, v in zip(names, terms):
        arrs["out_" + nme] = torch.as_tensor(v).float()
    for k, v in info.items():
        arrs["info_" + k] = torch.as_tensor(v).float()
    mg.save(name, **arrs)


if __name__ == "__main__":
    make_case("loss_small_cond_eval", "small_cond", 21, [3, 20, 11], training=False)
    make_case("loss_small_cond_train", "small_cond", 22, [0, 20, 7], training=True)
    make_case("loss_small_joint_eval", "small_joint", 23, [5, 19], training=False)
    make_case("loss_small_joint_train", "small_joint", 24, [0, 13], training=True)
