"""GPU: the reference's own CALLERS of the hot path, unchanged, driving the HIP kernels (VERDICT r4 "missing" #2 / next #3).

`lightning_modules.py` (LigandPocketDDPM: `generate_ligands` :754-872, `forward` :236-302, `training_step` :337-363) is the
reference's file -- from /root/reference in the build container, from oracle/_ref/reference_path.zip (oracle/make_ref.py:
unmodified copies with a SHA-256 manifest, shipped with the push) on the GPU box -- imported through
oracle/ref_caller_shim.py, which stubs the third-party libraries that are absent here (pytorch_lightning, wandb, rdkit,
Bio, openbabel) and lets `equivariant_diffusion` resolve to THIS repository's drop-in package.  Nothing of the reference's
code is edited or monkeypatched except where a comment says so; the samplers and the training step run on cuda:0.

The CPU half (constructor, strict checkpoint load, pocket featurisation == ours) is tests/test_reference_caller.py."""
import os
from argparse import Namespace

import numpy as np
import pytest
import torch

from oracle import ddpm_oracle as do
from oracle import egnn_oracle as eo
from oracle import weights as W

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def lm():
    from oracle import ref_caller_shim
    root = ref_caller_shim.use_archive()
    if root is None:
        pytest.skip("neither /root/reference nor oracle/_ref/reference_path.zip (with the caller modules) is present")
    return ref_caller_shim.import_lightning_modules()


@pytest.fixture(scope="module")
def example(tmp_path_factory):
    from oracle import ref_caller_shim
    return ref_caller_shim.example_files(tmp_path_factory.mktemp("ref_example"))


def build(lm, name, auxiliary_loss=False):
    from tests.test_reference_caller import CASES
    mode, dataset, rep, arch = CASES[name]
    cfg, dd = W.arch_cfg(arch)
    egnn = Namespace(joint_nf=cfg["joint_nf"], device="cpu", hidden_nf=cfg["hidden_nf"], n_layers=cfg["n_layers"],
                     attention=cfg["attention"], tanh=cfg["tanh"], norm_constant=cfg["norm_constant"],
                     inv_sublayers=cfg["inv_sublayers"], sin_embedding=False,
                     normalization_factor=cfg["normalization_factor"], aggregation_method="sum",
                     edge_cutoff_ligand=cfg["edge_cutoff_ligand"], edge_cutoff_pocket=cfg["edge_cutoff_pocket"],
                     edge_cutoff_interaction=cfg["edge_cutoff_interaction"],
                     reflection_equivariant=cfg["reflection_equivariant"], edge_embedding_dim=cfg["edge_embedding_dim"])
    diff = Namespace(diffusion_steps=dd["timesteps"], diffusion_noise_schedule=dd["noise_schedule"],
                     diffusion_noise_precision=dd["noise_precision"], diffusion_loss_type="l2",
                     normalize_factors=list(dd["norm_values"]))
    model = lm.LigandPocketDDPM(
        outdir="/tmp/out", dataset=dataset, datadir="/tmp/data", batch_size=8, lr=1e-3, egnn_params=egnn,
        diffusion_params=diff, num_workers=0, augment_noise=0, augment_rotation=False, clip_grad=True,
        eval_epochs=1, eval_params=Namespace(smiles_file=None, eval_batch_size=4), visualize_sample_epoch=1,
        visualize_chain_epoch=1, auxiliary_loss=auxiliary_loss,
        loss_params=Namespace(max_weight=0.1, schedule="linear", clamp_lj=3.0) if auxiliary_loss else None, mode=mode,
        node_histogram=np.ones((40, 400)), pocket_representation=rep)
    sd = W.random_state_dict(cfg, seed=3)
    ckpt = {"ddpm.dynamics." + k: v for k, v in sd.items()}            # a Lightning checkpoint of the reference
    ckpt["ddpm.buffer"] = torch.zeros(1)
    ckpt["ddpm.gamma.gamma"] = model.ddpm.gamma.gamma.detach().clone()
    model.load_state_dict(ckpt, strict=True)
    return model.to(DEV), cfg, dd, sd, rep, mode


def oracle_model(sd, cfg, dd):
    return do.OracleModel(sd, cfg, cfg["atom_nf"], cfg["residue_nf"], dd["timesteps"], dd["noise_schedule"],
                          dd["noise_precision"], norm_values=dd["norm_values"], conditional=dd["conditional"])


@pytest.mark.parametrize("name,n,T", [("crossdock_ca_cond", 4, 20), ("crossdock_fullatom_cond", 3, 8), ("moad_fullatom_joint", 3, 6)])
def test_reference_generate_ligands_runs_the_hip_samplers(lm, example, name, n, T):
    """`LigandPocketDDPM.generate_ligands` (lightning_modules.py:754-872) unchanged: its own PDB / SDF handling, pocket
    selection and featurisation (on the stub structure model), its `type(self.ddpm) == ...` dispatch, the sampler of THIS
    repository on the GPU, the move back into the pocket frame (:843-852) and the per-molecule loop.  The molecules it
    hands to `build_molecule` must be those of the ORACLE's chain on the same pocket and the same noise (the drop-in
    records its draws, the oracle replays them): coordinates 1e-3 over the free-running chain, identical atom types."""
    from diffsbdd_amd import pocket as pk
    model, cfg, dd, sd, rep, mode = build(lm, name)
    pdb, sdf = example
    n_lig = torch.tensor([9, 12, 7, 10][:n])
    tape = do.NoiseTape(7)
    model.ddpm.set_noise_source(tape)
    kwargs = dict(resamplings=2, jump_length=1) if mode == "joint" else {}
    with torch.no_grad():
        mols = model.generate_ligands(pdb, n, ref_ligand=sdf, num_nodes_lig=n_lig.to(DEV), timesteps=T, **kwargs)
    assert len(mols) == n
    # the oracle on the same pocket (diffsbdd_amd.pocket == the reference's featurisation: tests/test_reference_caller.py)
    pocket = pk.pocket_from_files(pdb, sdf, representation=rep, repeats=n)
    com0 = eo.segment_mean(pocket["x"].float(), pocket["mask"], n)
    om = oracle_model(sd, cfg, dd)
    replay = do.NoiseReplay(tape.draws)
    lig_mask = torch.repeat_interleave(torch.arange(n), n_lig)
    with torch.no_grad():
        if mode == "joint":
            ligand = {"x": torch.zeros(len(lig_mask), 3), "one_hot": torch.zeros(len(lig_mask), cfg["atom_nf"]),
                      "size": n_lig, "mask": lig_mask}
            o_l, o_p, _, p_mask = do.joint_inpaint(om, ligand, pocket, torch.zeros(len(lig_mask)),
                                                   torch.ones(len(pocket["mask"])), replay, resamplings=2, jump_length=1,
                                                   timesteps=T)
        else:
            o_l, o_p, _, p_mask = do.cond_sample_given_pocket(om, pocket, n_lig, replay, timesteps=T)
    shift = com0 - eo.segment_mean(o_p[:, :3], p_mask, n)                          # lightning_modules.py:843-852
    x_ref = o_l[:, :3] + shift[lig_mask]
    t_ref = o_l[:, 3:].argmax(1)
    lo = 0
    worst = 0.0
    for k, (pos, types_) in enumerate(mols):                                       # what build_molecule was handed
        hi = lo + int(n_lig[k])
        # the reference's utils.batch_to_list (utils.py:131-143) orders the rows by an UNSTABLE argsort of the mask: the
        # atoms of a molecule may come permuted (positions and types by the same permutation) -- match them by position
        d = (pos.cpu()[:, None, :] - x_ref[lo:hi][None, :, :]).abs().amax(-1)          # [ours, oracle]
        match = d.argmin(1)
        assert sorted(match.tolist()) == list(range(hi - lo)), (name, k, match.tolist())   # a permutation
        assert torch.equal(types_.cpu(), t_ref[lo:hi][match]), (name, k)
        worst = max(worst, d.gather(1, match[:, None]).max().item())
        # free-running over the chain: 1e-3 absolute + 1e-4 relative (the untrained joint model drives |x| to several
        # hundred Angstrom, where one fp32 ulp is already 3e-5; same criterion as tests/test_gpu_parity.py's joint chains)
        ref_k = x_ref[lo:hi][match]
        assert ((pos.cpu() - ref_k).abs() - (1e-3 + 1e-4 * ref_k.abs())).max().item() <= 0, (name, k, worst, ref_k.abs().max().item())
        lo = hi
    print(f"[{name}] reference generate_ligands on the HIP samplers: {n} molecules, T = {T}, max |x - oracle| = {worst:.2e} "
          f"(max |x| = {x_ref.abs().max().item():.1f})")


def _batch(cfg, dd, rep, n=3, seed=0):
    """A training batch as dataset.collate_fn builds it (dataset.py:52-70): float masks, COM-centred complexes."""
    g = torch.Generator().manual_seed(seed)
    nl = [9, 6, 11][:n]
    npk = [20, 17, 25][:n]
    lm_ = torch.cat([i * torch.ones(k) for i, k in enumerate(nl)])
    pm_ = torch.cat([i * torch.ones(k) for i, k in enumerate(npk)])
    xl = torch.randn(sum(nl), 3, generator=g) * 1.5
    xp = torch.randn(sum(npk), 3, generator=g) * 4.0
    hl = torch.nn.functional.one_hot(torch.randint(0, cfg["atom_nf"], (sum(nl),), generator=g), cfg["atom_nf"]).float()
    hp = torch.nn.functional.one_hot(torch.randint(0, cfg["residue_nf"], (sum(npk),), generator=g), cfg["residue_nf"]).float()
    return {"lig_coords": xl, "lig_one_hot": hl, "num_lig_atoms": torch.tensor(nl), "lig_mask": lm_,
            "pocket_coords": xp, "pocket_one_hot": hp, "num_pocket_nodes": torch.tensor(npk), "pocket_mask": pm_}


@pytest.mark.parametrize("name,aux", [("crossdock_ca_cond", False), ("crossdock_ca_cond", True), ("moad_fullatom_joint", False),
                                      ("crossdock_fullatom_cond", True)])
def test_reference_training_step_on_the_hip_kernels(lm, name, aux):
    """`training_step` -> `forward` (lightning_modules.py:337-363, 236-302) unchanged, in training mode on the GPU: the
    drop-in's `ddpm(ligand, pocket, return_info=True)` runs forward AND backward on the HIP kernels (train_hip.py).
    `auxiliary_loss=True` adds the Lennard-Jones term on `xh_lig_hat` (:283-291), which must carry its graph.
    Checked: the reference's nll / loss against the same formula evaluated on the oracle's loss terms (same t, same
    noise) <= 1e-4 relative; `loss.backward()` leaves a finite gradient on every parameter that the oracle's autograd
    also reaches, equal to it at 1e-4 of the gradient's largest entry."""
    model, cfg, dd, sd, rep, mode = build(lm, name, auxiliary_loss=aux)
    data = _batch(cfg, dd, rep)
    n = len(data["num_lig_atoms"])
    t_int = torch.tensor([[3.0], [250.0], [477.0]])[:n]
    tape = do.NoiseTape(11)
    model.ddpm.set_noise_source(tape)
    model.ddpm.t_int_source = lambda b: t_int
    if aux:
        # (the reference indexes a CPU weight table with the device's t_int, lightning_modules.py:900-914 -- torch >= 2
        #  refuses mixed devices there; moving the table is data placement, not a code change)
        model.auxiliary_weight_schedule.weights = model.auxiliary_weight_schedule.weights.to(DEV)
    model.train()
    info = model.training_step({k: v.clone() for k, v in data.items()})
    loss = info["loss"]
    assert loss.requires_grad and torch.isfinite(loss)
    loss.backward()
    # ---- the same objective on the oracle: loss terms -> the reference's l2 training formula (:254-275, 283-291)
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    om = oracle_model(sdg, cfg, dd)
    ligand = {"x": data["lig_coords"].clone(), "one_hot": data["lig_one_hot"].clone(), "size": data["num_lig_atoms"],
              "mask": data["lig_mask"].long()}
    pocket = {"x": data["pocket_coords"].clone(), "one_hot": data["pocket_one_hot"].clone(), "size": data["num_pocket_nodes"],
              "mask": data["pocket_mask"].long()}
    terms = do.loss_terms(om, ligand, pocket, t_int, do.NoiseReplay(tape.draws), True)
    (delta_log_px, err_l, err_p, snr, l0_xl, l0_xp, l0_h, nlc0, kl_prior, log_pn, t_o, xh_hat) = terms
    a, r = cfg["atom_nf"], cfg["residue_nf"]
    sz_l, sz_p = ligand["size"].float(), pocket["size"].float()
    loss_t = 0.5 * (err_l / (3 * sz_l + a * sz_l) + err_p / ((3 + r) * sz_p))
    loss_0 = l0_xl / (3 * sz_l) + l0_xp / (3 * sz_p) + l0_h
    nll = loss_t + loss_0 + kl_prior
    if aux:
        ref_lm = model                                            # the reference's own lj_potential / schedule on the oracle's x_hat
        w = ref_lm.auxiliary_weight_schedule.weights.cpu()[torch.as_tensor(t_o).long().view(-1)]
        cpu_self = Namespace(lj_rm=ref_lm.lj_rm, ddpm=Namespace(norm_values=dd["norm_values"]), clamp_lj=ref_lm.clamp_lj)
        lj = type(ref_lm).lj_potential(cpu_self, xh_hat[:, :3], xh_hat[:, 3:], ligand["mask"])
        nll = nll + w * lj
    ref_loss = nll.mean(0)
    rel = abs(loss.item() - ref_loss.item()) / max(1.0, abs(ref_loss.item()))
    print(f"[{name} aux={aux}] reference training_step on the HIP kernels: loss {loss.item():.6f} vs oracle {ref_loss.item():.6f} (rel {rel:.1e})")
    assert rel <= 1e-4, (loss.item(), ref_loss.item())
    ref_loss.backward()
    n_checked = 0
    for pname, p in model.ddpm.dynamics.named_parameters():
        g_ref = sdg[pname].grad
        if pname.endswith("coord_mlp.4.weight"):                 # one Parameter shared by both coordinate MLPs
            twin = pname.replace("coord_mlp", "cross_product_mlp")
            if twin in sdg and sdg[twin].grad is not None:
                g_ref = g_ref + sdg[twin].grad
        if g_ref is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, pname
            continue
        assert p.grad is not None and torch.isfinite(p.grad).all(), pname
        scale = max(g_ref.abs().max().item(), 1e-6)
        err = (p.grad.cpu() - g_ref).abs().max().item()
        assert err <= 1e-4 * scale + 1e-7, (pname, err, scale)
        n_checked += 1
    assert n_checked >= 40
