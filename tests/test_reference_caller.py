"""CPU: the reference's own caller of the hot path -- /root/reference/lightning_modules.py,
imported UNCHANGED -- against this repository's drop-in `equivariant_diffusion` package
(SURVEY.md 8b).  Third-party libraries that are missing here are stubbed by
oracle/ref_caller_shim.py; skipped where /root/reference does not exist (the GPU box)."""
import os
from argparse import Namespace

import numpy as np
import pytest
import torch

from oracle import ref_shim
from oracle import weights as W

pytestmark = pytest.mark.skipif(not ref_shim.reference_available(), reason="reference checkout not present")

CASES = {   # mode, dataset, pocket representation, architecture (diffsbdd_amd/synthetic.py)
    "crossdock_ca_cond": ("pocket_conditioning", "crossdock", "CA", "crossdock_ca_cond"),
    "crossdock_fullatom_cond": ("pocket_conditioning", "crossdock", "full-atom", "crossdock_fullatom_cond"),
    "moad_fullatom_joint": ("joint", "bindingmoad", "full-atom", "moad_fullatom_joint"),
}


@pytest.fixture(scope="module")
def lm():
    from oracle import ref_caller_shim
    return ref_caller_shim.import_lightning_modules()


def build(lm, name):
    mode, dataset, rep, arch = CASES[name]
    cfg, dd = W.arch_cfg(arch)
    egnn = Namespace(joint_nf=cfg["joint_nf"], device="cpu", hidden_nf=cfg["hidden_nf"], n_layers=cfg["n_layers"],
                     attention=cfg["attention"], tanh=cfg["tanh"], norm_constant=cfg["norm_constant"],
                     inv_sublayers=cfg["inv_sublayers"], sin_embedding=False,
                     normalization_factor=cfg["normalization_factor"], aggregation_method="sum",
                     edge_cutoff_ligand=cfg["edge_cutoff_ligand"], edge_cutoff_pocket=cfg["edge_cutoff_pocket"],
                     edge_cutoff_interaction=cfg["edge_cutoff_interaction"],
                     reflection_equivariant=cfg["reflection_equivariant"],
                     edge_embedding_dim=cfg["edge_embedding_dim"])
    diff = Namespace(diffusion_steps=dd["timesteps"], diffusion_noise_schedule=dd["noise_schedule"],
                     diffusion_noise_precision=dd["noise_precision"], diffusion_loss_type="l2",
                     normalize_factors=list(dd["norm_values"]))
    model = lm.LigandPocketDDPM(
        outdir="/tmp/out", dataset=dataset, datadir="/tmp/data", batch_size=8, lr=1e-3, egnn_params=egnn,
        diffusion_params=diff, num_workers=0, augment_noise=0, augment_rotation=False, clip_grad=True,
        eval_epochs=1, eval_params=Namespace(smiles_file=None, eval_batch_size=4), visualize_sample_epoch=1,
        visualize_chain_epoch=1, auxiliary_loss=False, loss_params=None, mode=mode,
        node_histogram=np.ones((40, 400)), pocket_representation=rep)
    return model, cfg, dd


@pytest.mark.parametrize("name", sorted(CASES))
def test_reference_lightning_module_builds_and_loads_the_drop_in(lm, name):
    """lightning_modules.py:59-61,137-173: the unchanged LightningModule constructs THIS repo's
    classes (exact class identity, as its `type(self.ddpm) == ...` dispatch needs) and a Lightning
    checkpoint (`ddpm.`-prefixed state_dict) loads strictly."""
    import diffsbdd_amd.conditional_model as cm
    import diffsbdd_amd.dynamics as dy
    import diffsbdd_amd.en_diffusion as ed
    model, cfg, dd = build(lm, name)
    assert lm.EGNNDynamics is dy.EGNNDynamics and lm.ConditionalDDPM is cm.ConditionalDDPM
    assert lm.EnVariationalDiffusion is ed.EnVariationalDiffusion
    want = ed.EnVariationalDiffusion if CASES[name][0] == "joint" else cm.ConditionalDDPM
    assert type(model.ddpm) == want                                   # lightning_modules.py:814,837
    assert type(model.ddpm.dynamics) == dy.EGNNDynamics
    assert model.ddpm.dynamics.update_pocket_coords == (CASES[name][0] == "joint")
    assert model.ddpm.T == dd["timesteps"] and tuple(model.ddpm.norm_values) == tuple(dd["norm_values"])
    # a Lightning checkpoint of the reference: state_dict under the `ddpm.` prefix
    sd = W.random_state_dict(cfg, seed=3)
    ckpt = {"ddpm.dynamics." + k: v for k, v in sd.items()}
    ckpt["ddpm.buffer"] = torch.zeros(1)
    ckpt["ddpm.gamma.gamma"] = model.ddpm.gamma.gamma.detach().clone()
    missing, unexpected = model.load_state_dict(ckpt, strict=True)
    assert not missing and not unexpected
    assert set(model.state_dict()) == set(ckpt)
    got = model.ddpm.dynamics.state_dict()
    assert all(torch.equal(got[k], sd[k]) for k in sd)
    # the aliased last layer of the two coordinate MLPs is ONE parameter (egnn_new.py:78,85,91)
    if not cfg["reflection_equivariant"]:
        eq = model.ddpm.dynamics.egnn.e_block_0.gcl_equiv
        assert eq.coord_mlp[4].weight is eq.cross_product_mlp[4].weight
    # configure_optimizers (lightning_modules.py:183-185) sees every parameter once
    opt = model.configure_optimizers()
    n_opt = sum(p.numel() for g in opt.param_groups for p in g["params"])
    assert n_opt == sum(p.numel() for p in model.ddpm.parameters())


@pytest.mark.parametrize("name", sorted(CASES))
def test_reference_generate_ligands_drives_the_drop_in_sampler(lm, name, monkeypatch):
    """The reference's `generate_ligands` (lightning_modules.py:754-872), unchanged, up to and after
    the sampler call: its own pocket selection (utils.py:103-128) and featurisation (:714-752) on the
    3rfm example must equal diffsbdd_amd.pocket's result, and the sampler of THIS repo receives the
    argument types its API documents.  (The sampler itself needs a GPU: it is replaced by a recorder
    here; the GPU suite runs it.)"""
    from diffsbdd_amd import pocket as pk
    model, cfg, dd = build(lm, name)
    pdb = os.path.join(ref_shim.REF_ROOT, "example", "3rfm.pdb")
    sdf = os.path.join(ref_shim.REF_ROOT, "example", "3rfm_B_CFF.sdf")
    rep = CASES[name][2]
    ours = pk.pocket_from_files(pdb, sdf, representation=rep, repeats=3)
    seen = {}
    a = cfg["atom_nf"]

    def fake_outputs(pocket, num_nodes_lig):
        n = len(num_nodes_lig)
        lig_mask = torch.repeat_interleave(torch.arange(n), num_nodes_lig)
        xh_lig = torch.cat([torch.randn(len(lig_mask), 3), torch.nn.functional.one_hot(
            torch.randint(0, a, (len(lig_mask),)), a).float()], 1)
        xh_pocket = torch.cat([pocket["x"].float(), pocket["one_hot"].float()], 1)
        return xh_lig, xh_pocket, lig_mask, pocket["mask"]

    def rec_sample_given_pocket(pocket, num_nodes_lig, return_frames=1, timesteps=None):
        seen.update(kind="sample_given_pocket", pocket=pocket, n=num_nodes_lig, timesteps=timesteps)
        return fake_outputs(pocket, num_nodes_lig)

    def rec_inpaint(ligand, pocket, lig_fixed, pocket_fixed, resamplings=1, jump_length=1, return_frames=1,
                    timesteps=None):
        seen.update(kind="inpaint", pocket=pocket, ligand=ligand, n=ligand["size"], lig_fixed=lig_fixed,
                    pocket_fixed=pocket_fixed, resamplings=resamplings, timesteps=timesteps)
        return fake_outputs(pocket, ligand["size"])

    if hasattr(model.ddpm, "sample_given_pocket"):
        monkeypatch.setattr(model.ddpm, "sample_given_pocket", rec_sample_given_pocket, raising=True)
    monkeypatch.setattr(model.ddpm, "inpaint", rec_inpaint, raising=True)
    kwargs = dict(resamplings=2, jump_length=1) if CASES[name][0] == "joint" else {}
    mols = model.generate_ligands(pdb, 3, ref_ligand=sdf, timesteps=50, n_nodes_min=1, **kwargs)
    assert len(mols) == 3
    assert seen["kind"] == ("inpaint" if CASES[name][0] == "joint" else "sample_given_pocket")
    p = seen["pocket"]
    # f-1: the reference's selection + featurisation == ours, entry by entry
    assert torch.equal(p["x"].cpu(), ours["x"]) and torch.equal(p["one_hot"].cpu(), ours["one_hot"])
    assert torch.equal(p["size"].cpu(), ours["size"]) and torch.equal(p["mask"].cpu(), ours["mask"])
    assert p["mask"].dtype == torch.int64 and p["x"].dtype == torch.float32
    assert seen["timesteps"] == 50 and len(seen["n"]) == 3 and seen["n"].dtype == torch.int64
    if CASES[name][0] == "joint":
        assert seen["resamplings"] == 2 and float(seen["lig_fixed"].sum()) == 0
        assert float(seen["pocket_fixed"].sum()) == len(p["mask"])
        assert seen["ligand"]["x"].shape == (int(seen["n"].sum()), 3)


def test_learned_noise_schedule_matches_the_reference_network():
    """noise_schedule='learned' (the reference's DEFAULT constructor argument): our GammaNetwork carries the
    reference's parameter names, loads its state_dict and returns its gamma(t) (en_diffusion.py:1030-1102);
    the samplers' table is the network at t = k / T."""
    import io
    from contextlib import redirect_stdout
    mods = ref_shim.import_reference()
    ref_ed = mods["en_diffusion"] if isinstance(mods, dict) else mods[1]
    from diffsbdd_amd.en_diffusion import EnVariationalDiffusion, GammaNetwork
    torch.manual_seed(3)
    with redirect_stdout(io.StringIO()):
        ref = ref_ed.GammaNetwork()
    ours = GammaNetwork(timesteps=40)
    assert sorted(ours.state_dict().keys()) == sorted(ref.state_dict().keys())
    ours.load_state_dict(ref.state_dict())
    t = torch.rand(17, 1)
    assert torch.allclose(ours(t), ref(t), atol=1e-6, rtol=1e-6)
    tab = ours.gamma
    assert tab.shape == (41,) and abs(tab[0].item() + 5.0) < 1e-5 and abs(tab[-1].item() - 10.0) < 1e-5
    assert (tab[1:] >= tab[:-1]).all()                       # monotone
    assert torch.allclose(tab, ref(torch.arange(41).float().view(-1, 1) / 40).view(-1), atol=1e-5)
    # the DDPM accepts the schedule (vlb only, as the reference asserts)
    dyn = torch.nn.Linear(1, 1)
    dyn.update_pocket_coords = True
    m = EnVariationalDiffusion(dynamics=dyn, atom_nf=10, residue_nf=10, n_dims=3, size_histogram=np.ones((4, 4)),
                               timesteps=40, noise_schedule="learned", loss_type="vlb", norm_values=(1.0, 1.0))
    assert "gamma.l2.weight" in m.state_dict() and m._coefs(40) is not None
    # norm_values (1, 4) -- every full-atom / moad config -- with a FRESH learned schedule: gamma_0 = -5 gives
    # sigma_0 * 8 = 0.65 > 1 / 4; the reference skips check_issues_norm_values for learned schedules
    # (en_diffusion.py:65), so construction must not raise
    m4 = EnVariationalDiffusion(dynamics=dyn, atom_nf=10, residue_nf=10, n_dims=3, size_histogram=np.ones((4, 4)),
                                timesteps=40, noise_schedule="learned", loss_type="vlb", norm_values=(1.0, 4.0))
    # the step coefficients follow the parameters: an in-place update (optimiser step) must not be served from the cache
    c0 = m4._coefs(40)
    with torch.no_grad():
        m4.gamma.gamma_1.add_(1.0)
    c1 = m4._coefs(40)
    assert c1 is not c0
    with torch.no_grad():
        m4.gamma.gamma_1.sub_(1.0)
    assert m4._coefs(40) is not c1
