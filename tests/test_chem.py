"""Molecule post-processing / generation driver.

CPU part: tables and oracle against the reference-generated goldens
(tests/golden/make_golden_chem.py), host logic (fragments, SDF, PDB selection,
checkpoint round trip).  GPU part: `dsbdd_bond_orders` bit-exact against the
oracle, and the generation driver end to end on the HIP path.
"""
import argparse
import os

import numpy as np
import pytest
import torch

from diffsbdd_amd import chem_tables, synthetic
from oracle import chem_oracle

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _hp(arch, mode, pocket_representation):
    """Hyper-parameters in the layout the reference's LightningModule saves
    (lightning_modules.py:32-55: Namespace objects for the *_params groups)."""
    cfg, dd = synthetic.arch_cfg(arch)
    egnn = argparse.Namespace(
        device="cuda", joint_nf=cfg["joint_nf"], hidden_nf=cfg["hidden_nf"], n_layers=cfg["n_layers"],
        attention=cfg["attention"], tanh=cfg["tanh"], norm_constant=cfg["norm_constant"],
        inv_sublayers=cfg["inv_sublayers"], sin_embedding=cfg["sin_embedding"],
        normalization_factor=cfg["normalization_factor"], aggregation_method=cfg["aggregation_method"],
        edge_cutoff_ligand=cfg["edge_cutoff_ligand"], edge_cutoff_pocket=cfg["edge_cutoff_pocket"],
        edge_cutoff_interaction=cfg["edge_cutoff_interaction"],
        reflection_equivariant=cfg["reflection_equivariant"], edge_embedding_dim=cfg["edge_embedding_dim"])
    diff = argparse.Namespace(
        diffusion_steps=dd["timesteps"], diffusion_noise_schedule=dd["noise_schedule"],
        diffusion_noise_precision=dd["noise_precision"], diffusion_loss_type="l2",
        normalize_factors=list(dd["norm_values"]))
    return dict(outdir="x", dataset="crossdock", datadir="x", batch_size=8, lr=1e-3, egnn_params=egnn,
                diffusion_params=diff, num_workers=0, augment_noise=0, augment_rotation=False,
                clip_grad=True, eval_epochs=1, eval_params=argparse.Namespace(), visualize_sample_epoch=1,
                visualize_chain_epoch=1, auxiliary_loss=False, loss_params=argparse.Namespace(),
                mode=mode, node_histogram=np.ones((40, 400)), pocket_representation=pocket_representation,
                virtual_nodes=False)


# a tiny synthetic complex: 4 residues around a 3-atom HETATM ligand, one far residue
_PDB = """\
ATOM      1  N   ALA A   1       0.000   0.000   0.000  1.00  0.00           N
ATOM      2  CA  ALA A   1       1.458   0.000   0.000  1.00  0.00           C
ATOM      3  C   ALA A   1       2.009   1.420   0.000  1.00  0.00           C
ATOM      4  O   ALA A   1       1.251   2.390   0.000  1.00  0.00           O
ATOM      5  N   GLY A   2       3.332   1.536   0.000  1.00  0.00           N
ATOM      6  CA  GLY A   2       3.988   2.839   0.000  1.00  0.00           C
ATOM      7  C   GLY A   2       5.503   2.693   0.000  1.00  0.00           C
ATOM      8  O   GLY A   2       6.042   1.587   0.000  1.00  0.00           O
ATOM      9  N   SER A   3       6.190   3.830   0.000  1.00  0.00           N
ATOM     10  CA  SER A   3       7.646   3.830   0.000  1.00  0.00           C
ATOM     11  OG  SER A   3       8.100   5.100   0.500  1.00  0.00           O
ATOM     12  N   CYS B   7       4.000   6.000   2.000  1.00  0.00           N
ATOM     13  CA  CYS B   7       4.500   7.300   2.300  1.00  0.00           C
ATOM     14  SG  CYS B   7       5.900   7.200   3.500  1.00  0.00           S
ATOM     15  N   LEU A   9      40.000  40.000  40.000  1.00  0.00           N
ATOM     16  CA  LEU A   9      41.458  40.000  40.000  1.00  0.00           C
HETATM   17  C1  LIG A 100       4.000   4.000   3.000  1.00  0.00           C
HETATM   18  O1  LIG A 100       5.200   4.300   3.200  1.00  0.00           O
HETATM   19  N1  LIG A 100       3.300   5.100   3.400  1.00  0.00           N
END
"""


# --------------------------------------------------------------------------- CPU
def test_tables_match_reference_constants():
    z = np.load(os.path.join(GOLD, "chem_tables.npz"))
    assert tuple(z["margins"]) == chem_tables.MARGINS_PM
    for name in ("crossdock", "bindingmoad", "crossdock_full"):
        info = chem_tables.dataset_info(name)
        for k in ("bonds1", "bonds2", "bonds3"):
            assert np.array_equal(z[f"{name}_{k}"], info[k]), (name, k)
        assert list(z[f"{name}_atom_decoder"]) == info["atom_decoder"]
        assert list(z[f"{name}_aa_decoder"]) == info["aa_decoder"]


def test_oracle_bond_orders_match_reference():
    z = np.load(os.path.join(GOLD, "chem_bonds.npz"))
    info = chem_tables.dataset_info("crossdock")
    got = chem_oracle.bond_orders_dense(z["x"], z["atom_type"], z["sizes"], info, z["order"].shape[1])
    assert np.array_equal(got, z["order"])
    assert all((z["order"] == k).sum() > 0 for k in (1, 2, 3))


def test_molecule_fragments_and_sdf(tmp_path):
    from diffsbdd_amd.molecules import Molecule, write_sdf
    pos = np.arange(18, dtype=np.float32).reshape(6, 3) / 7
    m = Molecule(pos, list("CCNOCS"), [(1, 0, 1), (4, 3, 2), (5, 4, 1)])
    assert m.fragments() == [[3, 4, 5], [0, 1], [2]]
    big = m.largest_fragment()
    assert big.symbols == ["O", "C", "S"] and big.bonds == [(1, 0, 2), (2, 1, 1)]
    assert np.array_equal(big.positions, pos[[3, 4, 5]])
    p = tmp_path / "out.sdf"
    write_sdf(p, [m, big])
    blocks = p.read_text().split("$$$$\n")[:-1]
    assert len(blocks) == 2
    lines = blocks[0].splitlines()
    assert lines[3].startswith("  6  3") and lines[3].endswith("V2000")
    assert lines[4].split()[:4] == ["0.0000", "0.1429", "0.2857", "C"]
    assert lines[10].split() == ["1", "2", "1", "0"] and lines[-1] == "M  END"
    # our own SDF coordinate reader understands what we wrote
    from diffsbdd_amd.pocket import read_sdf_coords
    assert np.allclose(read_sdf_coords(p), pos, atol=5e-5)


def test_virtual_nodes_checkpoints_build_like_the_reference():
    """lightning_modules.py:116-135,161-173: `virtual_nodes=True` adds the class 'Ne' to the ligand vocabulary (atom_nf
    11), hands its index to the DDPM as virtual_node_idx, and -- the encoder / decoder objects being shared -- widens a
    full-atom pocket's features too; C-alpha pockets keep their 20 classes.  Sampling drops the virtual atoms."""
    from diffsbdd_amd.generate import LigandGenerator
    keys = ("dataset", "egnn_params", "diffusion_params", "mode", "node_histogram", "pocket_representation", "virtual_nodes")
    for rep, arch, want_r in (("full-atom", "small_cond", 11), ("CA", "small_variant", 20)):
        hp = _hp(arch, "pocket_conditioning", rep)
        hp["virtual_nodes"] = True
        gen = LigandGenerator(**{k: hp[k] for k in keys}, device="cpu")
        assert gen.atom_nf == 11 and gen.aa_nf == want_r and gen.virtual_atom == 10 and gen.lig_type_decoder[-1] == "Ne"
        assert gen.ddpm.vnode_idx == 10 and gen.max_num_nodes == 39
        sd = gen.ddpm.dynamics.state_dict()
        assert sd["atom_encoder.0.weight"].shape[1] == 11 and sd["residue_encoder.0.weight"].shape[1] == want_r
        xh = torch.zeros(5, 3 + 11)
        xh[torch.arange(5), 3 + torch.tensor([0, 10, 2, 10, 10])] = 1.0
        x, at, m = gen._drop_virtual(xh, torch.tensor([0, 0, 0, 1, 1]))
        assert at.tolist() == [0, 2] and m.tolist() == [0, 0] and x.shape == (2, 3)
        # a positive size bias never asks for more nodes than max_num_nodes; the minimum still applies
        assert gen._ligand_sizes(torch.tensor([39, 39, 20]), n_nodes_bias=5).tolist() == [39, 39, 25]
        assert gen._ligand_sizes(torch.tensor([3]), n_nodes_bias=-2, n_nodes_min=4).tolist() == [4]
    plain = LigandGenerator(**{k: _hp("small_cond", "pocket_conditioning", "full-atom")[k] for k in keys}, device="cpu")
    assert plain.atom_nf == 10 and plain.virtual_atom is None and len(plain.dataset_info["atom_decoder"]) == 10
    assert plain._ligand_sizes(torch.tensor([39]), n_nodes_bias=5).tolist() == [44]      # no virtual nodes: the reference's rule


def test_pocket_selection_like_generate_ligands(tmp_path):
    from diffsbdd_amd.generate import LigandGenerator
    pdb = tmp_path / "c.pdb"
    pdb.write_text(_PDB)
    gen = LigandGenerator(**{k: v for k, v in _hp("small_cond", "pocket_conditioning", "full-atom").items()
                             if k in ("dataset", "egnn_params", "diffusion_params", "mode", "node_histogram",
                                      "pocket_representation", "virtual_nodes")}, device="cpu")
    by_lig = gen.select_pocket_residues(str(pdb), ref_ligand="A:100")
    assert [(r["chain"], r["resseq"]) for r in by_lig] == [("A", 1), ("A", 2), ("A", 3), ("B", 7)]
    by_ids = gen.select_pocket_residues(str(pdb), pocket_ids=["A:2", "B:7"])
    assert [r["resname"] for r in by_ids] == ["GLY", "CYS"]
    sdf = tmp_path / "lig.sdf"
    from diffsbdd_amd.molecules import Molecule
    sdf.write_text(Molecule(np.asarray([[4, 4, 3], [5.2, 4.3, 3.2], [3.3, 5.1, 3.4]], np.float32),
                            ["C", "O", "N"], []).to_sdf_block())
    by_sdf = gen.select_pocket_residues(str(pdb), ref_ligand=str(sdf))
    assert [(r["chain"], r["resseq"]) for r in by_sdf] == [("A", 1), ("A", 2), ("A", 3), ("B", 7)]
    pocket = gen.prepare_pocket(by_lig, repeats=3)
    assert pocket["x"].shape == (3 * 14, 3) and pocket["one_hot"].shape == (3 * 14, 10)
    assert pocket["size"].tolist() == [14, 14, 14] and pocket["mask"].tolist() == sum(([b] * 14 for b in range(3)), [])
    # sulphur of CYS B7 is class 3 of the atom vocabulary
    assert int(pocket["one_hot"][13].argmax()) == 3
    with pytest.raises(AssertionError):
        gen.select_pocket_residues(str(pdb))
    with pytest.raises(NotImplementedError):
        gen.generate_ligands(str(pdb), 2, ref_ligand="A:100", sanitize=True)


@pytest.mark.parametrize("arch,mode,rep", [("small_cond", "pocket_conditioning", "full-atom"),
                                           ("small_joint", "joint", "full-atom"),
                                           ("crossdock_ca_cond", "pocket_conditioning", "CA")])
def test_checkpoint_round_trip(tmp_path, arch, mode, rep):
    """A Lightning-format checkpoint (hyper_parameters + 'ddpm.'-prefixed state_dict) written
    from reference-named weights loads into the drop-in modules unchanged."""
    from diffsbdd_amd.generate import LigandGenerator
    hp = _hp(arch, mode, rep)
    cfg, _ = synthetic.arch_cfg(arch)
    dyn_sd = synthetic.random_state_dict(cfg, seed=3)
    T = hp["diffusion_params"].diffusion_steps
    sd = {"ddpm.dynamics." + k: v for k, v in dyn_sd.items()}
    sd["ddpm.buffer"] = torch.zeros(1)
    sd["ddpm.gamma.gamma"] = torch.zeros(T + 1)      # overwritten below with the real table
    from diffsbdd_amd.en_diffusion import PredefinedNoiseSchedule
    sd["ddpm.gamma.gamma"] = PredefinedNoiseSchedule(
        hp["diffusion_params"].diffusion_noise_schedule, T,
        hp["diffusion_params"].diffusion_noise_precision).gamma.detach().clone()
    path = tmp_path / "last.ckpt"
    torch.save({"state_dict": sd, "hyper_parameters": hp, "epoch": 3, "global_step": 7}, path)
    gen = LigandGenerator.from_checkpoint(str(path), device="cpu")
    assert type(gen.ddpm).__name__ == {"joint": "EnVariationalDiffusion",
                                       "pocket_conditioning": "ConditionalDDPM"}[mode]
    got = gen.ddpm.state_dict()
    for k, v in sd.items():
        assert torch.equal(got[k[len("ddpm."):]].cpu(), v), k
    assert gen.ddpm.dynamics.update_pocket_coords == (mode == "joint")
    assert gen.aa_nf == (20 if rep == "CA" else 10)
    # a checkpoint with a wrong tensor name is refused
    bad = dict(sd)
    bad["ddpm.dynamics.egnn.bogus.weight"] = torch.zeros(1)
    torch.save({"state_dict": bad, "hyper_parameters": hp}, path)
    with pytest.raises(ValueError):
        LigandGenerator.from_checkpoint(str(path), device="cpu")


def test_bond_orders_refuses_cpu_tensors():
    from diffsbdd_amd import _lib
    from diffsbdd_amd.molecules import bond_orders
    with pytest.raises(_lib.HipLibraryError):
        bond_orders(torch.zeros(3, 3), torch.zeros(3, dtype=torch.int64), torch.zeros(3, dtype=torch.int64),
                    chem_tables.dataset_info("crossdock"))


# --------------------------------------------------------------------------- GPU
@pytest.mark.gpu
def test_bond_orders_kernel_matches_golden_and_oracle():
    from diffsbdd_amd.molecules import bond_orders, build_molecules
    z = np.load(os.path.join(GOLD, "chem_bonds.npz"))
    info = chem_tables.dataset_info("crossdock")
    d = torch.device("cuda", 0)
    mask = torch.repeat_interleave(torch.arange(len(z["sizes"])), torch.as_tensor(z["sizes"])).to(d)
    got, sizes = bond_orders(torch.as_tensor(z["x"]).to(d), torch.as_tensor(z["atom_type"]).to(d), mask, info)
    assert sizes.tolist() == z["sizes"].tolist()
    assert np.array_equal(got.cpu().numpy(), z["order"])          # bit-exact vs the reference's matrices
    # a bigger random batch against the oracle (256 molecules, ragged, incl. single atoms)
    rng = np.random.default_rng(0)
    sz = rng.integers(1, 48, size=256)
    x = (rng.normal(size=(int(sz.sum()), 3)) * 2.2).astype(np.float32)
    t = rng.integers(0, 10, size=int(sz.sum()))
    want = chem_oracle.bond_orders_dense(x, t, sz, info, 47)
    mask = torch.repeat_interleave(torch.arange(256), torch.as_tensor(sz)).to(d)
    got, _ = bond_orders(torch.as_tensor(x).to(d), torch.as_tensor(t).to(d), mask, info, n_max=47)
    assert np.array_equal(got.cpu().numpy(), want)
    assert (want > 0).sum() > 100
    mols = build_molecules(torch.as_tensor(x).to(d), torch.as_tensor(t).to(d), mask, info, largest_frag=True)
    assert len(mols) == 256 and all(len(m.fragments()) == 1 for m in mols)


@pytest.mark.gpu
@pytest.mark.parametrize("arch,mode", [("small_cond", "pocket_conditioning"), ("small_joint", "joint")])
def test_generate_ligands_end_to_end(tmp_path, arch, mode):
    """PDB -> molecules on the HIP path; the tensor logic equals calling the sampler directly
    and moving the result back into the pocket frame (lightning_modules.py:841-846)."""
    from diffsbdd_amd.generate import LigandGenerator
    pdb = tmp_path / "c.pdb"
    pdb.write_text(_PDB)
    hp = _hp(arch, mode, "full-atom")
    gen = LigandGenerator(hp["dataset"], hp["egnn_params"], hp["diffusion_params"], mode, hp["node_histogram"],
                          pocket_representation="full-atom", device="cuda:0")
    cfg, _ = synthetic.arch_cfg(arch)
    gen.ddpm.dynamics.load_state_dict(synthetic.random_state_dict(cfg, seed=0))
    n = 5
    sizes = [4, 7, 3, 9, 6]
    gen.ddpm.seed(11)
    mols = gen.generate_ligands(str(pdb), n, ref_ligand="A:100", num_nodes_lig=torch.tensor(sizes),
                                timesteps=8, n_nodes_bias=1, n_nodes_min=5)
    want_sizes = [max(s + 1, 5) for s in sizes]
    assert [m.num_atoms for m in mols] == want_sizes
    assert all(np.isfinite(m.positions).all() for m in mols)
    assert all(s in gen.lig_type_decoder for m in mols for s in m.symbols)
    # same seed, sampler called directly
    residues = gen.select_pocket_residues(str(pdb), ref_ligand="A:100")
    pocket = gen.prepare_pocket(residues, repeats=n)
    com = pocket["x"][:14].mean(0)
    gen.ddpm.seed(11)
    xh, lm = gen.sample_for_pocket(pocket, n, torch.tensor(sizes), timesteps=8, n_nodes_bias=1, n_nodes_min=5)
    pos = np.concatenate([m.positions for m in mols])
    assert np.allclose(xh[:, :3].cpu().numpy(), pos, atol=1e-5)
    # generated atoms sit in the pocket's frame, not at the origin-centred sampling frame (random
    # weights keep the conditional chain bounded; the joint one wanders, so only checked there)
    if mode == "pocket_conditioning":
        assert float((xh[:, :3].mean(0) - com).norm()) < 15.0 and float(com.norm()) > 4.0
    # largest_frag never returns more atoms
    gen.ddpm.seed(11)
    frag = gen.generate_ligands(str(pdb), n, ref_ligand="A:100", num_nodes_lig=torch.tensor(sizes),
                                timesteps=8, n_nodes_bias=1, n_nodes_min=5, largest_frag=True)
    assert all(f.num_atoms <= m.num_atoms for f, m in zip(frag, mols))


@pytest.mark.gpu
def test_generate_for_several_pockets_in_one_batch(tmp_path):
    """Two different pockets packed into one batch give the molecules each pocket gets alone
    at the same global sample indices (block-diagonal graph + keyed noise)."""
    from diffsbdd_amd.generate import LigandGenerator
    pdb = tmp_path / "c.pdb"
    pdb.write_text(_PDB)
    hp = _hp("small_cond", "pocket_conditioning", "full-atom")
    gen = LigandGenerator(hp["dataset"], hp["egnn_params"], hp["diffusion_params"], hp["mode"],
                          hp["node_histogram"], pocket_representation="full-atom", device="cuda:0")
    cfg, _ = synthetic.arch_cfg("small_cond")
    gen.ddpm.dynamics.load_state_dict(synthetic.random_state_dict(cfg, seed=0))
    res_a = gen.select_pocket_residues(str(pdb), ref_ligand="A:100")          # 14 atoms
    res_b = gen.select_pocket_residues(str(pdb), pocket_ids=["A:2", "A:3"])   # 7 atoms
    jobs = [(res_a, 3, torch.tensor([5, 8, 6])), (res_b, 2, torch.tensor([7, 4]))]
    gen.ddpm.seed(5)
    packed = gen.generate_for_pockets(jobs, timesteps=6)
    assert [len(p) for p in packed] == [3, 2]
    assert [m.num_atoms for m in packed[0]] == [5, 8, 6] and [m.num_atoms for m in packed[1]] == [7, 4]
    gen.ddpm.seed(5, sample_offset=0)
    alone_a = gen.generate_for_pockets(jobs[:1], timesteps=6)[0]
    gen.ddpm.seed(5, sample_offset=3)
    alone_b = gen.generate_for_pockets(jobs[1:], timesteps=6)[0]
    for got, want in zip(packed[0] + packed[1], alone_a + alone_b):
        assert got.symbols == want.symbols
        assert np.allclose(got.positions, want.positions, atol=2e-4)


@pytest.mark.gpu
def test_sample_whose_atoms_are_all_virtual_is_an_empty_molecule_at_its_index():
    """With virtual nodes the atoms of the class 'Ne' are dropped before molecule building (lightning_modules.py:531-537);
    a sample that keeps no atom must still occupy its slot of the per-sample list (callers index molecules by sample)."""
    from diffsbdd_amd.generate import LigandGenerator
    from diffsbdd_amd.molecules import build_molecules
    keys = ("dataset", "egnn_params", "diffusion_params", "mode", "node_histogram", "pocket_representation", "virtual_nodes")
    hp = _hp("small_cond", "pocket_conditioning", "full-atom")
    hp["virtual_nodes"] = True
    gen = LigandGenerator(**{k: hp[k] for k in keys}, device="cuda")
    types = torch.tensor([0, 1, 10, 10, 10, 10, 2, 0])          # sample 1: virtual atoms only
    mask = torch.tensor([0, 0, 0, 1, 1, 1, 2, 2])
    xh = torch.zeros(8, 3 + 11, device="cuda")
    xh[:, :3] = torch.arange(8, dtype=torch.float32, device="cuda")[:, None] * 1.4
    xh[torch.arange(8), 3 + types] = 1.0
    x, at, m = gen._drop_virtual(xh, mask.cuda())
    mols = build_molecules(x, at, m, gen.dataset_info, batch=3)
    assert [mol.num_atoms for mol in mols] == [2, 0, 2]
    assert mols[0].symbols == ["C", "N"] or len(mols[0].symbols) == 2
