"""Test-set driver (/root/reference/test.py:59-176): the slot scheduler on the CPU with
a stand-in sampler, and the real packed sampling on the GPU (packing independence, bitwise)."""
import os

import numpy as np
import pytest
import torch

from diffsbdd_amd import testset as ts


def fake_jobs(sizes, n_samples):
    return ts.number_jobs([ts.PocketJob(f"p{i}", [], n, n_samples) for i, n in enumerate(sizes)])


def test_scheduler_fills_slots_refills_deficits_and_accounts_time():
    """Slots of a batch are shared by pockets of similar size; molecules rejected by the filter come
    back as a deficit that is refilled next to new pockets; time is split by slots."""
    jobs = fake_jobs([300, 36, 280, 40, 310, 290, 35], n_samples=10)
    calls = []
    tick = iter(range(1000))

    def sample_batch(plan, batch_no):
        calls.append([(job.name, n) for job, n in plan])
        # deterministic "molecules": (job index, running sample number); every 4th one is invalid
        return [[(job.index, job.n_generated + k) for k in range(n)] for job, n in plan]

    drv = ts.TestSetDriver(sample_batch, batch_size=16, is_valid=lambda m: m[1] % 4 != 3,
                           clock=lambda: float(next(tick)))
    done = drv.run(jobs)
    assert all(len(j.valid) == 10 for j in done)
    assert all(sum(n for _, n in c) <= 16 for c in calls)
    assert all(sum(n for _, n in c) == 16 for c in calls[:-1])        # full batches (tail slots are handed out as spares)
    # pockets of similar size share batches: the small pockets (35 / 36 / 40 nodes) come first, together
    assert [n for n, _ in calls[0]] == ["p6", "p1"] and [n for n, _ in calls[1]] == ["p6", "p1", "p3"]
    # a molecule is never sampled twice and never lost
    for j in done:
        assert [m[1] for m in j.raw] == list(range(j.n_generated))
        assert j.valid == [m for m in j.raw if m[1] % 4 != 3][:10]
    # fewer chains than pocket-at-a-time sampling (the reference: ceil-rounds per pocket, one batch each)
    assert len(calls) < len(jobs)
    # every batch took 1 tick; the per-pocket times add up to the total
    assert abs(sum(j.seconds for j in done) - len(calls)) < 1e-9


def test_scheduler_gives_up_like_the_reference():
    jobs = fake_jobs([50], n_samples=4)
    drv = ts.TestSetDriver(lambda plan, b: [[None] * n for _, n in plan], batch_size=8, max_rounds=3)
    with pytest.raises(RuntimeError, match="maximum number of iterations"):
        drv.run(jobs)


def test_budget_counts_samples_and_oversampling_follows_the_pass_rate():
    """The reference gives a pocket MAXITER rounds of batch_size samples (test.py:101-104).  Here a round hands a
    pocket its deficit only, so the budget is counted in SAMPLES: with a filter that passes one molecule in five a
    pocket still finishes (its deficit shrinks geometrically), the request is scaled by the observed pass rate,
    and when the budget is exhausted the collected molecules survive in the exception."""
    jobs = fake_jobs([50, 60], n_samples=20)
    sizes = []

    def sample_batch(plan, batch_no):
        sizes.append(sum(n for _, n in plan))
        return [[(job.index, job.n_generated + k) for k in range(n)] for job, n in plan]

    drv = ts.TestSetDriver(sample_batch, batch_size=32, is_valid=lambda m: m[1] % 5 == 0, max_rounds=10)
    done = drv.run(jobs)
    assert all(len(j.valid) == 20 for j in done)
    assert all(j.n_generated <= 10 * 32 for j in done) and max(j.rounds for j in done) > 3
    # a filter nothing passes: the budget of max_rounds * batch_size samples per pocket ends the run, results are kept
    jobs = fake_jobs([50], n_samples=4)
    drv = ts.TestSetDriver(lambda plan, b: [[(0, k) for k in range(n)] for _, n in plan], batch_size=8,
                           is_valid=lambda m: False, max_rounds=3)
    with pytest.raises(ts.IterationLimit) as exc:
        drv.run(jobs)
    assert exc.value.jobs[0].n_generated >= 24 and len(exc.value.jobs[0].raw) == exc.value.jobs[0].n_generated


def test_default_accepts_every_molecule_and_sanitize_filter_rejects_overvalent_ones():
    """The default acceptance is the reference's (test.py:99-135 with sanitize=False: every molecule process_molecule
    returns); `--sanitize` maps to the valence / connectivity filter on molecules built from the bond-order matrix
    (molecules.is_valid_molecule; allowed valences = the reference's constants.py:19-22)."""
    from diffsbdd_amd.molecules import Molecule, is_valid_molecule
    pos = np.zeros((5, 3), dtype=np.float32)
    ok = Molecule(pos, ["C", "O", "N", "C", "F"], [(1, 0, 2), (2, 0, 1), (3, 2, 1), (4, 3, 1)])
    assert ok.valences() == [3, 2, 2, 2, 1] and not ok.valence_violations() and ts.valence_filter(ok)
    bad = Molecule(pos, ["C", "O", "N", "C", "F"], [(1, 0, 2), (2, 0, 1), (3, 1, 1), (4, 3, 1)])   # O with 3 bonds
    assert bad.valence_violations() == [1] and not ts.valence_filter(bad) and ts.default_is_valid(bad)
    assert not ts.default_is_valid(None) and ts.default_is_valid("anything else that exists")
    assert not ts.valence_filter(None) and ts.valence_filter("anything else that exists")
    assert ts.TestSetDriver(lambda plan, b: [], 4).is_valid is ts.default_is_valid
    two = Molecule(pos, ["C", "C", "C", "C", "C"], [(1, 0, 1), (2, 1, 1), (4, 3, 1)])
    frag = two.largest_fragment()
    assert frag.num_atoms == 3 and frag.n_generated_atoms == 5
    assert is_valid_molecule(frag) and not is_valid_molecule(frag, min_fragment_fraction=0.8)
    assert is_valid_molecule(two, max_fragments=2) and not is_valid_molecule(two, max_fragments=1)


def test_rank_partition_balances_cost_and_outputs_follow_the_reference_layout(tmp_path):
    jobs = fake_jobs([300, 36, 280, 40, 310, 290, 35, 305, 295], n_samples=10)
    parts = ts.assign_to_ranks(jobs, 2)
    assert sorted(j.name for p in parts for j in p) == sorted(j.name for j in jobs)
    cost = [sum(j.n_samples * (j.n_nodes + 30) ** 2 for j in p) for p in parts]
    assert max(cost) / min(cost) < 1.25
    assert [j.index for j in jobs] == list(range(9))                   # global ids survive the partition
    drv = ts.TestSetDriver(lambda plan, b: [[f"{job.name}:{job.n_generated + k}" for k in range(n)]
                                            for job, n in plan], batch_size=16)
    done = drv.run(parts[0])
    written = {}
    ts.TestSetDriver.write_outputs(done, str(tmp_path), lambda path, mols: written.__setitem__(path, list(mols)))
    for j in done:
        assert len(written[os.path.join(str(tmp_path), "processed", f"{j.name}_gen.sdf")]) == 10
        assert os.path.isfile(tmp_path / "pocket_times" / f"{j.name}.txt")
    lines = (tmp_path / "pocket_times.txt").read_text().splitlines()
    assert [l.split()[0] for l in lines] == [j.name for j in done]
    # several ranks: the summary of ALL pockets is written by the rank that is handed it, the others write none
    all_times = [(j.name, 1.5) for j in jobs]
    os.remove(tmp_path / "pocket_times.txt")
    ts.TestSetDriver.write_outputs(done, str(tmp_path), lambda path, mols: None, summary=False)
    assert not os.path.exists(tmp_path / "pocket_times.txt")
    ts.TestSetDriver.write_outputs(done, str(tmp_path), lambda path, mols: None, summary=all_times)
    assert len((tmp_path / "pocket_times.txt").read_text().splitlines()) == len(jobs)


@pytest.mark.gpu
def test_packed_sampling_is_independent_of_the_packing():
    """Two pockets (3rfm, 5ndu; C-alpha model), 6 molecules each, 4 reverse steps: the molecules of a
    pocket are identical -- coordinates bit for bit -- whether the driver packs 12, 7 or 4 slots per
    batch, because noise and ligand sizes are keyed by the global sample id."""
    from oracle import weights as W
    from diffsbdd_amd import pocket as pk
    from diffsbdd_amd.generate import LigandGenerator
    from tests._golden import GOLDEN_DIR
    cfg, dd = W.arch_cfg("crossdock_ca_cond")
    egnn = dict(joint_nf=cfg["joint_nf"], hidden_nf=cfg["hidden_nf"], n_layers=cfg["n_layers"], attention=True,
                tanh=True, norm_constant=1, inv_sublayers=1, sin_embedding=False, normalization_factor=100,
                aggregation_method="sum", edge_cutoff_ligand=None, edge_cutoff_pocket=5.0,
                edge_cutoff_interaction=5.0, reflection_equivariant=False, edge_embedding_dim=None)
    diff = dict(diffusion_steps=500, diffusion_noise_schedule="polynomial_2", diffusion_noise_precision=5e-4,
                diffusion_loss_type="l2", normalize_factors=[1, 1])
    gen = LigandGenerator("crossdock", egnn, diff, "pocket_conditioning", np.ones((40, 400)), "CA", device="cuda:0")
    gen.ddpm.dynamics.load_state_dict(W.random_state_dict(cfg, 0))
    residues = {}
    for name in ("3rfm", "5ndu"):
        z = np.load(os.path.join(GOLDEN_DIR, f"pocket_{name}.npz"))
        # rebuild residue records from the fixture: one CA atom per residue
        inv = {v: k for k, v in pk.AA_ENCODER.items()}
        one_to_three = {v: k for k, v in pk.AA3_TO_1.items()}
        residues[name] = [dict(chain="A", resseq=i, icode=" ", resname=one_to_three[inv[int(t)]],
                               atoms=[("CA", "C", tuple(map(float, xyz)))], hetero=False)
                          for i, (xyz, t) in enumerate(zip(z["ca_x"], z["ca_types"]))]

    def run(batch_size):
        jobs = ts.number_jobs([ts.PocketJob(n, residues[n], len(residues[n]), 6) for n in ("3rfm", "5ndu")])
        drv = ts.TestSetDriver(ts.make_hip_sampler(gen, timesteps=4, seed=5, largest_frag=False), batch_size,
                               is_valid=lambda m: m is not None)     # (the packing is the subject, not the filter)
        return {j.name: j.valid for j in drv.run(jobs)}, len(drv.batches)

    a, na = run(12)
    b, nb = run(7)
    c, nc = run(4)
    assert (na, nb, nc) == (1, 2, 3)
    for name in a:
        for x, y, z_ in zip(a[name], b[name], c[name]):
            assert x.symbols == y.symbols == z_.symbols
            assert np.array_equal(x.positions, y.positions) and np.array_equal(x.positions, z_.positions)


@pytest.mark.gpu
def test_packed_sampling_full_atom_pockets_is_independent_of_the_packing():
    """The same with FULL-ATOM pockets (3rfm 286 atoms, 5ndu 287 atoms): here the chain hands the engine a pocket frame
    with one representative per distinct pocket of the batch and the ligand-output-only calls run the forward /
    backward cones (csrc/engine.hip).  A pocket's molecules are bit-identical whether it shares its batches with the
    other pocket (12 or 7 slots) or not (4 slots: mostly one pocket per batch)."""
    from oracle import weights as W
    from diffsbdd_amd import pocket as pk
    from diffsbdd_amd.generate import LigandGenerator
    from tests._golden import GOLDEN_DIR
    cfg, dd = W.arch_cfg("crossdock_fullatom_cond")
    egnn = dict(joint_nf=cfg["joint_nf"], hidden_nf=cfg["hidden_nf"], n_layers=cfg["n_layers"], attention=True,
                tanh=True, norm_constant=1, inv_sublayers=1, sin_embedding=False, normalization_factor=100,
                aggregation_method="sum", edge_cutoff_ligand=None, edge_cutoff_pocket=5.0,
                edge_cutoff_interaction=5.0, reflection_equivariant=False, edge_embedding_dim=None)
    diff = dict(diffusion_steps=500, diffusion_noise_schedule="polynomial_2", diffusion_noise_precision=5e-4,
                diffusion_loss_type="l2", normalize_factors=[1, 4])
    gen = LigandGenerator("crossdock", egnn, diff, "pocket_conditioning", np.ones((40, 400)), "full-atom",
                          device="cuda:0")
    gen.ddpm.dynamics.load_state_dict(W.random_state_dict(cfg, 0))
    elem = {v: k for k, v in pk.ATOM_ENCODER.items()}
    residues = {}
    for name in ("3rfm", "5ndu"):
        z = np.load(os.path.join(GOLDEN_DIR, f"pocket_{name}.npz"))
        # one pseudo-residue per atom: the featuriser only looks at (element, xyz) in full-atom mode
        residues[name] = [dict(chain="A", resseq=i, icode=" ", resname="GLY",
                               atoms=[("X", elem[int(t)], tuple(map(float, xyz)))], hetero=False)
                          for i, (xyz, t) in enumerate(zip(z["fa_x"], z["fa_types"]))]

    def run(batch_size):
        jobs = ts.number_jobs([ts.PocketJob(n, residues[n], len(residues[n]), 6) for n in ("3rfm", "5ndu")])
        drv = ts.TestSetDriver(ts.make_hip_sampler(gen, timesteps=4, seed=5, largest_frag=False, n_nodes_min=2), batch_size,
                               is_valid=lambda m: m is not None)
        out = {j.name: j.valid for j in drv.run(jobs)}
        return out, len(drv.batches), gen.ddpm.dynamics.engine().last_plan()

    a, na, plan = run(12)
    assert plan[0] == [1, 2, 3, 3, 2, 1] and plan[1] == [1, 1, 1, 0, 0, 0]      # frame + cones were on
    b, nb, _ = run(7)
    c, nc, _ = run(4)
    assert (na, nb, nc) == (1, 2, 3)
    for name in a:
        for x, y, z_ in zip(a[name], b[name], c[name]):
            assert x.symbols == y.symbols == z_.symbols
            assert np.array_equal(x.positions, y.positions) and np.array_equal(x.positions, z_.positions)


class _KeyedNoise:
    """The oracle's noise source for a chain whose HIP counterpart draws keyed noise (dsbdd_randn_keyed): draw i of the
    chain for the ligand rows of the samples with the given GLOBAL ids -- the same values whatever batch they sit in."""

    def __init__(self, seed, sample_ids, lig_mask, dev):
        import torch
        self.seed, self.i, self.dev = int(seed), 0, dev
        self.ids = torch.as_tensor(sample_ids, dtype=torch.int64, device=dev).contiguous()
        self.mask = lig_mask.to(device=dev, dtype=torch.int64).contiguous()

    def __call__(self, shape):
        import ctypes as C
        import torch
        from diffsbdd_amd import _lib
        n, cols = shape
        assert n == self.mask.numel()
        out = torch.empty((n, cols), dtype=torch.float32, device=self.dev)
        _lib.check(_lib.load().dsbdd_randn_keyed(torch.cuda.current_stream(self.dev).cuda_stream, out.data_ptr(),
                                                 self.mask.data_ptr(), n, cols, self.ids.numel(), 0, self.ids.data_ptr(),
                                                 C.c_uint64(self.seed), C.c_uint64(self.i), 0), "dsbdd_randn_keyed")
        self.i += 1
        return out.cpu()


@pytest.mark.gpu
def test_driver_molecules_vs_oracle_chains():
    """SURVEY.md 8f-4 against the ORACLE (not packing against packing): the test-set driver (two C-alpha pockets, 3
    molecules each, 4 slots per batch -> the pockets share batches and one pocket is split over two, keyed noise, T = 4)
    must return, per pocket, the molecules of `oracle.ddpm_oracle.cond_sample_given_pocket` run on that pocket alone
    with the noise of the same global sample ids: coordinates (moved back into the pocket's frame like
    lightning_modules.py:843-852) to 1e-3 free-running over the chain, identical atom types.  The reference's driver
    (test.py:99-135) is this chain per pocket + process_molecule."""
    import torch
    from oracle import ddpm_oracle as do
    from oracle import egnn_oracle as eo
    from oracle import weights as W
    from diffsbdd_amd import pocket as pk
    from diffsbdd_amd.generate import LigandGenerator
    from tests._golden import GOLDEN_DIR
    cfg, dd = W.arch_cfg("crossdock_ca_cond")
    egnn = dict(joint_nf=cfg["joint_nf"], hidden_nf=cfg["hidden_nf"], n_layers=cfg["n_layers"], attention=True,
                tanh=True, norm_constant=1, inv_sublayers=1, sin_embedding=False, normalization_factor=100,
                aggregation_method="sum", edge_cutoff_ligand=None, edge_cutoff_pocket=5.0,
                edge_cutoff_interaction=5.0, reflection_equivariant=False, edge_embedding_dim=None)
    diff = dict(diffusion_steps=500, diffusion_noise_schedule="polynomial_2", diffusion_noise_precision=5e-4,
                diffusion_loss_type="l2", normalize_factors=[1, 1])
    gen = LigandGenerator("crossdock", egnn, diff, "pocket_conditioning", np.ones((40, 400)), "CA", device="cuda:0")
    sd = W.random_state_dict(cfg, 0)
    gen.ddpm.dynamics.load_state_dict(sd)
    inv = {v: k for k, v in pk.AA_ENCODER.items()}
    one_to_three = {v: k for k, v in pk.AA3_TO_1.items()}
    residues = {}
    for name in ("3rfm", "5ndu"):
        z = np.load(os.path.join(GOLDEN_DIR, f"pocket_{name}.npz"))
        residues[name] = [dict(chain="A", resseq=i, icode=" ", resname=one_to_three[inv[int(t)]],
                               atoms=[("CA", "C", tuple(map(float, xyz)))], hetero=False)
                          for i, (xyz, t) in enumerate(zip(z["ca_x"], z["ca_types"]))]
    n_lig, n_samples, T, seed = 9, 3, 4, 5
    jobs = ts.number_jobs([ts.PocketJob(n, residues[n], len(residues[n]), n_samples, num_nodes_lig=n_lig)
                           for n in ("3rfm", "5ndu")])
    drv = ts.TestSetDriver(ts.make_hip_sampler(gen, timesteps=T, seed=seed, largest_frag=False), batch_size=4)
    done = drv.run(jobs)
    assert len(drv.batches) >= 2 and any(len(plan) == 2 for _, plan in drv.batches)     # shared and split batches
    om = do.OracleModel(sd, cfg, cfg["atom_nf"], cfg["residue_nf"], dd["timesteps"], dd["noise_schedule"],
                        dd["noise_precision"], norm_values=dd["norm_values"], conditional=True)
    dec = gen.dataset_info["atom_decoder"]
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    for job in done:
        assert len(job.valid) == n_samples
        pocket = {k: v.cpu().clone() for k, v in gen.prepare_pocket(job.residues, repeats=n_samples).items()}
        sizes = torch.full((n_samples,), n_lig, dtype=torch.int64)
        lig_mask = torch.repeat_interleave(torch.arange(n_samples), sizes)
        ids = job.index * (1 << 20) + torch.arange(n_samples)
        com0 = eo.segment_mean(pocket["x"].float(), pocket["mask"], n_samples)
        noise = _KeyedNoise(seed, ids, lig_mask, torch.device("cuda:0"))
        o_l, o_p, _, p_mask = do.cond_sample_given_pocket(om, pocket, sizes, noise, timesteps=T)
        shift = com0 - eo.segment_mean(o_p[:, :3], p_mask, n_samples)                  # lightning_modules.py:843-852
        x_ref = (o_l[:, :3] + shift[lig_mask]).numpy()
        t_ref = o_l[:, 3:].argmax(1).numpy()
        for k, mol in enumerate(job.valid):
            sl = slice(k * n_lig, (k + 1) * n_lig)
            assert mol.symbols == [dec[int(t)] for t in t_ref[sl]], (job.name, k)
            err = np.abs(mol.positions - x_ref[sl]).max()
            assert err < 1e-3, (job.name, k, err)
