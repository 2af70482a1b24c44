"""Test-set driver: sustained sampling over many pockets.

The reference's `test.py` (/root/reference/test.py:59-176) walks the test pockets
one at a time: per pocket it calls `generate_ligands` with `batch_size` copies of
that one pocket until `n_samples` molecules pass its filters (at most MAXITER = 10
rounds, :14,101-104), writes `raw/<name>_gen.sdf`, `processed/<name>_gen.sdf` and
`pocket_times/<name>.txt`, and finally `pocket_times.txt` with the mean time per
pocket (:179-186).  A pocket that needs 100 molecules with batch_size 120 wastes
the rest of the batch; a pocket that is 3 molecules short costs a whole extra
chain.

Here the unit of scheduling is the SAMPLE SLOT, not the pocket.  The graph of a
batch is block diagonal per sample (dynamics.py:170-172) and the HIP path's
per-sample results do not depend on the batch composition, so one sampling batch
can carry slots of several different pockets:

  * `plan_batch` fills the `batch_size` slots from the queue of open requests,
    taking pockets of similar size together (the queue is kept sorted by node
    count, so the per-slot cost inside a batch is even and the workspace of the
    engine is re-bound rarely);
  * after every chain the driver counts the molecules that pass the filter and
    puts each pocket's deficit back into the queue -- the freed slots are refilled
    by the next pockets instead of idling (the reference would run a whole extra
    batch for that pocket);
  * every global sample gets a fixed index (pocket order x sample number), which
    keys the noise: the molecules of a pocket do not depend on which other pockets
    shared its batches nor on the number of GPUs;
  * across GPUs the pockets are distributed by estimated cost (longest-processing-
    time first over sum n_nodes^2 * n_samples); there is no communication on the
    data path, results are written per pocket.

Per-pocket time (the reference's `pocket_times`): a batch's wall time is attributed
to its pockets in proportion to the slots they used.
"""
from __future__ import annotations

import argparse
import os
import sys
import time
from dataclasses import dataclass, field
from typing import Callable, List, Optional

MAXITER = 10          # test.py:14


@dataclass
class PocketJob:
    """One test pocket.  `residues`: parsed pocket residues (pocket.read_pdb_residues
    selection); `n_nodes`: pocket nodes in the model's representation (scheduling
    weight); `num_nodes_lig`: fixed ligand size (test.py --fix_n_nodes) or None."""
    name: str
    residues: list
    n_nodes: int
    n_samples: int
    num_nodes_lig: Optional[int] = None
    index: int = 0                       # position in the job list: base of the global sample ids
    # bookkeeping
    valid: list = field(default_factory=list)
    raw: list = field(default_factory=list)
    n_generated: int = 0
    rounds: int = 0
    seconds: float = 0.0

    @property
    def deficit(self):
        return max(self.n_samples - len(self.valid), 0)


def plan_batch(queue: List[PocketJob], batch_size: int, oversample: float = 1.0, fill: bool = True):
    """Slots of the next batch: list of (job, n_slots), pockets of similar size together.

    `queue` is sorted by n_nodes.  Each open job asks for ceil(deficit * oversample)
    slots; jobs are taken in queue order until the batch is full, the last one is
    cut to what fits (its rest stays queued).  When the open requests do not fill the
    batch (the tail of the test set), `fill` hands the free slots to the planned jobs
    round-robin (at most 4x a job's request): spare samples are cheap in an under-full
    batch and spare a pocket another whole chain when some of its molecules are rejected."""
    plan, free = [], batch_size
    for job in queue:
        if free == 0:
            break
        want = job.deficit
        if want == 0:
            continue
        want = min(int(-(-want * oversample // 1)), free)
        plan.append([job, want])
        free -= want
    if fill and plan:
        cap = [4 * n for _, n in plan]
        while free > 0 and any(p[1] < c for p, c in zip(plan, cap)):
            for p, c in zip(plan, cap):
                if free > 0 and p[1] < c:
                    p[1] += 1
                    free -= 1
    return [(job, n) for job, n in plan]


def number_jobs(jobs: List[PocketJob]):
    for i, j in enumerate(jobs):
        j.index = i
    return jobs


def assign_to_ranks(jobs: List[PocketJob], world: int):
    """Longest-processing-time-first partition of the pockets over `world` ranks by the cost
    estimate n_samples * n_nodes^2 (edges of the complete graph).  Returns a list of job lists."""
    loads, parts = [0.0] * world, [[] for _ in range(world)]
    for job in sorted(jobs, key=lambda j: -(j.n_samples * (j.n_nodes + 30) ** 2)):
        r = loads.index(min(loads))
        parts[r].append(job)
        loads[r] += job.n_samples * (job.n_nodes + 30) ** 2
    return parts


class IterationLimit(RuntimeError):
    """A pocket used up its sample budget (test.py:101-104).  `jobs`: all jobs with what they collected so far."""

    def __init__(self, msg, jobs):
        super().__init__(msg)
        self.jobs = jobs


def default_is_valid(m):
    """The reference's default run (test.py:99-135 with sanitize=False, relax=False) accepts every molecule
    `process_molecule` returns: it only returns None when sanitisation or relaxation fails
    (analysis/molecule_builder.py:162-214).  So by default every built molecule counts."""
    return m is not None


def valence_filter(m):
    """`--sanitize` without RDKit: the valence / connectivity filter of molecules.is_valid_molecule on the molecule built
    from the GPU bond-order matrix (allowed valences: constants.py:19-22).  It APPROXIMATES what RDKit's sanitisation
    rejects for distance-table bonds (over-valent atoms); it does not kekulise or check aromaticity."""
    from .molecules import Molecule, is_valid_molecule
    return is_valid_molecule(m) if isinstance(m, Molecule) else m is not None


class TestSetDriver:
    """Runs `sample_batch(plan, batch_no) -> list of per-plan-entry molecule lists` until every job
    has `n_samples` molecules that pass `is_valid`.  Budget per pocket as in the reference (test.py:101-104: at most
    MAXITER rounds of `batch_size` samples each): a job may GENERATE max_rounds * batch_size samples -- counted in
    samples, because a round here hands a pocket only its deficit, not a whole batch."""

    __test__ = False    # not a pytest class

    def __init__(self, sample_batch: Callable, batch_size: int, is_valid: Callable = None,
                 oversample: float = 1.0, max_rounds: int = MAXITER, clock: Callable = time.perf_counter):
        self.sample_batch, self.batch_size = sample_batch, batch_size
        self.is_valid = is_valid or default_is_valid
        self.oversample, self.max_rounds, self.clock = oversample, max_rounds, clock
        self.batches = []          # [(wall seconds, [(job name, slots)])]

    def run(self, jobs: List[PocketJob]):
        """`job.index` (the base of a job's global sample ids) must be set by the caller -- it is the
        job's position in the WHOLE test set, not in this rank's share (number_jobs)."""
        queue = sorted(jobs, key=lambda j: (j.n_nodes, j.index))
        while True:
            open_jobs = [j for j in queue if j.deficit > 0]
            for j in open_jobs:
                if j.n_generated >= self.max_rounds * self.batch_size:        # test.py:101-104, in samples
                    for k in jobs:
                        k.valid = k.valid[:k.n_samples]
                    raise IterationLimit(f"{j.name}: maximum number of iterations has been exceeded "
                                         f"({j.n_generated} samples generated, {len(j.valid)} of {j.n_samples} accepted)", jobs)
            if not open_jobs:
                break
            # ask for more than the deficit when the filter rejects a share of the molecules: the observed pass rate
            # of the jobs in the queue scales the request (at least `oversample`, at most 4x)
            gen_n = sum(j.n_generated for j in open_jobs)
            rate = sum(len(j.valid) for j in open_jobs) / gen_n if gen_n else 1.0
            over = max(self.oversample, min(4.0, 1.0 / max(rate, 0.25))) if gen_n else self.oversample
            plan = plan_batch(open_jobs, self.batch_size, over)
            t0 = self.clock()
            results = self.sample_batch(plan, len(self.batches))
            dt = self.clock() - t0
            slots = sum(n for _, n in plan)
            for (job, n), mols in zip(plan, results):
                assert len(mols) == n, (job.name, len(mols), n)
                job.n_generated += n
                job.rounds += 1
                job.seconds += dt * n / slots
                job.raw.extend(mols)
                job.valid.extend(m for m in mols if self.is_valid(m))
            self.batches.append((dt, [(job.name, n) for job, n in plan]))
        for j in jobs:
            j.valid = j.valid[:j.n_samples]                           # test.py:136
        return jobs

    # ---- the reference's output files (test.py:139-186) ---------------------------------------
    @staticmethod
    def write_outputs(jobs, outdir, write_sdf, summary=None):
        """`summary`: [(name, seconds)] of ALL ranks' pockets for `pocket_times.txt` (the rank that passes it writes
        the file; None = this call's jobs, the single-process case)."""
        for sub in ("raw", "processed", "pocket_times"):
            os.makedirs(os.path.join(outdir, sub), exist_ok=True)
        for j in jobs:
            keep = set(id(m) for m in j.valid)
            ordered = [m for m in j.raw if id(m) in keep] + [m for m in j.raw if id(m) not in keep]
            write_sdf(os.path.join(outdir, "raw", f"{j.name}_gen.sdf"), ordered)
            write_sdf(os.path.join(outdir, "processed", f"{j.name}_gen.sdf"), j.valid)
            with open(os.path.join(outdir, "pocket_times", f"{j.name}.txt"), "w") as f:
                f.write(f"{j.name} {j.seconds}")
        if summary is None:
            summary = [(j.name, j.seconds) for j in jobs]
        if summary is not False:
            with open(os.path.join(outdir, "pocket_times.txt"), "w") as f:
                for name, sec in summary:
                    f.write(f"{name} {sec}\n")


def cone_mode_for_jobs(jobs: List[PocketJob]) -> int:
    """Forward-cone mode of a WHOLE test-set run (one value for every batch and every rank, so that molecules do not
    depend on the packing or the sharding): on (2) when the jobs ask for at least 5 samples per pocket on average --
    the measured break-even of the canonical-pocket network is one distinct pocket per 5 samples of a batch
    (profiles/r4n_cone_rule.md) and a batch packs jobs in proportion to their requests; the reference's runs
    (n_samples = 100, test.py:75) are far above it.  Decide on the complete job list, before assign_to_ranks."""
    n = sum(j.n_samples for j in jobs)
    return 2 if n >= 5 * max(len(jobs), 1) else 0


def make_hip_sampler(gen, timesteps=None, seed=0, largest_frag=True, n_nodes_bias=0, n_nodes_min=0, cone_mode=2, **kwargs):
    """`sample_batch` for TestSetDriver on a `generate.LigandGenerator`: one packed sampling batch on
    the GPU (LigandGenerator.generate_for_pockets).  Global sample id of slot k of a job's r-th round =
    index * 2^20 + (samples generated for the job so far) + k: independent of the packing.
    cone_mode: the engine's forward cone for EVERY batch of the run (2 = on, the default; 0 = off;
    `cone_mode_for_jobs(all jobs)` picks by the requested samples per pocket)."""
    import torch
    if cone_mode not in (0, 2):
        raise ValueError("cone_mode must be 0 or 2 (a per-batch choice would make molecules depend on the packing)")

    def ligand_sizes(job, ids):
        """Ligand sizes of the slots `ids` of one job: fixed, or drawn from p(n_lig | n_pocket)
        (DistributionNodes.sample_conditional, en_diffusion.py:993-1003) with a generator keyed by
        (seed, global sample id), so that a sample's size does not depend on the packing either."""
        if job.num_nodes_lig is not None:
            return torch.full((len(ids),), job.num_nodes_lig, dtype=torch.int64)
        prob = gen.ddpm.size_distribution.prob[:, job.n_nodes]
        out = []
        for g in ids.tolist():
            rng = torch.Generator().manual_seed((int(seed) * 1000003 + int(g)) % (2 ** 63 - 1))
            out.append(int(torch.multinomial(prob, 1, generator=rng)))
        return torch.tensor(out, dtype=torch.int64)

    def sample_batch(plan, batch_no):
        ids = [job.index * (1 << 20) + job.n_generated + torch.arange(n) for job, n in plan]
        jobs = [(job.residues, n, ligand_sizes(job, i)) for (job, n), i in zip(plan, ids)]
        # bit-identical molecules for any packing: the engine's forward cone must not switch with the number of
        # distinct pockets in a batch (en_diffusion.cone_mode) -- pinned to the run's mode for this batch only, the
        # generator's own setting is restored afterwards
        saved, saved_g = gen.ddpm.cone_mode, gen.ddpm.edge_granule16
        gen.ddpm.cone_mode = cone_mode
        if saved_g == "auto":
            gen.ddpm.edge_granule16 = 0        # (the automatic choice looks at the batch: pinned for the same reason)
        try:
            return gen.generate_for_pockets(jobs, timesteps=timesteps, largest_frag=largest_frag,
                                            n_nodes_bias=n_nodes_bias, n_nodes_min=n_nodes_min, seed=seed,
                                            sample_ids=torch.cat(ids), **kwargs)
        finally:
            gen.ddpm.cone_mode, gen.ddpm.edge_granule16 = saved, saved_g

    return sample_batch


def jobs_from_test_dir(gen, test_dir, n_samples, test_list=None, fix_n_nodes=False):
    """The reference's test-set layout (test.py:52-70,86-94): `<pdb>_<pocket>*.sdf` reference ligands next
    to `<pdb>.pdb` and `<ligand name>.txt` (pocket residue ids)."""
    from . import pocket as pocket_io
    jobs = []
    names = sorted(f[:-4] for f in os.listdir(test_dir) if f.endswith(".sdf") and not f.startswith("."))
    if test_list is not None:
        names = [n for n in names if n in set(test_list)]
    for name in names:
        pdb_name = name.split("_")[0]
        pdb_file = os.path.join(test_dir, pdb_name + ".pdb")
        with open(os.path.join(test_dir, name + ".txt")) as f:
            resi_list = f.read().split()
        residues = gen.select_pocket_residues(pdb_file, pocket_ids=resi_list)
        coords, _, _ = pocket_io.featurize_pocket(
            residues, "CA" if gen.pocket_representation == "CA" else "full-atom",
            atom_encoder=None if gen.pocket_representation == "CA" else gen.pocket_type_encoder)
        n_lig = len(pocket_io.read_sdf_coords(os.path.join(test_dir, name + ".sdf"))) if fix_n_nodes else None
        jobs.append(PocketJob(name, residues, len(coords), n_samples, n_lig))
    return jobs


def main(argv=None):
    """`python -m diffsbdd_amd.testset <checkpoint> --test_dir ... --outdir ...` -- the options of the
    reference's test.py (:18-36).  Without `--sanitize` every built molecule is accepted, as in the reference's default
    run; `--sanitize` applies `valence_filter` (an approximation of RDKit's sanitisation, which is absent here) and the
    driver refills the rejected slots; `--relax` (UFF, RDKit) is refused."""
    import torch

    from . import sharding
    from .generate import LigandGenerator
    from .molecules import write_sdf
    ap = argparse.ArgumentParser(description=main.__doc__)
    ap.add_argument("checkpoint")
    ap.add_argument("--test_dir", required=True)
    ap.add_argument("--test_list", default=None)
    ap.add_argument("--outdir", required=True)
    ap.add_argument("--n_samples", type=int, default=100)
    ap.add_argument("--all_frags", action="store_true")
    ap.add_argument("--sanitize", action="store_true")
    ap.add_argument("--relax", action="store_true")
    ap.add_argument("--batch_size", type=int, default=120)
    ap.add_argument("--resamplings", type=int, default=10)
    ap.add_argument("--jump_length", type=int, default=1)
    ap.add_argument("--timesteps", type=int, default=None)
    ap.add_argument("--fix_n_nodes", action="store_true")
    ap.add_argument("--n_nodes_bias", type=int, default=0)
    ap.add_argument("--n_nodes_min", type=int, default=0)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--trusted-checkpoint", action="store_true")
    a = ap.parse_args(argv)
    from .molecules import PROCESS_MOLECULE_COVERAGE
    print("[testset] " + PROCESS_MOLECULE_COVERAGE, file=sys.stderr)
    if a.relax:
        raise SystemExit("--relax (UFF relaxation) is an RDKit operation (see generate.py)")
    if a.sanitize:
        print("[testset] --sanitize: RDKit is absent; using the valence / connectivity filter on the distance-table "
              "bonds (testset.valence_filter), which approximates RDKit's sanitisation", file=sys.stderr)
    rank, local_rank, world = sharding.init_distributed()
    torch.cuda.set_device(local_rank)
    gen = LigandGenerator.from_checkpoint(a.checkpoint, device=f"cuda:{local_rank}", trusted=a.trusted_checkpoint)
    test_list = None
    if a.test_list:
        with open(a.test_list) as f:
            test_list = f.read().split(",")
    jobs = jobs_from_test_dir(gen, a.test_dir, a.n_samples, test_list, a.fix_n_nodes)
    mine = assign_to_ranks(number_jobs(jobs), world)[rank]
    extra = dict(resamplings=a.resamplings, jump_length=a.jump_length) if gen.mode == "joint" else {}
    driver = TestSetDriver(make_hip_sampler(gen, a.timesteps, a.seed, largest_frag=not a.all_frags,
                                            n_nodes_bias=a.n_nodes_bias, n_nodes_min=a.n_nodes_min,
                                            cone_mode=cone_mode_for_jobs(jobs), **extra),
                           a.batch_size, is_valid=valence_filter if a.sanitize else None)
    failed = None
    try:
        driver.run(mine)
    except IterationLimit as exc:          # keep what was collected (the reference loses the run here)
        failed = exc
    # pocket_times.txt lists every pocket of every rank (test.py:179-186): gathered to rank 0
    times = [(j.name, j.seconds) for j in mine]
    if world > 1:
        gathered = [None] * world
        torch.distributed.all_gather_object(gathered, times)
        times = [t for part in gathered for t in part]
    order = {j.name: k for k, j in enumerate(jobs)}
    times.sort(key=lambda t: order.get(t[0], 0))
    TestSetDriver.write_outputs(mine, a.outdir, write_sdf, summary=times if rank == 0 else False)
    if rank == 0:
        with open(os.path.join(a.outdir, "process_molecule.txt"), "w") as f:
            f.write(PROCESS_MOLECULE_COVERAGE + "\n")
    if rank == 0 and times:
        secs = [t[1] for t in times]
        mean = sum(secs) / len(secs)
        std = (sum((s - mean) ** 2 for s in secs) / len(secs)) ** 0.5
        print(f"Time per pocket: {mean:.3f} \\pm {std:.2f}")
    if failed is not None:
        raise SystemExit(f"[rank {rank}] {failed}")


if __name__ == "__main__":
    main()
