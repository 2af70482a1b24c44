"""Pocket sharding across the GPUs of one node.

Samples of a sampling batch are independent diffusion chains (the graph is
block-diagonal per sample: /root/reference/equivariant_diffusion/dynamics.py:170-172),
so the data-parallel scheme has NO collective on the data path:

  * one process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on
    ROCm, "gloo" on CPU for tests);
  * rank r owns the contiguous global sample range shard_range(n, W, r);
  * noise is keyed by the GLOBAL sample index (dsbdd_randn_keyed), so a chain
    does not depend on W;
  * one exchange at the very end: all_gather of the per-rank row counts, then a
    padded all_gather of the finished ligands and of their global sample ids
    (KB-scale: latency bound).

The reference has no multi-GPU sampling path at all.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def shard_range(n_total: int, world: int, rank: int):
    """Contiguous block partition, ceil(n/W) per rank (the last ranks may get
    fewer / zero samples)."""
    per = (n_total + world - 1) // world
    lo = min(rank * per, n_total)
    hi = min(lo + per, n_total)
    return lo, hi


def init_distributed(backend=None, one_gpu_per_rank=True, force=False):
    """Initialise torch.distributed from the torchrun environment
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT).  Returns
    (rank, local_rank, world).  A single process (no env) is world 1 and needs no process group; `force` creates one
    anyway (a one-rank RCCL communicator: the de-risking test of the multi-GPU path on a one-GPU box)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    if (world > 1 or force) and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        # dmabuf IPC is the only mode the host driver supports (see task notes)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl" and one_gpu_per_rank:
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local_rank, world


def gather_ligands(out_lig: torch.Tensor, lig_mask: torch.Tensor, sample_lo: int, group=None, force_collective=False):
    """Gather the finished ligands of every rank.

    out_lig  [n_rows, D] this rank's ligand atoms (x | one-hot), lig_mask [n_rows]
    LOCAL sample ids; sample_lo = global index of this rank's first sample.
    Returns (all_lig [sum rows, D], all_mask with GLOBAL sample ids) on every
    rank (all_gather; the payload is ~1 KB per molecule).  force_collective: run the all_gather exchange at world
    size 1 as well (tests)."""
    if not dist.is_initialized() or (dist.get_world_size(group) == 1 and not force_collective):
        return out_lig, lig_mask + sample_lo
    world = dist.get_world_size(group)
    # RCCL ("nccl") moves device tensors; gloo (CPU tests, or several ranks sharing one GPU)
    # needs host tensors
    out_dev = out_lig.device
    dev = out_dev if dist.get_backend(group) == "nccl" else torch.device("cpu")
    out_lig, lig_mask = out_lig.to(dev), lig_mask.to(dev)
    # row count and feature width of every rank (a rank with an empty shard does not know D): one small all_gather,
    # read back with ONE host sync
    shape = torch.tensor([out_lig.shape[0], out_lig.shape[1] if out_lig.dim() == 2 else 0],
                         dtype=torch.int64, device=dev)
    shapes = torch.zeros((world, 2), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(shapes, shape, group=group) if dist.get_backend(group) == "nccl" else \
        dist.all_gather(list(shapes.unbind(0)), shape, group=group)
    table = shapes.tolist()
    counts = [int(c[0]) for c in table]
    D = max(int(c[1]) for c in table)
    max_rows = max(max(counts), 1)
    # padded payloads in their own types (fp32 rows as they are, int64 global sample ids): no float64 detour
    dt = out_lig.dtype if out_lig.dim() == 2 and out_lig.shape[1] == D and out_lig.is_floating_point() else torch.float32
    rows = torch.zeros((max_rows, D), dtype=dt, device=dev)
    ids = torch.zeros((max_rows,), dtype=torch.int64, device=dev)
    if out_lig.shape[0]:
        rows[:out_lig.shape[0]] = out_lig
        ids[:out_lig.shape[0]] = lig_mask + sample_lo
    rbuf = [torch.zeros_like(rows) for _ in range(world)]
    ibuf = [torch.zeros_like(ids) for _ in range(world)]
    dist.all_gather(rbuf, rows, group=group)
    dist.all_gather(ibuf, ids, group=group)
    all_rows = torch.cat([b[:c] for b, c in zip(rbuf, counts)], dim=0).to(out_dev)
    all_ids = torch.cat([b[:c] for b, c in zip(ibuf, counts)], dim=0).to(out_dev)
    return all_rows, all_ids


def sample_sharded(sample_fn, n_total: int, group=None, force_collective=False):
    """Run `sample_fn(lo, hi) -> (out_lig, lig_mask_local)` on this rank's
    shard and gather.  `sample_fn` must key its randomness by the global sample
    index (e.g. model.seed(seed, sample_offset=lo))."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    lo, hi = shard_range(n_total, world, rank)
    if hi > lo:
        out_lig, mask = sample_fn(lo, hi)
    else:
        out_lig, mask = None, None
    if out_lig is None:   # empty shard: still has to take part in the gather (width comes from the peers)
        probe_dev = torch.device("cuda", torch.cuda.current_device()) if (
            dist.is_initialized() and dist.get_backend(group) == "nccl") else torch.device("cpu")
        out_lig = torch.zeros((0, 0), device=probe_dev)
        mask = torch.zeros((0,), dtype=torch.int64, device=probe_dev)
    return gather_ligands(out_lig, mask, lo, group, force_collective=force_collective)
