"""Helper of tests/test_gpu_rccl.py: ONE rank under torch.distributed.run with backend "nccl" (= RCCL): creates the
communicator, runs the barrier and the exchange steps of diffsbdd_amd/sharding.py on CUDA tensors -- incl. the
empty-shard branch -- and prints one JSON line."""
import json
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffsbdd_amd import sharding  # noqa: E402


def main():
    rank, local_rank, world = sharding.init_distributed(backend="nccl", force=True)
    dev = torch.device("cuda", local_rank)
    dist.barrier(device_ids=[local_rank])
    g = torch.Generator().manual_seed(0)
    lig = torch.randn(46, 13, generator=g).to(dev)
    mask = torch.repeat_interleave(torch.arange(2), 23).to(dev)
    a_lig, a_mask = sharding.gather_ligands(lig, mask, 5)                              # world-1 short cut
    b_lig, b_mask = sharding.gather_ligands(lig, mask, 5, force_collective=True)       # the padded all_gather over RCCL
    full = sharding.sample_sharded(lambda lo, hi: (lig, mask), 2, force_collective=True)
    empty = sharding.sample_sharded(lambda lo, hi: (lig, mask), 0, force_collective=True)   # hi == lo: the CUDA probe tensors
    red = torch.ones(4, device=dev)
    dist.all_reduce(red)
    torch.cuda.synchronize()
    out = {"backend": dist.get_backend(), "rccl_ranks": dist.get_world_size(), "rank": rank,
           "gather_equal": bool(torch.equal(a_lig, b_lig) and torch.equal(a_mask, b_mask)),
           "gather_device": str(b_lig.device), "sharded_equal": bool(torch.equal(full[0], lig) and torch.equal(full[1], mask)),
           "empty_rows": int(empty[0].shape[0]), "empty_mask": int(empty[1].numel()), "all_reduce": red.tolist()}
    dist.barrier(device_ids=[local_rank])
    dist.destroy_process_group()
    print("NCCL1 " + json.dumps(out))


if __name__ == "__main__":
    main()
