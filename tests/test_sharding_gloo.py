"""CPU, world_size 2 over gloo: the N>1 path (shard -> independent chains ->
single gather) gives the same global result as world_size 1.  The sampler is a
stand-in keyed by the GLOBAL sample index, exactly the contract the HIP sampler
honours through dsbdd_randn_keyed(sample_offset=lo)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from diffsbdd_amd import sharding


def fake_sampler(n_lig_of, width=13):
    def fn(lo, hi):
        rows, mask = [], []
        for g in range(lo, hi):
            n = n_lig_of(g)
            gen = torch.Generator().manual_seed(1000 + g)          # keyed by GLOBAL index
            rows.append(torch.randn(n, width, generator=gen))
            mask.append(torch.full((n,), g - lo, dtype=torch.int64))  # LOCAL sample ids
        return torch.cat(rows), torch.cat(mask)
    return fn


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, q, width=13, n_fixed=0):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    r, lr, w = sharding.init_distributed("gloo")
    assert (r, w) == (rank, world)
    out, mask = sharding.sample_sharded(fake_sampler((lambda g: n_fixed) if n_fixed else (lambda g: 5 + g % 4), width), n_total)
    q.put((rank, out.clone(), mask.clone()))
    dist.barrier()
    dist.destroy_process_group()


def _run_world(world, n_total, width=13, n_fixed=0):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, q, width, n_fixed)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return sorted(res, key=lambda x: x[0])


def test_shard_range_partitions():
    for n in (0, 1, 7, 64, 512, 513):
        for w in (1, 2, 3, 8):
            spans = [sharding.shard_range(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(hi - lo for lo, hi in spans) <= (n + w - 1) // w
    assert sharding.shard_range(512, 8, 3) == (192, 256)   # BASELINE configs[3]: 64 per GPU


def test_world2_equals_world1():
    n_total = 7    # uneven: 4 + 3
    single_out, single_mask = sharding.gather_ligands(*fake_sampler(lambda g: 5 + g % 4)(0, n_total), 0)
    res = _run_world(2, n_total)
    for rank, out, mask in res:          # all_gather: every rank holds the global result
        assert torch.equal(mask, single_mask)
        assert torch.equal(out, single_out)


def test_world8_config3_shape_equals_eight_sequential_world1_runs():
    """BASELINE configs[3]: 512 samples over 8 ranks -- the shards are [0, 64), [64, 128), ... [448, 512), and the padded
    all_gather returns, on every rank, exactly the rows of 8 sequential world-1 runs (one per shard) in global order."""
    n_total, world = 512, 8
    fn = fake_sampler(lambda g: 23)
    seq = [sharding.gather_ligands(*fn(lo, hi), lo) for lo, hi in (sharding.shard_range(n_total, world, r) for r in range(world))]
    assert [int(m[0]) for _, m in seq] == [0, 64, 128, 192, 256, 320, 384, 448]
    ref_out, ref_mask = torch.cat([o for o, _ in seq]), torch.cat([m for _, m in seq])
    assert ref_out.shape == (512 * 23, 13)
    res = _run_world(world, n_total, n_fixed=23)
    assert len(res) == world
    for rank, out, mask in res:
        assert torch.equal(mask, ref_mask) and torch.equal(out, ref_out)


def test_world2_with_empty_shard():
    res = _run_world(2, 1)               # rank 1 owns nothing but must join the gather
    ref_out, ref_mask = sharding.gather_ligands(*fake_sampler(lambda g: 5 + g % 4)(0, 1), 0)
    for rank, out, mask in res:
        assert torch.equal(out, ref_out) and torch.equal(mask, ref_mask)


def test_world2_empty_shard_other_feature_width():
    """The empty rank learns the feature width from its peers (3 + atom_nf is not 13 for every model)."""
    res = _run_world(2, 1, width=7)
    ref_out, ref_mask = sharding.gather_ligands(*fake_sampler(lambda g: 5 + g % 4, 7)(0, 1), 0)
    for rank, out, mask in res:
        assert out.shape[1] == 7
        assert torch.equal(out, ref_out) and torch.equal(mask, ref_mask)


def test_bench_self_launch_command(monkeypatch):
    """`python bench.py --gpus N` without torchrun: bench.py re-executes itself under
    torch.distributed.run with N ranks, rendezvous on 127.0.0.1 (command checked here; the
    launch itself runs in the GPU suite)."""
    import subprocess
    import sys
    import bench
    seen = {}

    class R:
        returncode = 0

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return R()

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "1"])
    assert bench.self_launch(4) == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "1"] and cmd[-5].endswith("bench.py")
    assert seen["env"]["MASTER_ADDR"] == "127.0.0.1" and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
