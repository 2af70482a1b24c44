"""De-risk the first multi-GPU run (task 7): on a ONE-GPU box, start one rank under torch.distributed.run
with backend "nccl" (RCCL on ROCm) and run `sharding.init_distributed`, `barrier(device_ids=...)`, the padded all_gather
of `sharding.gather_ligands` / `sample_sharded` (incl. the empty-shard branch with its CUDA probe tensors) on the
device -- RCCL loaded, a communicator created, collectives executed -- and compare with the world-1 short cut."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_one_rank_rccl_communicator_and_gather():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "_nccl_world1.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("NCCL1 ")][-1]
    out = json.loads(line[6:])
    assert out["backend"] == "nccl" and out["rccl_ranks"] == 1
    assert out["gather_equal"] and out["sharded_equal"] and out["gather_device"].startswith("cuda")
    assert out["empty_rows"] == 0 and out["empty_mask"] == 0
    assert out["all_reduce"] == [1.0, 1.0, 1.0, 1.0]
