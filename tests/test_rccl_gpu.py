"""GPU: one rank under torch.distributed.run with backend "nccl" (RCCL) -- see tests/rccl_one_rank.py.  The only part of the
eight-GPU run that can fail for reasons unrelated to scaling (two HIP runtimes in one process with an RCCL communicator
alive, the async collectives of playsnark_amd/dist.py on device tensors) is exercised on the one GPU a test box has."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_rccl_first_contact_one_rank():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "rccl_one_rank.py")]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=540, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    for line in ("RCCL communicator up", "sharded MSM over RCCL ok", "sharded provers over RCCL ok", "rccl one rank ok"):
        assert line in res.stdout, res.stdout[-2000:]
