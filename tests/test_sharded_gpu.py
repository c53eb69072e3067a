"""BASELINE cfg 4 in miniature on the GPU: two ranks (gloo rendezvous on 127.0.0.1, both on the
test box's one GPU) each score their shard of 1517 candidates with the HIP kernel, all-gather the
scores and rank them; the result must equal the unsharded CPU chain bit for bit.  The RCCL
transport itself only runs on the driver's 8-GPU node (two ranks cannot share a GPU under RCCL);
the sharding, the ragged all-gather and the product kernels are what this covers."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_two_ranks_score_shards_and_rank_identically(hiplib, oracle):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port),
                          os.path.join(ROOT, "tests", "sharded_gpu_worker.py")],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "scores bit-identical True" in out.stdout


@pytest.mark.gpu
def test_two_ranks_hip_bilinear_backward_and_sharded_loss(hiplib, oracle):
    """The HIP bilinear backward feeds all_reduce_param_grads (dW, dbias == unsharded oracle), and the fused triplet
    step runs on shards with sharded.shard_loss_weight (gradients / loss == the unsharded HIP run; the unscaled call
    is world_size times too large)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port),
                          os.path.join(ROOT, "tests", "sharded_gpu_worker.py"), "bilinear_and_loss"],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "unsharded oracle: True" in out.stdout and "== unsharded: True" in out.stdout
