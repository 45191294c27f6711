"""Two data-parallel ranks on one MI355X (gloo, both on cuda:0): the trainer's eager per-block all-reduce and its
four-graph replay keep the replicas identical and agree with each other.  RCCL itself needs one GPU per rank and is
exercised by the driver's multi-GPU bench; everything else of the N > 1 path runs here."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_two_ranks_one_gpu_eager_vs_graph_replay():
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29547", os.path.join(here, "dp_gpu_worker.py")]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=540)
    assert out.returncode == 0 and "DP_GPU_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


def test_sync_batchnorm_matches_single_process_batch_statistics():
    """TTSTrainingConfig.sync_batchnorm: two ranks, each with a part of the batch (2 + 2 and 3 + 1 samples), reproduce the
    BatchNorm of one process over the whole batch -- outputs, input gradients, summed parameter gradients and running
    statistics (reference modules.py:29,127) -- through the all-reduced per-channel sums of edges.ConvBNAct."""
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29549", os.path.join(here, "dp_syncbn_worker.py")]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=540)
    assert out.returncode == 0 and "SYNCBN_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]
