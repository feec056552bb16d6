"""-m gpu: the RCCL ("nccl") branch of the multi-GPU path on the one GPU a test box has: bench.py launched through
torch.distributed.run with one rank, the packed all-gather forced on (DISCO_FORCE_GATHER=1) - process-group init on the HIP
device, the event-count exchange and the asynchronous packed collective on device tensors, the closing barrier - must give the
same result checksum as the plain single-process run.  (Multi-rank behaviour is covered on gloo: tests/test_dist_gloo.py.)"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(extra_env, launcher):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
    cmd = launcher + [os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "4", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(line) == 1, r.stdout
    return json.loads(line[0])


def test_bench_on_rccl_with_forced_gather_matches_single_process():
    plain = _bench({}, [sys.executable])
    port = 34500 + os.getpid() % 1000
    dist_line = _bench({"DISCO_FORCE_GATHER": "1"},
                       [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", str(port)])
    assert dist_line["n_gpus"] == 1 and dist_line["kmeans_events"] == 0
    assert dist_line["result_checksum"] == plain["result_checksum"]
