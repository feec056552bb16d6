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


def _bench(extra_env, launcher, args=("--gpus", "1", "--batch", "4")):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
    cmd = launcher + [os.path.join(REPO, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline"] + list(args)
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


@pytest.mark.parametrize("n_global", [16, 17])
def test_world_2_on_one_gpu_with_device_tensors(n_global):
    """Two ranks SHARING cuda:0: bench.py's real N>1 configuration - real forwards, steps pipelined over two HIP streams per rank, results
    written in place, the packed all-gather enqueued behind each forward with `record_stream` bookkeeping on device tensors, the event-count
    exchange of the exact mode, the closing barrier - with the collectives on gloo (RCCL wants a device per rank, so it has only ever
    seen world size 1 here; this torch build's gloo takes host tensors only for all_gather_into_tensor: runner.all_gather_into stages
    device rows through pinned host buffers for that backend).  Equal shards (8 + 8: gathered straight into the result) and ragged
    ones (9 + 8: padded and unpacked); the result checksum must equal the single-process run of the same global batch."""
    plain = _bench({}, [sys.executable], ("--gpus", "1", "--global-batch", str(n_global)))
    port = 35600 + os.getpid() % 1000 + n_global
    # the two ways an N-rank bench is started: through torch.distributed.run (17) and - the driver's command - as plain
    # `python bench.py --gpus 2`, which starts its own ranks (16)
    launcher = [sys.executable] if n_global == 16 else [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                                                        "--master-addr", "127.0.0.1", "--master-port", str(port)]
    two = _bench({"DISCO_DIST_BACKEND": "gloo"}, launcher, ("--gpus", "2", "--global-batch", str(n_global)))
    assert two["n_gpus"] == 2 and two["world_size_seen_by_backend"] == 2 and two["kmeans_events"] == 0
    assert two["config"]["global_batch"] == n_global
    assert two["result_checksum"] == plain["result_checksum"]


@pytest.mark.parametrize("config,batch", [("3", 24), ("5a", 6), ("5b", 6)])
def test_named_configs_on_two_ranks_equal_one_rank(config, batch):
    """bench.py --config 3 / 5a / 5b (BASELINE's 8-GPU configurations by name) with the global batch cut down to what a test can afford:
    two self-launched ranks sharing cuda:0 must give the single-process checksum - K = 8 / --diverse K = 16 (three colorizations per image,
    packed gather of 3 rows per image) / random_hint K = 16 (`random` draws in global image order)."""
    one = _bench({}, [sys.executable], ("--gpus", "1", "--config", config, "--global-batch", str(batch)))
    two = _bench({"DISCO_DIST_BACKEND": "gloo"}, [sys.executable], ("--gpus", "2", "--config", config, "--global-batch", str(batch)))
    assert two["world_size_seen_by_backend"] == 2 and two["config"]["name"] == config and two["kmeans_events"] == 0
    assert two["config"]["colorizations_per_step"] == batch * (3 if config == "5a" else 1)
    assert one["result_checksum"] == two["result_checksum"]


@pytest.mark.parametrize("config", ["3", "5a"])
def test_eight_ranks_sharing_one_gpu_equal_the_single_process_run(config):
    """BASELINE's 8-GPU configurations at their FULL global batch and their real world size, on the one GPU a test box has: `python bench.py
    --gpus 8 --config 3` (512 images, 64 per rank) / `--config 5a` (256 images --diverse K = 16, 32 per rank = 96 colorizations each) with the
    eight self-launched ranks sharing cuda:0 (collectives on gloo, staged through pinned host buffers; RCCL wants a device per rank).  Everything
    an 8 x MI355X run does except the xGMI transport: the launch, the thread caps, shard bounds, the global draws sliced at eight offsets, the
    collective range check of the first batches, pipelined forwards with the packed gather behind each, the exactness check, max-over-ranks
    timing.  The result checksum must equal the single-process run's: the gathered 512 (768) results are bit for bit the one-GPU ones."""
    args = ("--config", config, "--steps", "1", "--warmup", "0", "--no-cpu-baseline")
    one = _bench({}, [sys.executable], ("--gpus", "1") + args)
    eight = _bench({"DISCO_DIST_BACKEND": "gloo"}, [sys.executable], ("--gpus", "8") + args)
    assert eight["n_gpus"] == 8 and eight["world_size_seen_by_backend"] == 8 and eight["kmeans_events"] == 0
    assert eight["config"]["global_batch"] == (512 if config == "3" else 256) and eight["config"]["images_per_gpu"] == (64 if config == "3" else 32)
    assert eight["result_checksum"] == one["result_checksum"]
