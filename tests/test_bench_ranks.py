"""bench.py's multi-rank path on ONE device: `--gpus 2` spawns two child ranks (they share the GPU and rendezvous
over gloo, the rehearsal backend) -- the same code the driver runs over RCCL with one GPU per rank.  Checks that the
line says n_gpus == 2 and that the sharded job (strong scaling: one global list dealt to the ranks,
vall_e_amd.inference.shard_utterances) returns, utterance for utterance, the codes of the single-rank job."""
import json
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _bench(tmp_path, gpus, tag):
    out = tmp_path / f"codes_{tag}.npy"
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", str(gpus), "--steps", "1", "--warmup", "0", "--frames", "10",
           "--global-batch", "6", "--no-cpu-baseline", "--dump-codes", str(out)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    return line, np.load(out)


def test_two_ranks_equal_one_rank(gpu_device, tmp_path):
    one, c1 = _bench(tmp_path, 1, "one")
    two, c2 = _bench(tmp_path, 2, "two")
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert two["scaling"] == "strong" and two["config"]["global_batch"] == 6
    assert len(two["rank_seconds"]) == 2 and all(t > 0 for t in two["rank_seconds"])
    assert two["value"] > 0 and two["roofline"] is not None
    assert c1.shape == c2.shape == (6, 10, 8)
    np.testing.assert_array_equal(c1, c2)


def test_gpus_flag_must_match_the_launcher(gpu_device):
    import os
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True, text=True,
                       env=env, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr


def test_rccl_calls_at_world_size_one(gpu_device, tmp_path):
    """The collective calls of the multi-rank path on the real RCCL backend: VALLE_BENCH_FORCE_DIST=1 initialises the
    process group at world size 1 (a one-GPU box cannot hold two RCCL ranks), so init, the barriers and the two
    all-gathers on device tensors run as they do with one GPU per rank; codes equal the run without a process group."""
    import os
    one, c1 = _bench(tmp_path, 1, "plain")
    out = tmp_path / "codes_rccl.npy"
    env = dict(os.environ, VALLE_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    env.pop("VALLE_BENCH_BACKEND", None)
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "0", "--frames", "10",
                        "--global-batch", "6", "--no-cpu-baseline", "--dump-codes", str(out)],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["collective_backend"] == "rccl" and line["n_gpus"] == 1
    np.testing.assert_array_equal(c1, np.load(out))
