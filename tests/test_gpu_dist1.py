"""The multi-rank path on ONE GPU: bench.py --force-dist starts its rank through the launcher child
(python -m torch.distributed.run, as for N > 1), initialises the RCCL process group with device_id,
and runs the exchange step's collectives (counts all-gather + padded all_gather_into_tensor) and the
barriers at world size 1 -- so an 8-GPU run is not the first contact with RCCL.  The gathered
contours must be bit-identical to the run without a process group.  reference: the launcher and
collectives of train_net.py:76-83, core/utils/comm.py:8-14."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(tmp_path, name, *flags):
    dump = str(tmp_path / (name + ".pt"))
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--workload", "cfg5", "--images", "8",
           "--per-image", "40", "--steps", "2", "--warmup", "1", "--no-extras", "--no-cpu-baseline",
           "--dump-output", dump] + list(flags)
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=170)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    return line, torch.load(dump, weights_only=True)


def test_force_dist_world1_runs_rccl_and_matches(tmp_path):
    plain, out_plain = _bench(tmp_path, "plain")
    assert plain["n_gpus"] == 1 and plain["rccl"]["initialised"] is False
    assert plain["rccl"]["all_gather_into_tensor_calls_in_timed_steps"] == 0
    dist1, out_dist = _bench(tmp_path, "dist", "--force-dist")
    assert dist1["n_gpus"] == 1 and dist1["scaling"] == "strong"
    assert dist1["rccl"]["initialised"] is True and dist1["rccl"]["backend"] == "nccl"
    # two collectives per step: the counts, then the padded contours
    assert dist1["rccl"]["all_gather_into_tensor_calls_in_timed_steps"] == 2 * dist1["steps"]
    assert out_dist.shape == (8 * 40, 256, 2) and torch.equal(out_dist, out_plain)
    assert dist1["config"]["contours_total"] == 320 and dist1["value"] > 0


def test_force_dist_weak_workload(tmp_path):
    """the weak workload's exchange (one all_gather_into_tensor of equal shards) at world size 1"""
    dump = str(tmp_path / "weak.pt")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--images", "2", "--steps", "2",
           "--warmup", "1", "--no-extras", "--cpu-reps", "1", "--dump-output", dump]
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=170)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["rccl"]["initialised"] is True
    assert line["rccl"]["all_gather_into_tensor_calls_in_timed_steps"] == line["steps"]
    assert line["cpu_baseline"]["value"] > 0 and line["cpu_baseline"]["kind"] == "port"
    assert torch.load(dump, weights_only=True).shape == (200, 128, 2)
