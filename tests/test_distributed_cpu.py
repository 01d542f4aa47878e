"""CPU, world_size 2, gloo: image sharding, the contour all-gather and bench.py's own launcher."""
import json
import os
import socket
import subprocess
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dance_amd import distributed as dd


def test_image_range_partition():
    for b in (1, 7, 16, 256):
        for world in (1, 2, 3, 8):
            spans = [dd.image_range(b, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == b
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    idx = torch.tensor([0., 0., 1., 2., 2., 3.])
    mask, lo = dd.shard_by_image(idx, 4, 1, 2)
    assert mask.tolist() == [False, False, False, True, True, True] and lo == 2


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        v = 8
        full = torch.arange(10 * v * 2, dtype=torch.float32).view(10, v, 2)
        # equal shards
        local = full[rank * 5:(rank + 1) * 5].clone()
        out = dd.all_gather_contours(local)
        ok1 = torch.equal(out, full)
        # ragged shards: 7 + 3 contours
        lo, hi = (0, 7) if rank == 0 else (7, 10)
        counts = dd.gather_counts(hi - lo, torch.device("cpu"))
        out2 = dd.all_gather_contours(full[lo:hi].clone(), counts)
        ok2 = counts == [7, 3] and torch.equal(out2, full)
        # the strong-scaling split of one sweep (bench.py --workload cfg5): ragged instances per
        # image, a stub "deformer" that depends on the GLOBAL image number; the union of the ranks'
        # shards must be the full set in the original order
        per = torch.tensor([3, 1, 4, 1, 5])
        idx = torch.repeat_interleave(torch.arange(5), per).float()
        contours = torch.rand((int(per.sum()), v, 2), generator=torch.Generator().manual_seed(3))
        stub = lambda c, i: c * 2 + i[:, None, None]          # noqa: E731
        shard = dd.ImageShard(idx, 5, rank, world)
        assert shard.images == ((0, 3) if rank == 0 else (3, 5)) and len(shard) == (8 if rank == 0 else 6)
        assert float(shard.local_image_idx.min()) == 0.0
        local = stub(shard.take(contours), shard.local_image_idx + shard.images[0])
        ok3 = torch.equal(dd.gather_shards(local), stub(contours, idx))
        ret[rank] = bool(ok1 and ok2 and ok3)
    finally:
        dist.destroy_process_group()


def test_all_gather_contours_gloo():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    assert ret[0] and ret[1]


def test_image_shard_needs_grouped_contours():
    import pytest
    with pytest.raises(ValueError):
        dd.ImageShard(torch.tensor([0., 1., 0.]), 2, 0, 2)
    one = dd.ImageShard(torch.tensor([0., 0., 1.]), 2, 0, 1)      # world 1: everything
    assert len(one) == 3 and one.images == (0, 2)


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*argv, env=None):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(argv), env=e,
                          capture_output=True, text=True, timeout=300)


def test_bench_gpus_flag_starts_the_ranks_itself():
    """`python bench.py --gpus 2` (no launcher around it, as the driver calls it) must run TWO
    ranks: launcher -> by-image shard -> counts + contour all-gather -> one JSON line from rank 0.
    --dry-run-cpu swaps RCCL for gloo and the HIP deformer for a stub; everything else is the
    code path of the real run."""
    for workload, scaling in (("cfg5", "strong"), ("cfg2", "weak")):
        r = _bench("--gpus", "2", "--workload", workload, "--dry-run-cpu")
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, r.stdout
        line = json.loads(lines[0])
        assert line["n_gpus"] == 2 and line["ok"] is True and line["scaling"] == scaling and line["dry_run"] is True


def test_bench_refuses_a_world_that_differs_from_gpus():
    r = _bench("--gpus", "4", "--dry-run-cpu", env={"WORLD_SIZE": "2", "RANK": "0"})
    assert r.returncode != 0 and "--gpus 4" in r.stderr


def test_force_dist_runs_the_group_at_world_one():
    """--force-dist: one rank through the launcher child, a process group and the exchange-step
    collectives at world size 1 (gloo here; tests/test_gpu_dist1.py runs the same with RCCL)."""
    r = _bench("--gpus", "1", "--force-dist", "--workload", "cfg5", "--dry-run-cpu")
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["ok"] is True and line["process_group"] is True
    r = _bench("--gpus", "1", "--workload", "cfg5", "--dry-run-cpu")
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["ok"] is True and line["process_group"] is False
