"""CPU, world_size 2, gloo: image sharding and the contour all-gather."""
import os
import socket

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
        ret[rank] = bool(ok1 and ok2)
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
