"""Data-parallel sharding of the contour path over the GPUs of one node.

The path shards by IMAGE: every contour only reads its own image's feature map and its own
vertices (snake_head.py:1561-1576; all convolutions run per instance along the ring), so
rank r deforms the contours of images [r*B/G, (r+1)*B/G) with replicated weights and no
data-path collective.  The one exchange step is the all-gather of the deformed contours
(RCCL over xGMI through torch.distributed, backend "nccl" on ROCm; "gloo" in the CPU
tests): equal-sized padded shards so it is a single `all_gather_into_tensor`.
"""
import torch
import torch.distributed as dist


def image_range(num_images, rank, world):
    """Contiguous block of images of rank `rank` (first `num_images % world` ranks get one more)."""
    base, extra = divmod(num_images, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def shard_by_image(image_idx, num_images, rank, world):
    """Boolean mask of the contours rank `rank` owns + the image offset to subtract from
    their image_idx when only the owned images' feature maps are resident."""
    lo, hi = image_range(num_images, rank, world)
    idx = image_idx.to(torch.int64)
    return (idx >= lo) & (idx < hi), lo


def all_gather_contours(local, counts=None, group=None):
    """All-gather per-rank contour tensors (n_r, V, 2) into (sum n_r, V, 2) on every rank,
    rank order preserved.

    counts: list of n_r for all ranks when they differ (shards are padded to max(counts) so
    one all_gather_into_tensor moves everything); None when every rank holds the same n.
    """
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    n, v, two = local.shape
    if counts is None:
        out = torch.empty((world * n, v, two), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    if len(counts) != world or counts[dist.get_rank(group)] != n:
        raise ValueError("counts must list every rank's contour count")
    m = max(counts)
    padded = local
    if n < m:
        padded = torch.zeros((m, v, two), dtype=local.dtype, device=local.device)
        padded[:n] = local
    out = torch.empty((world * m, v, two), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, padded.contiguous(), group=group)
    return torch.cat([out[r * m:r * m + counts[r]] for r in range(world)], dim=0)


def gather_counts(n_local, device, group=None):
    """Every rank's contour count (one tiny all_gather), for variable instances per image."""
    world = dist.get_world_size(group)
    t = torch.tensor([n_local], dtype=torch.int64, device=device)
    out = torch.empty((world,), dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(out, t, group=group)
    return [int(x) for x in out.tolist()]
