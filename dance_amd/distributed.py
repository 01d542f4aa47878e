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


class ImageShard:
    """The share of one fixed sweep that rank `rank` of `world` deforms (strong scaling, BASELINE
    config 5): the contours of the contiguous image block `image_range(num_images, rank, world)`.

    image_idx (N,) must be non-decreasing -- contours grouped by image, as the reference's
    `sample_bboxes_fast` emits them (snake_head.py:865-886) -- so that the concatenation of the
    ranks' shards in rank order IS the original contour order and the all-gather needs no permutation.
      index            positions of the owned contours in the full set (int64)
      local_image_idx  their image index relative to the first owned image (only the owned images'
                       feature maps are resident on the rank)
      images           (lo, hi) block of owned images
    """

    def __init__(self, image_idx, num_images, rank, world):
        idx = image_idx.to(torch.int64)
        if idx.numel() > 1 and bool((idx[1:] < idx[:-1]).any()):
            raise ValueError("ImageShard: image_idx must be non-decreasing (contours grouped by image)")
        mask, lo = shard_by_image(idx, num_images, rank, world)
        self.images = image_range(num_images, rank, world)
        self.index = mask.nonzero().flatten()
        self.local_image_idx = (idx[self.index] - lo).to(image_idx.dtype)

    def take(self, t):
        """Rows of a per-contour tensor that belong to this rank."""
        return t.index_select(0, self.index.to(t.device))

    def __len__(self):
        return int(self.index.numel())


def gather_shards(local, group=None):
    """The exchange step of a sharded sweep: every rank's deformed contours (n_r, V, 2), n_r
    differing with the instances per image, into the full (N, V, 2) set on every rank in the
    original order: one tiny all-gather of the counts, then one padded all_gather_into_tensor.
    Without a process group there is nothing to exchange; WITH one the collectives run whatever the
    world size -- a world of one rank (bench.py --force-dist) takes the same RCCL calls as eight."""
    if not dist.is_available() or not dist.is_initialized():
        return local
    counts = gather_counts(local.size(0), local.device, group)
    return all_gather_contours(local, counts, group)


def all_gather_contours(local, counts=None, group=None):
    """All-gather per-rank contour tensors (n_r, V, 2) into (sum n_r, V, 2) on every rank,
    rank order preserved.

    counts: list of n_r for all ranks when they differ (shards are padded to max(counts) so
    one all_gather_into_tensor moves everything); None when every rank holds the same n.
    No early return at world size 1 (see gather_shards).
    """
    if not dist.is_available() or not dist.is_initialized():
        return local
    world = dist.get_world_size(group)
    n, v, two = local.shape
    if counts is None:
        out = torch.empty((world * n, v, two), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        CALLS["all_gather_into_tensor"] += 1
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
    CALLS["all_gather_into_tensor"] += 1
    return torch.cat([out[r * m:r * m + counts[r]] for r in range(world)], dim=0)


CALLS = {"all_gather_into_tensor": 0}       # collectives issued by this process (bench.py reports it)


def gather_counts(n_local, device, group=None):
    """Every rank's contour count (one tiny all_gather), for variable instances per image."""
    world = dist.get_world_size(group)
    t = torch.tensor([n_local], dtype=torch.int64, device=device)
    out = torch.empty((world,), dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(out, t, group=group)
    CALLS["all_gather_into_tensor"] += 1
    return [int(x) for x in out.tolist()]
