"""Polygon -> RLE masks on the GPU: the step right after the contour path.

Mirrors the reference's `get_polygon_rles(polygons, image_shape)`
(core/evaluation/coco_evaluation.py:311-318, core/modeling/postprocessing.py:266-274,
core/modeling/dsnake_baseline/postprocessing.py:13-21), which copies the contours to the host
and calls pycocotools once per instance.  Here the contours stay on the device; only the
packed "counts" strings (a few hundred bytes per instance) cross to the host.

Validation, output allocation and stream plumbing only -- the arithmetic is in
libdance_hip.so (csrc/raster.hip).  GPU tensors required; no CPU or torch fallback.
"""
import torch

from ._lib import check, current_stream, lib, ptr

MAX_VERTICES = 512


def _as_polys(polygons):
    if not polygons.is_cuda:
        raise RuntimeError("dance_amd.rle needs GPU tensors (got a %s tensor)" % polygons.device)
    if polygons.dtype != torch.float32:
        raise RuntimeError("expected float32 polygons, got %s" % polygons.dtype)
    if polygons.dim() == 2:                      # the reference's flattened (N, 2V)
        if polygons.size(1) % 2:
            raise RuntimeError("flattened polygons must have an even number of columns")
        polygons = polygons.reshape(polygons.size(0), polygons.size(1) // 2, 2)
    if polygons.dim() != 3 or polygons.size(2) != 2:
        raise RuntimeError("polygons must be (N,V,2) or (N,2V)")
    if not 1 <= polygons.size(1) <= MAX_VERTICES:
        raise RuntimeError("polygons need 1..%d vertices" % MAX_VERTICES)
    return polygons.contiguous()


class _CapacityExceeded(Exception):
    """an instance has more runs than the slots it was given (deferred check)"""

    def __init__(self, need):
        super().__init__(need)
        self.need = int(need)


class PolygonRLE:
    """Run lengths of N masks on the device.

    counts (N,cap) int32 holding the library's uint32 values (column-major runs, first run =
    zeros), m (N,) int32 number of runs, area (N,) int32 mask areas (rleArea)."""

    def __init__(self, counts, m, area, image_shape):
        self.counts, self.m, self.area = counts, m, area
        self.image_shape = (int(image_shape[0]), int(image_shape[1]))

    def __len__(self):
        return self.counts.size(0)

    def strings(self):
        """list of N `bytes`: pycocotools' compressed "counts" (rleToString).  Two host round trips:
        the packed sizes (with the run counts of a deferred capacity check), then the packed bytes."""
        n, cap = self.counts.shape
        if n == 0:
            return []
        dev = self.counts.device
        lens = torch.empty((n,), dtype=torch.int32, device=dev)
        with torch.cuda.device(dev):
            check(lib().dance_rle_to_string(ptr(self.counts), ptr(self.m), n, cap, None, None, 0, ptr(lens),
                                            current_stream()), "rle_to_string")
            offsets = torch.zeros((n + 2,), dtype=torch.int64, device=dev)
            torch.cumsum(lens, 0, out=offsets[1:n + 1])
            offsets[n + 1] = self.m.max()        # rides along: one copy instead of two
            off = offsets.cpu().tolist()
            need = off.pop()
            if need > cap:                       # (only reachable with the capacity check deferred)
                raise _CapacityExceeded(need)
            total = off[-1]
            packed = torch.empty((max(total, 1),), dtype=torch.uint8, device=dev)
            check(lib().dance_rle_to_string(ptr(self.counts), ptr(self.m), n, cap, ptr(packed), ptr(offsets),
                                            total, ptr(lens), current_stream()), "rle_to_string")
        raw = packed.cpu().numpy().tobytes()
        return [raw[off[i]:off[i + 1]] for i in range(n)]

    def to_coco(self):
        """The reference's return value: [{"size": [h, w], "counts": bytes}, ...]."""
        h, w = self.image_shape
        return [{"size": [h, w], "counts": s} for s in self.strings()]

    def decode(self):
        """(N,h,w) uint8 masks (a transposed view of the library's column-major memory, as
        pycocotools' Fortran-ordered `decode` output)."""
        h, w = self.image_shape
        n, cap = self.counts.shape
        mask = torch.empty((n, w, h), dtype=torch.uint8, device=self.counts.device)
        if n:
            with torch.cuda.device(mask.device):
                check(lib().dance_rle_decode(ptr(self.counts), ptr(self.m), n, cap, ptr(mask), h, w,
                                             current_stream()), "rle_decode")
        return mask.transpose(1, 2)


def polygon_rle(polygons, image_shape, cap=None, defer_check=False):
    """frPyObjects + merge for N single-polygon instances: polygons (N,V,2) or (N,2V) float32
    on the GPU, image_shape (h, w) -> PolygonRLE.  `cap` = run slots per instance (default
    2*w + 2, enough for any polygon that crosses each column at most twice); if an instance
    needs more the call is repeated once with the exact maximum (this check synchronises, as
    the reference's host copy does).  defer_check: no synchronisation here; `strings()` looks at the run
    counts with its first host copy and raises _CapacityExceeded (get_polygon_rles repeats the call)."""
    polys = _as_polys(polygons)
    h, w = int(image_shape[0]), int(image_shape[1])
    if h < 1 or w < 1 or h > 327680 or h * w >= 2 ** 31:
        raise RuntimeError("polygon_rle: unsupported image shape %s" % (image_shape,))
    n, v, _ = polys.shape
    dev = polys.device
    cap = int(cap) if cap else 2 * w + 2
    while True:
        counts = torch.empty((n, cap), dtype=torch.int32, device=dev)
        m = torch.empty((n,), dtype=torch.int32, device=dev)
        area = torch.empty((n,), dtype=torch.int32, device=dev)
        if n == 0:
            break
        with torch.cuda.device(dev):
            check(lib().dance_polygon_rle(ptr(polys), None, n, v, h, w, ptr(counts), cap, ptr(m), ptr(area),
                                          None, current_stream()), "polygon_rle")
        if defer_check:
            break
        need = int(m.max())
        if need <= cap:
            break
        cap = need
    return PolygonRLE(counts, m, area, (h, w))


def get_polygon_rles(polygons, image_shape):
    """Drop-in for the reference's function of the same name (coco_evaluation.py:311-318):
    polygons N x (p*2) (a GPU tensor here -- no `.cpu()` of the contours), image_shape (h, w)
    -> list of {"size": [h, w], "counts": bytes}."""
    try:
        return polygon_rle(polygons, image_shape, defer_check=True).to_coco()
    except _CapacityExceeded as e:
        return polygon_rle(polygons, image_shape, cap=e.need).to_coco()
