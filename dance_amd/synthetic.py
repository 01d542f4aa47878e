"""Seeded synthetic weights and inputs for tests and the benchmark.

There is no network for checkpoints or datasets, and the reference's own initialisation
(`normal_(0, 0.01)` + identity BN, snake_head.py:639) drives activations into denormals,
so weights are He-scaled and BN statistics randomised.  Everything is drawn from a CPU
`torch.Generator`, in state-dict key order, so the same seed reproduces the same tensors
for the reference modules, the oracle and the HIP path.
"""
import math

import torch


def fill_state_dict_(sd, seed):
    """Overwrite every tensor of an (ordered) state-dict in place; returns a checksum."""
    g = torch.Generator().manual_seed(int(seed))
    checksum = 0.0
    for name, t in sd.items():
        if name.endswith("num_batches_tracked"):
            continue
        shape = tuple(t.shape)
        if name.endswith("running_mean"):
            v = torch.randn(shape, generator=g) * 0.1
        elif name.endswith("running_var"):
            v = torch.rand(shape, generator=g) + 0.5
        elif ".norm." in name or name.startswith("norm."):
            if name.endswith("weight"):
                v = torch.rand(shape, generator=g) + 0.5
            else:
                v = torch.randn(shape, generator=g) * 0.1
        elif name.endswith("weight"):
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            v = torch.randn(shape, generator=g) * math.sqrt(2.0 / fan_in)
        elif name.endswith("bias"):
            v = torch.randn(shape, generator=g) * 0.05
        else:
            raise KeyError("no synthetic rule for %s" % name)
        t.copy_(v.to(t.dtype))
        checksum += float(v.double().abs().sum())
    return checksum


def make_features(batch, channels, hf, wf, seed, attention=True):
    """(B, [1+]C, Hf, Wf) fp32 NCHW; channel 0 in [0,1] when `attention` (it is a sigmoid
    output in the reference, edge_det.py:94)."""
    g = torch.Generator().manual_seed(int(seed))
    c = channels + (1 if attention else 0)
    f = torch.randn((batch, c, hf, wf), generator=g)
    if attention:
        f[:, 0] = torch.rand((batch, hf, wf), generator=g)
    return f


def make_smooth_features(batch, channels, hf, wf, seed, attention=True, cell=8):
    """Like make_features but band-limited: white noise on a (hf/cell, wf/cell) lattice,
    bilinearly interpolated with explicit elementwise ops (bit-reproducible on any CPU).
    Real FPN features vary slowly at the pixel scale; white-noise features make the
    3-iteration deformation chaotic (a 0.005 px difference grows ~10x per iteration), which
    says nothing about an implementation.  Used where a whole chain is compared."""
    g = torch.Generator().manual_seed(int(seed))
    c = channels + (1 if attention else 0)
    gh, gw = hf // cell + 2, wf // cell + 2
    coarse = torch.randn((batch, c, gh, gw), generator=g)
    if attention:
        coarse[:, 0] = torch.rand((batch, gh, gw), generator=g)
    ys = torch.arange(hf, dtype=torch.float32) / cell
    xs = torch.arange(wf, dtype=torch.float32) / cell
    y0, x0 = ys.floor().long(), xs.floor().long()
    ty, tx = (ys - y0.float())[None, None, :, None], (xs - x0.float())[None, None, None, :]
    top = coarse[:, :, y0][:, :, :, x0] * (1 - tx) + coarse[:, :, y0][:, :, :, x0 + 1] * tx
    bot = coarse[:, :, y0 + 1][:, :, :, x0] * (1 - tx) + coarse[:, :, y0 + 1][:, :, :, x0 + 1] * tx
    return (top * (1 - ty) + bot * ty).contiguous()


def make_boxes(n, height, width, seed, r_min=20.0, r_max=100.0):
    """n xyxy boxes with centres in the central 80 % of the image."""
    g = torch.Generator().manual_seed(int(seed))
    cx = (0.1 + 0.8 * torch.rand(n, generator=g)) * width
    cy = (0.1 + 0.8 * torch.rand(n, generator=g)) * height
    rx = r_min + (r_max - r_min) * torch.rand(n, generator=g)
    ry = r_min + (r_max - r_min) * torch.rand(n, generator=g)
    b = torch.stack([cx - rx, cy - ry, cx + rx, cy + ry], dim=1)
    b[:, 0::2].clamp_(0, width - 1)
    b[:, 1::2].clamp_(0, height - 1)
    return b


def make_circle_contours(boxes, num_vertices):
    """Clockwise (image coordinates, y down) ellipse inscribed in each box: (N,V,2)."""
    cx = (boxes[:, 0] + boxes[:, 2]) * 0.5
    cy = (boxes[:, 1] + boxes[:, 3]) * 0.5
    rx = (boxes[:, 2] - boxes[:, 0]) * 0.5
    ry = (boxes[:, 3] - boxes[:, 1]) * 0.5
    t = torch.arange(num_vertices, dtype=torch.float32) * (2 * math.pi / num_vertices)
    x = cx[:, None] + rx[:, None] * torch.cos(t)[None]
    y = cy[:, None] + ry[:, None] * torch.sin(t)[None]
    return torch.stack([x, y], dim=2).contiguous()
