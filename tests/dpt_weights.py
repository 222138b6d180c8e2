"""Seeded, non-degenerate random weights for the DPT numerics tests (test infrastructure).

PyTorch's default initialisation leaves the depth head at ~0 behind its ReLUs and the RefineNet paths with a tiny
dynamic range, so a whole-model comparison says little.  ``seeded_init`` gives every stage a usable range:
variance-preserving (He) convolution / linear weights, non-trivial norm affine parameters, and a depth head whose
output spans the NYU checkpoint's working range (inverse-depth units of the reference's scale / shift,
/root/reference/hive/dataset_adaptors.py:1366-1374), i.e. depths between roughly 0.5 m and 7 m.

Deterministic for a given torch build (CPU generator); ``state_checksum`` lets a committed fixture detect drift.
"""
import hashlib

import numpy as np
import torch


from hive_amd.dpt.init import seeded_init  # noqa: E402,F401  (moved into the package: bench.py uses it too)


def state_checksum(model):
    """sha256 over the float32 bytes of the state dict (key order of ``state_dict()``)."""
    h = hashlib.sha256()
    for k, v in model.state_dict().items():
        if k.endswith("_std_weight"):
            continue
        h.update(k.encode())
        h.update(np.ascontiguousarray(v.detach().float().cpu().numpy()).tobytes())
    return h.hexdigest()


def seeded_input(batch, height, width, seed=99):
    """Network input in [-1, 1] (what NormalizeImage(0.5, 0.5) produces): smooth pattern + noise, float32 NCHW."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    v, u = torch.meshgrid(torch.linspace(0, 1, height), torch.linspace(0, 1, width), indexing="ij")
    imgs = []
    for b in range(batch):
        ph = torch.rand(6, generator=g) * 6.28
        fr = torch.rand(6, generator=g) * 6 + 1
        chans = [0.6 * torch.sin(fr[2 * c] * 6.28 * u + ph[2 * c]) * torch.cos(fr[2 * c + 1] * 6.28 * v + ph[2 * c + 1]) for c in range(3)]
        imgs.append(torch.stack(chans) + 0.2 * (torch.rand(3, height, width, generator=g) - 0.5))
    return torch.stack(imgs).clamp(-1, 1).contiguous()
