"""Pre-processing transforms of the reference's (absent) ``dpt.transforms`` module, as constructed at
/root/reference/hive/dataset_adaptors.py:1376-1392: ``Resize`` -> ``NormalizeImage`` -> ``PrepareForNet``,
each callable on ``{"image": ndarray}``.

``Resize`` follows the published MiDaS/DPT rule (keep aspect ratio, "minimal" scaling, sizes constrained
to a multiple of 32).  The reference interpolates with ``cv2.INTER_CUBIC``; OpenCV is not available here,
so the bicubic resampling runs through ``torch.nn.functional.interpolate(mode="bicubic",
align_corners=False)`` (same a = -0.75 kernel and pixel-centre convention; borders differ slightly).
At HIVE's fixed 640 x 480 network size on 640 x 480 frames the resize is the identity.
"""
import math

import numpy as np

INTER_CUBIC = 2  # value of cv2.INTER_CUBIC, accepted for call-site compatibility
INTER_AREA = 3


class Resize:
    """Target size of the network input.  Behaviour (what the published DPT pre-processing does; the sizing rule must give identical sizes):
    per-axis scale factors towards (width, height); with ``keep_aspect_ratio`` both axes take ONE of the two factors -- the larger for
    "lower_bound" (the result covers the target), the smaller for "upper_bound" (it fits inside), the one closer to 1 for "minimal";
    each side is then snapped to a multiple of ``ensure_multiple_of``: to the nearest one, but never below the target for "lower_bound"
    (snapped up) nor above it for "upper_bound" (snapped down)."""

    _PICK = {"lower_bound": max, "upper_bound": min, "minimal": lambda a, b: a if abs(1.0 - a) < abs(1.0 - b) else b}

    def __init__(self, width, height, resize_target=True, keep_aspect_ratio=False, ensure_multiple_of=1,
                 resize_method="lower_bound", image_interpolation_method=INTER_AREA):
        if resize_method not in self._PICK:
            raise ValueError(f"resize_method {resize_method} not implemented")
        self._target = (int(width), int(height))
        self._resize_target = resize_target
        self._keep_aspect = bool(keep_aspect_ratio)
        self._multiple = ensure_multiple_of
        self._method = resize_method
        self._interpolation = image_interpolation_method

    def constrain_to_multiple_of(self, x, min_val=0, max_val=None):
        """x snapped to the grid of multiples: nearest (ties to even, numpy's rounding), down if that exceeds max_val, up if it falls below min_val."""
        steps = x / self._multiple
        snapped = int(np.round(steps)) * self._multiple
        if max_val is not None and snapped > max_val:
            snapped = int(np.floor(steps)) * self._multiple
        if snapped < min_val:
            snapped = int(np.ceil(steps)) * self._multiple
        return snapped

    def get_size(self, width, height):
        target_w, target_h = self._target
        factor_w, factor_h = target_w / width, target_h / height
        if self._keep_aspect:
            factor_w = factor_h = self._PICK[self._method](factor_w, factor_h)  # ("minimal": equally close factors -> the height's)
        bounds_w, bounds_h = {}, {}
        if self._method == "lower_bound":
            bounds_w, bounds_h = {"min_val": target_w}, {"min_val": target_h}
        elif self._method == "upper_bound":
            bounds_w, bounds_h = {"max_val": target_w}, {"max_val": target_h}
        return (int(self.constrain_to_multiple_of(factor_w * width, **bounds_w)),
                int(self.constrain_to_multiple_of(factor_h * height, **bounds_h)))

    def __call__(self, sample):
        width, height = self.get_size(sample["image"].shape[1], sample["image"].shape[0])
        image = sample["image"]
        if (height, width) != image.shape[:2]:
            import torch
            import torch.nn.functional as F
            t = torch.from_numpy(np.ascontiguousarray(image)).permute(2, 0, 1).unsqueeze(0)
            t = F.interpolate(t, size=(height, width), mode="bicubic", align_corners=False)
            image = t.squeeze(0).permute(1, 2, 0).numpy()
        sample["image"] = image
        return sample


class NormalizeImage:
    """(image - mean) / std, per channel."""

    def __init__(self, mean, std):
        self._mean, self._std = mean, std

    def __call__(self, sample):
        sample["image"] = (sample["image"] - self._mean) / self._std
        return sample


class PrepareForNet:
    def __call__(self, sample):
        image = np.transpose(sample["image"], (2, 0, 1))
        sample["image"] = np.ascontiguousarray(image).astype(np.float32)
        return sample


class Compose:
    """torchvision.transforms.Compose stand-in (torchvision is not installed here)."""

    def __init__(self, transforms):
        self.transforms = transforms

    def __call__(self, x):
        for t in self.transforms:
            x = t(x)
        return x
