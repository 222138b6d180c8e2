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
    def __init__(self, width, height, resize_target=True, keep_aspect_ratio=False, ensure_multiple_of=1,
                 resize_method="lower_bound", image_interpolation_method=INTER_AREA):
        self.__width = width
        self.__height = height
        self.__resize_target = resize_target
        self.__keep_aspect_ratio = keep_aspect_ratio
        self.__multiple_of = ensure_multiple_of
        self.__resize_method = resize_method
        self.__image_interpolation_method = image_interpolation_method

    def constrain_to_multiple_of(self, x, min_val=0, max_val=None):
        y = (np.round(x / self.__multiple_of) * self.__multiple_of).astype(int)
        if max_val is not None and y > max_val:
            y = (np.floor(x / self.__multiple_of) * self.__multiple_of).astype(int)
        if y < min_val:
            y = (np.ceil(x / self.__multiple_of) * self.__multiple_of).astype(int)
        return y

    def get_size(self, width, height):
        scale_height = self.__height / height
        scale_width = self.__width / width
        if self.__keep_aspect_ratio:
            if self.__resize_method == "lower_bound":
                # scale such that output size is lower bound
                if scale_width > scale_height:
                    scale_height = scale_width
                else:
                    scale_width = scale_height
            elif self.__resize_method == "upper_bound":
                if scale_width < scale_height:
                    scale_height = scale_width
                else:
                    scale_width = scale_height
            elif self.__resize_method == "minimal":
                # scale as little as possible
                if abs(1 - scale_width) < abs(1 - scale_height):
                    scale_height = scale_width
                else:
                    scale_width = scale_height
            else:
                raise ValueError(f"resize_method {self.__resize_method} not implemented")
        if self.__resize_method == "lower_bound":
            new_height = self.constrain_to_multiple_of(scale_height * height, min_val=self.__height)
            new_width = self.constrain_to_multiple_of(scale_width * width, min_val=self.__width)
        elif self.__resize_method == "upper_bound":
            new_height = self.constrain_to_multiple_of(scale_height * height, max_val=self.__height)
            new_width = self.constrain_to_multiple_of(scale_width * width, max_val=self.__width)
        elif self.__resize_method == "minimal":
            new_height = self.constrain_to_multiple_of(scale_height * height)
            new_width = self.constrain_to_multiple_of(scale_width * width)
        else:
            raise ValueError(f"resize_method {self.__resize_method} not implemented")
        return int(new_width), int(new_height)

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
    def __init__(self, mean, std):
        self.__mean = mean
        self.__std = std

    def __call__(self, sample):
        sample["image"] = (sample["image"] - self.__mean) / self.__std
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
