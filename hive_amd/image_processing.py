"""``dilate_mask`` of /root/reference/hive/image_processing.py:30-45 on the MI355X."""
import numpy as np

from hive_amd import _lib
from hive_amd._lib import MEM_HOST, ptr
from hive_amd.options import MaskDilationOptions
from hive_amd.utils import validate_shape


def dilate_mask(mask, dilation_options: MaskDilationOptions):
    """Dilate an instance segmentation mask so that it covers a larger area.

    The reference runs ``cv2.dilate(mask, dilation_options.filter, iterations=num_iterations)``.  With the default 3x3 rectangle
    that equals one (2n+1) x (2n+1) box maximum with out-of-image pixels ignored (two separable launches); any other structuring
    element (``MaskDilationOptions(dilation_filter=...)``, /root/reference/hive/options.py:245-268) is iterated literally on the
    device with cv2's definition: anchor at the element's centre, taps outside the image ignored.

    :return: The dilated mask (bool).
    """
    mask = np.asarray(mask)
    validate_shape(mask, 'mask', expected_shape=(None, None))
    se = dilation_options.structuring_element()
    mask_u8 = np.ascontiguousarray(mask.astype(np.float32) != 0, dtype=np.uint8)
    out = np.empty_like(mask_u8)
    ctx = _lib.default_context()
    ctx.check(ctx.lib.hive_dilate_mask_se(ctx.handle, ptr(mask_u8), mask_u8.shape[0], mask_u8.shape[1], ptr(se), se.shape[0], se.shape[1],
                                          int(dilation_options.num_iterations), MEM_HOST, ptr(out)))
    return out.astype(bool)
