"""The data contract on either side of the hot path: HIVE's on-disk dataset format and the key-frame
selection that feeds TSDF fusion (/root/reference/hive/io.py:533-572, 713-790, 866-1189).

Only what the dense path reads is mirrored: ``ImageFolderDataset``, ``DatasetMetadata`` (the fields the path
uses), ``HiveDataset`` (folder layout, camera files, depth transform, background selection) and
``HiveDataset.select_key_frames`` -- whose inner loops (unprojection, projection, visibility bbox) run on
the GPU through the C ABI.
"""
import ctypes
import json
import logging
import os
from os.path import join as pjoin
from typing import List, Optional

import numpy as np
from PIL import Image

from hive_amd import _lib
from hive_amd._lib import MEM_DEVICE, MEM_HOST, ptr
from hive_amd.geometric import Trajectory, get_pose_components, pose_vec2mat


class ImageFolderDataset:
    """The image files of a folder, in sorted order, as numpy arrays (the reference's loader of the same name, io.py:533-572, whose behaviour the
    dataset code relies on): 32-bit integer PNGs come back as 16-bit ('I' -> 'I;16'), 16-bit and 8-bit grey images keep their type, every
    other mode is converted to RGB; an optional ``transform`` is applied to the array."""
    _KEEP_MODES = frozenset(('L', 'I;16'))

    def __init__(self, base_dir, transform=None):
        assert os.path.isdir(base_dir), f"Could not find the folder: {base_dir}"
        names = sorted(os.listdir(base_dir))
        assert names, f"No files found in the folder: {base_dir}"
        self.base_dir, self.transform = base_dir, transform
        self.image_filenames = names
        self.image_paths = [pjoin(base_dir, name) for name in names]

    def __len__(self):
        return len(self.image_paths)

    @classmethod
    def _decode(cls, path) -> np.ndarray:
        if path.endswith('.raw'):
            raise NotImplementedError("raw float32 depth (COLMAP) is outside the hot path")
        with Image.open(path) as image:
            if image.mode == 'I':
                image = image.convert('I;16')
            elif image.mode not in cls._KEEP_MODES:
                image = image.convert('RGB')
            return np.asarray(image)

    def __getitem__(self, idx) -> np.ndarray:
        array = self._decode(self.image_paths[idx])
        return self.transform(array) if self.transform else array


class DatasetMetadata:
    """The subset of the reference's metadata.json that the hot path reads (io.py:713-790)."""

    def __init__(self, num_frames: int, fps: float, width: int, height: int, estimate_pose: bool = False, estimate_depth: bool = False,
                 depth_scale: float = 1. / 1000., max_depth: float = 10.0, depth_mask_dilation_iterations: int = 10, frame_step: int = 1,
                 **extra):
        if not isinstance(num_frames, int) or num_frames < 1:
            raise ValueError(f"num_frames must be a positive integer, got {num_frames}.")
        if width < 1 or height < 1:
            raise ValueError(f"width and height must be positive, got {width}x{height}.")
        self.num_frames = num_frames
        self.fps = fps
        self.frame_step = frame_step
        self.width = width
        self.height = height
        self.depth_scale = depth_scale
        self.max_depth = max_depth
        self.depth_mask_dilation_iterations = depth_mask_dilation_iterations
        self.estimate_pose = estimate_pose
        self.estimate_depth = estimate_depth
        self.extra = extra

    def to_json(self) -> dict:
        d = {k: v for k, v in self.__dict__.items() if k != "extra"}
        d.update(self.extra)
        return d

    def save(self, path):
        with open(path, 'w') as f:
            json.dump(self.to_json(), f)

    @staticmethod
    def load(path) -> 'DatasetMetadata':
        with open(path, 'r') as f:
            return DatasetMetadata(**json.load(f))

    def __eq__(self, other):
        return isinstance(other, DatasetMetadata) and self.to_json() == other.to_json()

    def __repr__(self):
        return f"DatasetMetadata({self.to_json()})"


class HiveDataset:
    """The main dataset format of HIVE (io.py:866-1189): rgb/ depth/ mask/ (+ *_inpainted/), metadata.json,
    camera_matrix.txt (3 x 3), camera_trajectory.txt (N x 7, xyzw quaternion + t, world-to-camera)."""

    metadata_filename = "metadata.json"
    camera_matrix_filename = "camera_matrix.txt"
    camera_trajectory_filename = "camera_trajectory.txt"
    required_files = [metadata_filename, camera_trajectory_filename, camera_matrix_filename]
    rgb_folder, depth_folder, mask_folder = "rgb", "depth", "mask"
    inpainted_rgb_folder, inpainted_depth_folder, inpainted_mask_folder = "rgb_inpainted", "depth_inpainted", "mask_inpainted"
    required_folders = [rgb_folder, depth_folder, mask_folder]
    depth_scaling_factor = 1. / 1000.  # mm -> m

    def __init__(self, base_path):
        self.base_path = str(base_path)
        self._validate_dataset()
        self.metadata = DatasetMetadata.load(pjoin(self.base_path, self.metadata_filename))
        self.camera_matrix, self.camera_trajectory = self._load_camera_parameters()
        self.rgb_dataset = ImageFolderDataset(pjoin(self.base_path, self.rgb_folder))
        self.depth_dataset = ImageFolderDataset(pjoin(self.base_path, self.depth_folder), transform=self._get_depth_map_transform())
        self.mask_dataset = ImageFolderDataset(pjoin(self.base_path, self.mask_folder))
        self.inpainted_rgb_dataset, self.inpainted_depth_dataset = self._get_inpainted_frame_data()

    @classmethod
    def is_valid_folder_structure(cls, path) -> bool:
        """True when `path` holds the files and folders of a HIVE dataset (io.py `InvalidDatasetFormatError` check)."""
        path = str(path)
        return (os.path.isdir(path) and all(os.path.isfile(pjoin(path, f)) for f in cls.required_files)
                and all(os.path.isdir(pjoin(path, f)) for f in cls.required_folders))

    def _validate_dataset(self):
        if not os.path.isdir(self.base_path):
            raise RuntimeError(f"The folder {self.base_path} does not exist.")
        for name in self.required_files:
            if not os.path.isfile(pjoin(self.base_path, name)):
                raise RuntimeError(f"The dataset {self.base_path} is missing the file {name}.")
        for name in self.required_folders:
            if not os.path.isdir(pjoin(self.base_path, name)):
                raise RuntimeError(f"The dataset {self.base_path} is missing the folder {name}.")

    def _get_depth_map_transform(self):
        def transform(depth_map):  # io.py:1032-1039
            depth_map = self.depth_scaling_factor * depth_map.astype(np.float32)
            depth_map[depth_map > self.metadata.max_depth] = 0.0
            return depth_map
        return transform

    def _get_inpainted_frame_data(self):
        paths = [pjoin(self.base_path, f) for f in (self.inpainted_rgb_folder, self.inpainted_depth_folder, self.inpainted_mask_folder)]
        if not all(os.path.isdir(p) for p in paths):
            return None, None
        rgb = ImageFolderDataset(paths[0])
        depth = ImageFolderDataset(paths[1], transform=self._get_depth_map_transform())
        if len(rgb) != self.num_frames or len(depth) != self.num_frames:
            raise RuntimeError(f"Expected inpainted frame data to have {self.num_frames} frames, but got {len(rgb)} and {len(depth)}")
        return rgb, depth

    def _load_camera_parameters(self):
        camera_matrix = np.loadtxt(pjoin(self.base_path, self.camera_matrix_filename), dtype=np.float32)
        camera_trajectory = Trajectory.load(pjoin(self.base_path, self.camera_trajectory_filename))
        if camera_matrix.shape != (3, 3):
            raise RuntimeError(f"Expected camera matrix to be a 3x3 matrix, but got {camera_matrix.shape} instead.")
        if len(camera_trajectory.shape) != 2 or camera_trajectory.shape[1] != 7:
            raise RuntimeError(f"Expected camera trajectory to be a Nx7 matrix, but got {camera_trajectory.shape} instead.")
        return camera_matrix, camera_trajectory

    @property
    def bg_rgb_dataset(self):
        return self.inpainted_rgb_dataset or self.rgb_dataset

    @property
    def bg_depth_dataset(self):
        return self.inpainted_depth_dataset or self.depth_dataset

    @property
    def has_inpainted_frame_data(self) -> bool:
        return self.inpainted_rgb_dataset is not None and self.inpainted_depth_dataset is not None

    @property
    def num_frames(self) -> int:
        return self.metadata.num_frames

    @property
    def frame_width(self) -> int:
        return self.metadata.width

    @property
    def frame_height(self) -> int:
        return self.metadata.height

    def __len__(self):
        return self.num_frames

    @staticmethod
    def index_to_filename(index: int, file_extension="png") -> str:
        return f"{index:06d}.{file_extension}"

    def select_key_frames(self, threshold=0.3, frame_step=30) -> List[int]:
        return select_key_frames(self, threshold=threshold, frame_step=frame_step)


def select_key_frames(dataset, threshold=0.3, frame_step=30, ctx=None) -> List[int]:
    """Greedy key-frame set: a sampled frame joins the set unless the bounding box of its point cloud, projected
    into some key frame, covers at least ``threshold`` of the image (io.py:1117-1189).

    The point cloud of the candidate (``point_cloud_from_depth``) stays in HBM; for every key frame one kernel
    projects it (``world2image``, np.round) and reduces the visible pixels to their bounding box, so only five
    integers per (candidate, key frame) pair come back to the host.
    """
    logging.info(f"Selecting key frames (threshold={threshold})...")
    if not (0.0 <= threshold <= 1.0):
        raise ValueError(f"Threshold must be a real number between zero and one (inclusive), but got {threshold}.")
    if threshold == 0.0:
        return [0]
    elif threshold == 1.0:
        return list(range(dataset.num_frames))
    if threshold > 0.8:
        logging.warning("Setting the key frame threshold to a high value (> 0.8) may result in long runtimes.")
    if frame_step < 1:
        raise ValueError(f"Frame step must be a positive integer, but got {frame_step} instead.")

    import torch
    ctx = ctx or _lib.default_context()
    width, height = dataset.metadata.width, dataset.metadata.height
    K = dataset.camera_matrix
    K64 = np.ascontiguousarray(K, dtype=np.float64)
    Kinv = np.ascontiguousarray(np.linalg.inv(K), dtype=np.float64)  # inverted in K's dtype, as image2world does
    key_frames = [0]
    points_dev = torch.empty((height * width, 3), dtype=torch.float64, device="cuda")

    def pose_of(frame):
        R, t = get_pose_components(pose_vec2mat(dataset.camera_trajectory[frame]))
        return np.ascontiguousarray(R, dtype=np.float64), np.ascontiguousarray(t, dtype=np.float64).reshape(3)

    for frame in range(1, dataset.num_frames, frame_step):
        depth = torch.from_numpy(np.ascontiguousarray(dataset.bg_depth_dataset[frame], dtype=np.float32)).cuda()
        mask = torch.from_numpy(np.ascontiguousarray(np.asarray(dataset.mask_dataset[frame]) == 0, dtype=np.uint8)).cuda()
        R, t = pose_of(frame)
        n = ctypes.c_int64(0)
        ctx.check(ctx.lib.hive_unproject(ctx.handle, depth.data_ptr(), mask.data_ptr(), None, depth.shape[0], depth.shape[1], ptr(Kinv), ptr(R),
                                         ptr(t), MEM_DEVICE, points_dev.data_ptr(), None, points_dev.shape[0], ctypes.byref(n)))
        for key_frame in key_frames:
            Rk, tk = pose_of(key_frame)
            box = np.zeros(5, np.int32)
            ctx.check(ctx.lib.hive_project_bbox(ctx.handle, points_dev.data_ptr(), n.value, ptr(K64), ptr(Rk), ptr(tk), width, height,
                                                MEM_DEVICE, ptr(box)))
            if box[4] == 0:  # no visible point
                continue
            visible_area = int(box[1] - box[0]) * int(box[3] - box[2])
            overlap_ratio = visible_area / (width * height)
            if overlap_ratio >= threshold:
                logging.debug(f"Excluding frame {frame} from key frames: overlap with key frame {key_frame} is {overlap_ratio:.2f}.")
                break
        else:
            key_frames.append(frame)
    logging.debug(f"Selected key frames: {key_frames}.")
    return key_frames
