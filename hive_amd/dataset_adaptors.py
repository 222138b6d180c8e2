"""Dataset adaptors: the API surface of /root/reference/hive/dataset_adaptors.py for the inputs of the hot
path.  ``TUMAdaptor`` (RGB-D + ground-truth poses, BASELINE.json config 1) is implemented; the adaptors that
need COLMAP / ffmpeg / Unreal / StrayScanner captures are outside the dense-compute scope.
``estimate_depth_dpt`` is re-exported from ``hive_amd.depth``.
"""
import logging
import os
import shutil
from os.path import join as pjoin

import numpy as np
from PIL import Image
from scipy.spatial.transform import Rotation

from hive_amd.depth import estimate_depth_dpt  # noqa: F401  (reference location: dataset_adaptors.py:1346)
from hive_amd.geometric import Trajectory
from hive_amd.io import DatasetMetadata, HiveDataset, ImageFolderDataset
from hive_amd.options import BackgroundMeshOptions


class TUMAdaptor:
    """Converts a TUM RGB-D sequence (rgb.txt, depth.txt, groundtruth.txt, rgb/, depth/) to the HIVE format
    (dataset_adaptors.py:573-760): frames are associated by nearest timestamp per depth map, the cam-to-world
    ground-truth poses are re-based (``normalise_position``), inverted to world-to-camera and rotated -90 degrees
    about x; depth PNGs (1/5000 m units) are rewritten in millimetres."""
    fx, fy, cx, cy = 580.0, 580.0, 319.5, 239.5
    width, height = 640, 480
    intrinsic_matrix = np.array([[fx, 0., cx], [0., fy, cy], [0., 0., 1.]])
    fps = 30.0
    pose_path, rgb_files_path, depth_map_files_path = "groundtruth.txt", "rgb.txt", "depth.txt"
    required_files = [pose_path, rgb_files_path, depth_map_files_path]
    rgb_folder, depth_folder = "rgb", "depth"
    required_folders = [rgb_folder, depth_folder]

    def __init__(self, base_path, output_path, num_frames=-1, frame_step=1, is_16_bit=True):
        self.base_path, self.output_path = str(base_path), str(output_path)
        for name in self.required_files:
            if not os.path.isfile(pjoin(self.base_path, name)):
                raise RuntimeError(f"The TUM dataset {self.base_path} is missing the file {name}.")
        for name in self.required_folders:
            if not os.path.isdir(pjoin(self.base_path, name)):
                raise RuntimeError(f"The TUM dataset {self.base_path} is missing the folder {name}.")
        self.frame_step = frame_step
        self.depth_scale_factor = 1.0 / 5000.0 if is_16_bit else 1.0
        self.image_filenames, self.depth_filenames, trajectory = self._get_synced_frame_data()
        full = len(self.image_filenames)
        self.num_frames = full if num_frames == -1 or num_frames > full else num_frames
        trajectory = trajectory.normalise_position().inverse()
        rotation = np.eye(4)
        rotation[:3, :3] = Rotation.from_euler('xyz', [-90, 0, 0], degrees=True).as_matrix()
        self.camera_trajectory = trajectory.apply(rotation)

    @staticmethod
    def _load_list(path):
        stamps, data = [], []
        with open(path, 'r') as f:
            for line in f:
                line = line.strip()
                if not line or line.startswith('#'):
                    continue
                parts = line.split(' ')
                stamps.append(float(parts[0]))
                data.append(parts[1:])
        return np.array(stamps), data

    def _get_synced_frame_data(self):
        image_t, image_paths = self._load_list(pjoin(self.base_path, self.rgb_files_path))
        depth_t, depth_paths = self._load_list(pjoin(self.base_path, self.depth_map_files_path))
        pose_t, poses = self._load_list(pjoin(self.base_path, self.pose_path))

        def closest(query, target):  # index of the closest query timestamp for every target timestamp
            return np.abs(query.reshape(-1, 1) - target.reshape(1, -1)).argmin(axis=0)

        images = [image_paths[i][0][len("rgb/"):] for i in closest(image_t, depth_t)]
        depths = [p[0][len("depth/"):] for p in depth_paths]
        rows = []
        for i in closest(pose_t, depth_t):
            tx, ty, tz, qx, qy, qz, qw = map(float, poses[i])
            rows.append((qx, qy, qz, qw, tx, ty, tz))
        return images, depths, Trajectory(np.array(rows))

    def get_metadata(self, estimate_pose=False, estimate_depth=False) -> DatasetMetadata:
        return DatasetMetadata(num_frames=self.num_frames, frame_step=self.frame_step, fps=self.fps, width=self.width, height=self.height,
                               estimate_pose=estimate_pose, estimate_depth=estimate_depth,
                               depth_mask_dilation_iterations=BackgroundMeshOptions().depth_mask_dilation_iterations,
                               depth_scale=HiveDataset.depth_scaling_factor)

    def depth_to_mm(self, raw):
        """TUM depth PNG values -> uint16 millimetres, in the reference's operation order (dataset_adaptors.py:762-764:
        metres first, then x 1000, then truncation) -- dividing by the two scale factors in one step rounds differently
        and is off by one millimetre for 41 of the 65536 raw values."""
        depth_map = np.asarray(raw) * self.depth_scale_factor  # convert to metres from non-standard scale & units.
        return (1000 * depth_map).astype(np.uint16)  # convert to mm from metres.

    def convert(self, estimate_pose=False, estimate_depth=False, no_cache=False) -> HiveDataset:
        """Write the HIVE-format folder and return it as a ``HiveDataset`` (dataset_adaptors.py:176-266).
        Instance masks need detectron2 (out of scope): empty masks are written, i.e. a static scene.
        ``estimate_depth=True`` replaces the sensor depth by DPT-Hybrid estimates (needs the weights file)."""
        if estimate_pose:
            raise NotImplementedError("pose estimation runs COLMAP, which is outside the dense-compute scope")
        out = self.output_path
        if no_cache and os.path.isdir(out):
            shutil.rmtree(out)
        os.makedirs(out, exist_ok=True)
        for folder in HiveDataset.required_folders:
            os.makedirs(pjoin(out, folder), exist_ok=True)
        self.get_metadata(estimate_pose, estimate_depth).save(pjoin(out, HiveDataset.metadata_filename))
        logging.info("Copying frames...")
        for i in range(self.num_frames):
            name = HiveDataset.index_to_filename(i)
            Image.open(pjoin(self.base_path, self.rgb_folder, self.image_filenames[i])).convert('RGB').save(pjoin(out, "rgb", name))
            Image.fromarray(np.zeros((self.height, self.width), np.uint8)).save(pjoin(out, "mask", name))
            if not estimate_depth:
                raw = np.asarray(Image.open(pjoin(self.base_path, self.depth_folder, self.depth_filenames[i])))
                depth_mm = self.depth_to_mm(raw)
                Image.fromarray(depth_mm).save(pjoin(out, "depth", name))  # uint16 -> 16-bit PNG
        if estimate_depth:
            estimate_depth_dpt(ImageFolderDataset(pjoin(out, "rgb")), pjoin(out, "depth"))
        np.savetxt(pjoin(out, HiveDataset.camera_matrix_filename), self.intrinsic_matrix)
        Trajectory(self.camera_trajectory.values[:self.num_frames]).save(pjoin(out, HiveDataset.camera_trajectory_filename))
        return HiveDataset(out)


def get_dataset(dataset_path, output_path, num_frames=-1, frame_step=1, estimate_depth=False, no_cache=False) -> HiveDataset:
    """Open ``dataset_path`` as a HIVE dataset, converting it first if it is a TUM sequence
    (the dispatch of dataset_adaptors.py:1438-1498, restricted to the formats implemented here)."""
    if all(os.path.isfile(pjoin(dataset_path, f)) for f in HiveDataset.required_files):
        return HiveDataset(dataset_path)
    if all(os.path.isfile(pjoin(dataset_path, f)) for f in TUMAdaptor.required_files):
        return TUMAdaptor(dataset_path, output_path, num_frames=num_frames, frame_step=frame_step).convert(estimate_depth=estimate_depth,
                                                                                                           no_cache=no_cache)
    raise RuntimeError(f"Could not recognise the dataset format for the dataset at {dataset_path}.")
