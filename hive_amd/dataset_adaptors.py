"""Dataset adaptors: the API surface of /root/reference/hive/dataset_adaptors.py for the inputs of the hot
path.  ``TUMAdaptor`` (RGB-D + ground-truth poses, BASELINE.json config 1) is implemented; ``UnrealAdaptor``,
``VideoAdaptor`` and ``StrayScannerAdaptor`` (dataset_adaptors.py:769, 1023, 1158) need COLMAP / ffmpeg / capture
formats that are outside the dense-compute scope ("KEEP API, no logic"): their names, folder
checks and constructor signatures are here so that ``from hive.dataset_adaptors import ...`` lines and ``get_dataset``'s
dispatch keep working, and ``convert`` says what is missing instead of doing the work.
``estimate_depth_dpt`` is re-exported from ``hive_amd.depth``.
"""
import enum
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
from hive_amd.options import BackgroundMeshOptions, COLMAPOptions, InpaintingMode, PipelineOptions, StorageOptions


class DatasetAdaptor:
    """Base of the converters to the HIVE format (dataset_adaptors.py:57-572): ``convert()`` writes the HIVE folder."""
    required_files: list = []
    required_folders: list = []

    def __init__(self, base_path, output_path, num_frames=-1, frame_step=1, colmap_options=None):
        self.base_path, self.output_path = str(base_path), str(output_path)
        self.num_frames, self.frame_step = num_frames, frame_step
        self.colmap_options = colmap_options or COLMAPOptions()

    @classmethod
    def is_valid_folder_structure(cls, path) -> bool:
        path = str(path)
        return (os.path.isdir(path) and all(os.path.isfile(pjoin(path, f)) for f in cls.required_files)
                and all(os.path.isdir(pjoin(path, f)) for f in cls.required_folders))

    def convert(self, estimate_pose, estimate_depth, inpainting_mode=InpaintingMode.Off, static_camera=False, no_cache=False, profiling=None):
        raise NotImplementedError(f"{type(self).__name__}.convert: this capture format needs stages outside the dense-compute path of "
                                  f"this build (COLMAP / ffmpeg / capture-specific decoding, SURVEY.md section 2 row 11); convert the "
                                  f"sequence with the reference and open the resulting HIVE folder, or use a TUM-layout folder")


class TUMAdaptor(DatasetAdaptor):
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

    def __init__(self, base_path, output_path, num_frames=-1, frame_step=1, is_16_bit=True, colmap_options=None):
        self.base_path, self.output_path = str(base_path), str(output_path)
        self.colmap_options = colmap_options or COLMAPOptions()
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

    def convert(self, estimate_pose=False, estimate_depth=False, inpainting_mode=InpaintingMode.Off, static_camera=False, no_cache=False,
                profiling=None) -> HiveDataset:
        """Write the HIVE-format folder and return it as a ``HiveDataset`` (dataset_adaptors.py:176-266).
        Instance masks need detectron2 (out of scope): empty masks are written, i.e. a static scene.
        ``estimate_depth=True`` replaces the sensor depth by DPT-Hybrid estimates (needs the weights file)."""
        if estimate_pose:
            raise NotImplementedError("pose estimation runs COLMAP, which is outside the dense-compute scope")
        if inpainting_mode != InpaintingMode.Off or static_camera:
            raise NotImplementedError("inpainting (LaMa / cv2) and the static-camera override are outside the dense-compute scope")
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


class UnrealAdaptor(DatasetAdaptor):
    """Unreal Engine captures (dataset_adaptors.py:769-851): API only."""
    metadata_filename, camera_matrix_filename, camera_trajectory_filename = "info.json", "camera.txt", "trajectory.txt"
    required_files = [metadata_filename, camera_matrix_filename, camera_trajectory_filename]
    rgb_folder, depth_folder = "colour", "depth"
    required_folders = [rgb_folder, depth_folder]


class VideoAdaptorBase(DatasetAdaptor):
    """Common part of the adaptors that read frames from a video file (dataset_adaptors.py:854-1020): API only."""

    def __init__(self, base_path, output_path, video_path, num_frames=-1, frame_step=1, colmap_options=None, resize_to=640):
        super().__init__(base_path, output_path, num_frames=num_frames, frame_step=frame_step, colmap_options=colmap_options)
        self.video_path, self.resize_to = str(video_path), resize_to


class VideoAdaptor(VideoAdaptorBase):
    """A plain video file (poses from COLMAP, depth from DPT; dataset_adaptors.py:1023-1091): API only."""

    def __init__(self, base_path, output_path, num_frames=-1, frame_step=1, colmap_options=None, resize_to=640):
        path = str(base_path)
        super().__init__(os.path.dirname(path), output_path, video_path=path, num_frames=num_frames, frame_step=frame_step, colmap_options=colmap_options,
                         resize_to=resize_to)

    @classmethod
    def is_valid_folder_structure(cls, path) -> bool:
        path = str(path)
        return os.path.isfile(path) and os.path.splitext(path)[1] == ".mp4"  # (dataset_adaptors.py:1059)


class DeviceOrientation(enum.Enum):
    """How a StrayScanner capture was held (dataset_adaptors.py:1094-1155)."""
    Landscape = enum.auto()
    Portrait = enum.auto()
    LandscapeReverse = enum.auto()
    PortraitReverse = enum.auto()


class StrayScannerAdaptor(VideoAdaptorBase):
    """iOS StrayScanner captures (LiDAR depth + ARKit odometry; dataset_adaptors.py:1158-1335): API only."""
    video_filename, camera_matrix_filename, camera_trajectory_filename = "rgb.mp4", "camera_matrix.csv", "odometry.csv"
    required_files = [video_filename, camera_matrix_filename, camera_trajectory_filename]
    depth_folder, confidence_map_folder = "depth", "confidence"
    required_folders = [depth_folder, confidence_map_folder]

    def __init__(self, base_path, output_path, num_frames=-1, frame_step=1, colmap_options=None, resize_to=640, depth_confidence_filter_level=0,
                 fix_orientation=True):
        super().__init__(base_path, output_path, video_path=pjoin(str(base_path), self.video_filename), num_frames=num_frames, frame_step=frame_step,
                         colmap_options=colmap_options, resize_to=resize_to)
        self.depth_confidence_filter_level, self.fix_orientation = depth_confidence_filter_level, fix_orientation


def get_dataset(storage_options, colmap_options=None, pipeline_options=None, resize_to=640, depth_confidence_filter_level=0, profiling=None,
                **legacy) -> HiveDataset:
    """Open a HIVE dataset, converting it first if it is in another format (dataset_adaptors.py:1438-1498).

    Reference form: ``get_dataset(storage_options: StorageOptions, colmap_options, pipeline_options, resize_to, ...)``.  The
    shorthand of this build's earlier rounds still works: ``get_dataset(dataset_path, output_path, num_frames=-1, frame_step=1,
    estimate_depth=False, no_cache=False)``."""
    if not isinstance(storage_options, StorageOptions):  # shorthand: (dataset_path, output_path, ...)
        output_path = legacy.pop("output_path", colmap_options)
        pipeline_options = PipelineOptions(num_frames=legacy.pop("num_frames", -1), frame_step=legacy.pop("frame_step", 1),
                                           estimate_depth=legacy.pop("estimate_depth", False))
        storage_options = StorageOptions(dataset_path=storage_options, output_path=output_path, no_cache=legacy.pop("no_cache", False))
        colmap_options = None
        if HiveDataset.is_valid_folder_structure(storage_options.dataset_path):  # (the shorthand opens a HIVE folder in place)
            return HiveDataset(storage_options.dataset_path)
    assert not legacy, f"get_dataset: unexpected arguments {sorted(legacy)}"
    colmap_options = colmap_options or COLMAPOptions()
    pipeline_options = pipeline_options or PipelineOptions()
    dataset_path, output_path = str(storage_options.dataset_path), str(storage_options.output_path)
    if not storage_options.no_cache and HiveDataset.is_valid_folder_structure(output_path):
        return HiveDataset(output_path)
    base = dict(base_path=dataset_path, output_path=output_path, num_frames=pipeline_options.num_frames, frame_step=pipeline_options.frame_step,
                colmap_options=colmap_options)
    if HiveDataset.is_valid_folder_structure(dataset_path):
        return HiveDataset(dataset_path)
    if TUMAdaptor.is_valid_folder_structure(dataset_path):
        converter = TUMAdaptor(**base)
    elif UnrealAdaptor.is_valid_folder_structure(dataset_path):
        converter = UnrealAdaptor(**base)
    elif StrayScannerAdaptor.is_valid_folder_structure(dataset_path):
        converter = StrayScannerAdaptor(**base, resize_to=resize_to, depth_confidence_filter_level=depth_confidence_filter_level,
                                        fix_orientation=not pipeline_options.estimate_pose)
    elif VideoAdaptor.is_valid_folder_structure(dataset_path):
        converter = VideoAdaptor(resize_to=resize_to, **base)
    elif not os.path.isdir(dataset_path):
        raise RuntimeError(f"Could not open the path {dataset_path} or it is not a folder.")
    else:
        raise RuntimeError(f"Could not recognise the dataset format for the dataset at {dataset_path}.")
    return converter.convert(estimate_pose=pipeline_options.estimate_pose, estimate_depth=pipeline_options.estimate_depth,
                             inpainting_mode=pipeline_options.inpainting_mode, static_camera=pipeline_options.static_camera,
                             no_cache=storage_options.no_cache, profiling=profiling)
