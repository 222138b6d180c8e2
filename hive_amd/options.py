"""The option objects the hot path reads, with the reference's names, defaults and checks
(/root/reference/hive/options.py:44-67, 245-268, 310-439)."""
import abc
import argparse
import enum
from typing import Dict, Optional

import numpy as np


class Options(abc.ABC):
    """Base of the option groups: each can register its CLI flags and rebuild itself from them."""

    @staticmethod
    @abc.abstractmethod
    def add_args(parser: argparse.ArgumentParser):
        raise NotImplementedError

    @staticmethod
    @abc.abstractmethod
    def from_args(args: argparse.Namespace):
        raise NotImplementedError

    def __repr__(self):
        return f"{self.__class__.__name__}({', '.join(f'{k}={v!r}' for k, v in self.__dict__.items())})"

    def __eq__(self, other):
        return type(self) is type(other) and all(np.array_equal(v, other.__dict__.get(k)) for k, v in self.__dict__.items())


def _default_filter():
    # cv2.getStructuringElement(cv2.MORPH_RECT, (3, 3)) -- options.py:248
    return np.ones((3, 3), dtype=np.uint8)


class MaskDilationOptions(Options):
    """Options for `dilate_mask` (options.py:245-268)."""

    def __init__(self, num_iterations=0, dilation_filter=None):
        self.num_iterations = num_iterations
        self.filter = _default_filter() if dilation_filter is None else dilation_filter

    @property
    def is_default_filter(self):
        return np.array_equal(np.asarray(self.filter), _default_filter())

    @staticmethod
    def add_args(parser: argparse.ArgumentParser):
        group = parser.add_argument_group('Mask Dilation Options')
        group.add_argument('--dilate_mask_iter', type=int, default=0,
                           help='The number of times to run a dilation filter over the object masks. A higher number '
                                'results in larger masks and zero results in the original mask.')

    @staticmethod
    def from_args(args) -> 'MaskDilationOptions':
        return MaskDilationOptions(num_iterations=args.dilate_mask_iter)


class MeshFilteringOptions(Options):
    """Limits of the per-frame face filter (/root/reference/hive/options.py:271-306): a face survives when every edge is at
    most ``max_pixel_distance`` pixels long in image space and spans at most ``max_depth_distance`` metres of depth."""

    def __init__(self, max_pixel_distance=2, max_depth_distance=0.1, min_num_components=5):
        self.max_pixel_distance = max_pixel_distance
        self.max_depth_distance = max_depth_distance
        self.min_num_components = min_num_components

    @staticmethod
    def add_args(parser: argparse.ArgumentParser):
        group = parser.add_argument_group('Mesh Filtering Options')
        group.add_argument('--max_depth_dist', type=float, default=0.1, help='largest depth difference between the vertices of a face')
        group.add_argument('--max_pixel_dist', type=float, default=2, help='largest image-space distance between the vertices of a face')
        group.add_argument('--min_num_components', type=float, default=5, help='fragments with fewer connected faces are culled')

    @staticmethod
    def from_args(args) -> 'MeshFilteringOptions':
        return MeshFilteringOptions(max_pixel_distance=args.max_pixel_dist, max_depth_distance=args.max_depth_dist,
                                    min_num_components=args.min_num_components)


class MeshReconstructionMethod(enum.Enum):
    TSDFFusion = enum.auto()
    BundleFusion = enum.auto()
    RGBD = enum.auto()

    @classmethod
    def get_cli_names(cls) -> Dict['MeshReconstructionMethod', str]:
        return {cls.TSDFFusion: 'tsdf_fusion', cls.BundleFusion: 'bundle_fusion', cls.RGBD: 'rgbd'}

    @classmethod
    def get_choices(cls):
        return {name: method for method, name in cls.get_cli_names().items()}

    def get_cli_name(self) -> str:
        return self.get_cli_names()[self]

    @classmethod
    def from_string(cls, name):
        choices = cls.get_choices()
        if name.lower() in choices:
            return choices[name.lower()]
        raise RuntimeError(f"No method called {name}, valid choices are: {list(choices.keys())}")


class BackgroundMeshOptions(Options):
    """Static-scene reconstruction options (options.py:353-439); only TSDFFusion is implemented here."""
    supported_reconstruction_methods = [MeshReconstructionMethod.TSDFFusion, MeshReconstructionMethod.BundleFusion,
                                        MeshReconstructionMethod.RGBD]

    def __init__(self, reconstruction_method=MeshReconstructionMethod.TSDFFusion, depth_mask_dilation_iterations=10,
                 sdf_volume_size=5.0, sdf_voxel_size=0.005, sdf_max_voxels: Optional[int] = 320_000_000,
                 key_frame_threshold=0.3, key_frame_step=30):
        assert reconstruction_method in self.supported_reconstruction_methods, \
            f"Reconstruction method must be one of the following: " \
            f"{[m.name for m in self.supported_reconstruction_methods]}, but got {reconstruction_method} instead."
        assert depth_mask_dilation_iterations >= 0 and isinstance(depth_mask_dilation_iterations, int), \
            f"The depth mask dilation iterations must be a positive integer."
        assert sdf_volume_size > 0.0, f"Volume size must be a positive number, instead got {sdf_volume_size}"
        assert sdf_voxel_size > 0.0, f"Voxel size must be a positive number, instead got {sdf_voxel_size}"
        assert sdf_max_voxels is None or (isinstance(sdf_max_voxels, int) and sdf_max_voxels > 0), \
            f"Number of voxels number must be a positive integer or None, instead got {sdf_max_voxels}"
        if not (0.0 <= key_frame_threshold <= 1.0):
            raise ValueError(f"Key frame threshold must be between zero and one (inclusive), but got {key_frame_threshold}.")
        assert isinstance(key_frame_step, int) and key_frame_step > 1, \
            f"Key frame step must be a positive integer, but got {key_frame_step}."

        self.reconstruction_method = reconstruction_method
        self.depth_mask_dilation_iterations = depth_mask_dilation_iterations
        self.sdf_volume_size = sdf_volume_size
        self.sdf_voxel_size = sdf_voxel_size
        self.sdf_max_voxels = sdf_max_voxels
        self.key_frame_threshold = key_frame_threshold
        self.key_frame_step = key_frame_step

    @staticmethod
    def add_args(parser: argparse.ArgumentParser):
        group = parser.add_argument_group('Static Mesh Options')
        group.add_argument('--mesh_reconstruction_method', type=str, default='tsdf_fusion',
                           choices=[m.get_cli_name() for m in BackgroundMeshOptions.supported_reconstruction_methods],
                           help="The method to use for reconstructing the static mesh.")
        group.add_argument('--depth_mask_dilation_iterations', type=int, default=10,
                           help="The number of times to dilate the dynamic object masks for masking the depth maps.")
        group.add_argument('--sdf_volume_size', type=float, default=5.0, help="The size of the SDF volume in cubic meters.")
        group.add_argument('--sdf_voxel_size', type=float, default=0.005, help="The size of a voxel in the SDF volume.")
        group.add_argument('--sdf_max_voxels', type=int, default=320_000_000,
                           help="The maximum number of voxels allowed in the resulting voxel volume.")
        group.add_argument('--key_frame_threshold', type=float, default=0.3,
                           help="The maximum overlap ratio before a frame is excluded from the key frame set.")
        # the reference parses this flag as float and then fails its own int assert (options.py:424 vs :390);
        # parsed as int here so that the flag is usable
        group.add_argument('--key_frame_step', type=int, default=30,
                           help="The frequency to sample frames at for key frame selection.")

    @staticmethod
    def from_args(args: argparse.Namespace) -> 'BackgroundMeshOptions':
        return BackgroundMeshOptions(
            reconstruction_method=MeshReconstructionMethod.from_string(args.mesh_reconstruction_method),
            depth_mask_dilation_iterations=args.depth_mask_dilation_iterations,
            sdf_volume_size=args.sdf_volume_size,
            sdf_voxel_size=args.sdf_voxel_size,
            sdf_max_voxels=args.sdf_max_voxels,
            key_frame_threshold=args.key_frame_threshold,
            key_frame_step=args.key_frame_step,
        )
