"""The option groups the hot path reads.  Public names, constructor arguments, defaults, CLI flags and the exception type each
bad value raises follow /root/reference/hive/options.py (:44-67 base class, :245-268 mask dilation, :271-306 mesh filtering,
:310-350 reconstruction method, :353-439 static mesh); the implementation is a flag table per group, from which the parser
arguments and `from_args` are both derived."""
import abc
import argparse
import enum
from typing import Dict, NamedTuple, Optional

import numpy as np


class _Flag(NamedTuple):
    """One CLI flag of an option group: `--flag`, the constructor argument it feeds, its type / default and its help text."""
    flag: str
    field: str
    type: type
    default: object
    help: str
    choices: Optional[tuple] = None


class Options(abc.ABC):
    """An option group: `add_args` registers its flags on a parser, `from_args` rebuilds the group from the parsed namespace.
    Groups that list their flags in `_flags` / `_title` inherit both."""
    _title: str = ""
    _flags: tuple = ()

    @classmethod
    def _register(cls, parser: argparse.ArgumentParser):
        section = parser.add_argument_group(cls._title)
        for f in cls._flags:
            extra = {"choices": list(f.choices)} if f.choices else {}
            section.add_argument(f"--{f.flag}", type=f.type, default=f.default, help=f.help, **extra)

    @classmethod
    def _collect(cls, args: argparse.Namespace) -> dict:
        return {f.field: getattr(args, f.flag) for f in cls._flags}

    @staticmethod
    @abc.abstractmethod
    def add_args(parser: argparse.ArgumentParser):
        raise NotImplementedError

    @staticmethod
    @abc.abstractmethod
    def from_args(args: argparse.Namespace):
        raise NotImplementedError

    def __repr__(self):
        body = ", ".join(f"{name}={value!r}" for name, value in vars(self).items())
        return f"{type(self).__name__}({body})"

    def __eq__(self, other):
        if type(self) is not type(other):
            return False
        theirs = vars(other)
        return all(np.array_equal(value, theirs.get(name)) for name, value in vars(self).items())


def _check(ok: bool, message: str):
    """The reference validates with `assert`; callers (and its tests) catch AssertionError, so that is what a bad value raises."""
    if not ok:
        raise AssertionError(message)


def _box3x3():
    # what cv2.getStructuringElement(cv2.MORPH_RECT, (3, 3)) returns (reference default, options.py:248)
    return np.ones((3, 3), dtype=np.uint8)


class MaskDilationOptions(Options):
    """How far `image_processing.dilate_mask` grows the instance masks."""
    _title = 'Mask Dilation Options'
    _flags = (_Flag('dilate_mask_iter', 'num_iterations', int, 0,
                    'how many passes of the 3x3 dilation to apply to the object masks (0 leaves them as they are; each pass '
                    'grows every mask by one pixel in each direction)'),)

    def __init__(self, num_iterations=0, dilation_filter=None):
        self.num_iterations = num_iterations
        self.filter = _box3x3() if dilation_filter is None else dilation_filter

    @property
    def is_default_filter(self):
        """True for the 3x3 box -- the only structuring element the HIP kernel implements."""
        return np.array_equal(np.asarray(self.filter), _box3x3())

    @staticmethod
    def add_args(parser: argparse.ArgumentParser):
        MaskDilationOptions._register(parser)

    @staticmethod
    def from_args(args) -> 'MaskDilationOptions':
        return MaskDilationOptions(**MaskDilationOptions._collect(args))


class MeshFilteringOptions(Options):
    """Limits of the per-frame face filter: a face survives when every edge is at most ``max_pixel_distance`` pixels long in
    image space and spans at most ``max_depth_distance`` metres of depth; fragments of fewer than ``min_num_components``
    connected faces are dropped afterwards."""
    _title = 'Mesh Filtering Options'
    _flags = (_Flag('max_depth_dist', 'max_depth_distance', float, 0.1, 'largest depth difference between the vertices of a face'),
              _Flag('max_pixel_dist', 'max_pixel_distance', float, 2, 'largest image-space distance between the vertices of a face'),
              _Flag('min_num_components', 'min_num_components', float, 5, 'fragments with fewer connected faces are culled'))

    def __init__(self, max_pixel_distance=2, max_depth_distance=0.1, min_num_components=5):
        self.max_pixel_distance = max_pixel_distance
        self.max_depth_distance = max_depth_distance
        self.min_num_components = min_num_components

    @staticmethod
    def add_args(parser: argparse.ArgumentParser):
        MeshFilteringOptions._register(parser)

    @staticmethod
    def from_args(args) -> 'MeshFilteringOptions':
        return MeshFilteringOptions(**MeshFilteringOptions._collect(args))


class MeshReconstructionMethod(enum.Enum):
    TSDFFusion = 1
    BundleFusion = 2
    RGBD = 3

    @classmethod
    def get_cli_names(cls) -> Dict['MeshReconstructionMethod', str]:
        return dict(zip((cls.TSDFFusion, cls.BundleFusion, cls.RGBD), ('tsdf_fusion', 'bundle_fusion', 'rgbd')))

    @classmethod
    def get_choices(cls) -> Dict[str, 'MeshReconstructionMethod']:
        return {cli_name: member for member, cli_name in cls.get_cli_names().items()}

    def get_cli_name(self) -> str:
        return type(self).get_cli_names()[self]

    @classmethod
    def from_string(cls, name):
        member = cls.get_choices().get(name.lower())
        if member is None:
            raise RuntimeError(f"No method called {name}, valid choices are: {list(cls.get_choices())}")
        return member


_METHOD_NAMES = tuple(MeshReconstructionMethod.get_cli_names().values())


class BackgroundMeshOptions(Options):
    """How the static scene is reconstructed.  Only `TSDFFusion` runs on this path; the other two methods are accepted so that
    the reference's command lines parse, and the pipeline refuses them when it gets there."""
    supported_reconstruction_methods = list(MeshReconstructionMethod)

    _title = 'Static Mesh Options'
    _flags = (
        _Flag('mesh_reconstruction_method', 'reconstruction_method', str, 'tsdf_fusion',
              'which reconstruction builds the static-scene mesh', _METHOD_NAMES),
        _Flag('depth_mask_dilation_iterations', 'depth_mask_dilation_iterations', int, 10,
              'dilation passes applied to the dynamic-object masks before they blank the depth maps'),
        _Flag('sdf_volume_size', 'sdf_volume_size', float, 5.0, 'edge length of the SDF volume (bundle fusion only), metres'),
        _Flag('sdf_voxel_size', 'sdf_voxel_size', float, 0.005, 'edge length of one voxel, metres'),
        _Flag('sdf_max_voxels', 'sdf_max_voxels', int, 320_000_000,
              'voxel budget: a scene that would need more gets a coarser voxel instead'),
        _Flag('key_frame_threshold', 'key_frame_threshold', float, 0.3,
              'a sampled frame joins the key-frame set when its overlap with the last key frame is at most this ratio'),
        # the reference parses this flag as float and then fails its own int check (options.py:424 vs :390); int here, so that
        # the flag is usable
        _Flag('key_frame_step', 'key_frame_step', int, 30, 'key-frame selection looks at every n-th frame'),
    )

    def __init__(self, reconstruction_method=MeshReconstructionMethod.TSDFFusion, depth_mask_dilation_iterations=10,
                 sdf_volume_size=5.0, sdf_voxel_size=0.005, sdf_max_voxels: Optional[int] = 320_000_000,
                 key_frame_threshold=0.3, key_frame_step=30):
        _check(reconstruction_method in self.supported_reconstruction_methods,
               f"reconstruction_method: expected one of {[m.name for m in self.supported_reconstruction_methods]}, "
               f"got {reconstruction_method!r}")
        _check(isinstance(depth_mask_dilation_iterations, int) and depth_mask_dilation_iterations >= 0,
               f"depth_mask_dilation_iterations: expected an int >= 0, got {depth_mask_dilation_iterations!r}")
        _check(sdf_volume_size > 0.0, f"sdf_volume_size: expected a positive length, got {sdf_volume_size!r}")
        _check(sdf_voxel_size > 0.0, f"sdf_voxel_size: expected a positive length, got {sdf_voxel_size!r}")
        _check(sdf_max_voxels is None or (isinstance(sdf_max_voxels, int) and sdf_max_voxels > 0),
               f"sdf_max_voxels: expected None or an int > 0, got {sdf_max_voxels!r}")
        if not 0.0 <= key_frame_threshold <= 1.0:  # the one check the reference raises ValueError for (options.py:386-388)
            raise ValueError(f"key_frame_threshold: expected a ratio in [0, 1], got {key_frame_threshold!r}")
        _check(isinstance(key_frame_step, int) and key_frame_step > 1,
               f"key_frame_step: expected an int > 1, got {key_frame_step!r}")

        self.reconstruction_method = reconstruction_method
        self.depth_mask_dilation_iterations = depth_mask_dilation_iterations
        self.sdf_volume_size = sdf_volume_size
        self.sdf_voxel_size = sdf_voxel_size
        self.sdf_max_voxels = sdf_max_voxels
        self.key_frame_threshold = key_frame_threshold
        self.key_frame_step = key_frame_step

    @staticmethod
    def add_args(parser: argparse.ArgumentParser):
        BackgroundMeshOptions._register(parser)

    @staticmethod
    def from_args(args: argparse.Namespace) -> 'BackgroundMeshOptions':
        fields = BackgroundMeshOptions._collect(args)
        fields['reconstruction_method'] = MeshReconstructionMethod.from_string(fields['reconstruction_method'])
        return BackgroundMeshOptions(**fields)
