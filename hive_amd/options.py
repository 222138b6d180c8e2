"""The option groups of `hive.options`.  Public names, constructor arguments, defaults, CLI flags and the exception type each
bad value raises follow /root/reference/hive/options.py (:44-67 base class, :245-268 mask dilation, :271-306 mesh filtering,
:310-350 reconstruction method, :353-439 static mesh -- the groups the hot path reads); the implementation is a flag table per
group, from which the parser arguments and `from_args` are both derived.  The groups of stages that are out of this build's scope
(:70-104 storage, :107-207 COLMAP, :210-242 mesh decimation, :442-466 trajectory smoothing, :469-527 WebXR, :530-582 inpainting
mode, :585-689 pipeline) are kept as API -- same names, arguments, defaults and flags, so that the reference's command lines and
`from hive.options import ...` lines keep working -- with no stage behind them here ("KEEP API, no logic")."""
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
    required: bool = False  # (type bool = a switch: `store_true`, default False)


class Options(abc.ABC):
    """An option group: `add_args` registers its flags on a parser, `from_args` rebuilds the group from the parsed namespace.
    Groups that list their flags in `_flags` / `_title` inherit both."""
    _title: str = ""
    _flags: tuple = ()

    @classmethod
    def _register(cls, parser: argparse.ArgumentParser):
        section = parser.add_argument_group(cls._title)
        for f in cls._flags:
            if f.type is bool:
                section.add_argument(f"--{f.flag}", action="store_true", help=f.help)
                continue
            extra = {"choices": list(f.choices)} if f.choices else {}
            if f.required:
                extra["required"] = True
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
        """True for the reference's default, the 3x3 box (the separable fast path of the HIP kernels)."""
        return np.array_equal(np.asarray(self.filter), _box3x3())

    def structuring_element(self):
        """``self.filter`` as the uint8 [kh][kw] array the C ABI takes (non-zero = member), validated like cv2 would use it."""
        se = np.ascontiguousarray(np.asarray(self.filter) != 0, dtype=np.uint8)
        if se.ndim != 2 or se.size == 0 or se.shape[0] > 32 or se.shape[1] > 32:
            raise ValueError(f"dilation_filter must be a 2-D structuring element of at most 32 x 32, got shape {np.asarray(self.filter).shape}")
        if not se.any():
            raise ValueError("dilation_filter has no set element")
        return se

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


# ---- groups of the stages outside this build's scope: API only (names, arguments, defaults, flags of the reference) -------------

class StorageOptions(Options):
    """Where the inputs are read from and the outputs written to (options.py:70-104)."""
    _title = 'Storage Options'
    _flags = (_Flag('dataset_path', 'dataset_path', str, None, 'folder that holds the RGB and depth image folders', required=True),
              _Flag('output_path', 'output_path', str, None, 'folder the outputs are written to', required=True),
              _Flag('overwrite_ok', 'overwrite_ok', bool, False, 'allow replacing mesh data already present in the output / export folders'),
              _Flag('no_cache', 'no_cache', bool, False, 'ignore cached datasets and results'))

    def __init__(self, dataset_path, output_path, overwrite_ok=False, no_cache=False):
        self.dataset_path = dataset_path
        self.output_path = output_path
        self.overwrite_ok = overwrite_ok
        self.no_cache = no_cache

    @staticmethod
    def add_args(parser: argparse.ArgumentParser):
        StorageOptions._register(parser)

    @staticmethod
    def from_args(args) -> 'StorageOptions':
        return StorageOptions(**StorageOptions._collect(args))


class COLMAPOptions(Options):
    """Settings of the COLMAP pose-estimation stage (options.py:107-207); the stage itself is out of scope here."""
    quality_choices = ('low', 'medium', 'high', 'extreme')
    _title = 'COLMAP Options'
    _flags = (_Flag('multiple_cameras', 'multiple_cameras', bool, False, 'the video comes from several devices / per-frame camera settings'),
              _Flag('single_camera_per_folder', 'single_camera_per_folder', bool, False, 'frames are organised in sub-folders, one camera each'),
              _Flag('dense', 'dense', bool, False, 'run dense reconstruction too'),
              _Flag('quality', 'quality', str, 'low', 'quality preset of the reconstruction', quality_choices),
              _Flag('binary_path', 'binary_path', str, '/usr/local/bin/colmap', 'path of the COLMAP binary'),
              _Flag('vocab_path', 'vocab_path', str, '/root/.cache/colmap/vocab.bin', 'path of the COLMAP vocabulary file'))
    _json_fields = ('binary_path', 'vocab_path', 'is_single_camera', 'single_camera_per_folder', 'dense', 'quality')

    def __init__(self, is_single_camera=True, single_camera_per_folder=False, dense=False, quality='low',
                 binary_path='/usr/local/bin/colmap', vocab_path='/root/.cache/colmap/vocab.bin'):
        self.binary_path = binary_path
        self.vocab_path = vocab_path
        self.is_single_camera = is_single_camera
        self.single_camera_per_folder = single_camera_per_folder
        self.dense = dense
        self.quality = quality

    @property
    def quality(self) -> str:
        return self._quality

    @quality.setter
    def quality(self, quality: str):
        _check(quality in self.quality_choices, f"quality: expected one of {self.quality_choices}, got {quality!r}")
        self._quality = quality

    @staticmethod
    def add_args(parser: argparse.ArgumentParser):
        COLMAPOptions._register(parser)

    @staticmethod
    def from_args(args: argparse.Namespace) -> 'COLMAPOptions':
        fields = COLMAPOptions._collect(args)
        fields['is_single_camera'] = not fields.pop('multiple_cameras')
        return COLMAPOptions(**fields)

    def to_json(self) -> dict:
        return {name: getattr(self, name) for name in self._json_fields}

    @classmethod
    def from_json(cls, json_dict: dict) -> 'COLMAPOptions':
        fields = {name: json_dict[name] for name in cls._json_fields if name in json_dict}  # (older files lack single_camera_per_folder)
        for name in ('is_single_camera', 'single_camera_per_folder', 'dense'):
            if name in fields:
                fields[name] = bool(fields[name])
        return cls(**fields)

    def copy(self) -> 'COLMAPOptions':
        return COLMAPOptions(**self.to_json())

    def __eq__(self, other) -> bool:
        return type(other) is type(self) and self.to_json() == other.to_json()


class MeshDecimationOptions(Options):
    """Face budgets of the decimation stage (options.py:210-242); -1 disables decimation for that mesh class."""
    _title = 'Mesh Decimation Options'
    _flags = (_Flag('num_faces_background', 'num_faces_background', int, 2 ** 14, 'face budget of the background mesh'),
              _Flag('num_faces_object', 'num_faces_object', int, 2 ** 10, 'face budget of each object mesh'),
              _Flag('decimation_max_error', 'max_error', float, 0.001, 'error bound handed to the decimater'))

    def __init__(self, num_faces_background=2 ** 14, num_faces_object=2 ** 10, max_error=0.001):
        self.num_faces_background = num_faces_background
        self.num_faces_object = num_faces_object
        self.max_error = max_error

    @staticmethod
    def add_args(parser: argparse.ArgumentParser):
        MeshDecimationOptions._register(parser)

    @staticmethod
    def from_args(args) -> 'MeshDecimationOptions':
        return MeshDecimationOptions(**MeshDecimationOptions._collect(args))


class ForegroundTrajectorySmoothingOptions(Options):
    """Step size / iteration count of the foreground trajectory smoothing (options.py:442-466); 0 epochs = off."""
    _title = 'Foreground Trajectory Smoothing'
    _flags = (_Flag('fts_learning_rate', 'learning_rate', float, 1e-5, 'step size of one smoothing epoch'),
              _Flag('fts_num_epochs', 'num_epochs', int, 0, 'smoothing epochs (0 disables the smoothing)'))

    def __init__(self, learning_rate=1e-5, num_epochs=0):
        self.learning_rate = learning_rate
        self.num_epochs = num_epochs

    @staticmethod
    def add_args(parser: argparse.ArgumentParser):
        ForegroundTrajectorySmoothingOptions._register(parser)

    @staticmethod
    def from_args(args: argparse.Namespace) -> 'ForegroundTrajectorySmoothingOptions':
        return ForegroundTrajectorySmoothingOptions(**ForegroundTrajectorySmoothingOptions._collect(args))


class WebXROptions(Options):
    """Export target and switches of the WebXR renderer (options.py:469-527); the renderer is out of scope here."""
    _title = 'WebXR'
    _flags = (_Flag('webxr_source_path', 'webxr_source_path', str, 'third_party/HIVE_Renderer', 'source folder of the renderer'),
              _Flag('webxr_path', 'webxr_path', str, 'third_party/HIVE_Renderer/docs/video', 'folder the 3D video files are exported to'),
              _Flag('webxr_url', 'webxr_url', str, 'http://localhost:8080', 'URL of the WebXR player'),
              _Flag('webxr_add_ground_plane', 'webxr_add_ground_plane', bool, False, 'render a white ground plane (debugging aid)'),
              _Flag('webxr_add_sky_box', 'webxr_add_sky_box', bool, False, 'render a sky cube map behind the scene'),
              _Flag('webxr_run_server', 'webxr_run_server', bool, False, 'start the web server after the export'))

    def __init__(self, webxr_source_path: str = 'third_party/HIVE_Renderer', webxr_path='third_party/HIVE_Renderer/docs/video',
                 webxr_url='localhost:8080', webxr_add_ground_plane=False, webxr_add_sky_box=False, webxr_run_server=False):
        self.webxr_source_path = webxr_source_path
        self.webxr_path = webxr_path
        self.webxr_url = webxr_url
        self.webxr_add_ground_plane = webxr_add_ground_plane
        self.webxr_add_sky_box = webxr_add_sky_box
        self.webxr_run_server = webxr_run_server

    @staticmethod
    def add_args(parser: argparse.ArgumentParser):
        WebXROptions._register(parser)

    @staticmethod
    def from_args(args: argparse.Namespace) -> 'WebXROptions':
        return WebXROptions(**WebXROptions._collect(args))

    def copy(self) -> 'WebXROptions':
        return WebXROptions(**vars(self))


class InpaintingMode(enum.Flag):
    """Which inpainter fills the background behind the dynamic objects, for colour and for depth (options.py:530-582).  The five
    combinations the CLI exposes map to the integers 0..4."""
    Off = 0
    CV2_Image = enum.auto()
    CV2_Depth = enum.auto()
    Lama_Image = enum.auto()
    Lama_Depth = enum.auto()

    CV2_Image_Depth = CV2_Image | CV2_Depth
    Lama_Image_CV2_Depth = Lama_Image | CV2_Depth
    CV2_Image_Lama_Depth = CV2_Image | Lama_Depth
    Lama_Image_Depth = Lama_Image | Lama_Depth

    @classmethod
    def get_modes(cls):
        return [cls.Off, cls.CV2_Image_Depth, cls.Lama_Image_CV2_Depth, cls.CV2_Image_Lama_Depth, cls.Lama_Image_Depth]

    def to_integer(self) -> int:
        modes = type(self).get_modes()
        if self not in modes:
            raise RuntimeError(f"{self.name} does not have an integer mapping, only {modes} have one.")
        return modes.index(self)

    @classmethod
    def from_integer(cls, value: int) -> 'InpaintingMode':
        modes = cls.get_modes()
        if not (isinstance(value, int) and 0 <= value < len(modes)):
            raise RuntimeError(f"Unrecognised integer value for {cls.__name__}, expected one of {modes}.")
        return modes[value]

    @classmethod
    def get_name(cls, value: int) -> str:
        return cls.from_integer(value).name

    @classmethod
    def get_modes_as_integer(cls):
        return list(range(len(cls.get_modes())))


class PipelineOptions(Options):
    """Top-level switches of `hive.pipeline.Pipeline` (options.py:585-689).  This build's `Pipeline` reads `num_frames`,
    `estimate_depth` and `background_only`; the others belong to stages outside its scope and are carried as API."""
    _title = 'Pipeline'
    _flags = (_Flag('num_frames', 'num_frames', int, -1, 'process at most this many frames (-1: all)'),
              _Flag('frame_step', 'frame_step', int, 15, 'sample every n-th frame for COLMAP and pose optimisation'),
              _Flag('estimate_pose', 'estimate_pose', bool, False, 'estimate camera parameters with COLMAP instead of using the provided ones'),
              _Flag('estimate_depth', 'estimate_depth', bool, False, 'estimate depth maps (DPT) instead of using the provided ones'),
              _Flag('background_only', 'background_only', bool, False, 'reconstruct the static background only'),
              _Flag('static_camera', 'static_camera', bool, False, 'treat the camera as static (Kinect intrinsics, identity poses)'),
              _Flag('align_scene', 'align_scene', bool, False, 'align the scene with the ground plane'),
              _Flag('inpainting_mode', 'inpainting_mode', int, 0,
                    'inpainting of the background: ' + ', '.join(f'{i}={m.name}' for i, m in enumerate(InpaintingMode.get_modes())),
                    tuple(InpaintingMode.get_modes_as_integer())),
              _Flag('billboard', 'billboard', bool, False, 'flat billboards instead of meshes for the foreground objects'),
              _Flag('disable_scaling', 'disable_scaling', bool, False, 'keep the input resolution instead of rescaling to 640x480'),
              _Flag('disable_coverage_constraint', 'disable_coverage_constraint', bool, False, 'keep foreground objects that cover under 1 %% of the frame'),
              _Flag('log_file', 'log_file', str, 'logs.log', 'path of the log file'))

    def __init__(self, num_frames=-1, frame_step=15, estimate_pose=False, estimate_depth=False, background_only=False, static_camera=False,
                 align_scene=False, inpainting_mode=InpaintingMode.Off, billboard=False, disable_scaling=False, disable_coverage_constraint=False,
                 log_file='logs.log'):
        self.disable_scaling = disable_scaling
        self.disable_coverage_constraint = disable_coverage_constraint
        self.num_frames = num_frames
        self.frame_step = frame_step
        self.estimate_pose = estimate_pose
        self.estimate_depth = estimate_depth
        self.background_only = background_only
        self.static_camera = static_camera
        self.align_scene = align_scene
        self.inpainting_mode = inpainting_mode
        self.billboard = billboard
        self.log_file = log_file

    @staticmethod
    def add_args(parser: argparse.ArgumentParser):
        PipelineOptions._register(parser)

    @staticmethod
    def from_args(args: argparse.Namespace) -> 'PipelineOptions':
        fields = PipelineOptions._collect(args)
        fields['inpainting_mode'] = InpaintingMode.from_integer(fields['inpainting_mode'])
        return PipelineOptions(**fields)

    def copy(self) -> 'PipelineOptions':
        return PipelineOptions(**vars(self))
