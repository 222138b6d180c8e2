"""The background (static scene) half of HIVE's pipeline -- the part the dense-compute path serves:
dataset -> key frames -> TSDF fusion -> mesh (/root/reference/hive/pipeline.py:258-286, 871-901).

The foreground per-frame meshing, glTF export, draco compression and the WebXR viewer of the reference's
``Pipeline`` are outside the scope of this build.
"""
import argparse
import json
import logging
import os
from typing import Optional

import numpy as np

from hive_amd import fusion
from hive_amd.dataset_adaptors import get_dataset
from hive_amd.io import HiveDataset
from hive_amd.options import (BackgroundMeshOptions, COLMAPOptions, ForegroundTrajectorySmoothingOptions, MaskDilationOptions, MeshDecimationOptions,
                              MeshFilteringOptions, MeshReconstructionMethod, PipelineOptions, StorageOptions, WebXROptions)
from hive_amd.utils import timed_block


def write_ply(path, vertices, faces, vertex_colors=None, vertex_normals=None):
    """Binary little-endian PLY (trimesh, which the reference exports glb with, is not a dependency here)."""
    vertices = np.asarray(vertices, np.float32)
    faces = np.asarray(faces, np.int32)
    fields = [("x", "<f4"), ("y", "<f4"), ("z", "<f4")]
    if vertex_normals is not None:
        fields += [("nx", "<f4"), ("ny", "<f4"), ("nz", "<f4")]
    if vertex_colors is not None:
        fields += [("red", "u1"), ("green", "u1"), ("blue", "u1")]
    v = np.empty(len(vertices), dtype=fields)
    v["x"], v["y"], v["z"] = vertices.T
    if vertex_normals is not None:
        v["nx"], v["ny"], v["nz"] = np.asarray(vertex_normals, np.float32).T
    if vertex_colors is not None:
        c = np.asarray(vertex_colors)
        v["red"], v["green"], v["blue"] = c[:, 0], c[:, 1], c[:, 2]
    f = np.empty(len(faces), dtype=[("n", "u1"), ("i", "<i4", (3,))])
    f["n"], f["i"] = 3, faces
    names = {"<f4": "float", "u1": "uchar"}
    with open(path, "wb") as out:
        header = ["ply", "format binary_little_endian 1.0", f"element vertex {len(v)}"]
        header += [f"property {names[t]} {n}" for n, t in fields]
        header += [f"element face {len(f)}", "property list uchar int vertex_indices", "end_header"]
        out.write(("\n".join(header) + "\n").encode())
        out.write(v.tobytes())
        out.write(f.tobytes())


class Pipeline:
    """The reference's ``Pipeline`` (/root/reference/hive/pipeline.py:59-262) for the part this build serves: dataset -> key frames
    -> TSDF fusion -> background mesh.  Constructor, ``from_command_line`` and ``run`` take the reference's arguments; the stages
    behind the other option groups (foreground meshes, decimation, glTF / draco export, WebXR) are outside this build's scope
     and ``run`` says so where the reference would run them."""
    mesh_folder = "mesh"
    bundle_fusion_folder = "bundle_fusion"

    def __init__(self, options: Optional[PipelineOptions] = None, storage_options: Optional[StorageOptions] = None, decimation_options=None,
                 dilation_options=None, filtering_options=None, colmap_options=None, static_mesh_options: Optional[BackgroundMeshOptions] = None,
                 webxr_options=None, fts_options=None, *, background_mesh_options: Optional[BackgroundMeshOptions] = None, num_frames=None):
        """Reference form (pipeline.py:67-99): ``Pipeline(options, storage_options, decimation_options, dilation_options,
        filtering_options, colmap_options, static_mesh_options, webxr_options, fts_options)``.  The keyword shorthand of this
        build's earlier rounds -- ``Pipeline(background_mesh_options=..., num_frames=...)`` -- still works."""
        self.options = options or PipelineOptions()
        if num_frames is not None:
            self.options.num_frames = num_frames
        self.storage_options = storage_options
        self.decimation_options = decimation_options or MeshDecimationOptions()
        self.dilation_options = dilation_options or MaskDilationOptions()
        self.filtering_options = filtering_options or MeshFilteringOptions()
        self.colmap_options = colmap_options or COLMAPOptions()
        self.background_mesh_options = static_mesh_options or background_mesh_options or BackgroundMeshOptions()
        self.webxr_options = webxr_options or WebXROptions()
        self.fts_options = fts_options or ForegroundTrajectorySmoothingOptions()
        self.profiling = {}

    @staticmethod
    def from_command_line(argv=None) -> 'Pipeline':
        """pipeline.py:100-141: every option group registers its flags on one parser; ``argv=None`` reads ``sys.argv``."""
        parser = argparse.ArgumentParser("HIVE", description="Create 3D mesh videos from a RGB-D sequence with camera trajectory annotations.")
        groups = (PipelineOptions, StorageOptions, MaskDilationOptions, MeshFilteringOptions, MeshDecimationOptions, COLMAPOptions, BackgroundMeshOptions,
                  WebXROptions)
        for group in groups:
            group.add_args(parser)
        args = parser.parse_args(argv)
        logging.debug(args)
        return Pipeline(options=PipelineOptions.from_args(args), storage_options=StorageOptions.from_args(args),
                        decimation_options=MeshDecimationOptions.from_args(args), dilation_options=MaskDilationOptions.from_args(args),
                        filtering_options=MeshFilteringOptions.from_args(args), colmap_options=COLMAPOptions.from_args(args),
                        static_mesh_options=BackgroundMeshOptions.from_args(args), webxr_options=WebXROptions.from_args(args))

    @property
    def num_frames(self) -> int:
        return self.options.num_frames

    @property
    def estimate_pose(self) -> bool:
        return self.options.estimate_pose

    @property
    def estimate_depth(self) -> bool:
        return self.options.estimate_depth

    @property
    def mesh_path(self) -> str:
        return os.path.join(self.storage_options.output_path, self.mesh_folder)

    @staticmethod
    def create_static_mesh(dataset: HiveDataset, num_frames=-1, options: Optional[BackgroundMeshOptions] = None, frame_set=None):
        """pipeline.py:871-901: key frames (unless a frame set is given), then TSDF fusion."""
        options = options or BackgroundMeshOptions()
        if num_frames < 1:
            num_frames = dataset.num_frames
        if options.reconstruction_method != MeshReconstructionMethod.TSDFFusion:
            raise NotImplementedError(f"{options.reconstruction_method.name}: only TSDFFusion is part of the dense-compute path")
        if frame_set is None:
            frame_set = dataset.select_key_frames(threshold=options.key_frame_threshold, frame_step=options.key_frame_step)
            frame_set = [f for f in frame_set if f < num_frames]
        logging.info(f"Creating background mesh from {len(frame_set)} key frames...")
        return fusion.tsdf_fusion(dataset, options, num_frames=num_frames, frame_set=frame_set)

    def run(self, dataset=None, adaptor=None, compress=True, *, estimate_depth=None):
        """pipeline.py:172-262 up to the background mesh: load (or convert) the dataset, fuse the static scene, write
        ``<output>/mesh/bg.ply`` (vertex colours in sRGB as pipeline.py:281-282) and ``profiling.json``; returns the mesh.

        Reference form: ``run(dataset: Optional[HiveDataset] = None, adaptor: Optional[DatasetAdaptor] = None, compress=True)`` with the
        paths in ``storage_options``.  Shorthand of earlier rounds: ``run(dataset_path, output_path, estimate_depth=False)``."""
        if isinstance(dataset, (str, os.PathLike)):  # shorthand: run(dataset_path, output_path)
            self.storage_options = StorageOptions(dataset_path=str(dataset), output_path=str(adaptor))
            dataset = adaptor = None
        if estimate_depth is not None:
            self.options.estimate_depth = bool(estimate_depth)
        if self.storage_options is None and dataset is None and adaptor is None:
            raise ValueError("Pipeline.run needs storage_options (dataset_path, output_path), a dataset or an adaptor")
        with timed_block("Loaded dataset in", self.profiling, ("timing", "load_dataset", "total")):
            if adaptor is not None:
                dataset = adaptor.convert(estimate_pose=self.estimate_pose, estimate_depth=self.estimate_depth, inpainting_mode=self.options.inpainting_mode,
                                          static_camera=self.options.static_camera, no_cache=bool(self.storage_options and self.storage_options.no_cache),
                                          profiling=self.profiling)
            elif dataset is None:
                resize_to = None if self.options.disable_scaling else 640
                dataset = get_dataset(self.storage_options, self.colmap_options, self.options, resize_to=resize_to, profiling=self.profiling)
            n = dataset.num_frames if self.num_frames == -1 else min(self.num_frames, dataset.num_frames)
        with timed_block("Created background mesh in", self.profiling, ("timing", "background_reconstruction", "total")):
            mesh = self.create_static_mesh(dataset, num_frames=n, options=self.background_mesh_options)
        if not self.options.background_only:
            logging.info("Foreground meshes, decimation, glTF / draco export and the WebXR viewer are outside this build's scope: "
                         "the background mesh is written as PLY.")
        # vertex colours -> sRGB as pipeline.py:281-282
        colors = np.asarray(mesh.visual.vertex_colors)[:, :3]
        colors = (255 * np.power(colors / 255, 2.2)).astype(np.uint8)
        out = self.mesh_path if self.storage_options is not None else os.path.join(dataset.base_path, self.mesh_folder)
        os.makedirs(out, exist_ok=True)
        with timed_block("Wrote mesh data in", self.profiling, ("timing", "mesh_export")):
            write_ply(os.path.join(out, "bg.ply"), mesh.vertices, mesh.faces, colors, mesh.vertex_normals)
        with open(os.path.join(dataset.base_path, "profiling.json"), "w") as f:  # pipeline.py:251
            json.dump(self.profiling, f, indent=2)
        return mesh


def main(argv=None):
    """`python -m hive_amd ...` / `python -m hive_amd.pipeline ...`: /root/reference/hive/pipeline.py:1337-1339 (and hive/__main__.py:17-20) --
    build the pipeline from the command line and run it."""
    program = Pipeline.from_command_line(argv)
    return program.run()


if __name__ == '__main__':
    main()
