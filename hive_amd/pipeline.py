"""The background (static scene) half of HIVE's pipeline -- the part the dense-compute path serves:
dataset -> key frames -> TSDF fusion -> mesh (/root/reference/hive/pipeline.py:258-286, 871-901).

The foreground per-frame meshing, glTF export, draco compression and the WebXR viewer of the reference's
``Pipeline`` are outside the scope of this build.
"""
import logging
import os
from typing import Optional

import numpy as np

from hive_amd import fusion
from hive_amd.dataset_adaptors import get_dataset
from hive_amd.io import HiveDataset
from hive_amd.options import BackgroundMeshOptions, MeshReconstructionMethod
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
    """``create_static_mesh`` / ``run`` with the reference's names and option objects."""

    def __init__(self, background_mesh_options: Optional[BackgroundMeshOptions] = None, num_frames=-1):
        self.background_mesh_options = background_mesh_options or BackgroundMeshOptions()
        self.num_frames = num_frames
        self.profiling = {}

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

    def run(self, dataset_path, output_path, estimate_depth=False):
        """dataset (HIVE format or TUM) -> background mesh written to <output>/mesh/bg.ply; returns the mesh."""
        with timed_block("Loaded dataset in", self.profiling, ("timing", "load_dataset", "total")):
            dataset = get_dataset(dataset_path, output_path, num_frames=self.num_frames, estimate_depth=estimate_depth)
        with timed_block("Created background mesh in", self.profiling, ("timing", "background_reconstruction", "total")):
            mesh = self.create_static_mesh(dataset, num_frames=self.num_frames, options=self.background_mesh_options)
        # vertex colours -> sRGB as pipeline.py:281-282
        colors = np.asarray(mesh.visual.vertex_colors)[:, :3]
        colors = (255 * np.power(colors / 255, 2.2)).astype(np.uint8)
        os.makedirs(os.path.join(output_path, "mesh"), exist_ok=True)
        write_ply(os.path.join(output_path, "mesh", "bg.ply"), mesh.vertices, mesh.faces, colors, mesh.vertex_normals)
        return mesh
