"""Mesh container for ``tsdf_fusion``: a ``trimesh.Trimesh`` when trimesh is installed (as in
/root/reference/hive/fusion.py:132), otherwise a minimal stand-in with the same attribute names."""
import numpy as np


class _Visual:
    def __init__(self, vertex_colors):
        self.vertex_colors = vertex_colors


class Mesh:
    def __init__(self, vertices, faces, vertex_colors=None, vertex_normals=None):
        self.vertices = np.asarray(vertices)
        self.faces = np.asarray(faces)
        self.vertex_normals = None if vertex_normals is None else np.asarray(vertex_normals)
        colors = None
        if vertex_colors is not None:
            colors = np.asarray(vertex_colors)
            if colors.shape[1] == 3:  # trimesh stores RGBA
                colors = np.hstack([colors, np.full((len(colors), 1), 255, colors.dtype)])
        self.visual = _Visual(colors)

    @property
    def is_empty(self):
        return len(self.faces) == 0

    @property
    def euler_number(self):
        edges = np.sort(np.concatenate([self.faces[:, [0, 1]], self.faces[:, [1, 2]], self.faces[:, [2, 0]]]), axis=1)
        return len(self.vertices) - len(np.unique(edges, axis=0)) + len(self.faces)


def make_mesh(vertices, faces, vertex_colors=None, vertex_normals=None):
    try:
        import trimesh
    except ImportError:
        return Mesh(vertices, faces, vertex_colors, vertex_normals)
    return trimesh.Trimesh(vertices=vertices, faces=faces, vertex_colors=vertex_colors, vertex_normals=vertex_normals)
