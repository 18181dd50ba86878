"""Host-side scene types mirroring what the reference's conversion pass reads from RenderContext
(RenderContext.hpp:86-90): per mesh a de-indexed vertex buffer in the reference VBO layout,
`material.baseColorFactor`, the *cumulative* bbox and the RGBA8 texture map.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Sequence, Tuple

import numpy as np

# texture-map keys == params.hpp:12-16
BASE_COLOR_TEXTURE = "baseColorTexture"
NORMAL_TEXTURE = "normalTexture"
METALLIC_ROUGHNESS_TEXTURE = "metallicRoughnessTexture"
TEXTURE_SLOTS = (BASE_COLOR_TEXTURE, NORMAL_TEXTURE, METALLIC_ROUGHNESS_TEXTURE)  # tex units 0,1,2

MAX_GAUSSIANS_TO_SORT = 7_000_000  # RenderPass.hpp:9
RECORD_FLOATS = 24                 # utils::GaussianDataSSBO, utils.hpp:145-152 (6 x vec4 = 96 B)


@dataclass
class Mesh:
    """utils::Mesh + its meshToTextureData entry (utils.hpp:174-182, RenderContext.hpp:90)."""
    name: str
    vertices: np.ndarray                       # (3*T, stride) float32, stride >= 12
    base_color: Tuple[float, float, float, float] = (1.0, 1.0, 1.0, 1.0)
    textures: Dict[str, np.ndarray] = field(default_factory=dict)  # key -> (H, W, 4) uint8
    bbox_min: np.ndarray | None = None         # filled by Scene (cumulative, Q1)
    bbox_max: np.ndarray | None = None

    def __post_init__(self):
        v = np.ascontiguousarray(self.vertices, dtype=np.float32)
        if v.ndim != 2 or v.shape[1] < 12:
            raise ValueError("vertices must be (3*T, stride>=12) float32")
        if v.shape[0] % 3:
            raise ValueError("vertex count must be a multiple of 3 (de-indexed triangles)")
        self.vertices = v
        for k, t in list(self.textures.items()):
            t = np.ascontiguousarray(t, dtype=np.uint8)
            if t.ndim != 3 or t.shape[2] != 4:
                raise ValueError(f"texture {k} must be (H, W, 4) uint8 (tiny_gltf expands to RGBA8)")
            self.textures[k] = t

    @property
    def n_triangles(self) -> int:
        return self.vertices.shape[0] // 3

    @property
    def stride(self) -> int:
        return self.vertices.shape[1]


class Scene:
    """Ordered mesh list == RenderContext::dataMeshAndGlMesh after SceneManager::setupMeshBuffers.

    The bbox of mesh k is the AABB of meshes 0..k (SceneManager.cpp:476-477,514-520,527: minBB/maxBB
    are declared outside the mesh loop) unless the caller supplies explicit bboxes.
    """

    def __init__(self, meshes: Sequence[Mesh], cumulative_bbox: bool = True):
        self.meshes: List[Mesh] = list(meshes)
        mn = np.full(3, np.finfo(np.float32).max, np.float32)
        mx = np.full(3, -np.finfo(np.float32).max, np.float32)
        for m in self.meshes:
            if m.bbox_min is not None and m.bbox_max is not None:
                m.bbox_min = np.asarray(m.bbox_min, np.float32)
                m.bbox_max = np.asarray(m.bbox_max, np.float32)
                continue
            if not cumulative_bbox:
                mn = np.full(3, np.finfo(np.float32).max, np.float32)
                mx = np.full(3, -np.finfo(np.float32).max, np.float32)
            if m.n_triangles:
                p = m.vertices[:, 0:3]
                mn = np.minimum(mn, p.min(axis=0))
                mx = np.maximum(mx, p.max(axis=0))
            m.bbox_min, m.bbox_max = mn.copy(), mx.copy()

    @property
    def n_meshes(self) -> int:
        return len(self.meshes)

    @property
    def n_triangles(self) -> int:
        return sum(m.n_triangles for m in self.meshes)

    def mesh_first_triangle(self) -> np.ndarray:
        """Prefix of triangle counts: flattened (mesh-major) triangle index of each mesh's first triangle."""
        out = np.zeros(len(self.meshes) + 1, np.int64)
        for i, m in enumerate(self.meshes):
            out[i + 1] = out[i] + m.n_triangles
        return out


def reference_cap(R: int, n_meshes: int) -> int:
    """ConversionPass.cpp:21-24: `unsigned int maxGaussians = R*R*6*meshCount` (32-bit wrap) clamped to 7e6."""
    mc = max(1, int(n_meshes))
    mx = (int(R) * int(R) * 6 * mc) & 0xFFFFFFFF
    return min(mx, MAX_GAUSSIANS_TO_SORT)


def resolution_from_quality(quality: float, max_res: int = 1024) -> int:
    """ImGuiUI.cpp:512: R = int(16 + quality*(maxRes-16)); UI default quality 0.5, maxRes 1024 -> 520."""
    return int(16 + float(quality) * (int(max_res) - 16))
