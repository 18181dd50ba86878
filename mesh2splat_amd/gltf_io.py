"""Scene file I/O on the Python side.

* :func:`load_glb` / :func:`read_ply` — thin bindings of the C++ loader (m2s_load_glb, m2s_read_ply in
  include/m2s.h), i.e. the re-hosted SceneManager::loadModel / parsers::loadPlyFile.
* :func:`write_glb` — a small glTF-binary WRITER so that tests and benchmarks can push synthetic scenes
  through the real loader (the reference ships no assets and there is no network).  Pure numpy + zlib.
"""
from __future__ import annotations

import ctypes as C
import json
import os
import struct
import zlib
from typing import Dict, Optional, Sequence

import numpy as np

from . import _lib
from .scene import BASE_COLOR_TEXTURE, METALLIC_ROUGHNESS_TEXTURE, NORMAL_TEXTURE, TEXTURE_SLOTS, Mesh, Scene


# ---------------------------------------------------------------------------------------------------
# loader bindings
# ---------------------------------------------------------------------------------------------------
def load_glb(path: str) -> Scene:
    """.glb -> Scene (world-space de-indexed 17-float vertices, cumulative bboxes, RGBA8 textures)."""
    L = _lib.load()
    h = C.c_void_p()
    st = L.m2s_load_glb(os.fsencode(path), C.byref(h))
    if st != _lib.M2S_OK:
        raise _lib.M2SError(st, L.m2s_io_last_error().decode())
    try:
        n = L.m2s_host_scene_num_meshes(h)
        arr = L.m2s_host_scene_meshes(h)
        meshes = []
        for i in range(n):
            m = arr[i]
            v = np.ctypeslib.as_array(C.cast(m.vertices, C.POINTER(C.c_float)), shape=(m.n_vertices, m.stride_floats)).copy() \
                if m.n_vertices else np.zeros((0, m.stride_floats), np.float32)
            tex = {}
            for k, key in enumerate(TEXTURE_SLOTS):
                t = m.tex[k]
                if t.rgba8:
                    tex[key] = np.ctypeslib.as_array(C.cast(t.rgba8, C.POINTER(C.c_uint8)), shape=(t.height, t.width, 4)).copy()
            meshes.append(Mesh(name=L.m2s_host_scene_mesh_name(h, i).decode(), vertices=v, base_color=tuple(m.base_color),
                               textures=tex, bbox_min=np.array(m.bbox_min, np.float32), bbox_max=np.array(m.bbox_max, np.float32)))
        sc = Scene(meshes)
        sc.warnings = L.m2s_host_scene_warnings(h).decode().splitlines()
        return sc
    finally:
        L.m2s_free_host_scene(h)


def read_ply(path: str):
    """.ply (format 0 or 1) -> (records (n,24) float32, has_pbr) with parsers::loadPlyFile semantics."""
    L = _lib.load()
    rec = C.c_void_p()
    n = C.c_uint64()
    pbr = C.c_int()
    st = L.m2s_read_ply(os.fsencode(path), C.byref(rec), C.byref(n), C.byref(pbr))
    if st != _lib.M2S_OK:
        raise _lib.M2SError(st, L.m2s_io_last_error().decode())
    try:
        out = np.ctypeslib.as_array(C.cast(rec, C.POINTER(C.c_float)), shape=(n.value, 24)).copy() if n.value else \
            np.zeros((0, 24), np.float32)
    finally:
        L.m2s_free_records(rec)
    return out, bool(pbr.value)


# ---------------------------------------------------------------------------------------------------
# writer (test / benchmark tooling)
# ---------------------------------------------------------------------------------------------------
def encode_png(img: np.ndarray) -> bytes:
    """(H,W,4|3|2|1) uint8 -> PNG bytes (8-bit, filter 0, zlib level 3)."""
    a = np.ascontiguousarray(img, np.uint8)
    if a.ndim == 2:
        a = a[:, :, None]
    h, w, c = a.shape
    ctype = {1: 0, 2: 4, 3: 2, 4: 6}[c]
    raw = np.empty((h, 1 + w * c), np.uint8)
    raw[:, 0] = 0
    raw[:, 1:] = a.reshape(h, w * c)

    def chunk(tag: bytes, body: bytes) -> bytes:
        return struct.pack(">I", len(body)) + tag + body + struct.pack(">I", zlib.crc32(tag + body) & 0xFFFFFFFF)

    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ctype, 0, 0, 0)) +
            chunk(b"IDAT", zlib.compress(raw.tobytes(), 3)) + chunk(b"IEND", b""))


def _pad4(b: bytes, fill: bytes = b"\x00") -> bytes:
    return b + fill * ((4 - len(b) % 4) % 4)


def write_glb(scene: Scene, path: str, indexed: bool = True, with_normals: bool = True, with_tangents: bool = True,
              with_uvs: bool = True, node_trs: Optional[Sequence[dict]] = None, png_override: Optional[Dict[str, bytes]] = None,
              index_type: str = "auto", nested: bool = False) -> None:
    """Write `scene` as one .glb: one glTF mesh + node per Scene mesh.

    node_trs[i] may hold 'translation' (3), 'rotation' (x,y,z,w), 'scale' (3) or 'matrix' (16, column-major);
    the vertex data written is the Scene's as-is (so the loader's output = transform applied to it).
    png_override maps a texture key to raw PNG bytes used instead of encoding the array (decoder tests)."""
    bin_parts = []
    off = 0
    views, accessors, meshes_j, nodes_j, materials, textures, images = [], [], [], [], [], [], []
    image_of = {}

    def add_view(data: bytes, target: Optional[int] = None) -> int:
        nonlocal off
        data = _pad4(data)
        v = {"buffer": 0, "byteOffset": off, "byteLength": len(data)}
        if target:
            v["target"] = target
        views.append(v)
        bin_parts.append(data)
        off += len(data)
        return len(views) - 1

    def add_accessor(arr: np.ndarray, ctype: int, typ: str, target: Optional[int] = None, minmax: bool = False) -> int:
        a = np.ascontiguousarray(arr)
        acc = {"bufferView": add_view(a.tobytes(), target), "componentType": ctype, "count": int(a.shape[0]), "type": typ}
        if minmax and a.shape[0]:
            acc["min"] = [float(x) for x in a.min(axis=0)]
            acc["max"] = [float(x) for x in a.max(axis=0)]
        accessors.append(acc)
        return len(accessors) - 1

    def add_texture(key: str, img: np.ndarray) -> int:
        ident = (key if png_override and key in png_override else None, id(img))
        if ident not in image_of:
            data = png_override[key] if png_override and key in png_override else encode_png(img)
            images.append({"bufferView": add_view(data), "mimeType": "image/jpeg" if data[:2] == b"\xff\xd8" else "image/png"})
            textures.append({"source": len(images) - 1})
            image_of[ident] = len(textures) - 1
        return image_of[ident]

    for mi, m in enumerate(scene.meshes):
        v = m.vertices
        attr_cols = np.concatenate([v[:, 0:3], v[:, 3:6] if with_normals else np.zeros((len(v), 0), np.float32),
                                    v[:, 6:10] if with_tangents else np.zeros((len(v), 0), np.float32),
                                    v[:, 10:12] if with_uvs else np.zeros((len(v), 0), np.float32)], axis=1)
        if indexed and len(v):
            uniq, inv = np.unique(attr_cols, axis=0, return_inverse=True)
            idx = inv.reshape(-1).astype(np.uint32)
        else:
            uniq, idx = attr_cols, None
        col = 0
        attrs = {"POSITION": add_accessor(uniq[:, 0:3].astype(np.float32), 5126, "VEC3", 34962, minmax=True)}
        col = 3
        if with_normals:
            attrs["NORMAL"] = add_accessor(uniq[:, col:col + 3].astype(np.float32), 5126, "VEC3", 34962)
            col += 3
        if with_tangents:
            attrs["TANGENT"] = add_accessor(uniq[:, col:col + 4].astype(np.float32), 5126, "VEC4", 34962)
            col += 4
        if with_uvs:
            attrs["TEXCOORD_0"] = add_accessor(uniq[:, col:col + 2].astype(np.float32), 5126, "VEC2", 34962)
        prim = {"attributes": attrs, "mode": 4}
        if idx is not None:
            it = index_type
            if it == "auto":
                it = "u8" if len(uniq) <= 255 else "u16" if len(uniq) <= 65535 else "u32"
            npt, ct = {"u8": (np.uint8, 5121), "u16": (np.uint16, 5123), "u32": (np.uint32, 5125)}[it]
            prim["indices"] = add_accessor(idx.astype(npt), ct, "SCALAR", 34963)
        mat = {"name": f"mat_{mi}", "pbrMetallicRoughness": {"baseColorFactor": [float(x) for x in m.base_color]}}
        if BASE_COLOR_TEXTURE in m.textures:
            mat["pbrMetallicRoughness"]["baseColorTexture"] = {"index": add_texture(BASE_COLOR_TEXTURE, m.textures[BASE_COLOR_TEXTURE])}
        if METALLIC_ROUGHNESS_TEXTURE in m.textures:
            mat["pbrMetallicRoughness"]["metallicRoughnessTexture"] = {
                "index": add_texture(METALLIC_ROUGHNESS_TEXTURE, m.textures[METALLIC_ROUGHNESS_TEXTURE])}
        if NORMAL_TEXTURE in m.textures:
            mat["normalTexture"] = {"index": add_texture(NORMAL_TEXTURE, m.textures[NORMAL_TEXTURE])}
        materials.append(mat)
        prim["material"] = len(materials) - 1
        name = m.name.rsplit("_", 1)[0] if "_" in m.name else m.name
        meshes_j.append({"name": name, "primitives": [prim]})
        node = {"mesh": mi, "name": f"node_{mi}"}
        if node_trs and mi < len(node_trs) and node_trs[mi]:
            node.update({k: [float(x) for x in val] for k, val in node_trs[mi].items()})
        nodes_j.append(node)

    roots = list(range(len(nodes_j)))
    if nested and nodes_j:  # one extra parent node carrying an identity-ish transform chain
        nodes_j.append({"name": "root", "children": roots, "translation": [0.0, 0.0, 0.0]})
        roots = [len(nodes_j) - 1]
    doc = {"asset": {"version": "2.0", "generator": "mesh2splat_amd.gltf_io"}, "scene": 0, "scenes": [{"nodes": roots}],
           "nodes": nodes_j, "meshes": meshes_j, "materials": materials, "accessors": accessors, "bufferViews": views,
           "buffers": [{"byteLength": off}]}
    if textures:
        doc["textures"] = textures
        doc["images"] = images
    js = _pad4(json.dumps(doc, separators=(",", ":")).encode(), b" ")
    binc = b"".join(bin_parts)
    total = 12 + 8 + len(js) + 8 + len(binc)
    with open(path, "wb") as f:
        f.write(struct.pack("<III", 0x46546C67, 2, total))
        f.write(struct.pack("<II", len(js), 0x4E4F534A) + js)
        f.write(struct.pack("<II", len(binc), 0x004E4942) + binc)
