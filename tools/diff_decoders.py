"""Differential check of the image decoders against the REFERENCE's loader (tiny_gltf -> stb_image, oracle/_ref/ref_host_check,
needs /root/reference at build time): every image file given is wrapped into a one-quad .glb as its base-colour texture, loaded by
both, and the decoded RGBA bytes are compared.  usage: python tools/diff_decoders.py DIR_OR_FILES...   (see tools/fuzz_seeds.py)"""
import glob
import json
import os
import struct
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import refhost                                   # noqa: E402
from mesh2splat_amd import gltf_io, synth        # noqa: E402


def wrap(blob: bytes, mime: str, dst: str):
    scene = synth.unit_quad(textures={"baseColorTexture": np.full((4, 4, 4), 200, np.uint8)})
    gltf_io.write_glb(scene, dst)
    raw = open(dst, "rb").read()
    jl = struct.unpack_from("<I", raw, 12)[0]
    doc = json.loads(raw[20:20 + jl])
    pos = 20 + jl
    bl = struct.unpack_from("<I", raw, pos)[0]
    binc = bytearray(raw[pos + 8: pos + 8 + bl])
    off = len(binc)
    binc += blob
    doc["bufferViews"].append({"buffer": 0, "byteOffset": off, "byteLength": len(blob)})
    doc["images"][0] = {"bufferView": len(doc["bufferViews"]) - 1, "mimeType": mime}
    doc["buffers"][0]["byteLength"] = len(binc)
    js = json.dumps(doc).encode()
    js += b" " * (-len(js) % 4)
    binc += b"\0" * (-len(binc) % 4)
    body = struct.pack("<II", len(js), 0x4E4F534A) + js + struct.pack("<II", len(binc), 0x004E4942) + bytes(binc)
    open(dst, "wb").write(struct.pack("<III", 0x46546C67, 2, 12 + len(body)) + body)


def main():
    files = []
    for a in sys.argv[1:]:
        files += sorted(glob.glob(os.path.join(a, "*"))) if os.path.isdir(a) else [a]
    files = [f for f in files if f.lower().endswith((".png", ".jpg", ".jpeg"))]
    same = differ = both_reject = only_ref = only_mine = 0
    with tempfile.TemporaryDirectory() as d:
        for f in files:
            glb = os.path.join(d, "w.glb")
            wrap(open(f, "rb").read(), "image/png" if f.lower().endswith(".png") else "image/jpeg", glb)
            try:
                ref = refhost.load_scene(glb, d)[0]["textures"].get("baseColorTexture")
            except Exception:
                ref = None
            try:
                mine = gltf_io.load_glb(glb).meshes[0].textures.get("baseColorTexture")
            except Exception as e:
                mine, merr = None, str(e)[-80:]
            if ref is None and mine is None:
                both_reject += 1
            elif ref is None:
                only_mine += 1
                print("ONLY THIS DECODER ACCEPTS", os.path.basename(f))
            elif mine is None:
                only_ref += 1
                print("ONLY THE REFERENCE ACCEPTS", os.path.basename(f), merr)
            else:
                r = ref if ref.shape[2] == 4 else None
                if r is not None and r.shape == mine.shape and np.array_equal(r, mine):
                    same += 1
                else:
                    differ += 1
                    print("DIFFERENT", os.path.basename(f), ref.shape, mine.shape,
                          int((ref.reshape(-1)[: mine.size] != mine.reshape(-1)[: ref.size]).sum()) if ref.size == mine.size else "size")
    print(f"{len(files)} files: identical {same}, different {differ}, both reject {both_reject}, only reference accepts {only_ref}, only this decoder accepts {only_mine}")
    return 1 if (differ or only_ref or only_mine) else 0


if __name__ == "__main__":
    sys.exit(main())
