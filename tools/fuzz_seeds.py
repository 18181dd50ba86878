"""Seed corpus for tools/fuzz_host.sh: small PNG / JPEG files of every flavour the decoders handle (and a few they reject),
written with Pillow; the .glb / .ply seeds are the committed fixtures under tests/golden/ref_host.  usage: fuzz_seeds.py DIR"""
import os
import sys

import numpy as np
from PIL import Image

out = sys.argv[1]
os.makedirs(out, exist_ok=True)
rng = np.random.default_rng(7)


def pic(w, h, ch):
    y, x = np.mgrid[0:h, 0:w]
    base = np.stack([(x * 255 // max(w - 1, 1)), (y * 255 // max(h - 1, 1)), ((x ^ y) * 8) % 256, 255 - (x + y) % 256], -1).astype(np.uint8)
    noise = rng.integers(0, 40, (h, w, 4), dtype=np.uint8)
    return (base // 2 + noise)[..., :ch] if ch > 1 else (base // 2 + noise)[..., 0]


n = 0
for (w, h) in ((1, 1), (7, 5), (16, 16), (33, 17), (64, 48)):
    for mode, ch in (("L", 1), ("RGB", 3), ("RGBA", 4)):
        im = Image.fromarray(pic(w, h, ch), mode)
        im.save(os.path.join(out, f"s{n}_{mode}_{w}x{h}.png")); n += 1
        if mode != "RGBA":
            for kw in (dict(quality=85, subsampling=0), dict(quality=60, subsampling=1), dict(quality=40, subsampling=2),
                       dict(quality=75, subsampling=2, progressive=True), dict(quality=90, subsampling=0, progressive=True, optimize=True)):
                im.save(os.path.join(out, f"s{n}_{mode}_{w}x{h}.jpg"), **kw); n += 1
im = Image.fromarray(pic(40, 24, 3), "RGB")
im.convert("P", palette=Image.ADAPTIVE, colors=16).save(os.path.join(out, f"s{n}_pal16.png")); n += 1
im.convert("P", palette=Image.ADAPTIVE, colors=200).save(os.path.join(out, f"s{n}_pal200.png")); n += 1
im.convert("1").save(os.path.join(out, f"s{n}_1bit.png")); n += 1
im.convert("LA").save(os.path.join(out, f"s{n}_la.png")); n += 1
Image.fromarray((pic(24, 24, 1).astype(np.uint16) * 257), "I;16").save(os.path.join(out, f"s{n}_gray16.png")); n += 1
im.save(os.path.join(out, f"s{n}_nocompress.png"), compress_level=0); n += 1
im.save(os.path.join(out, f"s{n}_best.png"), compress_level=9, optimize=True); n += 1
im.convert("CMYK").save(os.path.join(out, f"s{n}_cmyk.jpg")); n += 1
big = Image.fromarray(pic(160, 96, 3), "RGB")
big.save(os.path.join(out, f"s{n}_restart.jpg"), quality=70, subsampling=2, restart_marker_blocks=3); n += 1
big.save(os.path.join(out, f"s{n}_restart_prog.jpg"), quality=70, subsampling=1, progressive=True, restart_marker_rows=1); n += 1
print(n, "image seeds in", out)
