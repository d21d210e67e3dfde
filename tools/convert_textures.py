#!/usr/bin/env python3
"""Writes an 8-bit binary PPM (P6) next to every JPG / PNG texture a scene's .mtl names with map_Kd, for the command-line
front end, whose loader reads PPM / PFM only (no JPEG or PNG codec in the C++ toolchain of this image).  The PPM holds the
decoded 8-bit sRGB samples; the loader applies the reference's / 255 and sRGB -> linear (src/bitmap.cpp:68-93) itself, and
falls back from <name>.jpg / .png to <name>.ppm on its own.
Usage: tools/convert_textures.py <scene dir> [<name>.mtl]   (output goes to --out DIR, default the scene dir)"""
import argparse
import glob
import os

from PIL import Image

ap = argparse.ArgumentParser()
ap.add_argument("scene_dir")
ap.add_argument("mtl", nargs="?")
ap.add_argument("--out", default=None)
a = ap.parse_args()
mtls = [os.path.join(a.scene_dir, a.mtl)] if a.mtl else glob.glob(os.path.join(a.scene_dir, "*.mtl"))
done = set()
for mtl in mtls:
    for line in open(mtl):
        tok = line.split()
        if tok and tok[0] == "map_Kd":
            rel = tok[-1].replace("\\", "/")
            src = os.path.join(a.scene_dir, rel)
            dst = os.path.splitext(os.path.join(a.out or a.scene_dir, rel))[0] + ".ppm"
            if src in done:
                continue
            done.add(src)
            if not os.path.exists(src):
                print("missing:", src)
                continue
            os.makedirs(os.path.dirname(dst), exist_ok=True)
            with Image.open(src) as im:
                rgb = im.convert("RGB")
                with open(dst, "wb") as f:
                    f.write(b"P6\n%d %d\n255\n" % rgb.size)
                    f.write(rgb.tobytes())
            print(src, "->", dst, rgb.size)
