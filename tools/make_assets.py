"""Decodes the reference's earth texture (image_texture/earthmap.jpg, used by
src/main.rs:445,665) once with PIL/libjpeg and stores the raw RGB8 texels so that
the CPU oracle and the GPU read identical bytes on any machine. The reference's own
decoder (image 0.23.7 / jpeg-decoder 0.1.20) may differ from libjpeg by a level or
two per texel; parity with it is unpinned.

Run in the build container only (needs /root/reference)."""
import hashlib
import os
import sys

import numpy as np
from PIL import Image

SRC = "/root/reference/image_texture/earthmap.jpg"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "raytracer_amd", "assets", "earthmap_1024x512.rgb")


def main():
    img = np.asarray(Image.open(SRC).convert("RGB"))
    assert img.shape == (512, 1024, 3), img.shape
    raw = img.tobytes()
    with open(DST, "wb") as f:
        f.write(raw)
    print("wrote", os.path.normpath(DST), len(raw), "bytes; sha256", hashlib.sha256(raw).hexdigest()[:16],
          "mean", img.reshape(-1, 3).mean(0))


if __name__ == "__main__":
    sys.exit(main())
