"""Decodes the reference's earth-map.jpg (crates/cli/assets/earth-map.jpg, 1024x512 RGB) once, in the
build container, and stores the raw texels so that no JPEG decoder is needed at test / bench time.
The reference decodes with image 0.25.9 / zune-jpeg; PIL's decoder can differ by 1-2 LSB per texel,
which no test here depends on (both the oracle and the GPU path read this same array).

Run from the repo root:  python tests/golden/make_earth_fixture.py
"""
import os
import sys

import numpy as np
from PIL import Image

src = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/crates/cli/assets/earth-map.jpg"
img = np.asarray(Image.open(src).convert("RGB"), dtype=np.uint8)
assert img.shape == (512, 1024, 3), img.shape
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "earth_map_texels.npz")
np.savez_compressed(out, rgb=img)
print(out, os.path.getsize(out))
