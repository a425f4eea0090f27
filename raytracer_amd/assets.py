"""Data files shipped with the package."""
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


def earthmap():
    """The earth texture of src/main.rs:445,665 as uint8 [512, 1024, 3] (decoded once by tools/make_assets.py)."""
    path = os.path.join(_HERE, "assets", "earthmap_1024x512.rgb")
    return np.fromfile(path, dtype=np.uint8).reshape(512, 1024, 3)
