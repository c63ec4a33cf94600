"""Writes tests/golden/jpeg_cases.npz: small JPEG files encoded by Pillow (libjpeg-turbo) together with Pillow's own
decode of them.  Pillow decodes with libjpeg's default parameters (JDCT_ISLOW, fancy upsampling) - the same library
family and settings as the reference's ``mx.image.imread`` (OpenCV imdecode -> libjpeg, reference dataset.py:204) - so
these pairs pin oracle/jpeg_np.py and the device decoder to the real decoder's arithmetic.

    python tests/golden/make_jpeg_fixtures.py      (needs Pillow; run in the build container)
"""
import io
import os

import numpy as np
from PIL import Image, features

CASES = [  # name, (h, w), kind, quality, subsampling (0 = 4:4:4, 1 = 4:2:2, 2 = 4:2:0, None = grey), extra save options
    ("c444_q90", (40, 56), "smooth", 90, 0, {}),
    ("c422_q75", (33, 47), "smooth", 75, 1, {}),
    ("c420_q85", (37, 53), "smooth", 85, 2, {}),
    ("c420_noise_q50", (48, 64), "noise", 50, 2, {}),
    ("c420_q100", (24, 24), "noise", 100, 2, {}),
    ("c420_tiny", (3, 5), "noise", 80, 2, {}),
    ("c420_1x1", (1, 1), "noise", 80, 2, {}),
    ("c422_narrow", (9, 4), "noise", 80, 1, {}),
    ("c420_optimized", (37, 53), "smooth", 80, 2, {"optimize": True}),
    ("c420_restart_blocks", (40, 72), "smooth", 80, 2, {"restart_marker_blocks": 2}),
    ("c444_restart_rows", (40, 72), "noise", 70, 0, {"restart_marker_rows": 1}),
    ("grey_q80", (35, 50), "smooth", 80, None, {}),
]


def image(rng, h, w, kind):
    if kind == "noise":
        return (rng.random((h, w, 3)) * 255).astype(np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    a = np.stack([127 + 120 * np.sin(xx / 7.0 + yy / 11.0), 127 + 120 * np.cos(xx / 5.0 - yy / 9.0), (xx * 3 + yy * 5) % 256], -1)
    return np.clip(a + rng.normal(0, 6, (h, w, 3)), 0, 255).astype(np.uint8)


def main():
    rng = np.random.default_rng(20240928)
    out = {}
    for name, (h, w), kind, q, ss, extra in CASES:
        a = image(rng, h, w, kind)
        buf = io.BytesIO()
        if ss is None:
            Image.fromarray(a[:, :, 0]).save(buf, "JPEG", quality=q, **extra)
        else:
            Image.fromarray(a).save(buf, "JPEG", quality=q, subsampling=ss, **extra)
        data = buf.getvalue()
        ref = np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))
        out[name + "__jpeg"] = np.frombuffer(data, np.uint8)
        out[name + "__rgb"] = ref
    out["decoder"] = np.array("Pillow %s, libjpeg-turbo %s" % (Image.__version__ if hasattr(Image, "__version__") else "?",
                                                               features.version("jpg")))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "jpeg_cases.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(CASES), "cases")


if __name__ == "__main__":
    main()
