"""Synthetic frame families for the calibrated fp16 conversion (tennis_amd/calibrate.py) and its robustness tests.

The reference evaluates fp32 parameters on whatever frames arrive (models/vision/definitions.py:27-33); a conversion that is
calibrated on frames must therefore hold on frames that do not look like its calibration set.  No video ships with the
reference, so the families below stand in for "different kinds of content": what matters to the conversion is how the
per-channel mean activations of a frame differ from those it was calibrated on (dark / bright scenes, flat regions, edges,
texture).  All generators are seeded and return NHWC uint8.
"""
from __future__ import annotations

import numpy as np

# the families of the built-in calibration set
FAMILIES = ["noise", "lowcontrast", "constant", "gradient", "halfblack", "blobs", "scene", "stripes", "checker", "dark", "bright", "tinted"]


def _noise(rng, n, s):
    return rng.integers(0, 256, (n, s, s, 3), dtype=np.uint8)


def _lowcontrast(rng, n, s):
    base = rng.integers(60, 200, (n, 1, 1, 3))
    return np.clip(base + rng.normal(0, 12, (n, s, s, 3)), 0, 255).astype(np.uint8)


def _constant(rng, n, s):
    return np.broadcast_to(rng.integers(0, 256, (n, 1, 1, 3), dtype=np.uint8), (n, s, s, 3)).copy()


def _gradient(rng, n, s):
    t = np.linspace(0.0, 1.0, s)
    out = np.empty((n, s, s, 3), np.float64)
    for i in range(n):
        c0, c1 = rng.uniform(0, 255, 3), rng.uniform(0, 255, 3)
        ang = rng.uniform(0, 2 * np.pi)
        g = np.clip(0.5 + (np.cos(ang) * (t[None, :] - 0.5) + np.sin(ang) * (t[:, None] - 0.5)), 0, 1)
        out[i] = c0 + g[..., None] * (c1 - c0)
    return out.astype(np.uint8)


def _halfblack(rng, n, s):
    f = _noise(rng, n, s)
    for i in range(n):
        if i & 1:
            f[i, :, : s // 2] = 0
        else:
            f[i, : s // 2] = 0
    return f


def _blobs(rng, n, s):
    """low-pass filtered noise (1/f-like): smooth coloured regions with soft edges"""
    fy = np.fft.fftfreq(s)[:, None]
    fx = np.fft.rfftfreq(s)[None, :]
    amp = 1.0 / np.maximum(np.hypot(fy, fx), 1.0 / s) ** 1.5
    out = np.empty((n, s, s, 3), np.float64)
    for i in range(n):
        for c in range(3):
            spec = (rng.normal(size=(s, s // 2 + 1)) + 1j * rng.normal(size=(s, s // 2 + 1))) * amp
            img = np.fft.irfft2(spec, (s, s))
            img = (img - img.mean()) / (img.std() + 1e-9)
            out[i, ..., c] = 128 + rng.uniform(-40, 40) + img * rng.uniform(25, 70)
    return np.clip(out, 0, 255).astype(np.uint8)


def _scene(rng, n, s):
    """a court-like composition: large flat rectangles, thin bright lines, a textured band, sensor noise"""
    out = np.empty((n, s, s, 3), np.float64)
    for i in range(n):
        img = np.empty((s, s, 3))
        img[:] = rng.uniform(20, 120, 3)
        for _ in range(int(rng.integers(2, 6))):
            y0, x0 = rng.integers(0, s - 8, 2)
            h, w = rng.integers(8, s // 2, 2)
            img[y0:y0 + h, x0:x0 + w] = rng.uniform(0, 255, 3)
        for _ in range(int(rng.integers(2, 7))):
            if rng.random() < 0.5:
                y = int(rng.integers(0, s - 2)); img[y:y + 2, :] = rng.uniform(200, 255)
            else:
                x = int(rng.integers(0, s - 2)); img[:, x:x + 2] = rng.uniform(200, 255)
        y0 = int(rng.integers(0, s - 32))
        img[y0:y0 + 32] += rng.normal(0, 40, (32, s, 3))
        out[i] = img + rng.normal(0, 4, (s, s, 3))
    return np.clip(out, 0, 255).astype(np.uint8)


def _stripes(rng, n, s):
    out = np.empty((n, s, s, 3), np.float64)
    t = np.arange(s)
    for i in range(n):
        period = rng.integers(2, 40)
        c0, c1 = rng.uniform(0, 255, 3), rng.uniform(0, 255, 3)
        m = ((t // period) & 1).astype(np.float64)
        m = m[None, :] if rng.random() < 0.5 else m[:, None]
        out[i] = c0 + np.broadcast_to(m, (s, s))[..., None] * (c1 - c0)
    return out.astype(np.uint8)


def _checker(rng, n, s):
    out = np.empty((n, s, s, 3), np.float64)
    t = np.arange(s)
    for i in range(n):
        p = rng.integers(1, 32)
        c0, c1 = rng.uniform(0, 255, 3), rng.uniform(0, 255, 3)
        m = (((t[:, None] // p) + (t[None, :] // p)) & 1).astype(np.float64)
        out[i] = c0 + m[..., None] * (c1 - c0)
    return out.astype(np.uint8)


def _dark(rng, n, s):
    return (_scene(rng, n, s).astype(np.float64) * rng.uniform(0.08, 0.3, (n, 1, 1, 1))).astype(np.uint8)


def _bright(rng, n, s):
    return (255 - (255 - _blobs(rng, n, s).astype(np.float64)) * rng.uniform(0.1, 0.35, (n, 1, 1, 1))).astype(np.uint8)


def _tinted(rng, n, s):
    """noise through a random per-frame colour transform (gain + offset per channel)"""
    f = _noise(rng, n, s).astype(np.float64)
    return np.clip(f * rng.uniform(0.2, 1.0, (n, 1, 1, 3)) + rng.uniform(0, 120, (n, 1, 1, 3)), 0, 255).astype(np.uint8)


def _text(rng, n, s):
    """dark glyph-like rectangles on a light page"""
    out = np.empty((n, s, s, 3), np.float64)
    for i in range(n):
        img = np.empty((s, s, 3)); img[:] = rng.uniform(190, 255, 3)
        ink = rng.uniform(0, 60, 3)
        for y in range(6, s - 10, int(rng.integers(10, 18))):
            x = 6
            while x < s - 10:
                w = int(rng.integers(2, 9))
                if rng.random() < 0.8:
                    img[y:y + int(rng.integers(5, 9)), x:x + w] = ink
                x += w + int(rng.integers(1, 5))
        out[i] = img
    return out.astype(np.uint8)


def _saturated(rng, n, s):
    """patchwork of fully saturated primaries / secondaries"""
    pal = np.array([[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 0], [0, 255, 255], [255, 0, 255], [255, 255, 255], [0, 0, 0]], np.float64)
    out = np.empty((n, s, s, 3), np.float64)
    for i in range(n):
        p = int(rng.integers(8, 80))
        idx = rng.integers(0, len(pal), ((s + p - 1) // p, (s + p - 1) // p))
        out[i] = pal[np.repeat(np.repeat(idx, p, 0), p, 1)[:s, :s]]
    return out.astype(np.uint8)


def _photo(rng, n, s):
    """photo-like: blobs + a few hard-edged objects + film grain + vignette"""
    base = _blobs(rng, n, s).astype(np.float64)
    yy, xx = np.mgrid[0:s, 0:s]
    for i in range(n):
        for _ in range(int(rng.integers(1, 5))):
            cy, cx, r = rng.integers(0, s), rng.integers(0, s), rng.integers(6, s // 3)
            m = (yy - cy) ** 2 + (xx - cx) ** 2 < r * r
            base[i][m] = base[i][m] * 0.3 + rng.uniform(0, 255, 3) * 0.7
        vign = 1.0 - rng.uniform(0.0, 0.5) * (((yy - s / 2) ** 2 + (xx - s / 2) ** 2) / (s * s / 2.0))
        base[i] = base[i] * vign[..., None] + rng.normal(0, rng.uniform(1, 10), (s, s, 3))
    return np.clip(base, 0, 255).astype(np.uint8)


# families that are NOT part of the built-in calibration set: the robustness tests evaluate on them
HELD_OUT = ["text", "saturated", "photo"]

_GEN = {"text": _text, "saturated": _saturated, "photo": _photo,
        "stripes": _stripes, "checker": _checker, "dark": _dark, "bright": _bright, "tinted": _tinted, "noise": _noise,
        "lowcontrast": _lowcontrast, "constant": _constant, "gradient": _gradient, "halfblack": _halfblack,
        "blobs": _blobs, "scene": _scene}


def frames(family: str, n: int, size: int = 224, seed: int = 0) -> np.ndarray:
    """``n`` NHWC uint8 frames of one family."""
    if family not in _GEN:
        raise ValueError(f"unknown frame family {family!r}; one of {FAMILIES}")
    rng = np.random.default_rng([seed, (FAMILIES + HELD_OUT).index(family)])
    return np.ascontiguousarray(_GEN[family](rng, n, size))


def default_calibration_frames(size: int = 224, n: int = 28, seed: int = 4321) -> np.ndarray:
    """The built-in calibration set: ``n`` frames dealt round-robin over all families (``n`` = 28: four of each).  A caller
    with real frames of the footage to be processed should ADD those (calibrate(frames) concatenates)."""
    per = [(n + len(FAMILIES) - 1 - i) // len(FAMILIES) for i in range(len(FAMILIES))]
    return np.concatenate([frames(f, k, size, seed) for f, k in zip(FAMILIES, per) if k > 0])


def fine_checkerboards(n: int, size: int = 224, seed: int = 0) -> np.ndarray:
    """Full-contrast checkerboards with cells of 2, 3 and 1 px in turn: the frames on which the calibrated fp16 conversion is
    weakest (round 5, profiles/r05_parity_wide.json: an exactly periodic pattern makes the same weight-rounding error in every
    pixel of a phase).  Colours black / white or two saturated primaries, random phase."""
    rng = np.random.default_rng([seed, 977])
    t = np.arange(size)
    out = np.empty((n, size, size, 3), np.uint8)
    pal = np.array([[255, 255, 255], [0, 0, 0], [255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 0], [0, 255, 255], [255, 0, 255]], np.uint8)
    for i in range(n):
        p = (2, 3, 1)[i % 3]
        oy, ox = rng.integers(0, 2 * p, 2)
        m = ((((t[:, None] + oy) // p) + ((t[None, :] + ox) // p)) & 1).astype(bool)
        c0, c1 = (pal[0], pal[1]) if i % 2 == 0 else pal[rng.choice(len(pal), 2, replace=False)]
        out[i] = np.where(m[..., None], c1, c0)
    return out


def frames_from_npz(path: str, n: int, size: int = 224, skip: int = 0) -> np.ndarray:
    """``n`` frames made of the decoded RGB images of an ``.npz`` of fixtures (keys ending in ``__rgb``, e.g. the repo's
    tests/golden/jpeg_cases.npz): enlarged 2-4x (nearest neighbour) and tiled to ``size`` x ``size``."""
    z = np.load(path)
    imgs = [z[k] for k in sorted(z.files) if k.endswith("__rgb") and z[k].shape[0] >= 24]
    out = []
    for i in range(skip, skip + n):
        im = imgs[i % len(imgs)]
        k = 2 + i % 3
        big = np.repeat(np.repeat(im, k, 0), k, 1)
        reps = (-(-size // big.shape[0]), -(-size // big.shape[1]), 1)
        out.append(np.tile(big, reps)[:size, :size])
    return np.ascontiguousarray(np.stack(out))


def mixed_batch(n: int, size: int = 224, seed: int = 0, jpeg_npz: str | None = None, fine: int | None = None):
    """``n`` frames dealt over every family (calibration families, held-out families, ``jpeg`` when a fixture file is given) plus
    ``fine`` fine checkerboards (default n // 16): the content mix the parity of the timed configuration is measured on (VERDICT r5
    item 1).  Returns ``(frames uint8 NHWC, labels)`` with ``labels[i]`` the family of frame i."""
    fams = FAMILIES + HELD_OUT + (["jpeg"] if jpeg_npz else [])
    fine = n // 16 if fine is None else fine
    rest = n - fine
    per = [(rest + len(fams) - 1 - i) // len(fams) for i in range(len(fams))]
    parts, labels = [], []
    for f, k in zip(fams, per):
        if k <= 0:
            continue
        parts.append(frames_from_npz(jpeg_npz, k, size, skip=seed % 7) if f == "jpeg" else frames(f, k, size, seed))
        labels += [f] * k
    if fine > 0:
        parts.append(fine_checkerboards(fine, size, seed))
        labels += ["finechecker"] * fine
    return np.ascontiguousarray(np.concatenate(parts)), labels
