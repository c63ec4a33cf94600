"""A tiny copy of the reference's ``data/`` directory for the captioning pipeline tests (data/README.md layout): three videos of
JPEG frames, label files, splits/02/{train,val,test}.txt, annotations/points.txt + captions.txt and a word-embedding file in the
text format train_embeddings.py writes (``token v1 ... vD`` per line, rows L2-normalised, as data/embeddings-ex.txt)."""
import os

import numpy as np

CAPTIONS = {"train": ["the near player serves wide", "far player hits a forehand return", "near player hits a backhand into the net",
                      "the far player serves an ace", "the near player hits a forehand winner down the line"],
            "val": ["the far player serves wide", "near player hits a forehand into the net"],
            "test": ["the near player serves an ace", "far player hits a backhand return wide"]}


def write(root, rng, frame_size=(48, 64), emb_dim=12, frames_per_point=4, missing=("winner",)):
    """-> {"points": {split: [(pid, video, start, end, caption)]}, "emb": {token: vector}}; tokens in ``missing`` are left out of
    the embedding file (``Vocab.set_embedding`` gives them the zero vector)."""
    from PIL import Image
    classes = ["OTH", "SFI", "SFF", "SFL", "SNI", "SNF", "SNL", "HFL", "HFR", "HNL", "HNR"]
    os.makedirs(os.path.join(root, "splits", "02"), exist_ok=True)
    os.makedirs(os.path.join(root, "annotations", "labels"), exist_ok=True)
    with open(os.path.join(root, "classes.names"), "w") as f:
        f.write("\n".join(classes) + "\n")
    points, pts_lines, cap_lines = {}, [], []
    for vi, split in enumerate(("train", "val", "test")):
        v = f"V{10 + vi:03d}"
        n = len(CAPTIONS[split]) * (frames_per_point + 1) + 2
        with open(os.path.join(root, "annotations", "labels", v + ".txt"), "w") as f:
            for fr in range(n):
                f.write(f"{fr} {classes[1 + fr % 10] if fr % (frames_per_point + 1) else 'OTH'}\n")
        for fr in range(n):
            path = os.path.join(root, "frames", v + ".mp4", "0000000000", f"{fr:010d}.jpg")
            os.makedirs(os.path.dirname(path), exist_ok=True)
            Image.fromarray(rng.integers(0, 256, frame_size + (3,), dtype=np.uint8)).save(path, quality=90)
        with open(os.path.join(root, "splits", "02", split + ".txt"), "w") as f:
            f.write("\n".join(f"{v} {fr}" for fr in range(n)) + "\n")
        points[split] = []
        for i, cap in enumerate(CAPTIONS[split]):
            start = 1 + i * (frames_per_point + 1)
            pid = f"P{split}{i}"
            points[split].append((pid, v, start, start + frames_per_point, cap))
            pts_lines.append(f"{pid} {v} {start} {start + frames_per_point}")
            cap_lines.append(f"{pid}\t{cap}")
    with open(os.path.join(root, "annotations", "points.txt"), "w") as f:
        f.write("\n".join(pts_lines) + "\n")
    with open(os.path.join(root, "annotations", "captions.txt"), "w") as f:
        f.write("\n".join(cap_lines) + "\n")
    words = sorted({w for caps in CAPTIONS.values() for c in caps for w in c.split()} - set(missing)) + ["court", "lob"]
    emb = {}
    with open(os.path.join(root, "embeddings-ex.txt"), "w") as f:
        for w in words:
            vec = rng.normal(0, 1, emb_dim)
            vec = (vec / np.linalg.norm(vec)).astype(np.float32)
            emb[w] = vec
            f.write(w + " " + " ".join(repr(float(x)) for x in vec) + "\n")
    return {"points": points, "emb": emb}
