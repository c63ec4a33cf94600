"""Evaluation / feature-extraction driver — mirror of reference evaluate.py for the
hot path: same flag names, ``save_features`` (evaluate.py:306-321) and
``evaluate_model`` (evaluate.py:274-303) with the same results.

Differences that are deliberate: one device per rank
(no serialised per-device inner loop, no ``idxs`` rebinding bug), one D2H copy per
batch instead of one per row, frames sharded by batch across ranks.
"""
from __future__ import annotations

import argparse
import os
import time

import numpy as np
import torch

from . import transforms
from .dataset import DataLoader, TennisSet
from .metrics.vision import PRF1
from .model_zoo import get_model
from .models.vision.definitions import CNNRNN, FrameModel, TemporalPooling


def evaluate_model(net, loader, dataset, metrics, ctx=None):
    """Reference evaluate.py:274-303: forward, metric.update, results[img_path] = raw
    logits, ground_truths[img_path] = class index."""
    results, ground_truths = dict(), dict()
    for data, labels, idxs in loader:
        outputs = net(data)
        lab = torch.from_numpy(labels).to(outputs.device)
        for metric in metrics:
            metric.update([lab], [outputs])
        out = outputs.cpu().numpy()
        for i, idx in enumerate(int(j) for j in idxs):
            sample = dataset._samples[idx]
            img_path = dataset.get_image_path(dataset._frames_dir, sample[0], sample[1])
            results[img_path] = out[i]
            ground_truths[img_path] = dataset.classes.index(sample[2])
    return results, ground_truths


def save_features(net, loader, dataset, ctx=None, verbose=True):
    """Reference evaluate.py:306-321: feat = net.backbone(x); one float32 ``.npy`` per
    frame at save_feature_path(idx), skipped when the file already exists."""
    written = 0
    for data, _labels, idxs in loader:
        feat = net.backbone(data).cpu().numpy()
        for i, idx in enumerate(int(j) for j in idxs):
            feat_path = dataset.save_feature_path(idx)
            if not os.path.exists(feat_path):
                os.makedirs(os.path.dirname(feat_path), exist_ok=True)
                np.save(feat_path, feat[i])
                written += 1
                if verbose:
                    print("Saving %s" % feat_path)
    return written


def build_parser():
    p = argparse.ArgumentParser(description="tennis_amd evaluate (flags of reference evaluate.py:30-75)")
    p.add_argument("--backbone", default="DenseNet121")
    p.add_argument("--model_id", default="0000")
    p.add_argument("--split_id", default="02")
    p.add_argument("--split", default="test")
    p.add_argument("--data_shape", type=int, default=224)
    p.add_argument("--every", default="1, 1, 1")
    p.add_argument("--window", type=int, default=1)
    p.add_argument("--padding", type=int, default=1)
    p.add_argument("--stride", type=int, default=1)
    p.add_argument("--batch_size", type=int, default=64)
    p.add_argument("--num_gpus", type=int, default=1)
    p.add_argument("--save_feats", action="store_true")
    p.add_argument("--feats_model", default=None)
    p.add_argument("--temp_pool", default=None, help="mean, max, gru or lstm")
    p.add_argument("--root", default="data")
    p.add_argument("--frames_per_video", type=int, default=16)
    return p


def main(argv=None):
    flags = build_parser().parse_args(argv)
    every = [int(s) for s in flags.every.split(",")]
    # evaluate.py:91-98: Resize(s + 32) / CenterCrop(s) / ToTensor / Normalize, here one GPU launch per batch; the
    # synthetic source already produces data_shape frames, so only data on disk goes through it
    transform_test = None
    if flags.feats_model is None and os.path.exists(os.path.join(flags.root, "splits", flags.split_id, flags.split + ".txt")):
        transform_test = transforms.Compose([transforms.Resize(flags.data_shape + 32),
                                             transforms.CenterCrop(flags.data_shape),
                                             transforms.ToTensor(),
                                             transforms.Normalize([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])])
    test_set = TennisSet(root=flags.root, transform=transform_test, split=flags.split, every=every[2], padding=flags.padding,
                         stride=flags.stride, window=flags.window, model_id=flags.model_id,
                         split_id=flags.split_id, balance=False, feats_model=flags.feats_model,
                         save_feats=flags.save_feats, data_shape=flags.data_shape,
                         frames_per_video=flags.frames_per_video)
    test_data = DataLoader(test_set, batch_size=flags.batch_size, shuffle=False)

    model = None
    if flags.feats_model is None:                                           # evaluate.py:118-135
        backbone_net = get_model(flags.backbone, pretrained=True).features
        model = FrameModel(backbone_net, len(test_set.classes))
    elif flags.temp_pool in ["max", "mean"]:                                # evaluate.py:136-138
        backbone_net = get_model(flags.backbone, pretrained=True).features
        model = FrameModel(backbone_net, len(test_set.classes))
    if flags.window > 1:                                                    # evaluate.py:139-162
        if flags.temp_pool in ["gru", "lstm"]:
            model = CNNRNN(model, num_classes=len(test_set.classes), type=flags.temp_pool, hidden_size=128)
        elif flags.temp_pool not in ["mean", "max"]:
            raise AssertionError("window > 1 needs --temp_pool (the 3-D rdnet backbone is out of scope)")
    model.initialize()
    model.hybridize()

    if flags.save_feats:                                                    # evaluate.py:186-204
        n = save_features(model, test_data, test_set)
        print("wrote %d feature files under %s" % (n, test_set.feat_dir))
        return 0

    if flags.temp_pool in ["max", "mean"]:                                  # evaluate.py:242-244
        model = TemporalPooling(model, pool=flags.temp_pool, num_classes=0, feats=flags.feats_model is not None)
    test_metrics = [PRF1(label_names=test_set.classes)]
    tic = time.time()
    results, gts = evaluate_model(model, test_data, test_set, test_metrics)
    str_ = "Test set:"                                                       # evaluate.py:250-255
    for i in range(len(test_set.classes)):
        str_ += "\n"
        for j in range(len(test_set.classes)):
            str_ += str(test_metrics[0].mat[i, j]) + "\t"
    print(str_)
    str_ = "[Finished] "
    for res in test_metrics[0].get():
        str_ += ", Test_{}={:.3f}".format(res[0], res[1])
    str_ += "  # Samples: {}, Time Taken: {:.1f}".format(len(test_set), time.time() - tic)
    print(str_)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
