"""ORACLE (test infrastructure only) — one fine-tuning step of ``FrameModel(DenseNet121.features, 11)`` on the CPU with
torch autograd (float64), BatchNorm in TRAINING mode (batch statistics, running statistics updated):

  model    reference models/vision/definitions.py:10-33 over gluoncv DenseNet-121 ``.features`` [EXT]
  loss     gluon.loss.SoftmaxCrossEntropyLoss per sample, ``ag.backward`` of the per-sample losses = gradient of their SUM
           (train.py:324,419-421)
  update   gluon.Trainer 'sgd' .step(batch_size): rescale 1/batch_size, momentum, wd (train.py:298-299,424) — the update
           itself is oracle/train_np.py::sgd_momentum
  BN       gluon nn.BatchNorm(momentum=0.9, epsilon=1e-5) [EXT]: normalise with the biased batch variance;
           running = 0.9 * running + 0.1 * batch (MXNet keeps the biased variance in the moving average)

PARITY UNPINNED against MXNet (absent); the backward pass is torch autograd's.  Only tests/ may import this module.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

BN_NAMES = ("_gamma", "_beta")


def forward(params: dict, x_nchw, prefix="densenet0_", cls_prefix="framemodel0_dense0_", dtype=torch.float64):
    """-> (logits, leaf tensors w, batch statistics {bn name: (mean, biased var)})"""
    w = {k: torch.tensor(np.asarray(v), dtype=dtype, requires_grad=not k.endswith(("_running_mean", "_running_var")))
         for k, v in params.items()}
    stats = {}

    def bn_relu(x, n):
        mean = x.mean(dim=(0, 2, 3))
        var = x.var(dim=(0, 2, 3), unbiased=False)
        stats[n] = (mean.detach().numpy(), var.detach().numpy())
        xh = (x - mean[None, :, None, None]) / torch.sqrt(var[None, :, None, None] + 1e-5)
        return F.relu(xh * w[n + "_gamma"][None, :, None, None] + w[n + "_beta"][None, :, None, None])

    pre = prefix
    x = torch.tensor(np.asarray(x_nchw), dtype=dtype)
    x = F.conv2d(x, w[pre + "conv0_weight"], stride=2, padding=3)
    x = F.max_pool2d(bn_relu(x, pre + "batchnorm0"), 3, 2, 1)
    outer = 1
    for st, nl in enumerate((6, 12, 24, 16), 1):
        sp = f"{pre}stage{st}_"
        for li in range(nl):
            y = F.conv2d(bn_relu(x, f"{sp}batchnorm{2 * li}"), w[f"{sp}conv{2 * li}_weight"])
            y = F.conv2d(bn_relu(y, f"{sp}batchnorm{2 * li + 1}"), w[f"{sp}conv{2 * li + 1}_weight"], padding=1)
            x = torch.cat([x, y], 1)
        if st != 4:
            x = F.conv2d(bn_relu(x, f"{pre}batchnorm{outer}"), w[f"{pre}conv{outer}_weight"])
            x = F.avg_pool2d(x, 2, 2)
            outer += 1
    x = F.avg_pool2d(bn_relu(x, f"{pre}batchnorm{outer}"), x.shape[-1]).flatten(1)
    logits = x @ w[cls_prefix + "weight"].T + w[cls_prefix + "bias"]
    return logits, w, stats


def loss_and_grads(params, x_nchw, labels, prefix="densenet0_", cls_prefix="framemodel0_dense0_"):
    logits, w, stats = forward(params, x_nchw, prefix, cls_prefix)
    loss = F.cross_entropy(logits, torch.tensor(np.asarray(labels), dtype=torch.long), reduction="none")
    loss.sum().backward()
    g = {k: v.grad.numpy() for k, v in w.items() if v.requires_grad}
    return loss.detach().numpy(), logits.detach().numpy(), g, stats
