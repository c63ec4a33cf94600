"""ORACLE (test infrastructure only) — the same DenseNet-121 ``.features`` graph
through ``torch.nn.functional`` on CPU (oneDNN), the closest available stand-in
for the reference's MXNet+MKL-DNN CPU path (``--num_gpus 0`` -> ``[mx.cpu()]``,
reference evaluate.py:85), which cannot be installed here.

Used (a) as an independent cross-check of oracle/densenet_np.py and (b) by
bench.py's ``cpu_baseline`` leg.  PARITY UNPINNED — see densenet_np.py header.
Never imported by the product path.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


class TorchDenseNet121:
    def __init__(self, p: dict, prefix: str = "densenet0_"):
        self.p = {k: torch.from_numpy(v) for k, v in p.items() if k.startswith(prefix)}
        self.pre = prefix

    def _bn(self, x, n):
        p = self.p
        return F.batch_norm(x, p[n + "_running_mean"], p[n + "_running_var"], p[n + "_gamma"], p[n + "_beta"],
                            False, 0.0, 1e-5)

    @torch.no_grad()
    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        p, pre = self.p, self.pre
        x = F.conv2d(x, p[pre + "conv0_weight"], stride=2, padding=3)
        x = F.max_pool2d(F.relu(self._bn(x, pre + "batchnorm0")), 3, 2, 1)
        outer = 1
        for st, nl in enumerate((6, 12, 24, 16), 1):
            sp = f"{pre}stage{st}_"
            for li in range(nl):
                y = F.conv2d(F.relu(self._bn(x, f"{sp}batchnorm{2 * li}")), p[f"{sp}conv{2 * li}_weight"])
                y = F.conv2d(F.relu(self._bn(y, f"{sp}batchnorm{2 * li + 1}")), p[f"{sp}conv{2 * li + 1}_weight"],
                             padding=1)
                x = torch.cat([x, y], 1)
            if st != 4:
                x = F.conv2d(F.relu(self._bn(x, f"{pre}batchnorm{outer}")), p[f"{pre}conv{outer}_weight"])
                x = F.avg_pool2d(x, 2, 2)
                outer += 1
        x = F.avg_pool2d(F.relu(self._bn(x, f"{pre}batchnorm{outer}")), 7)
        return x.flatten(1)
