"""``gluoncv.model_zoo.get_model`` stand-in for the one backbone on the hot path.

Reference call sites: evaluate.py:125, train.py:204, train_gnmt.py:150.  There is
no network, so ``pretrained=True`` cannot download ImageNet weights: it yields the
seeded synthetic parameters (tennis_amd.weights) unless ``params_file`` points at a
parameter container with Gluon names.
"""
from .nn import DenseNet121Backbone


class _ZooModel:
    def __init__(self, features):
        self.features = features


def get_model(name, pretrained=False, seed=0, params_file=None, **kwargs):
    if name.lower() != "densenet121":
        raise NotImplementedError(f"backbone '{name}' is outside the accelerated hot path "
                                  "(only DenseNet121, BASELINE.json north_star)")
    feats = DenseNet121Backbone(seed=seed, **kwargs)
    feats.initialize()
    if params_file is not None:
        feats.load_parameters(params_file)
    return _ZooModel(feats)
