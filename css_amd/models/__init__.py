"""Model registry -- same shape as css/models/__init__.py:19-27 (MODELS dict + build_model)."""
from . import conformer

MODELS = {
    "Conformer": conformer.Conformer,
}


def build_model(modelname, conf):
    return MODELS[modelname].build_model(conf)
