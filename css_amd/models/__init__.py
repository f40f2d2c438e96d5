"""Model registry -- same shape as css/models/__init__.py:19-27 (MODELS dict + build_model)."""
from . import blstm, conformer

MODELS = {
    "Conformer": conformer.Conformer,
    "BLSTM": blstm.BLSTM,
}


def build_model(modelname, conf):
    return MODELS[modelname].build_model(conf)
