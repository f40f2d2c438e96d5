"""Objective registry -- same shape as css/objectives/__init__.py:19-23."""
from . import mse, pit
from .pit import pit_loss  # noqa: F401

OBJECTIVES = {"MSE": mse.MeanSquaredError}


def build_objective(objectivename, conf):
    return OBJECTIVES[objectivename].build_objective(conf)
