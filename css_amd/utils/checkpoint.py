"""Checkpoint tools in the reference's on-disk format.

scripts/python/train.py:304-317 saves ``{"model", "objective", "optimizer", "lr_sched", "epoch", "loss"}`` to
``{expdir}/{epoch+1}.{job}.mdl``; scripts/python/combine_models.py:55-110 averages the job checkpoints of one
epoch into ``{epoch}.mdl``.  css_amd models use the reference's state_dict keys, so files written by either
side load in the other.
"""
import glob
import os

import torch

from .. import parallel


def save_checkpoint(path, model, objective=None, optimizer=None, lr_sched=None, epoch=0, loss=None):
    state = {
        "model": model.state_dict(),
        "objective": objective.state_dict() if objective is not None else {},
        "optimizer": optimizer.state_dict() if optimizer is not None else None,
        "lr_sched": lr_sched.state_dict() if lr_sched is not None else None,
        "epoch": epoch,
        "loss": loss,
    }
    torch.save(state, path)
    return path


def load_checkpoint(path, model, objective=None, optimizer=None, lr_sched=None, map_location="cpu"):
    """Mirror of the resume block of train.py:185-225; returns the stored epoch."""
    mdl = torch.load(path, map_location=map_location)
    model.load_state_dict(mdl["model"])
    if objective is not None and mdl.get("objective"):
        objective.load_state_dict(mdl["objective"])
    if optimizer is not None and mdl.get("optimizer") is not None:
        optimizer.load_state_dict(mdl["optimizer"])
    if lr_sched is not None and mdl.get("lr_sched") is not None:
        lr_sched.load_state_dict(mdl["lr_sched"])
    return mdl.get("epoch", 0)


def combine_models(out_path, model_paths, delete=False):
    """combine_models.py: elementwise mean of the job checkpoints' model parameters (integer buffers keep their
    dtype, combine_models.py:77-78); optimizer / scheduler state of the LAST job is kept (quirk 8)."""
    ckpts = [torch.load(p, map_location="cpu") for p in model_paths]
    if not ckpts:
        raise ValueError("no checkpoints to combine")
    out = dict(ckpts[-1])
    out["model"] = parallel.average_parameters([c["model"] for c in ckpts])
    torch.save(out, out_path)
    if delete:
        for p in model_paths:
            os.remove(p)
    return out_path


def epoch_checkpoints(expdir, epoch):
    """The ``{epoch}.{job}.mdl`` files train_parallel.sh:59-77 produces for one epoch."""
    return sorted(glob.glob(os.path.join(expdir, f"{epoch}.*.mdl")))
