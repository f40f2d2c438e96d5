"""Dropout sites of the training forward (the reference's nn.Dropout layers: css/models/conformer.py:127,249,254,302,
317,319 and css/models/blstm.py:147,158) and their stream numbering.

A mask is a pure function of (seed, stream, element index) (include/csship.h, "dropout"): the forward kernels, the
backward kernels and - in the tests - the oracle (oracle/dropout.py) all regenerate it from those three numbers.  The
seed is ONE uint32 in device memory owned by the model; the training forward adds 1 to it on the device before it runs
(also inside a captured step, so every replay draws fresh masks), the backward of the same step reads the same value.
"""
import ctypes as C

import torch

from . import _lib

SITES = {"embed": 0, "ffn_in.hidden": 1, "ffn_in.out": 2, "attn.p": 3, "attn.out": 4, "conv.out": 5,
         "ffn_out.hidden": 6, "ffn_out.out": 7, "blstm.layer": 8}


def stream_of(site):
    """site: "embed" or "<layer index>.<site name>" (e.g. "3.ffn_in.hidden") -> stream id."""
    if site == "embed":
        return SITES["embed"]
    layer, name = site.split(".", 1)
    return (int(layer) + 1) * 16 + SITES[name]


def _rank_offset():
    """Every rank of a data-parallel job drops its own shard independently: its stream ids are shifted by 4096 * rank
    (block ids stay far below 4096)."""
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        return 4096 * torch.distributed.get_rank()
    return 0


class DropState:
    """What a training forward / backward pair needs: the seed tensor, p, and cfg structs per site."""

    def __init__(self, seed, p, offset=None):
        self.seed, self.p = seed, float(p)
        self.offset = _rank_offset() if offset is None else offset
        self._keep = []

    @property
    def active(self):
        return self.p > 0.0

    def cfg(self, site):
        """ctypes pointer to a csship_dropout_cfg for ``site`` (None when dropout is off)."""
        if not self.active:
            return None
        c = _lib.DropoutCfg(self.seed.data_ptr(), stream_of(site) + self.offset, self.p)
        self._keep.append(c)  # the struct must outlive the (synchronous) ctypes call only; kept for the step anyway
        return C.pointer(c)


def model_drop_state(model, p):
    """The model's DropState for THIS training forward: bumps the device-side seed (a tiny kernel on the current stream,
    captured with the step) and returns the state.  p <= 0 or eval mode: inactive state."""
    if p <= 0.0 or not model.training:
        return DropState(None, 0.0)
    seed = model._drop_seed
    seed.add_(1)
    return DropState(seed, p)


def set_seed(model, value):
    """Set the model's dropout seed (the next training forward uses value + 1)."""
    model._drop_seed.fill_(int(value))
