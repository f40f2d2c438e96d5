"""css_amd -- MI355X-native (gfx950) hot path for continuous speech separation.

Drop-in for the reference's ``css.models`` / ``css.objectives`` / ``css.executor`` /
``css.trainer`` plugin API (desh2608/css).  All hot-path arithmetic runs in the
hand-written HIP library ``css_amd/csrc/libcsship.so`` (C ABI declared in
``include/csship.h``); PyTorch-ROCm provides device memory, streams and
``torch.distributed`` (RCCL).  There is no CPU fallback: calling an op without the
library or without a GPU raises.
"""
__version__ = "0.1.0"


# Bumped whenever parameter STORAGE is rewritten behind autograd's back (css_amd.optim.FusedAdam updates the flat
# parameter buffer through a raw HIP kernel / a graph replay, so neither data_ptr() nor Tensor._version changes).
# The bf16 weight-shadow caches of the models (Conformer._pack, BLSTM._pack) key on it.
_param_generation = 0


def bump_param_generation():
    global _param_generation
    _param_generation += 1
    return _param_generation


def param_generation():
    return _param_generation
