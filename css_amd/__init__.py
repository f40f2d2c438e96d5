"""css_amd -- MI355X-native (gfx950) hot path for continuous speech separation.

Drop-in for the reference's ``css.models`` / ``css.objectives`` / ``css.executor`` /
``css.trainer`` plugin API (desh2608/css).  All hot-path arithmetic runs in the
hand-written HIP library ``css_amd/csrc/libcsship.so`` (C ABI declared in
``include/csship.h``); PyTorch-ROCm provides device memory, streams and
``torch.distributed`` (RCCL).  There is no CPU fallback: calling an op without the
library or without a GPU raises.
"""
__version__ = "0.1.0"
