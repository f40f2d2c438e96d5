#!/usr/bin/env python3
"""Golden vectors of the CAUSAL ConvModule (css/models/conformer.py:265-273, 297-298) from the reference itself.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_causal.py [/root/reference]

The reference's ``Conformer.build_model`` never passes ``causal`` (no recipe reaches it), but the class supports it:
``ConvModule(causal=True)`` pads the depthwise convolution by kernel_size - 1 on both sides and drops the last
kernel_size - 1 outputs, and ``ConformerEncoder(causal=True)`` hands the flag to every block.  Same seeds / inputs as
make_golden.py's conv entries (rnd((2, 64, 256), 11), weights seed 24), plus a 2-block encoder + head forward.
Only outputs are stored (causal.npz)."""
import os
import sys

sys.dont_write_bytecode = True
REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
sys.path.insert(0, REF)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from css.models.conformer import ConformerEncoder, ConvModule  # noqa: E402
from css_amd.utils import synthetic as syn  # noqa: E402

torch.set_num_threads(8)


def rnd(shape, seed, scale=1.0):
    return torch.from_numpy((scale * np.random.RandomState(seed).standard_normal(shape)).astype(np.float32))


def load_synth(model, seed):
    sd = syn.fill_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed)
    model.load_state_dict(sd)
    model.eval()
    return sd


out = {}
with torch.no_grad():
    x = rnd((2, 64, 256), 11)
    conv = ConvModule(256, 33, 0.1, causal=True)
    load_synth(conv, 24)
    out["conv_eval"] = conv(x).numpy()
    conv.train()
    conv.dropout.p = 0.0
    out["conv_trainbn"] = conv(x).numpy()
    # a 2-block causal encoder on MVN'd features (conformer.py:95 + :152-165), eval mode
    enc = ConformerEncoder(257, 256, 4, 1024, 2, 33, 0.1, causal=True, relative_pos_emb=True)
    sd = load_synth(enc, 31)
    f = rnd((2, 90, 257), 12).abs()
    fn = (f - f.mean(-2, keepdim=True)) / (f.std(-2, keepdim=True) + float(np.finfo(np.float32).eps))
    out["encoder2"] = enc(fn, None)[0].numpy()
    out["encoder2_keys"] = np.array(sorted(sd.keys()))
np.savez_compressed(os.path.join(HERE, "causal.npz"), **out)
print({k: getattr(v, "shape", None) for k, v in out.items()})
