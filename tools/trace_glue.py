#!/usr/bin/env python3
"""Which torch (aten) operators does one training step issue besides the library's own launches, and from where?

The training step is replayed as a captured HIP graph, so every aten call in the forward / backward / optimiser path is
a kernel node of that graph (rocprofv3 of round 4: ~1.2 ms of an 11.3 ms step in ~280 torch launches and D2D copies).
A TorchDispatchMode sees each aten call on the host; this tool runs ONE eager training step of the bench workload under
it and prints, per (operator, calling line inside css_amd/), the call count and the bytes of the largest output.

    python tools/trace_glue.py [--blocks 16] [--idim 257]
"""
import argparse
import collections
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402

SKIP = ("aten.view", "aten._unsafe_view", "aten.detach", "aten.as_strided", "aten.t.", "aten.transpose", "aten.permute",
        "aten.select", "aten.slice", "aten.unsqueeze", "aten.squeeze", "aten.expand", "aten.alias", "aten.empty",
        "aten.reshape", "aten.unbind", "aten.split", "aten._reshape_alias", "aten.empty_strided", "aten.empty_like",
        "aten.lift_fresh", "aten.is_pinned", "aten.unfold")


class Trace(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.rows = collections.Counter()
        self.bytes = {}

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if not name.startswith(SKIP):
            where = "?"
            for fr in reversed(traceback.extract_stack()):
                if "/css_amd/" in fr.filename or fr.filename.endswith("bench.py"):
                    where = f"{os.path.relpath(fr.filename, ROOT)}:{fr.lineno} {fr.name}"
                    break
            nb = 0
            for o in (out if isinstance(out, (tuple, list)) else (out,)):
                if isinstance(o, torch.Tensor):
                    nb = max(nb, o.numel() * o.element_size())
            if not nb:   # in-place / foreach: size of the first tensor argument
                for a in args:
                    a0 = a[0] if isinstance(a, (list, tuple)) and a and isinstance(a[0], torch.Tensor) else a
                    if isinstance(a0, torch.Tensor):
                        nb = a0.numel() * a0.element_size() * (len(a) if isinstance(a, (list, tuple)) else 1)
                        break
            key = (name, where)
            self.rows[key] += 1
            self.bytes[key] = max(self.bytes.get(key, 0), nb)
        return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--blocks", type=int, default=16)
    ap.add_argument("--B", type=int, default=32)
    ap.add_argument("--T", type=int, default=400)
    args = ap.parse_args()
    from css_amd import ops
    from css_amd.models import MODELS
    from css_amd.objectives import OBJECTIVES
    from css_amd.optim import FusedAdam
    from css_amd.utils import synthetic as syn
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    conf = dict(attention_dim=256, attention_heads=4, linear_units=1024, num_blocks=args.blocks, kernel_size=33,
                dropout_rate=0.1, relative_pos_emb=True)
    model = MODELS["Conformer"](257, 257, 2, 1, conf)
    model.load_state_dict(syn.fill_state_dict(syn.conformer_shapes(num_blocks=args.blocks), 0))
    model = model.to(dev).train()
    mix, src = syn.synthetic_sources(args.B, 512 + 256 * (args.T - 1), seed=0)
    audio = torch.stack([mix, src[0], src[1]], 0).to(dev)
    obj = OBJECTIVES["MSE"].build_objective({})
    optim = FusedAdam(model.parameters(), lr=1e-4, max_norm=5.0)

    def step():
        with torch.no_grad():
            mags = ops.stft(audio, ("mag",))["mag"]
        batch = {"mix": mags[0], "source1": mags[1], "source2": mags[2]}
        optim.zero_grad()
        loss = obj(model, batch, device=dev)
        loss.backward()
        optim.step(max_norm=5.0)
        return loss

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    tr = Trace()
    with tr:
        step()
    torch.cuda.synchronize()
    tot = sum(tr.rows.values())
    print(f"{tot} aten calls in one training step ({args.blocks} blocks); per (operator, caller): calls, largest tensor")
    for (name, where), n in sorted(tr.rows.items(), key=lambda kv: -kv[1] * max(tr.bytes[kv[0]], 1 << 16)):
        print(f"{n:5d}  {tr.bytes[(name, where)] / 1e6:9.3f} MB  {name:42s} {where}")


if __name__ == "__main__":
    main()
