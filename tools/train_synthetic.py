#!/usr/bin/env python3
"""Training loop on synthetic mixtures, mirroring scripts/python/train.py:279-317 of the reference with css_amd parts:
on-GPU synthetic dataset -> Conformer -> PIT-MSE -> explicit HIP backward -> fused clip + Adam, the whole step replayed
as one HIP graph; LR schedule, validation and checkpoints in the reference's format.

    python tools/train_synthetic.py [--epochs 2] [--batches-per-epoch 20] [--blocks 16] [--out /tmp/css_ckpt]
"""
import argparse
import os
import sys
import time
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from css_amd.datasets import DATASETS  # noqa: E402
from css_amd.models import MODELS  # noqa: E402
from css_amd.objectives import OBJECTIVES  # noqa: E402
from css_amd.optim import FusedAdam  # noqa: E402
from css_amd.trainer import GraphedTrainStep, LRScheduler, validate  # noqa: E402
from css_amd.utils import checkpoint as ck  # noqa: E402

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=2)
    ap.add_argument("--batches-per-epoch", type=int, default=20)
    ap.add_argument("--blocks", type=int, default=16)
    ap.add_argument("--batch-size", type=int, default=32)
    ap.add_argument("--frames", type=int, default=400)
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--grad-thresh", type=float, default=5.0)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    dev = "cuda"
    conf = dict(attention_dim=256, attention_heads=4, linear_units=1024, num_blocks=a.blocks, kernel_size=33,
                dropout_rate=0.1, relative_pos_emb=True)
    model = MODELS["Conformer"](257, 257, 2, 1, conf).to(dev).train()
    objective = OBJECTIVES["MSE"].build_objective({})
    optim = FusedAdam(model.parameters(), lr=a.lr, max_norm=a.grad_thresh)
    sched = LRScheduler(optim, types.SimpleNamespace(warmup=a.batches_per_epoch, decay=0.0, fixed=0, min_lr=1e-9))
    train = iter(DATASETS["SyntheticSeparation"](batch_size=a.batch_size, frames=a.frames, seed=0, device=dev))
    dev_set = iter(DATASETS["SyntheticSeparation"](batch_size=a.batch_size, frames=a.frames, seed=1000, device=dev))
    step = GraphedTrainStep(model, objective, optim, next(train), max_norm=a.grad_thresh)
    for epoch in range(1, a.epochs + 1):
        t0 = time.perf_counter()
        tot = 0.0
        for _ in range(a.batches_per_epoch):
            loss, gnorm = step(next(train))
            sched.step(1.0)
            tot += float(loss)  # one host sync per step, like the reference's loss.item() logging
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        val = float(validate(dev_set, model, objective, device=dev, num_batches=5))
        print(f"epoch {epoch}: train {tot / a.batches_per_epoch:.5f}  dev {val:.5f}  lr {sched.curr_lr:.2e}  "
              f"{a.batches_per_epoch * a.batch_size * a.frames / dt / 1e6:.3f} M frames/s")
        if a.out:
            os.makedirs(a.out, exist_ok=True)
            ck.save_checkpoint(os.path.join(a.out, f"{epoch}.mdl"), model, objective, optim, sched, epoch)
