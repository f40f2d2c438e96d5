#!/usr/bin/env python3
"""End-to-end inference on a synthetic 7-channel session, mirroring scripts/python/separate_libricss.py:55-82 of the
reference with the css_amd executors: Separator (STFT + IPD features -> Conformer masks -> DOA merge) -> Stitcher ->
mask-based MVDR Beamformer.  The config dict has the keys of conf/config_7ch.yaml.

    python tools/separate_synthetic.py [--seconds 12] [--blocks 2]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from css_amd.executor.beamformer import Beamformer  # noqa: E402
from css_amd.executor.separator import Separator  # noqa: E402
from css_amd.executor.stitcher import Stitcher  # noqa: E402
from css_amd.models import MODELS  # noqa: E402
from css_amd.utils import synthetic as syn  # noqa: E402

IPD = "1,0;2,0;3,0;4,0;5,0;6,0"


def config():
    return {
        "sampling_rate": 16000,
        "separation": dict(batch_size=16, eval_hop=0.8, eval_win=2.4, frame_length=512, frame_shift=256, ipd=IPD, merge=True,
                           merge_threshold=16),
        "stitching": dict(eval_hop=0.8, eval_win=2.4, hop_size=256, n_fft=512, connector="mask", stitcher="mask"),
        "beamforming": dict(batch_size=32, type="SoudenMVDRBeamformer", hop_size=256, n_fft=512, eval_hop=0.8,
                            proceed_margin=2, eval_win=2.4, wta_thresh=0.0001),
    }


def session(seconds, sr=16000, seed=0):
    """two delayed noise-like talkers on a 7-microphone array, D x T; T = a whole number of 0.8 s hops + window + 256"""
    n_win = max(2, int((seconds - 2.4) / 0.8) + 1)
    T = int(0.8 * sr) * (n_win - 1) + int(2.4 * sr) + 256
    _, src = syn.synthetic_sources(1, T, seed=seed)                       # [2, 1, T] delayed talkers
    return syn.synthetic_multichannel(src, n_ch=7, seed=seed)[0]           # [7, T]


def executors(model, cfg):
    return (Separator(cfg["separation"], sr=cfg["sampling_rate"], model=model), Stitcher(cfg["stitching"]),
            Beamformer(cfg["beamforming"]))


def run(mixed, ex):
    separator, stitcher, beamformer = ex
    window_masks, mag_specs = separator.separate(mixed)
    perm = stitcher.get_stitch(mag_specs, window_masks)
    stitched = stitcher.get_connect(perm, window_masks)
    return beamformer.continuous_process(mixed, stitched), window_masks.shape[0]


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=12.0)
    ap.add_argument("--blocks", type=int, default=2)
    a = ap.parse_args()
    conf = dict(attention_dim=256, attention_heads=4, linear_units=1024, num_blocks=a.blocks, kernel_size=33,
                dropout_rate=0.1, relative_pos_emb=True)
    model = MODELS["Conformer"](257 * 7, 257, 2, 1, conf)
    model.load_state_dict(syn.fill_state_dict(syn.conformer_shapes(idim=257 * 7, num_blocks=a.blocks), 0))
    model = model.cuda().eval()
    mixed = session(a.seconds).cuda()
    with torch.no_grad():
        ex = executors(model, config())   # built once per process, like the reference's script (separate_libricss.py:55-63)
        run(mixed, ex)  # warm-up (the beamformer captures its per-batch launch sequence here)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        (w0, w1), n_win = run(mixed, ex)
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{mixed.shape[1] / 16000:.1f} s of 7-channel audio, {n_win} windows -> 2 x {w0.shape[1]} samples in {dt * 1e3:.1f} ms "
          f"({mixed.shape[1] / 16000 / dt:.0f}x real time), peaks {float(w0.abs().max()):.2f} / {float(w1.abs().max()):.2f}")
