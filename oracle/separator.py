"""CPU restatement of the sliding-window Separator, css/executor/separator.py:63-111 (TEST INFRASTRUCTURE ONLY).

Pinned: tests/golden/separator.npz holds the outputs of the reference's own ``Separator.separate`` (run in the build
container with a stub ``onnxruntime`` module and css_amd.utils.synthetic.FakeMaskSession as the session) for a
7-channel recording with the DOA merge on and a 1-channel recording; tests/test_oracle_golden.py checks this
restatement against them.
"""
import math

import numpy as np
import torch

from . import doa, feature


def separate(s, session, sr=16000, eval_win=2.4, eval_hop=0.8, batch_size=1, frame_len=512, frame_hop=256, ipd=None,
             merge=False, merge_threshold=16, lowcut=80, highcut=2000):
    """s [D, n] -> (masks [W, 257, T, 3] clamped to <= 1, mags [W, 257, T]); ``session`` has the onnxruntime
    interface (get_inputs()[0].name, run(None, {name: (1, F, B*T) float32}) -> three (1, 257, B*T) arrays)."""
    win = int(eval_win * sr) + 256
    s = s.unfold(-1, win, int(eval_hop * sr))                                # separator.py:67  D x W x n (tail dropped)
    if merge:
        sv, angles = doa.steervec_7ch(257, 30, samplerate=sr)               # :54-57
        freq_step = (sr // 2) / (257 - 1)
        lo, hi = int(math.floor(lowcut / freq_step)), int(math.ceil(highcut / freq_step))   # :59-61
    masks, mags = [], []
    for b0 in range(0, s.shape[1], batch_size):                              # :72-80
        sb = s[:, b0:b0 + batch_size, :].permute(1, 0, 2)                    # B x D x n
        mag, f, re, im = feature.feature_extractor(sb, frame_len, frame_hop, ipd_index=ipd)   # :91
        B, F, T = f.shape
        D = sb.shape[1]
        res = session.run(None, {session.get_inputs()[0].name: f.reshape(1, F, B * T).numpy()})   # :94-97
        m = torch.Tensor(np.stack((res[0], res[1], res[2]), axis=0))         # :99-101
        m = m.reshape(3, B, F // D, T).permute(1, 2, 3, 0)                   # :102  (B, 257, T, 3)
        m = torch.clamp(m, max=1.0)                                          # :103
        if merge:                                                            # :105-109
            spec = re + 1j * im
            m[:, :, :, :2] = doa.angle_merge(spec, m[:, :, :, :2], sv, angles, lo, hi, thresh=merge_threshold)[0]
        masks.append(m)
        mags.append(mag)
    return torch.cat(masks, 0), torch.cat(mags, 0)
