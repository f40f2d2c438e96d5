"""Mask stitcher -- drop-in for css.executor.stitcher.Stitcher (stitcher.py:4-110).

Same constructor (stitching_config, sr) and methods.  ``get_stitch`` returns the reference's Python list of
[0,1] / [1,0] decisions (that costs one small device->host copy of the (W-1) x 4 similarity sums);
``get_connect`` returns the three stitched masks [F, hop*(W-1)+T] on the device.
"""
import torch

from .. import ops
from .._lib import check, lib


class Stitcher:
    def __init__(self, stitching_config, sr=16000):
        self.eval_win = stitching_config["eval_win"]
        self.eval_hop = stitching_config["eval_hop"]
        self.fft_hop = stitching_config["hop_size"]
        self.sr = sr
        # stitcher.py:10-12 (kept in the reference's x10 / 10 arithmetic)
        self.stitch_margin = int((self.eval_win * 10 - self.eval_hop * 10) / 10 * self.sr / self.fft_hop)

    def get_stitch(self, x, masks):
        """x: [W,F,T] magnitudes, masks: [W,F,T,3] -> list of [0,1] | [1,0] per window boundary."""
        ops._need(x, torch.float32, "x")
        ops._need(masks, torch.float32, "masks")
        x, masks = x.contiguous(), masks.contiguous()
        W, F, T = x.shape
        assert tuple(masks.shape) == (W, F, T, 3)
        if W < 2:
            return []
        sim = torch.empty(W - 1, 4, device=x.device, dtype=torch.float32)
        check(lib().csship_stitch_similarity(ops._ptr(x), ops._ptr(masks), W, F, T, min(self.stitch_margin, T),
                                             ops._ptr(sim), ops._stream()))
        s = sim.cpu()
        keep = (s[:, 0] + s[:, 3]) >= (s[:, 1] + s[:, 2])  # sim0 >= sim1 (stitcher.py:42-49)
        return [[0, 1] if k else [1, 0] for k in keep.tolist()]

    def get_connect(self, PERM, mask, wta_floor=1e-4):
        ops._need(mask, torch.float32, "mask")
        mask = mask.contiguous()
        W, F, T, _ = mask.shape
        state, cur = [0], 0
        for item in PERM:  # running parity (stitcher.py:55-60)
            if item[0] == 1:
                cur = 1 - cur
            state.append(cur)
        assert len(state) == W
        st = torch.tensor(state, dtype=torch.int32, device=mask.device)
        hop = int(self.eval_hop * self.sr / self.fft_hop)
        out = torch.empty(3, F, hop * (W - 1) + T, device=mask.device, dtype=torch.float32)
        check(lib().csship_stitch_connect(ops._ptr(mask), ops._ptr(st), W, F, T, hop, wta_floor, ops._ptr(out),
                                          ops._stream()))
        return out[0], out[1], out[2]
