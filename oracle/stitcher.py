"""Oracle: mask stitcher (reference css/executor/stitcher.py).  Test infrastructure only."""
import torch


def stitch_margin(eval_win, eval_hop, hop_size, sr=16000):
    """stitcher.py:10-12 (kept in the reference's x10 / 10 arithmetic)."""
    return int((eval_win * 10 - eval_hop * 10) / 10 * sr / hop_size)


def get_stitch(mags, masks, margin):
    """Stitcher.get_stitch, stitcher.py:14-52.

    mags: [W,F,T]; masks: [W,F,T,3] (spk1, spk2, noise).  For each window
    boundary choose identity [0,1] or swap [1,0] by comparing
    -sum |E_prev[j,:, -margin:] - E_now[i,:, :margin]|^0.5 on the overlap.
    """
    perms = []
    for n in range(len(masks) - 1):
        e_prev = masks[n][:, :, :2].permute(2, 0, 1) * mags[n].abs()
        e_now = masks[n + 1][:, :, :2].permute(2, 0, 1) * mags[n + 1].abs()
        sim = torch.zeros(2, 2)
        for i in range(2):
            for j in range(2):
                d = e_prev[j, :, -margin:] - e_now[i, :, :margin]
                sim[i, j] = -d.abs().sqrt().sum()
        keep = sim[0, 0] + sim[1, 1]
        swap = sim[0, 1] + sim[1, 0]
        perms.append([0, 1] if keep >= swap else [1, 0])
    return perms


def get_connect(perms, masks, hop_frames, wta_floor=1e-4):
    """Stitcher.get_connect, stitcher.py:54-110.

    Running parity of swaps -> reorder speakers -> winner-take-all (non-max
    entries become 1e-4) -> overlap-add average.  Returns 3 x [F, hop*(W-1)+win].
    """
    W = len(masks)
    state = 0
    parity = [0]
    for p in perms:
        if p[0] == 1:
            state = 1 - state
        parity.append(state)
    F, win = masks[0].shape[0], masks[0].shape[1]
    total = hop_frames * (W - 1) + win
    acc = torch.zeros(3, F, total)
    cnt = torch.zeros(total)
    for w in range(W):
        m = masks[w]
        a, b = (m[:, :, 0], m[:, :, 1]) if parity[w] == 0 else (m[:, :, 1], m[:, :, 0])
        trio = torch.stack((a, b, m[:, :, 2]), dim=0)
        top = trio.amax(dim=0, keepdim=True)
        trio = torch.where(trio == top, trio, torch.full_like(trio, wta_floor))
        st = hop_frames * w
        acc[:, :, st:st + win] += trio
        cnt[st:st + win] += 1
    cnt[cnt == 0] = 1
    out = acc / cnt
    return out[0], out[1], out[2]
