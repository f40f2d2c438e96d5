"""CPU restatement of the mask-based MVDR beamformer (TEST INFRASTRUCTURE ONLY, like the rest of oracle/).

Two statements of the same math exist in the reference:
  * css/utils/mvdr_util.py:45-66 (numpy; importable, never called by the pipeline): un-normalised masked SCMs with
    1e-15 diagonal loading, W = (N^-1 T / tr(N^-1 T))[:, 0], y = sum_c conj(W_c) x_c.  tests/golden/mvdr.npz pins
    `mask_scm`, `souden_coeffs`, `apply_bf` below against it.
  * css/executor/beamformer.py:126-183, the live path, which delegates the arithmetic to asteroid
    (compute_scm + SoudenMVDRBeamformer, NOT in the reference checkout and not installed): PARITY UNPINNED for
    `compute_scm` / `souden_mvdr` (restated from asteroid's published algorithm: mask-normalised SCMs,
    eps = 1e-8 added to the trace, reference microphone 0); the scalar normalisation cancels in W, so the two agree up
    to the eps terms - tests check that.  Everything around it (STFT with centre padding, energy rescale, the
    deduplication gain, iSTFT, overlap-add with proceed_margin, peak normalisation) is in beamformer.py and restated
    line by line.
"""
import math

import numpy as np
import torch


# ---- mvdr_util.py:45-66 -------------------------------------------------------------------------------------
def mask_scm(mix, mask, diag=1e-15):
    """mix [D,F,T] complex, mask [F,T] -> (Ri [F,D,D], sum(mask))   (get_mask_scm)"""
    x = mix.transpose(1, 2, 0)
    Ri = np.einsum("FT,FTM,FTm->FMm", mask, x, x.conj())
    Ri = Ri + diag * np.eye(mix.shape[0])[np.newaxis]
    return Ri, np.sum(mask)


def souden_coeffs(noi_scm, tgt_scm):
    """W [F,D] = (N^-1 T / trace)[..., 0]; the reference adds 1e-15 to the trace of frequency bin 0 only (calc_bfcoeffs)"""
    num = np.linalg.solve(noi_scm, tgt_scm)
    den = np.trace(num, axis1=-2, axis2=-1)[..., np.newaxis, np.newaxis]
    den[0] += 1e-15
    return (num / den)[..., 0]


def apply_bf(mix, W):
    """y [F,T] = sum_c conj(W[f,c]) mix[c,f,t]   (get_bf)"""
    c, f, t = mix.shape
    return np.sum(W.reshape(f, c, 1).conj() * mix.transpose(1, 0, 2), axis=1)


# ---- beamformer.py:126-183 with asteroid's compute_scm / SoudenMVDRBeamformer restated ------------------------
def compute_scm(S, mask, normalize=True):
    """S [B,D,F,T] complex, mask [B,F,T] -> scm [B,F,D,D] (asteroid returns [B,D,D,F]; same numbers)"""
    scm = torch.einsum("bmft,bnft->bfmn", S * mask[:, None], S.conj())
    if normalize:
        scm = scm / mask.sum(-1)[..., None, None]
    return scm


def souden_mvdr(S, target_scm, noise_scm, ref=0, eps=1e-8):
    """-> [B,F,T] complex: w = (N^-1 T)[:, ref] / (tr(N^-1 T) + eps), y = w^H x"""
    num = torch.linalg.solve(noise_scm, target_scm)
    tr = torch.diagonal(num, dim1=-2, dim2=-1).sum(-1)
    w = num[..., ref] / (tr[..., None] + eps)                    # [B,F,D]
    return torch.einsum("bfm,bmft->bft", w.conj(), S)


def normalize_scale(result_spec, ch0_spec, mask):
    """beamformer.py:165-183"""
    masked = mask * ch0_spec
    masked_energy = torch.sqrt(torch.mean(torch.abs(masked) ** 2, dim=(1, 2), keepdim=True))
    mvdr_energy = torch.sqrt(torch.mean(torch.abs(result_spec) ** 2, dim=(1, 2), keepdim=True))
    return result_spec / mvdr_energy * masked_energy


def stft_center(x, n_fft=512, hop=256):
    return torch.stft(x, n_fft=n_fft, hop_length=hop, center=True, onesided=True, window=torch.hann_window(n_fft),
                      return_complex=True)


def process_one_batch(x, speech_mask, noise_mask, rescale=True, n_fft=512, hop=256):
    """beamformer.py:126-163.  x [B,D,N]; masks [B,F,T] -> [B,F,L] complex"""
    B, D, N = x.shape
    S = stft_center(x.reshape(B * D, N), n_fft, hop)
    _, F, T = S.shape
    S = S.reshape(B, D, F, T)
    L = min(S.shape[-1], speech_mask.shape[-1])
    tgt = compute_scm(S[..., :L], speech_mask[..., :L])
    noi = compute_scm(S[..., :L], noise_mask[..., :L])
    res = souden_mvdr(S[..., :L], tgt, noi)
    if rescale:
        res = normalize_scale(res, S[:, 0, :, :L], speech_mask[..., :L])
    return res


def dedup(S):
    """beamformer.py:67-80.  S [B,2,F,T] complex (modified copy returned)"""
    S = S.clone()
    S_pow = 10 * torch.log10(torch.sum(torch.abs(S) ** 2, dim=(2, 3)))
    S_abs = torch.abs(S)
    gain = torch.divide(S_abs, torch.amax(S_abs, dim=1, keepdim=True))
    for b in range(S.shape[0]):
        if S_pow[b, 0] - S_pow[b, 1] > 15:
            S[b, 1] = torch.multiply(torch.clamp(gain[b, 1], min=10 ** (-40 / 20)), S[b, 1])
        elif S_pow[b, 1] - S_pow[b, 0] > 15:
            S[b, 0] = torch.multiply(torch.clamp(gain[b, 0], min=10 ** (-40 / 20)), S[b, 0])
    return S


def continuous_process(X, masks, cfg, sr=16000, rescale=True):
    """beamformer.py:32-124 line by line (including the batch_end + 1 overlap of consecutive batches, whose
    duplicated window is simply added twice into disjoint... the SAME output span - kept as the reference has it)."""
    n_fft, hop = cfg["n_fft"], cfg["hop_size"]
    eval_win, eval_hop = int(cfg["eval_win"] * sr), int(cfg["eval_hop"] * sr)
    margin = int(cfg["proceed_margin"] * sr)
    bs = cfg["batch_size"]
    mask_hop, mask_win = int(eval_hop / hop), int(eval_win / hop)
    D, T = X.shape
    m0, m1, mn = masks
    Xw = X.unfold(-1, eval_win + 256, eval_hop).permute(1, 0, 2)
    m0 = m0.unfold(1, mask_win, mask_hop).permute(1, 0, 2)
    m1 = m1.unfold(1, mask_win, mask_hop).permute(1, 0, 2)
    mn = mn.unfold(1, mask_win, mask_hop).permute(1, 0, 2)
    n_batches = math.ceil(Xw.shape[0] / bs)
    r0, r1 = torch.zeros(T), torch.zeros(T)
    for i in range(n_batches):
        bstart = i * bs
        bend = min((i + 1) * bs, Xw.shape[0]) + 1
        xb = Xw[bstart:bend]
        o0 = process_one_batch(xb, m0[bstart:bend], mn[bstart:bend], rescale, n_fft, hop)
        o1 = process_one_batch(xb, m1[bstart:bend], mn[bstart:bend], rescale, n_fft, hop)
        S = dedup(torch.stack((o0, o1), dim=1))
        wavs = []
        for k in range(2):
            w = torch.istft(S[:, k], n_fft=n_fft, hop_length=hop, center=True, onesided=True,
                            window=torch.hann_window(n_fft), return_complex=False)
            wavs.append(w[:, : xb.shape[-1]])
        st, en = eval_hop * bstart, eval_hop * bend
        for b in range(len(wavs[0])):
            for res, w in ((r0, wavs[0]), (r1, wavs[1])):
                if st == 0:
                    res[st: st + margin] += w[b, :margin]
                elif en == T:
                    res[st + margin - eval_hop: st + w.shape[1]] += w[b, margin - eval_hop:]
                else:
                    res[st + margin - eval_hop: st + margin] += w[b, margin - eval_hop: margin]
            st += eval_hop
            en += eval_hop
    r0 = r0 * 0.9 / torch.max(torch.abs(r0))
    r1 = r1 * 0.9 / torch.max(torch.abs(r1))
    return r0.unsqueeze(0), r1.unsqueeze(0)
