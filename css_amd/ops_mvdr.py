"""Torch-tensor front door to the MVDR kernels of libcsship.so (css_amd/csrc/mvdr.hip)."""
import ctypes as C

import torch

from ._lib import check, lib
from .ops import _need, _ptr, _stream


def _xcheck(x_re, x_im):
    _need(x_re, torch.float32, "x_re"), _need(x_im, torch.float32, "x_im")
    if x_re.dim() != 4 or x_re.shape != x_im.shape or not (x_re.is_contiguous() and x_im.is_contiguous()):
        raise RuntimeError("css_amd: MVDR spectra must be dense [B, D, T, F] re / im planes")
    return x_re.shape


def _mcheck(mask, B, T, F):
    _need(mask, torch.float32, "mask")
    if mask.dim() != 3 or mask.shape[0] != B or mask.shape[1] < T or mask.shape[2] != F or mask.stride(2) != 1:
        raise RuntimeError(f"css_amd: MVDR mask must be [B, >=T, F], got {tuple(mask.shape)}")


def scm(x_re, x_im, mask, T=None, normalize=True, diag_load=0.0):
    """Masked spatial covariance: (scm complex64 [B,F,D,D], mask_sum f32 [B,F]).  mask: [B, >=T, F]."""
    B, D, Tx, F = _xcheck(x_re, x_im)
    T = Tx if T is None else T
    _mcheck(mask, B, T, F)
    out = torch.empty((B, F, D, D, 2), dtype=torch.float32, device=x_re.device)
    msum = torch.empty((B, F), dtype=torch.float32, device=x_re.device)
    check(lib().csship_mvdr_scm(_ptr(x_re), _ptr(x_im), _ptr(mask), mask.stride(1), mask.stride(0), B, D, T, Tx, F,
                                float(diag_load), int(normalize), _ptr(out), _ptr(msum), _stream()))
    return torch.view_as_complex(out), msum


def scm_multi(x_re, x_im, masks, T=None, normalize=True, diag_load=0.0):
    """scm() for up to four masks of equal shape and strides over the same spectra, ONE launch: list of (scm, mask_sum)."""
    B, D, Tx, F = _xcheck(x_re, x_im)
    T = Tx if T is None else T
    n = len(masks)
    if not 1 <= n <= 4:
        raise RuntimeError("css_amd: scm_multi takes 1..4 masks")
    for m in masks:
        _mcheck(m, B, T, F)
        if m.stride() != masks[0].stride():
            raise RuntimeError("css_amd: scm_multi needs masks with equal strides")
    outs = [torch.empty((B, F, D, D, 2), dtype=torch.float32, device=x_re.device) for _ in range(n)]
    sums = [torch.empty((B, F), dtype=torch.float32, device=x_re.device) for _ in range(n)]
    arr = lambda ts: (C.c_void_p * n)(*[t.data_ptr() for t in ts])  # noqa: E731
    check(lib().csship_mvdr_scm_multi(_ptr(x_re), _ptr(x_im), arr(masks), n, masks[0].stride(1), masks[0].stride(0), B, D, T, Tx,
                                      F, float(diag_load), int(normalize), arr(outs), arr(sums), _stream()))
    return [(torch.view_as_complex(o), s_) for o, s_ in zip(outs, sums)]


def souden(target_scm, noise_scm, noise_scm2=None, ref_mic=0, trace_eps=1e-8):
    """Souden MVDR weights complex64 [B,F,D] from SCMs complex64 [B,F,D,D] (noise = noise_scm [+ noise_scm2])."""
    t = torch.view_as_real(target_scm.contiguous())
    n = torch.view_as_real(noise_scm.contiguous())
    n2 = torch.view_as_real(noise_scm2.contiguous()) if noise_scm2 is not None else None
    B, F, D = target_scm.shape[:3]
    w = torch.empty((B, F, D, 2), dtype=torch.float32, device=t.device)
    check(lib().csship_mvdr_souden(_ptr(t), _ptr(n), _ptr(n2), B * F, D, ref_mic, float(trace_eps), _ptr(w), _stream()))
    return torch.view_as_complex(w)


def apply(x_re, x_im, w, mask=None, T=None):
    """y = w^H x -> (y_re, y_im f32 [B,F,T], energies f32 [B,2] = {sum |y|^2, sum |mask x_0|^2})."""
    B, D, Tx, F = _xcheck(x_re, x_im)
    T = Tx if T is None else T
    if mask is not None:
        _mcheck(mask, B, T, F)
    wr = torch.view_as_real(w.contiguous())
    yr = torch.empty((B, F, T), dtype=torch.float32, device=x_re.device)
    yi = torch.empty_like(yr)
    e = torch.empty((B, 2), dtype=torch.float32, device=x_re.device)
    nws = lib().csship_mvdr_apply_workspace(B, T, F)
    ws = torch.empty(max(nws, 16), dtype=torch.uint8, device=x_re.device)
    check(lib().csship_mvdr_apply(_ptr(x_re), _ptr(x_im), _ptr(wr), _ptr(mask), mask.stride(1) if mask is not None else 0,
                                  mask.stride(0) if mask is not None else 0, B, D, T, Tx, F, _ptr(yr), _ptr(yi), _ptr(e),
                                  _ptr(ws), nws, _stream()))
    return yr, yi, e


def finish(y0, y1, e0, e1, rescale=True, dedup=True):
    """In place on (re, im) pairs y0, y1 [B,F,T]: energy rescale + deduplication gain."""
    B = y0[0].shape[0]
    per_b = y0[0].numel() // B
    check(lib().csship_mvdr_finish(_ptr(y0[0]), _ptr(y0[1]), _ptr(y1[0]), _ptr(y1[1]), _ptr(e0), _ptr(e1), int(rescale),
                                   int(dedup), B, per_b, _stream()))
    return y0, y1
