"""CPU restatement of the DOA-based mask merge of css/executor/separator.py:113-250 (TEST INFRASTRUCTURE ONLY).

Pinned: css.executor.separator imports onnxruntime at its top (css/utils/model_util.py:1), which is not installed;
with a stub module of that name registered in sys.modules the reference's own Separator imports and runs in the
build container (tests/golden/make_golden_separator.py).  tests/golden/separator.npz holds ITS steering table,
``doa_likelihood`` (softmax off and on) and ``angle_merge`` outputs on css_amd.utils.synthetic.doa_inputs, and
tests/test_oracle_golden.py checks the functions below against them.  SURVEY quirk 6 is part of what is pinned:
``masked_magnitude[b]`` (separator.py:200) indexes the FREQUENCY axis with the batch index, the sum collapses to a
scalar and ``argmin`` of a scalar is 0 - the reference always silences output 0 when two directions coincide.
``fix_indexing=True`` gives the evidently intended behaviour (silence the output with less masked energy); that
variant has no reference counterpart and is build-defined.
"""
import math

import torch


def steervec_7ch(nfreqs, nvecs, radius=0.0425, sndvelocity=340, samplerate=16000, reference=0, inverse_shift=False):
    """separator.py:113-163 -> (steervecs complex64 [nfreqs, nvecs, 7], angles in degrees [nvecs])"""
    angles = torch.tensor([2 * math.pi * k / nvecs for k in range(nvecs)])
    distances = radius * torch.stack((torch.zeros(nvecs), torch.cos(angles + math.pi / 6), torch.cos(angles - math.pi / 6),
                                      torch.cos(angles - math.pi / 2), torch.cos(angles - 5 * math.pi / 6),
                                      torch.cos(angles + 5 * math.pi / 6), torch.cos(angles + math.pi / 2)), dim=1)
    if reference != 0:
        distances -= distances[:, reference, None]
    deltas = distances / sndvelocity * samplerate
    sign = -1j if inverse_shift else 1j
    sv = torch.stack([torch.exp(sign * deltas * math.pi * f / (nfreqs - 1)) for f in range(nfreqs)])
    sv = sv / math.sqrt(7)
    return sv, angles * 360 / (2 * math.pi)


def doa_likelihood(X, steervec, mask, compression=1 / 3, epsilon=1e-12, lowcut=0, highcut=None, softmax=True):
    """separator.py:206-250.  X [b,7,F,T] complex, steervec [F,A,7], mask [b,F,T,C] -> [b,C,A]"""
    Xb = X[:, :, lowcut:highcut, :]
    Xpow = torch.abs(torch.einsum("bmft,bmft->bft", Xb, Xb.conj()))
    XHpow = torch.abs(torch.einsum("bmft,fam->bfta", Xb.conj(), steervec[lowcut:highcut])) ** 2
    if compression <= 0:
        tf = -torch.log(Xpow[..., None] - XHpow / (1 + epsilon))
    else:
        tf = -torch.pow(Xpow[..., None] - XHpow / (1 + epsilon), compression)
    ret = torch.einsum("bftc,bfta->bca", mask[:, lowcut:highcut], tf)
    if softmax:
        ret = ret - torch.amax(ret, dim=-1, keepdim=True)
        ret = torch.exp(ret - torch.log(torch.sum(torch.exp(ret), dim=-1, keepdim=True)))
    return ret


def angle_merge(X, mask, steervec, angles, lo, hi, thresh=16, mask_binarize_thres=0.5, compression=0.5,
                fix_indexing=False):
    """separator.py:165-204.  mask [b,F,T,2] -> new mask (one output silenced to 1e-12 where the DOAs coincide)"""
    batch, nfreqs, nframes, _ = mask.shape
    binmask = (mask > mask_binarize_thres).float()
    like = doa_likelihood(X, steervec, binmask, compression=compression, epsilon=1e-12, lowcut=lo, highcut=hi,
                          softmax=False)
    idx = torch.argmax(like, dim=-1)
    new_mask = torch.clone(mask)
    for b in range(batch):
        DOA = angles[idx[b]]
        if min(float((DOA[0] - DOA[1]) % 360), float((DOA[1] - DOA[0]) % 360)) <= thresh:
            masked_magnitude = binmask[b] * torch.abs(X[b, 0:1, :, :].permute(1, 2, 0))     # [F,T,2]
            if fix_indexing:
                masked_energy = torch.sum(masked_magnitude, dim=(0, 1))
            else:
                masked_energy = torch.sum(masked_magnitude[b], dim=(0, 1))                    # quirk 6: a scalar
            kill = int(torch.argmin(masked_energy))
            new_mask[b, ..., kill] = torch.ones((nfreqs, nframes)) * 1e-12
    return new_mask, like
