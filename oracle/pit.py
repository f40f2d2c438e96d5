"""Oracle: permutation-invariant MSE (reference css/objectives/pit.py, mse.py).

Test infrastructure only -- see oracle/__init__.py.
"""
import itertools

import torch


def pairwise_sse(est, tgt):
    """D[i, j] = sum over (T,F) of (est[i] - tgt[j])^2.  est/tgt [K,T,F] -> [K,K] (float64 acc)."""
    e = est.double()[:, None]
    t = tgt.double()[None, :]
    return ((e - t) ** 2).sum(dim=(-1, -2))


def pit_mse(est, tgt, return_permutation=False):
    """pit_loss(axis=0, loss_fn=mse_loss), pit.py:5-62.

    Candidate p scores mean((est[p] - tgt)^2) over all K*T*F elements: est[p[i]] is
    matched to tgt[i].  Ties resolve to the first permutation in
    itertools.permutations order (torch.min returns the first minimum).
    """
    assert est.shape == tgt.shape, f"{est.shape} != {tgt.shape}"
    K = est.shape[0]
    assert K < 30
    perms = list(itertools.permutations(range(K)))
    cands = torch.stack([torch.mean((est[list(p)] - tgt) ** 2) for p in perms])
    best, idx = torch.min(cands, dim=0)
    if return_permutation:
        return best, perms[int(idx)]
    return best


def mse_objective(y_pred, src1, src2):
    """MeanSquaredError.forward, mse.py:26-37, with y_pred = model(x)[0]."""
    y_true = torch.stack((src1, src2), dim=1)
    assert y_pred.shape == y_true.shape, f"{y_pred.shape} != {y_true.shape}"
    B = y_pred.shape[0]
    total = 0
    for b in range(B):
        total = total + pit_mse(y_pred[b], y_true[b])
    return total / B


def objective_multi(y_pred, targets):
    """Same, for K>=2 targets stacked as [B,K,T,F] (config 5: 3 speakers)."""
    B = y_pred.shape[0]
    total = 0
    perms = []
    for b in range(B):
        l, p = pit_mse(y_pred[b], targets[b], return_permutation=True)
        total = total + l
        perms.append(p)
    return total / B, perms
