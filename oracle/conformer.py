"""Oracle: Conformer mask estimator forward (reference css/models/conformer.py).

Functional restatement over a plain ``state_dict`` (name -> fp32 tensor) whose
keys are the reference's (``conformer.encoders.0.self_attn.linear_q.weight``
...).  Test infrastructure only -- see oracle/__init__.py.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

EPS = float(np.finfo(np.float32).eps)  # conformer.py:21
LN_EPS = 1e-5  # torch.nn.LayerNorm / BatchNorm1d default eps


def _ln(x, sd, p):
    return F.layer_norm(x, (x.shape[-1],), sd[p + ".weight"], sd[p + ".bias"], LN_EPS)


def _lin(x, sd, p):
    return F.linear(x, sd[p + ".weight"], sd[p + ".bias"])


# every matrix product of the attention goes through this name, so that oracle/emulate.py can evaluate the SAME graph
# with the operand roundings of the HIP path (bf16 operands, f32 accumulation) - plain torch.matmul otherwise
_mm = torch.matmul


def mvn_time(f):
    """conformer.py:95 / blstm.py:80: MVN over the time axis (-2), unbiased std, +eps."""
    return (f - f.mean(-2, keepdim=True)) / (f.std(-2, keepdim=True) + EPS)


def rel_pos_table(pe_k, T, maxlen=1000):
    """conformer.py:156-159 + 179-185: pos_k[i, j] = pe_k[clamp(i-j, -maxlen, maxlen-1)+maxlen]."""
    idx = torch.arange(T)
    rel = (idx[:, None] - idx[None, :]).clamp(-maxlen, maxlen - 1) + maxlen
    return pe_k[rel]  # [T,T,d_k]


def _no_drop(x, site):
    return x


def feed_forward(x, sd, p, drop=_no_drop, site=""):
    """FeedForward.forward, conformer.py:322-326.  drop(x, site) applies the two nn.Dropout layers (:317,319) with the
    build's masks (oracle/dropout.py); the default is eval mode (identity)."""
    h = _ln(x, sd, p + ".layer_norm")
    h = drop(torch.relu(_lin(h, sd, p + ".net.0")), site + ".hidden")
    return drop(_lin(h, sd, p + ".net.3"), site + ".out")


def attention(x, sd, p, n_head, pos_k, drop=_no_drop, site=""):
    """MultiHeadedAttention.forward, conformer.py:213-254 (mask=None).  drop: the dropouts of the probabilities (:249,
    indexed as [B, H, T, pad8(T)] like the build's probability buffer) and of the output (:254)."""
    B, T, D = x.shape
    dk = D // n_head
    h = _ln(x, sd, p + ".layer_norm")
    q = _lin(h, sd, p + ".linear_q").view(B, T, n_head, dk).transpose(1, 2)  # [B,H,T,dk]
    k = _lin(h, sd, p + ".linear_k").view(B, T, n_head, dk).transpose(1, 2)
    v = _lin(h, sd, p + ".linear_v").view(B, T, n_head, dk).transpose(1, 2)
    scores = _mm(q, k.transpose(-1, -2))
    if pos_k is not None:
        # B_[b,h,i,j] = q[b,h,i,:] . pos_k[i,j,:], as the reference forms it (conformer.py:230-233): one batched product
        # per query position, (T, B*h, d_k) x (T, d_k, T)
        qt = q.permute(2, 0, 1, 3).reshape(T, B * n_head, dk)
        scores = scores + _mm(qt, pos_k.transpose(1, 2)).reshape(T, B, n_head, T).permute(1, 2, 0, 3)
    att = torch.softmax(scores / math.sqrt(dk), dim=-1)
    if drop is not _no_drop:
        Tp = (T + 7) // 8 * 8
        att = drop(F.pad(att, (0, Tp - T)), site + ".p")[..., :T]
    ctx = _mm(att, v).transpose(1, 2).reshape(B, T, D)
    return drop(_lin(ctx, sd, p + ".linear_out"), site + ".out")


def conv_module(x, sd, p, kernel_size, training_bn=False, drop=_no_drop, site="", causal=False):
    """ConvModule.forward, conformer.py:289-303.  causal (conformer.py:265-273, 297-298): the depthwise convolution is
    padded by kernel_size - 1 on both sides and the last kernel_size - 1 outputs are dropped - tap k of frame t reads
    frame t + k - (kernel_size - 1).

    The two "pointwise convs" are Conv2d(1,2,1) / Conv2d(1,1,1): scalar gains.
    training_bn=True uses batch statistics (biased variance) as BatchNorm1d does
    in train mode; running stats are not updated here.
    """
    h = _ln(x, sd, p + ".layer_norm")
    w1 = sd[p + ".pw_conv_1.weight"].reshape(2)
    b1 = sd[p + ".pw_conv_1.bias"].reshape(2)
    g = (w1[0] * h + b1[0]) * torch.sigmoid(w1[1] * h + b1[1])  # scalar GLU
    g = g.transpose(1, 2)  # [B,D,T]
    D = g.shape[1]
    y = F.conv1d(g, sd[p + ".dw_conv_1d.weight"], sd[p + ".dw_conv_1d.bias"],
                 padding=(kernel_size - 1) if causal else (kernel_size - 1) // 2, groups=D)
    if causal:
        y = y[:, :, : -(kernel_size - 1)]
    if training_bn:
        mu = y.mean(dim=(0, 2), keepdim=True)
        var = y.var(dim=(0, 2), unbiased=False, keepdim=True)
    else:
        mu = sd[p + ".BN.running_mean"].view(1, D, 1)
        var = sd[p + ".BN.running_var"].view(1, D, 1)
    y = (y - mu) / torch.sqrt(var + LN_EPS)
    y = y * sd[p + ".BN.weight"].view(1, D, 1) + sd[p + ".BN.bias"].view(1, D, 1)
    y = torch.relu(y).transpose(1, 2)
    return drop(y * sd[p + ".pw_conv_2.weight"].reshape(()) + sd[p + ".pw_conv_2.bias"].reshape(()), site + ".out")  # :302


def encoder_layer(x, sd, p, n_head, kernel_size, pos_k, training_bn=False, drop=_no_drop, layer=0, causal=False):
    """EncoderLayer.forward, conformer.py:348-362."""
    x = x + 0.5 * feed_forward(x, sd, p + ".feed_forward_in", drop, f"{layer}.ffn_in")
    x = x + attention(x, sd, p + ".self_attn", n_head, pos_k, drop, f"{layer}.attn")
    x = x + conv_module(x, sd, p + ".conv", kernel_size, training_bn, drop, f"{layer}.conv", causal)
    x = x + 0.5 * feed_forward(x, sd, p + ".feed_forward_out", drop, f"{layer}.ffn_out")
    return _ln(x, sd, p + ".layer_norm")


def embed(x, sd, p="conformer.embed", drop=_no_drop):
    """conformer.py:124-129: Linear + LayerNorm + Dropout + ReLU."""
    return torch.relu(drop(_ln(_lin(x, sd, p + ".0"), sd, p + ".1"), "embed"))


def encoder(x, sd, n_head=4, kernel_size=33, num_blocks=16, relative_pos_emb=True,
            training_bn=False, p="conformer", drop=_no_drop, causal=False):
    """ConformerEncoder.forward, conformer.py:152-165."""
    x = embed(x, sd, p + ".embed", drop)
    pos_k = rel_pos_table(sd[p + ".pos_emb.pe_k.weight"], x.shape[1]) if relative_pos_emb else None
    for l in range(num_blocks):
        x = encoder_layer(x, sd, f"{p}.encoders.{l}", n_head, kernel_size, pos_k, training_bn, drop, l, causal)
    return x


def forward(f, sd, num_spk=2, num_noise=1, n_head=4, kernel_size=33, num_blocks=16,
            relative_pos_emb=True, training_bn=False, drop=_no_drop, causal=False):
    """Conformer.forward, conformer.py:83-104 -> (y_pred [N,K,T,F], masks (K+1)x[N,T,F])."""
    if f.dim() == 4:
        f = f.squeeze(0)
    f_orig = f
    h = encoder(mvn_time(f), sd, n_head, kernel_size, num_blocks, relative_pos_emb, training_bn, drop=drop, causal=causal)
    m = torch.relu(_lin(h, sd, "linear"))
    masks = torch.chunk(m, num_spk + num_noise, dim=2)
    y_pred = torch.stack([mk * f_orig for mk in masks[:-1]], dim=1)
    return y_pred, masks
