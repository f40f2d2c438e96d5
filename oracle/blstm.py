"""Oracle: BLSTM mask estimator forward (reference css/models/blstm.py).

The LSTM recurrence is written out explicitly (gate order i,f,g,o as in
torch.nn.LSTM) instead of calling torch.nn.LSTM.  Test infrastructure only.
"""
import torch
import torch.nn.functional as F

from .conformer import mvn_time, LN_EPS


def lstm_direction(x, w_ih, w_hh, b_ih, b_hh, reverse=False):
    """One direction of nn.LSTM(batch_first=True), zero initial state.  x [B,T,I] -> [B,T,H]."""
    B, T, _ = x.shape
    H = w_hh.shape[1]
    gates_x = F.linear(x, w_ih, b_ih + b_hh)  # [B,T,4H]
    h = x.new_zeros(B, H)
    c = x.new_zeros(B, H)
    out = x.new_zeros(B, T, H)
    steps = range(T - 1, -1, -1) if reverse else range(T)
    for t in steps:
        g = gates_x[:, t] + F.linear(h, w_hh)
        i, f, gg, o = g.chunk(4, dim=1)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        h = torch.sigmoid(o) * torch.tanh(c)
        out[:, t] = h
    return out


def _no_drop(x, site):
    return x


def blstm_layer(x, sd, p, drop=_no_drop, layer=0):
    """BLSTMLayer.forward, blstm.py:149-160; drop(x, site): the layer's dropout (:158) with the build's masks."""
    fw = lstm_direction(x, sd[p + ".lstm.weight_ih_l0"], sd[p + ".lstm.weight_hh_l0"],
                        sd[p + ".lstm.bias_ih_l0"], sd[p + ".lstm.bias_hh_l0"])
    bw = lstm_direction(x, sd[p + ".lstm.weight_ih_l0_reverse"], sd[p + ".lstm.weight_hh_l0_reverse"],
                        sd[p + ".lstm.bias_ih_l0_reverse"], sd[p + ".lstm.bias_hh_l0_reverse"],
                        reverse=True)
    y = torch.cat([fw, bw], dim=-1)
    return drop(F.layer_norm(y, (y.shape[-1],), sd[p + ".layer_norm.weight"], sd[p + ".layer_norm.bias"], LN_EPS),
                f"{layer}.blstm.layer")


def encoder(x, sd, num_layers, p="blstm", drop=_no_drop):
    """BLSTMEncoder.forward, blstm.py:122-128 (embed: Linear, LayerNorm, Dropout, ReLU, :143-148)."""
    x = F.linear(x, sd[p + ".embed.0.weight"], sd[p + ".embed.0.bias"])
    x = torch.relu(drop(F.layer_norm(x, (x.shape[-1],), sd[p + ".embed.1.weight"], sd[p + ".embed.1.bias"], LN_EPS), "embed"))
    for l in range(num_layers):
        x = blstm_layer(x, sd, f"{p}.encoders.{l}", drop, l)
    return x


def forward(f, sd, num_layers, num_spk=2, num_noise=1, drop=_no_drop):
    """BLSTM.forward, blstm.py:68-88 -> (y_pred, masks)."""
    if f.dim() == 4:
        f = f.squeeze(0)
    f_orig = f
    h = encoder(mvn_time(f), sd, num_layers, drop=drop)
    m = torch.relu(F.linear(h, sd["linear.weight"], sd["linear.bias"]))
    masks = torch.chunk(m, num_spk + num_noise, dim=-1)
    y_pred = torch.stack([mk * f_orig for mk in masks[:-1]], dim=1)
    return y_pred, masks
