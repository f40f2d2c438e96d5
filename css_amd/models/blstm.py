"""BLSTM mask estimator -- drop-in for css.models.blstm.BLSTM (blstm.py:17-160).

Same constructor / ``add_args`` / ``build_model`` / ``state_dict`` keys as the reference; the torch sub-modules
are parameter containers only.  Forward = HIP kernels: MVN statistics -> embed GEMM (MVN folded into the A
load) -> LayerNorm+ReLU -> per layer [input-projection GEMM for both directions and all T, sequential LSTM
recurrence kernel, LayerNorm] -> mask-head GEMM + ReLU -> mask application.  Dropout is not applied
(see models/conformer.py for the list of deliberate deviations).
"""
import torch
from torch import nn

from .. import ops
from .._lib import check, lib

DEFAULT_BLSTM_CONF = {"hidden_dim": 512, "num_layers": 4, "dropout_rate": 0.1}


class _BLSTMLayer(nn.Module):  # blstm.py:136-147
    def __init__(self, h_dim, dropout_rate):
        super().__init__()
        self.lstm = nn.LSTM(input_size=h_dim, hidden_size=h_dim // 2, num_layers=1, batch_first=True, bidirectional=True)
        self.layer_norm = nn.LayerNorm(h_dim)
        self.dropout = nn.Dropout(dropout_rate, inplace=True)


class BLSTMEncoder(nn.Module):  # blstm.py:96-120
    def __init__(self, idim=257, hidden_dim=1024, num_layers=4, dropout_rate=0.1):
        super().__init__()
        self.embed = nn.Sequential(nn.Linear(idim, hidden_dim), nn.LayerNorm(hidden_dim), nn.Dropout(dropout_rate), nn.ReLU())
        self.encoders = nn.Sequential(*[_BLSTMLayer(hidden_dim, dropout_rate) for _ in range(num_layers)])
        self.dropout_rate = float(dropout_rate)  # applied by the training forward (css_amd/dropout.py)
        self.idim, self.hidden_dim = idim, hidden_dim


class BLSTM(nn.Module):
    @staticmethod
    def add_args(parser):  # blstm.py:23-30
        parser.add_argument("--idim", type=int, default=257)
        parser.add_argument("--num-bins", type=int, default=257)
        parser.add_argument("--num-spk", type=int, default=2)
        parser.add_argument("--num-noise", type=int, default=1)
        parser.add_argument("--blstm-hdim", type=int, default=1024)
        parser.add_argument("--blstm-num-layers", type=int, default=3)
        parser.add_argument("--blstm-dropout-rate", type=float, default=0.1)

    @classmethod
    def build_model(cls, conf):  # blstm.py:33-46
        blstm_conf = {"hidden_dim": int(conf["blstm_hdim"]), "num_layers": int(conf["blstm_num_layers"]),
                      "dropout_rate": float(conf["blstm_dropout_rate"])}
        return cls(in_features=conf["idim"], num_bins=conf["num_bins"], num_spk=conf["num_spk"],
                   num_noise=conf["num_noise"], blstm_conf=blstm_conf)

    def __init__(self, in_features=257, num_bins=257, num_spk=2, num_noise=1, blstm_conf=DEFAULT_BLSTM_CONF):
        super().__init__()
        self.blstm = BLSTMEncoder(in_features, **blstm_conf)
        self.num_bins, self.num_spk, self.num_noise = num_bins, num_spk, num_noise
        self.linear = nn.Linear(blstm_conf["hidden_dim"], num_bins * (num_spk + num_noise))
        self._packed, self._packed_key = None, None
        self.register_buffer("_drop_seed", torch.zeros(1, dtype=torch.int32), persistent=False)  # see Conformer

    def _pack(self):
        from .. import param_generation  # FusedAdam rewrites parameter storage without touching data_ptr / _version
        key = (param_generation(),) + tuple((p.data_ptr(), p._version) for p in self.parameters())
        if self._packed is not None and key == self._packed_key:
            return self._packed
        enc = self.blstm
        if self.linear.weight.device.type != "cuda":
            raise RuntimeError("css_amd.BLSTM runs on an MI355X only; move the module to 'cuda' (no CPU path)")
        h = enc.hidden_dim
        H = h // 2
        if h % 8 or H % 16 or (H > 128 and H % 128):
            raise NotImplementedError("css_amd.BLSTM needs hidden_dim % 32 == 0 (and % 256 == 0 above 256)")
        Hp = (H + 31) // 32 * 32
        P = {"embed_w": ops.pad_weight_bf16(enc.embed[0].weight), "head_w": ops.pad_weight_bf16(self.linear.weight),
             "layers": []}
        for lay in enc.encoders:
            l = lay.lstm
            w_ih = torch.cat([l.weight_ih_l0.detach(), l.weight_ih_l0_reverse.detach()], 0)        # [8H, h]
            b = torch.cat([l.bias_ih_l0.detach() + l.bias_hh_l0.detach(),
                           l.bias_ih_l0_reverse.detach() + l.bias_hh_l0_reverse.detach()], 0).float().contiguous()
            whh = []
            for w in (l.weight_hh_l0, l.weight_hh_l0_reverse):                                   # [4H, H] each
                wp = torch.zeros(4 * H, Hp, device=w.device)
                wp[:, :H] = w.detach().float()
                whh.append(ops.pack_frag(wp))
            P["layers"].append({"w_ih": ops.pad_weight_bf16(w_ih), "b": b, "whh": torch.cat(whh).contiguous()})
        self._packed, self._packed_key = P, key
        return P

    def forward_masks(self, f):
        """HIP forward.  In training mode under autograd the result carries a grad_fn (css_amd/autograd.py BlstmTrain:
        the recurrence also stores gates and cell states for the BPTT kernel); in eval mode the plain inference path
        runs unless ``self.eval_gradients`` is set."""
        if (torch.is_grad_enabled() and (self.training or getattr(self, "eval_gradients", False))
                and any(p.requires_grad for p in self.parameters())):
            from ..autograd import masks_for_training
            return masks_for_training(self, f)
        return self._forward_masks_hip(f)

    def _forward_masks_hip(self, f):
        ops._need(f, torch.float32, "f")
        enc = self.blstm
        if f.dim() != 3 or f.shape[2] != enc.idim:
            raise RuntimeError(f"expected [N, T, {enc.idim}] features, got {tuple(f.shape)}")
        f = f.contiguous()
        P = self._pack()
        B, T, _ = f.shape
        M, h = B * T, enc.hidden_dim
        H = h // 2
        mean, rinv = ops.mvn_stats(f)
        e = ops.gemm(f.view(M, -1), P["embed_w"], enc.embed[0].bias.detach(), K=enc.idim, mvn=(mean, rinv), rows_per_batch=T)
        x = torch.empty_like(e)
        check(lib().csship_layernorm_relu(ops._ptr(e), M, h, ops._ptr(enc.embed[1].weight.detach()),
                                          ops._ptr(enc.embed[1].bias.detach()), 1, None, ops._ptr(x), ops._stream()))
        for lay, lp in zip(enc.encoders, P["layers"]):
            gx = ops.gemm(x, lp["w_ih"], lp["b"], K=h)  # [M, 8H] = (dir, 4H): x W_ih^T + b_ih + b_hh
            y = ops.lstm_bidir(gx, lp["whh"], B, T, H)
            x = torch.empty(M, h, device=f.device, dtype=torch.float32)
            check(lib().csship_layernorm_relu(ops._ptr(y), M, h, ops._ptr(lay.layer_norm.weight.detach()),
                                              ops._ptr(lay.layer_norm.bias.detach()), 0, None, ops._ptr(x), ops._stream()))
        n_out = self.linear.out_features
        ld = (n_out + 3) // 4 * 4
        masks = torch.empty(B, T, ld, device=f.device, dtype=torch.float32)[:, :, :n_out]
        ops.gemm(x, P["head_w"], self.linear.bias.detach(), K=h, relu=True, out=masks.view(M, n_out) if ld == n_out
                 else masks.as_strided((M, n_out), (ld, 1)))
        return masks

    def forward(self, f):
        if f.ndim == 4:
            f = f.squeeze(0)  # blstm.py:75-76
        m = self.forward_masks(f)
        masks = torch.chunk(m, self.num_spk + self.num_noise, -1)
        y_pred = ops.apply_masks(m, f, self.num_spk, self.num_bins)
        return y_pred, masks
