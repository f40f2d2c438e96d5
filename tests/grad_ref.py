"""Reference gradients for the GPU gradient tests: torch.autograd (CPU, fp32) through the ORACLE's restatement of the
reference model (oracle/conformer.py, oracle/blstm.py) and of the PIT-MSE objective (oracle/pit.py).  Test
infrastructure only - nothing under css_amd/ imports this."""
import torch

import oracle


def drop_of(model):
    """The oracle-side drop(x, site) that applies exactly the masks of the model's LAST training forward (the device
    seed after its bump), or the identity when the model does not drop (eval mode / dropout_rate 0)."""
    from css_amd.dropout import stream_of
    enc = getattr(model, "conformer", None) or model.blstm
    if not model.training or enc.dropout_rate <= 0.0:
        return None
    return oracle.dropout.Masks(int(model._drop_seed.item()), enc.dropout_rate, stream_of)


def oracle_masks(model, f, training_bn, drop=None):
    """(masks [N,T,(K+1)*F] as one tensor with grad, leaf state dict) of the oracle forward for `model`'s weights."""
    sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    for v in sd.values():
        if v.is_floating_point():
            v.requires_grad_(True)
    f = f.detach().cpu()
    if hasattr(model, "conformer") and model.conformer.idim != model.num_bins:
        # 7-channel IPD features: the reference's own forward raises at `mask * f_orig` (conformer.py:103, widths differ),
        # so the oracle is evaluated up to the masks - conformer.py:95-101: MVN over time, encoder, Linear, ReLU, chunk
        enc = model.conformer
        h = oracle.conformer.encoder(oracle.conformer.mvn_time(f), sd, n_head=enc.n_head, kernel_size=enc.kernel_size,
                                     num_blocks=len(enc.encoders), relative_pos_emb=enc.pos_emb is not None,
                                     training_bn=training_bn, **({"drop": drop} if drop is not None else {}))
        masks = (torch.relu(torch.nn.functional.linear(h, sd["linear.weight"], sd["linear.bias"])),)
    elif hasattr(model, "conformer"):
        enc = model.conformer
        _, masks = oracle.conformer.forward(f, sd, num_spk=model.num_spk, num_noise=model.num_noise, n_head=enc.n_head,
                                            kernel_size=enc.kernel_size, num_blocks=len(enc.encoders),
                                            relative_pos_emb=enc.pos_emb is not None, training_bn=training_bn,
                                            causal=bool(getattr(enc, "causal", False)), **({"drop": drop} if drop is not None else {}))
    else:
        _, masks = oracle.blstm.forward(f, sd, num_layers=len(model.blstm.encoders), num_spk=model.num_spk,
                                        num_noise=model.num_noise, **({"drop": drop} if drop is not None else {}))
    return torch.cat(masks, -1), sd


def pit_loss_with_perm(masks, f, targets, perm, nbins=257):
    """mean_b MSE(mask_{perm[b][i]} * f, target_i): the objective's value at the permutation the GPU chose."""
    B = f.shape[0]
    tot = 0
    for b in range(B):
        est = torch.stack([masks[b, :, perm[b][i] * nbins:(perm[b][i] + 1) * nbins] * f[b][:, :nbins] for i in range(len(targets))])
        tot = tot + ((est - torch.stack([t[b] for t in targets])) ** 2).mean()
    return tot / B


def oracle_param_grads(model, loss_of_masks, f, training_bn, emulate_bf16=False):
    """{parameter name: gradient} of loss_of_masks(masks) by autograd through the oracle (with the dropout masks of the
    model's last training forward, if it dropped).  emulate_bf16: every Linear / attention product with bf16 operands
    in forward and backward (oracle/emulate.py) - the arithmetic of the HIP path up to summation order."""
    import contextlib
    with (oracle.emulate.bf16_operands() if emulate_bf16 else contextlib.nullcontext()):
        masks, sd = oracle_masks(model, f, training_bn, drop_of(model))
        loss_of_masks(masks).backward()
    return {k: sd[k].grad for k, _ in model.named_parameters()}


def per_parameter(grads, ref):
    """[(name, relative L2 error, |ref|)] sorted by error, largest first"""
    rows = []
    for k, g in grads.items():
        r = ref[k]
        d = float((r ** 2).sum()) ** 0.5
        rows.append((k, float(((g.detach().cpu() - r) ** 2).sum()) ** 0.5 / max(d, 1e-30), d))
    return sorted(rows, key=lambda t: -t[1])


def compare(grads, ref, worst_min_numel=0):
    """(global relative L2 error, (worst parameter, its relative error)) of {name: grad} against the reference.
    worst_min_numel: parameters with fewer elements count in the global error but not for the worst one (see
    scalar_sums_within_conditioning)."""
    num = den = 0.0
    worst = ("", 0.0)
    for k, g in grads.items():
        r = ref[k]
        n, d = float(((g.detach().cpu() - r) ** 2).sum()), float((r ** 2).sum())
        num, den = num + n, den + d
        if g.numel() >= worst_min_numel and d > 1e-16 and (n / d) ** 0.5 > worst[1]:
            worst = (k, (n / d) ** 0.5)
    return (num / den) ** 0.5, worst


def scalar_sums_within_conditioning(grads, ref32, ref16, max_numel=2, floor=1.0, factor=5.0):
    """The conv module's 1- and 2-element parameters (pw_conv_1 / pw_conv_2: Conv1d(1, 2, 1) / Conv1d(2, 1, 1)) get
    gradients that are sums over EVERY position and channel with heavy cancellation; their relative error is as large as
    the sum is ill-conditioned at the shape and seed at hand, whatever computes it (measured on four shapes with the
    kernels of round 4 and of round 5 alike: 0.06 / 0.16-0.20 / 0.42-0.48 / 2.3 on pw_conv_2.bias while every vector
    and matrix gradient stays below 0.1).  The kernel-level test (test_gpu_bwd.py, csship_convmod_bwd against float64
    autograd on the SAME upstream gradient) pins these sums to 1e-4; at model level this is a sanity bound only: the HIP
    value may be ``factor`` times as far from the fp32 oracle as the oracle's own bf16-operand evaluation is, or
    ``floor``.  Returns the worst (name, error, allowed)."""
    worst = ("", 0.0, floor)
    for k, g in grads.items():
        if g.numel() > max_numel:
            continue
        r = ref32[k]
        d = float((r ** 2).sum()) ** 0.5
        if d <= 1e-12:
            continue
        e = float(((g.detach().cpu() - r) ** 2).sum()) ** 0.5 / d
        own = float(((ref16[k] - r) ** 2).sum()) ** 0.5 / d   # how far the bf16-operand oracle itself moves this sum
        # (a sign or scale error must not pass where the sum is well conditioned - 0.5 instead of `floor` then)
        allowed = max(0.5 if own < 0.1 else floor, factor * own)
        assert e < allowed, f"{k}: relative error {e:.3e} vs the fp32 oracle, the bf16-operand oracle itself moves {allowed / factor:.3e}"
        if e > worst[1]:
            worst = (k, e, allowed)
    return worst


def check_against_emulation(grads, ref16, global_tol, param_tol, min_numel=0):
    """HIP gradients vs the oracle with bf16 operands in forward and backward (oracle/emulate.py): global relative L2
    error and the worst parameter among those whose reference gradient is not numerically zero (>= 1e-4 of the largest
    parameter-gradient norm: the key bias - softmax is shift invariant - and a bias in front of a batch-statistics
    BatchNorm have mathematically zero gradients, there both sides hold rounding noise).  Returns (global, worst)."""
    rows = per_parameter(grads, ref16)
    big = max(d for _, _, d in rows)
    live = [(k, e) for k, e, d in rows if d >= 1e-4 * big and grads[k].numel() >= min_numel]
    num = sum(float(((grads[k].detach().cpu() - ref16[k]) ** 2).sum()) for k in grads)
    den = sum(float((ref16[k] ** 2).sum()) for k in grads)
    glob = (num / den) ** 0.5
    worst = max(live, key=lambda t: t[1])
    assert glob < global_tol, f"global relative gradient error vs the bf16-operand oracle {glob:.3e} >= {global_tol}"
    assert worst[1] < param_tol, f"{worst[0]}: relative gradient error vs the bf16-operand oracle {worst[1]:.3e} >= {param_tol}"
    return glob, worst
