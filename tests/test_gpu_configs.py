"""Every BASELINE.json configuration at its FULL size through the C ABI (the other test files use shapes the oracle
finishes in seconds).  At full size the oracle checks a SLICE (eval mode is batch-independent: chunk b of the batch sees
the same arithmetic whatever else is in the batch) and size-independent properties cover the rest.

  configs[1]  Conformer-base 1ch, 32 x 400 frames, TRAIN mode (batch-statistics BatchNorm, dropout) - 2 blocks vs oracle
              on the whole batch (batch statistics need it), 16 blocks for determinism / finiteness of loss + gradients
  configs[2]  7-channel IPD features (32 x 7 x 400 frames -> [32, 400, 1799]) + Conformer-base idim 1799
  configs[3]  600 s recording -> 747 windows -> Separator -> Stitcher
  configs[4]  3 speakers + noise, Conformer-large (d 512, 8 heads, ffn 2048, 18 blocks), bf16 (fp8 decision)
"""
import numpy as np
import pytest
import torch

import oracle
from css_amd.utils import synthetic as syn

pytestmark = pytest.mark.gpu
DEV = "cuda"
B, T = 32, 400
N_SAMPLES = 512 + 256 * (T - 1)


def rnd(shape, seed, scale=1.0):
    return torch.from_numpy((scale * np.random.RandomState(seed).standard_normal(shape)).astype(np.float32))


def _conformer(nb, seed, idim=257, num_spk=2, d=256, h=4, ff=1024):
    from css_amd.models import MODELS
    conf = dict(attention_dim=d, attention_heads=h, linear_units=ff, num_blocks=nb, kernel_size=33, dropout_rate=0.1,
                relative_pos_emb=True)
    m = MODELS["Conformer"](idim, 257, num_spk, 1, conf)
    sd = syn.fill_state_dict(syn.conformer_shapes(idim=idim, num_spk=num_spk, attention_dim=d, attention_heads=h,
                                                  linear_units=ff, num_blocks=nb), seed)
    m.load_state_dict(sd)
    return m.to(DEV), sd


def test_config2_training_mode_full_batch_vs_oracle():
    """configs[1] in TRAIN mode at 32 x 400: features from raw audio, batch-statistics BatchNorm over all 12 800 frames,
    dropout, fused PIT-MSE; 2 encoder blocks so that the oracle can run the WHOLE batch (the statistics couple it)."""
    from css_amd import ops
    from css_amd.objectives import OBJECTIVES
    import grad_ref
    m, sd = _conformer(2, 81)
    m.train()
    mix, src = syn.synthetic_sources(B, N_SAMPLES, seed=4)
    audio = torch.stack([mix, src[0], src[1]], 0).to(DEV)
    mags = ops.stft(audio, ("mag",))["mag"]
    obj = OBJECTIVES["MSE"].build_objective({})
    loss = obj(m, {"mix": mags[0], "source1": mags[1], "source2": mags[2]}, device=DEV)
    loss.backward()
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in m.parameters())
    masks = m.forward_masks(mags[0])          # a second training forward: new dropout masks, same statistics path
    torch.set_num_threads(16)
    o_mag = [oracle.feature.stft(a)[0].transpose(1, 2).contiguous() for a in (mix, src[0], src[1])]
    with torch.no_grad():
        ref, _ = grad_ref.oracle_masks(m, o_mag[0], training_bn=True, drop=grad_ref.drop_of(m))
    mse = float(((masks.detach().cpu() - ref) ** 2).mean())
    print(f"[parity] config 2, train mode 32 x 400 (2 blocks): mask MSE vs oracle {mse:.3e}, loss {float(loss):.5f}")
    assert mse < 1e-4


def test_config2_eval_16_blocks_full_size_vs_oracle():
    """configs[1] as bench.py runs it - the seed-0 16-block Conformer-base, rank 0's mixtures, 32 x 400 frames, eval mode,
    features from raw audio, the fused head + PIT kernel - against the oracle at FULL size.
    The bench line's loss has read +7.8e-4 above the fp32 oracle for three rounds; tools/loss_bias.py
    (profiles/r06_loss_bias.txt) names the source: bf16 OPERAND rounding, a fixed perturbation of the model that does
    not average out over the 12 800 frames.  The oracle evaluated with the same operand roundings (oracle/emulate.py)
    reads +4.5e-4 itself, and the HIP path that rounds exactly those operands (no LayerNorm folding) agrees with it to
    3e-6; the default path folds the LayerNorm affines into the weights BEFORE rounding them (another rounding of the
    same weights, of the same size): +3.3e-4 from the emulation.  Asserted here: mask MSE of chunks 0-1 and the loss
    against fp32 (SURVEY 8d budgets), and the two internal bars against the bf16-operand oracle."""
    from css_amd import ops
    from oracle import emulate
    m, sd = _conformer(16, 0)
    m.eval()
    mix, src = syn.synthetic_sources(B, N_SAMPLES, seed=0)
    audio = torch.stack([mix, src[0], src[1]], 0).to(DEV)
    with torch.no_grad():
        mags = ops.stft(audio, ("mag",))["mag"]
        loss = float(m.forward_loss(mags[0], [mags[1], mags[2]])[0])
        masks = m.forward_masks(mags[0][:2].contiguous()).cpu()
        m.no_ln_fold = True
        loss_nf = float(m.forward_loss(mags[0], [mags[1], mags[2]])[0])
    torch.set_num_threads(16)
    o_mag = [oracle.feature.stft(x)[0].transpose(1, 2).contiguous() for x in (mix, src[0], src[1])]
    with torch.no_grad():
        y, o_masks = oracle.conformer.forward(o_mag[0], sd, num_blocks=16)
        l32 = float(oracle.pit.mse_objective(y, o_mag[1], o_mag[2]))
        with emulate.bf16_operands():
            y, _ = oracle.conformer.forward(o_mag[0], sd, num_blocks=16)
        l16 = float(oracle.pit.mse_objective(y, o_mag[1], o_mag[2]))
    mse = float(((masks - torch.cat(o_masks, -1)[:2]) ** 2).mean())
    print(f"[parity] config 2 eval, 16 blocks at 32 x 400: mask MSE of chunks 0-1 vs fp32 oracle {mse:.3e}; loss {loss:.7f} "
          f"(no LayerNorm folding {loss_nf:.7f}) vs fp32 oracle {l32:.7f} (rel {(loss - l32) / l32:+.2e}) and vs the "
          f"bf16-operand oracle {l16:.7f} (rel {(loss - l16) / l16:+.2e}; unfolded {(loss_nf - l16) / l16:+.2e})")
    assert mse < 1e-4
    assert abs(loss - l32) / l32 < 1e-3                 # bf16 operands vs the fp32 reference
    assert abs(loss_nf - l16) / l16 < 5e-5              # same operand roundings: summation order is all that is left
    assert abs(loss - l16) / l16 < 6e-4                 # the folded weights: a second rounding of the same size


def test_config2_training_step_16_blocks_full_size_is_deterministic():
    """The full Conformer-base training step at 32 x 400: two identical steps from identical state give bit-identical
    loss and gradients (no atomics, fixed reduction orders), everything finite."""
    from css_amd.objectives import OBJECTIVES
    from css_amd import dropout
    obj = OBJECTIVES["MSE"].build_objective({})
    g = torch.Generator().manual_seed(0)
    batch = {k: (torch.rand((B, T, 257), generator=g) * s).to(DEV) for k, s in (("mix", 1.0), ("source1", 0.5), ("source2", 0.5))}
    outs = []
    for _ in range(2):
        m, _ = _conformer(16, 82)
        m.train()
        dropout.set_seed(m, 41)
        loss = obj(m, batch, device=DEV)
        loss.backward()
        outs.append((float(loss), torch.cat([p.grad.reshape(-1) for p in m.parameters()]).clone()))
    assert outs[0][0] == outs[1][0] and torch.equal(outs[0][1], outs[1][1])
    assert np.isfinite(outs[0][0]) and bool(torch.isfinite(outs[0][1]).all()) and float(outs[0][1].abs().max()) > 0


def test_config3_seven_channel_full_size():
    """configs[2]: 32 chunks x 7 channels x 400 frames -> IPD features [32, 400, 1799] -> Conformer-base (idim 1799)
    -> PIT.  Chunks 0-1 vs the oracle (features and masks); all chunks: finite, equal to a separate run of the same
    chunks in a smaller batch (batch independence), IPD columns bounded by pi."""
    from css_amd.executor.feature import FeatureExtractor
    from css_amd import ops
    m, sd = _conformer(16, 83, idim=257 * 7)
    m.eval()
    mix, src = syn.synthetic_sources(B, N_SAMPLES, seed=6)
    x7 = syn.synthetic_multichannel(src, n_ch=7, seed=6)                     # [32, 7, n]
    fe = FeatureExtractor(512, 256, True, "1,0;2,0;3,0;4,0;5,0;6,0")
    with torch.no_grad():
        mag, f, _, _ = fe.forward_tf(x7.to(DEV), want_spectra=False)
        masks = m.forward_masks(f)
        tg = ops.stft(torch.stack([src[0], src[1]]).to(DEV), ("mag",))["mag"]
        loss, perm, _, _ = ops.pit_mse(masks, mag, [tg[0], tg[1]])
        f2 = fe.forward_tf(x7[:2].to(DEV), want_spectra=False)[1]
        masks2 = m.forward_masks(f2)
    assert f.shape == (B, T, 1799) and masks.shape == (B, T, 771)
    assert bool(torch.isfinite(f).all()) and bool(torch.isfinite(masks).all()) and bool(torch.isfinite(loss))
    assert float(f[..., 257:].abs().max()) <= np.pi + 1e-4
    assert torch.equal(f[:2], f2) and float(((masks[:2] - masks2) ** 2).mean()) < 1e-5
    torch.set_num_threads(16)
    with torch.no_grad():
        o_mag, o_f, _, _ = oracle.feature.feature_extractor(x7[:2], ipd_index="1,0;2,0;3,0;4,0;5,0;6,0")
        o_ft = o_f.transpose(1, 2)
        # masks from the GPU's own features (the IPD is ill-conditioned where a bin is near silent: its own test bounds it)
        # (the reference's own forward raises at idim != num_bins - mask * f_orig, conformer.py:103, DESIGN.md 2 - so the
        # oracle is called up to the mask head: encoder + Linear + ReLU, conformer.py:95-99)
        h = oracle.conformer.encoder(oracle.conformer.mvn_time(f[:2].cpu()), sd, num_blocks=16)
        o_masks = torch.relu(torch.nn.functional.linear(h, sd["linear.weight"], sd["linear.bias"]))
    assert float((mag[:2].cpu() - o_mag.transpose(1, 2)).abs().max()) < 1e-4 * float(o_mag.max())
    d_mag = (f[:2, :, :257].cpu() - o_ft[..., :257]).abs().max()
    assert float(d_mag) < 2e-3
    mse = float(((masks[:2].cpu() - o_masks) ** 2).mean())
    print(f"[parity] config 3, 7ch 32 x 400: mask MSE of chunks 0-1 vs oracle {mse:.3e}; loss {float(loss):.5f}")
    assert mse < 1e-4


def _seven_channel_features(b, t, seed):
    """[b, t, 1799] features with the statistics of config_7ch.yaml's extractor output: 257 magnitude columns, then 6 x 257
    IPD columns in (-pi, pi] (feature.py:105-134,169-171)."""
    mag = rnd((b, t, 257), seed).abs()
    ipd = torch.from_numpy(np.random.RandomState(seed + 1).uniform(-np.pi, np.pi, (b, t, 6 * 257)).astype(np.float32))
    return torch.cat([mag, ipd], -1)


def test_config3_seven_channel_training_gradients_vs_oracle():
    """configs[2] is a TRAINING configuration (conf/config_7ch.yaml:12, conformer.py:124-129 with idim 1799): 2-block
    Conformer on [B, T, 1799] features in train mode (batch-statistics BatchNorm, dropout on) -> PIT-MSE ->
    loss.backward() through the explicit HIP backward; every parameter gradient - the embedding weight gradient reduces
    over K = 1799, not a multiple of 8 or 32 - vs autograd through the oracle with the same dropout masks, same bounds
    as the 1-channel test (test_gpu_model.py::test_conformer_hip_backward_matches_oracle_autograd)."""
    from css_amd.objectives import OBJECTIVES
    import grad_ref
    m, sd = _conformer(2, 91, idim=257 * 7)
    m.train()
    b, t = 3, 60
    f = _seven_channel_features(b, t, 92).to(DEV)
    tg = [(rnd((b, t, 257), 95 + k).abs() * 0.5).to(DEV) for k in range(2)]
    obj = OBJECTIVES["MSE"].build_objective({})
    loss = obj(m, {"mix": f, "source1": tg[0], "source2": tg[1]}, device=DEV)
    loss.backward()
    grads = {k: p.grad.clone() for k, p in m.named_parameters()}
    assert all(g is not None and bool(torch.isfinite(g).all()) for g in grads.values())
    assert grads["conformer.embed.0.weight"].shape == (256, 1799)
    assert float(grads["conformer.embed.0.weight"][:, 257:].abs().max()) > 0      # the IPD columns reach the gradient
    perm = obj.last_permutation.cpu().tolist()
    tc = [x.cpu() for x in tg]
    of = lambda mk: grad_ref.pit_loss_with_perm(mk, f.cpu(), tc, perm)   # noqa: E731
    ref = grad_ref.oracle_param_grads(m, of, f, training_bn=True)
    ref16 = grad_ref.oracle_param_grads(m, of, f, training_bn=True, emulate_bf16=True)
    err, worst = grad_ref.compare(grads, ref)
    floor, floor_worst = grad_ref.compare({k: v for k, v in ref16.items()}, ref)   # what bf16 operands alone cost
    emb = "conformer.embed.0.weight"
    e_emb = grad_ref.compare({emb: grads[emb]}, ref)[0]
    e_emb16 = grad_ref.compare({emb: grads[emb]}, ref16)[0]
    f_emb = grad_ref.compare({emb: ref16[emb]}, ref)[0]
    print(f"[parity] config 3 training, 7ch idim 1799 (2 blocks): gradients vs fp32 oracle autograd {err:.3e} (worst {worst[0]} "
          f"{worst[1]:.3e}); the bf16-operand oracle itself vs fp32 {floor:.3e} (worst {floor_worst[0]} {floor_worst[1]:.3e}); "
          f"embedding weight [256, 1799]: {e_emb:.3e} vs fp32, {e_emb16:.3e} vs bf16-operand oracle (which is {f_emb:.3e} from fp32)")
    for k, e, d in grad_ref.per_parameter(grads, ref16)[:6]:
        print(f"    vs bf16-operand oracle: {k:55s} rel {e:.3e}  |ref| {d:.3e}")
    # the HIP path is no farther from the fp32 graph than its own arithmetic (bf16 operands, f32 accumulation) evaluated on
    # the CPU is, and tight against that evaluation (same bounds as the 1-channel test)
    assert err < max(3e-2, 1.5 * floor) and e_emb < max(3e-2, 1.5 * f_emb)
    # (measured: 5.0e-2 vs fp32 where the emulation itself is 5.1e-2 from fp32; 3.0e-2 against the emulation - two bf16
    # evaluations with different f32 summation orders decorrelate; the 1-channel model reads 2.4e-2 / 1.5e-2: the 1799-wide
    # embedding's input gradient noise is inherited by every parameter behind it)
    g16, w16 = grad_ref.check_against_emulation(grads, ref16, global_tol=4.5e-2, param_tol=0.15)
    print(f"[parity] ... vs the bf16-operand oracle: global {g16:.3e}, worst live parameter {w16[0]} {w16[1]:.3e}")


def test_config3_seven_channel_training_step_full_size_is_deterministic():
    """configs[2] at full size: one 16-block training step (forward, PIT, backward) on [32, 400, 1799] features from the
    7-channel extractor; two runs from identical state give bit-identical loss and gradients, everything finite, and
    the loss equals the eval-free statement of the same forward through ops.pit_mse on the training masks."""
    from css_amd.executor.feature import FeatureExtractor
    from css_amd.objectives import OBJECTIVES
    from css_amd import dropout, ops
    mix, src = syn.synthetic_sources(B, N_SAMPLES, seed=7)
    x7 = syn.synthetic_multichannel(src, n_ch=7, seed=7)
    fe = FeatureExtractor(512, 256, True, "1,0;2,0;3,0;4,0;5,0;6,0")
    with torch.no_grad():
        _, f, _, _ = fe.forward_tf(x7.to(DEV), want_spectra=False)
        tg = ops.stft(torch.stack([src[0], src[1]]).to(DEV), ("mag",))["mag"]
    assert f.shape == (B, T, 1799)
    obj = OBJECTIVES["MSE"].build_objective({})
    outs = []
    for _ in range(2):
        m, _ = _conformer(16, 93, idim=257 * 7)
        m.train()
        dropout.set_seed(m, 43)
        loss = obj(m, {"mix": f, "source1": tg[0], "source2": tg[1]}, device=DEV)
        loss.backward()
        outs.append((float(loss), torch.cat([p.grad.reshape(-1) for p in m.parameters()]).clone()))
        del m
    assert outs[0][0] == outs[1][0] and torch.equal(outs[0][1], outs[1][1])
    assert np.isfinite(outs[0][0]) and bool(torch.isfinite(outs[0][1]).all()) and float(outs[0][1].abs().max()) > 0


@pytest.mark.parametrize("batch_bn", [False, True])
def test_conformer_training_trajectory_vs_oracle(batch_bn):
    """N optimiser steps, not one gradient: 30 steps of the 16-block Conformer-base through the HIP training path
    (css_amd.optim.FusedAdam: clip 1.0 + Adam with weight decay, warm-up schedule of the reference's LRScheduler) against the SAME loop on the
    CPU - fp32 torch.autograd through the oracle + torch.nn.utils.clip_grad_norm_ + torch.optim.Adam
    (iteration_types.py:34-63).  Dropout 0 (the reference draws from Philox, the build from its counter hash: the
    trajectories would differ by mask, not by arithmetic); BatchNorm frozen at the running statistics (model.eval() +
    eval_gradients) or on batch statistics (model.train()).  B = 4, T = 100 - what the oracle steps through in ~1 s.
    Asserts the loss curve (1 % through the descent, 2 % on the plateau), the early gradient norms, the loss really
    falling, and the distance between the final parameter vectors small against the distance travelled (a wrong factor on any gradient family bends the curve or
    the end point; a 10 % per-parameter gradient tolerance cannot see that, VERDICT r3 weak 1b / missing 7)."""
    import types
    from css_amd.models import MODELS
    from css_amd.objectives import OBJECTIVES
    from css_amd.optim import FusedAdam
    from css_amd.trainer import LRScheduler
    steps, b, t = 30, 4, 100
    conf = dict(attention_dim=256, attention_heads=4, linear_units=1024, num_blocks=16, kernel_size=33, dropout_rate=0.0,
                relative_pos_emb=True)
    sd0 = syn.fill_state_dict(syn.conformer_shapes(num_blocks=16), 97)
    m = MODELS["Conformer"](257, 257, 2, 1, conf)
    m.load_state_dict(sd0)
    m = m.to(DEV)
    if batch_bn:
        m.train()
    else:
        m.eval()
        m.eval_gradients = True
    sched_args = types.SimpleNamespace(warmup=5, decay=0.0, fixed=0, min_lr=1e-9)
    lr, wd, clip = 1e-3, 1e-2, 1.0        # (the first steps' gradient norms are 2 - 4: the clip is active there)
    batches = [(rnd((b, t, 257), 500 + i).abs(), rnd((b, t, 257), 600 + i).abs() * 0.5, rnd((b, t, 257), 700 + i).abs() * 0.5)
               for i in range(4)]               # four batches, cycled: the loss falls on data it has seen
    opt = FusedAdam(m.parameters(), lr=lr, weight_decay=wd, max_norm=clip)
    sch = LRScheduler(opt, sched_args)
    obj = OBJECTIVES["MSE"].build_objective({})
    gpu_losses, gpu_norms = [], []
    for i in range(steps):
        mix, s1, s2 = (x.to(DEV) for x in batches[i % 4])
        opt.zero_grad()
        loss = obj(m, {"mix": mix, "source1": s1, "source2": s2}, device=DEV)
        loss.backward()
        gpu_norms.append(float(opt.step(max_norm=clip)))
        gpu_losses.append(float(loss))
        sch.step(1.0)
    got = {k: v.detach().cpu().clone() for k, v in m.named_parameters()}

    torch.set_num_threads(16)
    names = [k for k, _ in m.named_parameters()]
    sd = {k: v.clone() for k, v in sd0.items()}
    leaves = [sd[k].requires_grad_(True) for k in names]
    ref_opt = torch.optim.Adam(leaves, lr=lr, weight_decay=wd)
    ref_sch = LRScheduler(ref_opt, sched_args)
    ref_losses, ref_norms = [], []
    for i in range(steps):
        mix, s1, s2 = batches[i % 4]
        ref_opt.zero_grad()
        y_pred, _ = oracle.conformer.forward(mix, sd, num_blocks=16, training_bn=batch_bn)
        loss = oracle.pit.mse_objective(y_pred, s1, s2)
        loss.backward()
        ref_norms.append(float(torch.nn.utils.clip_grad_norm_(leaves, clip)))
        ref_opt.step()
        ref_losses.append(float(loss))
        ref_sch.step(1.0)
    rel = [abs(g - r) / abs(r) for g, r in zip(gpu_losses, ref_losses)]
    nrel = [abs(g - r) / abs(r) for g, r in zip(gpu_norms, ref_norms)]
    moved = sum(float(((sd[k].detach() - sd0[k]) ** 2).sum()) for k in names) ** 0.5
    apart = sum(float(((got[k] - sd[k].detach()) ** 2).sum()) for k in names) ** 0.5
    print(f"[parity] 30-step trajectory ({'batch-statistics' if batch_bn else 'frozen'} BatchNorm): loss {ref_losses[0]:.5f} -> "
          f"{ref_losses[-1]:.5f} (oracle), {gpu_losses[0]:.5f} -> {gpu_losses[-1]:.5f} (HIP); worst per-step loss error "
          f"{max(rel):.2e}, worst gradient-norm error {max(nrel):.2e}; |theta_hip - theta_ref| / |theta_ref - theta_0| = "
          f"{apart / moved:.3f}")
    for i in range(steps):
        print(f"    step {i:2d}: loss {ref_losses[i]:.5f} | {gpu_losses[i]:.5f}   grad norm {ref_norms[i]:.4f} | {gpu_norms[i]:.4f}")
    assert ref_losses[-1] < 0.8 * ref_losses[0]          # the recipe trains
    # the descent (steps 0 - 14, loss 0.50 -> 0.21) is tracked to 1 %; on the plateau behind it two bf16 / fp32 evaluations
    # of the same recipe wander apart slowly (measured: 1.2 % after 30 steps with the BatchNorm frozen, 0.5 % on batch
    # statistics - where the ORACLE's own gradient norm jumps between 0.09 and 0.23 from step to step)
    assert max(rel[:15]) < 1e-2 and max(rel) < 2e-2
    # gradient norms: while the model is far from a stationary point (the first steps, where the clip of 1.0 acts) the two
    # agree to a few per cent; near the plateau the true gradient is a small difference of large per-frame terms and the
    # bf16 operand noise does not cancel with it - not asserted there, the loss curve and the end point are
    # (the descent only - steps 0 - 14: on the plateau the ORACLE's own norm spikes to 0.3 - 0.36 at a step that differs
    # from run to run of the same CPU code - 16 threads, non-associative reductions - while the HIP column is bit-identical
    # between runs; a spike on one side only is not an error of the other)
    steep = [e for e, r in zip(nrel[:15], ref_norms[:15]) if r > 1.0]
    mid = [e for e, r in zip(nrel[:15], ref_norms[:15]) if r > 0.3]
    assert len(steep) >= 3 and max(steep) < 5e-2 and max(mid) < 0.1
    assert apart / moved < 0.05                          # measured 0.005 / 0.017


def test_config4_long_recording_747_windows():
    """configs[3]: one 600 s recording -> 747 sliding windows (2.4 s / 0.8 s hop) -> Separator (batches of 64) ->
    Stitcher.  Window-batch independence (the default pass takes 256 windows per launch sequence whatever the
    configured batch size; a second pass 13 at a time, whose batches are ragged), windows 0-5 and the
    last window vs the oracle, stitched length and the 1e-4 winner-take-all floor of stitcher.py:71-84."""
    from css_amd.executor import Separator, Stitcher
    m, sd = _conformer(2, 84)
    m.eval()
    sr = 16000
    n = 12800 * 746 + 38656 + 100            # 747 windows, ragged tail dropped like separator.py:67
    r = np.random.RandomState(11)
    env = np.repeat(r.uniform(0.2, 1.0, size=n // 16000 + 1), 16000)[:n].astype(np.float32)
    rec = torch.from_numpy((0.1 * r.standard_normal(n)).astype(np.float32) * env)[None]      # [1, n]
    cfg = {"batch_size": 64, "eval_win": 2.4, "eval_hop": 0.8, "frame_length": 512, "frame_shift": 256}
    with torch.no_grad():
        masks, mags = Separator(cfg, sr, device=DEV, backend="hip", model=m).separate(rec)
        cfg2 = dict(cfg, batch_size=13)
        masks_b, mags_b = Separator(cfg2, sr, device=DEV, backend="hip", model=m, device_batch=13).separate(rec)
    assert masks.shape == (747, 257, 150, 3) and mags.shape == (747, 257, 150)
    assert torch.equal(mags, mags_b) and float(((masks - masks_b) ** 2).mean()) < 1e-5
    assert bool(torch.isfinite(masks).all()) and float(masks.max()) <= 1.0
    win = rec.unfold(-1, 38656, 12800)[0]
    pick = [0, 1, 2, 3, 4, 5, 746]
    with torch.no_grad():
        o_mag, o_f, _, _ = oracle.feature.feature_extractor(win[pick])
        _, o_masks = oracle.conformer.forward(o_f.transpose(1, 2), sd, num_blocks=2)
    ref = torch.stack(o_masks, -1).permute(0, 2, 1, 3).clamp(max=1.0)
    mse = float(((masks[pick].cpu() - ref) ** 2).mean())
    print(f"[parity] config 4, 747 windows: mask MSE of windows 0-5 and 746 vs oracle {mse:.3e}")
    assert mse < 1e-4 and float((mags[pick].cpu() - o_mag).abs().max()) < 1e-4 * float(o_mag.max())
    st = Stitcher({"eval_win": 2.4, "eval_hop": 0.8, "hop_size": 256}, sr)
    perm = st.get_stitch(mags, masks)
    r1, r2, rn = st.get_connect(perm, masks)
    assert len(perm) == 746 and r1.shape == r2.shape == rn.shape == (257, 50 * 746 + 150)
    assert bool(torch.isfinite(r1).all()) and float(torch.minimum(r1, r2).min()) >= 0.0
    o_perm = oracle.stitcher.get_stitch(mags[:8].cpu(), masks[:8].cpu(), st.stitch_margin)
    assert [list(p) for p in o_perm] == [list(p) for p in perm[:7]]


def test_config5_conformer_large_full_size():
    """configs[4]: 3 speakers + noise head, Conformer-large (d 512 / 8 heads / ffn 2048 / 18 blocks) at 32 x 400,
    bf16 MFMA (the fp8 question is answered in DESIGN.md from tools/fp8_budget.py): run-to-run bitwise determinism,
    finite, K = 3 PIT permutations valid; chunk 0 vs the 18-block oracle."""
    from css_amd import ops
    m, sd = _conformer(18, 85, num_spk=3, d=512, h=8, ff=2048)
    m.eval()
    f = rnd((B, T, 257), 86).abs().to(DEV)
    tg = [(rnd((B, T, 257), 87 + k).abs() * 0.5).to(DEV) for k in range(3)]
    with torch.no_grad():
        masks = m.forward_masks(f)
        masks_again = m.forward_masks(f)
        loss, perm, _, _ = ops.pit_mse(masks, f, tg)
    assert masks.shape == (B, T, 4 * 257) and torch.equal(masks, masks_again)
    assert bool(torch.isfinite(masks).all()) and bool(torch.isfinite(loss))
    assert all(sorted(p) == [0, 1, 2] for p in perm.cpu().tolist())
    torch.set_num_threads(16)
    with torch.no_grad():
        _, o_masks = oracle.conformer.forward(f[:1].cpu(), sd, num_spk=3, n_head=8, num_blocks=18)
    mse = float(((masks[:1].cpu() - torch.cat(o_masks, -1)) ** 2).mean())
    print(f"[parity] config 5, Conformer-large 18 blocks at 32 x 400: mask MSE of chunk 0 vs oracle {mse:.3e}")
    assert mse < 1e-4


def test_config5_fp8_ffn_full_size():
    """configs[4] in the arithmetic it names, at its FULL size: Conformer-large (d 512 / 8 heads / ffn 2048 / 18 blocks),
    3 speakers + noise, 32 x 400 frames, every FeedForward on block-scaled fp8 operands (model.fp8_ffn = True ->
    csship_ffn_mx, v_mfma_scale_f32_16x16x128_f8f6f4); attention stays bf16 (priced, declined).
    All 32 chunks: bitwise run-to-run determinism, finite, valid K = 3 permutations, batch independence of chunk 0.
    Chunk 0 against (a) the oracle that emulates the same operand formats (oracle/mxfp8.py) and (b) the fp32 oracle.
    On these UNTRAINED seeded weights the format itself is far outside the 1e-4 budget (why the switch is opt-in; the
    trained-checkpoint statement is test_fp8_ffn_meets_the_mask_budget_on_a_trained_checkpoint), and through 18 blocks
    two evaluations that differ by bf16 rounding elsewhere take different e4m3 rounding decisions, so (a) is bounded
    relative to (b): the HIP path must be closer to the emulation than the emulation is to fp32."""
    from css_amd import ops
    m, sd = _conformer(18, 85, num_spk=3, d=512, h=8, ff=2048)
    m.eval()
    m.fp8_ffn = True
    f = rnd((B, T, 257), 86).abs().to(DEV)
    tg = [(rnd((B, T, 257), 87 + k).abs() * 0.5).to(DEV) for k in range(3)]
    with torch.no_grad():
        masks = m.forward_masks(f)
        masks_again = m.forward_masks(f)
        loss, perm, _, _ = ops.pit_mse(masks, f, tg)
        alone = m.forward_masks(f[:1].contiguous())
        m.fp8_ffn = False
        masks16 = m.forward_masks(f[:1].contiguous())
        loss16 = ops.pit_mse(m.forward_masks(f), f, tg)[0]
    assert masks.shape == (B, T, 4 * 257) and torch.equal(masks, masks_again)
    assert bool(torch.isfinite(masks).all()) and bool(torch.isfinite(loss))
    assert all(sorted(p) == [0, 1, 2] for p in perm.cpu().tolist())
    assert float(((masks[:1] - alone) ** 2).mean()) < 1e-5      # the hidden-chunk order rotates per panel: not bitwise
    torch.set_num_threads(16)
    kw = dict(num_spk=3, n_head=8, num_blocks=18)
    with torch.no_grad():
        with oracle.mxfp8.ffn_emulation("bf16"):
            _, o_emul = oracle.conformer.forward(f[:1].cpu(), sd, **kw)
        _, o_fp32 = oracle.conformer.forward(f[:1].cpu(), sd, **kw)
    o_emul, o_fp32 = torch.cat(o_emul, -1), torch.cat(o_fp32, -1)
    tight = float(((masks[:1].cpu() - o_emul) ** 2).mean())
    fmt = float(((masks[:1].cpu() - o_fp32) ** 2).mean())
    emul = float(((o_emul - o_fp32) ** 2).mean())
    bf16 = float(((masks16.cpu() - o_fp32) ** 2).mean())
    rel = abs(float(loss) - float(loss16)) / float(loss16)
    print(f"[parity] config 5 in fp8 (FeedForwards MXFP8, 18 blocks, 32 x 400), chunk 0: vs MXFP8-emulating oracle {tight:.2e}; "
          f"vs fp32 oracle {fmt:.2e} (emulation vs fp32 {emul:.2e}; bf16 path vs fp32 {bf16:.2e}); PIT loss {float(loss):.5f} "
          f"vs bf16 path {float(loss16):.5f} (rel {rel:.2e})")
    assert bf16 < 1e-4
    assert tight < 0.6 * emul and tight < 0.6 * fmt   # closer to its own emulation than the format is to fp32
    assert fmt > 3 * bf16 and fmt < 5e-2              # the switch changes the arithmetic, and stays a sane approximation
    assert rel < 5e-2


def _large(blocks, seed):
    from css_amd.models import MODELS
    conf = dict(attention_dim=512, attention_heads=8, linear_units=2048, num_blocks=blocks, kernel_size=33, dropout_rate=0.1,
                relative_pos_emb=True)
    sd = syn.fill_state_dict(syn.conformer_shapes(num_spk=3, attention_dim=512, attention_heads=8, linear_units=2048,
                                                  num_blocks=blocks), seed)
    m = MODELS["Conformer"](257, 257, 3, 1, conf)
    m.load_state_dict(sd)
    return m.to(DEV).eval(), sd


def test_config5_fp8_ffn_vs_mxfp8_emulating_oracle():
    """configs[4] ("fp8 MFMA attention/FFN", Conformer-large, 3 spk + noise) with model.fp8_ffn = True: every FeedForward
    on block-scaled fp8 operands (csship_ffn_mx).  What is TIGHT is the kernel: on identical inputs one FeedForward
    agrees with a float64 evaluation of the MX-quantised operands to 2e-6 (tests/test_gpu_ops.py::test_ffn_mx).  Through
    a whole model two evaluations that differ by bf16 rounding elsewhere (attention, projections) take different e4m3
    rounding decisions downstream - one flipped element is a 6 % step - so the model-level agreement with the oracle
    that emulates the same operand formats (oracle/mxfp8.py: MXFP8 in the FeedForwards, bf16 in every other Linear) is
    bounded at 3e-4 mask MSE here (measured 1.4e-4), an order of magnitude closer than the format's own distance to the
    fp32 oracle on these seeded random weights (1.1e-3: over the 1e-4 budget, which is why the switch is opt-in;
    DESIGN.md has the trained-checkpoint figure)."""
    m, sd = _large(2, 71)
    f = rnd((2, 120, 257), 72).abs()
    with torch.no_grad():
        m.fp8_ffn = True
        masks8 = m.forward_masks(f.to(DEV)).cpu()
        m.fp8_ffn = False
        masks16 = m.forward_masks(f.to(DEV)).cpu()
    kw = dict(num_spk=3, n_head=8, num_blocks=2)
    with oracle.mxfp8.ffn_emulation("bf16"):
        _, o_emul = oracle.conformer.forward(f, sd, **kw)
    _, o_fp32 = oracle.conformer.forward(f, sd, **kw)
    o_emul, o_fp32 = torch.cat(o_emul, -1), torch.cat(o_fp32, -1)
    tight = float(((masks8 - o_emul) ** 2).mean())
    fmt = float(((masks8 - o_fp32) ** 2).mean())
    bf16 = float(((masks16 - o_fp32) ** 2).mean())
    print(f"fp8_ffn masks: vs MXFP8-emulating oracle {tight:.2e}; vs fp32 oracle {fmt:.2e} (bf16 path: {bf16:.2e})")
    assert tight < 3e-4 and tight < 0.3 * fmt and bf16 < 1e-4
    assert fmt > 3 * bf16          # the switch really changes the arithmetic
    assert fmt < 5e-2              # ... and stays a sane approximation


def test_fp8_ffn_base_width_and_wide_bf16_path():
    """The same switch at the base width (d_model 256 through the any-width path), and the bf16 single-kernel
    FeedForward of that path (csship_ffn_wide) vs the unfused form it replaces."""
    from css_amd.models import MODELS
    conf = dict(attention_dim=256, attention_heads=4, linear_units=1024, num_blocks=2, kernel_size=33, dropout_rate=0.1,
                relative_pos_emb=True)
    sd = syn.fill_state_dict(syn.conformer_shapes(num_blocks=2), 73)
    m = MODELS["Conformer"](257, 257, 2, 1, conf)
    m.load_state_dict(sd)
    m = m.to(DEV).eval()
    f = rnd((3, 90, 257), 74).abs()
    _, o_fp32 = oracle.conformer.forward(f, sd, num_blocks=2)
    with oracle.mxfp8.ffn_emulation("bf16"):
        _, o_emul = oracle.conformer.forward(f, sd, num_blocks=2)
    o_fp32, o_emul = torch.cat(o_fp32, -1), torch.cat(o_emul, -1)
    with torch.no_grad():
        m.force_generic = True
        wide = m.forward_masks(f.to(DEV)).cpu()
        m.unfused_ffn = True
        unfused = m.forward_masks(f.to(DEV)).cpu()
        m.unfused_ffn = False
        m.fp8_ffn = True
        fp8 = m.forward_masks(f.to(DEV)).cpu()
    assert float(((wide - o_fp32) ** 2).mean()) < 1e-4 and float(((unfused - o_fp32) ** 2).mean()) < 1e-4
    assert float(((wide - unfused) ** 2).mean()) < 2e-5
    e_emul, e_fp32 = float(((fp8 - o_emul) ** 2).mean()), float(((fp8 - o_fp32) ** 2).mean())
    assert e_emul < 3e-4 and e_emul < 0.3 * e_fp32, (e_emul, e_fp32)


def test_fp8_ffn_meets_the_mask_budget_on_a_trained_checkpoint():
    """The second parity statement of the opt-in fp8 path: on a briefly TRAINED Conformer-base (the deterministic recipe of
    tools/fp8_trained_parity.py: 300 optimiser steps of css_amd training from the seeded initialisation) the HIP model
    with fp8_ffn = True stays within the 1e-4 mask-MSE budget against the fp32 oracle (the CPU emulation of round 2
    predicted 4.9e-5), while on the untrained seeded weights it does not (2.6e-3): trained weights have the block
    structure block-scaled fp8 is made for."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
    import fp8_trained_parity as tp
    r = tp.run(300, verbose=False)
    print(f"[parity] fp8_ffn after 300 training steps: vs fp32 oracle {r['fp8_vs_fp32']:.3e} (bf16 path {r['bf16_vs_fp32']:.3e}, "
          f"emulating oracle {r['emulation_vs_fp32']:.3e})")
    assert r["bf16_vs_fp32"] < 2e-5
    assert r["fp8_vs_fp32"] < 1e-4            # measured 2.0e-5 (profiles/r03_fp8_trained_parity.txt)
    # ... and tight against the oracle with the same operand formats.  The checkpoint is the product of 300 optimiser
    # steps, so any change of a summation order in the backward moves it (round 3's: 4.4e-6 against an emulation 2.0e-5
    # from the fp32 oracle; round 5's, after the backward's row kernels changed their instruction order: 1.8e-5 against
    # 5.6e-5): the bound is relative to how far the format itself takes THIS checkpoint from fp32
    assert r["fp8_vs_emulation"] < max(1e-5, 0.4 * r["emulation_vs_fp32"])
