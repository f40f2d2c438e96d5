#!/usr/bin/env python3
"""Per-parameter gradient table of the HIP training path against torch.autograd through the oracle, twice: the plain
fp32 oracle, and the oracle with bf16 GEMM operands in forward and backward (oracle/emulate.py) - the floor that
"bf16 operand noise" really is.  2-block Conformer-base, batch-statistics BatchNorm, dropout ON (same masks on both
sides), PIT-MSE at the permutation the GPU chose.   python tools/grad_floor.py > profiles/r03_grad_table.txt"""
import os, sys
import numpy as np
import torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import grad_ref  # noqa: E402
from css_amd.models import MODELS  # noqa: E402
from css_amd.objectives import OBJECTIVES  # noqa: E402
from css_amd.utils import synthetic as syn  # noqa: E402


def rnd(shape, seed):
    return torch.from_numpy(np.random.RandomState(seed).standard_normal(shape).astype(np.float32))


for blocks, frozen in ((2, False), (1, True)):
    conf = dict(attention_dim=256, attention_heads=4, linear_units=1024, num_blocks=blocks, kernel_size=33, dropout_rate=0.1,
                relative_pos_emb=True)
    m = MODELS["Conformer"](257, 257, 2, 1, conf)
    m.load_state_dict(syn.fill_state_dict(syn.conformer_shapes(num_blocks=blocks), 43))
    m = m.cuda()
    if frozen:
        m.eval(); m.eval_gradients = True
    else:
        m.train()
    B, T = 3, 60
    f = rnd((B, T, 257), 14).abs().cuda()
    tg = [(rnd((B, T, 257), 60 + k).abs() * 0.5).cuda() for k in range(2)]
    obj = OBJECTIVES["MSE"].build_objective({})
    loss = obj(m, {"mix": f, "source1": tg[0], "source2": tg[1]}, device="cuda")
    loss.backward()
    grads = {k: p.grad.clone() for k, p in m.named_parameters()}
    perm = obj.last_permutation.cpu().tolist()
    tc = [t.cpu() for t in tg]
    lossf = lambda mk: grad_ref.pit_loss_with_perm(mk, f.cpu(), tc, perm)  # noqa: E731
    ref32 = grad_ref.oracle_param_grads(m, lossf, f, training_bn=not frozen)
    ref16 = grad_ref.oracle_param_grads(m, lossf, f, training_bn=not frozen, emulate_bf16=True)
    e32, w32 = grad_ref.compare(grads, ref32)
    e16, w16 = grad_ref.compare(grads, ref16)
    ee, _ = grad_ref.compare(ref16, ref32)
    mode = "frozen BatchNorm (eval_gradients), no dropout" if frozen else "train mode: batch-statistics BatchNorm, dropout 0.1"
    print(f"== Conformer-base, {blocks} block(s), {B} x {T} frames, {mode}")
    print(f"global relative L2 error of the HIP gradients: vs fp32 oracle {e32:.3e}; vs bf16-operand oracle {e16:.3e}; "
          f"(bf16-operand oracle vs fp32 oracle: {ee:.3e})")
    print(f"worst parameter: vs fp32 {w32[0]} {w32[1]:.3e}; vs bf16-operand oracle {w16[0]} {w16[1]:.3e}")
    t32 = dict((k, e) for k, e, _ in grad_ref.per_parameter(grads, ref32))
    print(f"{'parameter':62s} {'|grad|':>10s} {'vs bf16-emul':>12s} {'vs fp32':>10s}")
    for k, e, d in grad_ref.per_parameter(grads, ref16):
        print(f"{k:62s} {d:10.3e} {e:12.3e} {t32[k]:10.3e}")
    print()
