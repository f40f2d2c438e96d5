#!/usr/bin/env python3
"""Golden vectors for the Separator row, the DOA merge (8f N4) and checkpoint averaging (a17), made
by RUNNING the reference in the build container:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_separator.py [/root/reference]

css.executor.separator imports onnxruntime at module top (css/utils/model_util.py:1); the package is not installed,
so a stub module named ``onnxruntime`` is registered in sys.modules first - every line of
separator.py:9-250 and combine_models.py:17-110 that is exercised below is the reference's own code; the ONNX model
itself (a third-party artefact) is replaced by css_amd.utils.synthetic.FakeMaskSession, a closed-form function of
the session input that both sides can evaluate.

  separator.npz   steervec_7ch table + angles; doa_likelihood (softmax off and on) and angle_merge on
                  synthetic.doa_inputs; separate() of a 7-channel and a 1-channel recording through the fake session
                  (unfold windows, (1, F, B*T) session layout, reshape, clamp, merge) -> masks, mags
  combine.npz     combine_models.py on three tiny Conformer job checkpoints whose BatchNorm num_batches_tracked
                  buffers (0-dim int64) differ: averaged parameters and the scalar-buffer result
"""
import json
import os
import sys
import tempfile
import types

sys.dont_write_bytecode = True
REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(REF, "scripts", "python"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from css_amd.utils import synthetic as syn  # noqa: E402

torch.set_num_threads(8)

import importlib.machinery  # noqa: E402

stub = types.ModuleType("onnxruntime")
stub.__spec__ = importlib.machinery.ModuleSpec("onnxruntime", None)  # torch._dynamo walks sys.modules with find_spec
stub.get_device = lambda: "CPU"
stub.InferenceSession = lambda *a, **k: syn.FakeMaskSession()
sys.modules["onnxruntime"] = stub

from css.executor.separator import Separator  # noqa: E402


def separator_goldens():
    cfg7 = dict(batch_size=1, eval_win=2.4, eval_hop=0.8, frame_length=512, frame_shift=256,
                ipd="1,0;2,0;3,0;4,0;5,0;6,0", merge=True, merge_threshold=16, model_path="unused.onnx")
    sep = Separator(cfg7, 16000, None, "cpu", "onnx")
    out = {"steervec_re": sep.steervec.real.numpy(), "steervec_im": sep.steervec.imag.numpy(),
           "angles": sep.angles.numpy(), "lo_hi": np.array([sep._lo, sep._hi])}
    X, mask = syn.doa_inputs(sep.steervec)
    binmask = (mask > 0.5).float()
    out["like"] = sep.doa_likelihood(X, sep.steervec, binmask, compression=0.5, epsilon=1e-12, lowcut=sep._lo,
                                     highcut=sep._hi, softmax=False).numpy()
    out["like_softmax"] = sep.doa_likelihood(X, sep.steervec, mask, compression=1 / 3, lowcut=sep._lo,
                                             highcut=sep._hi, softmax=True).numpy()
    out["merged"] = sep.angle_merge(X, mask.clone(), thresh=16).numpy()
    # separate(): 3 windows of 38 656 samples, hop 12 800, + a ragged tail that unfold drops (separator.py:67)
    n = 12800 * 2 + 38656 + 777
    mix, src = syn.synthetic_sources(1, n, seed=5)
    rec7 = syn.synthetic_multichannel(src, n_ch=7, seed=5)[0]            # [7, n]
    m7, g7 = sep.separate(rec7)
    out["sep7_masks"], out["sep7_mags"] = m7.numpy(), g7.numpy()
    cfg1 = dict(batch_size=1, eval_win=2.4, eval_hop=0.8, frame_length=512, frame_shift=256, merge=False,
                merge_threshold=None, model_path="unused.onnx")
    sep1 = Separator(cfg1, 16000, None, "cpu", "onnx")
    m1, g1 = sep1.separate(mix)                                            # [1, n]
    out["sep1_masks"], out["sep1_mags"] = m1.numpy(), g1.numpy()
    np.savez_compressed(os.path.join(HERE, "separator.npz"), **out)
    print("separator.npz", {k: v.shape for k, v in out.items()})


def combine_goldens():
    import combine_models as cm
    conf = dict(model="Conformer", objective="MSE", optim="adam", lr=1e-4, weight_decay=1e-2, idim=257, num_bins=257,
                num_spk=2, num_noise=1, conformer_attention_dim=64, conformer_attention_heads=1,
                conformer_linear_units=64, conformer_num_blocks=1, conformer_kernel_size=33, conformer_dropout_rate=0.1,
                conformer_causal=False, conformer_relative_pos_emb=True)
    import css.models as models
    tracked = [500, 500, 501]
    with tempfile.TemporaryDirectory() as d:
        json.dump(conf, open(os.path.join(d, "conf.json"), "w"))
        paths, sums = [], None
        for j in range(3):
            torch.manual_seed(100 + j)
            m = models.MODELS["Conformer"].build_model(conf)
            sd = m.state_dict()
            for k in sd:
                if sd[k].dim() == 0:
                    sd[k].fill_(tracked[j])
            opt = torch.optim.Adam(m.parameters(), lr=1e-4)
            p = os.path.join(d, f"1.{j + 1}.mdl")
            torch.save({"model": sd, "objective": {}, "optimizer": opt.state_dict(), "lr_sched": None, "epoch": 1}, p)
            paths.append(p)
        omodel = os.path.join(d, "1.mdl")
        argv, sys.argv = sys.argv, ["combine_models.py", omodel, os.path.join(d, "conf.json"), "--save-models", "--models"] + paths
        try:
            cm.main()
        finally:
            sys.argv = argv
        res = torch.load(omodel, map_location="cpu")["model"]
    keys = ["encoder.encoders.0.conv.bn.num_batches_tracked", "encoder.encoders.0.conv.bn.running_var",
            "encoder.encoders.0.feed_forward_in.w_1.weight", "linear.bias"]
    keys = [k for k in keys if k in res] or list(res)[:4]
    scalar = [k for k in res if res[k].dim() == 0]
    out = {"seeds": np.array([100, 101, 102]), "tracked": np.array(tracked)}
    for i, k in enumerate(sorted(set(keys + scalar))):
        out[f"key{i}"] = np.array(k)
        out[f"val{i}"] = res[k].numpy()
    np.savez_compressed(os.path.join(HERE, "combine.npz"), **out)
    print("combine.npz", {str(out[f'key{i}']): out[f'val{i}'].shape for i in range(len(set(keys + scalar)))},
          {k: int(res[k]) for k in scalar})


if __name__ == "__main__":
    separator_goldens()
    combine_goldens()
