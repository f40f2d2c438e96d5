"""ONNX-layout adapter -- the mask model behind the interface css/executor/separator.py:94-103 drives.

The reference's separator talks to an ``onnxruntime.InferenceSession`` (css/utils/model_util.py:12-25): input one
array ``(1, F, B*T)`` under the session's first input name, output three mask arrays (speaker 1, speaker 2, noise), which
``separate_one_batch`` stacks to ``(3, F', B*T)`` and reshapes to ``(B, F', T, 3)``.  onnxruntime and the ``.onnx``
checkpoints are third-party artefacts that are not in the reference repository; ``HipSession`` gives a css_amd model
that interface, so a caller written against the session (``get_inputs()[0].name``, ``run(None, {name: array})``) runs
unchanged with the masks coming from the HIP path.  ``get_onnx_model`` mirrors the reference's loader signature for
``.mdl`` checkpoints written by ``css_amd.utils.checkpoint`` (or by the reference's train.py: same keys).
"""
import numpy as np
import torch


class _Input:
    def __init__(self, name):
        self.name = name


class HipSession:
    """onnxruntime-session look-alike around a css_amd mask model (Conformer / BLSTM)."""

    def __init__(self, model, device="cuda", input_name="input"):
        self.model = model.to(device).eval()
        self.device = device
        self._inputs = [_Input(input_name)]

    def get_inputs(self):
        return self._inputs

    def run(self, output_names, feed):
        """feed: {input name: float array (1, F, L)} (L = B*T frames as ONE sequence, separator.py:95-96) ->
        [mask_1, ..., mask_K, mask_noise], each float32 (1, num_bins, L)."""
        (x,) = feed.values()
        x = torch.as_tensor(np.asarray(x), dtype=torch.float32)
        if x.dim() != 3 or x.shape[0] != 1:
            raise ValueError(f"expected an array of shape (1, F, frames), got {tuple(x.shape)}")
        f = x[0].t().contiguous()[None].to(self.device)          # [1, L, F]: frame-major, what the model consumes
        with torch.no_grad():
            m = self.model.forward_masks(f)                        # [1, L, (K+1)*num_bins]
        nb = self.model.num_bins
        parts = m[0].reshape(m.shape[1], -1, nb).permute(1, 2, 0)  # [K+1, num_bins, L]
        return [p[None].contiguous().cpu().numpy() for p in parts]


def get_onnx_model(onnx_model_path, sess_options=None, device="cuda", model=None):
    """Signature of css/utils/model_util.py:12 -> (session, device).  ``onnx_model_path``: a ``.mdl`` checkpoint in the
    reference's on-disk format ({"model": state_dict, ...}, train.py:304-317) loaded into ``model`` (a css_amd model of
    the matching architecture), or None to wrap ``model`` as it is."""
    if model is None:
        raise ValueError("css_amd has no ONNX runtime: pass model=<css_amd model> (its weights are loaded from the path)")
    if onnx_model_path is not None:
        from .checkpoint import load_checkpoint
        load_checkpoint(onnx_model_path, model)
    return HipSession(model, device=str(device)), torch.device(device)


def to_numpy(tensor):
    return tensor.detach().cpu().numpy()
