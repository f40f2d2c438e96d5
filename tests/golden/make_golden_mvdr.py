#!/usr/bin/env python3
"""Golden vectors for the MVDR row from the reference's own numpy implementation
css/utils/mvdr_util.py (get_mask_scm, calc_bfcoeffs, get_bf, make_wta) - the only statement of the beamformer
arithmetic inside the reference checkout (the live path calls asteroid, which is not available).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_mvdr.py [/root/reference]

Inputs are regenerated from seeds on both sides; only outputs are stored (tests/golden/mvdr.npz).
"""
import os
import sys

sys.dont_write_bytecode = True
REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))
sys.path.insert(0, REF)

import numpy as np  # noqa: E402

from css.utils import mvdr_util as ref  # noqa: E402
from css_amd.utils.synthetic import mvdr_inputs as inputs  # noqa: E402


if __name__ == "__main__":
    mix, masks = inputs()
    wta = ref.make_wta([masks[0].copy(), masks[1].copy(), masks[2].copy()])
    tgt, tsum = ref.get_mask_scm(mix, masks[0])
    itf, _ = ref.get_mask_scm(mix, masks[1])
    noi, _ = ref.get_mask_scm(mix, masks[2])
    coef = ref.calc_bfcoeffs(noi + itf, tgt)
    res = ref.get_bf(mix, coef)
    np.savez_compressed(os.path.join(HERE, "mvdr.npz"), tgt_scm=tgt, mask_sum=tsum, coef=coef, res=res, wta=wta)
    print("mvdr.npz", tgt.shape, tgt.dtype, coef.shape, coef.dtype, res.shape, res.dtype)
