"""CPU oracle for the css hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

This package is a plain fp32 CPU restatement (numpy / torch-CPU ops, own code)
of the reference's algorithm for the hot path named in BASELINE.json:

    STFT / IPD / MVN features  ->  Conformer | BLSTM encoder  ->  mask head
    ->  PIT-MSE,  plus the stitcher and the LR schedule around it.

Every function cites the reference file:line it restates (paths relative to
the desh2608/css checkout).  The reference has no tests of its own, so parity
is pinned by golden vectors produced by importing the reference in the build
container (tests/golden/make_golden.py, outputs committed under
tests/golden/*.npz) -- tests/test_oracle_golden.py checks every oracle
function against them.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this package, and only as the checker / the CPU baseline.  Nothing
under css_amd/ imports it: the product path fails loudly when the HIP
library is missing instead of falling back to this code.
"""

from . import feature, conformer, blstm, pit, stitcher, schedule, dropout, mixer, doa, separator, mxfp8, emulate  # noqa: F401
