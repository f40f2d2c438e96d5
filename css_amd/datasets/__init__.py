"""Dataset registry -- same shape as css/datasets/__init__.py:19-21.  The reference's only dataset needs Lhotse
(third-party, unpinned, out of scope); the synthetic counterpart produces the same batch dict on the GPU."""
from . import synthetic

DATASETS = {"SyntheticSeparation": synthetic.SyntheticSeparationDataset,      # a fixed pool of pre-mixed batches
            "OnDeviceMixture": synthetic.OnDeviceMixtureDataset}              # per-step random mixing on the GPU (N2)
