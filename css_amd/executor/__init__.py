from .feature import FeatureExtractor, STFT  # noqa: F401
from .stitcher import Stitcher  # noqa: F401
from .separator import Separator  # noqa: F401
