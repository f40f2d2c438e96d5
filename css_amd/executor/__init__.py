from .feature import FeatureExtractor, STFT  # noqa: F401
