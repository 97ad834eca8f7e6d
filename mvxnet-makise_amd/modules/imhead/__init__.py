from .Head import ImageHead  # noqa: F401
from .Pipe import ImageFeatureExtractor, ImageFeatureFusion, featureMaping  # noqa: F401
