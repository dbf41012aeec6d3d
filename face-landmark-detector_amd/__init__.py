"""MI355X-native facial-landmark inference and alignment (hot path only).

Drop-in for the `keypoints_detector.prediction` / `keypoints_detector.networks`
path of sandyz1000/face-landmark-detector: same registry (`LANDMARKS_MODELS`),
same model-object contract (`predict`, `load_weights`, the six attributes of
networks/utils.py:32-37), same function names (`keypts_predict`, `_prediction`,
`detect_marks`, `transfer_target`), plus the `predict()` / `align()` batch entry
points.  All arithmetic runs in the hand-written HIP library behind the C ABI in
``include/flm.h``; there is no CPU fallback -- calling an op without the built
library raises.
"""
__version__ = "0.1.0"

from . import weights  # noqa: F401

__all__ = ["weights", "__version__"]
