from .basic_models import LANDMARKS_MODELS  # noqa: F401
from .fcn import Fcn8Model, fcn_8  # noqa: F401
