from .basic_models import LANDMARKS_MODELS  # noqa: F401
from .fcn import Fcn8Model, Fcn32Model, fcn_8, fcn_32  # noqa: F401
