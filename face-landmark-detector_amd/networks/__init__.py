from .basic_models import LANDMARKS_MODELS  # noqa: F401
from .fcn import (Fcn8Model, Fcn32Model, Fcn8VggModel, Fcn32VggModel, Fcn8MobilenetModel, Fcn32MobilenetModel,  # noqa: F401
                  Fcn8Resnet50Model, Fcn32Resnet50Model, fcn_8, fcn_32, fcn_8_vgg, fcn_32_vgg, fcn_8_mobilenet,
                  fcn_32_mobilenet, fcn_8_resnet50, fcn_32_resnet50)
