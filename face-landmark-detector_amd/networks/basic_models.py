"""Registry mirror of keypoints_detector/networks/basic_models.py:59-64.

`LANDMARKS_MODELS[name](n_classes, input_height=..., input_width=...)` is how the
reference builds a model (prediction.py:122-126, training.py:129-133).  The reference's
registry lists only ImageNet-backbone variants (whose constructors download weights) and a
broken 'default'; it has no entry for the vanilla FCN-8 the hot path is built on, so this
registry adds 'fcn_8' and points 'default' at it.
"""
from .fcn import (fcn_8, fcn_32, fcn_8_vgg, fcn_32_vgg, fcn_8_mobilenet, fcn_32_mobilenet, fcn_8_resnet50,
                  fcn_32_resnet50)


LANDMARKS_MODELS = {
    "fcn_8": fcn_8,
    "fcn_32": fcn_32,
    "default": fcn_8,
    # the reference's keys (basic_models.py:59-64), every one built WITHOUT the ImageNet download its constructor
    # performs by default (`pretrained=None` graphs); fp32 and bf16
    "fcn_8_resnet50": fcn_8_resnet50,
    "fcn_32_resnet50": fcn_32_resnet50,
    "fcn_8_mobilenet": fcn_8_mobilenet,
    "fcn_32_mobilenet": fcn_32_mobilenet,
    "fcn_8_vgg": fcn_8_vgg,
    "fcn_32_vgg": fcn_32_vgg,
}
