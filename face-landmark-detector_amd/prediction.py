"""Inference workflows: mirror of reference keypoints_detector/prediction.py on the HIP path.

Same function names, argument order and error behaviour as the reference
(`keypts_predict` :158-196, `_prediction` :199-222, `detect_marks` :16-96,
`model_from_checkpoint_path` :116-133), with the defects of SURVEY.md section 3.4 routed
around rather than reproduced (`keypts_predict` returns the class map instead of dropping it;
`os.path.isdir` is only asked about strings), plus the batch entry points `predict()` and
`align()` that BASELINE.json names.
"""
from __future__ import annotations

import glob
import json
import os

import numpy as np
import six

from . import _lib, alignment
from .data.generator import get_image_array, imread_bgr, resize_u8_device
from .networks.config import IMAGE_ORDERING


# ---- checkpoint discovery (reference training.py:41-71, prediction.py:116-133) ---------------------
def find_latest_checkpoint(checkpoints_path, fail_safe=True):
    """Newest `<checkpoints_path>.<epoch>[.npz]` by numeric suffix (training.py:41-71; the
    reference strips TensorFlow's `.index`, this build's container suffix is `.npz`)."""
    def epoch_of(path):
        tail = path.replace(checkpoints_path, "").strip(".")
        for suf in (".npz", ".index"):
            if tail.endswith(suf):
                tail = tail[: -len(suf)]
        return tail

    files = [f for f in glob.glob(checkpoints_path + ".*") if epoch_of(f).isdigit()]
    if not files:
        if not fail_safe:
            raise ValueError("Checkpoint path {0} invalid".format(checkpoints_path))
        return None
    return max(files, key=lambda f: int(epoch_of(f)))


def model_from_checkpoint_path(checkpoints_path):
    """prediction.py:116-133.  The sidecar `<ckpt>_config.json` holds model_class, n_classes,
    input_height, input_width (the trainer omits the last two, training.py:195-200: they then
    default to output_height-8 / output_width-8, the vanilla FCN-8 relation)."""
    from .networks.basic_models import LANDMARKS_MODELS
    assert (os.path.isfile(checkpoints_path + "_config.json")), "Checkpoint not found."
    model_config = json.loads(open(checkpoints_path + "_config.json", "r").read())
    latest_weights = find_latest_checkpoint(checkpoints_path)
    assert (latest_weights is not None), "Checkpoint not found."
    ih = model_config.get("input_height", model_config.get("output_height", 264) - 8)
    iw = model_config.get("input_width", model_config.get("output_width", 264) - 8)
    model = LANDMARKS_MODELS[model_config["model_class"]](model_config["n_classes"], input_height=ih,
                                                          input_width=iw)
    print("loaded weights ", latest_weights)
    status = model.load_weights(latest_weights)
    if status is not None:
        status.expect_partial()
    return model


# ---- FCN class-map prediction ------------------------------------------------------------------------
def _prediction(model, inp, input_width, input_height, output_height, output_width, n_classes,
                colors=None, show_legends=False, class_names=None, pred_dim=None, overlay_img=False,
                out_fname=None):
    """prediction.py:199-222: preprocess -> model.predict -> per-pixel argmax, int64 [H',W'].

    The preprocess (get_image_array sub_mean, :207), the forward (:208) and the argmax (:209)
    run as one launch sequence on the device; only the class map returns to the host.
    """
    import torch
    dev = _lib.require_gpu()
    if not isinstance(inp, np.ndarray) or inp.ndim != 3 or inp.shape[2] != 3 or inp.dtype != np.uint8:
        # non-uint8 arrays take the reference's host route (float images are only mean-shifted)
        x = get_image_array(inp, input_width, input_height, ordering=IMAGE_ORDERING)
        xd = torch.from_numpy(np.ascontiguousarray(x[None])).to(dev)
    else:
        d = resize_u8_device(torch.from_numpy(np.ascontiguousarray(inp)).to(dev), input_height, input_width)
        xd = d[None].contiguous()
    pr = model.forward_device(xd, "classmap")[0].to(torch.int64).cpu().numpy()
    assert pr.shape == (output_height, output_width)
    if out_fname is not None:
        from .utils.plots import visualize_keypoints
        seg_img = visualize_keypoints(pr, inp, n_classes=n_classes, colors=colors, overlay_img=overlay_img,
                                      show_legends=show_legends, class_names=class_names, pred_dim=pred_dim)
        from PIL import Image
        Image.fromarray(np.ascontiguousarray(seg_img[:, :, ::-1].astype(np.uint8))).save(out_fname)
    return pr


def keypts_predict(model=None, inp=None, out_fname=None, checkpoints_path=None, overlay_img=False,
                   class_names=None, show_legends=False, colors=None, pred_dim=None, read_image_type=1):
    """prediction.py:158-196, returning the class map (the reference drops it)."""
    if model is None and checkpoints_path is None:
        raise ValueError("Both model and checkpoint_path cannot be empty")
    if model is None and (checkpoints_path is not None):
        model = model_from_checkpoint_path(checkpoints_path)
    assert (inp is not None), "Invalid input, should be either directory, ndarray or image path"
    assert ((type(inp) is np.ndarray) or isinstance(inp, six.string_types)), \
        "Input should be the CV image or the input file name"
    if isinstance(inp, six.string_types):
        inp = imread_bgr(inp, read_image_type)
    assert (len(inp.shape) == 3 or len(inp.shape) == 1 or len(inp.shape) == 4), "Image should be h,w,3 "
    return _prediction(model, inp, model.input_width, model.input_height, model.output_height,
                       model.output_width, model.n_classes, colors, show_legends, class_names, pred_dim,
                       overlay_img, out_fname)


# ---- batch entry points --------------------------------------------------------------------------------
def predict(crops, model, n_points=4, thresh=0.0, to_input_space=False):
    """Batched landmarks: crops [N,H,W,3] (numpy or CUDA tensor; uint8 BGR or float32
    preprocessed) -> float64 [N,C,2] (x,y).

    Coordinates are in output-grid pixels (0..W'-1), as the reference's decode leaves them
    (utils/metrics.py:80); `to_input_space=True` rescales to input-crop pixels
    (x * W / W').  n_points < 1 selects the all-pixel centroid, else top-n (utils/metrics.py:58,66);
    the reference as shipped always runs n_points=4, thresh=0.
    Returns the type it was given (numpy -> numpy, CUDA tensor -> CUDA tensor).
    """
    import torch
    was_np = not isinstance(crops, torch.Tensor)
    xd = torch.from_numpy(np.ascontiguousarray(crops)).to(_lib.require_gpu()) if was_np else crops
    lm = model.forward_device(xd, "landmarks", n_points=n_points, thresh=thresh)
    if to_input_space:
        scale = torch.tensor([model.input_width / model.output_width, model.input_height / model.output_height],
                             dtype=torch.float64, device=lm.device)
        lm = torch.where(lm < 0, lm, lm * scale)
    return lm.cpu().numpy() if was_np else lm


def align(crops, model=None, landmarks=None, template=None, out_size=None, n_points=4, thresh=0.0):
    """Align face crops by the similarity transform that maps their landmarks onto a template.

    crops [N,H,W,3] uint8/float32; `landmarks` float64 [N,C,2] in output-grid pixels (predicted
    with `model` when omitted); `template` float64 [C,2] in aligned-image pixels (default
    `alignment.canonical_template`); out_size (h, w) defaults to the crop size.
    Returns (aligned float32 [N,h,w,3], M float32 [N,2,3], landmarks).
    """
    import torch
    was_np = not isinstance(crops, torch.Tensor)
    dev = _lib.require_gpu()
    xd = torch.from_numpy(np.ascontiguousarray(crops)).to(dev) if was_np else crops
    if landmarks is None:
        if model is None:
            raise ValueError("align needs either landmarks or a model")
        lm = model.forward_device(xd, "landmarks", n_points=n_points, thresh=thresh)
    else:
        lm = torch.as_tensor(landmarks, dtype=torch.float64).to(dev)
    h, w = int(xd.shape[1]), int(xd.shape[2])
    oh, ow = out_size if out_size is not None else (h, w)
    k = int(lm.shape[1])
    tm = alignment.canonical_template(k, oh, ow) if template is None else np.asarray(template, np.float64)
    tmd = torch.from_numpy(np.ascontiguousarray(tm)).to(dev)
    # landmarks live on the model's output grid (W' = W + 8): bring them to crop pixels
    sc = (1.0, 1.0)
    if model is not None:
        sc = (model.input_width / model.output_width, model.input_height / model.output_height)
    aligned, m = alignment.align_device(xd, lm, tmd, oh, ow, sc)
    if was_np:
        return aligned.cpu().numpy(), m.cpu().numpy(), lm.cpu().numpy()
    return aligned, m, lm


# ---- multi-face stream: detect_marks (prediction.py:16-96) ---------------------------------------------
def get_square_box(box):
    """prediction.py:36-65, integer box squaring by symmetric expansion."""
    left_x, top_y, right_x, bottom_y = box
    box_width = right_x - left_x
    box_height = bottom_y - top_y
    diff = box_height - box_width
    delta = int(abs(diff) / 2)
    if diff == 0:
        return box
    elif diff > 0:
        left_x -= delta
        right_x += delta
        if diff % 2 == 1:
            right_x += 1
    else:
        top_y -= delta
        bottom_y += delta
        if diff % 2 == 1:
            bottom_y += 1
    assert ((right_x - left_x) == (bottom_y - top_y)), 'Box is not square.'
    return [left_x, top_y, right_x, bottom_y]


def move_box(box, offset):
    """prediction.py:67-74."""
    return [box[0] + offset[0], box[1] + offset[1], box[2] + offset[0], box[3] + offset[1]]


def face_boxes(faces):
    """The box maths of detect_marks for a list of faces (prediction.py:76-78)."""
    out = []
    for face in faces:
        offset_y = int(abs((face[3] - face[1]) * 0.1))
        out.append(get_square_box(move_box(list(face), [0, offset_y])))
    return out


def crop_faces_device(frame, boxes, out_h, out_w, out=None, boxes_dev=None):
    """frame: CUDA uint8 [H,W,3]; boxes: list of (x0,y0,x1,y1) -> CUDA uint8 [K,out_h,out_w,3]
    (crop + bilinear resize, prediction.py:80-82; BGR kept, the FCN loader handles the order).
    `out`: write into this contiguous [K,out_h,out_w,3] uint8 tensor (a slice of a larger batch) instead of a new one;
    `boxes_dev`: the boxes already on the device (int32 [K,4]), e.g. a slice of one upload for several frames."""
    import torch
    lib = _lib.load()
    k = len(boxes) if boxes_dev is None else int(boxes_dev.shape[0])
    if boxes_dev is None:
        boxes_dev = torch.tensor(np.asarray(boxes, np.int32).reshape(k, 4), dtype=torch.int32, device=frame.device)
    elif boxes_dev.dtype != torch.int32 or not boxes_dev.is_cuda or not boxes_dev.is_contiguous() or boxes_dev.shape[1:] != (4,):
        raise ValueError("boxes_dev must be a contiguous CUDA int32 [K,4] tensor")
    if out is None:
        out = torch.empty((k, out_h, out_w, 3), dtype=torch.uint8, device=frame.device)
    elif (tuple(out.shape) != (k, out_h, out_w, 3) or out.dtype != torch.uint8 or not out.is_cuda
          or not out.is_contiguous()):
        raise ValueError("out must be a contiguous CUDA uint8 [%d,%d,%d,3] tensor" % (k, out_h, out_w))
    if k:
        _lib.check(lib.flm_crop_resize(_lib.stream_ptr(), _lib.ptr(frame.contiguous()), int(frame.shape[0]),
                                       int(frame.shape[1]), _lib.ptr(boxes_dev), k, _lib.ptr(out), out_h, out_w),
                   "flm_crop_resize")
    return out


def crop_frames_device(frames, faces_per_frame, out_h, out_w, frame_index=None):
    """The faces of several frames as ONE batch: faces_per_frame: per frame a list of detector boxes (x0,y0,x1,y1) ->
    (CUDA uint8 [K_total,out_h,out_w,3], squared boxes per frame).  One upload of all boxes; no per-frame allocation,
    no concatenation: the shape a multi-face stream feeds the landmark model with (prediction.py:99-113 loops per face).
    frames: either a list of CUDA uint8 [H,W,3] tensors (one crop / resize launch per frame straight into its slice of
    the batch) or ONE CUDA uint8 [F,H,W,3] tensor -- a ring of stream frames in one allocation -- with `frame_index`
    naming the ring slot of every entry of faces_per_frame (default 0, 1, ...): then all faces are cut in one launch."""
    import torch
    boxes = [face_boxes(f) for f in faces_per_frame]
    total = sum(len(b) for b in boxes)
    ring = isinstance(frames, torch.Tensor)
    if ring and (frames.dim() != 4 or frames.dtype != torch.uint8 or not frames.is_cuda or not frames.is_contiguous()
                 or frames.shape[3] != 3):
        raise ValueError("frames must be a list of CUDA uint8 [H,W,3] tensors or one contiguous CUDA uint8 [F,H,W,3] tensor")
    dev = frames.device if ring else (frames[0].device if len(frames) else _lib.require_gpu())
    out = torch.empty((total, out_h, out_w, 3), dtype=torch.uint8, device=dev)
    if total == 0:
        return out, boxes
    flat = np.concatenate([np.asarray(b, np.int32).reshape(len(b), 4) for b in boxes if len(b)], 0)
    if ring:
        slots = list(range(len(boxes))) if frame_index is None else [int(v) for v in frame_index]
        if len(slots) != len(boxes) or any(not (0 <= v < frames.shape[0]) for v in slots):
            raise ValueError("frame_index must name a ring slot in [0, %d) for every frame" % frames.shape[0])
        idx = np.concatenate([np.full(len(b), v, np.int32) for b, v in zip(boxes, slots) if len(b)])
        both = torch.from_numpy(np.concatenate([flat.reshape(-1), idx])).to(dev)   # one upload: boxes, then slots
        lib = _lib.load()
        fh, fw = int(frames.shape[1]), int(frames.shape[2])
        _lib.check(lib.flm_crop_resize_frames(_lib.stream_ptr(), _lib.ptr(frames), fh * fw * 3, int(frames.shape[0]), fh, fw,
                                              _lib.ptr(both), _lib.ptr(both[4 * total:]), total, _lib.ptr(out), out_h, out_w),
                   "flm_crop_resize_frames")
        return out, boxes
    bdev = torch.from_numpy(flat).to(dev)
    o = 0
    for frame, b in zip(frames, boxes):
        if len(b):
            crop_faces_device(frame, None, out_h, out_w, out=out[o:o + len(b)], boxes_dev=bdev[o:o + len(b)])
            o += len(b)
    return out, boxes


def detect_marks_batch(img, model, faces, n_points=4, thresh=0.0):
    """All faces of one frame in one batch: crop -> FCN -> decode -> back-projection
    (prediction.py:76-94).  Returns uint [K,C,2] image coordinates."""
    import torch
    if len(faces) == 0:   # a frame without faces: the reference's per-face loop (prediction.py:105-107) does nothing
        return np.zeros((0, model.n_classes, 2), np.uint)
    dev = _lib.require_gpu()
    frame = torch.from_numpy(np.ascontiguousarray(img)).to(dev) if not isinstance(img, torch.Tensor) else img
    boxes = face_boxes(faces)
    crops = crop_faces_device(frame, boxes, model.input_height, model.input_width)
    lm = model.forward_device(crops, "landmarks", n_points=n_points, thresh=thresh).cpu().numpy()
    out = []
    for k, fb in enumerate(boxes):
        # marks in [0,1] of the crop, then prediction.py:91-94
        marks = (lm[k] / np.array([model.output_width, model.output_height], np.float64)).astype(np.float32)
        marks *= (fb[2] - fb[0])
        marks[:, 0] += fb[0]
        marks[:, 1] += fb[1]
        out.append(np.maximum(marks, 0).astype(np.uint))
    return np.stack(out)


def detect_marks(img, model, face):
    """prediction.py:16-96 for one face; `model` is this package's FCN-8 model object."""
    return detect_marks_batch(img, model, [face])[0]


def video_predict(facedetector_fn, landmark_model, frames=None, on_frame=None, n_points=4, thresh=0.0):
    """prediction.py:99-113: per frame `rects = facedetector_fn(img)`, landmarks of every face, `draw_marks(img, marks)`.

    The reference reads camera 0 through cv2.VideoCapture and shows each frame with cv2.imshow until 'q' is pressed;
    neither exists here, so the frame source and the sink are arguments: `frames` is any iterable of uint8 BGR
    [H,W,3] arrays, `on_frame(img, marks)` receives the annotated frame (marks: uint [K,C,2], K may be 0) and ends
    the loop by returning False -- the 'q' key.  All faces of a frame go through ONE batched launch sequence
    (`detect_marks_batch`) instead of the reference's per-face model calls.  Returns the number of frames processed."""
    from .utils.plots import draw_marks
    if frames is None:
        raise ValueError("video_predict needs `frames` (an iterable of BGR uint8 images): there is no camera capture "
                         "without cv2 (prediction.py:100)")
    count = 0
    for img in frames:
        rects = facedetector_fn(img)
        marks = detect_marks_batch(img, landmark_model, list(rects), n_points=n_points, thresh=thresh)
        for k in range(marks.shape[0]):
            draw_marks(img, marks[k])
        count += 1
        if on_frame is not None and on_frame(img, marks) is False:
            break
    return count
