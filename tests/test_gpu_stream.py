"""GPU parity for the multi-face stream shape (BASELINE configs[4]):
1080p frame + face boxes -> detect_marks box maths -> crop+resize -> batched FCN landmarks ->
back-projection (reference prediction.py:16-96), and predict()/align() end to end."""
import numpy as np
import pytest
import torch

from oracle import decode_ref, fcn_ref, warp_ref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup():
    import flm_amd  # noqa: F401
    from flm_amd import prediction
    from flm_amd.networks import LANDMARKS_MODELS
    from flm_amd.weights import synth_fcn8_weights
    params = synth_fcn8_weights(68, seed=2)
    model = LANDMARKS_MODELS["fcn_8"](68, input_height=256, input_width=256)
    model.load_weights(params)
    return prediction, model, params


def test_detect_marks_batch_1080p(setup):
    prediction, model, params = setup
    rng = np.random.default_rng(5)
    frame = rng.integers(0, 256, (1080, 1920, 3), dtype=np.uint8)
    faces = [[100, 120, 300, 360], [900, 400, 1296, 700], [1700, 800, 1900, 1060]]
    boxes = prediction.face_boxes(faces)
    assert boxes == [warp_ref.square_box_ref(f) for f in faces]
    fd = torch.from_numpy(frame).cuda()
    crops = prediction.crop_faces_device(fd, boxes, 256, 256)
    exp_crops = warp_ref.crop_resize_ref(frame, np.asarray(boxes), 256, 256)
    assert np.array_equal(crops.cpu().numpy(), exp_crops)      # integer fixed point on both sides
    marks = prediction.detect_marks_batch(frame, model, faces, n_points=0)
    assert marks.shape == (3, 68, 2) and marks.dtype == np.uint
    cg = exp_crops
    x = np.stack([fcn_ref.get_image_array_ref(c) for c in cg])
    pr = fcn_ref.fcn8_predict_ref(x, params).reshape(3, 264, 264, 68)
    with np.errstate(all="ignore"):
        lm = decode_ref.transfer_target_ref(pr, 0, 0).reshape(3, 68, 2)
    for k, fb in enumerate(boxes):
        m01 = (lm[k] / np.array([264.0, 264.0])).astype(np.float32)
        exp = warp_ref.backproject_marks_ref(np.maximum(m01, 0), fb)
        # astype(uint) truncates: a 1e-5 px difference can step an integer boundary
        assert np.abs(marks[k].astype(np.int64) - exp.astype(np.int64)).max() <= 1
    one = prediction.detect_marks(frame, model, faces[1])
    assert one.shape == (68, 2)


def test_predict_and_align_end_to_end(setup):
    prediction, model, params = setup
    rng = np.random.default_rng(6)
    crops = rng.integers(0, 256, (3, 256, 256, 3), dtype=np.uint8)
    lm = prediction.predict(crops, model, n_points=0)               # numpy in -> numpy out
    assert isinstance(lm, np.ndarray) and lm.shape == (3, 68, 2) and lm.dtype == np.float64
    lm_in = prediction.predict(crops, model, n_points=0, to_input_space=True)
    np.testing.assert_allclose(lm_in, lm * (256.0 / 264.0), rtol=1e-12)
    aligned, m, lm2 = prediction.align(crops, model=model, out_size=(112, 112), n_points=0)
    assert aligned.shape == (3, 112, 112, 3) and aligned.dtype == np.float32 and m.shape == (3, 2, 3)
    assert np.array_equal(lm2, lm)
    from flm_amd import alignment
    tm = alignment.canonical_template(68, 112, 112)
    # the fit: float64 sums in landmark order on both sides, the grid-to-crop scale a float64 product inside the kernel,
    # rounded to float32 once -> the same bits; the warp: 1 ULP of the restatement (north_star), every value
    m_ref = warp_ref.similarity_ref(lm * np.array([256.0 / 264.0, 256.0 / 264.0]), tm)
    assert np.array_equal(m, m_ref)
    exp = warp_ref.warp_affine_ref(crops, m, 112, 112)
    a, b = aligned.view(np.int32).astype(np.int64), exp.view(np.int32).astype(np.int64)
    a, b = np.where(a < 0, -(a & 0x7fffffff), a), np.where(b < 0, -(b & 0x7fffffff), b)   # monotone integer image of floats
    assert np.abs(a - b).max() <= 1
    # the aligned landmarks land on the template in the least-squares sense: residual no larger than
    # before alignment
    pts = lm * (256.0 / 264.0)
    mapped = np.einsum("nij,nkj->nki", m[:, :, :2].astype(np.float64), pts) + m[:, None, :, 2]
    assert np.linalg.norm(mapped - tm, axis=-1).mean() <= np.linalg.norm(pts * (112 / 256) - tm, axis=-1).mean() + 1e-6


def test_keypts_predict_returns_class_map(setup, tmp_path):
    prediction, model, params = setup
    rng = np.random.default_rng(8)
    img = rng.integers(0, 256, (256, 256, 3), dtype=np.uint8)
    out = str(tmp_path / "seg.png")
    cm = prediction.keypts_predict(model=model, inp=img, out_fname=out)
    assert cm.shape == (264, 264) and cm.dtype == np.int64
    ref_cm, pr = fcn_ref.prediction_ref(img, params, 68)
    diff = cm != ref_cm
    if diff.any():
        srt = np.sort(pr.reshape(264, 264, 68), axis=-1)
        assert (srt[..., -1] - srt[..., -2])[diff].max() < 2e-6
    assert diff.mean() < 1e-3
    import os
    assert os.path.getsize(out) > 0


def test_checkpoint_files_are_found_loaded_and_run(setup, tmp_path):
    """what the trainer leaves on disk (training.py:187-200: `<ckpt>.<epoch>` weight files and a
    `<ckpt>_config.json` WITHOUT input_height/input_width) -> `keypts_predict(checkpoints_path=...)`
    (prediction.py:116-133,166-170) -> class map on the device.  The newest epoch must be the one that runs."""
    import json
    from flm_amd.weights import save_weights, synth_fcn8_weights
    prediction, _, params = setup
    ckpt = str(tmp_path / "fcn8_run")
    old = synth_fcn8_weights(68, seed=99)
    save_weights(ckpt + ".00003.npz", old)            # an older epoch with other weights
    save_weights(ckpt + ".00012.npz", params)         # the newest by NUMERIC suffix (12 > 3; a string sort would agree,
    save_weights(ckpt + ".00007.npz", old)            #  a directory-order pick would not)
    with open(ckpt + "_config.json", "w") as f:       # training.py:195-200: no input dims
        json.dump({"model_class": "fcn_8", "n_classes": 68, "output_height": 264, "output_width": 264}, f)
    assert prediction.find_latest_checkpoint(ckpt).endswith(".00012.npz")
    rng = np.random.default_rng(77)
    img = rng.integers(0, 256, (256, 256, 3), dtype=np.uint8)
    cm = prediction.keypts_predict(inp=img, checkpoints_path=ckpt)
    assert cm.shape == (264, 264) and cm.dtype == np.int64
    ref_cm, pr = fcn_ref.prediction_ref(img, params, 68)
    diff = cm != ref_cm
    if diff.any():
        srt = np.sort(pr.reshape(264, 264, 68), axis=-1)
        assert (srt[..., -1] - srt[..., -2])[diff].max() < 2e-6
    assert diff.mean() < 1e-3
    old_cm, _ = fcn_ref.prediction_ref(img, old, 68)
    assert (cm != old_cm).mean() > 0.5               # not the stale epoch's weights
    model = prediction.model_from_checkpoint_path(ckpt)
    assert (model.input_height, model.input_width, model.n_classes, model.model_name) == (256, 256, 68, "fcn_8")
    # an image that is not at model size goes through the resize of get_image_array (generator.py:53) first
    big = rng.integers(0, 256, (300, 420, 3), dtype=np.uint8)
    cm2 = prediction.keypts_predict(model=model, inp=big)
    small = warp_ref.resize_u8_ref(big, 256, 256)
    ref2, pr2 = fcn_ref.prediction_ref(small, params, 68)
    d2 = cm2 != ref2
    if d2.any():
        srt = np.sort(pr2.reshape(264, 264, 68), axis=-1)
        assert (srt[..., -1] - srt[..., -2])[d2].max() < 2e-6
    with pytest.raises(AssertionError):
        prediction.model_from_checkpoint_path(str(tmp_path / "nothing_here"))


def test_video_predict_frame_loop(setup):
    """prediction.py:99-113 with the camera and the window replaced by arguments: every frame's faces in one batch,
    marks drawn into the frame, frames without faces pass through, the sink's False ends the loop ('q')."""
    prediction, model, _ = setup
    rng = np.random.default_rng(78)
    frames = [rng.integers(0, 200, (480, 640, 3), dtype=np.uint8) for _ in range(4)]
    rects = [[[100, 100, 260, 300], [350, 120, 500, 330]], [], [[200, 150, 400, 420]], [[10, 10, 100, 100]]]
    seen = []

    def detector(img):
        return rects[len(seen)]

    def sink(img, marks):
        seen.append((img.copy(), marks))
        return len(seen) < 3          # stop after the third frame

    originals = [f.copy() for f in frames]
    n = prediction.video_predict(detector, model, frames=frames, on_frame=sink)
    assert n == 3 and len(seen) == 3
    assert seen[0][1].shape == (2, 68, 2) and seen[1][1].shape == (0, 68, 2) and seen[2][1].shape == (1, 68, 2)
    assert np.array_equal(frames[1], originals[1]) and np.array_equal(frames[3], originals[3])   # no faces / not reached
    for k in (0, 2):
        exp = prediction.detect_marks_batch(originals[k], model, rects[k])
        assert np.array_equal(seen[k][1], exp)
        x, y = int(exp[0, 0, 0]), int(exp[0, 0, 1])
        if 2 <= x < 638 and 2 <= y < 478:
            assert frames[k][y, x].tolist() == [0, 255, 0]        # draw_marks' default colour at the first landmark
        assert (frames[k] != originals[k]).any()
    assert prediction.detect_marks_batch(frames[0], model, []).shape == (0, 68, 2)
    with pytest.raises(ValueError):
        prediction.video_predict(detector, model)


def test_empty_and_ragged_batches(setup):
    """Edge cases: an empty batch launches nothing; a batch that is not a multiple of any tile size
    (N = 3) and the single-face batch agree with each other face by face."""
    prediction, model, params = setup
    from flm_amd.utils import metrics
    empty = torch.empty((0, 256, 256, 3), dtype=torch.uint8, device="cuda")
    assert tuple(model.forward_device(empty, "probs").shape) == (0, 264 * 264, 68)
    assert tuple(model.forward_device(empty, "landmarks", n_points=4).shape) == (0, 68, 2)
    assert model.predict(np.zeros((0, 256, 256, 3), np.float32)).shape == (0, 264 * 264, 68)
    assert metrics.transfer_target(np.zeros((0, 8, 8, 3), np.float32)).shape == (0, 6)
    rng = np.random.default_rng(9)
    crops = torch.from_numpy(rng.integers(0, 256, (3, 256, 256, 3), dtype=np.uint8)).cuda()
    a = model.forward_device(crops, "landmarks", n_points=4).clone()
    for i in range(3):   # the same face alone gives bit-identical landmarks (no cross-face coupling)
        b = model.forward_device(crops[i:i + 1].contiguous(), "landmarks", n_points=4)
        assert torch.equal(a[i], b[0])
    # beyond 4 faces fc6 / fc7 stop splitting K: a different fp32 summation order, same landmarks within the bar
    six = torch.cat([crops, crops], 0).contiguous()
    c = model.forward_device(six, "landmarks", n_points=4)
    assert torch.equal(c[:3], c[3:])
    assert (c[:3] - a).abs().max().item() < 1e-3


def test_batches_beyond_the_launch_limit_are_sliced(setup):
    """`max_batch` (32-bit offsets inside the kernels) is honoured by slicing: same results as one launch
    (4 faces: whole and slices both run fc6 / fc7 with split K, the same summation order)."""
    prediction, model, params = setup
    rng = np.random.default_rng(10)
    crops = torch.from_numpy(rng.integers(0, 256, (4, 256, 256, 3), dtype=np.uint8)).cuda()
    whole = model.forward_device(crops, "landmarks", n_points=4).clone()
    assert model.max_batch >= 512
    saved = model.max_batch
    try:
        model.max_batch = 2
        sliced = model.forward_device(crops, "landmarks", n_points=4)
        cm = model.forward_device(crops, "classmap")
    finally:
        model.max_batch = saved
    assert torch.equal(whole, sliced)
    assert tuple(cm.shape) == (4, 264, 264)


def test_captured_pipeline_equals_eager_sequence(setup):
    """graphs.CapturedPipeline replays the landmark + alignment launch sequence as one HIP graph: same kernels on the
    same buffers, so landmarks, matrices and aligned crops equal the eager calls bit for bit, call after call."""
    from flm_amd import alignment, graphs
    _, model, _ = setup
    rng = np.random.default_rng(61)
    for n in (1, 5):
        pipe = graphs.CapturedPipeline(model, n, n_points=4)
        for rep in range(3):
            crops = torch.from_numpy(rng.integers(0, 256, (n, 256, 256, 3), dtype=np.uint8)).cuda()
            lm, aligned, m = [t.clone() for t in pipe(crops)]
            lm_e = model.forward_device(crops, "landmarks", n_points=4)
            al_e, m_e = alignment.align_device(crops, lm_e, pipe.template, 256, 256, pipe.scale)
            assert torch.equal(lm, lm_e) and torch.equal(m, m_e) and torch.equal(aligned, al_e), (n, rep)
        with pytest.raises(ValueError):
            pipe(torch.zeros((n + 1, 256, 256, 3), dtype=torch.uint8, device="cuda"))


def test_captured_pipelines_survive_workspace_cache_eviction(setup):
    """One pipe per face count on ONE model (the stream caller's use: 1-16 faces per frame) plus eager calls at other
    batch sizes push the model's workspace cache through several evictions; every graph owns its workspace, so the first
    pipe still replays onto live memory and equals the eager result."""
    from flm_amd import alignment, graphs
    _, model, _ = setup
    rng = np.random.default_rng(62)
    crops = torch.from_numpy(rng.integers(0, 256, (8, 256, 256, 3), dtype=np.uint8)).cuda()
    pipes = {n: graphs.CapturedPipeline(model, n, n_points=4) for n in (1, 2, 3, 4, 5, 6, 7)}
    for n in (8, 3, 6, 1, 5, 2, 7):                      # eager batch sizes: more keys than the cache holds
        model.forward_device(crops[:n].contiguous(), "landmarks", n_points=4)
        model.forward_device(crops[:n].contiguous(), "classmap")
    assert len(model._ws) <= model._ws_cap
    filler = [torch.full((1 << 22,), 0xAB, dtype=torch.uint8, device="cuda") for _ in range(64)]   # reuse freed blocks
    for n in (1, 4, 7):
        x = crops[:n].contiguous()
        lm, aligned, m = [t.clone() for t in pipes[n](x)]
        lm_e = model.forward_device(x, "landmarks", n_points=4)
        al_e, m_e = alignment.align_device(x, lm_e, pipes[n].template, 256, 256, pipes[n].scale)
        assert torch.equal(lm, lm_e) and torch.equal(m, m_e) and torch.equal(aligned, al_e), n
    assert all(int(f[0]) == 0xAB and int(f[-1]) == 0xAB for f in filler)     # nothing scribbled over other tensors


def test_bf16_faces_do_not_depend_on_their_batch(setup):
    """The bf16 stack splits K for fc6 / fc7 / enc4 / enc5 at a few faces (dense (chunk, tap) slices): one to four faces
    share every bracket, so a face's landmarks are the same bits alone, in a ragged batch of three, and in slices."""
    from flm_amd.networks import LANDMARKS_MODELS
    _, _, params = setup
    model = LANDMARKS_MODELS["fcn_8"](68, input_height=256, input_width=256, dtype="bf16")
    model.load_weights(params)
    rng = np.random.default_rng(12)
    crops = torch.from_numpy(rng.integers(0, 256, (4, 256, 256, 3), dtype=np.uint8)).cuda()
    three = model.forward_device(crops[:3].contiguous(), "landmarks", n_points=4).clone()
    for i in range(3):
        one = model.forward_device(crops[i:i + 1].contiguous(), "landmarks", n_points=4)
        assert torch.equal(three[i], one[0]), i
    whole = model.forward_device(crops, "landmarks", n_points=4).clone()
    saved = model.max_batch
    try:
        model.max_batch = 2
        sliced = model.forward_device(crops, "landmarks", n_points=4)
    finally:
        model.max_batch = saved
    assert torch.equal(whole, sliced) and torch.equal(whole[:3], three)


@pytest.mark.gpu
def test_crop_frames_device_equals_per_frame_crops(setup):
    """Several frames' faces as one batch (one box upload, launches into slices of the batch) == the per-frame crops
    concatenated; a frame without faces contributes nothing."""
    from flm_amd import prediction
    rng = np.random.default_rng(21)
    dev = torch.device("cuda", 0)
    frames = [torch.from_numpy(rng.integers(0, 256, (270, 480, 3), dtype=np.uint8)).to(dev) for _ in range(4)]
    faces = [[(10, 20, 110, 130), (200, 40, 330, 160)], [], [(5, 5, 60, 70)], [(300, 100, 470, 260), (0, 0, 90, 90), (50, 150, 140, 250)]]
    crops, boxes = prediction.crop_frames_device(frames, faces, 64, 64)
    assert crops.shape == (6, 64, 64, 3) and [len(b) for b in boxes] == [2, 0, 1, 3]
    ref = torch.cat([prediction.crop_faces_device(fr, prediction.face_boxes(fc), 64, 64) for fr, fc in zip(frames, faces) if fc], 0)
    assert torch.equal(crops, ref)
    # the same frames as one ring tensor: one launch, slots named explicitly (and out of order)
    ring = torch.stack([frames[3], frames[0], frames[2], frames[1]], 0)
    crops_r, _ = prediction.crop_frames_device(ring, faces, 64, 64, frame_index=[1, 3, 2, 0])
    assert torch.equal(crops_r, ref)
    with pytest.raises(ValueError):
        prediction.crop_frames_device(ring, faces, 64, 64, frame_index=[0, 1, 2, 4])
    empty, _ = prediction.crop_frames_device(frames[:1], [[]], 64, 64)
    assert empty.shape == (0, 64, 64, 3)
    with pytest.raises(ValueError):
        prediction.crop_faces_device(frames[0], [(0, 0, 10, 10)], 64, 64, out=torch.empty((2, 64, 64, 3), dtype=torch.uint8, device=dev))
