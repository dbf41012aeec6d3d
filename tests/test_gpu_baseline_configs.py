"""BASELINE.json configs[1] and configs[2] at their full sizes, through the C ABI.

configs[1] -- batch 64, 256x256, fp32 (the correctness gate): every intermediate, the probabilities, the
class map and the top-4 / top-25 / all-pixel landmarks of ALL 64 faces against oracle/fcn_ref.py evaluated in
float32 (layer bars) and float64 (landmark yardstick).  At this batch the launch shapes are the bench's: fc6 and fc7
without split-K, multi-face position-major fc6 tiles, every encoder layer at full tile counts.

configs[2] -- batch 512, bf16 operands: the 256x256 tiles at full occupancy, the five-classes-per-wave key merge
(cand_merge_kernel<5>, n >= 128 faces), the candidate path against the materialised decode bit for bit, all 512
faces against the fp32 HIP path and 16 faces from the first / middle / last tiles against the fp32 oracle.

Landmark bar (north_star): coordinates within 1e-4 px of the reference, NME <= 1e-4.  The yardstick is the float64
evaluation of the oracle pushed through the reference's decode arithmetic (float32 map, float32 hsum, float64 index
sums).  How the bar is applied, everything printed:
  * all-pixel centroid (utils/metrics.py:58-64): hard 1e-4 px on every (face, class) pair.
  * a top-n centroid SELECTS pixels.  A pair is "determined" when the float64 map separates its n-th and (n+1)-th
    largest values by more than GAP_REL (relative) = twice the bar on the float32 rounding of a probability, so that
    every float32 evaluation of the network makes the same selection; determined pairs must select the float64 pixels.
    Undetermined pairs (printed and asserted < 1 %: a property of the synthetic maps, not of the kernels) must still be
    a valid top-n of the float64 map within that rounding.
  * determined pairs, top-4 (what every call of the reference decodes, utils/metrics.py:98) and top-25: hard 1e-4 px on
    EVERY pair, NME <= 1e-4.  With random weights the four selected pixels lie ~90 px apart, so the centroid moves by
    (spread x relative error of the probabilities): the bar is met because the fp32 implicit GEMMs sum in two levels
    (csrc/flm_igemm.hip: chains of 32 + K/32 roundings; 1.4-3.7e-7 relative RMS per layer where one fmaf chain of
    K = 576..12544 left 4e-7..1.3e-6 and 3 of 4,352 pairs at 1.0-1.31e-4 px; tools/diag_landmark_error.py).  The float32
    CPU oracle's own error on the same pairs is printed beside it.  DESIGN.md section 2 has the budget.
"""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

GAP_REL = 2e-5      # "determined": relative gap between the n-th and (n+1)-th largest float64 probabilities
ROUND_REL = 1e-5    # bar on the float32 rounding of a probability (relative to the n-th value; measured 8.3e-6 at worst)
PX = 1e-4           # north_star: landmark coordinates within 1e-4 px, NME <= 1e-4


@pytest.fixture(scope="module")
def flm():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    import flm_amd
    from flm_amd import _lib
    _lib.load()
    return flm_amd


@pytest.fixture(scope="module")
def weights68():
    from flm_amd.weights import synth_fcn8_weights
    return synth_fcn8_weights(68, seed=2)


def _rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def topn_of_maps(maps, n):
    """maps [HW, C] -> (idx [n, C] ascending by (value, flat index), n-th largest value [C], (n+1)-th largest [C]).
    Same selection rule as oracle/decode_ref.py (stable argsort, last n), without sorting all HW values."""
    hw, c = maps.shape
    k = min(hw, n + 16)
    part = np.argpartition(maps, hw - k, axis=0)[hw - k:]              # [k, C] unordered top-k
    vals = np.take_along_axis(maps, part, axis=0)
    order = np.lexsort((part, vals), axis=0)                           # by value, then by flat index
    part = np.take_along_axis(part, order, axis=0)
    vals = np.take_along_axis(vals, order, axis=0)
    # a tie that reaches below the k kept values would make the partition's choice arbitrary: not on these maps
    assert (vals[0] < vals[k - n]).all() or k == hw
    return part[k - n:], vals[k - n], vals[k - n - 1]


def centroid_ref(maps, idx):
    """utils/metrics.py:69-80 for already selected pixels, in the reference's arithmetic: the map is float32 (what
    model.predict returns), `hsum` a sequential float32 sum in ascending order, the index sums float64 (as
    oracle/decode_ref.py; checked against it below).  A float32 hsum alone moves a coordinate by up to 264 * 2^-24 =
    1.6e-5 px, so the yardstick applies the same decode to the float32-rounded float64 probabilities: what is left
    is the forward's error only."""
    w = 264
    v = np.take_along_axis(maps, idx, axis=0).astype(np.float32)
    ys, xs = np.divmod(idx, w)
    hsum = np.zeros(v.shape[1], np.float32)
    i0 = np.zeros(v.shape[1], np.float64)
    i1 = np.zeros(v.shape[1], np.float64)
    for r in range(v.shape[0]):
        hsum = hsum + v[r]
        i0 = i0 + ys[r] * v[r].astype(np.float64)
        i1 = i1 + xs[r] * v[r].astype(np.float64)
    return np.stack([i1 / hsum, i0 / hsum], axis=-1)


def test_config2_batch64_fp32_against_the_oracle(flm, weights68):
    from flm_amd.networks import LANDMARKS_MODELS
    from oracle import decode_ref, fcn_ref
    n, h, w, c = 64, 256, 256, 68
    crops = np.random.default_rng(1).integers(0, 256, (n, h, w, 3), dtype=np.uint8)   # SURVEY 8(d) config 2 seeds
    model = LANDMARKS_MODELS["fcn_8"](c, input_height=h, input_width=w)
    model.load_weights(weights68)
    xd = torch.from_numpy(crops).cuda()

    lm_hip = {npts: model.forward_device(xd, "landmarks", n_points=npts).cpu().numpy() for npts in (0, 4, 25)}
    cm_hip = model.forward_device(xd, "classmap").cpu().numpy()
    probs_hip = model.forward_device(xd, "probs")
    torch.cuda.synchronize()
    inter_hip = {k: model.intermediate(k, n, "probs").cpu().numpy()
                 for k in ("f1", "f2", "f3", "f4", "f5", "fc6", "fc7", "fuse4", "seg_feats")}
    probs_hip = probs_hip.cpu().numpy().reshape(n, 264 * 264, c)

    # landmarks by the candidate path are the decode of the materialised map (bit for bit), at this batch too
    hm = torch.from_numpy(probs_hip.reshape(n, 264, 264, c)).cuda()
    from flm_amd.utils.metrics import transfer_target
    for npts in (4, 25):
        dec = np.asarray(transfer_target(hm, thresh=0, n_points=npts))
        assert np.array_equal(dec.reshape(n, c, 2), lm_hip[npts]), npts
    del hm

    worst = {k: 0.0 for k in inter_hip}
    stats = {npts: dict(err=[], err32=[], decided=[], spread=[]) for npts in (4, 25)}
    all_err, probs_err, round_rel, cm_diff, cm_gap = [], 0.0, 0.0, 0, 0.0
    chunk = 8
    for lo in range(0, n, chunk):
        sl = slice(lo, lo + chunk)
        x_ref = np.stack([fcn_ref.get_image_array_ref(im) for im in crops[sl]])
        logits32, inter = fcn_ref.fcn8_logits_ref(x_ref, weights68, torch.float32, return_intermediates=True)
        for k in inter_hip:
            got = inter_hip[k][sl][..., : inter[k].shape[-1]]
            worst[k] = max(worst[k], _rel(got, inter[k]))
        p32 = torch.softmax(torch.from_numpy(logits32).reshape(chunk, -1, c), dim=-1).numpy()
        probs_err = max(probs_err, float(np.abs(probs_hip[sl] - p32).max()))
        # class map: equal to the oracle's argmax except where the oracle's top two are within rounding
        cm_ref = p32.reshape(chunk, 264, 264, c).argmax(-1)
        diff = cm_hip[sl] != cm_ref
        if diff.any():
            srt = np.sort(p32.reshape(chunk, 264, 264, c), axis=-1)
            cm_gap = max(cm_gap, float((srt[..., -1] - srt[..., -2])[diff].max()))
        cm_diff += int(diff.sum())
        p64 = fcn_ref.fcn8_predict_ref(x_ref, weights68, torch.float64)          # [chunk, HW, C]
        for i in range(chunk):
            f = lo + i
            m64, mh = p64[i], probs_hip[f]
            # all-pixel centroid (smooth in the map): hard 1e-4 against both evaluations of the oracle
            with np.errstate(all="ignore"):
                e64 = decode_ref.transfer_target_ref(m64.reshape(1, 264, 264, c), 0, 0).reshape(c, 2)
            all_err.append(np.abs(lm_hip[0][f] - e64).max())
            for npts in (4, 25):
                idx64, vn, vn1 = topn_of_maps(m64, npts)
                idxh, _, _ = topn_of_maps(mh, npts)
                e64n = centroid_ref(m64, idx64)
                if f == 0:   # the fast selection above IS the restated reference decode
                    with np.errstate(all="ignore"):
                        chk = decode_ref.transfer_target_ref(m64.astype(np.float32).reshape(1, 264, 264, c), 0, npts)
                    same_sel = (vn.astype(np.float32) > vn1.astype(np.float32))   # float32 rounding may tie the n-th place
                    assert np.array_equal(chk.reshape(c, 2)[same_sel], e64n[same_sel])
                decided = (vn - vn1) / vn > GAP_REL
                # measured float32 rounding of the probabilities the selection looked at
                sel64 = np.take_along_axis(m64, idxh, axis=0)
                selh = np.take_along_axis(mh, idxh, axis=0)
                round_rel = max(round_rel, float((np.abs(selh - sel64).max(0) / vn).max()))
                # every pair, determined or not: the HIP selection is a valid top-n of the float64 map within rounding
                assert (sel64 >= vn * (1 - ROUND_REL)).all(), (f, npts)
                # determined pairs select the same pixels
                same = (np.sort(idxh, axis=0) == np.sort(idx64, axis=0)).all(0)
                assert same[decided].all(), (f, npts)
                stats[npts]["err"].append(np.abs(lm_hip[npts][f] - e64n))
                idx32, _, _ = topn_of_maps(p32[i], npts)          # the float32 CPU oracle through the same decode
                stats[npts]["err32"].append(np.abs(centroid_ref(p32[i], idx32) - e64n))
                stats[npts]["decided"].append(decided)
                xs = (idx64 % 264).astype(np.float64)
                stats[npts]["spread"].append(xs.max(0) - xs.min(0))
        del p64, p32, logits32, inter

    print("config 2 (64 faces fp32): intermediates rel err", {k: "%.2g" % v for k, v in worst.items()})
    print("probabilities max-abs err %.3g; measured relative rounding at the selected pixels %.3g" % (probs_err, round_rel))
    print("all-pixel centroid: max err %.3g px" % max(all_err))
    summary = {}
    for npts in (4, 25):
        err = np.stack(stats[npts]["err"])                 # [64, 68, 2]
        decided = np.stack(stats[npts]["decided"])         # [64, 68]
        excl = 1.0 - decided.mean()
        nme = np.linalg.norm(err, axis=-1)[decided].mean() / 256.0
        e_pair = err.max(-1)
        e_dec = e_pair[decided].max()
        e_und = e_pair[~decided].max() if (~decided).any() else 0.0
        over = int((e_pair[decided] > PX).sum())
        p999 = float(np.quantile(e_pair[decided], 0.999))
        e32 = np.stack(stats[npts]["err32"]).max(-1)[decided]
        print("top-%d: %d pairs, excluded (n-th/(n+1)-th gap <= %.0e) %.3f %%; determined: max err %.3g px, p99.9 %.3g, "
              "pairs over 1e-4: %d, NME %.3g (float32 CPU oracle on the same pairs: max %.3g, p99.9 %.3g, over 1e-4: %d); "
              "undetermined: max err %.3g px; median x-spread of the selected pixels %.0f px"
              % (npts, decided.size, GAP_REL, 100 * excl, e_dec, p999, over, nme, e32.max(), np.quantile(e32, 0.999),
                 int((e32 > PX).sum()), e_und, np.median(np.stack(stats[npts]["spread"]))))
        summary[npts] = (excl, nme, e_dec, over, p999, int(decided.sum()))
    for k, v in worst.items():
        assert v < 2e-5, (k, v)
    assert probs_err <= 1e-5
    assert round_rel < ROUND_REL
    assert cm_diff / cm_hip.size < 1e-3 and cm_gap < 2e-6, (cm_diff, cm_gap)
    assert max(all_err) <= PX, max(all_err)
    for npts in (4, 25):
        excl, nme, e_dec, over, p999, n_dec = summary[npts]
        assert excl < 0.01, excl
        assert nme <= PX
        assert p999 <= PX, (npts, p999)
        assert e_dec <= PX and over == 0, (npts, e_dec, over)     # the north_star bar, on every determined pair


@pytest.mark.parametrize("wseed,xseed", [(5, 11), (9, 23)])
def test_top4_bar_holds_for_other_weights_and_crops(flm, wseed, xseed):
    """The 1e-4 px bar on the as-shipped top-4 decode is not a property of one seed: other synthetic weight sets and
    crops, 16 faces each (1,088 pairs), every determined pair within 1e-4 px of the float64 oracle."""
    from flm_amd.networks import LANDMARKS_MODELS
    from flm_amd.weights import synth_fcn8_weights
    from oracle import decode_ref, fcn_ref
    n, c = 16, 68
    w = synth_fcn8_weights(c, seed=wseed)
    crops = np.random.default_rng(xseed).integers(0, 256, (n, 256, 256, 3), dtype=np.uint8)
    model = LANDMARKS_MODELS["fcn_8"](c, input_height=256, input_width=256)
    model.load_weights(w)
    lm = model.forward_device(torch.from_numpy(crops).cuda(), "landmarks", n_points=4).cpu().numpy()
    worst, undet = 0.0, 0
    for lo in range(0, n, 8):
        x_ref = np.stack([fcn_ref.get_image_array_ref(im) for im in crops[lo:lo + 8]])
        p64 = fcn_ref.fcn8_predict_ref(x_ref, w, torch.float64)
        for i in range(8):
            det = decode_ref.topn_gap_rel(p64[i], 4) > GAP_REL
            with np.errstate(all="ignore"):
                ref = decode_ref.transfer_target_ref(p64[i].astype(np.float32).reshape(1, 264, 264, c), 0, 4).reshape(c, 2)
            e = np.abs(lm[lo + i] - ref).max(-1)
            worst = max(worst, float(e[det].max()))
            undet += int((~det).sum())
    print("weights seed %d, crops seed %d: top-4 max err %.3g px over the determined pairs (%d undetermined of %d)"
          % (wseed, xseed, worst, undet, n * c))
    assert worst <= PX and undet <= 0.01 * n * c


def test_config2_faces_do_not_depend_on_the_batch_beyond_the_split_k_brackets(flm, weights68):
    """A face's bits are the same in a batch of 20 and in the batch of 64 (both beyond the split-K brackets, which end
    at 16 faces): tap skipping in fc6's position-major tiles drops only products with zero padding."""
    from flm_amd.networks import LANDMARKS_MODELS
    crops = np.random.default_rng(1).integers(0, 256, (64, 256, 256, 3), dtype=np.uint8)
    model = LANDMARKS_MODELS["fcn_8"](68, input_height=256, input_width=256)
    model.load_weights(weights68)
    xd = torch.from_numpy(crops).cuda()
    lm64 = model.forward_device(xd, "landmarks", n_points=4).cpu().numpy()
    p64 = model.forward_device(xd, "probs")[40:44].cpu().numpy()
    sub = xd[30:50].contiguous()
    lm20 = model.forward_device(sub, "landmarks", n_points=4).cpu().numpy()
    p20 = model.forward_device(sub, "probs")[10:14].cpu().numpy()
    assert np.array_equal(p64, p20)
    assert np.array_equal(lm64[30:50], lm20)


def test_config3_batch512_bf16(flm, weights68):
    from flm_amd.networks import LANDMARKS_MODELS
    from oracle import decode_ref, fcn_ref
    n, h, w, c = 512, 256, 256, 68
    crops = np.random.default_rng(3).integers(0, 256, (n, h, w, 3), dtype=np.uint8)
    model = LANDMARKS_MODELS["fcn_8"](c, input_height=h, input_width=w, dtype="bf16")
    model.load_weights(weights68)
    xd = torch.from_numpy(crops).cuda()

    # (1) candidate path (sampling launch, thresholds, keys, five-classes-per-wave merge) == materialised decode
    for npts in (4, 25):
        got = model.forward_device(xd, "landmarks", n_points=npts).cpu().numpy()
        ref = model.forward_device(xd, "landmarks", n_points=npts, opts=dict(landmark_candidates=0)).cpu().numpy()
        model._ws.clear()      # 10 GB of materialised probabilities: release before the next size
        assert np.array_equal(got, ref), (npts, np.abs(got - ref).max())
    lm4 = got if npts == 4 else model.forward_device(xd, "landmarks", n_points=4).cpu().numpy()

    # (2) 16 faces spread over the first, middle and last tiles against the fp32 oracle (bf16 bars)
    model.forward_device(xd, "classmap")
    torch.cuda.synchronize()
    faces = list(range(0, 6)) + list(range(253, 258)) + list(range(507, 512))
    x_ref = np.stack([fcn_ref.get_image_array_ref(crops[f]) for f in faces])
    inter = {}
    for lo in range(0, len(faces), 8):
        _, it = fcn_ref.fcn8_logits_ref(x_ref[lo:lo + 8], weights68, torch.float32, return_intermediates=True)
        for k, v in it.items():
            inter.setdefault(k, []).append(v)
    inter = {k: np.concatenate(v) for k, v in inter.items()}
    for name, tol in (("f1", 1e-2), ("f2", 2e-2), ("f3", 2e-2), ("f4", 3e-2), ("f5", 3e-2), ("fc6", 4e-2),
                      ("fc7", 4e-2), ("fuse4", 5e-2), ("seg_feats", 5e-2)):
        got = model.intermediate(name, n, "classmap")[faces].cpu().numpy()[..., : inter[name].shape[-1]]
        assert _rel(got, inter[name]) < tol, (name, _rel(got, inter[name]))
    probs_ref = np.concatenate([fcn_ref.fcn8_predict_ref(x_ref[lo:lo + 8], weights68) for lo in range(0, len(faces), 8)])
    lm0 = model.forward_device(xd, "landmarks", n_points=0).cpu().numpy()
    with np.errstate(all="ignore"):
        exp0 = decode_ref.transfer_target_ref(probs_ref.reshape(len(faces), 264, 264, c), 0, 0).reshape(len(faces), c, 2)
    err = np.linalg.norm(lm0[faces] - exp0, axis=-1)
    print("config 3 (512 faces bf16): all-pixel landmarks of 16 faces vs the fp32 oracle: max %.3g px, NME %.3g"
          % (err.max(), err.mean() / 256))
    assert err.max() < 0.5
    cm = model.forward_device(xd[faces].contiguous(), "classmap").cpu().numpy()
    assert (cm == probs_ref.reshape(len(faces), 264, 264, c).argmax(-1)).mean() > 0.9

    # (3) all 512 faces against the fp32 HIP path (same crops): all-pixel centroid
    m32 = LANDMARKS_MODELS["fcn_8"](c, input_height=h, input_width=w)
    m32.load_weights(weights68)
    lm0_32 = m32.forward_device(xd, "landmarks", n_points=0).cpu().numpy()
    d = np.linalg.norm(lm0 - lm0_32, axis=-1)
    print("all 512 faces, bf16 vs fp32 HIP, all-pixel centroid: NME %.3g, max %.3g px" % (d.mean() / 256, d.max()))
    assert d.mean() / 256 < 2e-4 and d.max() < 0.25
    # top-4 on random-weight maps: a bf16 rounding moves pixels in and out of the selection, so what is gated is the rate
    # of (near-)identical selections, with a floor two points under the measured rate, and that every landmark is a point
    # of the grid
    lm4_32 = m32.forward_device(xd, "landmarks", n_points=4).cpu().numpy()
    close = np.abs(lm4 - lm4_32).max(-1) < 0.5
    print("top-4: %.1f %% of the 512x68 landmarks within 0.5 px of the fp32 path" % (100 * close.mean()))
    assert close.mean() >= 0.899            # measured 91.9 %; the floor is that minus 2 points
    assert (lm4 >= 0).all() and lm4[..., 0].max() <= 263 and lm4[..., 1].max() <= 263
