"""Host-side visualisation: flm_amd.utils.plots against hand-worked arrays of the reference's
arithmetic (utils/plots.py:60-166).  cv2 is absent, so its glyphs and anti-aliasing are not pinned; sizes, colours,
swatch geometry, resize index rules and the disc shape are."""
import numpy as np
import pytest

import flm_amd  # noqa: F401
from flm_amd.utils import plots

COLORS = [(10, 20, 30), (200, 100, 50), (7, 8, 9)]


def test_colored_segmentation_image_is_the_class_colour_per_pixel():
    seg = np.array([[0, 1], [2, 1], [5, 0]])
    img = plots.get_colored_segmentation_image(seg, 3, colors=COLORS)
    assert img.dtype == np.float64 and img.shape == (3, 2, 3)
    assert img[0, 0].tolist() == [10, 20, 30] and img[1, 0].tolist() == [7, 8, 9] and img[0, 1].tolist() == [200, 100, 50]
    assert img[2, 0].tolist() == [0, 0, 0]            # class id beyond n_classes stays black (:66)


def test_legend_geometry_and_concat():
    names = ["jaw", "brow", "nose"]
    leg = plots.get_legends(names, colors=COLORS)
    assert leg.shape == (3 * 25 + 25, 125, 3) and leg.dtype == np.uint8        # :78-79
    for i, c in enumerate(COLORS):                                              # cv2.rectangle((100, 25i), (125, 25i+25), -1)
        assert (leg[25 * i + 1:25 * i + 25, 100:125] == np.array(c, np.uint8)).all()
    assert (leg[77:, 100:125] == 255).all()                                     # below the last swatch's closing row (75/76): white
    assert (leg[:, :5] == 255).all()                                            # left margin stays white
    text_zone = leg[:75, 5:100]
    assert (text_zone != 255).any()                                             # names were drawn...
    for i in range(3):                                                          # ...one per 25-row band, in black
        band = leg[25 * i:25 * i + 25, 5:100]
        assert (band != 255).any() and set(np.unique(band)) <= set(range(256))
    seg = np.full((40, 30, 3), 9, np.uint8)
    out = plots.concat_lenends(seg, leg)
    assert out.shape == (100, 155, 3) and out.dtype == np.uint8                # max height, widths add (:96-97)
    assert np.array_equal(out[:100, :125], leg) and (out[:40, 125:] == 9).all()
    assert (out[40:, 125:] == 255).all()                                        # fill = legend_img[0,0,0] (:99)


def test_visualize_keypoints_pipeline():
    seg = np.array([[0, 1], [2, 0]])
    inp = np.arange(4 * 6 * 3, dtype=np.uint8).reshape(4, 6, 3)
    out = plots.visualize_keypoints(seg, inp, n_classes=3, colors=COLORS)
    assert out.shape == (4, 6, 3)
    # INTER_NEAREST up-scaling 2x2 -> 4x6: src row = floor(y*2/4), src col = floor(x*2/6)
    assert out[0, 0].tolist() == [10, 20, 30] and out[0, 3].tolist() == [200, 100, 50] and out[3, 2].tolist() == [7, 8, 9]
    ov = plots.visualize_keypoints(seg, inp, n_classes=3, colors=COLORS, overlay_img=True)
    assert ov.dtype == np.uint8 and np.array_equal(ov, (inp / 2 + out / 2).astype("uint8"))     # :113
    full = plots.visualize_keypoints(seg, inp, n_classes=3, colors=COLORS, overlay_img=True, show_legends=True,
                                     class_names=["a", "b", "c"], pred_dim=(12, 8))
    assert full.shape == (100, 125 + 12, 3)                                     # legend 100 rows, picture 8x12 beside it
    with pytest.raises(AssertionError):
        plots.visualize_keypoints(seg, inp, n_classes=3, show_legends=True)     # class_names required (:142)
    assert plots.visualize_keypoints(seg).shape == (2, 2, 3)                    # pred_dim=None no longer raises
    # n_classes=None covers the largest class id too
    assert plots.visualize_keypoints(seg, colors=COLORS)[1, 0].tolist() == [7, 8, 9]


def test_linear_resize_of_the_input_image_is_the_fixed_point_spec():
    """`cv2.resize(inp_img, (w, h))` inside visualize_keypoints: same integers as the oracle's restatement of OpenCV's
    8-bit INTER_LINEAR (oracle/warp_ref.py), which the device kernel is tested against bit for bit."""
    from oracle import warp_ref
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    for h, w in ((20, 31), (74, 106), (37, 53), (64, 64)):
        assert np.array_equal(plots._resize_linear_u8(img, w, h), warp_ref.resize_u8_ref(img, h, w)), (h, w)
    even = rng.integers(0, 256, (40, 60, 3), dtype=np.uint8)
    assert np.array_equal(plots._resize_linear_u8(even, 30, 20), warp_ref.resize_u8_ref(even, 20, 30))   # 2x: area path


def test_draw_marks_is_cv2s_radius_2_disc():
    img = np.zeros((12, 12, 3), np.uint8)
    out = plots.draw_marks(img, np.array([[5, 6], [0, 0], [11, 3]], np.uint), color=(0, 255, 0))
    assert out is img                                                           # in place, like the reference
    g = img[:, :, 1] == 255
    disc = np.array([[0, 1, 1, 1, 0],
                     [1, 1, 1, 1, 1],
                     [1, 1, 1, 1, 1],
                     [1, 1, 1, 1, 1],
                     [0, 1, 1, 1, 0]], bool)
    assert np.array_equal(g[4:9, 3:8], disc)                                    # centred on (x=5, y=6)
    assert np.array_equal(g[0:3, 0:3], disc[2:, 2:])                            # clipped at the corner
    assert np.array_equal(g[1:6, 9:12], disc[:, :3])                            # clipped at the right edge
    assert g.sum() == 21 + 8 + 13 and (img[:, :, 0] == 0).all() and (img[:, :, 2] == 0).all()
