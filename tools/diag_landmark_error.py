#!/usr/bin/env python3
"""Where the top-n landmark error of the fp32 forward comes from (developer tool; uses oracle/ as the checker).

For a few 256x256 faces: HIP logits and HIP probabilities against the float64 oracle.
  A. softmax64(HIP logits) -> reference decode:   error of the conv stack alone
  B. HIP probabilities      -> reference decode:   conv stack + the kernel's softmax
  C. float32 oracle         -> reference decode:   what another fp32 implementation of the same network gives
all against decode(float32(softmax64(oracle64 logits))).
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import flm_amd  # noqa: F401
    from flm_amd.networks import LANDMARKS_MODELS
    from flm_amd.weights import synth_fcn8_weights
    from oracle import fcn_ref
    from test_gpu_baseline_configs import centroid_ref, topn_of_maps
    n, c = int(sys.argv[1]) if len(sys.argv) > 1 else 8, 68
    for kv in filter(None, os.environ.get("KNOBS", "").split(",")):   # e.g. KNOBS=f32_two_level=0
        from flm_amd import _lib
        k, v = kv.split("=")
        _lib.check(_lib.load().flm_set_tuning(k.encode(), int(v)), "set_tuning")
        print("knob", k, "=", v)
    w = synth_fcn8_weights(c, seed=2)
    crops = np.random.default_rng(1).integers(0, 256, (n, 256, 256, 3), dtype=np.uint8)
    model = LANDMARKS_MODELS["fcn_8"](c, input_height=256, input_width=256)
    model.load_weights(w)
    xd = torch.from_numpy(crops).cuda()
    lg_hip = model.forward_device(xd, "logits").cpu().numpy().reshape(n, -1, c)
    pr_hip = model.forward_device(xd, "probs").cpu().numpy().reshape(n, -1, c)
    x_ref = np.stack([fcn_ref.get_image_array_ref(im) for im in crops])
    names = ("f1", "f2", "f3", "f4", "f5", "fc6", "fc7", "fuse4", "seg_feats")
    inter_hip = {k: model.intermediate(k, n, "probs").cpu().numpy() for k in names}
    _, i64 = fcn_ref.fcn8_logits_ref(x_ref[:2], w, torch.float64, return_intermediates=True)
    _, i32 = fcn_ref.fcn8_logits_ref(x_ref[:2], w, torch.float32, return_intermediates=True)
    rms = lambda a, b: float(np.sqrt(np.mean((a - b) ** 2)) / np.sqrt(np.mean(b ** 2)))
    print("relative RMS error against the float64 oracle, layer by layer (first two faces):")
    for k in names:
        h = inter_hip[k][:2][..., : i64[k].shape[-1]].astype(np.float64)
        print("  %-9s hip %.3g   float32 oracle %.3g" % (k, rms(h, i64[k]), rms(i32[k].astype(np.float64), i64[k])))
    # up3 alone: exact (float64) transposed conv of the HIP path's own seg_feats against the HIP logits
    seg = torch.from_numpy(inter_hip["seg_feats"][:2][..., :c].astype(np.float64)).permute(0, 3, 1, 2)
    lg_from_seg = fcn_ref._convT(seg, w["up3/kernel"], 8, torch.float64).permute(0, 2, 3, 1).reshape(2, -1, c).numpy()
    l64 = fcn_ref.fcn8_logits_ref(x_ref[:2], w, torch.float64).reshape(2, -1, c)
    print("logits: rms |hip - exact up3 of hip seg_feats| %.3g (up3's own rounding), rms |exact up3 of hip seg_feats - o64| %.3g "
          "(everything before it), rms |hip - o64| %.3g" % (np.sqrt(np.mean((lg_hip[:2] - lg_from_seg) ** 2)),
          np.sqrt(np.mean((lg_from_seg - l64) ** 2)), np.sqrt(np.mean((lg_hip[:2] - l64) ** 2))))
    for npts in (4, 25):
        ea, eb, ec, sm, worst = [], [], [], [], []
        for i in range(n):
            l64 = fcn_ref.fcn8_logits_ref(x_ref[i:i + 1], w, torch.float64).reshape(-1, c)
            l32 = fcn_ref.fcn8_logits_ref(x_ref[i:i + 1], w, torch.float32).reshape(-1, c)
            p64 = torch.softmax(torch.from_numpy(l64), -1).numpy()
            p32 = torch.softmax(torch.from_numpy(l32), -1).numpy()
            pa = torch.softmax(torch.from_numpy(lg_hip[i].astype(np.float64)), -1).numpy()
            idx, vn, vn1 = topn_of_maps(p64, npts)
            dec = (vn - vn1) / vn > 2e-5
            ref = centroid_ref(p64, idx)
            for lst, pm in ((ea, pa), (eb, pr_hip[i]), (ec, p32)):
                ii, _, _ = topn_of_maps(pm, npts)
                lst.append(np.abs(centroid_ref(pm, ii) - ref)[dec])
            ii, _, _ = topn_of_maps(pr_hip[i], npts)
            eb_full = np.abs(centroid_ref(pr_hip[i], ii) - ref).max(-1)
            ea_full = np.abs(centroid_ref(pa, topn_of_maps(pa, npts)[0]) - ref).max(-1)
            for cls in np.argsort(-np.where(dec, eb_full, 0))[:2]:
                sel = idx[:, cls]
                worst.append((eb_full[cls], i, int(cls), ea_full[cls],
                              (pr_hip[i][sel, cls] / p64[sel, cls] - 1).tolist(), (pa[sel, cls] / p64[sel, cls] - 1).tolist(),
                              (lg_hip[i][sel, cls] - l64[sel, cls]).tolist(), p64[sel, cls].tolist(),
                              [(int(t) // 264, int(t) % 264) for t in sel]))
            sm.append((np.abs(np.take_along_axis(pr_hip[i], ii, 0) - np.take_along_axis(pa, ii, 0)) /
                       np.take_along_axis(pa, ii, 0)).max())
            if npts == 4 and i == 0:
                print("logits: |hip - o64| max %.3g, |o32 - o64| max %.3g, |logit| max %.3g" %
                      (np.abs(lg_hip[i] - l64).max(), np.abs(l32 - l64).max(), np.abs(l64).max()))
        cat = lambda l: np.concatenate([e.ravel() for e in l])
        for name, l in (("A conv stack only (softmax64 of HIP logits)", ea), ("B HIP probabilities", eb),
                        ("C float32 oracle", ec)):
            e = cat(l)
            print("top-%d %-46s max %.3g px  p99 %.3g  median %.3g  over 1e-4: %d of %d" %
                  (npts, name, e.max(), np.quantile(e, 0.99), np.median(e), (e > 1e-4).sum(), e.size))
        print("top-%d kernel softmax vs softmax64(HIP logits), relative, at the selected pixels: max %.3g" % (npts, max(sm)))
        for e, i, cls, eaa, rb, ra, dl, pv, px in sorted(worst, reverse=True)[:6]:
            print("  worst: face %d class %d err B %.3g (A %.3g); rel err of the selected probs B %s | A %s; logit err %s; p %s; (y,x) %s"
                  % (i, cls, e, eaa, ["%.2g" % v for v in rb], ["%.2g" % v for v in ra], ["%.2g" % v for v in dl],
                     ["%.3g" % v for v in pv], px))


if __name__ == "__main__":
    main()
