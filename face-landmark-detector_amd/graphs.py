"""One HIP graph for the whole landmark + alignment launch sequence of a fixed batch shape.

A step of the path is ~30 launches through the C ABI (flm_fcn_forward's conv stack, the candidate selection, the
similarity fit, the warp).  At the batch sizes the stream caller produces (1-16 faces per frame,
prediction.py:99-110) those kernels run for microseconds each and the step is bounded by launch dispatch, not by the
GPU: the sequence is captured once (stream capture of the very same C-ABI calls -- every launch already goes to the
caller's stream, the workspace is pre-sized, nothing synchronises or allocates inside) and replayed as one graph
launch per step.  Results are those of the eager sequence bit for bit (same kernels, same order, same buffers)."""
from __future__ import annotations

from . import _lib, alignment


class CapturedPipeline:
    """`lm, aligned, m = pipe(crops)` for crops [n,H,W,3] uint8 (BGR) on the GPU; `n` is fixed at construction.

    The returned tensors are the graph's static outputs: they are overwritten by the next call (clone to keep)."""

    def __init__(self, model, n, n_points=4, thresh=0.0, out_hw=None, template=None, warmup=3):
        import torch
        _lib.require_gpu()
        if n < 1:
            raise ValueError("CapturedPipeline needs a batch of at least one face")
        if n > model.max_batch:
            raise ValueError("batch %d exceeds the model's single-launch limit %d" % (n, model.max_batch))
        self.model, self.n, self.n_points, self.thresh = model, int(n), int(n_points), float(thresh)
        h, w = model.input_height, model.input_width
        self.out_hw = tuple(out_hw) if out_hw else (h, w)
        dev = torch.device("cuda", torch.cuda.current_device())
        tm = template if template is not None else alignment.canonical_template(model.n_classes, *self.out_hw)
        self.template = torch.as_tensor(tm, dtype=torch.float64).to(dev)
        self.scale = (model.input_width / model.output_width, model.input_height / model.output_height)
        self.crops = torch.zeros((self.n, h, w, 3), dtype=torch.uint8, device=dev)
        # the graph bakes raw pointers into its kernel arguments: it owns its workspace (the model's cache may evict)
        self.workspace = model.new_workspace(self.n, "landmarks", self.n_points)
        # warm-up on a side stream: first-launch work (function attributes, workspace allocation) must not be captured
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                self._sequence()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.landmarks, self.aligned, self.m = self._sequence()

    def _sequence(self):
        lm = self.model.forward_device(self.crops, "landmarks", n_points=self.n_points, thresh=self.thresh,
                                       workspace=self.workspace)
        aligned, m = alignment.align_device(self.crops, lm, self.template, self.out_hw[0], self.out_hw[1], self.scale)
        return lm, aligned, m

    def __call__(self, crops):
        if tuple(crops.shape) != tuple(self.crops.shape) or crops.dtype != self.crops.dtype:
            raise ValueError("CapturedPipeline was captured for %s uint8 crops" % (tuple(self.crops.shape),))
        self.crops.copy_(crops)
        self.graph.replay()
        return self.landmarks, self.aligned, self.m
