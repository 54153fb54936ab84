"""hipGraph capture of a fixed-shape forward (inference / evaluation) or of a whole training step.

A PointMamba forward is ~900 kernel launches; at evaluation batch sizes (the reference tests with B = 32,
tools/runner_finetune.py:427-467) the GPU finishes them faster than Python can enqueue them.  Every op of this
package only enqueues on the stream it is handed and allocates through PyTorch's caching allocator, so the
whole forward -- HIP kernels, hipBLASLt GEMMs and the side-stream eigen-ordering with its two event joins --
captures into one hipGraph and replays with a single launch.
"""
from __future__ import annotations

import torch


class GraphedForward:
    """Capture ``module(*example_inputs)`` once (no grad), then ``__call__`` copies new inputs into the static
    buffers, replays the graph and returns the static output tensor(s)."""

    def __init__(self, module, *example_inputs, warmup: int = 3):
        if not all(t.is_cuda for t in example_inputs):
            raise RuntimeError("GraphedForward needs CUDA/ROCm tensors")
        self.module = module
        self.static_in = [t.clone() for t in example_inputs]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(warmup):           # lazy initialisations (library handles, LDS attributes) happen here
                module(*self.static_in)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.static_out = module(*self.static_in)

    @torch.no_grad()
    def __call__(self, *inputs):
        for dst, src in zip(self.static_in, inputs):
            dst.copy_(src)
        self.graph.replay()
        return self.static_out


class GraphedTrainStep:
    """Capture one whole training step -- zero_grad, forward, loss, backward, optional gradient clipping, optimizer
    step -- into one hipGraph (single process; DDP steps keep the eager path).

        step = GraphedTrainStep(lambda pts, gt: loss_fn(model(pts), gt), optimizer, (pts, gt), params=..., clip=10.0)
        loss = step(new_pts, new_gt)        # copies the inputs into the static buffers, replays, returns the loss

    The small-batch configurations (MAE pre-training at 64 clouds, part segmentation at 16) are launch-bound in
    eager mode: ~2 000 launches per step against 15-20 ms of GPU work.  Requirements: an optimizer built with
    ``capturable=True``, no host synchronisation inside ``loss_fn`` (no ``.item()``, no shape that depends on data).
    Random ops (DropPath, Dropout, the HLT tie-break) draw from the graph-safe Philox stream.
    """

    def __init__(self, loss_fn, optimizer, example_inputs, params=None, clip=None, warmup: int = 3):
        if not all(t.is_cuda for t in example_inputs):
            raise RuntimeError("GraphedTrainStep needs CUDA/ROCm tensors")
        self.loss_fn, self.opt, self.clip = loss_fn, optimizer, clip
        self.params = [p for g in optimizer.param_groups for p in g["params"]] if params is None else list(params)
        self.static_in = [t.clone() for t in example_inputs]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):               # optimizer state, library handles and autotuned paths settle here
                self._eager()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.static_loss = self._eager()

    def _eager(self):
        self.opt.zero_grad(set_to_none=True)
        loss = self.loss_fn(*self.static_in)
        loss.backward()
        if self.clip is not None:
            torch.nn.utils.clip_grad_norm_(self.params, self.clip, foreach=True)
        self.opt.step()
        return loss.detach()

    def __call__(self, *inputs):
        for dst, src in zip(self.static_in, inputs):
            if src is not dst:
                dst.copy_(src)
        self.graph.replay()
        return self.static_loss
