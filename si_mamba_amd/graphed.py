"""hipGraph capture of a fixed-shape forward (inference / evaluation).

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
