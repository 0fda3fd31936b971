"""hipGraph capture of a fixed-shape step (HIP graphs instead of a tracing compiler).

A forward + NLL step of a 3-level Glow is ~20 launches through the C ABI plus torch's glue, a training step ~1600; launched
one by one from Python the host costs about as much time as the GPU spends computing.  `Graphed` captures the callable once
(after warm-up runs on a side stream, as torch.cuda.graphs requires) and replays it with new inputs copied into the captured
input tensors.  The callable must be free of host synchronisation and use only static shapes - which holds for
Glow.transform / invert, GaussianPrior, the training step with nfdp_hip.optim.FusedAdam, and the diffusion steps.

A captured graph reads the kernel-layout weight copies that existed at capture time.  An inference graph therefore has to be
re-captured after the parameters change; a training-step graph re-packs its weights inside the graph (the warm-up step's
optimizer update invalidates every cache, so the packing launches are recorded) and can be replayed indefinitely.
"""
from typing import Callable, Sequence

import torch


class Graphed:
    """graphed = Graphed(fn, example_inputs);  out = graphed(*new_inputs)   (out aliases the captured output tensors)."""

    def __init__(self, fn: Callable, example_inputs: Sequence[torch.Tensor], warmup: int = 2, mutates_params: bool = False):
        """mutates_params: the captured callable updates parameters on the device (an optimizer step).  Every replay then bumps the
        parameter epoch so that kernel-layout weight caches used by eager code afterwards are rebuilt (torch's version counters do
        not see a replayed update)."""
        self._mutates = mutates_params
        if not torch.cuda.is_available():
            raise RuntimeError("Graphed needs the GPU (there is no CPU fallback)")
        self._static_in = [t.clone() for t in example_inputs]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                fn(*self._static_in)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self._graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._graph):
            self._out = fn(*self._static_in)

    def __call__(self, *inputs: torch.Tensor):
        if len(inputs) != len(self._static_in):
            raise ValueError(f"expected {len(self._static_in)} inputs, got {len(inputs)}")
        for dst, src in zip(self._static_in, inputs):
            if dst.shape != src.shape or dst.dtype != src.dtype:
                raise ValueError(f"static shape/dtype {tuple(dst.shape)}/{dst.dtype} != {tuple(src.shape)}/{src.dtype}")
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src, non_blocking=True)
        self._graph.replay()
        if self._mutates:
            from . import bump_param_epoch
            bump_param_epoch()
        return self._out
