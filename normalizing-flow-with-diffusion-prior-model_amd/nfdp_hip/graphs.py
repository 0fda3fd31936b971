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
import contextlib
import threading
from typing import Callable, Sequence

import torch


# ---------------------------------------------------------------------------------------------------------------- side branches under capture
# ROCm 7.2's hipStreamEndCapture dies with a host segfault when the capture contains a NESTED fork: a stream forked from a stream that
# was itself forked from the capture's origin stream.  Pure torch reproduces it - origin -> s (wait_stream) -> side (wait_stream), one
# elementwise kernel each, joins in reverse order; `tools/probe_nested_capture.py`, profiles/r05_b_nested_capture.txt: every nested
# variant exits with SIGSEGV inside capture_end, re-forking a joined stream and two non-nested forks are fine.  Round 4 met it as "two
# Glow L5 passes on two streams in one graph": each pass forks a side stream for its 1x1-conv log-dets, which is a nested fork when
# the pass itself runs on a forked stream.
# The package therefore forks a side stream under capture ONLY from the capture's origin stream.  HIP has no query for "is this the
# origin stream", so the origin must be known: torch.cuda.graph()'s default capture stream is (that covers `with torch.cuda.graph(g):`,
# capture_step and Graphed), a stream passed as torch.cuda.graph(g, stream=s) is declared with `with capture_origin(s):`.  On any
# other capturing stream the side work is issued inline on the calling stream - correct, just not overlapped.
_declared = threading.local()


@contextlib.contextmanager
def capture_origin(stream: "torch.cuda.Stream"):
    """Declare `stream` as the origin stream of the capture made inside the block (needed only with torch.cuda.graph(g, stream=stream))."""
    held = getattr(_declared, "streams", None)
    if held is None:
        held = _declared.streams = []
    held.append(stream.cuda_stream)
    try:
        yield stream
    finally:
        held.remove(stream.cuda_stream)


def side_branch_allowed(device=None) -> bool:
    """May the caller fork a side stream from the current stream?  Always outside a capture; under capture only when the current stream
    is the (known) origin of the capture - see the note above."""
    if not torch.cuda.is_available() or not torch.cuda.is_current_stream_capturing():
        return True
    cur = torch.cuda.current_stream(device).cuda_stream
    if cur in getattr(_declared, "streams", ()):
        return True
    default = getattr(torch.cuda.graph, "default_capture_stream", None)
    return default is not None and default.cuda_stream == cur


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


class CapturedStep:
    """A whole training (or evaluation) step captured once as a hipGraph: `step = capture_step(fn); step()` replays it and returns the
    captured outputs (tensors written in place by every replay).  See capture_step."""

    def __init__(self, graph, outputs, mutates_params: bool):
        self.graph, self.outputs, self._mutates = graph, outputs, mutates_params

    def __call__(self):
        self.graph.replay()
        if self._mutates:
            from . import bump_param_epoch
            bump_param_epoch()          # the replay updated parameters behind torch's version counters: weight-layout caches follow
        return self.outputs

    replay = __call__


def capture_step(fn: Callable, warmup: int = 3, mutates_params: bool = True, capture_error_mode: str = None,
                 warmup_fn: Callable = None) -> CapturedStep:
    """Capture `fn()` - a complete step with no arguments that reads its inputs from fixed tensors, e.g.

        def fn():
            loss = sum(prior(latents)); opt.zero_grad(); loss.backward(); opt.step(); return loss

    - as ONE hipGraph, the one way this package captures steps (bench.py, tools/bench_diffusion_train.py):
      * `warmup` eager calls first, on a SIDE stream (torch's recipe for whole-step capture: lazily created state - packed weight
        images, workspaces, autograd's accumulation nodes, optimizer state - is then built outside the capture and not bound to the
        legacy default stream), joined back before the capture;
      * refuses (RuntimeError) to nest inside a running capture, to capture with warmup < 1, and a step that syncs with the host
        fails inside torch.cuda.graph with torch's own error rather than hanging;
      * with a process group of more than one rank the capture runs in "thread_local" error mode (RCCL's watchdog thread may call
        into HIP meanwhile); pass capture_error_mode to override.
    The warm-up must leave the state every replay will start from.  For a training step that is "right after an optimizer update":
    the kernel-layout weight caches are then stale, so the capture records the re-packing launches and every replay re-packs the
    weights the previous replay updated.  A forward+backward captured after a warm-up WITHOUT an optimizer step finds valid caches,
    records no re-packing and replays on the weights of capture time for ever (seen as a diverging loss in the two-graph
    data-parallel form of bench.py --train before warmup_fn existed).
    warmup_fn: what the warm-up calls run when `fn` alone is not such a step - e.g. fn = optimizer.step under data parallelism,
    whose eager form needs the gradient all-reduce in front of it: pass the whole training step.
    fn must use static shapes and draw its randomness on the device (GaussianDiffusion.forward does; self_condition=True does not
    and raises).  The returned object replays the graph; its outputs alias fn's return value.

    Round 2 recorded a host segfault in hipStreamEndCapture for a step captured after eager steps on the legacy default stream
    and blamed autograd's AccumulateGrad nodes.  tools/probe_capture_abort.py (profiles/r03_probe_capture_abort.txt: 15 child
    processes - Glow and diffusion-prior steps, default- and side-stream warm-up, direct / AccumulateGrad / torch.optim.Adam
    gradients, including the original configuration) does NOT reproduce it on the current code: every combination captures and
    replays.  The side-stream warm-up is kept because it is torch's documented contract, not because of that crash."""
    if not torch.cuda.is_available():
        raise RuntimeError("capture_step needs the GPU (there is no CPU fallback)")
    if torch.cuda.is_current_stream_capturing():
        raise RuntimeError("capture_step: a capture is already running on this stream (captures do not nest)")
    if warmup < 1:
        raise RuntimeError("capture_step: at least one warm-up call is needed (lazily built state must exist before the capture)")
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(warmup):
            (warmup_fn or fn)()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    if capture_error_mode is None:
        import torch.distributed as dist
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
        capture_error_mode = "thread_local" if multi else "global"
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, capture_error_mode=capture_error_mode):
        out = fn()
    # The capture RECORDED launches without running them, but the Python-side bookkeeping ran: weight-layout caches whose re-packing was
    # just recorded are marked up to date although their buffers still hold the previous weights.  Eager code that runs before the first
    # replay (e.g. the warm-up of a second capture) must not trust them.
    from . import bump_param_epoch
    bump_param_epoch()
    return CapturedStep(graph, out, mutates_params)
