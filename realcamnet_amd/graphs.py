"""HIP-graph replay of a whole forward (the codec at low batch is "a HIP-graph / persistent-kernel candidate").

Every op of the path enqueues on the current stream, allocates through the caching allocator and never syncs with the host (torch_ops.py), so a forward
is capturable.  At B = 8 the GPU is the bound and a graph buys nothing; at B = 1 the codec forward is ~760 launches of which most are 5-40 us at 72 x 120
(upstream models/raw2bit.py:1817-1846, the serial 5-slice loop): the host needs ~20 ms to enqueue what the GPU runs in ~11 ms.  `GraphedCall` captures the
callable once per input signature and replays it: no Python, no dispatcher, no ctypes per launch.  The two-stream forks of the slice loop
(ops.fork_join) are captured as graph branches.
"""
from __future__ import annotations

from typing import Callable, Dict, Tuple

import torch


def _map(obj, fn):
    if isinstance(obj, torch.Tensor):
        return fn(obj)
    if isinstance(obj, dict):
        return {k: _map(v, fn) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_map(v, fn) for v in obj)
    return obj


class GraphedCall:
    """`g = GraphedCall(fn)`; `out = g(*tensors)`: fn(*tensors) captured as a HIP graph per (shape, dtype, device) signature of the inputs and replayed.

    * inputs are copied into the capture's static buffers (device-to-device, on the current stream), then the graph is launched;
    * the result has fn's structure (tensor / tuple / dict) and refers to the graph's OWN output buffers: it is overwritten by the next call with the
      same signature -- `.clone()` what must outlive it;
    * fn must be capturable: inference ops of this package only (no host sync, no `.item()`, no allocation outside the caching allocator);
      parameters must not be re-assigned between calls (in-place updates are seen, re-packing is not: call `reset()` after load_state_dict / .to()).
    """

    def __init__(self, fn: Callable, warmup: int = 2):
        self.fn, self.warmup = fn, int(warmup)
        self._graphs: Dict[Tuple, Tuple] = {}

    def reset(self) -> None:
        self._graphs.clear()

    @staticmethod
    def _sig(tensors) -> Tuple:
        return tuple((tuple(t.shape), t.dtype, str(t.device)) for t in tensors)

    def _capture(self, tensors):
        if not all(t.is_cuda for t in tensors):
            raise RuntimeError("GraphedCall: inputs must be HIP-device tensors")
        static_in = [t.clone() for t in tensors]
        side = torch.cuda.Stream(device=static_in[0].device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():          # warm-up outside the capture: weight packing, launch attributes, side streams
            for _ in range(self.warmup):
                self.fn(*static_in)
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph), torch.no_grad():
            out = self.fn(*static_in)
        return graph, static_in, out

    def __call__(self, *tensors):
        sig = self._sig(tensors)
        hit = self._graphs.get(sig)
        if hit is None:
            hit = self._graphs[sig] = self._capture(tensors)
        graph, static_in, out = hit
        for s, t in zip(static_in, tensors):
            if s.data_ptr() != t.data_ptr():
                s.copy_(t, non_blocking=True)
        graph.replay()
        return out
