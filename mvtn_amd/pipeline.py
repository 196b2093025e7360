"""Inference throughput: independent batches in flight on several HIP streams.

A forward is two dependent launches (pre-pass, rasteriser); on one stream the next forward's pre-pass — one workgroup
per view, ~15 us on a fraction of the CUs — starts only when the previous rasteriser's last workgroups have drained
(its final ~20 us run on a partly empty GPU).  Forwards that do not depend on each other (rendering a dataset, evaluation)
can be issued round-robin on two streams: the pre-pass of one fills the tail of the other.  Measured at C2:
114 -> 103 us per forward (1.68 -> 1.87 M views/s, `tools/time_two_streams.py`); a third stream adds nothing.
Training steps depend on each other through the view parameters and stay on one stream.
"""
import torch


class StreamPipeline:
    """Round-robin issue of independent calls on `n_streams` side streams of the current device.

        pipe = StreamPipeline(2)
        outs = [pipe.submit(renderer, packed_k, None, azim_k, elev_k, dist_k) for k in range(n)]   # runs under no_grad
        pipe.join()          # the caller's current stream now waits for everything submitted

    Every result tensor is produced on a side stream: use it on the caller's stream only after `join()`; results are
    registered with the caller's stream (`record_stream`) so the allocator keeps them alive.  `inputs_ready=True`
    skips the per-call dependency on the caller's stream (one barrier packet per call, ~6 us at C2) when the inputs
    were produced long before (e.g. resident meshes and view tensors made once)."""

    def __init__(self, n_streams=2, device=None):
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.streams = [torch.cuda.Stream(self.device) for _ in range(int(n_streams))]
        self._events = [torch.cuda.Event() for _ in self.streams]
        self._k = 0

    def submit(self, fn, *args, inputs_ready=False, **kwargs):
        cur = torch.cuda.current_stream(self.device)
        side = self.streams[self._k % len(self.streams)]
        self._k += 1
        if not inputs_ready:
            side.wait_stream(cur)                   # inputs prepared on the caller's stream are ready
        with torch.cuda.stream(side), torch.no_grad():
            out = fn(*args, **kwargs)
        for t in _tensors(out):
            t.record_stream(cur)
        return out

    def join(self):
        """The caller's current stream waits for everything submitted so far (one event per side stream)."""
        cur = torch.cuda.current_stream(self.device)
        for side, ev in zip(self.streams, self._events):
            ev.record(side)
            cur.wait_event(ev)


def _tensors(x):
    if torch.is_tensor(x):
        yield x
    elif isinstance(x, (tuple, list)):
        for y in x:
            yield from _tensors(y)
    elif hasattr(x, "R") and hasattr(x, "T"):      # Cameras
        yield x.R
        yield x.T
