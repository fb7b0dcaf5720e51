"""Device -> host copies that do not wait for later work on the compute stream.

`tensor.cpu()` / `.tolist()` enqueue the copy on the CURRENT stream and block until it is done, i.e. until everything
enqueued before the call has finished.  The drop-in generator enqueues timestep k+1 before it looks at the results of
timestep k (PNG frames, loss scalars: /root/reference/cgd/cgd.py:180-186,234-238,266-270), so those reads go through a
side stream that only depends on the event recorded right after the value was produced, into pinned memory.
"""
import torch as th

_SIDE = {}


def side_stream(device):
    device = th.device(device)
    key = device.index if device.index is not None else th.cuda.current_device()
    s = _SIDE.get(key)
    if s is None:
        s = _SIDE[key] = th.cuda.Stream(device=key)
    return s


class HostCopy:
    """Asynchronous snapshot of a device tensor; `get()` waits for this copy only and returns the pinned host tensor."""

    def __init__(self, t):
        t = t.detach()
        with th.cuda.device(t.device):  # events and streams of the tensor's GPU, whatever the caller's current device is
            produced = th.cuda.Event()
            produced.record()  # on the current (compute) stream: everything `t` depends on has been enqueued
            side = side_stream(t.device)
            self.host = th.empty(t.shape, dtype=t.dtype, pin_memory=True)
            with th.cuda.stream(side):
                side.wait_event(produced)
                self.host.copy_(t, non_blocking=True)
                self.done = th.cuda.Event(blocking=True)  # get() sleeps, it does not spin: one driver process per GPU shares the host
                self.done.record()
            t.record_stream(side)  # the caching allocator must not hand t's block out before the side-stream copy has read it
        self._src = t

    def get(self):
        self.done.synchronize()
        self._src = None
        return self.host
