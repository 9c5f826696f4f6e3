"""Input side of the Trainer loop on the GPU: batches arrive on the device one iteration ahead.

The reference's loop (torchgan Trainer.train driven at src/histopathology_gan.py:298-314) hands every loss plugin the
loader's HOST batch, and each ``train_ops`` moves what it needs itself -- ``real_inputs['image'].to(device)`` in the D-loss
and in the penalty plugin (src/wgan_loss.py:221-222, :352-353), the RNA rows in all three (:96, :223, :353): two blocking
50 MB copies and three small ones per iteration, each on the compute stream between two train_ops.  At the reference's
speed that is noise; at 11.5 ms per iteration it is 15-20 % of the step.

``DevicePrefetcher(loader, device)`` iterates the same loader and yields the same batch structure (dict / tuple / list /
tensor, anything else passed through) with every tensor already resident on ``device``: the copy of batch k + 1 is issued
on a side stream BEFORE batch k is handed out, from pinned memory (the loader's own ``pin_memory=True`` thread, else pinned
here), so it runs under batch k's kernels; the consumer's stream waits on the copy's event, and the tensors are recorded on
it so the caching allocator does not recycle them early.  The plugins' own ``.to(device)`` calls then find the tensors in
place and return them unchanged.  On a CPU device it is the identity wrapper.
"""
from __future__ import annotations

import torch


def _map(obj, fn):
    if torch.is_tensor(obj):
        return fn(obj)
    if isinstance(obj, dict):
        return type(obj)((k, _map(v, fn)) for k, v in obj.items())
    if isinstance(obj, tuple) and hasattr(obj, "_fields"):         # namedtuple
        return type(obj)(*(_map(v, fn) for v in obj))
    if isinstance(obj, (list, tuple)):
        return type(obj)(_map(v, fn) for v in obj)
    return obj


class DevicePrefetcher:
    def __init__(self, loader, device):
        self.loader = loader
        self.device = torch.device(device)
        self.batch_size = getattr(loader, "batch_size", None)
        self._stream = None

    def __len__(self):
        return len(self.loader)

    def _stage(self, batch):
        """Issue the host -> device copies of one batch on the side stream; returns (device batch, event)."""
        side = self._stream

        def move(t):
            if t.device == self.device:
                return t
            if t.device.type == "cpu" and not t.is_pinned():
                t = t.pin_memory()
            return t.to(self.device, non_blocking=True)
        with torch.cuda.stream(side):
            out = _map(batch, move)
            ev = torch.cuda.Event()
            ev.record(side)
        return out, ev

    def __iter__(self):
        if self.device.type != "cuda":
            yield from self.loader
            return
        if self._stream is None:
            self._stream = torch.cuda.Stream(self.device)
        it = iter(self.loader)
        try:
            staged = self._stage(next(it))
        except StopIteration:
            return
        while staged is not None:
            cur, ev = staged
            try:
                host = next(it)
            except StopIteration:
                host = None
            # batch k + 1 goes out on the side stream before batch k is consumed: its copy runs under batch k's kernels
            staged = self._stage(host) if host is not None else None
            main = torch.cuda.current_stream(self.device)
            main.wait_event(ev)
            _map(cur, lambda t: (t.record_stream(main), t)[1] if t.is_cuda else t)
            yield cur
