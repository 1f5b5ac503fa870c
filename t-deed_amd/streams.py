"""Distinct HIP streams.

`torch.cuda.Stream()` does not create a stream: it hands out the next one of a pool of 32 per device (per priority) and
wraps around.  In a long-lived process two "different" stream objects are therefore sooner or later the same HIP stream.
That is harmless for plain ordering, but not for the fork/join structure of a captured forward: a sub-batch "forked" onto
the very stream that is being captured records a join event on that stream and makes it wait for itself inside the capture
-- the instantiated graph then crashed the host in hipGraphLaunch (found with a test sequence that happened to advance the
pool by the right amount).  `new_stream` returns a stream whose handle differs from the current stream and from the ones to
avoid; callers that fork additionally run a sub-batch inline when its stream turns out to be the launching one.
"""
import torch


def new_stream(device=None, avoid=()):
    taken = {s.cuda_stream for s in avoid if s is not None}
    taken.add(torch.cuda.current_stream(device).cuda_stream)
    for _ in range(64):
        s = torch.cuda.Stream(device=device)
        if s.cuda_stream not in taken:
            return s
    raise RuntimeError("no distinct HIP stream available from torch's stream pool")
