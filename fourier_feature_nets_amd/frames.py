"""Frame output that does not stall the renderer.

The reference writes each orbit frame synchronously -- render, copy to the host, encode, write
(orbit_video.py:83-93) -- so frames/sec is bounded by the sum of the four.  ``FrameSink`` takes
the finished uint8 frame as a DEVICE tensor and overlaps the rest with the next frames' kernels:
the device-to-host copy runs on a side HIP stream into a ring of pinned buffers, and PNG encoding
+ file writes run on a small thread pool (zlib releases the GIL).  ``submit`` only blocks when
every ring slot is still busy, i.e. when encoding is the bottleneck.
"""

import collections
from concurrent.futures import ThreadPoolExecutor
from typing import Optional

import numpy as np
import torch


def write_png(path: str, image: np.ndarray, compress_level: int = 3):
    from PIL import Image
    Image.fromarray(image).save(path, compress_level=compress_level)


class FrameSink:
    """Asynchronous PNG writer for device-resident (H,W,3) uint8 frames."""

    def __init__(self, slots: int = 4, workers: int = 4, compress_level: int = 3):
        self._slots = int(slots)
        self._pool = ThreadPoolExecutor(max_workers=int(workers))
        self._free = collections.deque()          # pinned host buffers ready for reuse
        self._busy = collections.deque()          # (future, pinned buffer) in submission order
        self._allocated = 0
        self._stream: Optional[torch.cuda.Stream] = None
        self._compress = int(compress_level)
        self.frames = 0

    def _buffer(self, like: torch.Tensor) -> torch.Tensor:
        while self._free:
            buf = self._free.popleft()
            if buf.shape == like.shape:
                return buf
            self._allocated -= 1                  # frame size changed: drop the old buffer
        if self._allocated >= self._slots and self._busy:   # ring full: wait for the oldest frame
            future, buf = self._busy.popleft()
            try:
                future.result()
            except BaseException:
                self._free.append(buf)            # the buffer stays in the ring when a write failed
                raise
            if buf.shape == like.shape:
                return buf
            self._allocated -= 1
        self._allocated += 1
        return torch.empty(like.shape, dtype=torch.uint8).pin_memory()

    def _reap(self):
        while self._busy and self._busy[0][0].done():
            future, buf = self._busy.popleft()
            self._free.append(buf)                # recycled whether or not the write succeeded
            future.result()                       # surfaces encoder errors

    def submit(self, image: torch.Tensor, path: str):
        """Queues ``image`` (uint8, on a GPU; produced on torch's current stream) for writing
        to ``path``.  Returns immediately unless the ring is full."""
        if image.dtype != torch.uint8 or not image.is_cuda:
            raise TypeError("FrameSink takes uint8 device tensors")
        self._reap()
        host = self._buffer(image)
        if self._stream is None or self._stream.device != image.device:
            self._stream = torch.cuda.Stream(device=image.device)
        produced = torch.cuda.Event()
        produced.record(torch.cuda.current_stream(image.device))
        copied = torch.cuda.Event()
        with torch.cuda.stream(self._stream):
            self._stream.wait_event(produced)
            host.copy_(image, non_blocking=True)
            copied.record(self._stream)
        image.record_stream(self._stream)         # the allocator must not recycle it early
        compress = self._compress

        def finish():
            copied.synchronize()
            write_png(path, host.numpy(), compress)

        self._busy.append((self._pool.submit(finish), host))
        self.frames += 1

    def drain(self):
        """Blocks until every submitted frame is on disk (or has failed): every future is
        joined and every buffer recycled before the FIRST failure is re-raised."""
        first_error = None
        while self._busy:
            future, buf = self._busy.popleft()
            self._free.append(buf)
            try:
                future.result()
            except BaseException as err:          # noqa: B902 -- re-raised below
                if first_error is None:
                    first_error = err
        if first_error is not None:
            raise first_error

    def close(self):
        try:
            self.drain()
        finally:
            self._pool.shutdown(wait=True)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False
