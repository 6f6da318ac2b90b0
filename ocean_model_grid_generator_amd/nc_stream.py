"""Device -> NetCDF-3 file stream for write_nc's layout (OGG:773-829).

The classic format is a fixed header followed by each variable's data, contiguous and big-endian, so every band of every
field that sits in HBM after a pass maps to ONE contiguous byte range of the file.  A chunk of rows goes through
``ogg_bswap64_dev`` (8-byte reversal on the device: no host-side byte swap) into a slot of a small device staging ring, from
there with the copy engine into the matching slot of a pinned host ring, and a small pool of writer threads ``pwrite`` the
slot at its offset while the next chunks are swapped and copied.  (``OGG_NC_STAGE=host`` lets the kernel store straight into
the pinned slot instead -- no staging buffer, no copy -- but kernel stores cross PCIe no faster than the copy engine.)  Measured (profiles/r02_nc_write_sweep.json): 11 GB/s into a
fresh 1.2 GB file on /tmp whatever the staging mode, the slot size or the number of writer threads (1 to 16) -- buffered writes to
one file are serialised by the file system; the copy engine alone moves 56 GB/s.  The rings are allocated once per process (pinning memory is slow) and reused by later calls.

torch is used for the pinned allocation, events and the stream only.
"""
import os
import threading
from concurrent.futures import ThreadPoolExecutor

from . import _lib as L

_RING = {}   # (slots, slot_bytes, device or None) -> list of pinned / device uint8 tensors
_LOCK = threading.Lock()


def _ring(torch, slots, slot_bytes, device=None):
    with _LOCK:
        key = (slots, slot_bytes, str(device))
        if key not in _RING:
            if device is None:
                _RING[key] = [torch.empty(slot_bytes, dtype=torch.uint8, pin_memory=True) for _ in range(slots)]
            else:
                _RING[key] = [torch.empty(slot_bytes, dtype=torch.uint8, device=device) for _ in range(slots)]
        return _RING[key]


class DeviceToFile(object):
    """stream = DeviceToFile(fd, device); stream.put(tensor_2d, file_offset) ...; stream.finish()."""

    def __init__(self, fd, device, slot_bytes=None, slots=None, threads=None):
        import torch
        self.torch, self.fd = torch, fd
        self.device = torch.device(device)
        self.slot_bytes = int(slot_bytes or os.environ.get("OGG_NC_SLOT_BYTES", 16 << 20))
        slots = int(slots or os.environ.get("OGG_NC_SLOTS", 8))
        self.ring = _ring(torch, slots, self.slot_bytes)
        self.stage = None if os.environ.get("OGG_NC_STAGE", "device") == "host" else _ring(torch, slots, self.slot_bytes, self.device)
        self.busy = [None] * slots   # future of the pwrite that reads the slot
        self.pool = ThreadPoolExecutor(max_workers=int(threads or os.environ.get("OGG_NC_THREADS", 2)))
        self.k = 0
        self.bytes = 0

    def _write(self, slot, event, nbytes, offset):
        event.synchronize()   # the swap kernel of this slot has finished: its bytes are in host memory
        view = memoryview(self.ring[slot].numpy())[:nbytes]
        done = 0
        while done < nbytes:
            done += os.pwrite(self.fd, view[done:], offset + done)

    def put(self, t, offset):
        """t: contiguous fp64 device tensor (rows x cols); its big-endian image goes to file offset `offset`."""
        torch = self.torch
        assert t.is_contiguous() and t.dtype == torch.float64
        n = t.numel()
        per = self.slot_bytes // 8
        st = torch.cuda.current_stream(self.device)
        base = t.data_ptr()
        for k0 in range(0, n, per):
            cnt = min(per, n - k0)
            slot = self.k % len(self.ring)
            self.k += 1
            if self.busy[slot] is not None:
                self.busy[slot].result()   # the previous pwrite from this slot has finished
            if self.stage is None:
                L.call("ogg_bswap64_dev", cnt, base + 8 * k0, self.ring[slot].data_ptr(), st.cuda_stream)
            else:
                L.call("ogg_bswap64_dev", cnt, base + 8 * k0, self.stage[slot].data_ptr(), st.cuda_stream)
                self.ring[slot][: cnt * 8].copy_(self.stage[slot][: cnt * 8], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(st)
            self.busy[slot] = self.pool.submit(self._write, slot, ev, cnt * 8, offset + 8 * k0)
            self.bytes += cnt * 8

    def finish(self):
        """Wait for every pending write (re-raising the first failure) and stop the writer threads."""
        first = None
        for f in self.busy:
            if f is not None:
                try:
                    f.result()
                except Exception as exc:   # keep draining: the threads must be done before the caller closes the file
                    first = first or exc
        self.busy = [None] * len(self.ring)
        self.pool.shutdown(wait=True)
        if first is not None:
            raise first
