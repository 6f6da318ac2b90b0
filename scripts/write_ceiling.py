"""Pure-write and copy bandwidth of the GPU as seen by simple kernels (torch fill_/copy_, ogg_fill_dev): the ceiling against
which the lat-lon kernel's write rate should be read."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ocean_model_grid_generator_amd import _lib as L  # noqa: E402

n = 1 << 27  # 1 GiB of doubles
a = torch.empty(n, dtype=torch.float64, device="cuda")
b = torch.empty(n, dtype=torch.float64, device="cuda")


def bench(fn, nbytes, label, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print("%-28s %8.3f ms  %7.1f GB/s" % (label, ms, nbytes / ms / 1e6))


st = torch.cuda.current_stream().cuda_stream
bench(lambda: a.fill_(1.5), n * 8, "torch fill_ (write 1 GiB)")
bench(lambda: a.zero_(), n * 8, "torch zero_ (write 1 GiB)")
bench(lambda: L.call("ogg_fill_dev", n, 1.5, a.data_ptr(), st), n * 8, "ogg_fill_dev (write 1 GiB)")
bench(lambda: b.copy_(a), n * 16, "torch copy_ (read+write 2 GiB)")
