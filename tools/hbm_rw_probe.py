"""tools/hbm_rw_probe.py -- what the device sustains for write-only, read-only and copy streams (torch kernels, 2 GiB buffers that
do not fit the 256 MiB Infinity Cache): context for the roofline of the write-heavy period warp."""
import torch

def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e-3

N = 2 << 30
x = torch.empty(N // 4, dtype=torch.float32, device="cuda")
y = torch.empty_like(x)
x.fill_(1.0)
w = t(lambda: x.fill_(2.0))
r = t(lambda: x.sum())
c = t(lambda: y.copy_(x))
z = t(lambda: x.zero_())
print(f"write-only fill_ {N / w / 1e12:.2f} TB/s   zero_ {N / z / 1e12:.2f} TB/s   read-only sum {N / r / 1e12:.2f} TB/s   copy {2 * N / c / 1e12:.2f} TB/s (read + write)")
