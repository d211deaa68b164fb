"""Does a working set below the 256 MB of Infinity Cache (MALL) stream faster than HBM on this box?
In-place read-modify-write (x.mul_) and write-only (x.fill_) over buffers of 32 MB .. 4 GB, GB/s of bytes touched."""
import time
import torch

dev = torch.device("cuda:0")
for mb in (32, 64, 96, 128, 160, 192, 256, 384, 512, 1024, 4096):
    n = mb * 1024 * 1024 // 2
    x = torch.ones(n, dtype=torch.bfloat16, device=dev)
    reps = max(4, 8192 // mb)
    for kind in ("rmw", "fill", "read"):
        f = (lambda: x.mul_(1.0)) if kind == "rmw" else (lambda: x.fill_(1.0)) if kind == "fill" else (lambda: x.max())
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            f()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        moved = mb * 1048576 * (2 if kind == "rmw" else 1)
        print(f"{mb:5d} MB {kind:5s} {dt * 1e6:9.1f} us  {moved / dt / 1e12:6.2f} TB/s", flush=True)
    del x
