"""Isolated duration of conv1's bordered bf16 image kernels at the benchmark's ten-batch width (320 x 1000 x 80 -> 3.36 GB):
the matrix-core form (cn_op_conv1_bordered) against torch's fill of the same buffer (the write roof of this box)."""
import ctypes as C
import sys, os
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cassnat_asr_public_amd import hip  # noqa: E402

B, T, F, Cc = 320, 1000, 80, 256
T1, F1 = (T - 1) // 2 + 1, (F - 1) // 2 + 1
L = hip.lib()
dev = torch.device("cuda:0")
x = torch.randn(B, T, F, device=dev)
w = torch.randn(9, Cc, device=dev) / 3
b = torch.randn(Cc, device=dev) * 0.1
img = torch.empty(B, T1 + 2, F1 + 2, Cc, dtype=torch.bfloat16, device=dev)
p = lambda t: C.c_void_p(t.data_ptr())
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


gb = img.numel() * 2 / 1e9
t = timed(lambda: hip.check(L.cn_op_conv1_bordered(p(x), p(w), p(b), p(img), B, T, F, Cc, s)))
print(f"conv1 matrix-core form, bf16 image {gb:.2f} GB: {t:.3f} ms = {gb / t:.2f} TB/s")
t = timed(lambda: img.fill_(1.0))
print(f"torch fill of the image: {t:.3f} ms = {gb / t:.2f} TB/s")
