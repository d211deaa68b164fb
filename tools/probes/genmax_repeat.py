"""Timing probe: the fused generator tail launched back to back on the same operands (CASSNAT_GENMAX_REPEAT), bf16 and
split-bf16, at the benchmark's CTC shape.  Run under `rocprofv3 --kernel-trace`: the first dispatch sees the weight stream
cold (just uploaded), the following ones find it in L2 - the gap between the two is the stream's miss latency."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cassnat_asr_public_amd import build as _B  # noqa: E402

os.environ.setdefault("CASSNAT_GENMAX_REPEAT", "12")  # (an experiment switch: only the -DCASSNAT_EXPERIMENTS build reads it)
os.environ.setdefault("CASSNAT_HIP_LIB", _B.experiments_lib())
from cassnat_asr_public_amd import hip  # noqa: E402

M, V = int(os.environ.get("GM_M", 7968)), 5000
g = torch.Generator().manual_seed(1)
h = torch.randn(M, 256, generator=g).contiguous()
w = (torch.randn(V, 256, generator=g) / 16).contiguous()
b = (0.1 * torch.randn(V, generator=g)).contiguous()
arg = torch.zeros(M, dtype=torch.int32, device="cuda")
mlp = torch.zeros(M, dtype=torch.float32, device="cuda")
hd = h.to("cuda", torch.bfloat16)
hp = lambda t: C.c_void_p(t.data_ptr())
L = hip.lib()
for with_lse in (False, True):
    hip.check(L.cn_op_genmax(hp(hd), hp(w), hp(b), M, V, hp(arg), hp(mlp) if with_lse else None, None))
    hip.check(L.cn_op_genmax_x3(hp(h), hp(w), hp(b), M, V, hp(arg), hp(mlp) if with_lse else None, None, 0, 0, None, None))
torch.cuda.synchronize()
print("ok")
