"""Phase stamps of the bf16 self-attention launch of the benchmark shape (the last workgroup of the grid: steady state, not
the launch's first burst) and the launch's duration, for B utterances:  CASSNAT_ATTN_STAMPS=1 python tools/attn_stamps.py [B]"""
import ctypes as C
import os
import sys

os.environ.setdefault("CASSNAT_ATTN_STAMPS", "1")
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cassnat_asr_public_amd import build as _B  # noqa: E402

# (an experiment switch: only the -DCASSNAT_EXPERIMENTS build of the library reads it)
os.environ.setdefault("CASSNAT_HIP_LIB", _B.experiments_lib())
from cassnat_asr_public_amd import hip  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 320
    H, L = 4, 250
    g = torch.Generator().manual_seed(0)
    q, k, v = (torch.randn(B * L, H * 64, generator=g).to(torch.bfloat16).cuda() for _ in range(3))
    o = torch.empty_like(q)
    km = torch.ones(B, L, dtype=torch.uint8, device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr())
    L_ = hip.lib()
    for it in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        hip.check(L_.cn_op_attention(hip.PRECISION["bf16"], p(q), H * 64, p(k), H * 64, p(v), H * 64, p(o), H * 64, B, H, L, L, p(km),
                                     None, None, 0, 0, 0.125, C.c_void_p(torch.cuda.current_stream().cuda_stream)))
        e1.record()
        torch.cuda.synchronize()
        print("launch %d: %.1f us for %d workgroups" % (it, e0.elapsed_time(e1) * 1e3, B * H), file=sys.stderr)


if __name__ == "__main__":
    main()
