"""Phase stamps of the row-chain kernel (workgroup 0, s_memtime cycles and the shader clock it saw) for one encoder-layer
launch of M rows, on a build of the library with extra flags (e.g. -DCH_EXP_NO_DMA: the blocks without their LDS-DMA requests -
wrong results, right timing of everything else):
    python tools/chain_stamps.py [M] [-D...]          (GPU box)"""
import ctypes as C
import os
import sys

os.environ.setdefault("CASSNAT_CHAIN_STAMPS", "1")  # 2: also skip the tail's stores
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cassnat_asr_public_amd import build as B  # noqa: E402


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
    extra = sys.argv[2:]
    out = os.environ.get("TMPDIR", "/tmp")
    tag = "".join(c for c in "".join(extra) if c.isalnum())
    # the stamps are an experiment switch (csrc/common.h: cn_exp_env): an experiments build of the library, with the extra flags if
    # any (CASSNAT_CHAIN_STAMP_BLOCK chooses the workgroup that writes the stamps)
    lib = os.environ.get("CASSNAT_HIP_LIB") or B.experiments_lib(extra, tag="chain" + tag)
    L = C.CDLL(lib)
    L.cn_op_chain.argtypes = [C.c_void_p, C.c_void_p, C.c_int32] + [C.c_void_p] * 12 + [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                                                                       C.c_float, C.c_int32, C.c_void_p]
    d, dff, tail = 256, 2048, 768
    g = torch.Generator().manual_seed(1)
    rn = lambda *s: torch.randn(*s, generator=g)
    x = (rn((M + 31) // 32 * 32, d) * 2).cuda()  # (blocked layout in and out, as between the engine's layers: any values do)
    ctx = rn(M, d).to(torch.bfloat16).cuda()
    host = [rn(d, d) / 16, 0.1 * rn(d), 1 + 0.1 * rn(d), 0.1 * rn(d), rn(dff, d) / 16, 0.1 * rn(dff), rn(d, dff) / 45, 0.1 * rn(d), 1 + 0.1 * rn(d),
            0.1 * rn(d), rn(tail, d) / 16, 0.1 * rn(tail)]
    host = [t.contiguous() for t in host]
    outt = torch.empty((M + 31) // 32 * 32, tail, dtype=torch.bfloat16, device="cuda")
    p = lambda t: C.c_void_p(t.data_ptr())
    for _ in range(2):
        rc = L.cn_op_chain(p(x), p(ctx), d, *[p(t) for t in host], p(outt), tail, M, dff, tail, 1e-6, int(os.environ.get("CHAIN_X_MODE", "19")), None)  # 19: blocked x in and out + blocked tail output (the engine's form)
        assert rc == 0, rc
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
