"""Helper of tests/test_gpu_multirank.py::test_rccl_branch_world_size_one (run as a child process: a process group per
pytest process would outlive the test).

Executes, on ONE GPU, every RCCL call the N-rank path makes - `init_process_group("nccl", device_id=...)`, the weight-blob
broadcast on the `_CudaBlob` view (cassnat_asr_public_amd.dist.broadcast_weights) and the `all_gather_into_tensor` branch of
`all_gather_records` through `DecodePipelines.decode(gather=True)` - in a world of size 1, and checks the hypotheses against
the non-distributed pipelines.  Reference fan-out this replaces: egs/librispeech/run_hubert.sh:94-116.
Prints one JSON line.
"""
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))


def main():
    import numpy as np
    import torch
    import torch.distributed as dist

    from conftest import config2_b8_case
    from cassnat_asr_public_amd import dist as cdist
    from cassnat_asr_public_amd.models.cassnat import make_model
    from cassnat_asr_public_amd.pipeline import DecodePipelines

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    assert dist.get_backend() == "nccl"

    args, state, feats, sizes = config2_b8_case()
    args.hip_precision = sys.argv[1] if len(sys.argv) > 1 else "bf16"
    model = make_model(80, args).cuda(0)
    with torch.no_grad():
        for k, p in model.named_parameters():
            p.copy_(torch.from_numpy(state[k]))
    B, T, _ = feats.shape
    f, s = torch.from_numpy(feats).cuda(), torch.from_numpy(sizes).cuda()
    batches = [(f, s, k) for k in range(5)]

    with DecodePipelines(model, 2, B, T, coalesce=2) as plain:
        want = [(h, sc.tolist()) for _, h, sc in plain.decode(batches, args, sos=1)]

    info = {}

    def receive(eng):
        t0 = time.perf_counter()
        info["blob_bytes"] = cdist.broadcast_weights(eng, src=0)  # ncclBroadcast on the zero-copy view of the packed blob
        info["weight_broadcast_ms"] = round((time.perf_counter() - t0) * 1e3, 3)

    called = {"into_tensor": 0}
    real = dist.all_gather_into_tensor

    def counting(out, inp, group=None, **kw):
        called["into_tensor"] += 1
        return real(out, inp, group=group, **kw)

    dist.all_gather_into_tensor = counting
    with DecodePipelines(model, 2, B, T, with_weights=True, after_engine=receive, coalesce=2) as pipes:
        got = [(h, sc.tolist()) for _, h, sc in pipes.decode(batches, args, sos=1, gather=True)]
    dist.all_gather_into_tensor = real
    assert called["into_tensor"] >= 1, "the RCCL all-gather branch did not run"
    assert len(got) == len(want) == 5
    for (h1, s1), (h2, s2) in zip(got, want):
        assert h1 == h2, "hypotheses differ between the gathered and the plain path"
        assert np.array_equal(np.asarray(s1), np.asarray(s2))
    info.update(ok=True, backend=dist.get_backend(), all_gather_into_tensor_calls=called["into_tensor"], steps=len(got),
                utterances=len(got) * B)
    print(json.dumps(info), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
