"""Command-line surface of the recogniser: same flags as the reference's DecodeParser
(src/utils/parser.py:32-54) plus the engine switches of this build."""
import argparse


class DecodeParser(object):
    def __init__(self, description="Decode argument parser"):
        p = argparse.ArgumentParser(description=description)
        p.add_argument("--test_config")
        p.add_argument("--lm_config")
        p.add_argument("--data_path")
        p.add_argument("--text_label", default="", type=str, help="text label")
        p.add_argument("--task", default="cassnat", type=str, help="'cassnat' (NAT) or 'art' (autoregressive transformer): the tasks on the accelerated path")
        p.add_argument("--batch_size", default=32, type=int)
        p.add_argument("--load_data_workers", default=1, type=int)
        p.add_argument("--resume_model", default="", type=str, help="checkpoint with a 'model_state' dict")
        p.add_argument("--result_file", default="", type=str)
        p.add_argument("--print_freq", default=100, type=int)
        p.add_argument("--rnnlm", type=str, default=None)
        p.add_argument("--rank_model", type=str, default="lm")
        p.add_argument("--lm_weight", type=float, default=0.0)
        p.add_argument("--seed", default=1, type=int)
        # engine switches (not in the reference)
        p.add_argument("--hip_precision", default="bf16", choices=["bf16", "fp32", "fp8", "bf16x3"],
                       help="bf16 MFMA (throughput) or exact-f32 MFMA (parity with the reference CPU path)")
        p.add_argument("--hip_max_frames", default=4096, type=int, help="workspace size in input frames")
        p.add_argument("--hip_pipelines", default=3, type=int,
                       help="decode pipelines per GPU for greedy decoding of a test set (1 = batch after batch)")
        p.add_argument("--hip_coalesce", default=3, type=int,
                       help="equal-shaped batches a decode pipeline may take through one engine pass (hypotheses per batch unchanged)")
        p.add_argument("--hip_dist_backend", default="nccl", choices=["nccl", "gloo"],
                       help="torch.distributed backend under torch.distributed.run (nccl = RCCL over xGMI; gloo: rehearsal of the "
                            "N-rank path, also with several ranks on one GPU)")
        self.parser = p

    def get_args(self, argv=None):
        return self.parser.parse_args(argv)
