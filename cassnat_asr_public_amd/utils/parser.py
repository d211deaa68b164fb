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
        p.add_argument("--hip_precision", default="bf16", choices=["bf16", "fp32", "fp8", "bf16x3", "fp16"],
                       help="fp16: the bf16 engine with half-precision MFMA operands (same speed, operand roundings 8x smaller, range +-65504); bf16 MFMA (throughput), bf16x3 (split-bf16: the reference's tolerance at MFMA speed), fp32 (exact-f32 MFMA), "
                            "fp8 (e4m3 encoder products of the NAT recogniser, BASELINE config 5; a ranking LM / AT model or the "
                            "autoregressive recogniser of --task art run bf16 under it)")
        p.add_argument("--hip_fp8_scope", default="all",
                       help="--hip_precision fp8: which encoder-side products take e4m3 operands - all, or conv2 / linear / "
                            "ffn[:first layer] joined by + (e.g. conv2+ffn:8); fewer products = fewer arg-max flips, less speed-up")
        p.add_argument("--hip_max_frames", default=4096, type=int, help="workspace size in input frames")
        p.add_argument("--hip_pipelines", default=2, type=int,
                       help="decode pipelines per GPU for greedy decoding of a test set (1 = batch after batch, the reference's loop)")
        p.add_argument("--hip_coalesce", default=10, type=int,
                       help="sizes a pipeline's workspace: the area of that many batches of batch_size x 1024 frames; consecutive "
                            "batches - of different frame counts too - share one engine pass while they fit it (every batch's "
                            "hypotheses and scores stay exactly those of a pass of its own)")
        p.add_argument("--hip_ragged", default=0.75, type=float,
                       help="batches share an engine pass while the shortest has at least this fraction of the longest one's frames "
                            "(1 = equal shapes only)")
        p.add_argument("--hip_bucket", default=0, type=int,
                       help="1: form the batches from the utterance list sorted by length (frame counts from <scp dir>/utt2num_frames "
                            "or the ark headers) instead of file order - less padding per batch, neighbours that merge well; the "
                            "result file stays in file order.  0 (default): the reference's batches")
        p.add_argument("--hip_packed_reader", default=1, type=int,
                       help="1 (default): the pipelined decoder reads the utterances' rows straight from the memory-mapped archives into "
                            "page-locked memory, a pass at a time, and pads / normalises on the device (float32 archives without splicing "
                            "or frame skipping; --load_data_workers then sets the number of copy threads); 0: the DataLoader's collated batches")
        p.add_argument("--hip_device_cmvn", default=1, type=int,
                       help="1 (default): the pipelined decoder applies the global CMVN on the device, behind the host-to-device copy "
                            "(float64 arithmetic, bit-identical to the dataset's), when the dataset neither splices nor skips frames "
                            "and runs without loader workers; 0: always in the dataset, as the reference does")
        p.add_argument("--hip_dist_backend", default="nccl", choices=["nccl", "gloo"],
                       help="torch.distributed backend under torch.distributed.run (nccl = RCCL over xGMI; gloo: rehearsal of the "
                            "N-rank path, also with several ranks on one GPU)")
        self.parser = p

    def get_args(self, argv=None):
        return self.parser.parse_args(argv)
