"""decode_asr on a synthetic test set of ragged utterances (300..1500 frames): the merged-pass decoder against the plain loop.

    python tools/ragged_cli_bench.py [--utts 2000] [--batch 32] [--precision bf16]

Writes a Kaldi ark/scp of --utts utterances (seeded N(0,1) features, the bench's config-2 model and weights), then runs the
package's recogniser (`cassnat_asr_public_amd.bin.decode_asr`, in process) three ways on the SAME batches (`--hip_bucket 1`:
batches formed from the length-sorted list; the result file stays in file order):

  plain      --hip_pipelines 1: one beam_decode call per batch, the reference's loop (src/tasks/cassnat_task.py:317-356)
  pipelined  the defaults: 2 decode pipelines, consecutive batches of different frame counts merged into engine passes by
             workspace area (cn_decode_nast_merged), decoder side on a predicted row count
  preloaded  the same with the collated batches already in (pinned) host memory: the decode loop without the ark reader

and checks that the result files are identical line for line.  Prints one JSON line (kept under profiles/).
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch
import yaml

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from cassnat_asr_public_amd import synth  # noqa: E402
from cassnat_asr_public_amd.data import kaldi_io  # noqa: E402
from cassnat_asr_public_amd.tasks import CassNATTask  # noqa: E402
from cassnat_asr_public_amd.utils.parser import DecodeParser  # noqa: E402


def make_task(cli, conf):
    args = DecodeParser().get_args(cli)
    for k, v in conf.items():
        setattr(args, k, v)
    args.rank = 0
    task = CassNATTask("test", args)
    task.load_lm_model(args)
    return task, args


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--utts", type=int, default=2000)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--min-frames", type=int, default=300)
    ap.add_argument("--max-frames", type=int, default=1500)
    ap.add_argument("--cmvn", type=int, default=1, help="1: the test config has use_cmvn + a global CMVN stats file (as the recipes do)")
    a = ap.parse_args()
    torch.set_num_threads(1)
    margs = synth.make_args("config2")
    state = synth.make_state(margs, seed=0, blank_bias=synth.BENCH_BLANK_BIAS)
    rng = np.random.default_rng(5)
    lengths = [int(x) for x in rng.integers(a.min_frames, a.max_frames + 1, size=a.utts)]
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        t0 = time.perf_counter()

        def mats():
            for b, n in enumerate(lengths):
                f, _ = synth.make_feats(1, n, margs.input_size, seed=4000 + b)
                yield f"spk-utt{b:05d}", f[0]

        scp = os.path.join(tmp, "feats.scp")
        kaldi_io.write_ark_scp(os.path.join(tmp, "feats.ark"), scp, mats())
        with open(os.path.join(tmp, "utt2num_frames"), "w") as f:
            for b, n in enumerate(lengths):
                f.write(f"spk-utt{b:05d} {n}\n")
        vocab_file = os.path.join(tmp, "vocab.txt")
        with open(vocab_file, "w") as f:
            f.write("".join(f"w{i}\n" for i in range(margs.vocab_size - 4)))
        ckpt = os.path.join(tmp, "model.mdl")
        torch.save({"model_state": {k: torch.from_numpy(v) for k, v in state.items()}}, ckpt)
        keys = ("input_size", "d_model", "n_head", "d_ff", "N_enc", "model_type", "n_features", "left_ctx", "right_ctx", "skip_frame",
                "padding_idx", "beam_width", "length_penalty", "d_encff", "d_decff", "N_extra", "N_self_dec", "N_mix_dec", "use_trigger")
        conf = {k: getattr(margs, k) for k in keys}
        conf.update(vocab_file=vocab_file, use_gpu=True, test_paths=[{"name": "test", "scp_path": scp}])
        if a.cmvn:  # Kaldi global CMVN stats (sums, sums of squares, count) of N(0.2, 1.5^2) features
            n = float(sum(lengths))
            stats = np.zeros((2, margs.input_size + 1))
            stats[0, :-1], stats[0, -1], stats[1, :-1] = 0.2 * n, n, (1.5 ** 2 + 0.2 ** 2) * n
            kaldi_io.write_ark_scp(os.path.join(tmp, "cmvn.ark"), os.path.join(tmp, "cmvn.scp"), [("global", stats)])
            conf.update(use_cmvn=True, global_cmvn=kaldi_io.read_scp(os.path.join(tmp, "cmvn.scp"))[0][1])
        cfg = os.path.join(tmp, "decode.yaml")
        with open(cfg, "w") as f:
            yaml.safe_dump({k: v for k, v in conf.items() if k != "test_paths"}, f)
        out["setup_s"] = round(time.perf_counter() - t0, 1)
        base = ["--task", "cassnat", "--test_config", cfg, "--data_path", scp, "--resume_model", ckpt, "--batch_size", str(a.batch),
                "--hip_precision", a.precision, "--hip_bucket", "1", "--hip_max_frames", str(a.max_frames),
                "--print_freq", "100000"]
        audio_s = sum(lengths) * 0.01
        results = {}
        # pipelined* = the packed reader (archive rows -> page-locked memory -> device, padding and CMVN on the device);
        # *_dataloader = round 3's form (the DataLoader's collated batches, CMVN on the device); *_cmvn_in_dataset = the reference's
        for name, extra, preload in (("plain", ["--hip_pipelines", "1", "--load_data_workers", "0"], False),
                                     ("pipelined", ["--load_data_workers", "0"], False),
                                     ("pipelined_2_copy_threads", ["--load_data_workers", "2"], False),
                                     ("pipelined_4_loader_workers", ["--load_data_workers", "4"], False),
                                     ("pipelined_8_copy_threads", ["--load_data_workers", "8"], False),
                                     ("pipelined_dataloader", ["--load_data_workers", "0", "--hip_packed_reader", "0"], False),
                                     ("pipelined_cmvn_in_dataset", ["--load_data_workers", "0", "--hip_device_cmvn", "0"], False),
                                     ("pipelined_again", ["--load_data_workers", "0"], False),
                                     ("preloaded", ["--load_data_workers", "0"], True)):
            print(f"[ragged_cli_bench] {name} ...", file=sys.stderr, flush=True)
            res = os.path.join(tmp, f"result_{name}.txt")
            task, args = make_task(base + extra + ["--result_file", res], conf)
            if preload:  # the collated batches as pinned host tensors: what a loader with enough workers hands over
                task.test_loader = [(u, f.pin_memory(), l, r.pin_memory(), s) for u, f, l, r, s in task.test_loader]
            torch.cuda.synchronize()
            c0 = time.perf_counter()
            task.decode(args)
            torch.cuda.synchronize()
            first = time.perf_counter() - c0
            c0 = time.perf_counter()  # again on the same task: engines, workspaces, threads and the row-count predictor exist
            task.decode(args)
            torch.cuda.synchronize()
            el = time.perf_counter() - c0
            results[name] = open(res).read().splitlines()
            rec = {"seconds": round(el, 3), "first_call_seconds": round(first, 3), "utt_per_s": round(a.utts / el, 1),
                   "audio_seconds_per_second": round(audio_s / el, 1)}
            st = getattr(task, "pipeline_stats", None)
            if st:
                rec["engine_passes"] = st["passes"]
                rec["batches"] = st["batches"]
                rec["passes_mixing_frame_counts"] = st["merged_ragged"]
                rec["row_predictions_missed"] = st["missed"]
                rec["worker_host_seconds"] = {k: round(v, 3) for k, v in st.items() if k.startswith("s_")}
            out[name] = rec
            eng = getattr(task.model, "_engine", None)
            if eng is not None:
                eng.close()
                task.model._engine = None
            if hasattr(task, "close"):
                task.close()
            del task
        assert len(results["plain"]) == a.utts
        assert results["pipelined"] == results["plain"], "result files differ between the merged-pass decoder and the plain loop"
        assert results["preloaded"] == results["plain"] and results["pipelined_4_loader_workers"] == results["plain"]
        assert results["pipelined_cmvn_in_dataset"] == results["plain"]
        for name in ("pipelined_2_copy_threads", "pipelined_8_copy_threads", "pipelined_dataloader", "pipelined_again"):
            assert results[name] == results["plain"], name
        out["result_files_identical"] = True
    out.update(global_cmvn=bool(a.cmvn), utterances=a.utts, batch_size=a.batch, precision=a.precision, frames_min_max=[min(lengths), max(lengths)],
               mean_frames=round(float(np.mean(lengths)), 1), audio_seconds=round(audio_s, 1),
               note="plain / pipelined include reading and collating the ark on the host (single process, no loader workers); "
                    "preloaded = the decode loop alone; bench.py's rtfx is audio seconds per second on resident features")
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
