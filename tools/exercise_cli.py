"""A test set of realistic size through the recogniser's command line: N utterances of 300..1500 frames (Kaldi ark / scp written
here), config 2 model, `decode_asr --task cassnat` once batch after batch (--hip_pipelines 1) and once with the default
pipelines and merged passes (identical files); also two ranks over gloo on this GPU, whose file must equal - exactly - the merge of
one-process runs over each rank's own utterance list (the ranks add the deal and the merge, nothing else).
    python tools/exercise_cli.py [--utts 160] [--batch 16]"""
import argparse
import os
import subprocess
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


def run_cli(cli, result, world=1):
    cmd = [sys.executable, "-m", "cassnat_asr_public_amd.bin.decode_asr"] + cli + ["--result_file", result]
    env = dict(os.environ, PYTHONPATH=REPO + os.pathsep + os.environ.get("PYTHONPATH", ""), HSA_ENABLE_IPC_MODE_LEGACY="0")
    t0 = time.perf_counter()
    if world == 1:
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
            env.pop(k, None)
        out = subprocess.run(cmd, env=env, cwd=REPO, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    else:
        procs = [subprocess.Popen(cmd + ["--hip_dist_backend", "gloo"],
                                  env=dict(env, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT="29733"),
                                  cwd=REPO, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
        try:
            for pr in procs:
                text, _ = pr.communicate(timeout=900)
                assert pr.returncode == 0, text[-4000:]
        finally:  # never leave a rank blocked in a collective behind a failed peer
            for pr in procs:
                if pr.poll() is None:
                    pr.kill()
                    pr.wait()
    return open(result).read().splitlines(), time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--utts", type=int, default=160)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--task", default="cassnat", choices=["cassnat", "art"], help="art: the autoregressive model (config 4), beam 10")
    a = ap.parse_args()
    ast = a.task == "art"
    args = synth.make_args_ast("config4") if ast else synth.make_args("config2")
    state = synth.make_state(args, seed=0, blank_bias=synth.BENCH_BLANK_BIAS)
    rng = np.random.default_rng(5)
    # lengths in runs, as a length-sorted test set has them: many utterances share a padded batch shape
    lengths = sorted((int(x) // 40 * 40 for x in rng.integers(300, 1501, size=a.utts)), reverse=True)
    with tempfile.TemporaryDirectory() as tmp:
        mats = []
        for b, n in enumerate(lengths):
            f, _ = synth.make_feats(1, n, args.input_size, seed=4000 + b)
            mats.append((f"spk-utt{b:04d}", f[0]))
        scp = os.path.join(tmp, "feats.scp")
        kaldi_io.write_ark_scp(os.path.join(tmp, "feats.ark"), scp, mats)
        vocab_file = os.path.join(tmp, "vocab.txt")
        open(vocab_file, "w").write("".join(f"w{i}\n" for i in range(args.vocab_size - 4)))
        ckpt = os.path.join(tmp, "model.mdl")
        torch.save({"model_state": {"module." + k: torch.from_numpy(v) for k, v in state.items()}}, ckpt)
        keys = ("input_size", "d_model", "n_head", "d_ff", "N_enc", "model_type", "n_features", "left_ctx", "right_ctx", "skip_frame", "padding_idx",
                "beam_width", "length_penalty")
        keys += ("N_dec", "ctc_beam", "ctc_weight", "max_decode_ratio", "T") if ast else ("d_encff", "d_decff", "N_extra", "N_self_dec", "N_mix_dec",
                                                                                      "use_trigger")
        conf = {k: getattr(args, k) for k in keys}
        conf.update(vocab_file=vocab_file, use_gpu=True)
        cfg = os.path.join(tmp, "decode.yaml")
        open(cfg, "w").write(yaml.safe_dump(conf))
        cli = ["--task", a.task, "--test_config", cfg, "--data_path", scp, "--resume_model", ckpt, "--batch_size", str(a.batch),
               "--hip_max_frames", "1500", "--hip_precision", a.precision]
        plain, t_plain = run_cli(cli + ["--hip_pipelines", "1"], os.path.join(tmp, "r_plain.txt"))
        piped, t_piped = run_cli(cli, os.path.join(tmp, "r_piped.txt"))
        two, t_two = run_cli(cli, os.path.join(tmp, "r_two.txt"), world=2)
        assert len(plain) == a.utts, len(plain)
        assert piped == plain, "pipelined result file differs from the batch-after-batch one"
        # Two ranks deal the utterances differently: other batches - other padding, another T' behind src_size = (ratio * T').long(),
        # another batch maximum of the token count (cassnat.py:436, 580-636) - so a line may differ from the one-process file, as it
        # would between two batchings of the reference.  What must hold EXACTLY: a rank's lines are those of a one-process run over
        # that rank's own list (the snake deal's order, same batch size), i.e. the ranks add nothing but the deal and the merge.
        assert [ln.split()[0] for ln in two] == [ln.split()[0] for ln in plain]
        from cassnat_asr_public_amd import dist as cdist

        entries = [ln for ln in open(scp).read().splitlines() if ln.strip()]
        per_rank = {}
        for r in range(2):
            idx = cdist.shard_indices(np.array(lengths), 2, r)
            sub = os.path.join(tmp, f"feats_rank{r}.scp")
            open(sub, "w").write("".join(entries[i] + "\n" for i in idx))
            cli_r = [x if x != scp else sub for x in cli]
            lines, _ = run_cli(cli_r, os.path.join(tmp, f"r_rank{r}.txt"))
            per_rank.update({ln.split()[0]: ln for ln in lines})
        merged = [per_rank[ln.split()[0]] for ln in plain]
        differ_from_one = sum(x != y for x, y in zip(two, plain))
        print("two ranks vs one process: %d of %d lines differ (other batch mates); two ranks vs one-process runs over each rank's own "
              "list: %d differ" % (differ_from_one, len(plain), sum(x != y for x, y in zip(two, merged))))
        assert two == merged, "a rank's lines differ from a one-process run over that rank's own utterance list"
        audio_s = sum(lengths) * 0.01
        print("%d utterances (%.0f s of audio), batch %d: result files identical; wall incl. start-up: plain %.1f s, pipelines %.1f s, "
              "two ranks on one GPU %.1f s" % (a.utts, audio_s, a.batch, t_plain, t_piped, t_two))


if __name__ == "__main__":
    main()
