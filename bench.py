#!/usr/bin/env python3
"""Benchmark of the CASS-NAT inference hot path on MI355X (BASELINE.json: utterances/sec + RTF).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path (CassNAT.beam_decode: conv subsampling -> 12L encoder -> CTC greedy
alignment -> token-acoustic extractor -> NAT decoder -> greedy finish) over one synthetic batch of
32 utterances x 1000 frames x 80-dim fbank per GPU (BASELINE configs[1]), features resident in HBM, bf16 MFMA
with fp32 accumulation, random-init weights of the named architecture (seeded; blank-biased for a realistic
token count).  With N > 1 every rank decodes its own shard (weak scaling); rank 0 packs the checkpoint once and
the weight blob is broadcast over RCCL; each step ends with one all-gather of the hypothesis records.

Rank 0 prints ONE JSON line.  `roofline` describes the dominant kernel (timed live with HIP events on the launch
stream inside the timed region); `cpu_baseline` is the oracle (torch-CPU restatement of the reference's ATen op
sequence, kind "port") timed on this host's cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

PEAK_BF16_DENSE_TFLOPS = 2500.0  # MI355X_MICROARCH.md: ~2.5 PF dense bf16 MFMA
PEAK_FP8_DENSE_TFLOPS = 5000.0  # MI355X_MICROARCH.md: ~5 PF dense fp8 (e4m3) MFMA
PEAK_HBM_GBS = 8000.0


def flops_per_batch(B, T, F, U, a):
    """Algorithmic FLOPs of one batch (SURVEY 8d formula, FLOP = 2 MAC)."""
    d, V, dff_e, dff_d = a.d_model, a.vocab_size, a.d_encff, a.d_decff
    T1, F1 = (T - 1) // 2 + 1, (F - 1) // 2 + 1
    Tp, F2 = (T1 - 1) // 2 + 1, (F1 - 1) // 2 + 1
    mac = B * d * T1 * F1 * 9 + B * d * Tp * F2 * 9 * d + B * Tp * d * F2 * d
    mac += a.N_enc * B * Tp * (4 * d * d + 2 * Tp * d + 2 * d * dff_e)
    mac += B * Tp * d * V
    mac += a.N_extra * (2 * B * U * d * d + 2 * B * Tp * d * d + 2 * B * U * Tp * d + 2 * B * U * d * dff_d)
    mac += a.N_self_dec * B * U * (4 * d * d + 2 * U * d + 2 * d * dff_d)
    mac += a.N_mix_dec * (B * U * (6 * d * d + 2 * U * d + 2 * Tp * d + 2 * d * dff_d) + 2 * B * Tp * d * d)
    mac += B * U * d * V
    return 2.0 * mac


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--batch", type=int, default=32, help="utterances per GPU per step")
    ap.add_argument("--frames", type=int, default=1000)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32", "fp8", "bf16x3", "fp16"],
                    help="fp8: BASELINE config 5's mode - the bf16 engine with the encoder layers' products on the e4m3fn MFMA")
    ap.add_argument("--fp8-scope", default="all", help="--precision fp8: which products take e4m3 operands (--hip_fp8_scope of the CLI)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity-engine", action="store_true",
                    help="skip timing the engines that meet the parity gate (fp32 MFMA, split-bf16) beside the headline")
    ap.add_argument("--cpu-batches", type=int, default=3)
    ap.add_argument("--stage-profile", action="store_true", help="also print a per-kernel-tag table to stderr")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="gloo = rehearsal of the N>1 path on one GPU")
    ap.add_argument("--streams", type=int, default=2,
                    help="decode pipelines per GPU (own engine handle, HIP stream and host thread each): keeps the GPU fed "
                         "across the two host syncs every batch needs (token count readback, hypotheses to host)")
    ap.add_argument("--coalesce", type=int, default=10,
                    help="batches of 32 a decode pipeline may take through ONE engine pass (wider launches; every batch's "
                         "hypotheses and scores stay exactly those of a pass of its own: cn_decode_opts.sub_batch)")
    ap.add_argument("--exit-after-timed", action="store_true",
                    help="tracing aid: print value / ms_per_step only and leave right after the timed region (so that it is the "
                         "end of a kernel trace: tools/trace_tail.py)")
    ap.add_argument("--no-uncoalesced", action="store_true",
                    help="skip the extra measurement with one batch per engine pass (reported beside `value`)")
    ap.add_argument("--force-dist", action="store_true",
                    help="N = 1 through the N > 1 code: a world-size-1 RCCL process group, the weight-blob broadcast and the "
                         "all_gather_into_tensor of the hypothesis records (what the driver's multi-GPU command runs, on one GPU)")
    ap.add_argument("--no-ragged-leg", action="store_true",
                    help="skip the extra measurement on a length-sorted list of batches of DIFFERENT frame counts (300..1500), "
                         "merged by workspace area as decode_asr does (reported beside `value`)")
    ap.add_argument("--no-predict", action="store_true", help="decode every pass with the mid-pass host sync on the row count (round 2's form)")
    ap.add_argument("--ragged", type=float, default=0.75, help="--hip_ragged of the CLI for the ragged leg (its default)")
    ap.add_argument("--cpu-threads", type=int, default=16, help="upper bound on the host threads of the cpu_baseline leg")
    ap.add_argument("--host-timeline", action="store_true", help="stderr: what the pipelines' host threads did when, in the timed region")
    ap.add_argument("--plan", default="", help="explicit pass sizes of the timed run, e.g. 8,8,4 (default: equal shares)")
    a = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    from cassnat_asr_public_amd import dist as cdist
    from cassnat_asr_public_amd import hip, synth
    from cassnat_asr_public_amd.models.cassnat import make_model

    # host-side torch ops of the GPU legs run single-threaded: an OpenMP team spun up by a CPU tensor op keeps spinning and, under
    # the box's per-process CPU quota, gets the whole process throttled for tens of milliseconds (tools/esa_stall_probe.py); the
    # cpu_baseline leg sets its own thread count
    torch.set_num_threads(1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")
        a.gpus = world
    local_rank = min(local_rank, torch.cuda.device_count() - 1)  # gloo rehearsal: several ranks may share one GPU
    torch.cuda.set_device(local_rank)
    dist_on = world > 1 or a.force_dist
    if dist_on:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if world == 1:  # --force-dist: a world of one rank on this GPU
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")

    class Vocab:
        word2index = {"blank": 0, "sos": 1, "eos": 2, "unk": 3}

    args = synth.make_args("config2")
    args.hip_precision = a.precision
    args.hip_fp8_scope = a.fp8_scope
    args.hip_max_batch, args.hip_max_frames = a.batch, a.frames
    B, T, F = a.batch, a.frames, args.input_size

    # ---- model: rank 0 owns the checkpoint, everyone else receives the packed blob over RCCL.
    # cassnat_asr_public_amd.pipeline.DecodePipelines = NS decode pipelines (engine handle with its weights + workspace, HIP
    # stream and persistent host thread each; ONE shared device copy of the weights): the object the package's own test-set decoder uses.
    from cassnat_asr_public_amd.pipeline import DecodePipelines

    NS = max(1, a.streams)
    state = None
    model = make_model(F, args).cuda(local_rank)
    if rank == 0:
        state = synth.make_state(args, seed=0, blank_bias=synth.BENCH_BLANK_BIAS)
        with torch.no_grad():
            for k, p in model.named_parameters():
                p.copy_(torch.from_numpy(state[k]))
    bcast = {"ms": None}

    def receive_weights(eng):
        if dist_on:
            t0 = time.perf_counter()
            cdist.broadcast_weights(eng, src=0)
            bcast["ms"] = (time.perf_counter() - t0) * 1e3

    CO = max(1, a.coalesce) if a.precision != "fp32" else 1
    pipes = DecodePipelines(model, NS, B, T, with_weights=(rank == 0), after_engine=receive_weights, coalesce=CO,
                            predict_rows=not a.no_predict)
    plan = [int(x) for x in a.plan.split(",")] if a.plan else None
    engines = pipes.engines
    bcast_ms = bcast["ms"]
    blob_bytes = engines[0].weight_blob()[1]

    # ---- synthetic batch, resident in HBM before the timed region (each rank its own shard)
    feats_h, sizes_h = synth.make_feats(B, T, F, seed=1234 + rank)
    feats = torch.from_numpy(feats_h).cuda()
    sizes = torch.from_numpy(sizes_h).cuda()

    def fence():
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    last = {}

    def run_steps(n_steps):
        """n_steps batches over the NS decode pipelines.  The per-batch all-gather (N > 1) and the hypotheses' trip to the
        host happen here, in THIS thread only and in step order, so every rank issues the same sequence of collectives."""
        # (hypotheses reach the host as arrays - tokens, lengths, scores of all ranks' utterances - without per-utterance Python
        # work: at 8 ranks a step carries 256 of them)
        for _, hyps_, scores_ in pipes.decode([(feats, sizes, k) for k in range(n_steps)], args, sos=1, gather=dist_on,
                                              as_lists=False, plan=list(plan) if plan and n_steps == a.steps else None):
            last[0] = (hyps_, scores_)

    run_steps(max(a.warmup, NS * CO))
    if pipes.predict:  # (the first passes run exactly and teach the row-count predictor; one more, predicted, warms that form up)
        run_steps(NS * CO)
    stats0 = dict(pipes.stats)
    U = int(engines[0].fetch("ymax")[0])
    eng = engines[0]
    fence()
    # the dominant kernel (by time; its encoder-side and decoder-side launches are timed apart) and the largest single product
    ROOF_TAGS = ["row_chain", "row_chain_dec", "conv2"]
    # HIP-event pairs on the launch stream around these kernels only, on ONE of the pipelines (22 pairs per step: on all
    # of them the event traffic itself costs a few percent of throughput)
    for e in engines[:1]:
        e.profile_begin(ROOF_TAGS)
    if a.host_timeline:
        pipes.timeline = []
    t0 = time.perf_counter()
    run_steps(a.steps)
    t_run = time.perf_counter()
    fence()
    elapsed = time.perf_counter() - t0
    if a.host_timeline and rank == 0:
        tl, pipes.timeline = pipes.timeline, None
        for label, k, t in sorted(tl, key=lambda x: x[2]):
            print("%9.3f ms  %-12s %s" % ((t - t0) * 1e3, label, "" if k < 0 else "pipeline %d" % k), file=sys.stderr)
        print("%9.3f ms  run_steps returned\n%9.3f ms  fenced" % ((t_run - t0) * 1e3, elapsed * 1e3), file=sys.stderr)
    pass_stats = {k: pipes.stats[k] - stats0[k] for k in stats0}
    if a.exit_after_timed:
        if rank == 0:
            print(json.dumps({"value": round(a.steps * B * world / elapsed, 2), "ms_per_step": round(elapsed / a.steps * 1e3, 4), "steps": a.steps}))
        pipes.close()
        if dist_on:
            dist.destroy_process_group()
        return
    prof = {t: {"count": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0} for t in ROOF_TAGS}
    for e in engines[:1]:
        got = e.profile_end()
        for t in ROOF_TAGS:
            if got.get(t):
                for k in prof[t]:
                    prof[t][k] += got[t][k]
    (toks_, lens_), scores = last[0]
    hyps = [toks_[b, : lens_[b]].tolist() for b in range(toks_.shape[0])]  # the last step's hypotheses, as token lists
    if dist_on:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    # ---- per-stage breakdown outside the timed region
    eng.profile_begin(None)
    for _ in range(3):  # one pipeline, the GPU to itself
        cdist.unpack_records(cdist.pack_records(*model.decode_device(feats, sizes, args, engine=eng)))
    stages = eng.profile_end()
    best_gpu = eng.fetch("best_paths")[:B]  # the timed engine's CTC arg-max per frame on the benchmark batch (for cpu_baseline)
    stage_ms = {k: round(v["ms"] / 3, 4) for k, v in sorted(stages.items(), key=lambda kv: -kv[1]["ms"])}
    # ... and the roofline kernels once more with the GPU to themselves at the WIDTH of the timed run (CO batches per pass)
    stages_wide = {}
    COW = min(CO, int(eng.cfg.max_batch) // B)  # (the workspace is sized by area: at --frames > 1024 it holds fewer batches than CO)
    if COW > 1:
        fw, sw = torch.cat([feats] * COW, 0), torch.cat([sizes] * COW, 0)
        model.decode_device(fw, sw, args, 1, engine=eng, sub_batch=B)
        eng.profile_begin(ROOF_TAGS)
        for _ in range(2):
            cdist.unpack_records(cdist.pack_records(*model.decode_device(fw, sw, args, 1, engine=eng, sub_batch=B)))
        stages_wide = eng.profile_end()
        del fw, sw
    if a.stage_profile and rank == 0:
        for k, v in sorted(stages.items(), key=lambda kv: -kv[1]["ms"]):
            tf = v["flops"] / (v["ms"] * 1e-3) / 1e12 if v["ms"] > 0 else 0
            gb = v["bytes"] / (v["ms"] * 1e-3) / 1e9 if v["ms"] > 0 else 0
            print(f"{k:20s} n={v['count'] // 3:3d} {v['ms'] / 3:8.3f} ms  {tf:8.1f} TFLOP/s  {gb:8.1f} GB/s(alg)", file=sys.stderr)

    # ---- extra (never `value`; N = 1 only): the same steps with one batch per engine pass
    uncoalesced = None
    if not a.no_uncoalesced and CO > 1 and world == 1:  # (like cpu_baseline: at N = 1 only)
        NS1 = max(NS, 3)  # one batch per pass needs more pipelines in flight than wide passes do
        pipes2 = DecodePipelines(model, NS1, B, T, coalesce=1, share_from=engines[0])  # (the same device copy of the weights)

        def run2(n_steps):
            for _ in pipes2.decode([(feats, sizes, k) for k in range(n_steps)], args, sos=1, as_lists=False):
                pass

        run2(max(a.warmup, NS1))
        fence()
        c0 = time.perf_counter()
        run2(a.steps)
        fence()
        el2 = time.perf_counter() - c0
        uncoalesced = {"value": round(a.steps * B * world / el2, 2), "unit": "utt/s", "ms_per_step": round(el2 / a.steps * 1e3, 4),
                       "decode_pipelines": NS1, "note": "same workload and step count with one batch of 32 per engine pass"}
        pipes2.close()

    # ---- extra (never `value`; N = 1 only): the same pipelines over ten times as many steps - at the driver's 20 steps the timed
    # region is one engine pass per pipeline (~15 ms: the first launch's wake-up and the last pass's copy home are 10 % of it)
    steady = None
    if world == 1 and not dist_on and a.steps < 100 and not a.no_uncoalesced:
        run_steps(10 * a.steps)  # (once untimed: the passes of a longer list are cut differently)
        fence()
        c0 = time.perf_counter()
        run_steps(10 * a.steps)
        fence()
        el_s = time.perf_counter() - c0
        steady = {"value": round(10 * a.steps * B / el_s, 2), "unit": "utt/s", "steps": 10 * a.steps,
                  "ms_per_step": round(el_s / (10 * a.steps) * 1e3, 4), "note": "same pipelines, same batch, ten times the steps"}

    # ---- extra (never `value`; N = 1 only): what decode_asr meets - a length-sorted list of batches of DIFFERENT frame counts
    # (300..1500, every batch padded to its own longest utterance), merged into engine passes by workspace area
    ragged_leg = None
    if not a.no_ragged_leg and world == 1 and not dist_on and a.precision in ("bf16", "bf16x3", "fp16"):
        rng = np.random.default_rng(99)
        n_b = 192  # 6144 utterances: the size of a test set (dev-clean + dev-other); neighbouring batches of the sorted list differ by ~6 frames
        lens = np.sort(rng.integers(300, 1501, size=n_b * B))[::-1]
        rb = []
        for k in range(n_b):
            lk = [int(x) for x in lens[k * B:(k + 1) * B]]
            fh, sh = synth.make_feats(B, lk[0], F, lengths=lk, seed=7000 + k)
            rb.append((torch.from_numpy(fh).cuda(), torch.from_numpy(sh).cuda(), k))
        audio_r = float(lens.sum()) * 0.01
        pipes3 = DecodePipelines(model, NS, B, 1500, coalesce=-CO, share_from=engines[0], predict_rows=not a.no_predict,
                                 ragged=a.ragged)

        def run3():
            for _ in pipes3.decode(rb, args, sos=1, as_lists=False):
                pass

        run3()
        st3 = dict(pipes3.stats)
        fence()
        c0 = time.perf_counter()
        run3()
        fence()
        el3 = time.perf_counter() - c0
        ragged_leg = {"value": round(n_b * B / el3, 2), "unit": "utt/s", "audio_seconds_per_second": round(audio_r / el3, 1),
                      "batches": n_b, "frames_min_max": [int(lens.min()), int(lens.max())], "mean_frames": round(float(lens.mean()), 1),
                      "engine_passes": pipes3.stats["passes"] - st3["passes"],
                      "passes_mixing_frame_counts": pipes3.stats["merged_ragged"] - st3["merged_ragged"],
                      "row_predictions_missed": pipes3.stats["missed"] - st3["missed"],
                      "note": "length-sorted list of 192 batches of 32 utterances of 300..1500 frames (6144 utterances), "
                              "each batch padded to its own longest utterance; consecutive batches share an engine pass while they fit the workspace area "
                              "(cn_decode_nast_merged: per-batch results identical to separate passes); compare "
                              "audio_seconds_per_second with rtfx"}
        pipes3.close()
        del rb

    if rank != 0:
        if dist_on:
            dist.destroy_process_group()
        return

    utts = a.steps * B * world
    value = utts / elapsed
    audio_s = utts * T * 0.01
    flops = flops_per_batch(B, T, F, U, args)
    # (the non-scaled fp8 MFMA of gfx950 runs at the bf16 rate: same peak)
    peak = PEAK_BF16_DENSE_TFLOPS if a.precision in ("bf16", "fp8", "bf16x3", "fp16") else 157.3

    def merged(table, tags):  # one row for several profile tags (the chain kernel's encoder-side + decoder-side launches)
        rows = [table[t] for t in tags if table.get(t) and table[t].get("count")]
        if not rows:
            return None
        return {k: sum(r_.get(k, 0.0) for r_ in rows) for k in ("count", "ms", "flops", "bytes")}

    def side(pr):  # a launch class of the chain kernel on its own
        if not pr or not pr.get("count"):
            return None
        avg_s = pr["ms"] / pr["count"] * 1e-3
        ach = pr["flops"] / pr["count"] / avg_s / 1e12
        return {"launches_timed": pr["count"], "flops_per_launch": round(pr["flops"] / pr["count"]), "avg_launch_us": round(avg_s * 1e6, 2),
                "achieved": round(ach, 2), "frac": round(ach / peak, 4)}

    def roof(tag, kernel, extra):
        tags = ["row_chain", "row_chain_dec"] if tag == "row_chain" else [tag]
        pr = merged(prof, tags)
        if not pr or not pr["count"]:
            return None
        avg_s = pr["ms"] / pr["count"] * 1e-3
        ach = pr["flops"] / pr["count"] / avg_s / 1e12
        pmc = None
        pmc_path = os.path.join(REPO, "profiles", f"pmc_{tag}.json")
        if os.path.exists(pmc_path):
            try:
                # counted in separate rocprofv3 --pmc passes on launches of a given width and grouped by launch width
                # (tools/pmc_summary.py): the mean over the launches of a pass of that width, quoted only beside such launches
                pj = json.load(open(pmc_path))
                pmc = pj.get("hbm_bytes_per_launch") if pj.get("batches_per_engine_pass", 3) == CO and "by_grid" in pj else None
            except Exception:
                pmc = None
        alg_bytes = pr["bytes"] / pr["count"] if pr.get("bytes") else None
        iso = merged(stages, tags)
        iso_tf = round(iso["flops"] / (iso["ms"] * 1e-3) / 1e12, 2) if iso and iso["ms"] > 0 else None
        isow = merged(stages_wide, tags)
        isow_tf = round(isow["flops"] / (isow["ms"] * 1e-3) / 1e12, 2) if isow and isow["ms"] > 0 else None
        r = {"kernel": kernel,
             "note": f"timed with HIP events inside the timed region while {NS} decode pipelines share the GPU; "
                     "'isolated_achieved' is the same kernel with the GPU to itself (one pipeline, one batch per pass, outside the "
                     f"timed region), 'isolated_at_width_achieved' likewise at {COW} batches per pass (the timed run's width where the workspace holds it)",
             "bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4),
             "traffic": pmc, "algorithmic_bytes_per_launch": None if alg_bytes is None else round(alg_bytes),
             "traffic_over_algorithmic": None if (pmc is None or not alg_bytes) else round(pmc / alg_bytes, 3),
             "flops_per_launch": round(pr["flops"] / pr["count"]), "avg_launch_us": round(avg_s * 1e6, 2),
             "launches_timed": pr["count"], "isolated_achieved": iso_tf,
             "isolated_frac": None if iso_tf is None else round(iso_tf / peak, 4),
             "isolated_at_width_achieved": isow_tf, "isolated_at_width_frac": None if isow_tf is None else round(isow_tf / peak, 4)}
        if a.precision == "bf16":
            # context for `peak` (not a replacement): what a bare register-resident v_mfma_f32_32x32x16_bf16 loop on random operands
            # sustains on this pool - the chip holds ~1.63 GHz under that load (tools/probes/mfma_shape_probe.hip)
            r["bare_mfma_loop_sustained"] = {"value": 1712.0, "unit": "TFLOP/s", "source": "profiles/r04n_mfma_shape_probe.txt"}
        if tag == "row_chain":
            # the same kernel at two launch widths: 12 encoder-side launches of every row of the pass (93 % of the kernel's FLOPs)
            # and 9 decoder-side ones of a tenth of the rows on a quarter of the CUs; the fields above are all 21 together
            r["encoder_launches"] = side(prof.get("row_chain"))
            r["decoder_side_launches"] = side(prof.get("row_chain_dec"))
            ew = stages_wide.get("row_chain")
            if r["encoder_launches"] and ew and ew.get("ms"):
                r["encoder_launches"]["isolated_at_width_frac"] = round(ew["flops"] / (ew["ms"] * 1e-3) / 1e12 / peak, 4)
        r.update(extra)
        return r

    Tp_ = ((T - 1) // 2) // 2 + 1
    enc_wgs = -(-B * Tp_ * CO // 128)  # 128-row workgroups over the merged pass's rows
    roofline = None
    if a.precision == "bf16":
        roofline = roof("row_chain",
                        "chain_kernel (per layer: attention out-projection + residual + LayerNorm + FFN + residual + next "
                        "LayerNorm + next Q|K|V projection; 21 launches per step: 12 encoder, 9 decoder-side)",
                        {"workgroups_encoder_launch": enc_wgs,
                         "design_note": "a launch deliberately occupies ceil(rows / 128) CUs (63 of 256 for the encoder): its weight "
                                        "stream is bound per CU, so the remaining CUs are left to the other decode pipelines; "
                                        "'frac' is against the whole chip's peak all the same"})
    roofline_conv2 = roof("conv2", "conv2_kernel (3x3 / stride 2, 256 -> 256 channels, LDS-DMA implicit GEMM, 188.7 GFLOP per batch)"
                          if a.precision in ("bf16", "fp8") else "gemm_kernel<implicit-conv> (conv2)", {})
    if roofline is None:
        roofline = roofline_conv2
    if roofline_conv2 is not None and a.precision == "fp8" and (hip.parse_fp8_scope(a.fp8_scope)[0] in (0, 1, 3, 5, 7)):
        # the e4m3 form of conv2 runs on v_mfma_scale_f32_32x32x64_f8f6f4: priced against the dense fp8 peak, not the bf16 one
        r_ = roofline_conv2
        r_["peak"] = PEAK_FP8_DENSE_TFLOPS
        r_["frac"] = round(r_["achieved"] / r_["peak"], 4)
        for k_ in ("isolated", "isolated_at_width"):
            if r_.get(k_ + "_achieved") is not None:
                r_[k_ + "_frac"] = round(r_[k_ + "_achieved"] / r_["peak"], 4)
        r_["kernel"] = "conv2_kernel<false, false, true> (e4m3 operands, K = 64 block-scaled MFMA; 188.7 GFLOP per batch)"
    if roofline is not None and a.precision == "bf16x3":  # three MFMAs per product: algorithmic FLOPs against peak / 3
        for r_ in (roofline, roofline_conv2):
            if r_:
                r_["peak"] = round(PEAK_BF16_DENSE_TFLOPS / 3, 1)
                r_["frac"] = round(r_["achieved"] / r_["peak"], 4)
                for k_ in ("isolated", "isolated_at_width"):
                    if r_.get(k_ + "_achieved") is not None:
                        r_[k_ + "_frac"] = round(r_[k_ + "_achieved"] / r_["peak"], 4)

    def margin_fields(best, ref_):
        from cassnat_asr_public_amd.utils.agreement import flips_by_margin

        m = flips_by_margin(best, ref_["best_paths"], ref_["ctc_margin"])
        return {"frames": m["frames"], "flips": m["flips"], "max_flip_margin": round(m["max_flip_margin"], 5),
                "flip_rate_margin_ge_0.05": round(m["flip_rate_margin_ge_0.05"], 5), "frames_margin_ge_0.05": m["frames_margin_ge_0.05"],
                "flip_rate_margin_ge_0.2": round(m["flip_rate_margin_ge_0.2"], 5), "frames_margin_ge_0.2": m["frames_margin_ge_0.2"]}

    cpu, ref = None, None
    if not a.no_cpu_baseline and world == 1:
        from oracle import cassnat_oracle as orc

        # host threads: this process's CPU share, not the machine's core count (a 1-GPU box gets 16 cores)
        try:
            ncores = len(os.sched_getaffinity(0))
        except AttributeError:
            ncores = os.cpu_count() or 1
        ncores = max(1, min(ncores, a.cpu_threads))
        torch.set_num_threads(ncores)
        st = orc.to_torch_state(state)
        orc.decode_nast(st, feats_h, sizes_h, args)  # warm-up
        times = []
        for _ in range(a.cpu_batches):
            c0 = time.perf_counter()
            ref = orc.decode_nast(st, feats_h, sizes_h, args)
            times.append(time.perf_counter() - c0)
        med = float(np.median(times))
        cpu = {"value": round(B / med, 3), "unit": "utt/s", "cores": torch.get_num_threads(), "kind": "port",
               "sample": f"{a.cpu_batches} batches of {B} x {T} frames (same workload), median of per-batch wall time "
                         f"{med:.2f} s after 1 warm-up; RTF {med / (B * T * 0.01):.5f}",
               "hyp_agreement_with_gpu": round(float(np.mean([h == r for h, r in zip(hyps[:B], ref["hyps"])])), 3),
               "note": "hyp_agreement_with_gpu: whole hypotheses of the timed engine (`dtype`) equal to this fp32 CPU run's; "
                       "ctc_argmax_vs_gpu: the timed engine's CTC arg-max per frame against this run's, by this run's top-2 margin "
                       "(a random-weight model's posteriors are near-flat; the flips of a reduced-precision engine sit on "
                       "low-margin frames: tests/test_gpu_pipeline.py gates that)"}
        cpu["ctc_argmax_vs_gpu"] = margin_fields(best_gpu, ref)
        torch.set_num_threads(1)  # (back to single-threaded host ops for the GPU legs below)

    # ---- the engines that meet north_star's tolerance (1e-3 on the CTC log-posteriors, token-exact alignment: tests/
    # test_gpu_pipeline.py::test_fp32_parity_gate), timed through the SAME decode pipelines for the SAME number of steps on
    # the same resident batch (N = 1 only, like cpu_baseline): the headline `value` is the throughput mode, these are the
    # numbers whose hypotheses are the reference's
    def time_engine(prec, fp8_scope="all"):
        ax = synth.make_args("config2")
        ax.hip_precision, ax.hip_max_batch, ax.hip_max_frames, ax.hip_fp8_scope = prec, B, T, fp8_scope
        mx = make_model(F, ax).cuda(local_rank)
        with torch.no_grad():
            for k, p in mx.named_parameters():
                p.copy_(torch.from_numpy(state[k]))
        nsx = max(NS, 3) if prec == "fp32" else NS  # (the f32-MFMA engine runs one batch per pass)
        px = DecodePipelines(mx, nsx, B, T, coalesce=1 if prec == "fp32" else CO)
        got = {}

        def runx(n_steps):
            for _, h_, s_ in px.decode([(feats, sizes, k) for k in range(n_steps)], ax, sos=1, as_lists=False):
                got[0] = h_

        runx(max(2, nsx * (1 if prec == "fp32" else CO)))
        if px.predict and prec != "fp32":  # (as the headline: the first passes run exactly and teach the row-count predictor; one more, predicted, warms that form up)
            runx(nsx * CO)
        fence()
        c0 = time.perf_counter()
        runx(a.steps)
        fence()
        el = time.perf_counter() - c0
        tk, ln = got[0]
        hx = [tk[b, : ln[b]].tolist() for b in range(tk.shape[0])]
        r = {"dtype": prec, "value": round(a.steps * B / el, 2), "unit": "utt/s", "ms_per_step": round(el / a.steps * 1e3, 4),
             "steps": a.steps, "decode_pipelines": nsx, "batches_per_engine_pass": 1 if prec == "fp32" else CO,
             "hyp_agreement": None if ref is None else round(float(np.mean([h == list(q) for h, q in zip(hx, ref["hyps"])])), 3)}
        if ref is not None:  # CTC arg-max agreement by the reference's top-2 margin (one more pass of the benchmark batch alone)
            mx.decode_device(feats, sizes, ax, engine=px.engines[0])
            r["ctc_argmax_vs_cpu"] = margin_fields(px.engines[0].fetch("best_paths")[:B], ref)
        px.close()
        return r

    parity_engine, fp32_engine = None, None
    if world == 1 and not a.no_parity_engine and a.precision == "bf16":
        fp32_engine = time_engine("fp32")
        fp32_engine["mfma_peak_tflops"] = 157.3
        fp32_engine["mfma_frac_end_to_end"] = round(flops / B * fp32_engine["value"] / 157.3e12, 5)
        # split-bf16: every value a (bf16 hi, bf16 lo) pair, every product three bf16 MFMAs -> its roofline is the dense bf16
        # peak / 3 in algorithmic FLOPs
        parity_engine = time_engine("bf16x3")
        parity_engine["mfma_peak_tflops"] = round(PEAK_BF16_DENSE_TFLOPS / 3, 1)
        parity_engine["mfma_frac_end_to_end"] = round(flops / B * parity_engine["value"] / (PEAK_BF16_DENSE_TFLOPS / 3 * 1e12), 5)
        parity_engine["speedup_over_fp32_engine"] = round(parity_engine["value"] / fp32_engine["value"], 3)
        parity_engine["note"] = ("the fastest engine that passes north_star's gate (tests/test_gpu_pipeline.py::test_fp32_parity_gate"
                                 "[bf16x3-*]: 0 arg-max flips, 1e-5 logit error, hypotheses token-exact); `hyp_agreement` = whole "
                                 "hypotheses of the benchmark batch equal to the fp32 CPU oracle's.  Its conv2 and feed-forward "
                                 "products run a mixed arithmetic - half-precision hi x hi + e4m3 cross terms, 2 MFMA units per "
                                 "product instead of the split form's 3 (DESIGN 9) - so `mfma_peak_tflops` = peak / 3 understates "
                                 "the roofline of those kernels (their own is peak / 2)")

    # the bf16 engine's kernels with IEEE half operands (libcassnat_hip_f16.so, csrc/common.h): the same matrix pipe at the same
    # rate, operand roundings 8 x smaller - CTC log-posteriors within north_star's 1e-3 of the fp32 reference
    # (tests/test_gpu_pipeline.py::test_fp16_engine_meets_the_logit_tolerance); range +-65504
    fp16_engine = None
    if world == 1 and not a.no_parity_engine and a.precision == "bf16":
        fp16_engine = time_engine("fp16")
        fp16_engine["speedup_over_value"] = round(fp16_engine["value"] / value, 3)
        fp16_engine["note"] = ("half-precision (11-bit) MFMA operands in the bf16 engine's kernels, fp32 accumulation / residual stream / "
                               "LayerNorm / softmax as there; `ctc_argmax_vs_cpu`: its flips against the fp32 CPU oracle by margin")

    # BASELINE configs[4]'s arithmetic on this workload: the fp8 engine (e4m3 feed-forward products inside the chain kernel and an
    # e4m3 conv front-end on the block-scaled K = 64 MFMA) through the same pipelines - a throughput form without a parity
    # claim (its agreement with the fp32 reference is measured by tests/test_gpu_pipeline.py::test_config5_fp8_encoder_products)
    fp8_engine = None
    if world == 1 and not a.no_parity_engine and a.precision == "bf16":
        fp8_engine = time_engine("fp8")
        fp8_engine["note"] = ("e4m3fn operands for the encoder's feed-forward products and both subsampling convolutions (v_mfma_scale_f32_32x32x64_"
                              "f8f6f4), everything else as the bf16 engine; `hyp_agreement` as for the other engines")
        fp8_engine["speedup_over_value"] = round(fp8_engine["value"] / value, 3)
        # fewer e4m3 products = fewer arg-max flips against the fp32 reference (tools/fp8_accuracy.py, 9681 frames of config 5's
        # shape: all 4.7 %, conv2+ffn 3.8 %, conv2+ffn:8 2.9 %, conv2 2.1 %, bf16 0.4 %) for less of the speed-up
        fp8_engine["scope"] = "all (conv2 + linear_out + feed-forward products of every encoder layer)"
        fp8_engine["other_scopes"] = {}
        for sc in ("conv2+ffn", "conv2+ffn:8", "conv2"):
            r = time_engine("fp8", sc)
            fp8_engine["other_scopes"][sc] = {"value": r["value"], "speedup_over_value": round(r["value"] / value, 3),
                                              "hyp_agreement": r["hyp_agreement"]}

    out = {
        "metric": "utterances_per_sec", "value": round(value, 2), "unit": "utt/s", "n_gpus": world, "steps": a.steps,
        "warmup": a.warmup, "ms_per_step": round(elapsed / a.steps * 1e3, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": a.precision, "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: CASS-NAT 12L-enc / 1+3+2 dec blocks, d_model 256, 4 heads, d_ff 2048, "
                               "V 5000, greedy NAST; 32 utterances x 1000 frames x 80 fbank per GPU per step",
                   "batch_per_gpu": B, "frames": T, "feat_dim": F, "global_batch": B * world, "tokens_U_max": U,
                   "parallelism": f"utterance-sharded x{world}", "decode_pipelines_per_gpu": NS,
                   "batches_per_engine_pass": CO, "utterances_per_engine_pass": B * CO,
                   "regime": "the package's own test-set decoder (pipeline.DecodePipelines, decode_asr's defaults: 2 pipelines, "
                             "consecutive batches merged into engine passes by workspace area with per-batch results unchanged, "
                             "decoder side launched on a predicted row count and verified)",
                   "engine_passes_timed": pass_stats["passes"], "row_predictions": pass_stats["predicted"],
                   "row_predictions_missed": pass_stats["missed"], "collectives": "rccl" if dist_on and a.backend == "nccl" else (a.backend if dist_on else None),
                   "blank_bias": synth.BENCH_BLANK_BIAS},
        "rtf": round(elapsed / audio_s, 8), "rtfx": round(audio_s / elapsed, 1),
        "gflop_per_utt": round(flops / B / 1e9, 3),
        # algorithmic FLOP/s of the whole path against the chip's peak for this precision's products (bf16x3: three MFMAs each)
        "mfma_frac_end_to_end": round(flops / B * value / (world * peak * 1e12 / (3.0 if a.precision == "bf16x3" else 1.0)), 5),
        "roofline": roofline, "roofline_conv2": roofline_conv2, "cpu_baseline": cpu, "parity_engine": parity_engine,
        "fp32_engine": fp32_engine, "fp16_engine": fp16_engine, "fp8_engine": fp8_engine, "one_batch_per_pass": uncoalesced, "ragged_set": ragged_leg, "steady_state": steady,
        "stage_ms": stage_ms,
        "weight_blob_mb": round(blob_bytes / 1e6, 2), "weight_broadcast_ms": None if bcast_ms is None else round(bcast_ms, 2),
    }
    print(json.dumps(out), flush=True)
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
