"""CPU-only tests: host-side mirror of the reference interface, C-ABI surface, multi-process sharding helpers."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import REPO, load_golden
from cassnat_asr_public_amd import dist as cdist
from cassnat_asr_public_amd import hip, synth
from cassnat_asr_public_amd.data import kaldi_io
from cassnat_asr_public_amd.data.feat_op import context_feat, skip_feat
from cassnat_asr_public_amd.data.speech_loader import SpeechDataLoader, SpeechDataset, collate
from cassnat_asr_public_amd.data.vocab import Vocab
from cassnat_asr_public_amd.models.cassnat import create_pe, make_model
from cassnat_asr_public_amd.tasks.cassnat_task import hyp_to_words
from cassnat_asr_public_amd.utils.parser import DecodeParser


# ------------------------------------------------------------------------------------------- C ABI surface
def test_library_exports_every_declared_symbol():
    names = hip.declared_symbols()
    assert len(names) >= 20
    # both builds of the sources (bf16 operands; half-precision operands: the fp16 engine) export the one interface
    for flavour, operand in ((None, b"bf16"), ("f16", b"fp16")):
        L = hip.lib(flavour)  # raises if include/cassnat_hip.h declares something the .so does not export
        for n in names:
            assert hasattr(L, n), n
        assert L.cn_version().startswith(b"cassnat_hip") and L.cn_operand16() == operand


def test_product_library_reads_no_environment_variable():
    """Kernel-selection / stamp / repeat switches exist only in -DCASSNAT_EXPERIMENTS builds (csrc/common.h: cn_exp_env): the
    shipped library neither imports getenv nor carries the name of a switch; the package's only environment variable is the
    loader-level CASSNAT_HIP_LIB (which library file to load)."""
    import re

    for path in (hip.LIB_PATH, hip.LIB_PATH_F16):
        out = subprocess.run(["nm", "-D", "--undefined-only", path], capture_output=True, text=True, check=True).stdout
        assert not re.search(r"\bgetenv\b", out)
        blob = open(path, "rb").read()
        assert b"CASSNAT_" not in blob
    pkg = os.path.join(REPO, "cassnat_asr_public_amd")
    used = set()
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                used |= set(re.findall(r"environ[^\n]*?[\"'](CASSNAT_\w+)[\"']", open(os.path.join(root, f)).read()))
    assert used == {"CASSNAT_HIP_LIB"}, used


def test_create_rejects_unsupported_geometry_without_touching_the_gpu():
    L = hip.lib()
    h = C.c_void_p()
    cfg = hip.CnConfig(input_size=80, d_model=100, n_head=4, d_encff=256, d_decff=256, n_enc=1, n_extra=1, n_self_dec=1,
                       n_mix_dec=1, vocab_size=50, precision=1, max_batch=2, max_frames=64, device=0)
    assert L.cn_model_create(C.byref(cfg), C.byref(h)) != 0
    assert b"d_model" in L.cn_last_error()
    cfg.d_model, cfg.precision = 256, 7
    assert L.cn_model_create(C.byref(cfg), C.byref(h)) != 0 and b"precision" in L.cn_last_error()
    cfg.precision, cfg.fp8_scope = 2, 2  # linear_out in e4m3 without the convolution that feeds it
    assert L.cn_model_create(C.byref(cfg), C.byref(h)) != 0 and b"fp8_scope" in L.cn_last_error()


def test_fp8_scope_strings():
    """--hip_fp8_scope -> (cn_config.fp8_scope, fp8_ffn_first_layer)"""
    assert hip.parse_fp8_scope("all") == (0, 0) and hip.parse_fp8_scope(None) == (0, 0)
    assert hip.parse_fp8_scope("conv2") == (1, 0) and hip.parse_fp8_scope("conv2+linear+ffn") == (7, 0)
    assert hip.parse_fp8_scope("conv2 + ffn:8") == (5, 8) and hip.parse_fp8_scope("ffn") == (4, 0)
    for bad in ("linear", "ffn+linear", "conv", "conv2:3", "ffn:-1"):
        with pytest.raises(ValueError):
            hip.parse_fp8_scope(bad)
    a = DecodeParser().get_args(["--test_config", "c.yaml", "--data_path", "f.scp", "--hip_precision", "fp8", "--hip_fp8_scope", "conv2+ffn:8"])
    assert hip.parse_fp8_scope(a.hip_fp8_scope) == (5, 8)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(hip, "_libs", {})
    monkeypatch.setattr(hip, "LIB_PATH", str(tmp_path / "nope.so"))
    monkeypatch.setattr(hip, "LIB_PATH_F16", str(tmp_path / "nope_f16.so"))
    with pytest.raises(hip.HipError, match="no CPU fallback"):
        hip.lib()
    with pytest.raises(hip.HipError, match="nope_f16.so is missing"):
        hip.lib_for("fp16")


# ------------------------------------------------------------------------------------------- model surface
def test_parameter_names_are_the_reference_checkpoint_keys():
    for preset in ("tiny", "config2"):
        args = synth.make_args(preset)
        model = make_model(args.input_size, args)
        assert {k: tuple(v.shape) for k, v in model.named_parameters()} == dict(synth.param_shapes(args))
        assert [k for k, _ in model.named_parameters()] == list(synth.param_shapes(args).keys())
    assert sum(p.numel() for p in model.parameters()) == 28673296


def test_positional_table_matches_oracle_closed_form():
    from oracle.cassnat_oracle import sinusoid_table

    assert torch.equal(create_pe(256), sinusoid_table(256))


def test_make_model_conformer_blocks_hold_the_reference_parameter_names():
    """use_conv_enc / use_conv_dec build conformer parameter holders whose names and shapes are the reference's checkpoint
    keys (the list in synth.param_shapes_conformer is checked against the instantiated reference in oracle/make_goldens.py);
    like the reference (src/models/cassnat.py:31) they insist on relative positions."""
    with pytest.raises(AssertionError):
        make_model(80, synth.make_args("tiny", use_conv_dec=True))
    for ov in (dict(), dict(use_conv_enc=False)):
        args = synth.make_args("tiny_conf", **ov)
        model = make_model(80, args)
        assert [(k, tuple(v.shape)) for k, v in model.named_parameters()] == list(synth.param_shapes_conformer(args).items())


# ------------------------------------------------------------------------------------------- data side
def test_splice_and_skip_match_reference_outputs():
    g = load_golden("feat_op")
    np.testing.assert_array_equal(context_feat(g["feat"], 2, 1), g["ctx_l2_r1"])
    np.testing.assert_array_equal(context_feat(g["feat"], 0, 2), g["ctx_r2"])
    padded = np.vstack([g["feat"], np.zeros((1, 4))])
    np.testing.assert_array_equal(skip_feat(context_feat(padded, 1, 1), 3), g["ctx_l1_r1_skip3"])
    assert context_feat(g["feat"], 0, 0) is g["feat"] and skip_feat(g["feat"], 1) is g["feat"]


def _write_dataset(tmp_path, lengths, dim=6):
    rng = np.random.default_rng(0)
    mats = [(f"utt{i:02d}", rng.standard_normal((n, dim)).astype(np.float32)) for i, n in enumerate(lengths)]
    ark, scp = str(tmp_path / "feats.ark"), str(tmp_path / "feats.scp")
    kaldi_io.write_ark_scp(ark, scp, mats)
    vocab_file = tmp_path / "vocab.txt"
    vocab_file.write_text("a\nb\nc 7\nutt a b d\n")
    text = tmp_path / "text"
    text.write_text("".join(f"utt{i:02d} a b zzz\n" for i in range(len(lengths))))
    return mats, scp, str(vocab_file), str(text)


def test_vocab_rules(tmp_path):
    _, _, vocab_file, _ = _write_dataset(tmp_path, [3])
    v = Vocab(vocab_file, rank=1)
    # single-token lines add the token; multi-field lines drop the first field
    assert v.word2index == {"blank": 0, "sos": 1, "eos": 2, "unk": 3, "a": 4, "b": 5, "7": 6, "d": 7}
    assert v.n_words == 8 and v.index2word[6] == "7"


def test_kaldi_roundtrip_cmvn_and_collate(tmp_path):
    mats, scp, vocab_file, text = _write_dataset(tmp_path, [9, 5, 7])
    entries = kaldi_io.read_scp(scp)
    assert [u for u, _ in entries] == [u for u, _ in mats]
    for (_, spec), (_, m) in zip(entries, mats):
        np.testing.assert_array_equal(kaldi_io.load_mat(spec), m)
    # global CMVN stats file in Kaldi layout (double matrix, 2 x (dim+1))
    allf = np.vstack([m for _, m in mats]).astype(np.float64)
    stats = np.zeros((2, 7))
    stats[0, :6], stats[0, 6], stats[1, :6] = allf.sum(0), len(allf), (allf ** 2).sum(0)
    kaldi_io.write_ark_scp(str(tmp_path / "cmvn.ark"), str(tmp_path / "cmvn.scp"), [("global", stats)])
    args = synth.make_args("tiny", left_ctx=1, right_ctx=1, skip_frame=2)
    ds = SpeechDataset(Vocab(vocab_file, 1), [{"name": "test", "scp_path": scp, "text_label": text}], args)
    ds._load_cmvn(kaldi_io.read_scp(str(tmp_path / "cmvn.scp"))[0][1])
    np.testing.assert_allclose(ds.mean, allf.mean(0))
    np.testing.assert_allclose(ds.std, allf.std(0))
    utt, feat, label = ds[1]  # 5 frames -> padded to 6 -> spliced (18 dims) -> every 2nd frame -> 3
    assert utt == "utt01" and feat.shape == (3, 18) and label == [1, 4, 5, 3, 2]
    norm = (mats[1][1] - ds.mean) / ds.std
    np.testing.assert_allclose(feat[0], np.concatenate([norm[0], norm[0], norm[1]]))  # edge replication
    utts, feats, texts, ratios, sizes = next(iter(SpeechDataLoader(ds, 3, padding_idx=0)))
    assert feats.shape == (3, 5, 18) and feats.dtype == torch.float32
    assert torch.equal(ratios, torch.tensor([5 / 5, 3 / 5, 4 / 5]))
    assert (feats[1, 3:] == 0).all() and sizes.tolist() == [3, 3, 3] and texts.shape == (3, 5)
    # the padding mask the task derives (src/tasks/cassnat_task.py:328)
    assert ((feats[:, :, 0] != 0).sum(1)).tolist() == [5, 3, 4]


def test_fast_feature_path_equals_the_general_one_bit_for_bit(tmp_path):
    """No splicing / no skipping (the shipped configuration): the dataset reads the matrix in place from a memory map, does the
    CMVN into a float64 scratch and hands on float32 - the same values as the reference's order of operations (float64 all the
    way, rounded to float32 where collate converts).  With `device_cmvn` it hands on the raw rows (the consumer normalises)."""
    rng = np.random.default_rng(3)
    mats = [(f"utt{b:02d}", (rng.standard_normal((n, 6)) * 3 + 1).astype(np.float32)) for b, n in enumerate([9, 5, 70, 33, 4, 1])]
    scp = str(tmp_path / "feats.scp")
    kaldi_io.write_ark_scp(str(tmp_path / "feats.ark"), scp, mats)
    allf = np.vstack([m for _, m in mats]).astype(np.float64)
    stats = np.zeros((2, 7))
    stats[0, :6], stats[0, 6], stats[1, :6] = allf.sum(0), len(allf), (allf ** 2).sum(0)
    kaldi_io.write_ark_scp(str(tmp_path / "cmvn.ark"), str(tmp_path / "cmvn.scp"), [("global", stats)])
    (tmp_path / "v").mkdir()
    _, _, vocab_file, _ = _write_dataset(tmp_path / "v", [3])
    args = synth.make_args("tiny", left_ctx=0, right_ctx=0, skip_frame=1)
    ds = SpeechDataset(Vocab(vocab_file, 1), [{"name": "test", "scp_path": scp}], args)
    ds._load_cmvn(kaldi_io.read_scp(str(tmp_path / "cmvn.scp"))[0][1])
    assert ds.can_defer_cmvn()
    utts, feats, _, ratios, _ = next(iter(SpeechDataLoader(ds, 6, padding_idx=0)))
    for b, (_, m) in enumerate(mats):
        want = torch.Tensor((m - ds.mean) / ds.std)  # the reference: float64 arithmetic, float32 at collate (speech_loader.py:340)
        assert torch.equal(feats[b, : len(m)], want) and (feats[b, len(m):] == 0).all()
        np.testing.assert_array_equal(kaldi_io.load_mat_view(kaldi_io.read_scp(scp)[b][1]), m)
    ds.device_cmvn = True
    _, raw, _, ratios2, _ = next(iter(SpeechDataLoader(ds, 6, padding_idx=0)))
    assert torch.equal(ratios, ratios2)
    for b, (_, m) in enumerate(mats):
        assert torch.equal(raw[b, : len(m)], torch.from_numpy(m))
    # what the pipelines do with raw batches on a CPU-only host (on the GPU: hip.cmvn_, tests/test_gpu_kernels.py)
    lens = (ratios.double() * raw.shape[1]).round().to(torch.int32).tolist()
    assert lens == [len(m) for _, m in mats]


def test_loader_workers_read_the_same_batches(tmp_path):
    """Main-process loader and forked loader workers (memory maps of the archives are inherited) hand over identical batches."""
    rng = np.random.default_rng(5)
    mats = [(f"utt{b:03d}", rng.standard_normal((int(rng.integers(20, 60)), 6)).astype(np.float32)) for b in range(40)]
    scp = str(tmp_path / "feats.scp")
    kaldi_io.write_ark_scp(str(tmp_path / "feats.ark"), scp, mats)
    (tmp_path / "v").mkdir()
    _, _, vocab_file, _ = _write_dataset(tmp_path / "v", [3])
    ds = SpeechDataset(Vocab(vocab_file, 1), [{"name": "test", "scp_path": scp}], synth.make_args("tiny", left_ctx=0, right_ctx=0, skip_frame=1))
    a = [b[1].clone() for b in SpeechDataLoader(ds, 8, padding_idx=0)]
    b = [x[1].clone() for x in SpeechDataLoader(ds, 8, padding_idx=0, num_workers=2)]
    assert len(a) == 5 and all(torch.equal(x, y) for x, y in zip(a, b))


def test_collate_ratio_is_float32_of_python_division():
    batch = [("a", np.ones((1000, 2)), [1, 2]), ("b", np.ones((333, 2)), [1, 2])]
    _, _, _, ratios, _ = collate(batch)
    assert ratios[1].item() == np.float32(333 / 1000)


def test_hyp_to_words_follows_reference_rule():
    class V:
        word2index = {"blank": 0, "sos": 1, "eos": 2, "unk": 3}
        index2word = {4: "x", 5: "y", 6: "z"}

    assert hyp_to_words([1, 4, 0, 5, 2, 6], V, 0) == ["x", "y"]
    from oracle.cassnat_oracle import hyp_to_text

    assert hyp_to_text([1, 4, 0, 5, 2, 6], V.index2word) == ["x", "y"]


def test_decode_parser_has_the_reference_flags():
    a = DecodeParser().get_args(["--test_config", "c.yaml", "--data_path", "f.scp", "--task", "cassnat", "--batch_size", "8",
                                 "--resume_model", "m.mdl", "--result_file", "r.txt", "--print_freq", "5", "--seed", "3",
                                 "--lm_config", "lm.yaml", "--text_label", "t", "--load_data_workers", "2",
                                 "--rnnlm", "x", "--rank_model", "lm", "--lm_weight", "0"])
    assert a.batch_size == 8 and a.task == "cassnat" and a.hip_precision == "bf16"


# ------------------------------------------------------------------------------------------- sharding helpers
def test_shard_indices_balance_lengths():
    lens = np.array([100, 900, 500, 300, 700, 200, 800, 400])
    parts = [cdist.shard_indices(lens, 2, r) for r in range(2)]
    assert sorted(np.concatenate(parts).tolist()) == list(range(8))
    assert abs(lens[parts[0]].sum() - lens[parts[1]].sum()) <= 100
    assert lens[parts[0]][0] == 900 and lens[parts[1]][0] == 800


def test_record_pack_roundtrip():
    hyp = torch.tensor([[1, 5, 6, 0], [1, 9, 0, 0]], dtype=torch.int32)
    hl = torch.tensor([3, 2], dtype=torch.int32)
    sc = torch.tensor([-12.34567890123, -0.5], dtype=torch.float64)
    hyps, scores = cdist.unpack_records(cdist.pack_records(hyp, hl, sc))
    assert hyps == [[1, 5, 6], [1, 9]] and scores.tolist() == sc.tolist()


_WORKER = r"""
import os, sys, torch, numpy as np
import torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from cassnat_asr_public_amd import dist as cdist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % sys.argv[2], rank=rank, world_size=world)
lens = np.array([50, 10, 40, 20, 30, 60])
mine = cdist.shard_indices(lens, world, rank)
hyp = torch.zeros(len(mine), 8, dtype=torch.int32)
for i, u in enumerate(mine):
    hyp[i, :3] = torch.tensor([1, 100 + int(u), 2])
rec = cdist.pack_records(hyp, torch.full((len(mine),), 3, dtype=torch.int32), torch.tensor([-float(u) for u in mine], dtype=torch.float64))
allrec = cdist.all_gather_records(rec)
hyps, scores = cdist.unpack_records(allrec)
got = sorted((int(-s), h[1] - 100) for h, s in zip(hyps, scores))
assert got == [(u, u) for u in range(6)], got
# weight hand-off: rank 0 owns the blob, rank 1 receives it
blob = torch.arange(1000, dtype=torch.uint8) if rank == 0 else torch.zeros(1000, dtype=torch.uint8)
dist.broadcast(blob, src=0)
assert blob[999].item() == 999 % 256
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_two_process_gloo_gather_and_broadcast(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    port = str(29500 + os.getpid() % 2000)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        procs.append(subprocess.Popen([sys.executable, str(script), REPO, port], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    for r, pr in enumerate(procs):
        out, _ = pr.communicate(timeout=120)
        assert pr.returncode == 0, out
        assert f"rank {r} ok" in out


def test_ast_parameter_names_are_the_reference_checkpoint_keys():
    from cassnat_asr_public_amd.models.transformer import make_model as make_ast

    args = synth.make_args_ast("tiny_ast")
    model = make_ast(args.input_size, args)
    assert {k: tuple(v.shape) for k, v in model.named_parameters()} == dict(synth.param_shapes_ast(args))
    assert [k for k, _ in model.named_parameters()] == list(synth.param_shapes_ast(args).keys())


def test_decode_pipelines_keep_submission_order_and_surface_errors():
    """Host logic of pipeline.DecodePipelines with a stub model (no GPU): results come back in submission order whatever
    the workers' speeds, every batch is decoded exactly once, a worker's exception reaches the consumer, and an early
    exit of the consumer stops the workers."""
    import time as _time

    import torch as _torch

    from cassnat_asr_public_amd.pipeline import DecodePipelines

    class StubEngine:
        def close(self):
            pass

    class StubModel:
        def __init__(self, fail_at=None):
            self.seen, self.fail_at, self.pairs = [], fail_at, 0

        def new_engine(self, batch, frames, with_weights=True, share=None):
            return StubEngine()

        def decode_device(self, feats, ratio, args, sos, engine=None, sub_batch=0, sub_rows=None, sub_frames=None, u_hint=0,
                          want_ticket=False):
            if want_ticket:  # (the pipelines' call: a fourth value, the ticket - none for a stub)
                return self.decode_device(feats, ratio, args, sos) + (-1,)
            if feats.shape[0] > 1:  # a coalesced pair: one pass, records of both in order
                parts = [self.decode_device(feats[j : j + 1], ratio[j : j + 1], args, sos) for j in range(feats.shape[0])]
                self.pairs += 1
                return tuple(_torch.cat([p_[i] for p_ in parts], 0) for i in range(3))
            k = int(feats[0, 0, 0])
            if self.fail_at is not None and k == self.fail_at:
                raise RuntimeError("boom")
            _time.sleep(0.002 * ((k * 7) % 5))  # uneven pipelines
            self.seen.append(k)
            hyp = _torch.tensor([[sos, 10 + k, 0]], dtype=_torch.int32)
            return hyp, _torch.tensor([2], dtype=_torch.int32), _torch.tensor([float(k)], dtype=_torch.float64)

    def batches(n):
        for k in range(n):
            yield _torch.full((1, 4, 2), float(k)), _torch.ones(1), ("tag", k)

    m = StubModel()
    pipes = DecodePipelines(m, 3, 1, 4)
    out = list(pipes.decode(batches(25), args=None, sos=1))
    assert [t for t, _, _ in out] == [("tag", k) for k in range(25)]
    assert [h for _, h, _ in out] == [[[1, 10 + k]] for k in range(25)]
    assert [float(s[0]) for _, _, s in out] == [float(k) for k in range(25)]
    assert sorted(m.seen) == list(range(25))
    # coalescing: pairs of equal-shaped consecutive batches ride one pass; order, tags and contents unchanged
    def mixed(n):
        for k in range(n):
            t = 4 if k % 7 else 6  # every seventh batch has another shape: it must not be merged with its neighbours
            yield _torch.full((1, t, 2), float(k)), _torch.ones(1), ("tag", k)

    m4 = StubModel()
    out4 = list(DecodePipelines(m4, 3, 1, 6, coalesce=2).decode(mixed(31), args=None, sos=1))
    assert [t for t, _, _ in out4] == [("tag", k) for k in range(31)]
    assert [h for _, h, _ in out4] == [[[1, 10 + k]] for k in range(31)]
    assert sorted(m4.seen) == list(range(31)) and m4.pairs >= 3  # (how many pairs form depends on how far the consumer is behind)
    # a failing batch surfaces in the consumer
    m2 = StubModel(fail_at=6)
    with pytest.raises(RuntimeError, match="boom"):
        list(DecodePipelines(m2, 2, 1, 4).decode(batches(20), args=None, sos=1))
    # early exit of the consumer: the generator closes, the workers stop after the batch they are on
    m3 = StubModel()
    gen = DecodePipelines(m3, 2, 1, 4).decode(batches(1000), args=None, sos=1)
    for n, _ in enumerate(gen):
        if n == 4:
            break
    gen.close()
    assert len(m3.seen) < 40


def test_decode_pipelines_cut_a_list_of_known_length_into_equal_passes():
    """pipeline._Job: with len(batches) known the passes are planned up front - n x rounds of them, sizes differing by at most
    one, no one-batch tail - and the merged input of a pass is the batches in order."""
    import torch as _torch

    from cassnat_asr_public_amd.pipeline import DecodePipelines

    class StubEngine:
        def close(self):
            pass

    class StubModel:
        def __init__(self):
            self.passes, self.subs = [], []

        def new_engine(self, batch, frames, with_weights=True, share=None):
            return StubEngine()

        def decode_device(self, feats, ratio, args, sos, engine=None, sub_batch=0, sub_rows=None, sub_frames=None, u_hint=0,
                          want_ticket=False):
            ks = [int(feats[j, 0, 0]) for j in range(feats.shape[0])]
            self.passes.append(ks)
            self.subs.append((sub_rows, sub_frames, tuple(feats.shape)))
            hyp = _torch.tensor([[sos, 10 + k, 0] for k in ks], dtype=_torch.int32)
            out = (hyp, _torch.full((len(ks),), 2, dtype=_torch.int32), _torch.tensor([float(k) for k in ks], dtype=_torch.float64))
            return out + (-1,) if want_ticket else out

    for n, c, total, want in ((3, 3, 20, [2, 2, 2, 2, 2, 2, 2, 3, 3]), (2, 10, 20, [10, 10]), (2, 10, 5, [2, 3]), (1, 4, 9, [3, 3, 3]),
                              (3, 3, 200, None)):
        m = StubModel()
        items = [(_torch.full((1, 4, 2), float(k)), _torch.ones(1), k) for k in range(total)]
        out = list(DecodePipelines(m, n, 1, 4, coalesce=c).decode(items, args=None, sos=1))
        assert [t for t, _, _ in out] == list(range(total))
        assert [h for _, h, _ in out] == [[[1, 10 + k]] for k in range(total)]
        sizes = sorted(len(p) for p in m.passes)
        assert sum(sizes) == total and sizes[-1] <= c, sizes
        if want is not None:  # (a long list may also see passes cut short by the bound on decoded-but-unconsumed batches)
            assert sizes == want, sizes
        for p in m.passes:  # consecutive batches, in order
            assert p == list(range(p[0], p[0] + len(p)))


def test_decode_pipelines_merge_batches_of_different_frame_counts():
    """pipeline.DecodePipelines: consecutive batches whose frame counts are close enough share ONE engine pass - the merged
    input holds them one after the other, each padded with padding frames to the longest, and the call names every batch's
    utterance and frame counts (cn_decode_nast_merged); a batch that is too short waits for the next pass; a pass never
    exceeds the engines' workspace area."""
    import torch as _torch

    from cassnat_asr_public_amd.pipeline import DecodePipelines

    class StubEngine:
        def close(self):
            pass

    class StubModel:
        def __init__(self):
            self.calls = []

        def new_engine(self, batch, frames, with_weights=True, share=None):
            return StubEngine()

        def decode_device(self, feats, ratio, args, sos, engine=None, sub_batch=0, sub_rows=None, sub_frames=None, u_hint=0,
                          want_ticket=False):
            self.calls.append((sub_rows, sub_frames, feats.clone(), ratio.clone()))
            n = feats.shape[0]
            ks = [int(feats[j, 0, 0]) for j in range(n)]
            hyp = _torch.tensor([[sos, 10 + k, 0] for k in ks], dtype=_torch.int32)
            return hyp, _torch.full((n,), 2, dtype=_torch.int32), _torch.tensor([float(k) for k in ks], dtype=_torch.float64), -1

    class Args:
        padding_idx = 0

    # (utterances, frames) per batch; values = 100 * batch + utterance so that every row is recognisable
    shapes = [(3, 40), (3, 36), (2, 33), (3, 20), (3, 19), (1, 40)]
    items = []
    for k, (nb, t) in enumerate(shapes):
        f = _torch.stack([_torch.full((t, 2), float(100 * k + j + 1)) for j in range(nb)])
        items.append((f, _torch.full((nb,), 1.0 - 0.01 * k), k))
    m = StubModel()
    pipes = DecodePipelines(m, 1, 3, 40, coalesce=4, ragged=0.8)
    out = list(pipes.decode(items, Args(), sos=1, plan=[4, 4, 4]))
    pipes.close()
    assert [t for t, _, _ in out] == list(range(len(shapes)))
    got = [(r, f) for r, f, _, _ in m.calls]
    # 40 / 36 / 33 frames merge (33 >= 0.8 * 40); 20 does not (it waits and opens the next pass, with 19); the last one is alone
    assert got == [([3, 3, 2], [40, 36, 33]), ([3, 3], [20, 19]), (None, None)], got
    feats, ratio = m.calls[0][2], m.calls[0][3]
    assert tuple(feats.shape) == (8, 40, 2)
    assert _torch.all(feats[3:6, :36] != 0) and _torch.all(feats[3:6, 36:] == 0) and _torch.all(feats[6:8, 33:] == 0)
    assert [int(feats[j, 0, 0]) for j in range(8)] == [1, 2, 3, 101, 102, 103, 201, 202]
    assert _torch.allclose(ratio, _torch.tensor([1.0] * 3 + [0.99] * 3 + [0.98] * 2))
    for k, hyps, _ in out:
        assert hyps == [[1, 10 + 100 * k + j + 1] for j in range(shapes[k][0])]
    # the area bound: a workspace of 3 x 40 frames (coalesce 1: no extra room) never takes 3 + 3 utterances of 36+ frames at once
    m2 = StubModel()
    p2 = DecodePipelines(m2, 1, 3, 40, coalesce=2, ragged=0.5, area_frames=3 * 40)
    assert p2.fits(3, 40) and not p2.fits(6, 36) and p2.fits(6, 19)
    list(p2.decode(items, Args(), sos=1))
    p2.close()
    assert all(r is None or sum(r) * max(f) <= 4 * 41 for r, f, _, _ in m2.calls)
    assert any(r == [3, 3] and f == [20, 19] for r, f, _, _ in m2.calls)


_WORKER_PIPES = r"""
import os, sys, time, torch
import torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from cassnat_asr_public_amd.pipeline import DecodePipelines
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % sys.argv[2], rank=rank, world_size=world)

class StubEngine:
    def close(self):
        pass

class StubModel:
    def new_engine(self, batch, frames, with_weights=True, share=None):
        return StubEngine()
    def decode_device(self, feats, ratio, args, sos, engine=None, sub_batch=0, sub_rows=None, sub_frames=None, u_hint=0, want_ticket=False):
        ks = [int(feats[j, 0, 0]) for j in range(feats.shape[0])]
        time.sleep(0.001 * (1 + rank))  # the ranks' pipelines run at different speeds
        hyp = torch.tensor([[sos, 10 + k, 100 + rank] for k in ks], dtype=torch.int32)
        out = (hyp, torch.full((len(ks),), 3, dtype=torch.int32), torch.tensor([float(k) + 0.25 * rank for k in ks], dtype=torch.float64))
        return out + (-1,) if want_ticket else out

# 23 steps of two utterances each; steps 9..11 have another shape (the gather groups break there)
items = []
for k in range(23):
    t = 6 if 9 <= k < 12 else 4
    items.append((torch.stack([torch.full((t, 2), float(2 * k)), torch.full((t, 2), float(2 * k + 1))]), torch.ones(2), k))
pipes = DecodePipelines(StubModel(), 2, 2, 6, coalesce=4)
out = list(pipes.decode(items, args=None, sos=1, gather=True))
pipes.close()
assert [t for t, _, _ in out] == list(range(23))
for k, hyps, scores in out:  # rank-major: rank 0's two utterances, then rank 1's
    assert hyps == [[1, 10 + 2 * k, 100], [1, 11 + 2 * k, 100], [1, 10 + 2 * k, 101], [1, 11 + 2 * k, 101]], (k, hyps)
    assert scores.tolist() == [2.0 * k, 2.0 * k + 1, 2.0 * k + 0.25, 2.0 * k + 1.25], (k, scores)
dist.destroy_process_group()
print("rank", rank, "ok")
"""


def test_two_process_gloo_decode_pipelines_gather(tmp_path):
    """DecodePipelines.decode(gather=True): the all-gather is issued once per group of consecutive equal-shaped steps, by step
    index - two ranks whose pipelines run at different speeds issue the same collectives and see every step rank-major."""
    script = tmp_path / "worker_pipes.py"
    script.write_text(_WORKER_PIPES)
    port = str(31500 + os.getpid() % 2000)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        procs.append(subprocess.Popen([sys.executable, str(script), REPO, port], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    for r, pr in enumerate(procs):
        out, _ = pr.communicate(timeout=180)
        assert pr.returncode == 0, out
        assert f"rank {r} ok" in out


def test_batched_hypothesis_to_words_equals_the_per_utterance_rule():
    """tasks.cassnat_task.hyps_to_words_batch (the pipelined decoder's form) against hyp_to_words, the reference's rule
    (src/tasks/cassnat_task.py:346-353: skip sos and padding ids, stop at the first eos), on random records incl. empty ones."""
    from cassnat_asr_public_amd.tasks.cassnat_task import hyp_to_words, hyps_to_words_batch

    class V:
        word2index = {"blank": 0, "sos": 1, "eos": 2, "unk": 3}
        n_words = 40
        index2word = {i: f"w{i}" for i in range(40)}

    rng = np.random.default_rng(3)
    toks = rng.integers(0, 40, size=(64, 25)).astype(np.int32)
    toks[rng.random(toks.shape) < 0.2] = 2  # plenty of eos
    lens = rng.integers(0, 26, size=64)
    got = hyps_to_words_batch(toks, lens, V, 0)
    assert got == [hyp_to_words(toks[b, : lens[b]].tolist(), V, 0) for b in range(64)]


def test_flips_by_margin_counts_and_gate():
    """The margin-conditioned agreement report (utils/agreement.py) that the bf16 / fp8 GPU gates and bench.py use."""
    from cassnat_asr_public_amd.utils.agreement import assert_flips_explained, flips_by_margin

    ref = np.array([[1, 2, 3, 4, 5, 6]])
    got = np.array([[1, 9, 3, 9, 5, 9]])
    margin = np.array([[0.5, 0.003, 0.3, 0.06, 0.0, 0.25]], np.float32)
    own = np.array([[1, 1, 1, 1, 1, 0]], bool)  # the last frame is padding: its flip does not count
    r = flips_by_margin(got, ref, margin, own)
    assert (r["frames"], r["flips"]) == (5, 2) and abs(r["max_flip_margin"] - 0.06) < 1e-7
    assert (r["flips_margin_ge_0.05"], r["frames_margin_ge_0.05"]) == (1, 3)
    assert (r["flips_margin_ge_0.2"], r["frames_margin_ge_0.2"]) == (0, 2)
    assert sum(f for f, _ in r["flips_frames_by_margin"].values()) == 2 and sum(n for _, n in r["flips_frames_by_margin"].values()) == 5
    assert_flips_explained(r, 0.03, 2.0)  # 0.06 <= 2 x 0.03
    with pytest.raises(AssertionError, match="clear-margin frame flipped"):
        assert_flips_explained(r, 0.02, 2.0)
    assert flips_by_margin(ref, ref, margin)["max_flip_margin"] == 0.0


def test_snake_deal_balances_eight_ranks():
    """BASELINE configs[2]'s shape: 256 ragged utterances dealt over 8 ranks (dist.shard_indices) - a partition, every rank 32
    utterances, longest first, and a frame total within 3 % of the mean (what keeps eight GPUs busy for the same time)."""
    rng = np.random.default_rng(12)
    lens = rng.integers(300, 1501, size=256)
    parts = [cdist.shard_indices(lens, 8, r) for r in range(8)]
    assert sorted(np.concatenate(parts).tolist()) == list(range(256))
    totals = np.array([lens[p].sum() for p in parts], np.float64)
    assert all(len(p) == 32 for p in parts)
    assert all((np.diff(lens[p]) <= 0).all() for p in parts)
    assert np.abs(totals / totals.mean() - 1).max() < 0.03, totals


def test_eight_ranks_over_gloo_write_the_one_process_result_file(tmp_path):
    """configs[2] rehearsed without eight GPUs: `decode_asr --task cassnat` as 8 ranks over gloo on the CPU, 256 ragged utterances,
    --batch_size 1 (results independent of batch mates) - the product's host side end to end (CassNATTask, the snake deal, the one
    weight broadcast per rank, DecodePipelines with merged ragged passes, the gather of the ranks' results, rank 0's input-ordered
    file) around a stub engine (tests/_stub_rank_worker.py) whose tokens encode the utterance id and its unpadded frame count.
    The file equals the one-process run's line for line, and both equal what the stub must emit for every utterance."""
    import yaml

    rng = np.random.default_rng(21)
    lengths = [int(x) for x in rng.integers(30, 151, size=256)]
    mats = []
    for u, n in enumerate(lengths):
        m = np.full((n, 4), float(u + 1), np.float32)
        mats.append((f"spk{u % 7}-utt{u:03d}", m))
    perm = rng.permutation(256)  # the table is not in id order: the file must follow the table's order
    mats = [mats[i] for i in perm]
    scp = str(tmp_path / "feats.scp")
    kaldi_io.write_ark_scp(str(tmp_path / "feats.ark"), scp, mats)
    (tmp_path / "utt2num_frames").write_text("".join(f"{u} {m.shape[0]}\n" for u, m in mats))
    vocab_file = tmp_path / "vocab.txt"
    vocab_file.write_text("".join(f"w{i}\n" for i in range(20)))
    ckpt = str(tmp_path / "model.mdl")
    torch.save({"model_state": {"w": torch.zeros(4)}}, ckpt)
    conf = dict(input_size=4, n_features=4, left_ctx=0, right_ctx=0, skip_frame=1, padding_idx=0, beam_width=1, length_penalty=0,
                use_trigger=True, vocab_file=str(vocab_file), use_gpu=False, model_type="transformer")
    cfg = tmp_path / "decode.yaml"
    cfg.write_text(yaml.safe_dump(conf))
    worker = os.path.join(REPO, "tests", "_stub_rank_worker.py")

    def run(world, tag):
        result = str(tmp_path / f"result_{tag}.txt")
        cmd = [sys.executable, worker, REPO, "--task", "cassnat", "--test_config", str(cfg), "--data_path", scp, "--resume_model", ckpt,
               "--result_file", result, "--batch_size", "1", "--load_data_workers", "0", "--hip_dist_backend", "gloo", "--hip_max_frames", "160"]
        env = dict(os.environ, PYTHONPATH=REPO + os.pathsep + os.environ.get("PYTHONPATH", ""), OMP_NUM_THREADS="1")
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
            env.pop(k, None)
        if world == 1:
            out = subprocess.run(cmd, env=env, cwd=REPO, capture_output=True, text=True, timeout=300)
            assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
        else:
            port = str(33500 + os.getpid() % 2000)
            procs = [subprocess.Popen(cmd, env=dict(env, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                                                    MASTER_PORT=port), cwd=REPO, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
                     for r in range(world)]
            try:
                for pr in procs:
                    text, _ = pr.communicate(timeout=300)
                    assert pr.returncode == 0, text[-4000:]
            finally:
                for pr in procs:
                    if pr.poll() is None:
                        pr.kill()
                        pr.wait()
        return open(result).read().splitlines()

    one = run(1, "w1")
    eight = run(8, "w8")
    expect = []
    for utt, m in mats:
        uid, n = int(m[0, 0]) - 1, m.shape[0]
        expect.append(f"{utt} w{uid % 20} w{(uid // 20) % 20} w{n % 20} w{(n // 20) % 20}")
    assert one == expect
    assert eight == one


def test_host_gather_copies_archive_rows_back_to_back(tmp_path):
    """cn_host_gather (the host half of the packed reader; no GPU involved): read-only views into a memory-mapped archive copied
    back to back into one buffer by a single GIL-free call, alone and dealt over several threads; and pipeline.PackedBatch's
    shape / ratios / padded tensor against the dataset's collate."""
    from cassnat_asr_public_amd.pipeline import PackedBatch

    rng = np.random.default_rng(4)
    mats = [(f"u{i}", rng.standard_normal((int(n), 7)).astype(np.float32)) for i, n in enumerate(rng.integers(1, 90, size=37))]
    scp = str(tmp_path / "f.scp")
    kaldi_io.write_ark_scp(str(tmp_path / "f.ark"), scp, mats)
    views = [kaldi_io.load_mat_view(spec) for _, spec in kaldi_io.read_scp(scp)]
    assert all(not v.flags.writeable and v.dtype == np.float32 for v in views)
    want = np.concatenate([m for _, m in mats], 0)
    for threads in (1, 3, 8, 64):
        dst = np.full(want.shape, np.nan, np.float32)
        offs = hip.host_gather(dst.ctypes.data, views, threads)
        np.testing.assert_array_equal(dst, want)
        assert offs.tolist() == np.concatenate([[0], np.cumsum([m.nbytes for _, m in mats])[:-1]]).tolist()
    assert hip.host_gather(np.empty(4, np.float32).ctypes.data, [], 4).size == 0
    # the batch object stands where collate's padded tensor would
    pb = PackedBatch(views[:5])
    utts, feats, _, ratios, _ = collate([(u, m, [1]) for u, m in mats[:5]], padding_idx=0)
    assert pb.shape == tuple(feats.shape) and torch.equal(pb.ratios(), ratios) and torch.equal(pb.padded(0.0), feats)


def test_fp16_score_check_refuses_non_finite_scores():
    """Second line of the fp16 engine's range guard (hip.check_fp16_range): a NaN / infinite hypothesis score is an error, never a result."""
    hip.check_fp16_range(np.array([-3.5, -120.25]))
    for bad in (np.nan, np.inf, -np.inf):
        with pytest.raises(hip.HipError, match="half-precision range"):
            hip.check_fp16_range(np.array([-1.0, bad]))
    assert hip.PRECISION["fp16"] == 4 and hip.lib_for("fp16") is hip.lib("f16") and hip.lib_for("bf16x3") is hip.lib()
