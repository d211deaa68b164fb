import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


@pytest.fixture(scope="session")
def golden():
    return load_golden


# The seeded inputs every golden fixture was generated from (oracle/make_goldens.py).
def tiny_case(**overrides):
    from cassnat_asr_public_amd import synth

    args = synth.make_args("tiny", **overrides)
    state = synth.make_state(args, seed=0, gain=2.0)
    feats, sizes = synth.make_feats(3, 61, 80, lengths=[61, 50, 37], seed=11)
    return args, state, feats, sizes


def config1_case():
    from cassnat_asr_public_amd import synth

    args = synth.make_args("config1")
    state = synth.make_state(args, seed=1, blank_bias=0.0)
    feats, sizes = synth.make_feats(1, 837, 80, seed=21)
    return args, state, feats, sizes


def config2_b8_case():
    from cassnat_asr_public_amd import synth

    args = synth.make_args("config2")
    state = synth.make_state(args, seed=0, blank_bias=0.35)
    lens = synth.ragged_lengths(8, 1000, 400, seed=7)
    feats, sizes = synth.make_feats(8, 1000, 80, lengths=lens, seed=1234)
    return args, state, feats, sizes


def config5_shape_case():
    """BASELINE configs[4] shape: V = 4230 + 4 characters, same 12L / 1-3-2 model, B=4 ragged."""
    from cassnat_asr_public_amd import synth

    args = synth.make_args("config2", vocab_size=4234)
    state = synth.make_state(args, seed=5, blank_bias=0.35)
    lens = synth.ragged_lengths(4, 600, 300, seed=9)
    feats, sizes = synth.make_feats(4, 600, 80, lengths=lens, seed=77)
    return args, state, feats, sizes


def conf_tiny_case(seed=3, **overrides):
    """Conformer CASS-NAT (use_conv_enc / use_conv_dec, relative positions), tiny, ragged batch."""
    from cassnat_asr_public_amd import synth

    args = synth.make_args("tiny_conf", **overrides)
    state = synth.make_state(args, seed=seed, gain=2.0)
    feats, sizes = synth.make_feats(3, 61, 80, lengths=[61, 50, 37], seed=11)
    return args, state, feats, sizes


def conf_small_case():
    """The shipped decode YAML's shape at d_model 256: 12 conformer encoder layers, 1 + 1 + 6 conformer decoder blocks."""
    from cassnat_asr_public_amd import synth

    args = synth.make_args("conf_small")
    state = synth.make_state(args, seed=6, blank_bias=0.35)
    lens = synth.ragged_lengths(2, 400, 250, seed=3)
    feats, sizes = synth.make_feats(2, 400, 80, lengths=lens, seed=99)
    return args, state, feats, sizes


def esa_case(which):
    """ESA (sample_num = 4) + TransformerLM ranking cases of oracle/make_goldens.py."""
    from cassnat_asr_public_amd import synth

    preset, lmp, (B, T, lens) = {"esa_tiny": ("tiny", "tiny_lm", (3, 61, [61, 50, 37])),
                                 "esa_config2": ("config2", "lm_small", (2, 300, [300, 231])),
                                 "esa_conf_tiny": ("tiny_conf", "tiny_lm", (3, 61, [61, 50, 37]))}[which]
    args = synth.make_args(preset, sample_num=4, threshold=0.9, rank_model="lm")
    lm_args = synth.make_args_lm(lmp, vocab_size=args.vocab_size)
    state = synth.make_state(args, seed=0, gain=2.0) if preset.startswith("tiny") else synth.make_state(args, seed=0, blank_bias=0.35)
    lm_state = synth.make_state(lm_args, seed=9, gain=2.0)
    feats, sizes = synth.make_feats(B, T, 80, lengths=lens, seed=11)
    return args, lm_args, state, lm_state, feats, sizes


def ctcbeam_case(which):
    """decode_type ctc_only / ctc_att cases of oracle/make_goldens.py ('ctcbeam' group)."""
    from cassnat_asr_public_amd import synth

    preset, (B, T, lens), cfg = {"ctcbeam_tiny": ("tiny", (3, 61, [61, 50, 37]), dict(ctc_beam=5, ctc_pruning=8, ctc_lp=0.2)),
                                 "ctcbeam_config2": ("config2", (2, 300, [300, 231]), dict(ctc_beam=10, ctc_pruning=15, ctc_lp=0.0))}[which]
    args = synth.make_args(preset, decode_type="ctc_att", sample_num=1, ctc_lm_weight=0, **cfg)
    state = synth.make_state(args, seed=0, gain=2.0) if preset == "tiny" else synth.make_state(args, seed=0, blank_bias=0.35)
    feats, sizes = synth.make_feats(B, T, 80, lengths=lens, seed=11)
    return args, state, feats, sizes


def config2_b32_case():
    """The benchmark workload (bench.py): B=32 x 1000 frames, blank bias 0.9."""
    from cassnat_asr_public_amd import synth

    args = synth.make_args("config2")
    state = synth.make_state(args, seed=0, blank_bias=synth.BENCH_BLANK_BIAS)
    feats, sizes = synth.make_feats(32, 1000, 80, seed=1234)
    return args, state, feats, sizes


def ast_tiny_case(**overrides):
    from cassnat_asr_public_amd import synth

    args = synth.make_args_ast("tiny_ast", beam_width=3, ctc_beam=5, max_decode_ratio=0.75, **overrides)
    state = synth.make_state(args, seed=3, gain=2.0)
    # every utterance keeps >= 13 valid subsampled frames for the 12 decode steps: once a hypothesis has more tokens than
    # frames its CTC scores are +-1e10 dominated, all candidates tie in float32 and torch.topk's tie order decides
    feats, _ = synth.make_feats(3, 61, 80, lengths=[61, 57, 51], seed=11)
    return args, state, feats


def ast_config4_case(**overrides):
    from cassnat_asr_public_amd import synth

    args = synth.make_args_ast("config4", max_decode_ratio=0.3, **overrides)
    state = synth.make_state(args, seed=5)
    feats, _ = synth.make_feats(2, 400, 80, lengths=[400, 333], seed=31)
    return args, state, feats


# ---- the branches closed in round 4 (oracle/make_goldens.py, group 'branches')
def notrigger_case(which):
    """use_trigger = False (src/models/cassnat.py:469-473)."""
    from cassnat_asr_public_amd import synth

    preset, (B, T, lens) = {"tiny_notrigger": ("tiny", (3, 61, [61, 50, 37])), "config2_notrigger": ("config2", (2, 300, [300, 231]))}[which]
    args = synth.make_args(preset, use_trigger=False)
    state = synth.make_state(args, seed=0, gain=2.0) if preset == "tiny" else synth.make_state(args, seed=0, blank_bias=0.35)
    feats, sizes = synth.make_feats(B, T, 80, lengths=lens, seed=11)
    return args, state, feats, sizes


def esa_beam3_case():
    """ESA (sample_num 4, TransformerLM ranking) finished with beam_width 3."""
    from cassnat_asr_public_amd import synth

    args = synth.make_args("tiny", sample_num=4, threshold=0.9, rank_model="lm", beam_width=3, length_penalty=0.1)
    lm_args = synth.make_args_lm("tiny_lm", vocab_size=args.vocab_size)
    state, lm_state = synth.make_state(args, seed=0, gain=2.0), synth.make_state(lm_args, seed=9, gain=2.0)
    feats, sizes = synth.make_feats(3, 61, 80, lengths=[61, 50, 37], seed=11)
    return args, lm_args, state, lm_state, feats, sizes


def art_case(which, beam_width):
    """ArtTask decode_type 'ctc_only' / 'ctc_correct' on the autoregressive model."""
    from cassnat_asr_public_amd import synth

    preset, (B, T, lens), seed, fseed = {"art_tiny": ("tiny_ast", (3, 61, [61, 57, 51]), 3, 11),
                                         "art_config4": ("config4", (2, 400, [400, 333]), 5, 31)}[which]
    args = synth.make_args_ast(preset, beam_width=beam_width, ctc_beam=5, ctc_pruning=8, ctc_lp=0.2, ctc_lm_weight=0, length_penalty=0.1,
                               use_gpu=False, lm_weight=0)
    state = synth.make_state(args, seed=seed, gain=2.0) if preset == "tiny_ast" else synth.make_state(args, seed=seed)
    feats, sizes = synth.make_feats(B, T, 80, lengths=lens, seed=fseed)
    return args, state, feats, sizes
