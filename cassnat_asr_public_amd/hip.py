"""ctypes binding of libcassnat_hip.so (the C ABI declared in include/cassnat_hip.h).

The product path has no CPU fallback: if the library is missing or a call fails, this raises.
"""
import ctypes as C
import os
import re

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# (CASSNAT_HIP_LIB: another build of the same library - A/B measurements of kernel variants, tools/scripts/ab_bench.sh)
LIB_PATH = os.environ.get("CASSNAT_HIP_LIB") or os.path.join(_HERE, "libcassnat_hip.so")
# the second build of the same sources (csrc/common.h: -DCN_OP16_F16): the fast engine with IEEE half-precision MFMA operands -
# --hip_precision fp16.  Same C ABI; an engine keeps the library it was created in.
LIB_PATH_F16 = os.path.join(_HERE, "libcassnat_hip_f16.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "cassnat_hip.h")

PRECISION = {"fp32": 0, "f32": 0, "float32": 0, "bf16": 1, "bfloat16": 1, "fp8": 2, "bf16x3": 3, "fp16": 4, "float16": 4}
DTYPES = {0: np.float32, 1: np.int32, 2: np.uint8, 3: np.float64}
FP8_SCOPE_BITS = {"conv2": 1, "linear": 2, "ffn": 4}


def parse_fp8_scope(text):
    """``--hip_fp8_scope``: which encoder-side products of the fp8 engine take e4m3 operands - "all" or a "+"-joined list
    of ``conv2``, ``linear`` (needs conv2) and ``ffn`` / ``ffn:N`` (the feed-forward products of the encoder layers >= N).
    Returns (cn_config.fp8_scope, cn_config.fp8_ffn_first_layer)."""
    text = (text or "all").strip().lower()
    if text == "all":
        return 0, 0
    scope, first = 0, 0
    for part in text.split("+"):
        name, _, arg = part.strip().partition(":")
        if name not in FP8_SCOPE_BITS or (arg and name != "ffn"):
            raise ValueError(f"hip_fp8_scope: unknown part {part!r} (all, or conv2 / linear / ffn[:first layer] joined by +)")
        scope |= FP8_SCOPE_BITS[name]
        if arg:
            first = int(arg)
            if first < 0:
                raise ValueError("hip_fp8_scope: ffn:N needs N >= 0")
    if scope & 2 and not scope & 1:
        raise ValueError("hip_fp8_scope: linear needs conv2 (conv2 hands its rows to linear_out in e4m3)")
    return scope, first


class CnConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "input_size", "d_model", "n_head", "d_encff", "d_decff", "n_enc", "n_extra", "n_self_dec", "n_mix_dec",
        "vocab_size", "precision", "max_batch", "max_frames", "device", "ast", "conf_enc", "conf_dec", "enc_max_rel", "dec_max_rel",
        "enc_kernel", "dec_kernel", "d_ff", "esa_group", "fp8_scope", "fp8_ffn_first_layer")]


class CnDecodeOpts(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "padding_idx", "sos", "left_trigger", "right_trigger", "src_trigger", "use_unimask", "beam_width",
        "capture", "sub_batch", "no_trigger")] + [("reserved", C.c_int32 * 6)]


class CnAstOpts(C.Structure):
    _fields_ = [("ctc_weight", C.c_float), ("temperature", C.c_float), ("ctc_beam", C.c_int32), ("beam_width", C.c_int32),
                ("max_step", C.c_int32), ("eos", C.c_int32), ("use_length_penalty", C.c_int32), ("one_minus_ctc_weight", C.c_float),
                ("length_penalty", C.c_double), ("reserved", C.c_int32 * 4)]


class CnFbankOpts(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("sample_rate", "frame_length_ms", "frame_shift_ms", "preemph", "low_freq", "high_freq")] + \
               [(n, C.c_int32) for n in ("num_mel", "window_type", "remove_dc", "use_power", "use_log")] + \
               [("reserved", C.c_int32 * 5)]


class HipError(RuntimeError):
    pass


_libs = {}


def declared_symbols():
    """Every function name include/cassnat_hip.h declares."""
    with open(HEADER_PATH) as f:
        text = f.read()
    return sorted(set(re.findall(r"\b(cn_[a-z0-9_]+)\s*\(", text)) - {"cn_model", "cn_config", "cn_decode_opts"})


def lib_for(precision):
    """The library that holds engines of ``precision`` (a name of ``PRECISION`` or its number)."""
    code = PRECISION[precision] if isinstance(precision, str) else int(precision)
    return lib("f16" if code == PRECISION["fp16"] else None)


def lib(flavour=None):
    """Load the shared library (built by __graft_entry__.build / cassnat_asr_public_amd.build).  ``flavour`` "f16": the build with
    half-precision operands (libcassnat_hip_f16.so) - everything that is not an fp16 engine uses the default library."""
    path = LIB_PATH_F16 if flavour == "f16" else LIB_PATH
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise HipError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(there is no CPU fallback for the CASS-NAT hot path)")
    L = C.CDLL(path)
    L.cn_last_error.restype = C.c_char_p
    L.cn_version.restype = C.c_char_p
    L.cn_operand16.restype = C.c_char_p
    L.cn_model_destroy.restype = None
    for name in declared_symbols():
        fn = getattr(L, name)  # AttributeError here = header and library disagree
        if name not in ("cn_last_error", "cn_version", "cn_operand16", "cn_model_destroy"):
            fn.restype = C.c_int
    L.cn_op_gemm.argtypes = [C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                             C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32,
                             C.c_float, C.c_void_p]
    L.cn_op_convert.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]
    L.cn_op_conv1.argtypes = [C.c_int32] + [C.c_void_p] * 4 + [C.c_int32] * 4 + [C.c_void_p]
    L.cn_op_conv1_bordered.argtypes = [C.c_void_p] * 4 + [C.c_int32] * 4 + [C.c_void_p]
    L.cn_op_conv2.argtypes = [C.c_int32] + [C.c_void_p] * 4 + [C.c_int32] * 4 + [C.c_void_p]
    L.cn_op_conv_frontend_fp8.argtypes = [C.c_void_p] * 7 + [C.c_int32] * 4 + [C.c_float, C.c_float, C.c_void_p, C.c_void_p]
    L.cn_op_conv_frontend_mix.argtypes = [C.c_void_p] * 7 + [C.c_int32] * 4 + [C.c_void_p]
    L.cn_op_linear256_fp8.argtypes = [C.c_void_p] * 4 + [C.c_int32] * 2 + [C.c_float, C.c_float, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
    L.cn_op_layernorm.argtypes = [C.c_int32] + [C.c_void_p] * 4 + [C.c_int32, C.c_int32, C.c_float, C.c_void_p]
    L.cn_op_attention.argtypes = [C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32,
                                  C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p,
                                  C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_void_p]
    L.cn_op_logsoftmax_argmax.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    L.cn_op_ctc_align.argtypes = [C.c_void_p] * 3 + [C.c_int32] * 5 + [C.c_void_p] * 6
    L.cn_op_greedy_pack.argtypes = [C.c_void_p] * 3 + [C.c_int32] * 4 + [C.c_void_p] * 4
    L.cn_op_ctc_prefix_beam.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_double,
                                        C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p]
    L.cn_op_ctc_viterbi.argtypes = [C.c_void_p] * 5 + [C.c_int32] * 6 + [C.c_void_p, C.c_void_p]
    L.cn_op_topk.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.cn_op_gemm_fp8.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_float,
                                 C.c_int32, C.POINTER(C.c_float), C.c_void_p]
    L.cn_op_cmvn.argtypes = [C.c_void_p] * 4 + [C.c_int32] * 3 + [C.c_void_p]
    L.cn_host_gather.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]
    L.cn_op_unpack_rows.argtypes = [C.c_void_p] * 4 + [C.c_int32] * 3 + [C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
    L.cn_op_quantize_fp8.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_void_p]
    L.cn_op_logsoftmax_topk.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.cn_model_create.argtypes = [C.POINTER(CnConfig), C.POINTER(C.c_void_p)]
    L.cn_model_create_shared.argtypes = [C.POINTER(CnConfig), C.c_void_p, C.POINTER(C.c_void_p)]
    L.cn_model_destroy.argtypes = [C.c_void_p]
    L.cn_model_load_weights.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.POINTER(C.c_int64), C.c_int32]
    L.cn_model_load_pe.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
    L.cn_model_finalize.argtypes = [C.c_void_p]
    L.cn_model_weight_blob.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]
    L.cn_decode_nast.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                 C.POINTER(CnDecodeOpts), C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.cn_decode_nast_merged.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(CnDecodeOpts),
                                        C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int32, C.c_void_p, C.c_int32,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]
    L.cn_decode_ticket.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.cn_take_range_fault.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_float)]
    L.cn_encode_align.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                  C.POINTER(CnDecodeOpts), C.POINTER(C.c_int32), C.c_void_p]
    L.cn_fetch.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_int32),
                           C.POINTER(C.c_int32)]
    L.cn_op_ffn_fused.argtypes = [C.c_void_p] * 10 + [C.c_int32, C.c_int32, C.c_float, C.c_int32, C.c_void_p]
    L.cn_op_ffn_x3.argtypes = [C.c_void_p] * 10 + [C.c_int32, C.c_int32, C.c_float, C.c_int32, C.c_void_p]
    L.cn_op_x3_chain.argtypes = [C.c_void_p] * 16 + [C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_int32, C.c_void_p]
    L.cn_op_chain.argtypes = ([C.c_void_p, C.c_void_p, C.c_int32] + [C.c_void_p] * 13 +
                              [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_int32, C.c_void_p])
    L.cn_op_genmax.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.cn_op_genmax_gather.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int32,
                                      C.c_void_p, C.c_void_p]
    L.cn_op_genmax_x3.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    L.cn_op_proj_x3.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                C.c_void_p]
    L.cn_ast_begin.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(CnDecodeOpts), C.c_int32,
                               C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
    L.cn_ast_step.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                              C.c_float, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.cn_fbank_default_opts.argtypes = [C.POINTER(CnFbankOpts)]
    L.cn_fbank_default_opts.restype = None
    L.cn_fbank_num_frames.argtypes = [C.POINTER(CnFbankOpts), C.c_int32]
    L.cn_fbank_num_frames.restype = C.c_int32
    L.cn_fbank.argtypes = [C.POINTER(CnFbankOpts), C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                           C.c_int32, C.c_float, C.c_void_p]
    L.cn_esa_begin.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(CnDecodeOpts), C.c_void_p]
    L.cn_esa_sample.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_float, C.c_void_p, C.POINTER(CnDecodeOpts), C.c_void_p,
                                C.c_void_p, C.c_int32, C.c_void_p, C.POINTER(C.c_int32), C.c_int32, C.c_void_p]
    L.cn_lm_score.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    L.cn_decode_ast.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(CnDecodeOpts),
                                C.POINTER(CnAstOpts), C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.cn_ast_ctc_score.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                   C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    L.cn_ctc_beam.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(CnDecodeOpts), C.c_int32,
                              C.c_int32, C.c_double, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                              C.c_void_p]
    L.cn_decode_nast_forced.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(CnDecodeOpts),
                                        C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                        C.c_void_p]
    L.cn_ast_teacher_score.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(CnDecodeOpts), C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    L.cn_ast_ctc_correct.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(CnDecodeOpts), C.c_int32,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32), C.c_void_p]
    L.cn_profile_begin.argtypes = [C.c_void_p, C.c_char_p]
    L.cn_profile_end.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
    want = "fp16" if flavour == "f16" else "bf16"
    if L.cn_operand16().decode() != want:
        raise HipError(f"{path}: the library's 16-bit operand is {L.cn_operand16().decode()}, expected {want}")
    _libs[path] = L
    return L


def check(rc, what="", L=None):
    """``L``: the library the failing call was made in (its cn_last_error() holds the message); default: the bf16 build."""
    if rc != 0:
        msg = (L or lib()).cn_last_error().decode(errors="replace")
        raise HipError(f"{what or 'libcassnat_hip'} failed (rc={rc}): {msg}")


def check_fp16_range(scores, what="decode"):
    """Second line behind ``Engine.check_range`` (which guards what the features' scale drives): a value beyond +-65504 in front of
    a matrix product INSIDE the layers - a matter of the checkpoint, not of the input - becomes an infinity; where that reaches a
    hypothesis score as a NaN the engine fails loudly instead of returning the hypothesis (a ReLU can still turn a NaN into a 0:
    a model whose activations leave the half range belongs on the bf16 engine)."""
    if not np.isfinite(np.asarray(scores, dtype=np.float64)).all():
        raise HipError(f"{what}: non-finite hypothesis score from the fp16 engine - an MFMA operand left the half-precision range "
                       "(+-65504); use --hip_precision bf16 / bf16x3 for this model or these features")


def _ptr(t):
    """Device/host pointer of a torch tensor (or None)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def current_stream():
    import torch

    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def cmvn_(feats, lens, mean, std):
    """In-place global CMVN of a padded (B, T, F) float32 CUDA batch on the current stream: frames t < lens[b] become
    float((double(x) - mean) / std) - SpeechDataset's arithmetic bit for bit (cn_op_cmvn); lens int32, mean / std float64, all
    on the batch's device."""
    B, T, F = feats.shape
    assert feats.is_contiguous() and feats.dtype.is_floating_point and feats.element_size() == 4 and lens.numel() >= B
    check(lib().cn_op_cmvn(_ptr(feats), _ptr(lens), _ptr(mean), _ptr(std), B, T, F, current_stream()), "cn_op_cmvn")
    return feats


def host_gather(dst_ptr, srcs, threads=1):
    """Copy the numpy arrays ``srcs`` (C-contiguous; e.g. read-only views into a memory-mapped archive) back to back to the host
    address ``dst_ptr`` in one GIL-free call (cn_host_gather); returns the byte offsets of the pieces."""
    n = len(srcs)
    ptrs = np.fromiter((a.ctypes.data for a in srcs), np.uint64, n)
    sizes = np.fromiter((a.nbytes for a in srcs), np.uint64, n)
    offs = np.zeros(n, np.uint64)
    np.cumsum(sizes[:-1], out=offs[1:])
    check(lib().cn_host_gather(C.c_void_p(int(dst_ptr)), ptrs.ctypes.data_as(C.c_void_p), offs.ctypes.data_as(C.c_void_p),
                               sizes.ctypes.data_as(C.c_void_p), n, int(threads)), "cn_host_gather")
    return offs


def unpack_rows(packed, off, lens, out, pad, mean=None, std=None):
    """The reader's collate on the device (cn_op_unpack_rows): ``packed`` float32 CUDA rows of F features, utterance r at row off[r]
    with lens[r] frames -> the padded (rows, T, F) batch ``out`` on the current stream; with ``mean`` / ``std`` (float64, (F,)) the
    global CMVN is applied in float64 on the way."""
    rows, T, F = out.shape
    assert out.is_contiguous() and packed.is_contiguous() and off.numel() >= rows and lens.numel() >= rows
    check(lib().cn_op_unpack_rows(_ptr(packed), _ptr(off), _ptr(lens), _ptr(out), rows, T, F, float(pad), _ptr(mean), _ptr(std),
                                  current_stream()), "cn_op_unpack_rows")
    return out


class Engine:
    """Owns one ``cn_model`` handle: weights, workspace and the decode pipeline on one GPU."""

    def __init__(self, args, precision="bf16", max_batch=32, max_frames=2048, device=0, esa_group=1, share_with=None):
        """``share_with``: a finalized Engine of the same model - the new handle uses ITS device copy of the packed weights
        (reference counted in the library) and only allocates a workspace of its own; it is ready to decode."""
        self.L = lib_for(precision)
        ast = int(getattr(args, "ast", 0))
        self.cfg = CnConfig(
            input_size=args.input_size, d_model=args.d_model, n_head=args.n_head, d_encff=args.d_encff,
            d_decff=args.d_decff, n_enc=args.N_enc, n_extra=args.N_extra, n_self_dec=args.N_self_dec,
            n_mix_dec=args.N_mix_dec, vocab_size=args.vocab_size, precision=PRECISION[precision],
            max_batch=max_batch, max_frames=max_frames, device=device, ast=ast,
            conf_enc=int(getattr(args, "conf_enc", 0)), conf_dec=int(getattr(args, "conf_dec", 0)),
            enc_max_rel=int(getattr(args, "enc_max_rel", 0)), dec_max_rel=int(getattr(args, "dec_max_rel", 0)),
            enc_kernel=int(getattr(args, "enc_kernel", 0)), dec_kernel=int(getattr(args, "dec_kernel", 0)),
            d_ff=int(getattr(args, "d_ff", 0)), esa_group=int(esa_group),
            fp8_scope=int(getattr(args, "fp8_scope", 0)), fp8_ffn_first_layer=int(getattr(args, "fp8_ffn_first_layer", 0)))
        self.precision = precision
        self.handle = C.c_void_p()
        if share_with is not None:
            if not share_with.finalized or not share_with.handle:
                raise HipError("share_with needs a finalized engine")
            self._chk(self.L.cn_model_create_shared(C.byref(self.cfg), share_with.handle, C.byref(self.handle)), "cn_model_create_shared")
            self.finalized = True
        else:
            self._chk(self.L.cn_model_create(C.byref(self.cfg), C.byref(self.handle)), "cn_model_create")
            self.finalized = False

    def _chk(self, rc, what=""):
        check(rc, what, self.L)

    def close(self):
        if getattr(self, "handle", None):
            self.L.cn_model_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def load_state(self, state, pe_table):
        """state: name -> float32 array/tensor (reference parameter names); pe_table: (rows, d_model)."""
        for name, w in state.items():
            a = np.ascontiguousarray(w.detach().cpu().numpy() if hasattr(w, "detach") else w, dtype=np.float32)
            shape = (C.c_int64 * a.ndim)(*a.shape)
            self._chk(self.L.cn_model_load_weights(self.handle, name.encode(), a.ctypes.data_as(C.c_void_p), shape, a.ndim),
                  f"load {name}")
        pe = np.ascontiguousarray(pe_table.detach().cpu().numpy() if hasattr(pe_table, "detach") else pe_table,
                                  dtype=np.float32)
        self._chk(self.L.cn_model_load_pe(self.handle, pe.ctypes.data_as(C.c_void_p), pe.shape[0]), "load pe")
        self.finalize()

    def finalize(self):
        self._chk(self.L.cn_model_finalize(self.handle), "cn_model_finalize")
        self.finalized = True

    def weight_blob(self):
        """(device pointer, bytes) of the packed weights - what RCCL broadcasts from rank 0."""
        p, n = C.c_void_p(), C.c_int64()
        self._chk(self.L.cn_model_weight_blob(self.handle, C.byref(p), C.byref(n)), "cn_model_weight_blob")
        return p.value, n.value

    @staticmethod
    def make_opts(args, capture=False):
        # (an autoregressive model's args carry no trigger options: ctc_beam_decode runs on either kind of model)
        return CnDecodeOpts(padding_idx=int(args.padding_idx), sos=1, left_trigger=int(getattr(args, "left_trigger", 0)),
                            right_trigger=int(getattr(args, "right_trigger", 0)), src_trigger=int(bool(getattr(args, "src_trigger", False))),
                            use_unimask=int(bool(getattr(args, "use_unimask", False))), beam_width=int(args.beam_width),
                            capture=int(bool(capture)), no_trigger=int(not getattr(args, "use_trigger", True)))

    def decode(self, feats, size_ratio, opts, hyp, hyp_len, score):
        """feats (B,T,F) f32 cuda, size_ratio (B,) f32 cuda; outputs are caller-owned cuda tensors."""
        B, T, F = feats.shape
        self._chk(self.L.cn_decode_nast(self.handle, _ptr(feats), _ptr(size_ratio), B, T, F, C.byref(opts), _ptr(hyp),
                                    hyp.shape[1], _ptr(hyp_len), _ptr(score), current_stream()), "cn_decode_nast")

    def decode_merged(self, feats, size_ratio, opts, sub_rows, sub_frames, hyp, hyp_len, score, u_hint=0):
        """One engine pass over several reference batches (``cn_decode_nast_merged``): ``feats`` (B, T, F) holds the batches one
        after the other, batch k = ``sub_rows[k]`` utterances of ``sub_frames[k]`` <= T frames (padded to T with padding_idx
        frames); every utterance's result is that of a pass of its own batch.  ``sub_rows`` empty / None: a plain call.
        ``u_hint`` > 0: no mid-pass host sync - the decoder side runs on that many rows; returns the ticket whose
        ``ticket(t) -> (ymax, rows_used)`` the caller reads once the stream has drained (rows_used < ymax: decode again)."""
        B, T, F = feats.shape
        n = len(sub_rows) if sub_rows is not None else 0
        rows = (C.c_int32 * max(1, n))(*[int(x) for x in (sub_rows or [])])
        frames = (C.c_int32 * max(1, n))(*[int(x) for x in (sub_frames or [])])
        ticket = C.c_int32(-1)
        self._chk(self.L.cn_decode_nast_merged(self.handle, _ptr(feats), _ptr(size_ratio), B, T, F, C.byref(opts), n, rows, frames,
                                           int(u_hint), _ptr(hyp), hyp.shape[1], _ptr(hyp_len), _ptr(score), current_stream(),
                                           C.byref(ticket)), "cn_decode_nast_merged")
        return ticket.value

    def check_range(self, what="decode"):
        """fp16 engines: raise if a pass since the last call saw features beyond the range the half-precision operands hold
        (``cn_take_range_fault``; call it once the passes' results are on the host).  Other engines: nothing to check."""
        if self.precision not in ("fp16", "float16", "bf16x3"):
            return
        fault, limit = C.c_int32(), C.c_float()
        self._chk(self.L.cn_take_range_fault(self.handle, C.byref(fault), C.byref(limit)), "cn_take_range_fault")
        if fault.value and self.precision == "bf16x3":
            raise HipError(f"{what}: features beyond the range the split-bf16 engine's mixed-arithmetic convolution holds at its "
                           f"tolerance (|x| > {limit.value:.4g} lets conv1 outputs pass the e4m3 operands' 448): normalise the features "
                           "(CMVN) or decode with --hip_precision fp32")
        if fault.value:
            raise HipError(f"{what}: features beyond the fp16 engine's half-precision range (|x| > {limit.value:.4g} lets a subsampling "
                           "convolution's output pass 65504 / 2): normalise the features (CMVN) or decode with --hip_precision bf16 / bf16x3")

    def ticket(self, t):
        """(true row count, rows the decoder side ran on) of the pass that returned ticket ``t``; valid once its stream work is done."""
        ymax, used = C.c_int32(), C.c_int32()
        self._chk(self.L.cn_decode_ticket(self.handle, int(t), C.byref(ymax), C.byref(used)), "cn_decode_ticket")
        return ymax.value, used.value

    def encode_align(self, feats, size_ratio, opts):
        B, T, F = feats.shape
        ymax = C.c_int32()
        self._chk(self.L.cn_encode_align(self.handle, _ptr(feats), _ptr(size_ratio), B, T, F, C.byref(opts),
                                     C.byref(ymax), current_stream()), "cn_encode_align")
        return ymax.value

    # ---- autoregressive (AST) path
    def ast_begin(self, feats, opts, want_ctc, max_len, max_slots, ctc_beam):
        B, T, F = feats.shape
        self._chk(self.L.cn_ast_begin(self.handle, _ptr(feats), B, T, F, C.byref(opts), int(want_ctc), max_len, max_slots,
                                  ctc_beam, current_stream()), "cn_ast_begin")

    def ast_step(self, pos, tok, utt, anc, keyok, temperature, K, topk_idx, topk_val):
        self._chk(self.L.cn_ast_step(self.handle, tok.shape[0], pos, _ptr(tok), _ptr(utt), _ptr(anc), _ptr(keyok), anc.shape[1],
                                 float(temperature), K, _ptr(topk_idx), _ptr(topk_val), current_stream()), "cn_ast_step")

    def ast_ctc_score(self, out_len, utt, last_tok, cand, prev_ref, parity, eos, score):
        self._chk(self.L.cn_ast_ctc_score(self.handle, cand.shape[0], out_len, _ptr(utt), _ptr(last_tok), _ptr(cand), cand.shape[1],
                                      _ptr(prev_ref), parity, eos, _ptr(score), current_stream()), "cn_ast_ctc_score")

    # ---- ESA sampling + LM ranking
    def esa_begin(self, feats, opts):
        B, T, F = feats.shape
        self._chk(self.L.cn_esa_begin(self.handle, _ptr(feats), B, T, F, C.byref(opts), current_stream()), "cn_esa_begin")

    def esa_sample(self, select, threshold, size_ratio, opts, tok, val, ylen, force_U=0):
        """One pass over n sampled alignments per utterance (n <= cfg.esa_group): select uint8 (n, B, T') cuda draws (all-zero
        = the best path) -> rows (U) of this pass; tok / val (n, B, stride), ylen (n, B).  force_U > 0: decode on that many
        rows; force_U = -1: only count the rows (tok / val / ylen may be None)."""
        ymax = C.c_int32()
        n = select.shape[0]
        assert select.is_contiguous()
        if force_U >= 0 and opts.beam_width == 1:  # (beam_width > 1: the per-row top-k stays in the engine, nothing is copied out)
            assert tok.shape[0] == n and tok.is_contiguous() and val.is_contiguous() and ylen.is_contiguous()
        self._chk(self.L.cn_esa_sample(self.handle, _ptr(select), n, float(threshold), _ptr(size_ratio), C.byref(opts), _ptr(tok),
                                   _ptr(val), tok.shape[2] if tok is not None else 0, _ptr(ylen), C.byref(ymax), int(force_U),
                                   current_stream()), "cn_esa_sample")
        return ymax.value

    def lm_score(self, tok, tgt, length, U, score):
        """tok / tgt int32 (N, ld), length int32 (N,), score float32 (N, ld): log p(tgt[n][u] | tok[n][..u]) for u < U."""
        self._chk(self.L.cn_lm_score(self.handle, _ptr(tok), _ptr(tgt), _ptr(length), tok.shape[0], int(U), tok.shape[1], _ptr(score),
                                 current_stream()), "cn_lm_score")

    def ast_decode(self, feats, opts, ast_opts, hyp, hyp_len, score):
        """Whole beam search on the device: hyp int32 (B, beam, max_len), hyp_len int32 (B, beam), score float64 (B, beam)."""
        B, T, F = feats.shape
        self._chk(self.L.cn_decode_ast(self.handle, _ptr(feats), B, T, F, C.byref(opts), C.byref(ast_opts), _ptr(hyp),
                                   hyp.shape[2], _ptr(hyp_len), _ptr(score), current_stream()), "cn_decode_ast")

    # ---- decode_type ctc_only / ctc_att, rank_model at_baseline
    def ctc_beam(self, feats, size_ratio, opts, beam, pruning, length_penalty):
        """CTC prefix beam search on the device -> (hyp (B, beam, T'+1) int32, hyp_len (B, beam) int32, score_ctc / p_blk /
        p_nblk (B, beam) float64, nbeam (B,) int32) as cuda tensors; hypotheses carry no sos, best first."""
        import torch

        B, T, F = feats.shape
        cap = ((T - 1) // 2 + 1 - 1) // 2 + 1 + 1
        dev = feats.device
        hyp = torch.empty(B, beam, cap, dtype=torch.int32, device=dev)
        hlen = torch.empty(B, beam, dtype=torch.int32, device=dev)
        sc, pb, pnb = (torch.empty(B, beam, dtype=torch.float64, device=dev) for _ in range(3))
        nb = torch.empty(B, dtype=torch.int32, device=dev)
        self._chk(self.L.cn_ctc_beam(self.handle, _ptr(feats), _ptr(size_ratio), B, T, F, C.byref(opts), int(beam), int(pruning),
                                 float(length_penalty), _ptr(hyp), cap, _ptr(hlen), _ptr(sc), _ptr(pb), _ptr(pnb), _ptr(nb),
                                 current_stream()), "cn_ctc_beam")
        return hyp, hlen, sc, pb, pnb, nb

    def decode_forced(self, feats, size_ratio, opts, labels, label_len, max_label_len, hyp, hyp_len, score):
        B, T, F = feats.shape
        self._chk(self.L.cn_decode_nast_forced(self.handle, _ptr(feats), _ptr(size_ratio), B, T, F, C.byref(opts), _ptr(labels),
                                           _ptr(label_len), labels.shape[1], int(max_label_len), _ptr(hyp), hyp.shape[1],
                                           _ptr(hyp_len), _ptr(score), current_stream()), "cn_decode_nast_forced")

    def ast_teacher_score(self, feats, opts, tok, tgt, length, n_per_utt, U, score):
        B, T, F = feats.shape
        self._chk(self.L.cn_ast_teacher_score(self.handle, _ptr(feats), B, T, F, C.byref(opts), _ptr(tok), _ptr(tgt), _ptr(length),
                                          int(n_per_utt), int(U), tok.shape[1], _ptr(score), current_stream()), "cn_ast_teacher_score")

    def ast_ctc_correct(self, feats, opts, k):
        """Transformer.fast_decode_with_ctc's device half: -> (length (B,) int32 cuda, tok (B, U, k) int32 cuda, val (B, U, k) f32 cuda)."""
        import torch

        B, T, F = feats.shape
        Tp = ((T - 1) // 2 + 1 - 1) // 2 + 1
        tok = torch.empty(B * (Tp + 1) * k, dtype=torch.int32, device=feats.device)
        val = torch.empty(B * (Tp + 1) * k, dtype=torch.float32, device=feats.device)
        length = torch.empty(B, dtype=torch.int32, device=feats.device)
        rows = C.c_int32()
        self._chk(self.L.cn_ast_ctc_correct(self.handle, _ptr(feats), B, T, F, C.byref(opts), int(k), _ptr(tok), _ptr(val), _ptr(length),
                                        C.byref(rows), current_stream()), "cn_ast_ctc_correct")
        U = rows.value
        return length, tok[: B * U * k].view(B, U, k), val[: B * U * k].view(B, U, k)

    def profile_begin(self, tags=None):
        """Start HIP-event timing of the tagged kernels (None = all) on the launch stream."""
        self._chk(self.L.cn_profile_begin(self.handle, None if not tags else "|".join(tags).encode()), "cn_profile_begin")

    def profile_end(self):
        """-> {tag: {count, ms, flops, bytes}} accumulated since profile_begin (synchronises the device)."""
        import json

        buf = C.create_string_buffer(1 << 16)
        self._chk(self.L.cn_profile_end(self.handle, buf, len(buf)), "cn_profile_end")
        return json.loads(buf.value.decode())

    def fetch(self, name):
        shape = (C.c_int64 * 4)()
        ndim, dtype = C.c_int32(), C.c_int32()
        self._chk(self.L.cn_fetch(self.handle, name.encode(), None, 0, shape, C.byref(ndim), C.byref(dtype)), f"fetch {name}")
        out = np.empty([shape[i] for i in range(ndim.value)], dtype=DTYPES[dtype.value])
        self._chk(self.L.cn_fetch(self.handle, name.encode(), out.ctypes.data_as(C.c_void_p), out.nbytes, shape,
                              C.byref(ndim), C.byref(dtype)), f"fetch {name}")
        return out
