"""Waveform -> padded log-mel filterbank batch on the GPU (csrc/fbank.hip through ``cn_fbank``).

The reference reads fbank features that Kaldi's ``compute-fbank-feats`` wrote (egs/librispeech/conf/fbank.conf:1-6) and
normalises them in ``SpeechDataset._load_cmvn`` (src/data/speech_loader.py:109-115); this module is that front-end for
callers that start from audio: ``Fbank()(waves) -> (feats (B, T, 80) float32 cuda, feat_sizes (B,) float32)`` in the
layout ``CassNAT.beam_decode`` takes (padded frames are exactly ``pad_value``; sizes are length ratios as the
reference's collate produces them, speech_loader.py:327-356).
"""
import numpy as np
import torch

from .. import hip

WINDOWS = {"hamming": 0, "povey": 1, "hanning": 2, "rectangular": 3}


class Fbank:
    def __init__(self, cmvn_mean=None, cmvn_std=None, pad_value=0.0, device=None, **opts):
        """opts: sample_rate, frame_length_ms, frame_shift_ms, preemph, low_freq, high_freq, num_mel, window, remove_dc, ..."""
        self.L = hip.lib()
        self.o = hip.CnFbankOpts()
        self.L.cn_fbank_default_opts(self.o)
        if "window" in opts:
            opts["window_type"] = WINDOWS[opts.pop("window")]
        for k, v in opts.items():
            if not hasattr(self.o, k):
                raise TypeError(f"unknown fbank option {k}")
            setattr(self.o, k, v)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.pad_value = float(pad_value)
        self.mean = self.istd = None
        if cmvn_mean is not None:
            self.mean = torch.as_tensor(np.asarray(cmvn_mean, np.float32)).to(self.device)
            self.istd = torch.as_tensor((1.0 / np.asarray(cmvn_std, np.float64)).astype(np.float32)).to(self.device)

    def num_frames(self, num_samples):
        return int(self.L.cn_fbank_num_frames(self.o, int(num_samples)))

    def __call__(self, waves):
        """waves: list of 1-D arrays / tensors on the int16 scale (what Kaldi reads from a wav file)."""
        ns = [int(len(w)) for w in waves]
        B, max_s = len(waves), max(ns)
        frames = [self.num_frames(n) for n in ns]
        T = max(frames)
        if T == 0:
            raise ValueError("every waveform is shorter than one analysis window")
        host = np.zeros((B, max_s), np.float32)
        for b, w in enumerate(waves):
            host[b, : ns[b]] = np.asarray(w.cpu() if isinstance(w, torch.Tensor) else w, dtype=np.float32)
        wave_d = torch.from_numpy(host).to(self.device)
        ns_d = torch.tensor(ns, dtype=torch.int32, device=self.device)
        feats = torch.empty(B, T, self.o.num_mel, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            hip.check(self.L.cn_fbank(self.o, hip._ptr(wave_d), hip._ptr(ns_d), B, max_s, hip._ptr(self.mean) if self.mean is not None else None,
                                      hip._ptr(self.istd) if self.istd is not None else None, hip._ptr(feats), T, self.pad_value,
                                      hip.current_stream()), "cn_fbank")
        sizes = torch.tensor([f / T for f in frames], dtype=torch.float32)
        return feats, sizes
