"""ORACLE - test infrastructure only.  CPU restatement (numpy, float64) of Kaldi's `compute-fbank-feats` as the
reference's recipe configures it (egs/librispeech/conf/fbank.conf:1-6: hamming window, 16 kHz, 80 mel bins, no energy;
every other option at Kaldi's default).

PARITY UNPINNED: Kaldi is a third-party dependency that is not under /root/reference (no version is pinned by the
recipe, no script of this fork invokes it, and the reference's tests hold no fbank vectors), so this file restates the
PUBLISHED algorithm of kaldi-asr/kaldi `src/feat/` (feature-window.cc: ExtractWindow / ProcessWindow, feature-fbank.cc:
FbankComputer::Compute, mel-computations.cc: MelBanks::MelBanks) and the HIP kernel is validated against this file
only.  Kaldi's default dither = 1.0 adds random noise per sample; it is 0 here (deterministic), as it must be for any
comparison.
"""
import numpy as np

DEFAULTS = dict(sample_rate=16000.0, frame_length_ms=25.0, frame_shift_ms=10.0, preemph=0.97, remove_dc=True,
                window="hamming", num_mel=80, low_freq=20.0, high_freq=0.0, use_power=True, use_log=True, snip_edges=True)


def mel_scale(f):
    return 1127.0 * np.log(1.0 + np.asarray(f, dtype=np.float64) / 700.0)


def mel_banks(num_mel, n_fft, sample_rate, low_freq, high_freq):
    """MelBanks::MelBanks (no VTLN): returns a list of (first fft bin, weights) per mel bin over fft bins 0..n_fft/2-1."""
    nyquist = 0.5 * sample_rate
    if high_freq <= 0.0:
        high_freq += nyquist
    n_bins = n_fft // 2
    fft_bin_width = sample_rate / n_fft
    mel_lo, mel_hi = mel_scale(low_freq), mel_scale(high_freq)
    delta = (mel_hi - mel_lo) / (num_mel + 1)
    banks = []
    for b in range(num_mel):
        left, center, right = mel_lo + b * delta, mel_lo + (b + 1) * delta, mel_lo + (b + 2) * delta
        first, w = -1, []
        for i in range(n_bins):
            mel = mel_scale(fft_bin_width * i)
            if left < mel < right:
                w.append((mel - left) / (center - left) if mel <= center else (right - mel) / (right - center))
                if first < 0:
                    first = i
        banks.append((first, np.asarray(w, dtype=np.float64)))
    return banks


def window_fn(kind, n):
    i = np.arange(n, dtype=np.float64)
    a = 2.0 * np.pi / (n - 1)
    if kind == "hamming":
        return 0.54 - 0.46 * np.cos(a * i)
    if kind == "hanning":
        return 0.5 - 0.5 * np.cos(a * i)
    if kind == "povey":
        return (0.5 - 0.5 * np.cos(a * i)) ** 0.85
    if kind == "rectangular":
        return np.ones(n)
    raise ValueError(kind)


def num_frames(n_samples, frame_length, frame_shift):
    """snip_edges = true (feature-window.cc: NumFrames)."""
    return 0 if n_samples < frame_length else 1 + (n_samples - frame_length) // frame_shift


def fbank(wave, **opts):
    """wave: 1-D array on the int16 scale (as Kaldi reads a wav) -> (frames, num_mel) log-mel energies, float64."""
    o = dict(DEFAULTS)
    o.update(opts)
    sr = o["sample_rate"]
    flen, fshift = int(sr * 0.001 * o["frame_length_ms"]), int(sr * 0.001 * o["frame_shift_ms"])
    n_fft = 1
    while n_fft < flen:
        n_fft *= 2
    wave = np.asarray(wave, dtype=np.float64)
    T = num_frames(len(wave), flen, fshift)
    win = window_fn(o["window"], flen)
    banks = mel_banks(o["num_mel"], n_fft, sr, o["low_freq"], o["high_freq"])
    out = np.zeros((T, o["num_mel"]))
    eps = float(np.finfo(np.float32).eps)
    for t in range(T):
        fr = wave[t * fshift : t * fshift + flen].copy()
        if o["remove_dc"]:
            fr -= fr.sum() / flen
        if o["preemph"] != 0.0:  # feature-window.cc Preemphasize: back to front, sample 0 uses itself
            fr[1:] -= o["preemph"] * fr[:-1].copy()
            fr[0] -= o["preemph"] * fr[0]
        fr *= win
        spec = np.fft.rfft(fr, n_fft)
        power = spec.real**2 + spec.imag**2  # bins 0 .. n_fft/2
        if not o["use_power"]:
            power = np.sqrt(power)
        for b, (first, w) in enumerate(banks):
            out[t, b] = np.dot(w, power[first : first + len(w)])
        if o["use_log"]:
            out[t] = np.log(np.maximum(out[t], eps))
    return out
