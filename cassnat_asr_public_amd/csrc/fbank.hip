// Waveform -> log-mel filterbank (+ global CMVN) on the device: the front-end the reference leaves to Kaldi's
// `compute-fbank-feats` (egs/librispeech/conf/fbank.conf:1-6) followed by SpeechDataset._load_cmvn
// (src/data/speech_loader.py:109-115, (feat - mean) / std).  Algorithm = kaldi-asr/kaldi src/feat/feature-window.cc
// (ExtractWindow, ProcessWindow), feature-fbank.cc (FbankComputer::Compute), mel-computations.cc (MelBanks), dither 0.
// Parity is pinned to oracle/fbank_oracle.py only (Kaldi is not in the reference tree; see that file's header).
//
// One wave per frame, four frames per workgroup.  A frame is <= 512 samples: DC removal (wave reduction), pre-emphasis
// and the window are applied on the way into LDS in bit-reversed order, a radix-2 FFT runs in LDS (9 stages x 4
// butterflies per lane, twiddles from a table computed in double on the host), and the triangular mel filters - each a
// contiguous band of FFT bins - are summed one mel bin per lane in ascending bin order (the order of the oracle).
// HBM-bound: 2 bytes/sample in (10 ms hop, 25 ms window: every sample is read 2.5 times, from L2), 320 bytes/frame out.
#include <cmath>
#include <vector>

#include "kernels.h"

constexpr int FB_MAX_FFT = 512;

struct FbankParams {
    const float* wave;        // [B][max_samples]
    const int* num_samples;   // [B]
    float* out;               // [B][Tmax][num_mel]
    const float* window;      // [frame_len]
    const float2* twiddle;    // [n_fft / 2]: exp(-2 pi i k / n_fft)
    const int* band_first;    // [num_mel]
    const int* band_len;      // [num_mel]
    const float* band_w;      // [num_mel][n_fft / 2] (row b: band_len[b] weights)
    const float* cmvn_mean;   // [num_mel] or null
    const float* cmvn_istd;   // [num_mel] or null
    int B, max_samples, Tmax, frame_len, frame_shift, n_fft, log2_fft, num_mel;
    float preemph, pad_value;
    int remove_dc, use_power, use_log;
};

__global__ __launch_bounds__(256) void fbank_kernel(FbankParams p) {
    __shared__ float re[4][FB_MAX_FFT], im[4][FB_MAX_FFT];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int t = blockIdx.x * 4 + wave;
    const int ns = p.num_samples[b];
    const int T = ns < p.frame_len ? 0 : 1 + (ns - p.frame_len) / p.frame_shift;  // snip_edges = true
    const bool have = t < T;  // (frames past the utterance are written as padding; the barriers below stay uniform)
    float* re_w = re[wave];
    float* im_w = im[wave];
    const float* x = p.wave + (long long)b * p.max_samples + (long long)t * p.frame_shift;
    // ---- load, DC offset
    float v[FB_MAX_FFT / 64];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < FB_MAX_FFT / 64; ++i) {
        const int n = lane + 64 * i;
        v[i] = (have && n < p.frame_len) ? x[n] : 0.f;
        s += v[i];
    }
    if (p.remove_dc) {
        const float mean = wave_sum(s) / (float)p.frame_len;
#pragma unroll
        for (int i = 0; i < FB_MAX_FFT / 64; ++i) v[i] -= mean;
    }
    // neighbours for the pre-emphasis come through LDS (natural order), the windowed frame goes back bit-reversed
#pragma unroll
    for (int i = 0; i < FB_MAX_FFT / 64; ++i) re_w[lane + 64 * i] = v[i];
    __syncthreads();
    float wv[FB_MAX_FFT / 64];
#pragma unroll
    for (int i = 0; i < FB_MAX_FFT / 64; ++i) {
        const int n = lane + 64 * i;
        float y = 0.f;
        if (n < p.frame_len) {
            const float prev = re_w[n > 0 ? n - 1 : 0];
            y = (v[i] - p.preemph * prev) * p.window[n];
        }
        wv[i] = y;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < FB_MAX_FFT / 64; ++i) {
        const int n = lane + 64 * i;
        if (n < p.n_fft) {
            const int r = (int)(__brev((unsigned)n) >> (32 - p.log2_fft));
            re_w[r] = wv[i];
            im_w[r] = 0.f;
        }
    }
    __syncthreads();
    // ---- radix-2 decimation-in-time FFT
    for (int st = 0; st < p.log2_fft; ++st) {
        const int hlf = 1 << st, tw_step = p.n_fft >> (st + 1);
        for (int k = lane; k < p.n_fft / 2; k += 64) {
            const int pos = k & (hlf - 1);
            const int i0 = ((k >> st) << (st + 1)) + pos, i1 = i0 + hlf;
            const float2 w = p.twiddle[pos * tw_step];
            const float ar = re_w[i0], ai = im_w[i0], br = re_w[i1], bi = im_w[i1];
            const float tr = br * w.x - bi * w.y, ti = br * w.y + bi * w.x;
            re_w[i0] = ar + tr;
            im_w[i0] = ai + ti;
            re_w[i1] = ar - tr;
            im_w[i1] = ai - ti;
        }
        __syncthreads();
    }
    // ---- power spectrum of bins 0 .. n_fft/2 - 1 in place (re_w), then one mel bin per lane
    for (int k = lane; k < p.n_fft / 2; k += 64) {
        const float pw = re_w[k] * re_w[k] + im_w[k] * im_w[k];
        im_w[k] = p.use_power ? pw : sqrtf(pw);
    }
    __syncthreads();
    if (t >= p.Tmax) return;
    float* o = p.out + ((long long)b * p.Tmax + t) * p.num_mel;
    for (int mb = lane; mb < p.num_mel; mb += 64) {
        float val = p.pad_value;
        if (have) {
            const int first = p.band_first[mb], len = p.band_len[mb];
            const float* w = p.band_w + (long long)mb * (p.n_fft / 2);
            float e = 0.f;
            for (int i = 0; i < len; ++i) e = fmaf(w[i], im_w[first + i], e);
            if (p.use_log) e = logf(fmaxf(e, 1.1920929e-07f));
            if (p.cmvn_mean) e = (e - p.cmvn_mean[mb]) * p.cmvn_istd[mb];
            val = e;
        }
        o[mb] = val;
    }
}

// ---- host: tables (window, twiddles, mel bands) built in double, kept on the device per option set --------------
namespace {
struct FbankTables {
    FbankOpts o;
    int n_fft = 0, log2_fft = 0, frame_len = 0, frame_shift = 0, device = -1;
    float *window = nullptr, *band_w = nullptr;
    float2* twiddle = nullptr;
    int *band_first = nullptr, *band_len = nullptr;
};
FbankTables g_fb;

bool same_opts(const FbankOpts& a, const FbankOpts& b) {
    return a.sample_rate == b.sample_rate && a.frame_length_ms == b.frame_length_ms && a.frame_shift_ms == b.frame_shift_ms &&
           a.preemph == b.preemph && a.low_freq == b.low_freq && a.high_freq == b.high_freq && a.num_mel == b.num_mel &&
           a.window_type == b.window_type;
}
double mel_of(double f) { return 1127.0 * std::log(1.0 + f / 700.0); }
}  // namespace

static int fbank_tables(const FbankOpts& o, FbankTables** out) {
    int dev = 0;
    CN_HIP_CHECK(hipGetDevice(&dev));
    if (g_fb.window && g_fb.device == dev && same_opts(g_fb.o, o)) {
        *out = &g_fb;
        return 0;
    }
    FbankTables t;
    t.o = o;
    t.device = dev;
    t.frame_len = (int)(o.sample_rate * 0.001 * o.frame_length_ms);
    t.frame_shift = (int)(o.sample_rate * 0.001 * o.frame_shift_ms);
    t.n_fft = 1;
    while (t.n_fft < t.frame_len) t.n_fft *= 2, ++t.log2_fft;
    if (t.frame_len < 2 || t.frame_shift < 1 || t.n_fft > FB_MAX_FFT || o.num_mel < 1 || o.num_mel > 256) {
        cn_set_error("fbank: frame length must fit a 512-point FFT, 1 <= num_mel <= 256");
        return -1;
    }
    const int nb = t.n_fft / 2;
    std::vector<float> win(t.frame_len), bw((size_t)o.num_mel * nb, 0.f);
    std::vector<float2> tw(nb);
    std::vector<int> first(o.num_mel, 0), len(o.num_mel, 0);
    const double a = 2.0 * M_PI / (t.frame_len - 1);
    for (int i = 0; i < t.frame_len; ++i) {
        double w = 1.0;
        if (o.window_type == 0) w = 0.54 - 0.46 * std::cos(a * i);                       // hamming
        else if (o.window_type == 1) w = std::pow(0.5 - 0.5 * std::cos(a * i), 0.85);     // povey
        else if (o.window_type == 2) w = 0.5 - 0.5 * std::cos(a * i);                     // hanning
        win[i] = (float)w;
    }
    for (int k = 0; k < nb; ++k) {
        const double ang = -2.0 * M_PI * k / t.n_fft;
        tw[k] = make_float2((float)std::cos(ang), (float)std::sin(ang));
    }
    const double nyq = 0.5 * o.sample_rate, hi = o.high_freq <= 0.0 ? o.high_freq + nyq : o.high_freq;
    const double mlo = mel_of(o.low_freq), mhi = mel_of(hi), delta = (mhi - mlo) / (o.num_mel + 1);
    const double binw = o.sample_rate / t.n_fft;
    for (int b = 0; b < o.num_mel; ++b) {
        const double left = mlo + b * delta, center = left + delta, right = center + delta;
        int f = -1, n = 0;
        for (int i = 0; i < nb; ++i) {
            const double mel = mel_of(binw * i);
            if (mel > left && mel < right) {
                if (f < 0) f = i;
                bw[(size_t)b * nb + n++] = (float)(mel <= center ? (mel - left) / (center - left) : (right - mel) / (right - center));
            }
        }
        first[b] = f < 0 ? 0 : f;
        len[b] = n;
    }
    if (g_fb.window) {
        (void)hipFree(g_fb.window);
        (void)hipFree(g_fb.band_w);
        (void)hipFree(g_fb.twiddle);
        (void)hipFree(g_fb.band_first);
        (void)hipFree(g_fb.band_len);
    }
    CN_HIP_CHECK(hipMalloc((void**)&t.window, win.size() * 4));
    CN_HIP_CHECK(hipMalloc((void**)&t.band_w, bw.size() * 4));
    CN_HIP_CHECK(hipMalloc((void**)&t.twiddle, tw.size() * 8));
    CN_HIP_CHECK(hipMalloc((void**)&t.band_first, first.size() * 4));
    CN_HIP_CHECK(hipMalloc((void**)&t.band_len, len.size() * 4));
    CN_HIP_CHECK(hipMemcpy(t.window, win.data(), win.size() * 4, hipMemcpyHostToDevice));
    CN_HIP_CHECK(hipMemcpy(t.band_w, bw.data(), bw.size() * 4, hipMemcpyHostToDevice));
    CN_HIP_CHECK(hipMemcpy(t.twiddle, tw.data(), tw.size() * 8, hipMemcpyHostToDevice));
    CN_HIP_CHECK(hipMemcpy(t.band_first, first.data(), first.size() * 4, hipMemcpyHostToDevice));
    CN_HIP_CHECK(hipMemcpy(t.band_len, len.data(), len.size() * 4, hipMemcpyHostToDevice));
    g_fb = t;
    *out = &g_fb;
    return 0;
}

int fbank_num_frames(const FbankOpts& o, int num_samples) {
    const int fl = (int)(o.sample_rate * 0.001 * o.frame_length_ms), fs = (int)(o.sample_rate * 0.001 * o.frame_shift_ms);
    return num_samples < fl || fs < 1 ? 0 : 1 + (num_samples - fl) / fs;
}

int launch_fbank(const FbankOpts& o, const float* wave, const int* num_samples, int B, int max_samples, const float* cmvn_mean,
                 const float* cmvn_istd, float* out, int Tmax, float pad_value, hipStream_t s) {
    if (B <= 0 || Tmax <= 0) return 0;
    FbankTables* t = nullptr;
    CN_TRY(fbank_tables(o, &t));
    FbankParams p;
    p.wave = wave;
    p.num_samples = num_samples;
    p.out = out;
    p.window = t->window;
    p.twiddle = t->twiddle;
    p.band_first = t->band_first;
    p.band_len = t->band_len;
    p.band_w = t->band_w;
    p.cmvn_mean = cmvn_mean;
    p.cmvn_istd = cmvn_mean ? cmvn_istd : nullptr;
    p.B = B;
    p.max_samples = max_samples;
    p.Tmax = Tmax;
    p.frame_len = t->frame_len;
    p.frame_shift = t->frame_shift;
    p.n_fft = t->n_fft;
    p.log2_fft = t->log2_fft;
    p.num_mel = o.num_mel;
    p.preemph = o.preemph;
    p.pad_value = pad_value;
    p.remove_dc = o.remove_dc;
    p.use_power = o.use_power;
    p.use_log = o.use_log;
    hipLaunchKernelGGL(fbank_kernel, dim3(cn_ceil_div(Tmax, 4), B), dim3(256), 0, s, p);
    CN_HIP_CHECK(hipGetLastError());
    return 0;
}
