// F1 + F2: log-mel spectrogram frontend.
// torchaudio MelSpectrogram(power=2, center=True, pad_mode="reflect") -> AmplitudeToDB("power")
// as called by models/audio_encoder.py:113-124,183-184 (Cnn8Rnn) and :29-37,68-69 (CrnnEncoder).
//
// One wave (64 lanes) per STFT frame, four frames per workgroup.  A frame of n_fft real samples
// is gathered straight from the waveform (reflect padding resolved per sample, coalesced reads),
// windowed, packed into n_fft/2 complex points, transformed by a radix-2 Stockham FFT in LDS
// (ping-pong buffers, twiddles in an LDS table), unpacked to the n_fft/2+1 one-sided bins, squared,
// projected on the mel filterbank (each lane owns one mel bin and walks only its non-zero band)
// and written as dB, time-major (B, F, n_mels).  HBM traffic = waveform read once (hop/n_fft
// re-reads hit L2) + 256 B per frame written: the kernel is bound by LDS/VALU, not by HBM.
#include "tag_common.h"

namespace {

template <int NFFT>
__global__ __launch_bounds__(256) void logmel_kernel(const float* __restrict__ wave, int B, int S, int F,
                                                     int win_length, int hop,
                                                     const float* __restrict__ window,
                                                     const float* __restrict__ fb, int n_mels,
                                                     float* __restrict__ out_db,
                                                     float* __restrict__ power_out) {
    constexpr int NC = NFFT / 2;      // complex points
    constexpr int LOGNC = (NFFT == 1024) ? 9 : 10;
    __shared__ float2 tw[NC];         // tw[k] = exp(-2 pi i k / NFFT), k < NFFT/2
    __shared__ float2 buf[4][2][NC];  // per wave ping-pong
    const int lane = threadIdx.x & 63;
    const int wid = threadIdx.x >> 6;

    for (int k = threadIdx.x; k < NC; k += 256) {
        float s, c;
        sincospif(-2.0f * (float)k / (float)NFFT, &s, &c);
        tw[k] = make_float2(c, s);
    }
    // non-zero band of this lane's mel filter (data-driven: scan the column once per workgroup)
    int lo = NC + 1, hi = -1;
    if (lane < n_mels) {
        for (int k = 0; k <= NC; ++k) {
            if (fb[(size_t)k * n_mels + lane] != 0.0f) {
                if (lo > k) lo = k;
                hi = k;
            }
        }
    }
    __syncthreads();

    const long total = (long)B * F;
    const long stride = (long)gridDim.x * 4;
    const long iters = (total + stride - 1) / stride;
    const int left = (NFFT - win_length) / 2;
    float2* z0 = buf[wid][0];
    float2* z1 = buf[wid][1];

    for (long it = 0; it < iters; ++it) {
        const long frame = (long)blockIdx.x * 4 + wid + it * stride;
        const bool valid = frame < total;
        const int b = valid ? (int)(frame / F) : 0;
        const int t = valid ? (int)(frame % F) : 0;
        const float* x = wave + (size_t)b * S;
        const int start = t * hop - NFFT / 2;
        // gather + window + pack two real samples per complex point
#pragma unroll
        for (int j = 0; j < NC / 64; ++j) {
            const int m = lane + 64 * j;
            float v[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int n = 2 * m + e;
                int s = start + n;
                if (s < 0) s = -s;
                if (s >= S) s = 2 * (S - 1) - s;
                const int wi = n - left;
                const float w = (wi >= 0 && wi < win_length) ? window[wi] : 0.0f;
                v[e] = valid ? x[s] * w : 0.0f;
            }
            z0[m] = make_float2(v[0], v[1]);
        }
        __syncthreads();
        // radix-2 Stockham autosort FFT, LOGNC passes
        float2* src = z0;
        float2* dst = z1;
#pragma unroll
        for (int pass = 0; pass < LOGNC; ++pass) {
            const int p = 1 << pass;
#pragma unroll
            for (int j = 0; j < NC / 128; ++j) {
                const int i = lane + 64 * j;          // butterfly index < NC/2
                const int k = i & (p - 1);
                const int o = ((i - k) << 1) + k;
                const float2 w = tw[k * (NC / p)];     // exp(-i pi k / p) = exp(-2 pi i k (NC/p) / NFFT)
                const float2 u0 = src[i];
                const float2 u1r = src[i + NC / 2];
                const float2 u1 = make_float2(u1r.x * w.x - u1r.y * w.y, u1r.x * w.y + u1r.y * w.x);
                dst[o] = make_float2(u0.x + u1.x, u0.y + u1.y);
                dst[o + p] = make_float2(u0.x - u1.x, u0.y - u1.y);
            }
            __syncthreads();
            float2* tmp = src;
            src = dst;
            dst = tmp;
        }
        // src holds Z[0..NC); unpack to the one-sided spectrum and take the power.
        // X[k] = E[k] + W^k O[k],  E = (Z[k] + conj Z[NC-k]) / 2,  O = (Z[k] - conj Z[NC-k]) / (2i)
        float* P = reinterpret_cast<float*>(dst);     // NC+1 floats fit in NC float2
        float pw[NC / 64 + 1];
#pragma unroll
        for (int j = 0; j <= NC / 64; ++j) {
            const int k = lane + 64 * j;
            float val = 0.0f;
            if (k <= NC) {
                const float2 a = src[k & (NC - 1)];
                const float2 bq = src[(NC - k) & (NC - 1)];
                const float er = 0.5f * (a.x + bq.x), ei = 0.5f * (a.y - bq.y);
                const float orr = 0.5f * (a.y + bq.y), oi = -0.5f * (a.x - bq.x);
                float2 w = (k == NC) ? make_float2(-1.0f, 0.0f) : tw[k];
                const float xr = er + (w.x * orr - w.y * oi);
                const float xi = ei + (w.x * oi + w.y * orr);
                val = xr * xr + xi * xi;
            }
            pw[j] = val;
        }
        __syncthreads();   // all reads of src/dst done before P overwrites dst
#pragma unroll
        for (int j = 0; j <= NC / 64; ++j) {
            const int k = lane + 64 * j;
            if (k <= NC) P[k] = pw[j];
        }
        __syncthreads();
        if (lane < n_mels) {
            float acc = 0.0f;
            for (int k = lo; k <= hi; ++k) acc = fmaf(P[k], fb[(size_t)k * n_mels + lane], acc);
            if (valid) {
                const size_t o = (size_t)frame * n_mels + lane;
                if (power_out) power_out[o] = acc;
                // clamp(x, 1e-10) then 10 log10: the floor is exactly -100 dB (as on the CPU path)
                out_db[o] = acc <= 1e-10f ? -100.0f : 10.0f * log10f(acc);
            }
        }
        __syncthreads();
    }
}

}  // namespace

extern "C" int tag_logmel_forward(const float* wave, int B, int S, int n_fft, int win_length, int hop,
                                  const float* window, const float* fb, int n_mels, float* out_db,
                                  float* power_out, void* stream) {
    TAG_CHECK_ARG(wave && window && fb && out_db);
    TAG_CHECK_ARG(n_fft == 1024 || n_fft == 2048);
    TAG_CHECK_ARG(win_length > 0 && win_length <= n_fft && hop > 0);
    TAG_CHECK_ARG(n_mels > 0 && n_mels <= 64);
    TAG_CHECK_ARG(B > 0 && S > n_fft / 2);   // reflect padding needs pad < S (torch.stft raises too)
    const int F = S / hop + 1;
    const long total = (long)B * F;
    int grid = (int)((total + 3) / 4);
    if (grid > 2048) grid = 2048;
    if (n_fft == 1024)
        hipLaunchKernelGGL(logmel_kernel<1024>, dim3(grid), dim3(256), 0, as_stream(stream), wave, B, S, F,
                           win_length, hop, window, fb, n_mels, out_db, power_out);
    else
        hipLaunchKernelGGL(logmel_kernel<2048>, dim3(grid), dim3(256), 0, as_stream(stream), wave, B, S, F,
                           win_length, hop, window, fb, n_mels, out_db, power_out);
    TAG_LAUNCH_CHECK();
    return 0;
}
