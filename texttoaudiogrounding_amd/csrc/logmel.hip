// F1 + F2: log-mel spectrogram frontend.
// torchaudio MelSpectrogram(power=2, center=True, pad_mode="reflect") -> AmplitudeToDB("power")
// as called by models/audio_encoder.py:113-124,183-184 (Cnn8Rnn) and :29-37,68-69 (CrnnEncoder).
//
// One wave (64 lanes) per STFT frame, four independent frames per workgroup, NO workgroup barrier inside the frame loop:
// every wave owns its LDS buffer and DS operations of a wave execute in order.
//   * a frame of n_fft real samples is read straight from the waveform as coalesced 8-byte loads (two consecutive samples
//     = one complex point; reflect padding only on the frames that touch a clip edge), multiplied by the window values the
//     lane keeps in registers (the lane -> sample map is the same for every frame);
//   * NC = n_fft/2 complex points go through a Stockham autosort FFT of radix-8 / radix-4 stages (512 = 8.8.8,
//     1024 = 8.8.4.4): a lane holds its 8 / 16 points in registers, the first stage needs no LDS read, the others exchange
//     through the wave's buffer (index padded by i/8 so that the strided scatter of the early stages is conflict-free);
//   * real-FFT unpack -> |X|^2 -> mel projection: lane = mel bin, the non-zero band of every filter is compacted once per
//     workgroup into LDS as fbc[i][mel] (zero padded to the widest band), so the projection is a fixed-trip loop of two
//     conflict-free LDS reads + one FMA (terms in ascending-frequency order, like the dense sum) -> dB, time-major.
// HBM traffic = waveform read once (hop/n_fft re-reads hit L2) + 256 B per frame written: the kernel is LDS/VALU-bound.
#include "tag_common.h"

namespace {

struct c32 {
    float x, y;
};
__device__ __forceinline__ c32 cadd(c32 a, c32 b) { return {a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ c32 csub(c32 a, c32 b) { return {a.x - b.x, a.y - b.y}; }
__device__ __forceinline__ c32 cmul(c32 a, c32 w) { return {fmaf(a.x, w.x, -(a.y * w.y)), fmaf(a.x, w.y, a.y * w.x)}; }
__device__ __forceinline__ c32 mul_mi(c32 a) { return {a.y, -a.x}; }            // a * (-i)

template <int R>
__device__ __forceinline__ void dft(c32* v);
template <>
__device__ __forceinline__ void dft<2>(c32* v) {
    const c32 a = v[0], b = v[1];
    v[0] = cadd(a, b);
    v[1] = csub(a, b);
}
template <>
__device__ __forceinline__ void dft<4>(c32* v) {
    const c32 s0 = cadd(v[0], v[2]), d0 = csub(v[0], v[2]), s1 = cadd(v[1], v[3]), d1 = mul_mi(csub(v[1], v[3]));
    v[0] = cadd(s0, s1);
    v[1] = cadd(d0, d1);
    v[2] = csub(s0, s1);
    v[3] = csub(d0, d1);
}
template <>
__device__ __forceinline__ void dft<8>(c32* v) {
    c32 e[4] = {v[0], v[2], v[4], v[6]}, o[4] = {v[1], v[3], v[5], v[7]};
    dft<4>(e);
    dft<4>(o);
    const float h = 0.70710678118654752440f;
    o[1] = {h * (o[1].x + o[1].y), h * (o[1].y - o[1].x)};                       // * (1 - i)/sqrt2
    o[2] = mul_mi(o[2]);
    o[3] = {h * (o[3].y - o[3].x), -h * (o[3].x + o[3].y)};                      // * (-1 - i)/sqrt2
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        v[m] = cadd(e[m], o[m]);
        v[m + 4] = csub(e[m], o[m]);
    }
}

__device__ __forceinline__ int padi(int i) { return i + (i >> 3); }

// One Stockham stage of radix R after NS = (product of the earlier radices) points are already combined.  v[b][r] holds
// in[j + r NC/R] for the lane's butterflies j = lane + 64 b; the results go to z (padded natural Stockham order).
template <int NC, int R, int NS>
__device__ __forceinline__ void stage_from_regs(c32 (*v)[R], c32* z, const c32* twn, int lane) {
    constexpr int NB = NC / R / 64;
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int j = lane + 64 * b;
        const int k = j & (NS - 1);
        if (NS > 1) {
#pragma unroll
            for (int r = 1; r < R; ++r) v[b][r] = cmul(v[b][r], twn[r * k * (NC / (NS * R))]);
        }
        dft<R>(v[b]);
        const int j0 = (j - k) * R + k;
#pragma unroll
        for (int r = 0; r < R; ++r) z[padi(j0 + r * NS)] = v[b][r];
    }
}

template <int NC, int R, int NS>
__device__ __forceinline__ void stage(c32* z, const c32* twn, int lane) {
    constexpr int NB = NC / R / 64;
    c32 v[NB][R];
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int r = 0; r < R; ++r) v[b][r] = z[padi(lane + 64 * b + r * (NC / R))];
    __builtin_amdgcn_wave_barrier();                    // every read of this stage precedes its writes (one buffer)
    stage_from_regs<NC, R, NS>(v, z, twn, lane);
    __builtin_amdgcn_wave_barrier();
}

template <int NFFT>
struct Plan;
template <>
struct Plan<1024> {
    static constexpr int R0 = 8, WCAP = 96;
    template <int NC>
    static __device__ __forceinline__ void rest(c32* z, const c32* twn, int lane) {
        stage<NC, 8, 8>(z, twn, lane);
        stage<NC, 8, 64>(z, twn, lane);
    }
};
template <>
struct Plan<2048> {
    static constexpr int R0 = 8, WCAP = 192;
    template <int NC>
    static __device__ __forceinline__ void rest(c32* z, const c32* twn, int lane) {
        stage<NC, 8, 8>(z, twn, lane);
        stage<NC, 4, 64>(z, twn, lane);
        stage<NC, 4, 256>(z, twn, lane);
    }
};

// waves per workgroup: 4, or 8 at n_fft 2048 -- that instance needs 246 VGPRs and 103 KB of tables + exchange buffers, i.e. ONE
// workgroup per CU: with four waves every SIMD ran a single wave (0.45 ms for 64 x 10 s CrnnEncoder clips against 0.16 ms for twice as
// many frames of the 1024 instance); eight waves share the same tables at 140 KB and give every SIMD two
template <int NFFT> constexpr int logmel_waves() { return NFFT == 2048 ? 8 : 4; }

template <int NFFT>
__global__ __launch_bounds__(64 * logmel_waves<NFFT>()) void logmel_kernel(const float* __restrict__ wave, int B, int S, int F,
                                                     int win_length, int hop,
                                                     const float* __restrict__ window,
                                                     const float* __restrict__ fb, int n_mels,
                                                     float* __restrict__ out_db,
                                                     float* __restrict__ power_out) {
    constexpr int NC = NFFT / 2;                        // complex points
    constexpr int R0 = Plan<NFFT>::R0, NB0 = NC / R0 / 64, WCAP = Plan<NFFT>::WCAP;
    constexpr int ZP = NC + NC / 8 + 8;                 // padded buffer length
    constexpr int NW = logmel_waves<NFFT>();            // waves per workgroup
    __shared__ c32 twn[NC];                             // exp(-2 pi i m / NC)
    __shared__ c32 twh[NC];                             // exp(-2 pi i k / NFFT) (real-FFT unpack)
    __shared__ c32 zbuf[NW][ZP];
    __shared__ float fbc[WCAP][64];
    __shared__ int band[2][64];
    const int lane = threadIdx.x & 63;
    const int wid = threadIdx.x >> 6;

    for (int k = threadIdx.x; k < NC; k += 64 * NW) {
        float s, c;
        sincospif(-2.0f * (float)k / (float)NC, &s, &c);
        twn[k] = {c, s};
        sincospif(-2.0f * (float)k / (float)NFFT, &s, &c);
        twh[k] = {c, s};
    }
    if (threadIdx.x < 64) {
        band[0][threadIdx.x] = NC + 1;
        band[1][threadIdx.x] = -1;
    }
    __syncthreads();
    // non-zero band of every mel filter (data-driven): wave w scans the rows k = w (mod NW), lanes = mel bins
    {
        int lo = NC + 1, hi = -1;
        if (lane < n_mels) {
            for (int k = wid; k <= NC; k += NW) {
                if (fb[(size_t)k * n_mels + lane] != 0.0f) {
                    lo = min(lo, k);
                    hi = max(hi, k);
                }
            }
            atomicMin(&band[0][lane], lo);
            atomicMax(&band[1][lane], hi);
        }
    }
    __syncthreads();
    const int lo = band[0][lane], hi = band[1][lane];
    for (int i = wid; i < WCAP; i += NW)
        fbc[i][lane] = (lane < n_mels && lo + i <= hi) ? fb[(size_t)(lo + i) * n_mels + lane] : 0.0f;
    // widest band of the workgroup's filterbank (uniform trip count of the projection loop)
    int wmax = (lane < n_mels) ? hi - lo + 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) wmax = max(wmax, __shfl_xor(wmax, o));
    const int wloop = min(wmax, WCAP);
    __syncthreads();

    // the lane's window values and real-FFT unpack twiddle indices are the same for every frame
    const int left = (NFFT - win_length) / 2;
    float wv[NB0][R0][2];
#pragma unroll
    for (int b = 0; b < NB0; ++b)
#pragma unroll
        for (int r = 0; r < R0; ++r)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int wi = 2 * (lane + 64 * b + r * (NC / R0)) + e - left;
                wv[b][r][e] = (wi >= 0 && wi < win_length) ? window[wi] : 0.0f;
            }

    const long total = (long)B * F;
    const long stride = (long)gridDim.x * NW;
    c32* z = zbuf[wid];
    float* P = reinterpret_cast<float*>(z);             // NC + 1 floats, written after the last read of z

    for (long frame = (long)blockIdx.x * NW + wid; frame < total; frame += stride) {
        const int b = (int)(frame / F);
        const int t = (int)(frame % F);
        const float* x = wave + (size_t)b * S;
        const int start = t * hop - NFFT / 2;
        c32 v[NB0][R0];
        if (start >= 0 && start + NFFT <= S && ((((size_t)b * S + start) & 1) == 0)) {
            const float2* x2 = reinterpret_cast<const float2*>(x + start);
#pragma unroll
            for (int bb = 0; bb < NB0; ++bb)
#pragma unroll
                for (int r = 0; r < R0; ++r) {
                    const float2 q = x2[lane + 64 * bb + r * (NC / R0)];
                    v[bb][r] = {q.x * wv[bb][r][0], q.y * wv[bb][r][1]};
                }
        } else {
#pragma unroll
            for (int bb = 0; bb < NB0; ++bb)
#pragma unroll
                for (int r = 0; r < R0; ++r) {
                    float q[2];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        int s = start + 2 * (lane + 64 * bb + r * (NC / R0)) + e;
                        if (s < 0) s = -s;
                        if (s >= S) s = 2 * (S - 1) - s;
                        q[e] = x[s];
                    }
                    v[bb][r] = {q[0] * wv[bb][r][0], q[1] * wv[bb][r][1]};
                }
        }
        stage_from_regs<NC, R0, 1>(v, z, twn, lane);
        __builtin_amdgcn_wave_barrier();
        Plan<NFFT>::template rest<NC>(z, twn, lane);
        // z holds Z[0..NC) in natural order; unpack to the one-sided spectrum and take the power.
        // X[k] = E[k] + W^k O[k],  E = (Z[k] + conj Z[NC-k]) / 2,  O = (Z[k] - conj Z[NC-k]) / (2i)
        float pw[NC / 64 + 1];
#pragma unroll
        for (int j = 0; j <= NC / 64; ++j) {
            const int k = lane + 64 * j;
            float val = 0.0f;
            if (k <= NC) {
                const c32 a = z[padi(k & (NC - 1))];
                const c32 bq = z[padi((NC - k) & (NC - 1))];
                const float er = 0.5f * (a.x + bq.x), ei = 0.5f * (a.y - bq.y);
                const float orr = 0.5f * (a.y + bq.y), oi = -0.5f * (a.x - bq.x);
                const c32 w = (k == NC) ? c32{-1.0f, 0.0f} : twh[k];
                const c32 wo = cmul(c32{orr, oi}, w);
                const float xr = er + wo.x;
                const float xi = ei + wo.y;
                val = fmaf(xr, xr, xi * xi);
            }
            pw[j] = val;
        }
        __builtin_amdgcn_wave_barrier();                // all reads of z done before P overwrites it
#pragma unroll
        for (int j = 0; j <= NC / 64; ++j) {
            const int k = lane + 64 * j;
            if (k <= NC) P[k] = pw[j];
        }
        __builtin_amdgcn_wave_barrier();
        float acc = 0.0f;
#pragma unroll 4
        for (int i = 0; i < wloop; ++i) acc = fmaf(P[min(lo + i, NC)], fbc[i][lane], acc);
        if (lane < n_mels) {
            for (int k = lo + WCAP; k <= hi; ++k) acc = fmaf(P[k], fb[(size_t)k * n_mels + lane], acc);   // bands wider than WCAP
            const size_t o = (size_t)frame * n_mels + lane;
            if (power_out) power_out[o] = acc;
            // clamp(x, 1e-10) then 10 log10: the floor is exactly -100 dB (as on the CPU path)
            out_db[o] = acc <= 1e-10f ? -100.0f : 10.0f * log10f(acc);
        }
        __builtin_amdgcn_wave_barrier();                // P is read before the next frame's stage writes z
    }
}

// Waveform packs store float16 samples (utils/data/pack_waveform.py:46-52); the loaders widen them to float32 and zero-pad the
// batch to its longest clip (datasets/single_phrase_dataset.py:44-45, utils/train_util.py:211-216).  Here the RAGGED float16
// samples are what crosses PCIe (half the bytes, no padding) and this kernel does both steps on the device: out[b][s] =
// float(packed[off[b] + s]) for s < off[b+1] - off[b], else 0.  8 samples (16 B in, 32 B out) per thread; exact (every
// float16 is a float32).
__global__ __launch_bounds__(256) void waveform_f16_unpack_kernel(const _Float16* __restrict__ packed,
                                                                  const long* __restrict__ off, int B, int S,
                                                                  float* __restrict__ out, long* __restrict__ len_out) {
    const int b = blockIdx.y;
    const long o0 = off[b], n = off[b + 1] - o0;
    if (len_out && blockIdx.x == 0 && threadIdx.x == 0) len_out[b] = n < S ? n : S;   // a clip cut to S reports the cut length (the reference's collate never yields len > width)
    const _Float16* src = packed + o0;
    float* dst = out + (size_t)b * S;
    // head: scalar samples until the SOURCE is 16-byte aligned (ragged offsets), then 8-sample vector items
    const int head = (int)(((16 - (reinterpret_cast<uintptr_t>(src) & 15)) & 15) >> 1);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; ; i += (long)gridDim.x * 256) {
        const long s0 = i == 0 ? 0 : head + (i - 1) * 8;
        const int cnt = i == 0 ? head : 8;
        if (s0 >= S) break;
        if (i > 0 && s0 + 8 <= n && s0 + 8 <= S) {
            typedef _Float16 h8 __attribute__((ext_vector_type(8)));
            const h8 v = *reinterpret_cast<const h8*>(src + s0);
#pragma unroll
            for (int j = 0; j < 8; ++j) dst[s0 + j] = (float)v[j];
        } else {
            for (int j = 0; j < cnt && s0 + j < S; ++j) dst[s0 + j] = s0 + j < n ? (float)src[s0 + j] : 0.0f;
        }
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
    const int nw = n_fft == 1024 ? logmel_waves<1024>() : logmel_waves<2048>();
    int grid = (int)((total + nw - 1) / nw);
    int cus = tag_device_cu_count();
    // persistent workgroups: the per-workgroup table set-up is amortised (three resident per CU at n_fft 1024, one at 2048)
    const int cap = (n_fft == 1024 ? 3 : 1) * (cus > 0 ? cus : 256);
    if (grid > cap) grid = cap;
    if (n_fft == 1024)
        hipLaunchKernelGGL(logmel_kernel<1024>, dim3(grid), dim3(64 * logmel_waves<1024>()), 0, as_stream(stream), wave, B, S, F,
                           win_length, hop, window, fb, n_mels, out_db, power_out);
    else
        hipLaunchKernelGGL(logmel_kernel<2048>, dim3(grid), dim3(64 * logmel_waves<2048>()), 0, as_stream(stream), wave, B, S, F,
                           win_length, hop, window, fb, n_mels, out_db, power_out);
    TAG_LAUNCH_CHECK();
    return 0;
}

extern "C" int tag_waveform_f16_to_f32_padded(const void* packed_f16, const long* offsets, int B, int S, float* out,
                                              long* len_out, void* stream) {
    TAG_CHECK_ARG(packed_f16 && offsets && out && B > 0 && S > 0);
    TAG_CHECK_ARG(reinterpret_cast<uintptr_t>(packed_f16) % 2 == 0);
    int gx = cdiv((long)S / 8 + 2, 256);
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(waveform_f16_unpack_kernel, dim3(gx, B), dim3(256), 0, as_stream(stream),
                       static_cast<const _Float16*>(packed_f16), offsets, B, S, out, len_out);
    TAG_LAUNCH_CHECK();
    return 0;
}
