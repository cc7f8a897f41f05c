// Windowed stereo STFT (+ |X| + PHAT coherence) and inverse STFT + overlap-add on gfx950.
// Reference: gccNMF/librosaSTFT.py:126-181 (stft), :241-286 (istft); gccNMF/runGCCNMF.py:40,44;
// gccNMF/gccNMFFunctions.py:61-67, :153-163.
//
// These stages are <0.1 % of the path's FLOPs and are HBM/latency-bound, so the design goal is
// "few passes, coalesced": one workgroup transforms TB = 8 consecutive frames in LDS
// (radix-2, all 8 frames advance through each butterfly stage together -> log2(N) barriers per
// 8 frames) and then writes [f][t]-major outputs with the 8 frames innermost, i.e. 32/64-byte
// contiguous segments instead of a scattered column per frame.
//
// Two real signals ride in one complex transform:
//   forward : z = w*(xL + j xR);  F_L[k] = (Z[k]+conj(Z[N-k]))/2,  F_R[k] = (Z[k]-conj(Z[N-k]))/(2j)
//   inverse : Z = F_a + j F_b (both Hermitian-extended)  ->  ifft(Z) = y_a + j y_b
#include "fft_core.h"

// grid = batch * ceil(T / TB); dynamic LDS = (TB*(N+ZPAD) + N/2) float2.
// PCM16 = true: x points at interleaved int16 frames [n][2] (a wav file's data chunk); the int16 -> float32 / 32768
// conversion of wavfile.pcm2float (gccNMF/wavfile.py:57-89) and the de-interleave ride on the load (4 bytes per lane, coalesced).
template <bool PCM16>
__global__ __launch_bounds__(FFT_NT) void stft_stereo_kernel(const float* __restrict__ x, long x_stride, int n_samples, int N,
                                                             int logN, int hop, int T, const float* __restrict__ window,
                                                             const float2* __restrict__ twiddle, float2* __restrict__ X,
                                                             float* __restrict__ V, float* __restrict__ CC, int F, int Fp,
                                                             int Np, int Tp) {
    extern __shared__ __attribute__((aligned(16))) float2 fft_smem[];
    const int zstride = N + FFT_ZPAD;
    float2* z = fft_smem;
    float2* tw = fft_smem + FFT_TB * zstride;
    const int groups = (T + FFT_TB - 1) / FFT_TB;
    const int b = blockIdx.x / groups, t0 = (blockIdx.x - b * groups) * FFT_TB;
    const float* xl = x + b * x_stride;
    const float* xr = xl + n_samples;
    const short2* pcm = (const short2*)x + b * x_stride;      // PCM16: x_stride counts stereo frames

    for (int i = threadIdx.x; i < N / 2; i += FFT_NT) tw[i] = twiddle[i];
    for (int idx = threadIdx.x; idx < FFT_TB * N; idx += FFT_NT) {
        const int tb = idx / N, n = idx - tb * N;
        const int t = t0 + tb;
        float2 v = make_float2(0.f, 0.f);
        if (t < T) {
            const float w = window[n];
            const long s = (long)t * hop + n;
            if (PCM16) {
                const short2 q = pcm[s];
                v = make_float2(w * ((float)q.x / 32768.f), w * ((float)q.y / 32768.f));
            } else {
                v = make_float2(w * xl[s], w * xr[s]);
            }
        }
        z[tb * zstride + bitrev(n, logN)] = v;
    }
    __syncthreads();
    fft_stages<false>(z, tw, N, logN, zstride);

    const long plane = (long)Fp * Tp;
    float2* Xb = X + (long)b * 2 * plane;
    for (int idx = threadIdx.x; idx < F * FFT_TB; idx += FFT_NT) {
        const int f = idx / FFT_TB, tb = idx - f * FFT_TB;
        const int t = t0 + tb;
        if (t >= T) continue;
        const float2 zk = z[tb * zstride + f];
        const float2 zn = z[tb * zstride + ((N - f) & (N - 1))];
        // reference stores the CONJUGATE of the FFT (librosaSTFT.py:176-179)
        const float2 XL = make_float2(0.5f * (zk.x + zn.x), -0.5f * (zk.y - zn.y));
        const float2 XR = make_float2(0.5f * (zk.y + zn.y), 0.5f * (zk.x - zn.x));
        Xb[(long)f * Tp + t] = XL;
        Xb[plane + (long)f * Tp + t] = XR;
        const float aL = hypotf(XL.x, XL.y), aR = hypotf(XR.x, XR.y);
        if (V) {
            float* Vb = V + (long)b * Fp * Np + (long)f * Np;
            Vb[t] = aL;
            Vb[T + t] = aR;
        }
        if (CC) {
            // X0 * conj(X1) / |X0| / |X1|  (runGCCNMF.py:44)
            float re = XL.x * XR.x + XL.y * XR.y, im = XL.y * XR.x - XL.x * XR.y;
            if (aL > 0.f && aR > 0.f) {
                re = re / aL / aR;
                im = im / aL / aR;
            } else {
                re = 0.f;   // a bin that is exactly 0 in f32 carries no phase: it contributes nothing (NaN policy)
                im = 0.f;
            }
            float* Cb = CC + (long)b * 2 * plane + (long)f * Tp + t;
            Cb[0] = re;
            Cb[plane] = im;
        }
    }
}

// grid = batch * (nsig/2) * ceil(T / TB).  Writes windowed time frames [batch][nsig][T][N].
__global__ __launch_bounds__(FFT_NT) void istft_frames_kernel(const float2* __restrict__ spec, int nsig, int N, int logN, int T,
                                                              const float* __restrict__ window,
                                                              const float2* __restrict__ twiddle, float* __restrict__ frames,
                                                              int F, int Fp, int Tp) {
    extern __shared__ __attribute__((aligned(16))) float2 fft_smem[];
    const int zstride = N + FFT_ZPAD;
    float2* z = fft_smem;
    float2* tw = fft_smem + FFT_TB * zstride;
    const int groups = (T + FFT_TB - 1) / FFT_TB;
    const int npairs = nsig / 2;
    int id = blockIdx.x;
    const int g = id % groups;
    id /= groups;
    const int pr = id % npairs, b = id / npairs;
    const int t0 = g * FFT_TB;
    const long plane = (long)Fp * Tp;
    const float2* Sa = spec + ((long)b * nsig + 2 * pr) * plane;
    const float2* Sb = Sa + plane;

    for (int i = threadIdx.x; i < N / 2; i += FFT_NT) tw[i] = twiddle[i];
    for (int idx = threadIdx.x; idx < F * FFT_TB; idx += FFT_NT) {
        const int f = idx / FFT_TB, tb = idx - f * FFT_TB;
        const int t = t0 + tb;
        float2 fa = make_float2(0.f, 0.f), fb = fa;
        if (t < T) {
            const float2 sa = Sa[(long)f * Tp + t], sb = Sb[(long)f * Tp + t];
            fa = make_float2(sa.x, -sa.y);   // istft undoes the stored conjugate (librosaSTFT.py:278)
            fb = make_float2(sb.x, -sb.y);
        }
        if (f == 0 || f == N / 2) {          // ifft(...).real keeps only the real part of these two bins
            fa.y = 0.f;
            fb.y = 0.f;
        }
        float2* zz = z + tb * zstride;
        zz[bitrev(f, logN)] = make_float2(fa.x - fb.y, fa.y + fb.x);
        if (f != 0 && f != N / 2) zz[bitrev(N - f, logN)] = make_float2(fa.x + fb.y, fb.x - fa.y);
    }
    __syncthreads();
    fft_stages<true>(z, tw, N, logN, zstride);

    const float invN = 1.0f / (float)N;
    float* Fa = frames + (((long)b * nsig + 2 * pr) * T) * N;
    float* Fb = Fa + (long)T * N;
    for (int idx = threadIdx.x; idx < FFT_TB * N; idx += FFT_NT) {
        const int tb = idx / N, n = idx - tb * N;
        const int t = t0 + tb;
        if (t >= T) continue;
        const float2 v = z[tb * zstride + n];
        const float w = window[n];
        Fa[(long)t * N + n] = w * (v.x * invN);
        Fb[(long)t * N + n] = w * (v.y * invN);
    }
}

// Overlap-add in ASCENDING frame order (the order librosaSTFT.py:275-281 accumulates in), centre trim
// of `trim` samples at both ends (:283-284), times the gain (gccNMFFunctions.py:155,163).
// grid = (ceil(L/256), nsig, batch).
__global__ __launch_bounds__(256) void istft_ola_kernel(const float* __restrict__ frames, int N, int hop, int T, int L, int trim,
                                                        float gain, float* __restrict__ y) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= L) return;
    const long sig = (long)blockIdx.z * gridDim.y + blockIdx.y;
    const float* fr = frames + sig * (long)T * N;
    const int s = m + trim;
    int t_lo = (s - N + hop) / hop;       // ceil((s - N + 1) / hop) for s - N + 1 > 0
    if (s - N + 1 <= 0) t_lo = 0;
    int t_hi = s / hop;
    if (t_hi > T - 1) t_hi = T - 1;
    float acc = 0.f;
    for (int t = t_lo; t <= t_hi; ++t) acc = acc + fr[(long)t * N + (s - t * hop)];
    y[sig * L + m] = acc * gain;
}

// ---- int16 egress (gccNMF/wavfile.py:39-48, :92-131) ---------------------------------------------------------
// peak[g] = max |y| over the 2*L samples of group g (= one target of one file: what one wavwrite call sees).
// Non-negative floats order like their bit patterns, so atomicMax on the uint image is exact and order independent -- and
// +Inf / NaN images (>= 0x7F800000) rank above every finite value, so a non-finite sample always surfaces in peak[g]: the host
// checks for it (the reference's wavwrite would write platform-defined garbage for NaN; here NaN -> 0, +-Inf clip, no rescale).
__global__ __launch_bounds__(256) void pcm_peak_kernel(const float* __restrict__ y, long group_len, unsigned int* __restrict__ peak) {
    const long g = blockIdx.y;
    const float* yg = y + g * group_len;
    unsigned int m = 0u;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < group_len; i += (long)gridDim.x * 256) m = max(m, __float_as_uint(fabsf(yg[i])));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned int)__shfl_xor((int)m, o));
    if ((threadIdx.x & 63) == 0) atomicMax(peak + g, m);
}

// y [g][2][L] float -> pcm [g][L][2] int16: clip protection (peak >= 1 -> x / peak * 0.99), x * 32768, clip, truncate.
__global__ __launch_bounds__(256) void pcm_pack_kernel(const float* __restrict__ y, int L, const unsigned int* __restrict__ peak,
                                                       short2* __restrict__ pcm) {
    const long g = blockIdx.y;
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= L) return;
    const float pk = __uint_as_float(peak[g]);
    float a = y[(g * 2) * L + m], b = y[(g * 2 + 1) * L + m];
    if (pk >= 1.f && peak[g] < 0x7F800000u) {
        a = a / pk * 0.99f;
        b = b / pk * 0.99f;
    }
    a = (a == a) ? a : 0.f;
    b = (b == b) ? b : 0.f;
    a = fminf(fmaxf(a * 32768.f, -32768.f), 32767.f);
    b = fminf(fmaxf(b * 32768.f, -32768.f), 32767.f);
    pcm[g * L + m] = make_short2((short)(int)a, (short)(int)b);
}

extern "C" {

static int launch_stft(const void* x, long x_stride, int n_samples, int n_fft, int hop, int T, int batch, const float* window,
                       const float* twiddle, float* X, float* V, float* CC, bool pcm16, void* stream) {
    const int logN = ilog2_exact(n_fft);
    if (!x || !window || !twiddle || !X || logN < 6 || logN > 12 || hop < 1 || T < 1 || batch < 1) return GCCNMF_ERR_ARG;
    if ((long)(T - 1) * hop + n_fft > n_samples) return GCCNMF_ERR_ARG;
    const int F = n_fft / 2 + 1;
    GccNmfPitches p = gccnmf_make_pitches(F, T, 1);
    const size_t lds = sizeof(float2) * ((size_t)FFT_TB * (n_fft + FFT_ZPAD) + n_fft / 2);
    if (lds > 160 * 1024) return GCCNMF_ERR_UNSUPPORTED;
    const void* fn = pcm16 ? (const void*)stft_stereo_kernel<true> : (const void*)stft_stereo_kernel<false>;
    if (lds > 64 * 1024) {
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return GCCNMF_ERR_LAUNCH;
    }
    const int groups = gccnmf_ceil_div(T, FFT_TB);
    if (pcm16)
        hipLaunchKernelGGL(stft_stereo_kernel<true>, dim3(batch * groups), dim3(FFT_NT), lds, (hipStream_t)stream, (const float*)x,
                           x_stride, n_samples, n_fft, logN, hop, T, window, (const float2*)twiddle, (float2*)X, V, CC, F, p.Fp, p.Np, p.Tp);
    else
        hipLaunchKernelGGL(stft_stereo_kernel<false>, dim3(batch * groups), dim3(FFT_NT), lds, (hipStream_t)stream, (const float*)x,
                           x_stride, n_samples, n_fft, logN, hop, T, window, (const float2*)twiddle, (float2*)X, V, CC, F, p.Fp, p.Np, p.Tp);
    GCCNMF_CHECK_LAUNCH();
    return GCCNMF_OK;
}

int gccnmf_stft_stereo(const float* x, long x_stride, int n_samples, int n_fft, int hop, int T, int batch, const float* window,
                       const float* twiddle, float* X, float* V, float* CC, void* stream) {
    return launch_stft(x, x_stride, n_samples, n_fft, hop, T, batch, window, twiddle, X, V, CC, false, stream);
}

int gccnmf_stft_stereo_pcm16(const short* pcm, long frame_stride, int n_samples, int n_fft, int hop, int T, int batch,
                             const float* window, const float* twiddle, float* X, float* V, float* CC, void* stream) {
    return launch_stft(pcm, frame_stride, n_samples, n_fft, hop, T, batch, window, twiddle, X, V, CC, true, stream);
}

int gccnmf_istft_ola(const float* spec, int nsig, int n_fft, int hop, int T, int batch, const float* window,
                     const float* twiddle, float gain, int center, float* frames, float* y, void* stream) {
    const int logN = ilog2_exact(n_fft);
    if (!spec || !window || !twiddle || !frames || !y || logN < 6 || logN > 12 || hop < 1 || T < 1 || batch < 1 || nsig < 2 ||
        (nsig & 1))
        return GCCNMF_ERR_ARG;
    const int F = n_fft / 2 + 1;
    GccNmfPitches p = gccnmf_make_pitches(F, T, 1);
    const size_t lds = sizeof(float2) * ((size_t)FFT_TB * (n_fft + FFT_ZPAD) + n_fft / 2);
    if (lds > 160 * 1024) return GCCNMF_ERR_UNSUPPORTED;
    if (lds > 64 * 1024) {
        if (hipFuncSetAttribute((const void*)istft_frames_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return GCCNMF_ERR_LAUNCH;
    }
    const int groups = gccnmf_ceil_div(T, FFT_TB);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(istft_frames_kernel, dim3(batch * (nsig / 2) * groups), dim3(FFT_NT), lds, s, (const float2*)spec, nsig,
                       n_fft, logN, T, window, (const float2*)twiddle, frames, F, p.Fp, p.Tp);
    GCCNMF_CHECK_LAUNCH();
    const int trim = center ? n_fft / 2 : 0;
    const int L = n_fft + hop * (T - 1) - 2 * trim;
    if (L < 1) return GCCNMF_ERR_ARG;
    hipLaunchKernelGGL(istft_ola_kernel, dim3(gccnmf_ceil_div(L, 256), nsig, batch), dim3(256), 0, s, frames, n_fft, hop, T, L,
                       trim, gain, y);
    GCCNMF_CHECK_LAUNCH();
    return GCCNMF_OK;
}

int gccnmf_ola_frames(const float* frames, int nsig, int n_fft, int hop, int T, int batch, int first_sample, int L, float gain,
                      float* y, void* stream) {
    if (!frames || !y || nsig < 1 || n_fft < 2 || hop < 1 || T < 1 || batch < 1 || first_sample < 0 || L < 1 ||
        (long)first_sample + L > (long)n_fft + (long)hop * (T - 1))
        return GCCNMF_ERR_ARG;
    hipLaunchKernelGGL(istft_ola_kernel, dim3(gccnmf_ceil_div(L, 256), nsig, batch), dim3(256), 0, (hipStream_t)stream, frames, n_fft, hop, T,
                       L, first_sample, gain, y);
    GCCNMF_CHECK_LAUNCH();
    return GCCNMF_OK;
}

int gccnmf_pack_pcm16(const float* y, int groups, int L, unsigned int* peak_scratch, short* pcm, void* stream) {
    if (!y || !peak_scratch || !pcm || groups < 1 || L < 1) return GCCNMF_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(peak_scratch, 0, sizeof(unsigned int) * groups, s) != hipSuccess) return GCCNMF_ERR_LAUNCH;
    hipLaunchKernelGGL(pcm_peak_kernel, dim3(64, groups), dim3(256), 0, s, y, 2L * L, peak_scratch);
    GCCNMF_CHECK_LAUNCH();
    hipLaunchKernelGGL(pcm_pack_kernel, dim3(gccnmf_ceil_div(L, 256), groups), dim3(256), 0, s, y, L, peak_scratch, (short2*)pcm);
    GCCNMF_CHECK_LAUNCH();
    return GCCNMF_OK;
}

}  // extern "C"
