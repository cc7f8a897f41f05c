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
#ifndef FFT_ABL
#define FFT_ABL 0     // build-time timing experiments (results invalid): 1 = no butterfly stages, 2 = no output stage (STFT kernel)
#endif

// grid = batch * ceil(T / TB); dynamic LDS = (TB*(N+ZPAD) + N/2) float2.
// PCM16 = true: x points at interleaved int16 frames [n][2] (a wav file's data chunk); the int16 -> float32 / 32768
// conversion of wavfile.pcm2float (gccNMF/wavfile.py:57-89) and the de-interleave ride on the load (4 bytes per lane, coalesced).
template <bool PCM16, int TB>
__global__ __launch_bounds__(FFT_NT) void stft_stereo_kernel(const float* __restrict__ x, long x_stride, int n_samples, int N,
                                                             int logN, int hop, int T, const float* __restrict__ window,
                                                             const float2* __restrict__ twiddle, float2* __restrict__ X,
                                                             float* __restrict__ V, float* __restrict__ CC, int F, int Fp,
                                                             int Np, int Tp, int ps) {
    extern __shared__ __attribute__((aligned(16))) float2 fft_smem[];
    const int zstride = N + (N >> ps) + FFT_ZPAD;
    float2* z = fft_smem;
    float2* tw = fft_smem + TB * zstride;
    const int groups = (T + TB - 1) / TB;
    const int b = blockIdx.x / groups, t0 = (blockIdx.x - b * groups) * TB;
    const float* xl = x + b * x_stride;
    const float* xr = xl + n_samples;
    const short2* pcm = (const short2*)x + b * x_stride;      // PCM16: x_stride counts stereo frames

    for (int i = threadIdx.x; i < N / 2; i += FFT_NT) tw[i] = twiddle[i];
    // Eight samples per thread at a time, every load before the first use (clamped addresses, masked afterwards): as one plain loop the
    // compiler's code was load - wait - store per trip, TB N / 256 = 32 dependent round trips per workgroup.
    constexpr int UN = 8;
    for (int base = threadIdx.x; base < TB * N; base += FFT_NT * UN) {
        float w[UN], xa[UN], xb[UN];
        short2 q[UN];
#pragma unroll
        for (int j = 0; j < UN; ++j) {
            const int idx = min(base + j * FFT_NT, TB * N - 1);
            const int tb = idx / N, n = idx - tb * N;
            const long s = (long)min(t0 + tb, T - 1) * hop + n;
            w[j] = window[n];
            if (PCM16) {
                q[j] = pcm[s];
            } else {
                xa[j] = xl[s];
                xb[j] = xr[s];
            }
        }
#pragma unroll
        for (int j = 0; j < UN; ++j) {
            const int idx = base + j * FFT_NT;
            if (idx < TB * N) {
                const int tb = idx / N, n = idx - tb * N;
                float2 v = make_float2(0.f, 0.f);
                if (t0 + tb < T) {
                    if (PCM16) v = make_float2(w[j] * ((float)q[j].x / 32768.f), w[j] * ((float)q[j].y / 32768.f));
                    else v = make_float2(w[j] * xa[j], w[j] * xb[j]);
                }
                z[tb * zstride + fft_pad(bitrev(n, logN), ps)] = v;
            }
        }
    }
    __syncthreads();
#if !(FFT_ABL & 1)
    fft_stages_any<false, TB>(z, tw, N, logN, zstride, ps);
#endif
#if FFT_ABL & 2
    if (z[threadIdx.x].x != 123.456f) return;       // timing experiment: no output stage
#endif

    const long plane = (long)Fp * Tp;
    float2* Xb = X + (long)b * 2 * plane;
    for (int idx = threadIdx.x; idx < F * TB; idx += FFT_NT) {
        const int f = idx / TB, tb = idx - f * TB;
        const int t = t0 + tb;
        if (t >= T) continue;
        const float2 zk = z[tb * zstride + fft_pad(f, ps)];
        const float2 zn = z[tb * zstride + fft_pad((N - f) & (N - 1), ps)];
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
template <int TB>
__global__ __launch_bounds__(FFT_NT) void istft_frames_kernel(const float2* __restrict__ spec, int nsig, int N, int logN, int T,
                                                              const float* __restrict__ window,
                                                              const float2* __restrict__ twiddle, float* __restrict__ frames,
                                                              int F, int Fp, int Tp, int ps) {
    extern __shared__ __attribute__((aligned(16))) float2 fft_smem[];
    const int zstride = N + (N >> ps) + FFT_ZPAD;
    float2* z = fft_smem;
    float2* tw = fft_smem + TB * zstride;
    const int groups = (T + TB - 1) / TB;
    const int npairs = nsig / 2;
    int id = blockIdx.x;
    const int g = id % groups;
    id /= groups;
    const int pr = id % npairs, b = id / npairs;
    const int t0 = g * TB;
    const long plane = (long)Fp * Tp;
    const float2* Sa = spec + ((long)b * nsig + 2 * pr) * plane;
    const float2* Sb = Sa + plane;

    for (int i = threadIdx.x; i < N / 2; i += FFT_NT) tw[i] = twiddle[i];
    // (eight bins per thread at a time, every fetch before the first placement: see istft_fused_kernel)
    constexpr int UN = 8;
    for (int base = threadIdx.x; base < F * TB; base += FFT_NT * UN) {
        float2 va[UN], vb[UN];
#pragma unroll
        for (int j = 0; j < UN; ++j) {
            const int idx = min(base + j * FFT_NT, F * TB - 1);
            const int f = idx / TB, tb = idx - f * TB;
            const long o = (long)f * Tp + min(t0 + tb, T - 1);
            va[j] = Sa[o];
            vb[j] = Sb[o];
        }
#pragma unroll
        for (int j = 0; j < UN; ++j) {
            const int idx = base + j * FFT_NT;
            if (idx < F * TB) {
                const int f = idx / TB, tb = idx - f * TB;
                float2 fa = make_float2(0.f, 0.f), fb = fa;
                if (t0 + tb < T) {
                    fa = make_float2(va[j].x, -va[j].y);   // istft undoes the stored conjugate (librosaSTFT.py:278)
                    fb = make_float2(vb[j].x, -vb[j].y);
                }
                if (f == 0 || f == N / 2) {          // ifft(...).real keeps only the real part of these two bins
                    fa.y = 0.f;
                    fb.y = 0.f;
                }
                float2* zz = z + tb * zstride;
                zz[fft_pad(bitrev(f, logN), ps)] = make_float2(fa.x - fb.y, fa.y + fb.x);
                if (f != 0 && f != N / 2) zz[fft_pad(bitrev(N - f, logN), ps)] = make_float2(fa.x + fb.y, fb.x - fa.y);
            }
        }
    }
    __syncthreads();
    fft_stages_any<true, TB>(z, tw, N, logN, zstride, ps);

    const float invN = 1.0f / (float)N;
    float* Fa = frames + (((long)b * nsig + 2 * pr) * T) * N;
    float* Fb = Fa + (long)T * N;
    for (int idx = threadIdx.x; idx < TB * N; idx += FFT_NT) {
        const int tb = idx / N, n = idx - tb * N;
        const int t = t0 + tb;
        if (t >= T) continue;
        const float2 v = z[tb * zstride + fft_pad(n, ps)];
        const float w = window[n];
        Fa[(long)t * N + n] = w * (v.x * invN);
        Fb[(long)t * N + n] = w * (v.y * invN);
    }
}

// Overlap-add in ASCENDING frame order (the order librosaSTFT.py:275-281 accumulates in), centre trim
// of `trim` samples at both ends (:283-284), times the gain (gccNMFFunctions.py:155,163).
// grid = (ceil(L/256), nsig, batch).
// halo > 0: the frame sequence of a signal is `halo` frames from `prev` ([sig][halo][N], e.g. the previous time shard's last frames)
// followed by the T - halo frames of `frames` ([sig][T - halo][N]).
__global__ __launch_bounds__(256) void istft_ola_kernel(const float* __restrict__ frames, int N, int hop, int T, int L, int trim,
                                                        float gain, float* __restrict__ y, const float* __restrict__ prev = nullptr,
                                                        int halo = 0) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= L) return;
    const long sig = (long)blockIdx.z * gridDim.y + blockIdx.y;
    const float* fr = frames + sig * (long)(T - halo) * N;
    const float* pv = prev + sig * (long)halo * N;
    const int s = m + trim;
    int t_lo = (s - N + hop) / hop;       // ceil((s - N + 1) / hop) for s - N + 1 > 0
    if (s - N + 1 <= 0) t_lo = 0;
    int t_hi = s / hop;
    if (t_hi > T - 1) t_hi = T - 1;
    float acc = 0.f;
    for (int t = t_lo; t <= t_hi; ++t) acc = acc + (t < halo ? pv[(long)t * N + (s - t * hop)] : fr[(long)(t - halo) * N + (s - t * hop)]);
    y[sig * L + m] = acc * gain;
}


// Inverse STFT AND overlap-add in one pass (the windowed time frames never go to HBM: 15 MB written + read back per file at
// K = 1024 in the two-kernel form).  A workgroup owns the output samples of G = ISTFT_SUB * ISTFT_TB consecutive hops of one
// signal pair; it transforms frames t0-H .. t0+G-1 (H = ceil(N/hop)-1 halo frames that also reach into its range: 9 % more
// spectrogram reads at N/hop = 4) in ascending sub-batches of ISTFT_TB frames and adds each frame into a sliding LDS accumulator
// in ASCENDING frame order, every sample starting from 0 -- the order librosaSTFT.py:275-281 accumulates in; the result is bit for
// bit that of istft_frames_kernel + istft_ola_kernel (no fma contraction in this file, same accumulation).
// grid = batch * (nsig/2) * ceil(T / G).
// (lab build: shader clocks per phase of a workgroup -- hand-out / shift, spectrogram fetch + placement, butterflies, overlap-add -- into the trace
// buffer of gccnmf_debug_set_trace, scripts/ktrace_istft.py; the product build carries none of it)
#ifdef GCCNMF_EXPERIMENTS
extern long long* gccnmf_trace_buf;
extern int gccnmf_trace_blocks;
#define ISTFT_TRACE_BEGIN() long long tr_acc[4] = {0, 0, 0, 0}, tr_t = 0, tr_t0 = 0; if (trace && threadIdx.x == 0) { tr_t0 = __builtin_amdgcn_s_memrealtime(); tr_t = __builtin_amdgcn_s_memtime(); }
#define ISTFT_TRACE(k) if (trace && threadIdx.x == 0) { const long long n_ = __builtin_amdgcn_s_memtime(); tr_acc[k] += n_ - tr_t; tr_t = n_; }
#define ISTFT_TRACE_END() if (trace && threadIdx.x == 0) { long long* r_ = trace + 8L * blockIdx.x; r_[0] = tr_t0; r_[1] = tr_acc[0]; r_[2] = tr_acc[1]; r_[3] = tr_acc[2]; r_[4] = tr_acc[3]; r_[5] = __builtin_amdgcn_s_memrealtime(); }
#else
#define ISTFT_TRACE_BEGIN()
#define ISTFT_TRACE(k)
#define ISTFT_TRACE_END()
#endif
#ifndef ISTFT_TB
#define ISTFT_TB 4
#endif
#ifndef ISTFT_SUB
#define ISTFT_SUB 8
#endif
__global__ __launch_bounds__(FFT_NT, 2) void istft_fused_kernel(const float2* __restrict__ spec, int nsig, int N, int logN, int hop, int T,
                                                             const float* __restrict__ window, const float2* __restrict__ twiddle,
                                                             int F, int Fp, int Tp, int trim, int L, float gain, float* __restrict__ y, int ps, long long* trace) {
    extern __shared__ __attribute__((aligned(16))) float2 fft_smem[];
    constexpr int TB = ISTFT_TB, G = ISTFT_TB * ISTFT_SUB;
    ISTFT_TRACE_BEGIN();
    const int zstride = N + (N >> ps) + FFT_ZPAD;
    float2* z = fft_smem;
    float2* tw = fft_smem + TB * zstride;
    const int span = N + hop * (TB - 1);                  // samples one sub-batch touches
    float* acc_a = (float*)(tw + N / 2);                  // [span] per signal of the pair
    float* acc_b = acc_a + span;
    float* s_win = acc_b + span;                          // [N] the synthesis window
    const int groups = (T + G - 1) / G, npairs = nsig / 2;
    int id = blockIdx.x;
    const int g = id % groups;
    id /= groups;
    const int pr = id % npairs, b = id / npairs;
    const int t0 = g * G, t_end = min(t0 + G, T);
    const int halo = (N + hop - 1) / hop - 1;
    const int t_first = max(t0 - halo, 0);
    const long plane = (long)Fp * Tp;
    const float2* Sa = spec + ((long)b * nsig + 2 * pr) * plane;
    const float2* Sb = Sa + plane;
    float* ya = y + ((long)b * nsig + 2 * pr) * L;
    float* yb = ya + L;
    // owned samples of the untrimmed stream: [t0 * hop, t_end * hop), the last group to the end of the last frame
    const long own_lo = (long)t0 * hop, own_hi = (t_end == T) ? (long)(T - 1) * hop + N : (long)t_end * hop;
    const float invN = 1.0f / (float)N;

    for (int i = threadIdx.x; i < N / 2; i += FFT_NT) tw[i] = twiddle[i];
    for (int i = threadIdx.x; i < N; i += FFT_NT) s_win[i] = window[i];
    for (int i = threadIdx.x; i < 2 * span; i += FFT_NT) acc_a[i] = 0.f;
    long base = (long)t_first * hop;                      // untrimmed sample index of acc[0]
    for (int fs = t_first; fs < t_end; fs += TB) {
        // (an opaque copy of the thread index per sub-batch: the unrolled phases below derive ~80 addresses from it, and as invariants of this loop they
        // were hoisted in front of it and spilled -- 320 bytes of scratch, reloaded inside the loop)
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        // samples before this sub-batch's first frame are final: hand them out, slide the accumulator
        const int shift = (int)((long)fs * hop - base);
        if (shift > 0) {
            __syncthreads();
            // (all LDS reads of the thread first -- clamped, masked afterwards; span <= 8 FFT_NT for n_fft <= 2048)
            float out_a[8], out_b[8], keep_a[8], keep_b[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int i = tid + r * FFT_NT;
                const int ic = min(i, span - 1), is = min(i + shift, span - 1);
                out_a[r] = acc_a[ic];
                out_b[r] = acc_b[ic];
                keep_a[r] = acc_a[is];
                keep_b[r] = acc_b[is];
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int i = tid + r * FFT_NT;
                if (i < shift && i < span) {
                    const long sg = base + i;
                    if (sg >= own_lo && sg < own_hi) {
                        const long m = sg - trim;
                        if (m >= 0 && m < L) {
                            ya[m] = out_a[r] * gain;
                            yb[m] = out_b[r] * gain;
                        }
                    }
                }
                if (i + shift >= span) keep_a[r] = keep_b[r] = 0.f;
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int i = tid + r * FFT_NT;
                if (i < span) {
                    acc_a[i] = keep_a[r];
                    acc_b[i] = keep_b[r];
                }
            }
            base += shift;
        }
        __syncthreads();
        ISTFT_TRACE(0);
        // bin f of frame row tb of the sub-batch: the pair's two spectra as ONE complex input, in bit-reversed order
        auto place = [&](int f, int tb, bool valid, float2 sa, float2 sb) {
            float2 fa = make_float2(0.f, 0.f), fb = fa;
            if (valid) {
                fa = make_float2(sa.x, -sa.y);   // istft undoes the stored conjugate (librosaSTFT.py:278)
                fb = make_float2(sb.x, -sb.y);
            }
            if (f == 0 || f == N / 2) {          // ifft(...).real keeps only the real part of these two bins
                fa.y = 0.f;
                fb.y = 0.f;
            }
            float2* zz = z + tb * zstride;
            zz[fft_pad(bitrev(f, logN), ps)] = make_float2(fa.x - fb.y, fa.y + fb.x);
            if (f != 0 && f != N / 2) zz[fft_pad(bitrev(N - f, logN), ps)] = make_float2(fa.x + fb.y, fb.x - fa.y);
        };
        // Every fetch of the thread is issued before the first placement (clamped addresses, masked afterwards: a conditional load ends a basic
        // block and its wait).  Left as one loop, the compiler's code was fetch - wait - place per trip: nine dependent round trips to the memory
        // side per sub-batch, 35 % of a workgroup's clocks (profiles/r06ah_ktrace_istft.txt).
        constexpr int KI = (513 * TB + FFT_NT - 1) / FFT_NT;      // (bin, frame) items per thread, n_fft <= 1024
        if (F * TB <= KI * FFT_NT) {
            float2 va[KI], vb[KI];
#pragma unroll
            for (int j = 0; j < KI; ++j) {
                const int idx = min(tid + j * FFT_NT, F * TB - 1);
                const int f = idx / TB, tb = idx - f * TB;
                const long o = (long)f * Tp + min(fs + tb, t_end - 1);
                va[j] = Sa[o];
                vb[j] = Sb[o];
            }
#pragma unroll
            for (int j = 0; j < KI; ++j) {
                const int idx = tid + j * FFT_NT;
                if (idx < F * TB) {
                    const int f = idx / TB, tb = idx - f * TB;
                    place(f, tb, fs + tb < t_end, va[j], vb[j]);
                }
            }
        } else {
            for (int idx = threadIdx.x; idx < F * TB; idx += FFT_NT) {
                const int f = idx / TB, tb = idx - f * TB;
                const int t = fs + tb;
                float2 sa = make_float2(0.f, 0.f), sb = sa;
                if (t < t_end) {
                    sa = Sa[(long)f * Tp + t];
                    sb = Sb[(long)f * Tp + t];
                }
                place(f, tb, t < t_end, sa, sb);
            }
        }
        __syncthreads();
        ISTFT_TRACE(1);
        fft_stages_any<true, TB>(z, tw, N, logN, zstride, ps);
        ISTFT_TRACE(2);
        // every thread owns the accumulator positions i = tid, tid + NT, ...: frames added in ascending order, no hazards.  Branch-free reads
        // (clamped indexes, the window from LDS) so that the reads of all positions can be in flight together; the additions are masked.
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if ((r & 1) == 0) __builtin_amdgcn_sched_barrier(0);      // (two positions' reads at a time: all eight together spill)
            const int i = tid + r * FFT_NT;
            const int ic = min(i, span - 1);
            float va = acc_a[ic], vb = acc_b[ic];
            float2 v[TB];
            float w[TB];
#pragma unroll
            for (int tb = 0; tb < TB; ++tb) {
                const int nc = min(max(ic - tb * hop, 0), N - 1);
                v[tb] = z[tb * zstride + fft_pad(nc, ps)];
                w[tb] = s_win[nc];
            }
#pragma unroll
            for (int tb = 0; tb < TB; ++tb) {
                const int n = ic - tb * hop;
                // (no fma across the window product: this file is compiled with -ffp-contract=off; the reference adds whole frames)
                const float na = va + w[tb] * (v[tb].x * invN), nb = vb + w[tb] * (v[tb].y * invN);
                const bool on = n >= 0 && n < N && fs + tb < t_end;
                va = on ? na : va;
                vb = on ? nb : vb;
            }
            if (i < span) {
                acc_a[i] = va;
                acc_b[i] = vb;
            }
        }
        ISTFT_TRACE(3);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < span; i += FFT_NT) {
        const long sg = base + i;
        if (sg >= own_lo && sg < own_hi) {
            const long m = sg - trim;
            if (m >= 0 && m < L) {
                ya[m] = acc_a[i] * gain;
                yb[m] = acc_b[i] * gain;
            }
        }
    }
    ISTFT_TRACE(0);
    ISTFT_TRACE_END();
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

static inline long long* istft_trace(int grid) {
#ifdef GCCNMF_EXPERIMENTS
    return (gccnmf_trace_buf && grid <= gccnmf_trace_blocks) ? gccnmf_trace_buf : nullptr;
#else
    return nullptr;
#endif
}
static inline int fft_ps() { return gccnmf_tune_fft_r16 ? 4 : FFT_NOPAD; }
static inline size_t fft_rows_bytes(int tb, int n_fft, int ps) { return sizeof(float2) * ((size_t)tb * fft_row_floats2(n_fft, ps) + n_fft / 2); }

extern "C" {

static int launch_stft(const void* x, long x_stride, int n_samples, int n_fft, int hop, int T, int batch, const float* window,
                       const float* twiddle, float* X, float* V, float* CC, bool pcm16, void* stream) {
    const int logN = ilog2_exact(n_fft);
    if (!x || !window || !twiddle || !X || logN < 6 || logN > 12 || hop < 1 || T < 1 || batch < 1) return GCCNMF_ERR_ARG;
    if ((long)(T - 1) * hop + n_fft > n_samples) return GCCNMF_ERR_ARG;
    const int F = n_fft / 2 + 1;
    GccNmfPitches p = gccnmf_make_pitches(F, T, 1);
    // eight frames per workgroup; four when eight no longer fit the 160 KB of LDS (n_fft = 4096)
    const int ps = fft_ps();
    const bool small_tb = fft_rows_bytes(FFT_TB, n_fft, ps) > 160 * 1024;
    const int tb = small_tb ? FFT_TB / 2 : FFT_TB;
    const size_t lds = fft_rows_bytes(tb, n_fft, ps);
    if (lds > 160 * 1024) return GCCNMF_ERR_UNSUPPORTED;
    const void* fn = pcm16 ? (small_tb ? (const void*)stft_stereo_kernel<true, FFT_TB / 2> : (const void*)stft_stereo_kernel<true, FFT_TB>)
                           : (small_tb ? (const void*)stft_stereo_kernel<false, FFT_TB / 2> : (const void*)stft_stereo_kernel<false, FFT_TB>);
    if (lds > 64 * 1024) {
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return GCCNMF_ERR_LAUNCH;
    }
    const int groups = gccnmf_ceil_div(T, tb);
#define GCCNMF_LAUNCH_STFT(P_, TB_)                                                                                                       \
    hipLaunchKernelGGL((stft_stereo_kernel<P_, TB_>), dim3(batch * groups), dim3(FFT_NT), lds, (hipStream_t)stream, (const float*)x, x_stride, \
                       n_samples, n_fft, logN, hop, T, window, (const float2*)twiddle, (float2*)X, V, CC, F, p.Fp, p.Np, p.Tp, ps)
    if (pcm16) {
        if (small_tb) GCCNMF_LAUNCH_STFT(true, FFT_TB / 2);
        else GCCNMF_LAUNCH_STFT(true, FFT_TB);
    } else {
        if (small_tb) GCCNMF_LAUNCH_STFT(false, FFT_TB / 2);
        else GCCNMF_LAUNCH_STFT(false, FFT_TB);
    }
#undef GCCNMF_LAUNCH_STFT
    GCCNMF_CHECK_LAUNCH();
    return GCCNMF_OK;
}

int gccnmf_stft_stereo(const float* x, long x_stride, int n_samples, int n_fft, int hop, int T, int batch, const float* window,
                       const float* twiddle, float* X, float* V, float* CC, void* stream) {
    GCCNMF_ENTER();
    return launch_stft(x, x_stride, n_samples, n_fft, hop, T, batch, window, twiddle, X, V, CC, false, stream);
}

int gccnmf_stft_stereo_pcm16(const short* pcm, long frame_stride, int n_samples, int n_fft, int hop, int T, int batch,
                             const float* window, const float* twiddle, float* X, float* V, float* CC, void* stream) {
    GCCNMF_ENTER();
    return launch_stft(pcm, frame_stride, n_samples, n_fft, hop, T, batch, window, twiddle, X, V, CC, true, stream);
}

int gccnmf_istft_ola(const float* spec, int nsig, int n_fft, int hop, int T, int batch, const float* window,
                     const float* twiddle, float gain, int center, float* frames, float* y, void* stream) {
    GCCNMF_ENTER();
    const int logN = ilog2_exact(n_fft);
    if (!spec || !window || !twiddle || !y || logN < 6 || logN > 12 || hop < 1 || T < 1 || batch < 1 || nsig < 2 || (nsig & 1))
        return GCCNMF_ERR_ARG;
    const int F = n_fft / 2 + 1;
    const int ps = fft_ps();
    GccNmfPitches p = gccnmf_make_pitches(F, T, 1);
    if (!frames) {      // fused form: no frame buffer, one pass
        const int trim = center ? n_fft / 2 : 0;
        const int L = n_fft + hop * (T - 1) - 2 * trim;
        if (L < 1) return GCCNMF_ERR_ARG;
        const int span = n_fft + hop * (ISTFT_TB - 1);
        if (span > 8 * FFT_NT) return GCCNMF_ERR_UNSUPPORTED;          // hop > n_fft / 3 or so: use the two-kernel form
        const size_t lds = fft_rows_bytes(ISTFT_TB, n_fft, ps) + sizeof(float) * (2 * span + n_fft);
        if (lds > 64 * 1024) {
            if (hipFuncSetAttribute((const void*)istft_fused_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
                return GCCNMF_ERR_LAUNCH;
        }
        const int groups = gccnmf_ceil_div(T, ISTFT_TB * ISTFT_SUB);
        hipLaunchKernelGGL(istft_fused_kernel, dim3(batch * (nsig / 2) * groups), dim3(FFT_NT), lds, (hipStream_t)stream, (const float2*)spec,
                           nsig, n_fft, logN, hop, T, window, (const float2*)twiddle, F, p.Fp, p.Tp, trim, L, gain, y, ps, istft_trace(batch * (nsig / 2) * groups));
        GCCNMF_CHECK_LAUNCH();
        return GCCNMF_OK;
    }
    const bool small_tb = fft_rows_bytes(FFT_TB, n_fft, ps) > 160 * 1024;
    const int tb = small_tb ? FFT_TB / 2 : FFT_TB;
    const size_t lds = fft_rows_bytes(tb, n_fft, ps);
    if (lds > 160 * 1024) return GCCNMF_ERR_UNSUPPORTED;
    if (lds > 64 * 1024) {
        const void* fn = small_tb ? (const void*)istft_frames_kernel<FFT_TB / 2> : (const void*)istft_frames_kernel<FFT_TB>;
        if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return GCCNMF_ERR_LAUNCH;
    }
    const int groups = gccnmf_ceil_div(T, tb);
    hipStream_t s = (hipStream_t)stream;
    if (small_tb)
        hipLaunchKernelGGL(istft_frames_kernel<FFT_TB / 2>, dim3(batch * (nsig / 2) * groups), dim3(FFT_NT), lds, s, (const float2*)spec, nsig,
                           n_fft, logN, T, window, (const float2*)twiddle, frames, F, p.Fp, p.Tp, ps);
    else
        hipLaunchKernelGGL(istft_frames_kernel<FFT_TB>, dim3(batch * (nsig / 2) * groups), dim3(FFT_NT), lds, s, (const float2*)spec, nsig,
                           n_fft, logN, T, window, (const float2*)twiddle, frames, F, p.Fp, p.Tp, ps);
    GCCNMF_CHECK_LAUNCH();
    const int trim = center ? n_fft / 2 : 0;
    const int L = n_fft + hop * (T - 1) - 2 * trim;
    if (L < 1) return GCCNMF_ERR_ARG;
    hipLaunchKernelGGL(istft_ola_kernel, dim3(gccnmf_ceil_div(L, 256), nsig, batch), dim3(256), 0, s, frames, n_fft, hop, T, L,
                       trim, gain, y);
    GCCNMF_CHECK_LAUNCH();
    return GCCNMF_OK;
}

int gccnmf_ola_frames_halo(const float* prev, int halo, const float* frames, int nsig, int n_fft, int hop, int T, int first_sample, int L,
                           float gain, float* y, void* stream) {
    GCCNMF_ENTER();
    if (!frames || !y || nsig < 1 || n_fft < 2 || hop < 1 || T < 1 || halo < 0 || (halo > 0 && !prev) || first_sample < 0 || L < 1 ||
        (long)first_sample + L > (long)n_fft + (long)hop * (halo + T - 1))
        return GCCNMF_ERR_ARG;
    hipLaunchKernelGGL(istft_ola_kernel, dim3(gccnmf_ceil_div(L, 256), nsig, 1), dim3(256), 0, (hipStream_t)stream, frames, n_fft, hop, halo + T,
                       L, first_sample, gain, y, prev, halo);
    GCCNMF_CHECK_LAUNCH();
    return GCCNMF_OK;
}

int gccnmf_pack_pcm16(const float* y, int groups, int L, unsigned int* peak_scratch, short* pcm, void* stream) {
    GCCNMF_ENTER();
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

// ---- any n_fft: the DFT as a real GEMM on the matrix cores -------------------------------------------------------------------------
// The reference's stft / istft take ANY n_fft (scipy.fftpack, librosaSTFT.py:162-179, :276-279); the radix-2 kernels above take powers of
// two.  Other sizes are rare (off the GCC-NMF path's defaults) and small, so they do not get a mixed-radix FFT: the transform is the
// matrix product it is, on v_mfma_f32_32x32x2_f32 (exact-f32 fmaf chains) through the same GEMM kernels as the rest of the path --
//   forward:  [Re X; Im X] (2Fp x T) = basis^T (2Fp x N) . frames (N x T),   basis[n][f] = w[n] cos(2 pi f n / N), [n][Fp+f] = w[n] sin(..)
//             (the reference stores conj(fft): Im X = + sum w y sin)
//   inverse:  frames (T x N) = [Re S; Im S]^T (T x 2Fp) . ibasis (2Fp x N),  ibasis[k][n] = c_k w[n] cos(2 pi k n / N) / N, [Fp+k][n] = c_k w[n] sin(..) / N
//             with c_0 = c_{N/2} = 1, c_k = 2 otherwise: the real part of ifft([conj(S), S[-2:0:-1]]) (librosaSTFT.py:277-279)
// Both tables are evaluated in float64 on the host (like the twiddles).  2 N F flop per frame and signal: N = 1000 -> 1 MFLOP.
int gccnmf_gemm_nn_store(const float* A, const float* B, float* C, int M, int N, int Kd, int lda, int ldb, int ldc, int batch, long sA,
                         long sB, long sC, hipStream_t s);        // gcc.hip: C = A^T-layout . B, both operands [reduction][.]

// framesT[sig][n][t] = x[sig][t*hop + n]   (grid = (ceil(Tp/256), N, nsig); rows n >= N and columns t >= T are never written: zero)
__global__ __launch_bounds__(256) void dft_frames_kernel(const float* __restrict__ x, long x_stride, int N, int Np16, int hop, int T, int Tp,
                                                         float* __restrict__ framesT) {
    const int t = blockIdx.x * 256 + threadIdx.x, n = blockIdx.y;
    if (t >= T) return;
    framesT[((long)blockIdx.z * Np16 + n) * Tp + t] = x[blockIdx.z * x_stride + (long)t * hop + n];
}

// planes[sig][2Fp][Tp] -> X[sig][Fp][Tp] interleaved complex (valid F x T region only)
__global__ __launch_bounds__(256) void dft_pack_kernel(const float* __restrict__ planes, int F, int Fp, int T, int Tp, float2* __restrict__ X) {
    const int t = blockIdx.x * 256 + threadIdx.x, f = blockIdx.y;
    if (t >= T) return;
    const float* P = planes + (long)blockIdx.z * 2 * Fp * Tp;
    X[((long)blockIdx.z * Fp + f) * Tp + t] = make_float2(P[(long)f * Tp + t], P[(long)(Fp + f) * Tp + t]);
}

// spec[sig][Fp][Tp] complex -> planes[sig][2Fp][Tp]
__global__ __launch_bounds__(256) void dft_unpack_kernel(const float2* __restrict__ spec, int F, int Fp, int T, int Tp, float* __restrict__ planes) {
    const int t = blockIdx.x * 256 + threadIdx.x, f = blockIdx.y;
    if (t >= T) return;
    const float2 v = spec[((long)blockIdx.z * Fp + f) * Tp + t];
    float* P = planes + (long)blockIdx.z * 2 * Fp * Tp;
    P[(long)f * Tp + t] = v.x;
    P[(long)(Fp + f) * Tp + t] = v.y;
}

extern "C" {

long gccnmf_dft_workspace_floats(int n_fft, int T, int nsig) {
    GCCNMF_ENTER();
    if (n_fft < 2 || T < 1 || nsig < 1) return -1;
    GccNmfPitches p = gccnmf_make_pitches(n_fft / 2 + 1, T, 1);
    const long fwd = (long)nsig * ((long)gccnmf_round_up(n_fft, 16) + 2L * p.Fp) * p.Tp;      // framesT | planes
    const long inv = (long)nsig * (2L * p.Fp * p.Tp + (long)T * n_fft);                         // planes | frames
    return fwd > inv ? fwd : inv;
}

int gccnmf_stft_dft(const float* x, long x_stride, int n_samples, int n_fft, int hop, int T, int nsig, const float* basis,
                    float* workspace, float* X, void* stream) {
    GCCNMF_ENTER();
    if (!x || !basis || !workspace || !X || n_fft < 2 || n_fft > 8192 || hop < 1 || T < 1 || nsig < 1) return GCCNMF_ERR_ARG;
    if ((long)(T - 1) * hop + n_fft > n_samples) return GCCNMF_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int F = n_fft / 2 + 1, Np16 = gccnmf_round_up(n_fft, 16);
    GccNmfPitches p = gccnmf_make_pitches(F, T, 1);
    float* framesT = workspace;
    float* planes = framesT + (long)nsig * Np16 * p.Tp;
    if (hipMemsetAsync(framesT, 0, sizeof(float) * nsig * Np16 * p.Tp, s) != hipSuccess) return GCCNMF_ERR_LAUNCH;
    hipLaunchKernelGGL(dft_frames_kernel, dim3(gccnmf_ceil_div(T, 256), n_fft, nsig), dim3(256), 0, s, x, x_stride, n_fft, Np16, hop, T, p.Tp, framesT);
    GCCNMF_CHECK_LAUNCH();
    int rc = gccnmf_gemm_nn_store(basis, framesT, planes, 2 * p.Fp, T, n_fft, 2 * p.Fp, p.Tp, p.Tp, nsig, 0L, (long)Np16 * p.Tp, 2L * p.Fp * p.Tp, s);
    if (rc) return rc;
    hipLaunchKernelGGL(dft_pack_kernel, dim3(gccnmf_ceil_div(T, 256), F, nsig), dim3(256), 0, s, planes, F, p.Fp, T, p.Tp, (float2*)X);
    GCCNMF_CHECK_LAUNCH();
    return GCCNMF_OK;
}

int gccnmf_istft_dft(const float* spec, int nsig, int n_fft, int hop, int T, const float* ibasis, float gain, int center, float* workspace,
                     float* y, void* stream) {
    GCCNMF_ENTER();
    if (!spec || !ibasis || !workspace || !y || n_fft < 2 || n_fft > 8192 || (n_fft & 1) || hop < 1 || T < 1 || nsig < 1) return GCCNMF_ERR_ARG;
    const int trim = center ? n_fft / 2 : 0;
    const int L = n_fft + hop * (T - 1) - 2 * trim;
    if (L < 1) return GCCNMF_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const int F = n_fft / 2 + 1, Nb = gccnmf_round_up(n_fft, 64);
    GccNmfPitches p = gccnmf_make_pitches(F, T, 1);
    float* planes = workspace;
    float* frames = planes + (long)nsig * 2 * p.Fp * p.Tp;
    if (hipMemsetAsync(planes, 0, sizeof(float) * nsig * 2 * p.Fp * p.Tp, s) != hipSuccess) return GCCNMF_ERR_LAUNCH;
    hipLaunchKernelGGL(dft_unpack_kernel, dim3(gccnmf_ceil_div(T, 256), F, nsig), dim3(256), 0, s, (const float2*)spec, F, p.Fp, T, p.Tp, planes);
    GCCNMF_CHECK_LAUNCH();
    int rc = gccnmf_gemm_nn_store(planes, ibasis, frames, T, n_fft, 2 * p.Fp, p.Tp, Nb, n_fft, nsig, 2L * p.Fp * p.Tp, 0L, (long)T * n_fft, s);
    if (rc) return rc;
    hipLaunchKernelGGL(istft_ola_kernel, dim3(gccnmf_ceil_div(L, 256), nsig, 1), dim3(256), 0, s, frames, n_fft, hop, T, L, trim, gain, y);
    GCCNMF_CHECK_LAUNCH();
    return GCCNMF_OK;
}

}  // extern "C"
