// Radix-2 complex FFT in LDS shared by the offline STFT/iSTFT kernels (fft.hip) and the streaming frame
// processor (rt.hip): TB frames advance through the log2(N) butterfly stages together (log2(N) barriers per TB frames).
#pragma once
#include "common.h"

#define FFT_TB 8
#define FFT_NT 256
#define FFT_ZPAD 9   // row padding (in complex elements) of the per-frame LDS buffers

// fft.hip is compiled with -ffp-contract=off (Makefile): the compiler's own fma contraction differs from kernel to kernel, and the
// offline kernels that share this FFT must produce the same bits (the fused and the two-kernel inverse STFT are chosen by launch
// size).  HIP's __fmul_rn / __fadd_rn are plain operators to the optimiser and do not prevent it.
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// Input must already be in bit-reversed order.  INVERSE uses conjugated twiddles (no 1/N scaling here).
// The butterflies of a stage are independent, so a thread takes them FFT_UN at a time: all LDS loads of a batch are issued before
// the first store (the compiler cannot move a load across a store into the same array itself; one butterfly per round trip made a
// stage 16 dependent LDS latencies long: 0.24 of the STFT's 0.60 ms).  Same operations per butterfly in the same order: same bits.
// LDS position of element i of a frame: `ps` = 4 pads one element per 16 (the register passes below read 16 consecutive elements per
// lane: without it every lane of a wave would start in the same bank), `ps` = 30 is the plain layout (i >> 30 == 0).
#define FFT_NOPAD 30
__device__ __forceinline__ int fft_pad(int i, int ps) { return i + (i >> ps); }
static inline int fft_row_floats2(int N, int ps) { return N + (N >> ps) + FFT_ZPAD; }

template <bool INVERSE, int TB, int UN>
__device__ __forceinline__ void fft_stages_un(float2* z, const float2* tw, int N, int logN, int zstride, int ps = FFT_NOPAD) {
    const int half_n = N >> 1, total = TB * half_n;
    for (int s = 1; s <= logN; ++s) {
        const int half = 1 << (s - 1);
        const int tw_step = N >> s;
        for (int base = threadIdx.x; base < total; base += FFT_NT * UN) {
            float2 u[UN], v[UN], w[UN];
            int i0[UN];
#pragma unroll
            for (int j = 0; j < UN; ++j) {
                const int idx = base + j * FFT_NT;
                if (idx < total) {
                    const int tb = idx >> (logN - 1), bf = idx & (half_n - 1);
                    const int grp = bf >> (s - 1), pos = bf & (half - 1);
                    i0[j] = (grp << s) + pos;
                    w[j] = tw[pos * tw_step];
                    u[j] = z[tb * zstride + fft_pad(i0[j], ps)];
                    v[j] = z[tb * zstride + fft_pad(i0[j] + half, ps)];
                    i0[j] += tb << 16;               // (frame index and element index travel in one register)
                }
            }
#pragma unroll
            for (int j = 0; j < UN; ++j) {
                const int idx = base + j * FFT_NT;
                if (idx < total) {
                    float2 wj = w[j];
                    if (INVERSE) wj.y = -wj.y;
                    const float2 t = cmul(wj, v[j]);
                    const int tb = i0[j] >> 16, e0 = i0[j] & 0xffff;
                    z[tb * zstride + fft_pad(e0, ps)] = make_float2(u[j].x + t.x, u[j].y + t.y);
                    z[tb * zstride + fft_pad(e0 + half, ps)] = make_float2(u[j].x - t.x, u[j].y - t.y);
                }
            }
        }
        __syncthreads();
    }
}

// (eight at a time when a thread has at least four butterflies per stage -- the offline kernels: 16; one at a time otherwise -- the
// streaming processor's single frame: one butterfly per thread, where the eight predicated slots cost 20 us per block)
template <bool INVERSE, int TB = FFT_TB>
__device__ __forceinline__ void fft_stages(float2* z, const float2* tw, int N, int logN, int zstride, int ps = FFT_NOPAD) {
    if (TB * (N >> 1) >= 4 * FFT_NT) fft_stages_un<INVERSE, TB, 8>(z, tw, N, logN, zstride, ps);
    else fft_stages_un<INVERSE, TB, 1>(z, tw, N, logN, zstride, ps);
}

// The same radix-2 butterflies, up to four stages per trip through LDS (round 4).  Stages sa .. sa+R-1 only combine elements whose
// indexes differ in bits sa-1 .. sa+R-2, so a thread that holds the 2^R elements of one such group runs all R stages on registers:
// ten stages of a 1024-point frame become three passes (4 + 4 + 2 stages) -- three barriers and three LDS round trips instead of
// ten, and the twiddles of a pass (they depend on the thread's position inside the group pattern only) are fetched once.  Every
// butterfly computes t = w * v; (u + t, u - t) with the same w, u, v as in fft_stages_un, in the same operation order (this file's users
// are built with -ffp-contract=off): bit-identical results, whichever routine runs (tests/test_gpu_kernels.py checks it on hardware).
// Needs the padded layout (ps = 4) and a whole number of groups per thread stride: FFT_NT a multiple of 2^(sa-1) for every pass.
template <bool INVERSE, int TB, int R>
__device__ __forceinline__ void fft_pass(float2* z, const float2* tw, int N, int logN, int zstride, int sa) {
    constexpr int E = 1 << R;
    const int lo_bits = sa - 1, per_frame = N >> R, ngroups = TB * per_frame;
    const int lo = threadIdx.x & ((1 << lo_bits) - 1);            // the same for every group of this thread (FFT_NT is a multiple of 2^lo_bits)
    // twiddles of the pass: stage sa + r pairs j0 (bit r clear) with j0 | 1 << r; pos = lo + (low r bits of j0) << lo_bits
    float2 w[E - 1];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int tw_step = N >> (sa + r);
#pragma unroll
        for (int q = 0; q < (1 << r); ++q) {
            float2 t = tw[(lo + (q << lo_bits)) * tw_step];
            if (INVERSE) t.y = -t.y;
            w[(1 << r) - 1 + q] = t;
        }
    }
    for (int g = threadIdx.x; g < ngroups; g += FFT_NT) {
        const int tb = g / per_frame, q = g - tb * per_frame;
        const int base = ((q >> lo_bits) << (lo_bits + R)) + (q & ((1 << lo_bits) - 1));
        float2* row = z + tb * zstride;
        float2 e[E];
#pragma unroll
        for (int j = 0; j < E; ++j) e[j] = row[fft_pad(base + (j << lo_bits), 4)];
#pragma unroll
        for (int r = 0; r < R; ++r) {
#pragma unroll
            for (int j0 = 0; j0 < E; ++j0) {
                if (j0 & (1 << r)) continue;
                const int j1 = j0 | (1 << r);
                const float2 t = cmul(w[(1 << r) - 1 + (j0 & ((1 << r) - 1))], e[j1]);
                const float2 u = e[j0];
                e[j0] = make_float2(u.x + t.x, u.y + t.y);
                e[j1] = make_float2(u.x - t.x, u.y - t.y);
            }
        }
#pragma unroll
        for (int j = 0; j < E; ++j) row[fft_pad(base + (j << lo_bits), 4)] = e[j];
    }
    __syncthreads();
}

template <bool INVERSE, int TB, int R>
__device__ __forceinline__ void fft_pass_r(float2* z, const float2* tw, int N, int logN, int zstride, int sa, int r) {
    // r = stages of this pass (1 .. R), chosen at run time
    if (r == R) fft_pass<INVERSE, TB, R>(z, tw, N, logN, zstride, sa);
    else if constexpr (R > 1) fft_pass_r<INVERSE, TB, R - 1>(z, tw, N, logN, zstride, sa, r);
}

// logN in [6, 12], padded layout; input in bit-reversed order like fft_stages
template <bool INVERSE, int TB = FFT_TB>
__device__ __forceinline__ void fft_stages_r16(float2* z, const float2* tw, int N, int logN, int zstride) {
    fft_pass<INVERSE, TB, 4>(z, tw, N, logN, zstride, 1);
    fft_pass_r<INVERSE, TB, 4>(z, tw, N, logN, zstride, 5, logN - 4 < 4 ? logN - 4 : 4);
    if (logN > 8) fft_pass_r<INVERSE, TB, 4>(z, tw, N, logN, zstride, 9, logN - 8);
}

// either routine behind one call: ps = 4 -> register passes, ps = FFT_NOPAD -> one stage per LDS round trip (the layout follows ps)
template <bool INVERSE, int TB = FFT_TB>
__device__ __forceinline__ void fft_stages_any(float2* z, const float2* tw, int N, int logN, int zstride, int ps) {
    if (ps == 4) fft_stages_r16<INVERSE, TB>(z, tw, N, logN, zstride);
    else fft_stages<INVERSE, TB>(z, tw, N, logN, zstride, ps);
}

__device__ __forceinline__ int bitrev(int n, int logN) { return (int)(__brev((unsigned)n) >> (32 - logN)); }

static inline int ilog2_exact(int n) {
    int l = 0;
    while ((1 << l) < n) ++l;
    return ((1 << l) == n) ? l : -1;
}
