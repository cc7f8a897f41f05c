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
template <bool INVERSE, int TB = FFT_TB>
__device__ __forceinline__ void fft_stages(float2* z, const float2* tw, int N, int logN, int zstride) {
    const int half_n = N >> 1;
    for (int s = 1; s <= logN; ++s) {
        const int half = 1 << (s - 1);
        const int tw_step = N >> s;
        for (int idx = threadIdx.x; idx < TB * half_n; idx += FFT_NT) {
            const int tb = idx / half_n, bf = idx - tb * half_n;
            const int grp = bf >> (s - 1), pos = bf & (half - 1);
            const int i0 = (grp << s) + pos, i1 = i0 + half;
            float2 w = tw[pos * tw_step];
            if (INVERSE) w.y = -w.y;
            float2* zz = z + tb * zstride;
            const float2 u = zz[i0];
            const float2 t = cmul(w, zz[i1]);
            zz[i0] = make_float2(u.x + t.x, u.y + t.y);
            zz[i1] = make_float2(u.x - t.x, u.y - t.y);
        }
        __syncthreads();
    }
}

__device__ __forceinline__ int bitrev(int n, int logN) { return (int)(__brev((unsigned)n) >> (32 - logN)); }

static inline int ilog2_exact(int n) {
    int l = 0;
    while ((1 << l) < n) ++l;
    return ((1 << l) == n) ? l : -1;
}
