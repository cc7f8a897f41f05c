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
template <bool INVERSE, int TB, int UN>
__device__ __forceinline__ void fft_stages_un(float2* z, const float2* tw, int N, int logN, int zstride) {
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
                    i0[j] = tb * zstride + (grp << s) + pos;
                    w[j] = tw[pos * tw_step];
                    u[j] = z[i0[j]];
                    v[j] = z[i0[j] + half];
                }
            }
#pragma unroll
            for (int j = 0; j < UN; ++j) {
                const int idx = base + j * FFT_NT;
                if (idx < total) {
                    float2 wj = w[j];
                    if (INVERSE) wj.y = -wj.y;
                    const float2 t = cmul(wj, v[j]);
                    z[i0[j]] = make_float2(u[j].x + t.x, u[j].y + t.y);
                    z[i0[j] + half] = make_float2(u[j].x - t.x, u[j].y - t.y);
                }
            }
        }
        __syncthreads();
    }
}

// (eight at a time when a thread has at least four butterflies per stage -- the offline kernels: 16; one at a time otherwise -- the
// streaming processor's single frame: one butterfly per thread, where the eight predicated slots cost 20 us per block)
template <bool INVERSE, int TB = FFT_TB>
__device__ __forceinline__ void fft_stages(float2* z, const float2* tw, int N, int logN, int zstride) {
    if (TB * (N >> 1) >= 4 * FFT_NT) fft_stages_un<INVERSE, TB, 8>(z, tw, N, logN, zstride);
    else fft_stages_un<INVERSE, TB, 1>(z, tw, N, logN, zstride);
}

__device__ __forceinline__ int bitrev(int n, int logN) { return (int)(__brev((unsigned)n) >> (32 - logN)); }

static inline int ilog2_exact(int n) {
    int l = 0;
    while ((1 << l) < n) ++l;
    return ((1 << l) == n) ? l : -1;
}
