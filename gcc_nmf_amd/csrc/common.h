// Shared definitions for the gfx950 (MI355X / CDNA4) GCC-NMF kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define GCCNMF_OK 0
#define GCCNMF_ERR_ARG 1        // bad argument (null pointer, unsupported size, pitch too small)
#define GCCNMF_ERR_LAUNCH 2     // HIP launch failed (hipGetLastError != hipSuccess)
#define GCCNMF_ERR_UNSUPPORTED 3

static inline int gccnmf_ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int gccnmf_round_up(int a, int b) { return gccnmf_ceil_div(a, b) * b; }

#define GCCNMF_CHECK_LAUNCH()                                   \
    do {                                                        \
        hipError_t e__ = hipGetLastError();                     \
        if (e__ != hipSuccess) return GCCNMF_ERR_LAUNCH;        \
    } while (0)

// Storage geometry shared by every kernel (host mirrors it through gccnmf_pitches()):
//   Fp : rows of every [f][.] matrix          = round_up(F, 16)    (zero rows beyond F)
//   Kp : dictionary pitch / atom rows of H    = round_up(K, 64)    (zero atoms beyond K)
//   Np : column pitch of V, R, H (N = 2T)     = round_up(N, 64)    (zero columns beyond N)
//   Tp : column pitch of per-channel [.][t]   = round_up(T, 64)
struct GccNmfPitches {
    int Fp, Kp, Np, Tp;
};

static inline GccNmfPitches gccnmf_make_pitches(int F, int T, int K) {
    GccNmfPitches p;
    p.Fp = gccnmf_round_up(F, 16);
    p.Kp = gccnmf_round_up(K, 64);
    p.Np = gccnmf_round_up(2 * T, 64);
    p.Tp = gccnmf_round_up(T, 64);
    return p;
}
