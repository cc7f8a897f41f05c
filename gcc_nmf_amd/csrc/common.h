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

// ---- tuning state (gccnmf_set_tuning) ---------------------------------------------------------------------------------------
// The knobs are process-global ATOMICS; a library call works on a SNAPSHOT: every extern "C" entry point opens a GccNmfCall scope, which
// copies the atomics into the calling thread's `gccnmf_tune` once (a nested entry point keeps the outer call's copy), and everything
// below reads that copy.  A knob flipped by another thread while a call is running -- even inside an iteration loop that enqueues
// hundreds of launches -- cannot change the launches of that call; it takes effect at the next call.  Keys marked X exist in experiment
// builds only (make EXPERIMENTS=1): the product build rejects them and carries none of the code they select.
//   key  name           default  range   what
//    2   tile_policy       0     0..2    0 = tile by launch size, 1 = always the 512 x 64 throughput tile, 2 = always the 128 x 64 small-batch tile
//    3   dma               1     0..1    1 = throughput tiles stage operands by LDS-DMA (gemm_dma.h), 0 = through registers (gemm_mfma.h)
//    7   exact_div         1     0..1    1 = V / (W.H) is the IEEE quotient, 0 = v_rcp_f32 + one Newton step
//    8   shared_groups     3     1..4    file groups of a shared-dictionary shard that cannot fill the chip, on library-owned streams
//    9   tail_split        1     0..3    throughput-tile launch forms: 1 = by the launcher's rules, 0 = full tiles only, 2 = all narrow halves, 3 = all half-height (tests)
//   10   direct            1     0..1    1 = launches that cannot fill the chip take the direct-to-register kernels (direct.hip)
//   12   direct_batch      4     1..8    largest batch on the direct path
//   16   fused_k12         1     0..2    K <= 128: K1 + K2 as one launch of column tiles (0 never, 1 by cost model, 2 always)
//   17   fused_k34         1     0..2    K <= 128: K3 + K4a as one launch of bin slabs
//   21   chain             1   0,1,2,4,8 batch scale, K > 256: 1 = the whole gccnmf_klnmf call as ONE chained launch (gemm_dma.h: GemmSync) where it wins (>= 3 files per
//                                        XCD, balanced whole-file lists, no other file group beside it); 0 = never; forced forms: 2 = K1 | K2, 4 = the four GEMMs of
//                                        an iteration, 8 = every iteration of the call
//   23   chain_lists       1     0..3    lists of a chained launch: 1 = by rule -- whole files per XCD (hand-over through that XCD's L2) where they balance, else the
//                                        file-major tile list in equal eighths with agent-scope hand-over (a file may straddle XCDs); 0 = the plain launch's lists
//                                        (batch a multiple of 8); 2 = always spread; 3 = always whole files
//   24   chain_chunk     2048   1..65536 iterations per chained launch: a call of more iterations runs as several chained launches (the counters keep counting)
//    1 X ablate            0             timing ablations of the register-staged kernel (results invalid)
//    4 X ring              1     0..1    0 = small-batch tiles on the register-staged kernel instead of the LDS-DMA ring kernel
//    5 X wh_splits         3     1..4    parts of the single-file split-K W.H (the round-3 latency path, superseded by direct.hip)
//    6 X rht_splits        4     1..4    parts of the single-file split-K R.H^T
//   11 X direct_tile       0     0..8    a fixed tile for every direct launch
//   13 X direct_depth      0     0,2..4  register sets of the direct kernels' operand pipeline
//   14 X short_updh        1     0..1    0 = H updates of at most 128 atoms on the register-staged 128 x 256 tile
//   15 X fft_r16           1     0..1    0 = one radix-2 stage per LDS round trip in the offline STFT / iSTFT (same bits)
//   18 X persistent        0     0..1    1 = launches of more than 512 tiles as 512 resident workgroups pulling tiles by ticket
//   19 X prefetch          1     0..1    1 = a resident workgroup requests its next tile's first k-tile before the current epilogue
//   20 X wide_update_w     1     0..1    0 = the one-pass W update of short dictionaries at batch scale on 16 atoms per workgroup (round 4)
//   25 X chain_fault       0     0..1    1 = fault injection: a consumer of a chained launch that has to wait gives up at once (error flag -> NaN factors -> the engine's fallback)
//   22 X chain_solo        0     0..1    1 = chained launches reserve enough LDS that only ONE workgroup fits a CU (in-order dispatch => no deadlock: tested)
#define GCCNMF_TUNE_KEYS 26
struct GccNmfTune {
    int v[GCCNMF_TUNE_KEYS];
};
extern thread_local GccNmfTune gccnmf_tune;
struct GccNmfCall {
    GccNmfCall();
    ~GccNmfCall();
};
#define GCCNMF_ENTER() GccNmfCall gccnmf_call_scope_
#define gccnmf_tune_ablate (gccnmf_tune.v[1])
#define gccnmf_tune_tile_policy (gccnmf_tune.v[2])
#define gccnmf_tune_dma (gccnmf_tune.v[3])
#define gccnmf_tune_ring (gccnmf_tune.v[4])
#define gccnmf_tune_wh_splits (gccnmf_tune.v[5])
#define gccnmf_tune_rht_splits (gccnmf_tune.v[6])
#define gccnmf_tune_exact_div (gccnmf_tune.v[7])
#define gccnmf_tune_shared_groups (gccnmf_tune.v[8])
#define gccnmf_tune_tail_split (gccnmf_tune.v[9])
#define gccnmf_tune_direct (gccnmf_tune.v[10])
#define gccnmf_tune_direct_tile (gccnmf_tune.v[11])
#define gccnmf_tune_direct_batch (gccnmf_tune.v[12])
#define gccnmf_tune_direct_depth (gccnmf_tune.v[13])
#define gccnmf_tune_short_updh (gccnmf_tune.v[14])
#define gccnmf_tune_fft_r16 (gccnmf_tune.v[15])
#define gccnmf_tune_fused_k12 (gccnmf_tune.v[16])
#define gccnmf_tune_fused_k34 (gccnmf_tune.v[17])
#define gccnmf_tune_persistent (gccnmf_tune.v[18])
#define gccnmf_tune_prefetch (gccnmf_tune.v[19])
#define gccnmf_tune_wide_update_w (gccnmf_tune.v[20])
#define gccnmf_tune_chain (gccnmf_tune.v[21])
#define gccnmf_tune_chain_solo (gccnmf_tune.v[22])
#define gccnmf_tune_chain_rag (gccnmf_tune.v[23])
#define gccnmf_tune_chain_chunk (gccnmf_tune.v[24])
#define gccnmf_tune_chain_fault (gccnmf_tune.v[25])
