// KL-NMF multiplicative updates on gfx950 -- the >98 %-of-FLOPs part of the GCC-NMF path.
// Reference: performKLNMF, gccNMF/gccNMFFunctions.py:69-83.
//
// One iteration (reference lines in brackets) is five launches that cover the whole batch:
//   K1  R  = V / (W . (s*H))                     MFMA GEMM + divide epilogue            [:76 inner]
//   K2  H  = (s*H) * (W^T . R) / (colsum W + alpha + eps)   MFMA GEMM + update epilogue [:76]
//   K3  R  = V / (W . H)                         (new H)                                [:77 inner]
//   K4a U  = R . H^T ,  rowsumH = sum_n H        MFMA GEMM, row sums from the staged B  [:77]
//   K4b W  = normalise(W * U / rowsumH); s = atom norms; colsumW = sum_f W             [:77,:79-80]
// The compensating rescale H *= norms (:81) is NOT a separate pass over H: it is the vector s,
// applied while K1 stages its B operand and inside K2's epilogue (which rewrites H anyway).
// gccnmf_klnmf materialises it once after the last iteration.
#include <mutex>
#include "gemm_ring.h"
#include "direct.h"
#include "update_w.h"
#ifdef GCCNMF_EXPERIMENTS
int gccnmf_launch_gemm_stream(GemmArgs a, hipStream_t stream);       // the LDS-free throughput tile (direct.hip)
#endif
#include "../../include/gccnmf_hip.h"

#include <atomic>
#include <algorithm>
#include <vector>
#define GCCNMF_SHARED_STREAMS 4
#define GCCNMF_CHAIN_MAX_ITERATIONS gccnmf_tune_chain_chunk   // iterations per chained launch (tuning key 24, default 2048: the grid stays far below 2^30 workgroups)
#define GCCNMF_RAGGED_MAX_BATCH 248   // files of a ragged batch: 8 lists of at most GEMM_RAGGED_LMAX
#define GCCNMF_SPLITS 4           // parts of the round-3 single-file split-K (experiment builds); the workspace layout keeps room for them
#define GCCNMF_DIRECT_MAX_BATCH 8    // workspaces of at most this many files carry the transposed copies of the direct path (key 12 selects up to here; from 8 files on
                                     // the ring / throughput kernels win anyway: 8 files 41.5 against 41.3 ms, 12 files 61.9 against 59.3)
// The tuning table (common.h documents the keys): default, lowest and highest value, experiment-only.
struct GccNmfKnob {
    int def, lo, hi, experiment;
};
static const GccNmfKnob gccnmf_knobs[GCCNMF_TUNE_KEYS] = {
    {0, 0, 0, 1},                       //  0 (unused)
    {0, 0, 1 << 30, 1},                 //  1 ablate
    {0, 0, 2, 0},                       //  2 tile_policy
    {1, 0, 1, 0},                       //  3 dma
    {1, 0, 1, 1},                       //  4 ring
    {3, 1, 4, 1},                       //  5 wh_splits
    {4, 1, 4, 1},                       //  6 rht_splits
    {1, 0, 1, 0},                       //  7 exact_div: the IEEE quotient since round 5 (measured cost on K1 / K3: 0.0 %, profiles/r05b_kbench_exact_div.txt)
    {3, 1, GCCNMF_SHARED_STREAMS, 0},   //  8 shared_groups
    {1, 0, 3, 0},                       //  9 tail_split (3 = half-height tiles everywhere: tests / measurements)
    {1, 0, 1, 0},                       // 10 direct
    {0, 0, 8, 1},                       // 11 direct_tile
    {4, 1, GCCNMF_DIRECT_MAX_BATCH, 0}, // 12 direct_batch (measured, K = 1024: 4 files 21.8 ms against 25.8 on the ring kernel, 8 files 41.5 / 41.3)
    {0, 0, 4, 1},                       // 13 direct_depth (0, 2..4)
    {1, 0, 1, 1},                       // 14 short_updh
    {1, 0, 1, 1},                       // 15 fft_r16
    {1, 0, 2, 0},                       // 16 fused_k12
    {1, 0, 2, 0},                       // 17 fused_k34
    {0, 0, 1, 1},                       // 18 persistent
    {1, 0, 1, 1},                       // 19 prefetch
    {1, 0, 1, 1},                       // 20 wide_update_w
    {1, 0, 8, 0},                       // 21 chain: 1 = the whole call as one chained launch where the rule in chain_stages says so; 0 off; forced forms (tests, A/B):
                                        //    2 = K1 | K2, 4 = K1 | K2 | K3 | K4 per iteration, 8 = every iteration of the call
    {0, 0, 1, 1},                       // 22 chain_solo: chained launches with one workgroup per CU (the freedom-from-deadlock test)
    {1, 0, 3, 0},                       // 23 chain_lists: 1 by rule (whole files per XCD where they balance, else spread) | 0 the plain launch's lists (batch a multiple of 8) |
                                        //    2 always spread: file-major equal eighths, agent-scope hand-over | 3 always whole files
    {2048, 1, 65536, 0},                // 24 chain_chunk: iterations per chained launch (a call of more iterations is several launches; tests use small values)
    {0, 0, 1, 1},                       // 25 chain_fault: fault injection -- consumers of chained launches give up waiting at once (the failure path's test)
};
static std::atomic<int> gccnmf_knob_value[GCCNMF_TUNE_KEYS];
static std::atomic<int> gccnmf_knobs_ready{0};
static void gccnmf_knobs_init() {
    static std::mutex mu;
    if (gccnmf_knobs_ready.load(std::memory_order_acquire)) return;
    std::lock_guard<std::mutex> lock(mu);
    if (gccnmf_knobs_ready.load(std::memory_order_relaxed)) return;
    for (int k = 0; k < GCCNMF_TUNE_KEYS; ++k) gccnmf_knob_value[k].store(gccnmf_knobs[k].def, std::memory_order_relaxed);
    gccnmf_knobs_ready.store(1, std::memory_order_release);
}
static GccNmfTune gccnmf_tune_load() {
    gccnmf_knobs_init();
    GccNmfTune t;
    for (int k = 0; k < GCCNMF_TUNE_KEYS; ++k) t.v[k] = gccnmf_knob_value[k].load(std::memory_order_relaxed);
    return t;
}
thread_local GccNmfTune gccnmf_tune = gccnmf_tune_load();
static thread_local int gccnmf_call_depth = 0;
GccNmfCall::GccNmfCall() {
    if (gccnmf_call_depth++ == 0) gccnmf_tune = gccnmf_tune_load();      // one snapshot per outermost library call
}
GccNmfCall::~GccNmfCall() { --gccnmf_call_depth; }
long long* gccnmf_trace_buf = nullptr;
int gccnmf_trace_blocks = 0;

extern "C" {
int gccnmf_version(void) { return 106; }   // round 5: work lists with narrow items in the throughput tile (gccnmf_debug_gemm_plan), IEEE division by default,
                                           // tuning keys as atomics snapshotted per call, product / experiment build flavours (42 product entry points)

int gccnmf_set_tuning(int key, int value) {
    GCCNMF_ENTER();
    gccnmf_knobs_init();
    if (key < 1 || key >= GCCNMF_TUNE_KEYS) return GCCNMF_ERR_ARG;
    const GccNmfKnob& k = gccnmf_knobs[key];
#ifndef GCCNMF_EXPERIMENTS
    if (k.experiment) return GCCNMF_ERR_ARG;                 // the product build carries none of the code these select
#endif
    if (value < k.lo || value > k.hi || (key == 13 && value == 1) || (key == 21 && value != 0 && value != 1 && value != 2 && value != 4 && value != 8)) return GCCNMF_ERR_ARG;
    gccnmf_knob_value[key].store(value, std::memory_order_relaxed);      // takes effect at the next library call (GccNmfCall)
    return GCCNMF_OK;
}

}  // extern "C"

#ifdef GCCNMF_EXPERIMENTS
// Ticket blocks of the persistent throughput-tile launches (gemm_dma.h): 16 counters per (device, stream), zero between launches (the last
// workgroup of a launch resets them).  Launches on one stream are serialised, so a stream's launches share a block; every stream has its own.
// nullptr (pool exhausted, allocation refused -- e.g. inside a stream capture) = the launch falls back to the classic grid.
unsigned* gccnmf_ticket_block(hipStream_t stream) {
    constexpr int MAX_DEV = 16, PER_DEV = 128;
    struct Entry { hipStream_t s; };
    static std::mutex mu;
    static unsigned* pool[MAX_DEV] = {};
    static Entry entries[MAX_DEV][PER_DEV];
    static int used[MAX_DEV] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    for (int i = 0; i < used[dev]; ++i)
        if (entries[dev][i].s == stream) return pool[dev] + 16 * i;
    if (used[dev] == PER_DEV) return nullptr;
    if (!pool[dev]) {
        static const unsigned zeros[16 * PER_DEV] = {};
        unsigned* p = nullptr;
        if (hipMalloc((void**)&p, sizeof(zeros)) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
        if (hipMemcpy(p, zeros, sizeof(zeros), hipMemcpyHostToDevice) != hipSuccess) {      // blocking: the counters are zero before any launch reads them
            (void)hipGetLastError();
            (void)hipFree(p);
            return nullptr;
        }
        pool[dev] = p;
    }
    entries[dev][used[dev]].s = stream;
    return pool[dev] + 16 * used[dev]++;
}

#endif

extern "C" {
#ifdef GCCNMF_EXPERIMENTS
int gccnmf_debug_set_trace(long long* buf, int blocks) {
    GCCNMF_ENTER();
    gccnmf_trace_buf = buf;
    gccnmf_trace_blocks = buf ? blocks : 0;
    return GCCNMF_OK;
}

#endif

int gccnmf_pitches(int F, int T, int K, int* Fp, int* Kp, int* Np, int* Tp) {
    GCCNMF_ENTER();
    if (F < 2 || T < 1 || K < 1 || !Fp || !Kp || !Np || !Tp) return GCCNMF_ERR_ARG;
    GccNmfPitches p = gccnmf_make_pitches(F, T, K);
    *Fp = p.Fp;
    *Kp = p.Kp;
    *Np = p.Np;
    *Tp = p.Tp;
    return GCCNMF_OK;
}
}

#ifdef GCCNMF_EXPERIMENTS
// MFMA-only probe: what the f32 matrix pipe sustains on this box (clock included), nothing but 8 independent
// accumulator chains per wave, two waves per SIMD.  grid = 512 blocks of 256 threads.
__global__ __launch_bounds__(256, 2) void mfma_peak_kernel(float* out, int iters) {
    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = 1.0f + threadIdx.x * 1e-3f, b = 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 123.456f) out[0] = s;
}
#endif

// ------------------------------------------------------------------------------------------
// small K-vector kernels
// ------------------------------------------------------------------------------------------
// colsumW[k] = sum_f W[f][k]; hscale[k] = 1.   grid = batch * Kp/16, 256 threads = 16 atoms x 16 row phases (one file: 64 workgroups
// of 33-row chains instead of 16 of 129: 40 -> ~8 us).
__global__ __launch_bounds__(256) void nmf_prepare_kernel(const float* __restrict__ W, float* __restrict__ colsumW,
                                                          float* __restrict__ hscale, int F, int Fp, int Kp) {
    __shared__ float red[4][16];
    const int chunks = Kp / 16;
    const int b = blockIdx.x / chunks, ch = blockIdx.x - b * chunks;
    const int c = threadIdx.x & 15, q = threadIdx.x >> 4, wave = threadIdx.x >> 6;
    const int k = ch * 16 + c;
    const float* Wb = W + (long)b * Fp * Kp;
    float s = 0.f;
#pragma unroll 4
    for (int f = q; f < F; f += 16) s += Wb[(long)f * Kp + k];
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    if ((threadIdx.x & 63) < 16) red[wave][c] = s;
    __syncthreads();
    if (threadIdx.x < 16) {
        colsumW[(long)b * Kp + k] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
        hscale[(long)b * Kp + k] = 1.f;
    }
}

// W update + unit-L2 atom normalisation (gccNMFFunctions.py:77,79-80):
//   Wt = W * (U / rowsumH[k]);  norm[k] = sqrt(sum_f Wt^2);  W = Wt / norm;  hscale = norm;  colsumW = sum_f W
// grid = batch * Kp/AT, 256 threads = AT atoms x 256/AT row phases; W and U are read twice (L2-resident) instead of
// keeping F*AT/256 values per thread in registers.  AT = 64 for big batches (256-byte row segments); AT = 16 when the
// launch would otherwise be a handful of workgroups (single file: 16 -> 64 workgroups, 4x shorter row loops).
// nsplit > 1 (single-file split-K, below): U and rowsumH arrive as nsplit partial sums, sSplitU / sSplitR floats apart,
// and are added in ascending split order wherever they are read.
template <int AT>
__global__ __launch_bounds__(256) void nmf_update_w_kernel(float* __restrict__ W, const float* __restrict__ U,
                                                           const float* __restrict__ rowsumH, float* __restrict__ colsumW,
                                                           float* __restrict__ hscale, int F, int Fp, int K, int Kp,
                                                           long sW, long sU, long sVec, long sRowsum, int nsplit, long sSplitU,
                                                           long sSplitR) {
    constexpr int PH = 256 / AT;
    __shared__ float red[256];
    __shared__ float s_norm[AT];
    const int chunks = Kp / AT;
    const int b = blockIdx.x / chunks, ch = blockIdx.x - b * chunks;
    const int c = threadIdx.x % AT, q = threadIdx.x / AT;
    const int k = ch * AT + c;
    const bool valid = k < K;          // padded atoms stay exactly zero
    float* Wb = W + b * sW;
    const float* Ub = U + b * sU;
    float rs = 1.f;
    if (valid) {
        rs = rowsumH[b * sRowsum + k];
        for (int sp = 1; sp < nsplit; ++sp) rs += rowsumH[b * sRowsum + sp * sSplitR + k];
    }
    auto u_at = [&](long i) {
        float u = Ub[i];
        for (int sp = 1; sp < nsplit; ++sp) u += Ub[sp * sSplitU + i];
        return u;
    };
    float ss = 0.f;
    if (valid)
        for (int f = q; f < F; f += PH) {
            const float wt = Wb[(long)f * Kp + k] * (u_at((long)f * Kp + k) / rs);
            ss = fmaf(wt, wt, ss);
        }
    red[threadIdx.x] = ss;
    __syncthreads();
    if (q == 0) {
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < PH; ++j) t += red[c + AT * j];
        s_norm[c] = sqrtf(t);
    }
    __syncthreads();
    const float norm = s_norm[c];
    float cs = 0.f;
    if (valid)
        for (int f = q; f < F; f += PH) {
            const long i = (long)f * Kp + k;
            const float wn = (Wb[i] * (u_at(i) / rs)) / norm;
            Wb[i] = wn;
            cs += wn;
        }
    red[threadIdx.x] = cs;
    __syncthreads();
    if (q == 0 && valid) {
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < PH; ++j) t += red[c + AT * j];
        colsumW[b * sVec + k] = t;
        hscale[b * sVec + k] = norm;
    }
}

// The same update in ONE pass for launches of a few files (the two-pass kernel above re-reads W and every U partial, 64 dependent
// round trips per thread: 50 us for one file at K = 1024): 16 atoms per workgroup as float4 columns x 64 row phases, the
// R <= 17 rows of a thread stay in registers between the norm and the store, all loads of a thread are independent.
// Column reductions: xor-shuffles over the 16 row phases of a wave, then 4 waves through LDS (fixed order: deterministic).
template <int AT, int NSPLIT>
__global__ __launch_bounds__(256) void nmf_update_w_onepass_kernel(float* __restrict__ W, const float* __restrict__ U,
                                                                   const float* __restrict__ rowsumH, float* __restrict__ colsumW,
                                                                   float* __restrict__ hscale, int F, int K, int Kp, long sW, long sU,
                                                                   long sVec, long sRowsum, long sSplitU, long sSplitR, float* __restrict__ Wt,
                                                                   long sWt, int ldwt) {
    __shared__ __attribute__((aligned(16))) float smem[5 * AT];
    const int chunks = Kp / AT;
    const int b = blockIdx.x / chunks, ch = blockIdx.x - b * chunks;
    nmf_update_w_onepass_item<AT, NSPLIT>(W, U, rowsumH, colsumW, hscale, F, K, Kp, sW, sU, sVec, sRowsum, sSplitU, sSplitR, Wt, sWt, ldwt, b, ch, smem, (int)threadIdx.x);
}

static int launch_update_w(float* W, const float* U, const float* rowsumH, float* colsumW, float* hscale, int F, int Fp, int K,
                           int Kp, long sW, long sU, long sVec, long sRowsum, int batch, hipStream_t s, int nsplit = 1,
                           long sSplitU = 0, long sSplitR = 0, float* Wt = nullptr, long sWt = 0, int ldwt = 0) {
    if ((long)batch * (Kp / 64) < 256 && gccnmf_tune_ring && F <= 64 * 9 && (nsplit == 1 || nsplit == 2 || nsplit == 4)) {
        // 16 atoms per workgroup (64-byte row segments); 8 (twice the workgroups, 32-byte segments) measured slower: 13.0 vs 11.4 us for one
        // file at K = 1024 -- kept selectable for experiments (tuning key 1 = 64)
#define GCCNMF_ONEPASS(AT_, NS_) hipLaunchKernelGGL((nmf_update_w_onepass_kernel<AT_, NS_>), dim3(batch * (Kp / AT_)), dim3(256), 0, s, W, U, rowsumH, \
                                                    colsumW, hscale, F, K, Kp, sW, sU, sVec, sRowsum, sSplitU, sSplitR, Wt, sWt, ldwt)
        const bool narrow = gccnmf_tune_ablate == 64;
        // short dictionaries at batch scale (64 files, K = 128: the launch between the two fused GEMM launches): 32 atoms per workgroup =
        // whole 128-byte lines per row and workgroup -- with 16 atoms every line of W and U is fetched by two workgroups (17 us for 51 MB).
        // Only for K <= 128, where the launch forms follow the batch size anyway (a file's bits are batch-independent for K > 128).
        const bool wide = !narrow && nsplit == 1 && !Wt && K <= 128 && (long)batch * (Kp / 32) >= 256 && gccnmf_tune_wide_update_w;
        if (wide) {
            GCCNMF_ONEPASS(32, 1);
        } else if (narrow) {
            if (nsplit == 1) GCCNMF_ONEPASS(8, 1);
            else if (nsplit == 2) GCCNMF_ONEPASS(8, 2);
            else GCCNMF_ONEPASS(8, 4);
        } else {
            if (nsplit == 1) GCCNMF_ONEPASS(16, 1);
            else if (nsplit == 2) GCCNMF_ONEPASS(16, 2);
            else GCCNMF_ONEPASS(16, 4);
        }
#undef GCCNMF_ONEPASS
        GCCNMF_CHECK_LAUNCH();
        return GCCNMF_OK;
    }
    if ((long)batch * (Kp / 64) >= 256) {
        hipLaunchKernelGGL(nmf_update_w_kernel<64>, dim3(batch * (Kp / 64)), dim3(256), 0, s, W, U, rowsumH, colsumW, hscale, F, Fp, K,
                           Kp, sW, sU, sVec, sRowsum, nsplit, sSplitU, sSplitR);
    } else {
        hipLaunchKernelGGL(nmf_update_w_kernel<16>, dim3(batch * (Kp / 16)), dim3(256), 0, s, W, U, rowsumH, colsumW, hscale, F, Fp, K,
                           Kp, sW, sU, sVec, sRowsum, nsplit, sSplitU, sSplitR);
    }
    GCCNMF_CHECK_LAUNCH();
    if (Wt) return gccnmf_transpose_launch(W, sW, Kp, Wt, sWt, ldwt, F, Kp, batch, s);      // (the one-pass kernel writes it itself)
    return GCCNMF_OK;
}

// H[k][:] *= hscale[k].  grid = batch * K rows, 256 threads.  (sScale = 0: one shared scale vector; file b starts sH floats after
// file b-1, rows are ld floats apart, Np of them are touched)
__global__ __launch_bounds__(256) void nmf_scale_h_kernel(float* __restrict__ H, const float* __restrict__ hscale, long sScale,
                                                          int K, long sH, int ld, int Np) {
    const int b = blockIdx.x / K, k = blockIdx.x - b * K;
    const float s = hscale[b * sScale + k];
    float4* row = (float4*)(H + b * sH + (long)k * ld);
    for (int i = threadIdx.x; i < Np / 4; i += 256) {
        float4 v = row[i];
        v.x *= s;
        v.y *= s;
        v.z *= s;
        v.w *= s;
        row[i] = v;
    }
}

__global__ __launch_bounds__(256) void nmf_fill_kernel(float* __restrict__ p, float v, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = v;
}

// out[i] = sum_b in[b*stride + i], files added in ascending order (deterministic); accumulate: out[i] += that sum (a rank's
// second, third ... shard: the shards' sums are added in shard order).
__global__ __launch_bounds__(256) void nmf_reduce_files_kernel(const float* __restrict__ in, long stride, int batch, long n,
                                                               float* __restrict__ out, int accumulate) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int b = 0; b < batch; ++b) s += in[b * stride + i];
    out[i] = accumulate ? out[i] + s : s;
}

// ------------------------------------------------------------------------------------------
// GEMM dispatch: tall <4,1> tile for >128 output rows, wide <1,4> otherwise
// ------------------------------------------------------------------------------------------
// Tile choice.  Throughput tile: <4,1> x TM=4 = 512 x 64 per workgroup (128 x 64 per wave).  When the whole launch has
// fewer tall tiles than the chip has CUs (a single file: 16-40 of them), a workgroup's duration IS the launch duration,
// so the same 4 waves take a 128 x 64 tile instead (TM = 1: 32 x 64 per wave, 4x shorter, 4x more workgroups).
// <1,4> serves outputs of <= 128 rows at any batch.
static bool small_batch_tile(const GemmArgs& a) {
    if (a.M <= 128 || gccnmf_tune_tile_policy == 1) return false;
    if (gccnmf_tune_tile_policy == 2) return true;
    const long tall_tiles = (long)a.batch * gccnmf_ceil_div(a.M, 512) * gccnmf_ceil_div(a.N, 64);
    return tall_tiles < 256;
}

template <bool A_KC, bool B_KC, int EPI>
static int dispatch_gemm(const GemmArgs& a, bool tail, hipStream_t s) {
    const bool tall = a.M > 128;
    // The H update of a SMALL dictionary (K <= 128 atoms = output rows: the reference driver's default, runGCCNMF.py:41) used to ride the
    // register-staged wide tile (128 x 256 per workgroup, one k-tile prefetched): 154 us per launch for 64 files at K = 128 = 0.43 of the
    // matrix peak.  The LDS-DMA ring kernel's 128 x 64 tiles (six stages in flight, image-layout epilogue) take it instead (key 14).
    const bool short_updh = !tall && EPI == EPI_UPDH && gccnmf_tune_short_updh && gccnmf_tune_tile_policy != 1 && a.N >= 256;
    if ((small_batch_tile(a) || short_updh) && gccnmf_tune_ring && gccnmf_ring_supports(a.Kd)) {   // (longer reductions: the register-staged tile below)
        if (A_KC) return tail ? gccnmf_launch_gemm_ring<A_KC, B_KC, EPI, A_KC>(a, s) : gccnmf_launch_gemm_ring<A_KC, B_KC, EPI, false>(a, s);
        if (tail) return GCCNMF_ERR_ARG;
        return gccnmf_launch_gemm_ring<A_KC, B_KC, EPI, false>(a, s);
    }
    if (small_batch_tile(a)) {
        if (A_KC) return tail ? gccnmf_launch_gemm<4, 1, A_KC, B_KC, EPI, A_KC, 1>(a, s) : gccnmf_launch_gemm<4, 1, A_KC, B_KC, EPI, false, 1>(a, s);
        if (tail) return GCCNMF_ERR_ARG;
        return gccnmf_launch_gemm<4, 1, A_KC, B_KC, EPI, false, 1>(a, s);
    }
    if (tall && gccnmf_tune_dma) {
        if (A_KC) return tail ? gccnmf_launch_gemm_dma<A_KC, B_KC, EPI, A_KC>(a, s) : gccnmf_launch_gemm_dma<A_KC, B_KC, EPI, false>(a, s);
        if (tail) return GCCNMF_ERR_ARG;
        return gccnmf_launch_gemm_dma<A_KC, B_KC, EPI, false>(a, s);
    }
    if (A_KC) {
        if (tall) return tail ? gccnmf_launch_gemm<4, 1, A_KC, B_KC, EPI, A_KC>(a, s) : gccnmf_launch_gemm<4, 1, A_KC, B_KC, EPI, false>(a, s);
        return tail ? gccnmf_launch_gemm<1, 4, A_KC, B_KC, EPI, A_KC>(a, s) : gccnmf_launch_gemm<1, 4, A_KC, B_KC, EPI, false>(a, s);
    }
    if (tail) return GCCNMF_ERR_ARG;
    return tall ? gccnmf_launch_gemm<4, 1, A_KC, B_KC, EPI, false>(a, s) : gccnmf_launch_gemm<1, 4, A_KC, B_KC, EPI, false>(a, s);
}

struct NmfGeom {
    int F, N, K, Fp, Kp, Np;
    int ld;         // row pitch (floats) of V, R and H: Np for back-to-back files, the big matrix's pitch for column blocks
    int Fm;         // rows computed on the matrix cores
    bool tail;      // F % 128 == 1: the last bin rides on the VALU
    long sV, sW, sH, sU;
};

static NmfGeom make_geom(int F, int N, int K) {
    NmfGeom g;
    GccNmfPitches p = gccnmf_make_pitches(F, 1, K);
    g.F = F; g.N = N; g.K = K;
    g.Fp = p.Fp; g.Kp = p.Kp; g.Np = gccnmf_round_up(N, 64);
    g.ld = g.Np;
    g.tail = (F % 128) == 1 && F > 1;
    g.Fm = g.tail ? F - 1 : F;
    g.sV = (long)g.Fp * g.Np;
    g.sW = (long)g.Fp * g.Kp;
    g.sH = (long)g.Kp * g.Np;
    g.sU = (long)g.Fp * g.Kp;
    return g;
}

// R = V / (W . (bscale*H))
static int launch_wh_div(const NmfGeom& g, const float* V, const float* W, long sW, const float* H, const float* hscale,
                         long sScale, float* R, int batch, int xcd, hipStream_t s) {
    GemmArgs a = {};
    a.A = W; a.sA = sW; a.lda = g.Kp; a.a_clamp = g.Fp - 1;
    a.B = H; a.sB = g.sH; a.ldb = g.ld; a.b_clamp = g.Np - 4;
    a.M = g.Fm; a.N = g.N; a.Kd = g.K;
    a.batch = batch; a.xcd_affine = xcd & 1; a.concurrent = (xcd >> 1) & 1;
    a.bscale = hscale; a.s_bscale = sScale;
    a.tail_row = g.F - 1;
    a.C = R; a.sC = g.sV; a.ldc = g.ld;
    a.E0 = V; a.sE0 = g.sV;
    return dispatch_gemm<true, false, EPI_DIV>(a, g.tail, s);
}

// H = (hscale*H) * (W^T . R) / (colsumW + alpha + eps)
static int launch_update_h(const NmfGeom& g, const float* W, long sW, const float* R, float* H, const float* hscale,
                           long sScale, const float* colsumW, long sVec, float alpha, float eps, int batch, int xcd,
                           hipStream_t s) {
    GemmArgs a = {};
    a.A = W; a.sA = sW; a.lda = g.Kp; a.a_clamp = g.Kp - 4;
    a.B = R; a.sB = g.sV; a.ldb = g.ld; a.b_clamp = g.Np - 4;
    a.M = g.K; a.N = g.N; a.Kd = g.F;
    if ((g.F % 16) == 1) {   // F = 16*n + 1: bin F-1 leaves the matrix cores and becomes a rank-1 term of the epilogue
        a.Kd = g.F - 1;
        a.ktailA = W + (long)(g.F - 1) * g.Kp; a.s_ktailA = sW;
        a.ktailB = R + (long)(g.F - 1) * g.ld; a.s_ktailB = g.sV;
    }
    a.batch = batch; a.xcd_affine = xcd & 1; a.concurrent = (xcd >> 1) & 1;
    a.C = H; a.sC = g.sH; a.ldc = g.ld;
    a.E1 = hscale; a.sE1 = sScale;
    a.E2 = colsumW; a.sE2 = sVec;
    a.alpha = alpha; a.eps = eps;
    return dispatch_gemm<false, false, EPI_UPDH>(a, false, s);
}

// U = R . H^T, rowsumH = sum_n H
static int launch_rht(const NmfGeom& g, const float* R, const float* H, float* U, float* rowsumH, int batch, int xcd,
                      hipStream_t s) {
    GemmArgs a = {};
    a.A = R; a.sA = g.sV; a.lda = g.ld; a.a_clamp = g.Fp - 1;
    a.B = H; a.sB = g.sH; a.ldb = g.ld; a.b_clamp = g.Kp - 1;
    a.M = g.Fm; a.N = g.K; a.Kd = g.N;
    a.batch = batch; a.xcd_affine = xcd & 1; a.concurrent = (xcd >> 1) & 1;
    a.tail_row = g.F - 1;
    a.rowsumB = rowsumH; a.s_rowsumB = g.Kp;
    a.C = U; a.sC = g.sU; a.ldc = g.Kp;
    return dispatch_gemm<true, true, EPI_STORE>(a, g.tail, s);
}

// W = normalise(W * (R.H^T) / rowsumH), colsumW, hscale -- K4a and K4b in one launch (tall tile, all F rows in one workgroup)
static bool can_fuse_w_update(const NmfGeom& g, int batch) {
    // the fused epilogue needs the tall tile; tiny launches prefer the small-batch tile and the two-launch form
    if (g.Fm <= 128 || g.Fm > 512 || gccnmf_tune_tile_policy == 2) return false;
    return gccnmf_tune_tile_policy == 1 || (long)batch * gccnmf_ceil_div(g.K, 64) >= 256;
}

static int launch_rht_update_w(const NmfGeom& g, const float* R, const float* H, float* W, float* colsumW, float* hscale, int batch,
                               int xcd, hipStream_t s) {
    GemmArgs a = {};
    a.A = R; a.sA = g.sV; a.lda = g.ld; a.a_clamp = g.Fp - 1;
    a.B = H; a.sB = g.sH; a.ldb = g.ld; a.b_clamp = g.Kp - 1;
    a.M = g.Fm; a.N = g.K; a.Kd = g.N;
    a.batch = batch; a.xcd_affine = xcd & 1; a.concurrent = (xcd >> 1) & 1;
    a.tail_row = g.F - 1;
    a.C = W; a.sC = g.sW; a.ldc = g.Kp;
    a.out_colsum = colsumW; a.out_norm = hscale; a.s_out = g.Kp;
    // the LDS-DMA kernel carries only the full-tile form of this epilogue (every wave entirely inside or outside M, no
    // ragged atom tile); anything else takes the register-staged kernel with the generic one
    if (gccnmf_tune_dma && (a.M & 127) == 0 && (a.N & 63) == 0)
        return g.tail ? gccnmf_launch_gemm_dma<true, true, EPI_UPDW, true>(a, s) : gccnmf_launch_gemm_dma<true, true, EPI_UPDW, false>(a, s);
    return g.tail ? gccnmf_launch_gemm<4, 1, true, true, EPI_UPDW, true>(a, s) : gccnmf_launch_gemm<4, 1, true, true, EPI_UPDW, false>(a, s);
}

#ifdef GCCNMF_EXPERIMENTS
// ------------------------------------------------------------------------------------------
// One file alone (BASELINE config 2 as a single mixture): split-K
// ------------------------------------------------------------------------------------------
// A launch over ONE file has 64-80 small-batch tiles for 256 CUs, each a 64..78-step dependent chain: latency-bound.  The two
// long reductions are therefore cut into GCCNMF_SPLITS equal parts that run as independent "files" of the batched kernel
// (operand base + part * length, partial outputs side by side): W.H over the atoms, R.H^T over the columns.  The parts
// are added in ascending order by the consumer -- nmf_div_partials_kernel (R = V / sum) and nmf_update_w_kernel -- so the
// result does not depend on scheduling.  Zero padding makes the parts equal: Kp and Np are multiples of 64.

// R[f][n] = V[f][n] / (P_0 + P_1 + ... )[f][n] on the valid F x N region only (R's padding must stay zero)
// one float4 per thread; columns >= N of the last float4 are written as 0 (not 0/0)
__global__ __launch_bounds__(256) void nmf_div_partials_kernel(const float* __restrict__ V, const float* __restrict__ P, long sP,
                                                               int nsplit, int F, int N, int Np, float* __restrict__ R) {
    const int n4 = Np / 4;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)F * n4) return;
    const int f = (int)(idx / n4), n = 4 * (int)(idx - (long)f * n4);
    if (n >= N) return;
    const long i = (long)f * Np + n;
    float4 d = *(const float4*)(P + i);
    for (int sp = 1; sp < nsplit; ++sp) {
        const float4 t = *(const float4*)(P + sp * sP + i);
        d.x += t.x; d.y += t.y; d.z += t.z; d.w += t.w;
    }
    const float4 v = *(const float4*)(V + i);
    *(float4*)(R + i) = make_float4(v.x / d.x, n + 1 < N ? v.y / d.y : 0.f, n + 2 < N ? v.z / d.z : 0.f, n + 3 < N ? v.w / d.w : 0.f);
}

// reduction length (padded) worth cutting: at least 8 k-tiles per part
static bool single_file_split(const NmfGeom& g, int batch, int reduction, int splits) {
    const bool ring = gccnmf_tune_ring && gccnmf_ring_supports(reduction);        // balances unequal parts itself
    return batch == 1 && splits > 1 && gccnmf_tune_tile_policy != 1 && g.Fm > 128 && reduction >= GCCNMF_SPLITS * 128 &&
           (ring || (reduction / 16) % splits == 0);
}

// P_part = W[:, part] . (hscale * H)[part, :]   (EPI_STORE incl. the VALU tail row), then R = V / sum_part P_part
static int launch_wh_div_split(const NmfGeom& g, const float* V, const float* W, const float* H, const float* hscale, float* P, float* R,
                               hipStream_t s) {
    const int nsplit = gccnmf_tune_wh_splits;
    const bool ring = gccnmf_tune_ring && gccnmf_ring_supports(g.Kp);
    const int len = g.Kp / nsplit;                           // atoms per part (multiple of 16) -- equal parts for the register-staged kernel
    GemmArgs a = {};
    a.A = W; a.sA = len; a.lda = g.Kp; a.a_clamp = g.Fp - 1;
    a.B = H; a.sB = (long)len * g.Np; a.ldb = g.Np; a.b_clamp = g.Np - 4;
    a.M = g.Fm; a.N = g.N; a.Kd = len;
    a.batch = nsplit; a.xcd_affine = 0;
    a.bscale = hscale; a.s_bscale = len;
    if (ring) {                                              // the ring kernel balances the k-tiles itself (parts need not be equal: 3 parts of 64 tiles)
        a.kparts = nsplit;
        a.Kd = g.Kp;
    }
    a.tail_row = g.F - 1;
    a.C = P; a.sC = g.sV; a.ldc = g.Np;
    int rc;
    if (ring)
        rc = g.tail ? gccnmf_launch_gemm_ring<true, false, EPI_STORE, true>(a, s) : gccnmf_launch_gemm_ring<true, false, EPI_STORE, false>(a, s);
    else
        rc = g.tail ? gccnmf_launch_gemm<4, 1, true, false, EPI_STORE, true, 1>(a, s) : gccnmf_launch_gemm<4, 1, true, false, EPI_STORE, false, 1>(a, s);
    if (rc) return rc;
    hipLaunchKernelGGL(nmf_div_partials_kernel, dim3((unsigned)(((long)g.F * (g.Np / 4) + 255) / 256)), dim3(256), 0, s, V, P, g.sV, nsplit, g.F,
                       g.N, g.Np, R);
    GCCNMF_CHECK_LAUNCH();
    return GCCNMF_OK;
}

// U_part = R[:, part] . H[:, part]^T, rowsum_part = sum_{n in part} H   (consumed by nmf_update_w_kernel with nsplit parts)
static int launch_rht_split(const NmfGeom& g, const float* R, const float* H, float* Upart, float* rowsum_part, hipStream_t s) {
    const int nsplit = gccnmf_tune_rht_splits;
    const bool ring = gccnmf_tune_ring && gccnmf_ring_supports(g.Np);
    const int len = g.Np / nsplit;                           // columns per part (multiple of 16; beyond N both operands are zero)
    GemmArgs a = {};
    a.A = R; a.sA = len; a.lda = g.Np; a.a_clamp = g.Fp - 1;
    a.B = H; a.sB = len; a.ldb = g.Np; a.b_clamp = g.Kp - 1;
    a.M = g.Fm; a.N = g.K; a.Kd = len;
    a.batch = nsplit; a.xcd_affine = 0;
    if (ring) {                                              // balanced k-tile ranges over the N valid columns only (78 tiles, not 80)
        a.kparts = nsplit;
        a.Kd = g.N;
    }
    a.tail_row = g.F - 1;
    a.rowsumB = rowsum_part; a.s_rowsumB = g.Kp;
    a.C = Upart; a.sC = g.sU; a.ldc = g.Kp;
    if (ring)
        return g.tail ? gccnmf_launch_gemm_ring<true, true, EPI_STORE, true>(a, s) : gccnmf_launch_gemm_ring<true, true, EPI_STORE, false>(a, s);
    return g.tail ? gccnmf_launch_gemm<4, 1, true, true, EPI_STORE, true, 1>(a, s) : gccnmf_launch_gemm<4, 1, true, true, EPI_STORE, false, 1>(a, s);
}

#endif  // GCCNMF_EXPERIMENTS (single-file split-K)

// ------------------------------------------------------------------------------------------
// The direct path (csrc/direct.hip): launches that cannot fill the chip -- one mixture alone, the shape behind the reference's
// own function names (runGCCNMF.py:41 -> performKLNMF(V, dictionarySize, 100, 0))
// ------------------------------------------------------------------------------------------
// All four GEMMs of an iteration as "both operands reduction-major" products, one round of one workgroup per CU each, the
// reduction split inside the workgroup (no partial products through HBM, no combine launches):
//   K1  R  = V / (Wt^T . (s*H))           A = Wt [k][f]   B = H  [k][n]      -> R [f][n]
//   K2  H  = (s*H) * (W^T . R) / (...)    A = W  [f][k]   B = R  [f][n]      -> H [k][n] and Ht [n][k]
//   K3  Rt = (V / (Wt^T . H))^T           A = Wt [k][f]   B = H  [k][n]      -> Rt [n][f] (+ the Nyquist row into R)
//   K4  U  = Rt^T . Ht, rowsumH           A = Rt [n][f]   B = Ht [n][k]      -> U [f][k]
//   K4b W update (one pass)                                                   -> W [f][k] and Wt [k][f]
// The transposed copies (Wt, Ht, Rt) live behind the other scratch in the workspace.  F = 16 n + 1 (513): bin F-1 is the VALU tail
// row of K1 / K3 / K4 and the rank-1 epilogue term of K2.
struct DirectBufs {
    float *Wt, *Ht, *Rt;
    long sWt, sHt, sRt;
    int ldwt, ldht, ldrt;
};
static long direct_floats(const NmfGeom& g, int batch) { return batch <= GCCNMF_DIRECT_MAX_BATCH ? (long)batch * (g.sU + g.sH + g.sV) : 0; }
static bool direct_path(const NmfGeom& g, int batch) {
    return gccnmf_tune_direct && gccnmf_tune_tile_policy == 0 && batch <= gccnmf_tune_direct_batch && batch <= GCCNMF_DIRECT_MAX_BATCH;
}
static DirectBufs direct_bufs(const NmfGeom& g, float* base, int batch) {
    DirectBufs d;
    d.sWt = (long)g.Kp * g.Fp; d.sHt = (long)g.Np * g.Kp; d.sRt = (long)g.Np * g.Fp;
    d.ldwt = g.Fp; d.ldht = g.Kp; d.ldrt = g.Fp;
    d.Wt = base;
    d.Ht = d.Wt + batch * d.sWt;
    d.Rt = d.Ht + batch * d.sHt;
    return d;
}
static bool direct_tail(const NmfGeom& g) { return g.F > 16 && (g.F % 16) == 1; }

// K1 (hscale != nullptr) and K3 (transposed output)
static int direct_wh_div(const NmfGeom& g, const DirectBufs& d, const float* V, const float* W, const float* H, const float* hscale, float* R,
                         bool transposed, int batch, hipStream_t s) {
    DirectArgs a = {};
    a.A = d.Wt; a.sA = d.sWt; a.lda = d.ldwt;
    a.B = H; a.sB = g.sH; a.ldb = g.ld;
    a.M = direct_tail(g) ? g.F - 1 : g.F; a.N = g.N; a.Kd = g.K; a.batch = batch;
    a.bscale = hscale; a.s_bscale = g.Kp;
    if (direct_tail(g)) {
        a.tailA = W + (long)(g.F - 1) * g.Kp; a.s_tailA = g.sW; a.tail_row = g.F - 1;
    }
    a.C = R; a.sC = g.sV; a.ldc = g.ld;
    a.E0 = V; a.sE0 = g.sV; a.lde0 = g.ld;
    if (transposed) {
        a.Ct = d.Rt; a.sCt = d.sRt; a.ldct = d.ldrt;
    }
    return gccnmf_direct_launch(a, transposed ? DEPI_DIVT : DEPI_DIV, gccnmf_tune_direct_tile, s);
}

static int direct_update_h(const NmfGeom& g, const DirectBufs& d, const float* W, const float* R, float* H, const float* hscale,
                           const float* colsumW, float alpha, float eps, int batch, hipStream_t s) {
    DirectArgs a = {};
    a.A = W; a.sA = g.sW; a.lda = g.Kp;
    a.B = R; a.sB = g.sV; a.ldb = g.ld;
    a.M = g.K; a.N = g.N; a.Kd = g.F; a.batch = batch;
    if (direct_tail(g)) {
        a.Kd = g.F - 1;
        a.ktailA = W + (long)(g.F - 1) * g.Kp; a.s_ktailA = g.sW;
        a.ktailB = R + (long)(g.F - 1) * g.ld; a.s_ktailB = g.sV;
    }
    a.C = H; a.sC = g.sH; a.ldc = g.ld;
    a.Ct = d.Ht; a.sCt = d.sHt; a.ldct = d.ldht;
    a.E1 = hscale; a.sE1 = g.Kp;
    a.E2 = colsumW; a.sE2 = g.Kp;
    a.alpha = alpha; a.eps = eps;
    return gccnmf_direct_launch(a, DEPI_UPDH, gccnmf_tune_direct_tile, s);
}

static int direct_rht(const NmfGeom& g, const DirectBufs& d, const float* R, float* U, float* rowsumH, int batch, hipStream_t s) {
    DirectArgs a = {};
    a.A = d.Rt; a.sA = d.sRt; a.lda = d.ldrt;
    a.B = d.Ht; a.sB = d.sHt; a.ldb = d.ldht;
    a.M = direct_tail(g) ? g.F - 1 : g.F; a.N = g.K; a.Kd = g.N; a.batch = batch;
    if (direct_tail(g)) {
        a.tailA = R + (long)(g.F - 1) * g.ld; a.s_tailA = g.sV; a.tail_row = g.F - 1;
    }
    a.rowsumB = rowsumH; a.s_rowsumB = g.Kp;
    a.C = U; a.sC = g.sU; a.ldc = g.Kp;
    return gccnmf_direct_launch(a, DEPI_STORE, gccnmf_tune_direct_tile, s);
}

// Short dictionaries (K <= 128 -- the reference driver's K = 128): K1 and K2 as ONE launch (gccnmf_wh_updh_kernel: a workgroup per
// 64-frame column tile with its scaled H resident in registers, W in 64-bin chunks through LDS), R never written.
// Measured at K = 128, N = 1244 (profiles/r04w_fused_k12_sweep.txt; files: fused / two launches, us): 20: 82 / 92, 25: 86 / 109, 26: 127 / 123,
// 40: 153 / 186, 50: 160 / 189, 64: 208 / 240, 96: 312 / 333, 128: 379 / 431; K = 64, 64 files: 116 / 195.  A full round of 512 workgroups
// (two per CU) takes 78 us, a last round of at most 256 (one per CU) 45, a larger one 76; the two launches 25 + 0.17 us per column tile.
// Both scale alike with K, so the choice is made in those units.  (GCCNMF_FLAG_GROUPS: the file groups that run side by side are priced together.)
// file groups that run this call side by side on separate streams (GCCNMF_FLAG_GROUPS): launch forms are chosen for all of them together
static int concurrent_groups(int flags) {
    const int n = (flags >> 8) & 255;
    return (flags & 4) ? (n >= 2 ? n : 2) : 1;
}
static bool fused_wh_updh(const NmfGeom& g, int batch, int flags) {
    if (!gccnmf_tune_fused_k12 || gccnmf_tune_tile_policy != 0 || direct_path(g, batch) || batch < 2 || !g.tail || g.Fm < 64 || g.Fm > 512 ||
        (g.Fm % 64) != 0 || g.K > 128)
        return false;
    if (gccnmf_tune_fused_k12 == 2) return true;
    const long wgs = (long)batch * gccnmf_ceil_div(g.N, 64) * concurrent_groups(flags), rem = wgs % 512;
    const long fused = 780 * (wgs / 512) + (rem == 0 ? 0 : rem <= 256 ? 450 : 760), two = 250 + 17 * wgs / 10;      // tenths of a microsecond
    return fused < two;
}
static int launch_wh_updh(const NmfGeom& g, const float* V, const float* W, float* H, const float* hscale, const float* colsumW, float alpha,
                          float eps, int batch, hipStream_t s) {
    WhUpdhArgs a = {};
    a.W = W; a.sW = g.sW; a.lda = g.Kp;
    a.H = H; a.sH = g.sH; a.ldb = g.ld;
    a.V = V; a.sV = g.sV; a.ldv = g.ld;
    a.scale = hscale; a.colsum = colsumW; a.sVec = g.Kp;
    a.M = g.Fm; a.N = g.N; a.Kd = g.K; a.batch = batch;
    a.alpha = alpha; a.eps = eps;
    return gccnmf_wh_updh_launch(a, s);
}

// K3 and K4a as ONE launch (gccnmf_whdiv_rht_kernel: 64-bin slabs with their W rows in registers), K <= 128.
// Every slab workgroup walks ALL column tiles of its file, so the launch costs one "round" (two workgroups per CU, 512 at a time) however
// few workgroups it holds: measured at K = 128, N = 1244 a round takes 184 us against 3.72 us per file for the two launches it replaces
// (64 files: 184 against 238 us; 40 files: 184 against 149).  Both scale alike with K and N, so the choice is a ratio.  Returns the number of
// files (from the front of the batch) that take the slab launch: all of them, none, or whole rounds' worth with the REST of the files on
// the two launches behind it (72 files: 64 + 8 -> 184 + 40 us against 268 for the two launches over all of them, 368 for two slab rounds).
// (GCCNMF_FLAG_GROUPS: the file groups that run side by side share the rounds; they are not split.)
static int fused_whdiv_rht_files(const NmfGeom& g, int batch, int flags) {
    if (!gccnmf_tune_fused_k34 || gccnmf_tune_tile_policy != 0 || direct_path(g, batch) || batch < 2 || !g.tail || g.Fm < 64 || g.Fm > 512 ||
        (g.Fm % 64) != 0 || g.K > 128 || (g.Fm / 64) * 16 < 32 * gccnmf_ceil_div(g.K, 32))
        return 0;
    if (gccnmf_tune_fused_k34 == 2) return batch;
    const long slabs = g.Fm / 64, groups = concurrent_groups(flags), wgs = (long)batch * slabs * groups, rounds = (wgs + 511) / 512;
    const long per_round = 512 / slabs;                          // files per round
    // in hundredths of a microsecond: 3.72 us x slabs / 8 per file on the two launches, 184 us per round, 10 us for the extra launch pair
    const long two = 372 * slabs / 8;
    long best = two * batch * groups, head = 0;                  // all files on the two launches
    if (18400 * rounds < best) {
        best = 18400 * rounds;
        head = batch;
    }
    if (groups == 1) {
        const long r = batch / per_round;                        // whole rounds of files, the rest behind them on the two launches
        // (+ 10 %: the two launches are dearer per file on a small rest than the straight line says -- 104 files: 64 + 40 measured 717 us
        // per iteration against 699 for two slab rounds)
        if (r >= 1 && r * per_round < batch && 11 * (18400 * r + two * (batch - r * per_round) + 1000) < 10 * best) head = r * per_round;
    }
    return (int)head;
}
static int launch_whdiv_rht(const NmfGeom& g, const float* V, const float* W, const float* H, float* U, float* rowsumH, int batch, hipStream_t s) {
    WhdivRhtArgs a = {};
    a.W = W; a.sW = g.sW; a.lda = g.Kp;
    a.H = H; a.sH = g.sH; a.ldb = g.ld;
    a.V = V; a.sV = g.sV; a.ldv = g.ld;
    a.U = U; a.sU = g.sU; a.ldu = g.Kp;
    a.rowsumH = rowsumH; a.sVec = g.Kp;
    a.M = g.Fm; a.N = g.N; a.Kd = g.K; a.batch = batch;
    return gccnmf_whdiv_rht_launch(a, s);
}

// ---- chained launches of the iteration (tuning key 21; gemm_dma.h: GemmSync, gccnmf_gemm_chain_kernel) ---------------------------------
// Counters of one file group, behind everything else in the workspace: c12 [batch][tiles_n] (K1 -> K2, in 32-column blocks), c23
// [batch][tiles_n] (K2 -> K3), c34 [batch] (K3 -> K4, items), c41 [batch] (K4 -> the next iteration's K1, items), error [1].  Zeroed once per gccnmf_klnmf call; iteration `it` waits for
// (it + 1) x the per-iteration count, so nothing is reset between launches.
static long chain_counter_floats(const NmfGeom& g, int batch) { return (long)batch * (2L * gccnmf_ceil_div(g.N, 64) + 2) + 32; }      // ... error [1], 7 unused, tickets [8], XCCs seen per list [8], 8 unused
static long klnmf_workspace_base_floats(const NmfGeom& g, int batch) {
    long n = (long)batch * (g.sV + g.sU + 3L * g.Kp);
    if (batch == 1) n += GCCNMF_SPLITS * ((g.sV > g.sU ? g.sV : g.sU) + (long)g.Kp);
    return n + direct_floats(g, batch);
}
// Which iterations can be chained: the four GEMMs all on full-height LDS-DMA throughput tiles with one XCD-affine list per XCD, every
// file's tiles on ONE XCD in every stage (batch a multiple of 8: list x holds the files x, x + 8, ... in all four GEMMs).
static bool chain_capable(const NmfGeom& g, int batch, int flags, bool any_size = false);
static bool chain_rule(int batch, int flags);
// The lists of a chained launch (tuning key 23): 1 = whole files per XCD (hand-over through that XCD's L2), 2 = the file-major tile list cut into equal
// eighths (a file may straddle XCDs: agent-scope hand-over, the producer writes its L2 back before it signals), 0 = the plain launch's lists.
// By rule (key 23 = 1): whole files where they balance -- the longest list at most 2.7 % above the mean, i.e. multiples of 8 and e.g. 102 files -- else the
// spread lists.  Same box (profiles/r06y_files_sweep_*.txt, iteration as a fraction of peak, whole | spread): 64 files 0.845 | 0.830 (the write-back costs
// 1.7 %), 24: 0.774 | 0.761, 32: 0.840 | 0.826, 40: 0.849 | 0.832; 26: plain 0.698 | spread 0.786, 51: plain 0.781 | 0.826, 52: whole 0.795 | 0.826,
// 77: 0.826 | 0.829, 25: plain 0.763 | 0.774, 20: plain 0.642 | 0.685; 16 files: plain 0.747, chained 0.584 -- the four stages of two files per XCD
// cannot fill 64 slots, spread or not.
static int chain_list_mode(int batch) {
    const int key = gccnmf_tune_chain_rag;
    if (key == 0 || key == 2) return key;
    if (key == 3) return 1;
    const int longest = (batch + 7) / 8;
    return 1000L * 8 * longest <= 1027L * batch ? 1 : 2;
}
static int chain_stages(const NmfGeom& g, int batch, int flags) {
    const int want = gccnmf_tune_chain;
    if (!want || !chain_capable(g, batch, flags) || ((batch & 7) && chain_list_mode(batch) == 0)) return 0;
    if (gccnmf_tune_tile_policy != 1 && (long)batch * gccnmf_ceil_div(g.N, 64) < 256) return 0;      // (the small-batch tile's territory)
    if (want != 1) return want;                                   // a forced form
    // by rule: not beside another file group's launches (two chained launches that share the chip are slower than two plain ones: 155 k against 158 k
    // frames/s end to end); whole-file lists from three files per XCD on, spread lists from 20 files on
    if (flags & 4) return 0;
    return batch >= (chain_list_mode(batch) == 1 ? 24 : 20) ? 8 : 0;
}
// the shape and tuning conditions of any chained launch: the four GEMMs all on full-height LDS-DMA throughput tiles (K2 half-height up to 256 atoms)
// with one XCD-affine list per XCD
static bool chain_capable(const NmfGeom& g, int batch, int flags, bool any_size) {
    if (!gccnmf_tune_dma || gccnmf_tune_tile_policy == 2 || gccnmf_tune_tail_split > 1) return false;
    if (direct_path(g, batch) || (flags & 3) || batch < 8) return false;
    // (K a multiple of 128: a K2 tile whose last wave is partly beyond M takes the generic epilogue h * (acc / den) for that wave, a plain
    // launch on half-height tiles the lean one (h * acc) * (s / den) for the same rows: a few ulp apart, so the forms would not be bitwise equal)
    if (g.K <= 128 || g.Fm <= 128 || g.Fm > 512 || (g.Fm & 127) || (g.K & 127) || (g.F % 16) != 1 || !g.tail) return false;
    // (a plain launch of a few files takes the two-launch W update -- other kernels, other summation order; a ragged batch has no plain form to agree with)
    return any_size || can_fuse_w_update(g, batch);
}
static bool chain_rule(int batch, int flags) {
    // The rule (profiles/r06h_files_sweep_*.txt, one-stream iteration as a fraction of the f32 peak, plain launches -> whole-call chain):
    //   24 files 0.74 -> 0.78, 32: 0.81 -> 0.85, 40: 0.73 -> 0.86, 64: 0.83 -> 0.86, 72: 0.79 -> 0.86, 104: 0.80 -> 0.86 -- but 16 files (two per XCD:
    //   the four stages of so few files cannot fill 64 slots): 0.75 -> 0.59, and with whole-file lists the longest list sets the pace: 25
    //   files (4 on one XCD, 3 on the others) 0.77 -> 0.68, 51: a tie, 52 (7 | 6): 0.75 -> 0.80.  So: at least three files per XCD and a longest
    //   list at most 8 % above the mean.  Not beside another file group's launches (flag bit 2): two chained launches that share the chip
    //   are slower than two plain ones (155 k against 158 k frames/s end to end).
    if (flags & 4) return false;
    const int longest = (batch + 7) / 8;
    return batch >= 24 && 100L * 8 * longest <= 108L * batch;
}

// A ragged batch (gccnmf_klnmf_ragged): files of different lengths in ONE chained launch.  g is the geometry of the LONGEST file (every file's V, H,
// R live in blocks of that pitch); n[f] = the file's own column count.  The host deals the files out to the eight XCD lists (longest first, each to
// the list with the least work so far); the device tables sit behind the counters in the workspace.
struct RaggedPlan {
    const int* n;                                   // host [batch]
    int lists[8][GEMM_RAGGED_LMAX + 1];             // host: count, files
    const int* d_n;                                 // device copies
    const int* d_lists;
};

static int launch_klnmf_chain(int stages, const NmfGeom& g, const float* V, float* W, float* H, float* R, float* colsumW, float* hscale,
                              float alpha, float eps, int batch, int flags, unsigned* counters, int it0, int iterations, hipStream_t s,
                              const RaggedPlan* rg = nullptr) {
    GemmArgs a[4] = {};
    const int concurrent = (flags & 4) ? 1 : 0;
    for (int i = 0; i < 4; ++i) {
        a[i].batch = batch; a[i].xcd_affine = 1; a[i].concurrent = concurrent;
        a[i].ablate = gccnmf_tune_ablate; a[i].exact_div = gccnmf_tune_exact_div;
    }
    for (int i = 0; i < 3; i += 2) {             // K1 (pending row scale of H on the B fragments) and K3: R = V / (W.H)  (launch_wh_div)
        a[i].A = W; a[i].sA = g.sW; a[i].lda = g.Kp; a[i].a_clamp = g.Fp - 1;
        a[i].B = H; a[i].sB = g.sH; a[i].ldb = g.ld; a[i].b_clamp = g.Np - 4;
        a[i].M = g.Fm; a[i].N = g.N; a[i].Kd = g.K;
        a[i].tail_row = g.F - 1;
        a[i].C = R; a[i].sC = g.sV; a[i].ldc = g.ld;
        a[i].E0 = V; a[i].sE0 = g.sV;
    }
    a[0].bscale = hscale; a[0].s_bscale = g.Kp;
    // K2: H = (s*H) * (W^T.R) / (colsum W + alpha + eps)  (launch_update_h)
    a[1].A = W; a[1].sA = g.sW; a[1].lda = g.Kp; a[1].a_clamp = g.Kp - 4;
    a[1].B = R; a[1].sB = g.sV; a[1].ldb = g.ld; a[1].b_clamp = g.Np - 4;
    a[1].M = g.K; a[1].N = g.N; a[1].Kd = g.F - 1;
    a[1].ktailA = W + (long)(g.F - 1) * g.Kp; a[1].s_ktailA = g.sW;
    a[1].ktailB = R + (long)(g.F - 1) * g.ld; a[1].s_ktailB = g.sV;
    a[1].C = H; a[1].sC = g.sH; a[1].ldc = g.ld;
    a[1].E1 = hscale; a[1].sE1 = g.Kp;
    a[1].E2 = colsumW; a[1].sE2 = g.Kp;
    a[1].alpha = alpha; a[1].eps = eps;
    // K4: W = normalise(W * (R.H^T) / rowsum H), column sums, norms  (launch_rht_update_w)
    a[3].A = R; a[3].sA = g.sV; a[3].lda = g.ld; a[3].a_clamp = g.Fp - 1;
    a[3].B = H; a[3].sB = g.sH; a[3].ldb = g.ld; a[3].b_clamp = g.Kp - 1;
    a[3].M = g.Fm; a[3].N = g.K; a[3].Kd = g.N;
    a[3].tail_row = g.F - 1;
    a[3].C = W; a[3].sC = g.sW; a[3].ldc = g.Kp;
    a[3].out_colsum = colsumW; a[3].out_norm = hscale; a[3].s_out = g.Kp;
    const int tm1 = g.K <= 256 ? 2 : 4;            // K2's outputs are the K atoms: at most 256 rows -> half-height tiles, as in a plain launch
    GemmChain ch = {};
    for (int i = 0; i < 4; ++i) {
        int len = 0;
        if (i < stages) {
            // whole-file lists (key 23, default): any batch size, K4 never waits for a ragged K3 item at the end of a list
            const int tm = (i == 1 && tm1 == 2) ? 2 : 4;
            int grid = gemm_dma_plan(a[i], i != 3 && tm == 4, tm, rg ? 1 : chain_list_mode(batch));      // 0: the plain launch's lists | 1: whole files | 2: file-major equal eighths
            if (grid < 8 || a[i].lists != 8 || a[i].split) return GCCNMF_ERR_ARG;
            if (rg) {
                // per list: the sum over its files of tiles_m x the file's own column tiles (K4: the uniform atom tiles, the file's own reduction length)
                a[i].ragged_n = rg->d_n; a[i].ragged_lists = rg->d_lists; a[i].ragged_kd = i == 3 ? 1 : 0;
                long longest = 0;
                for (int l = 0; l < 8; ++l) {
                    long items = 0;
                    for (int k = 0; k < rg->lists[l][0]; ++k)
                        items += (long)a[i].tiles_m * (i == 3 ? a[i].tiles_n : gccnmf_ceil_div(rg->n[rg->lists[l][1 + k]], 64));
                    if (items > longest) longest = items;
                }
                if (longest < 1 || longest > (1L << 24)) return GCCNMF_ERR_ARG;
                a[i].cw = (int)longest;
                grid = 8 * a[i].cw;
            }
            len = grid / 8;
        }
        ch.first[i + 1] = ch.first[i] + len;
    }
    const int tn = gccnmf_ceil_div(g.N, 64);
    unsigned* c12 = counters;
    unsigned* c23 = c12 + (long)batch * tn;
    unsigned* c34 = c23 + (long)batch * tn;
    unsigned* c41 = c34 + batch;
    unsigned* err = c41 + batch;
    const bool wide = !rg && chain_list_mode(batch) == 2;          // a file's tiles spread over the XCDs: agent-scope hand-over, no XCC check
    for (int i = 0; i < 4; ++i) {
        ch.sync[i].error = err;
        ch.sync[i].timeout = gccnmf_tune_chain_fault ? 0 : GEMM_SYNC_TIMEOUT;      // (lab build, key 25: every consumer that has to wait gives up at once)
        ch.sync[i].xcc_seen = wide ? nullptr : err + 16;
        ch.sync[i].wide = wide ? 1 : 0;
    }
    // does a producer run a file's ragged last column tile as ONE narrow item?  (a uniform batch: decided for the launch; a ragged one: per file)
    auto narrow_items = [&](int i) { return rg ? a[i].narrow_ok : (a[i].rag ? 1 : 0); };
    // K1 -> K2: column tile j of a file is ready when its 32-column blocks (2; the ragged last tile as ONE narrow item: 1) are stored
    ch.sync[0].sig_cnt = c12; ch.sync[0].sig_stride = tn; ch.sync[0].sig_per_tile = 1;
    ch.sync[1].wait_cnt = c12; ch.sync[1].wait_stride = tn; ch.sync[1].wait_per_tile = 1;
    ch.sync[1].wait_need = 2u; ch.sync[1].prod_narrow = narrow_items(0);
    if (stages == 4) {
        // K2 -> K3: both atom tiles (tiles_m of K2) of column tile j;  K3 -> K4: every item of the file
        ch.sync[1].sig_cnt = c23; ch.sync[1].sig_stride = tn; ch.sync[1].sig_per_tile = 1;
        ch.sync[2].wait_cnt = c23; ch.sync[2].wait_stride = tn; ch.sync[2].wait_per_tile = 1;
        ch.sync[2].wait_need = 2u * a[1].tiles_m; ch.sync[2].prod_narrow = narrow_items(1);
        ch.sync[2].sig_cnt = c34; ch.sync[2].sig_stride = 1; ch.sync[2].sig_per_tile = 0;
        ch.sync[3].wait_cnt = c34; ch.sync[3].wait_stride = 1; ch.sync[3].wait_per_tile = 0;
        ch.sync[3].wait_need = (unsigned)(a[2].tiles_m * (rg ? 1 : a[2].tiles_n));
        ch.sync[3].wait_scale_tiles = rg ? 1 : 0;
        {
            // K4 -> the next iteration's K1 (also across launches: a call of very many iterations is a few chained launches, and forced per-iteration
            // launches keep counting -- the kernel boundary makes the wait trivially true there): every atom tile of the file (W, its column sums and the pending row scale are complete; R is free)
            ch.sync[3].sig_cnt = c41; ch.sync[3].sig_stride = 1; ch.sync[3].sig_per_tile = 0;
            ch.sync[0].wait_cnt = c41; ch.sync[0].wait_stride = 1; ch.sync[0].wait_per_tile = 0; ch.sync[0].wait_lag = 1;
            ch.sync[0].wait_need = (unsigned)(a[3].tiles_m * a[3].tiles_n);
        }
    }
    if (iterations > 1 && stages != 4) return GCCNMF_ERR_ARG;
    if (it0 == 0 && iterations == 1) ch.sync[0].wait_cnt = nullptr;      // (nothing to wait for: the counters were just zeroed)
    ch.it0 = it0; ch.iterations = iterations;
    ch.trace_it = it0 + (iterations > 4 ? iterations - 3 : iterations - 1);      // timeline builds: an iteration in the steady state of a whole-call launch, not its last
    if ((long)8 * ch.first[4] * iterations > (1L << 30)) return GCCNMF_ERR_ARG;
    for (int i = 0; i < stages; ++i) {           // timeline builds (gccnmf_debug_set_trace): stage i's item t of list x -> row 8 * (first[i] + t) + x
        a[i].trace = gccnmf_trace_buf ? gccnmf_trace_buf + 8L * 8 * ch.first[i] : nullptr;
        a[i].trace_rows = gccnmf_trace_buf ? gccnmf_trace_blocks - 8 * ch.first[i] : 0;
        if (a[i].trace_rows <= 0) a[i].trace = nullptr;
    }
    const int grid = 8 * ch.first[4] * iterations;
    const unsigned pad = gccnmf_tune_chain_solo ? 16384u : 0u;         // static 78 KB + 16 KB: one workgroup per CU
    if (!g.tail) return GCCNMF_ERR_ARG;
    if (stages == 2 && tm1 == 4) hipLaunchKernelGGL((gccnmf_gemm_chain_kernel<true, 2, 4>), dim3(grid), dim3(256), pad, s, a[0], a[1], a[2], a[3], ch);
    else if (stages == 2) hipLaunchKernelGGL((gccnmf_gemm_chain_kernel<true, 2, 2>), dim3(grid), dim3(256), pad, s, a[0], a[1], a[2], a[3], ch);
    else if (tm1 == 4) hipLaunchKernelGGL((gccnmf_gemm_chain_kernel<true, 4, 4>), dim3(grid), dim3(256), pad, s, a[0], a[1], a[2], a[3], ch);
    else hipLaunchKernelGGL((gccnmf_gemm_chain_kernel<true, 4, 2>), dim3(grid), dim3(256), pad, s, a[0], a[1], a[2], a[3], ch);
    GCCNMF_CHECK_LAUNCH();
    return GCCNMF_OK;
}

// Short dictionaries (K <= 128): the three launches of an iteration -- K1 + K2 on column tiles, K3 + K4a on bin slabs, the one-pass W update --
// chained the same way (direct.hip: gccnmf_short_chain_kernel).  Shapes both fused kernels and the one-pass W update take; the same batch
// rule as the throughput chain.  Returns the atoms per W-update group (16 / 32: what the plain launch would use at this batch), 0 = no chain.
static int short_chain_group(const NmfGeom& g, int batch, int flags) {
    const int want = gccnmf_tune_chain;
    if (!want || want == 2 || want == 4 || !gccnmf_tune_fused_k12 || !gccnmf_tune_fused_k34 || gccnmf_tune_tile_policy != 0) return 0;
    if (direct_path(g, batch) || (flags & 3) || batch < 8 || !g.tail || g.Fm < 64 || g.Fm > 512 || (g.Fm % 64) != 0 || g.K > 128) return 0;
    if ((g.Fm / 64) * 16 < 32 * gccnmf_ceil_div(g.K, 32) || g.F > 64 * 9 || !gccnmf_tune_ring || (long)batch * (g.Kp / 64) >= 256) return 0;
    // by rule: only where the plain call runs the SAME three item programs for every file (both fused launches, no files left to the two-launch
    // form) -- the chained call is then bit for bit the plain one, and a file's bits keep following the batch size exactly as before
    if (want == 1 && !(chain_rule(batch, flags) && fused_wh_updh(g, batch, flags) && fused_whdiv_rht_files(g, batch, flags) == batch)) return 0;
    if (gccnmf_tune_ablate == 64) return 0;
    return ((long)batch * (g.Kp / 32) >= 256 && gccnmf_tune_wide_update_w) ? 32 : 16;
}

static int launch_short_chain(const NmfGeom& g, const float* V, float* W, float* H, float* U, float* colsumW, float* rowsumH, float* hscale,
                              float alpha, float eps, int batch, int group, unsigned* counters, int it0, int iterations, hipStream_t s) {
    ShortChainArgs c = {};
    c.a12.W = W; c.a12.sW = g.sW; c.a12.lda = g.Kp;
    c.a12.H = H; c.a12.sH = g.sH; c.a12.ldb = g.ld;
    c.a12.V = V; c.a12.sV = g.sV; c.a12.ldv = g.ld;
    c.a12.scale = hscale; c.a12.colsum = colsumW; c.a12.sVec = g.Kp;
    c.a12.M = g.Fm; c.a12.N = g.N; c.a12.Kd = g.K; c.a12.batch = batch;
    c.a12.alpha = alpha; c.a12.eps = eps;
    c.a34.W = W; c.a34.sW = g.sW; c.a34.lda = g.Kp;
    c.a34.H = H; c.a34.sH = g.sH; c.a34.ldb = g.ld;
    c.a34.V = V; c.a34.sV = g.sV; c.a34.ldv = g.ld;
    c.a34.U = U; c.a34.sU = g.sU; c.a34.ldu = g.Kp;
    c.a34.rowsumH = rowsumH; c.a34.sVec = g.Kp;
    c.a34.M = g.Fm; c.a34.N = g.N; c.a34.Kd = g.K; c.a34.batch = batch;
    c.aw.W = W; c.aw.U = U; c.aw.rowsumH = rowsumH; c.aw.colsumW = colsumW; c.aw.hscale = hscale;
    c.aw.F = g.F; c.aw.K = g.K; c.aw.Kp = g.Kp; c.aw.sW = g.sW; c.aw.sU = g.sU; c.aw.sVec = g.Kp; c.aw.sRowsum = g.Kp;
    c.it0 = it0; c.iterations = iterations; c.atoms_per_group = group; c.solo = gccnmf_tune_chain_solo;
    c.counters = counters;
    c.error = counters + chain_counter_floats(g, batch) - 32;
    c.xcc_seen = c.error + 16;
    c.timeout = gccnmf_tune_chain_fault ? 0 : GEMM_SYNC_TIMEOUT;
    return gccnmf_short_chain_launch(c, s);
}

// a consumer gave up waiting (GEMM_SYNC_TIMEOUT), or the workgroups of a list were NOT all on one XCD (the data hand-over through that XCD's
// L2 is then not guaranteed): the factors cannot be trusted -- make them NaN so that nothing downstream looks plausible
__global__ void nmf_chain_poison_kernel(const unsigned* __restrict__ err, float* __restrict__ W, float* __restrict__ H, long nW, long nH) {
    bool bad = err[0] != 0u;
    for (int l = 0; l < 8; ++l) bad = bad || __popc(err[16 + l]) > 1;
    if (!bad) return;
    const float nan = __int_as_float(0x7fc00000);
    for (long i = blockIdx.x * 256L + threadIdx.x; i < nW; i += 256L * gridDim.x) W[i] = nan;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < nH; i += 256L * gridDim.x) H[i] = nan;
}

extern "C" {

// R [batch][Fp][Np] | U [batch][Fp][Kp] | colsumW, rowsumH, hscale [batch][Kp] each | (batch == 1) the split-K partials:
// GCCNMF_SPLITS x max(Fp*Np, Fp*Kp) (W.H parts and R.H^T parts use the same memory at different stages) + GCCNMF_SPLITS x Kp
// | (batch <= GCCNMF_DIRECT_MAX_BATCH) the transposed copies of the direct path: Wt [batch][Kp][Fp], Ht [batch][Np][Kp], Rt [batch][Np][Fp]
long gccnmf_klnmf_workspace_floats(int F, int N, int K, int batch) {
    GCCNMF_ENTER();
    if (F < 2 || N < 1 || K < 1 || batch < 1) return -1;
    NmfGeom g = make_geom(F, N, K);
    // R | U | colsumW | rowsumH | hscale | split-K scratch (one file) | Wt | Ht | Rt of the direct path (a handful of files at most) | chain counters
    return klnmf_workspace_base_floats(g, batch) + chain_counter_floats(g, batch);
}

// One launch group of the iteration, addressable on its own so that tests and the benchmark can time /
// check each kernel in isolation.  stage: 0 prepare | 1 K1 | 2 K2 | 3 K3 | 4 K4a | 5 K4b | 6 final H rescale
static int klnmf_stage(int stage, const float* V, float* W, float* H, float* workspace, const NmfGeom& g, int batch,
                       float alpha, float eps, int flags, hipStream_t s) {
    float* R = workspace;
    float* U = R + (long)batch * g.sV;
    float* colsumW = U + (long)batch * g.sU;
    float* rowsumH = colsumW + (long)batch * g.Kp;
    float* hscale = rowsumH + (long)batch * g.Kp;
    float* parts = hscale + (long)batch * g.Kp;                                   // batch == 1 only
    float* rowsum_parts = parts + GCCNMF_SPLITS * (g.sV > g.sU ? g.sV : g.sU);
    float* direct_base = batch == 1 ? rowsum_parts + GCCNMF_SPLITS * (long)g.Kp : parts;
    const bool fused12 = fused_wh_updh(g, batch, flags);
    const int head34 = fused_whdiv_rht_files(g, batch, flags), rest34 = batch - head34;      // files on the slab launch | behind it on the two launches
    const bool fused34 = head34 > 0;
    if (direct_path(g, batch)) {
        const DirectBufs d = direct_bufs(g, direct_base, batch);
        switch (stage) {
            case 0: {
                // zero: R's padding (reduction operand of K2), Rt / Ht rows n >= N (reduction operands of K4), Wt columns f >= F
                if (hipMemsetAsync(R, 0, sizeof(float) * batch * g.sV, s) != hipSuccess) return GCCNMF_ERR_LAUNCH;
                if (hipMemsetAsync(d.Wt, 0, sizeof(float) * direct_floats(g, batch), s) != hipSuccess) return GCCNMF_ERR_LAUNCH;
                hipLaunchKernelGGL(nmf_prepare_kernel, dim3(batch * (g.Kp / 16)), dim3(256), 0, s, W, colsumW, hscale, g.F, g.Fp, g.Kp);
                GCCNMF_CHECK_LAUNCH();
                return gccnmf_transpose_launch(W, g.sW, g.Kp, d.Wt, d.sWt, d.ldwt, g.F, g.Kp, batch, s);
            }
            case 1: return direct_wh_div(g, d, V, W, H, hscale, R, false, batch, s);
            case 2: return direct_update_h(g, d, W, R, H, hscale, colsumW, alpha, eps, batch, s);
            case 3: return direct_wh_div(g, d, V, W, H, nullptr, R, true, batch, s);
            case 4: return direct_rht(g, d, R, U, rowsumH, batch, s);
            case 5:
                return launch_update_w(W, U, rowsumH, colsumW, hscale, g.F, g.Fp, g.K, g.Kp, g.sW, g.sU, (long)g.Kp, (long)g.Kp, batch, s, 1, 0, 0,
                                       d.Wt, d.sWt, d.ldwt);
            case 6:
                hipLaunchKernelGGL(nmf_scale_h_kernel, dim3(batch * g.K), dim3(256), 0, s, H, hscale, (long)g.Kp, g.K, g.sH, g.ld, g.Np);
                GCCNMF_CHECK_LAUNCH();
                return GCCNMF_OK;
            default: return GCCNMF_ERR_ARG;
        }
    }
#ifdef GCCNMF_EXPERIMENTS
    const bool split_wh = single_file_split(g, batch, g.Kp, gccnmf_tune_wh_splits);      // the round-3 latency path: one file's reductions as parts
    const bool split_rht = single_file_split(g, batch, g.Np, gccnmf_tune_rht_splits);
#else
    constexpr bool split_rht = false;              // the split-K path is compiled out of the product library
#endif
    const int xcd = ((flags & 1) ? 0 : 1) | ((flags & 4) ? 2 : 0);      // bit 1: another file group's launches run beside these
    const int vec_grid = batch * (g.Kp / 16);
    switch (stage) {
        case 0:
            // R's padding (rows >= F, columns >= N) must be zero: it is a reduction operand of K2 and K4a.
            if (hipMemsetAsync(R, 0, sizeof(float) * batch * g.sV, s) != hipSuccess) return GCCNMF_ERR_LAUNCH;
            hipLaunchKernelGGL(nmf_prepare_kernel, dim3(vec_grid), dim3(256), 0, s, W, colsumW, hscale, g.F, g.Fp, g.Kp);
            break;
        case 1:
            if (fused12) return launch_wh_updh(g, V, W, H, hscale, colsumW, alpha, eps, batch, s);                         // K1 + K2
#ifdef GCCNMF_EXPERIMENTS
            if (split_wh) return launch_wh_div_split(g, V, W, H, hscale, parts, R, s);
#endif
            return launch_wh_div(g, V, W, g.sW, H, hscale, g.Kp, R, batch, xcd, s);
        case 2:
            if (fused12) return GCCNMF_OK;                                                                             // done by stage 1
            return launch_update_h(g, W, g.sW, R, H, hscale, g.Kp, colsumW, g.Kp, alpha, eps, batch, xcd, s);
        case 3:
            if (fused34) {                                                                                             // K3 + K4a
                const int rc = launch_whdiv_rht(g, V, W, H, U, rowsumH, head34, s);
                if (rc || !rest34) return rc;
                return launch_wh_div(g, V + head34 * g.sV, W + head34 * g.sW, g.sW, H + head34 * g.sH, nullptr, 0, R + head34 * g.sV, rest34, xcd, s);
            }
#ifdef GCCNMF_EXPERIMENTS
            if (split_wh) return launch_wh_div_split(g, V, W, H, nullptr, parts, R, s);
#endif
            return launch_wh_div(g, V, W, g.sW, H, nullptr, 0, R, batch, xcd, s);
        case 4:
            if (fused34)                                                                                               // done by stage 3 ...
                return rest34 ? launch_rht(g, R + head34 * g.sV, H + head34 * g.sH, U + head34 * g.sU, rowsumH + (long)head34 * g.Kp, rest34, xcd, s)
                              : GCCNMF_OK;                                                                             // ... but for the rest
#ifdef GCCNMF_EXPERIMENTS
            if (split_rht) return launch_rht_split(g, R, H, parts, rowsum_parts, s);
#endif
            if (can_fuse_w_update(g, batch) && !(flags & 2)) return launch_rht_update_w(g, R, H, W, colsumW, hscale, batch, xcd, s);
            return launch_rht(g, R, H, U, rowsumH, batch, xcd, s);
        case 5:
            if (split_rht)
                return launch_update_w(W, parts, rowsum_parts, colsumW, hscale, g.F, g.Fp, g.K, g.Kp, g.sW, g.sU, (long)g.Kp, (long)g.Kp, batch,
                                       s, gccnmf_tune_rht_splits, g.sU, (long)g.Kp);
            if (can_fuse_w_update(g, batch) && !(flags & 2) && !fused34) return GCCNMF_OK;     // done by stage 4's epilogue
            return launch_update_w(W, U, rowsumH, colsumW, hscale, g.F, g.Fp, g.K, g.Kp, g.sW, g.sU, (long)g.Kp, (long)g.Kp, batch, s);
        case 6:
            hipLaunchKernelGGL(nmf_scale_h_kernel, dim3(batch * g.K), dim3(256), 0, s, H, hscale, (long)g.Kp, g.K, g.sH, g.ld, g.Np);
            break;
        default: return GCCNMF_ERR_ARG;
    }
    GCCNMF_CHECK_LAUNCH();
    return GCCNMF_OK;
}

// Which launches gccnmf_klnmf would use for this problem under the current tuning: bit 0 the direct latency kernels, bit 1 the fused
// K1 + K2 launch, bit 2 the fused K3 + K4a slab launch, bit 3 chained launches of the iteration (benchmarks and tests name the kernel they
// time by this; the engine keeps ONE file group when the library chains).
int gccnmf_klnmf_plan(int F, int N, int K, int batch, int flags) {
    GCCNMF_ENTER();
    if (F < 2 || N < 1 || K < 1 || batch < 1) return -1;
    const NmfGeom g = make_geom(F, N, K);
    return (direct_path(g, batch) ? 1 : 0) | (fused_wh_updh(g, batch, flags) ? 2 : 0) | (fused_whdiv_rht_files(g, batch, flags) > 0 ? 4 : 0) |
           ((chain_stages(g, batch, flags) || short_chain_group(g, batch, flags)) ? 8 : 0);
}

int gccnmf_klnmf_stage(const float* V, float* W, float* H, float* workspace, int F, int N, int K, int batch,
                       float sparsity_alpha, float epsilon, int flags, int stage, void* stream) {
    GCCNMF_ENTER();
    if (!V || !W || !H || !workspace || F < 2 || N < 1 || K < 1 || batch < 1) return GCCNMF_ERR_ARG;
    return klnmf_stage(stage, V, W, H, workspace, make_geom(F, N, K), batch, sparsity_alpha, epsilon, flags, (hipStream_t)stream);
}

int gccnmf_klnmf(const float* V, float* W, float* H, float* workspace, int F, int N, int K, int batch, int iterations,
                 float sparsity_alpha, float epsilon, int flags, void* stream) {
    GCCNMF_ENTER();
    if (!V || !W || !H || !workspace || F < 2 || N < 1 || K < 1 || batch < 1 || iterations < 0) return GCCNMF_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    NmfGeom g = make_geom(F, N, K);
    int rc;
    if ((rc = klnmf_stage(0, V, W, H, workspace, g, batch, sparsity_alpha, epsilon, flags, s))) return rc;
    const int chained = chain_stages(g, batch, flags);          // 0 | 2: K1 | K2 in one launch | 4: the whole iteration | 8: the whole call
    unsigned* counters = (unsigned*)(workspace + klnmf_workspace_base_floats(g, batch));
    const int short_group = chained ? 0 : short_chain_group(g, batch, flags);      // K <= 128: the three launches of every iteration as one chained launch
    // the status words describe THIS call (gccnmf_klnmf_chain_status): a call that does not chain clears what an earlier, chained one may have left
    if (!chained && !short_group && hipMemsetAsync(counters + chain_counter_floats(g, batch) - 32, 0, 32 * sizeof(unsigned), s) != hipSuccess) return GCCNMF_ERR_LAUNCH;
    if (short_group && iterations > 0) {
        if (hipMemsetAsync(counters, 0, sizeof(unsigned) * chain_counter_floats(g, batch), s) != hipSuccess) return GCCNMF_ERR_LAUNCH;
        float* R0 = workspace;
        float* U0 = R0 + (long)batch * g.sV;
        float* colsum0 = U0 + (long)batch * g.sU;
        float* rowsum0 = colsum0 + (long)batch * g.Kp;
        float* hscale0 = rowsum0 + (long)batch * g.Kp;
        for (int it0 = 0; it0 < iterations; it0 += GCCNMF_CHAIN_MAX_ITERATIONS)
            if ((rc = launch_short_chain(g, V, W, H, U0, colsum0, rowsum0, hscale0, sparsity_alpha, epsilon, batch, short_group, counters, it0,
                                         std::min(GCCNMF_CHAIN_MAX_ITERATIONS, iterations - it0), s)))
                return rc;
        hipLaunchKernelGGL(nmf_chain_poison_kernel, dim3(64), dim3(256), 0, s, counters + chain_counter_floats(g, batch) - 32, W, H, (long)batch * g.sW, (long)batch * g.sH);
        GCCNMF_CHECK_LAUNCH();
        return klnmf_stage(6, V, W, H, workspace, g, batch, sparsity_alpha, epsilon, flags, s);
    }
    if (chained && iterations > 0 && hipMemsetAsync(counters, 0, sizeof(unsigned) * chain_counter_floats(g, batch), s) != hipSuccess) return GCCNMF_ERR_LAUNCH;
    float* R = workspace;
    float* colsumW = R + (long)batch * g.sV + (long)batch * g.sU;
    float* hscale = colsumW + 2L * batch * g.Kp;
    if (chained == 8 && iterations > 0) {
        for (int it0 = 0; it0 < iterations; it0 += GCCNMF_CHAIN_MAX_ITERATIONS)      // (one launch; a call of thousands of iterations: a few, the grid stays below 2^30 workgroups)
            if ((rc = launch_klnmf_chain(4, g, V, W, H, R, colsumW, hscale, sparsity_alpha, epsilon, batch, flags, counters, it0,
                                         std::min(GCCNMF_CHAIN_MAX_ITERATIONS, iterations - it0), s)))
                return rc;
    } else {
        for (int it = 0; it < iterations; ++it) {
            if (chained && (rc = launch_klnmf_chain(chained, g, V, W, H, R, colsumW, hscale, sparsity_alpha, epsilon, batch, flags, counters, it, 1, s))) return rc;
            for (int stage = chained + 1; stage <= 5; ++stage)
                if ((rc = klnmf_stage(stage, V, W, H, workspace, g, batch, sparsity_alpha, epsilon, flags, s))) return rc;
        }
    }
    if (chained && iterations > 0) {
        hipLaunchKernelGGL(nmf_chain_poison_kernel, dim3(64), dim3(256), 0, s, counters + chain_counter_floats(g, batch) - 32, W, H, (long)batch * g.sW, (long)batch * g.sH);
        GCCNMF_CHECK_LAUNCH();
    }
    return klnmf_stage(6, V, W, H, workspace, g, batch, sparsity_alpha, epsilon, flags, s);
}

// Did the chained launches of the LAST gccnmf_klnmf / gccnmf_klnmf_ragged call on this workspace hand over cleanly?  (The call itself is
// asynchronous and poisons W, H with NaN if they did not; this is the explicit check: a blocking 4-byte read.)  status: 0 = clean (or the call
// did not chain), bit 0 = a consumer gave up waiting for its producer (GEMM_SYNC_TIMEOUT), bit 1 = the workgroups of some list ran on more
// than one XCC (the hand-over through one XCD's L2 is then not guaranteed).  N: the column count the workspace was sized with (Nmax for a ragged batch).
int gccnmf_klnmf_chain_status(const float* workspace, int F, int N, int K, int batch, int* status) {
    GCCNMF_ENTER();
    if (!workspace || !status || F < 2 || N < 1 || K < 1 || batch < 1) return GCCNMF_ERR_ARG;
    NmfGeom g = make_geom(F, N, K);
    const unsigned* words = (const unsigned*)(workspace + klnmf_workspace_base_floats(g, batch)) + chain_counter_floats(g, batch) - 32;
    unsigned host[32];
    if (hipMemcpy(host, words, sizeof(host), hipMemcpyDeviceToHost) != hipSuccess) return GCCNMF_ERR_LAUNCH;
    int st = host[0] ? 1 : 0;
    for (int l = 0; l < 8; ++l)
        if (__builtin_popcount(host[16 + l]) > 1) st |= 2;
    *status = st;
    return GCCNMF_OK;
}

// ---- ragged batches: mixtures of different lengths in one call (gccNMF/runGCCNMF.py:30-36 separates a file of ANY length) -------------
struct RaggedTables {
    int v[GCCNMF_RAGGED_MAX_BATCH + 8 * (GEMM_RAGGED_LMAX + 1)];
};
__global__ void nmf_store_ragged_tables_kernel(RaggedTables t, int* dst, int n) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = t.v[i];
}
static long ragged_table_ints(int batch) { return (long)gccnmf_round_up(batch + 8 * (GEMM_RAGGED_LMAX + 1), 4); }

long gccnmf_klnmf_ragged_workspace_floats(int F, int Nmax, int K, int batch) {
    GCCNMF_ENTER();
    if (F < 2 || Nmax < 1 || K < 1 || batch < 1 || batch > GCCNMF_RAGGED_MAX_BATCH) return -1;
    NmfGeom g = make_geom(F, Nmax, K);
    return klnmf_workspace_base_floats(g, batch) + chain_counter_floats(g, batch) + ragged_table_ints(batch);
}

int gccnmf_klnmf_ragged(const float* V, float* W, float* H, float* workspace, int F, const int* N, int Nmax, int K, int batch, int iterations,
                        float sparsity_alpha, float epsilon, int flags, void* stream) {
    GCCNMF_ENTER();
    if (!V || !W || !H || !workspace || !N || F < 2 || Nmax < 1 || K < 1 || batch < 1 || iterations < 0) return GCCNMF_ERR_ARG;
    if (batch > GCCNMF_RAGGED_MAX_BATCH) return GCCNMF_ERR_UNSUPPORTED;
    long tiles = 0;
    for (int f = 0; f < batch; ++f) {
        if (N[f] < 1 || N[f] > Nmax) return GCCNMF_ERR_ARG;
        tiles += gccnmf_ceil_div(N[f], 64);
    }
    hipStream_t s = (hipStream_t)stream;
    NmfGeom g = make_geom(F, Nmax, K);
    // a ragged batch exists only as a chained launch (the plain launches take one N for the whole batch): where that form is not available
    // -- short dictionaries, a handful of files -- the caller runs the files of each length as a batch of their own
    if (!gccnmf_tune_chain || !chain_capable(g, batch, flags & ~4, true) || (gccnmf_tune_tile_policy != 1 && tiles < 256)) return GCCNMF_ERR_UNSUPPORTED;
    // deal the files out to the eight lists: longest first, each to the list with the fewest column tiles so far (ties: the lower list)
    RaggedPlan rg = {};
    rg.n = N;
    std::vector<int> order(batch);
    for (int f = 0; f < batch; ++f) order[f] = f;
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return N[x] > N[y]; });
    long load[8] = {};
    for (int f : order) {
        int best = 0;
        for (int l = 1; l < 8; ++l)
            if (load[l] < load[best]) best = l;
        int& cnt = rg.lists[best][0];
        if (cnt >= GEMM_RAGGED_LMAX) return GCCNMF_ERR_UNSUPPORTED;
        rg.lists[best][1 + cnt++] = f;
        load[best] += gccnmf_ceil_div(N[f], 64);
    }
    for (int l = 0; l < 8; ++l) std::sort(rg.lists[l] + 1, rg.lists[l] + 1 + rg.lists[l][0]);      // a list serves its files in ascending order
    unsigned* counters = (unsigned*)(workspace + klnmf_workspace_base_floats(g, batch));
    int* tables = (int*)(counters + chain_counter_floats(g, batch));
    RaggedTables t = {};
    for (int f = 0; f < batch; ++f) t.v[f] = N[f];
    for (int l = 0; l < 8; ++l)
        for (int k = 0; k <= GEMM_RAGGED_LMAX; ++k) t.v[batch + l * (GEMM_RAGGED_LMAX + 1) + k] = rg.lists[l][k];
    hipLaunchKernelGGL(nmf_store_ragged_tables_kernel, dim3(1), dim3(256), 0, s, t, tables, batch + 8 * (GEMM_RAGGED_LMAX + 1));
    GCCNMF_CHECK_LAUNCH();
    rg.d_n = tables;
    rg.d_lists = tables + batch;
    int rc;
    if ((rc = klnmf_stage(0, V, W, H, workspace, g, batch, sparsity_alpha, epsilon, flags, s))) return rc;
    if (iterations > 0) {
        if (hipMemsetAsync(counters, 0, sizeof(unsigned) * chain_counter_floats(g, batch), s) != hipSuccess) return GCCNMF_ERR_LAUNCH;
        float* R = workspace;
        float* colsumW = R + (long)batch * g.sV + (long)batch * g.sU;
        float* hscale = colsumW + 2L * batch * g.Kp;
        for (int it0 = 0; it0 < iterations; it0 += GCCNMF_CHAIN_MAX_ITERATIONS)
            if ((rc = launch_klnmf_chain(4, g, V, W, H, R, colsumW, hscale, sparsity_alpha, epsilon, batch, flags & ~4, counters, it0,
                                         std::min(GCCNMF_CHAIN_MAX_ITERATIONS, iterations - it0), s, &rg)))
                return rc;
        hipLaunchKernelGGL(nmf_chain_poison_kernel, dim3(64), dim3(256), 0, s, counters + chain_counter_floats(g, batch) - 32, W, H, (long)batch * g.sW, (long)batch * g.sH);
        GCCNMF_CHECK_LAUNCH();
    }
    return klnmf_stage(6, V, W, H, workspace, g, batch, sparsity_alpha, epsilon, flags, s);
}

// ---- shared dictionary (one W for every file / rank) ------------------------------------------
// A rank's columns come as SHARDS: `batch` files of N columns each, either whole padded matrices back to back
// ([batch][Fp][Np] / [batch][Kp][Np], ld = 0) or -- ld > 0 -- column blocks of ONE matrix with row pitch ld (file b = columns
// [b*N, b*N+N) of V [Fp][ld], H [Kp][ld]; N a multiple of 64 unless batch == 1): the frame windows of one long mixture need no
// gather / scatter that way.  Per shard scratch: R (same layout as V) | Upart [batch][Fp][Kp] | rowsum_part [batch][Kp].
// Shared by all shards of a rank: W, vec = colsumW [Kp] | hscale [Kp], partial = num [Fp][Kp] | den [Kp].
struct SharedShard {
    const float* V;
    float *H, *R, *Upart, *rowsum_part;
    float *parts, *rowsum_parts;      // one file in the plain layout: scratch of the latency path's split-K partials, else nullptr
    bool latency;                     // ... and the current tuning sends it down that path
    bool direct;                      // ... on the direct-to-register kernels (round 4), else the round-3 split-K launches
    DirectBufs d;                     // transposed copies (Wt, Ht, Rt) of such a shard, behind the split-K scratch
    NmfGeom g;
    int batch;
    long r_floats;
};

static long shared_r_floats(const NmfGeom& g, int batch, int ld) { return ld > 0 ? (long)g.Fp * ld : (long)batch * g.sV; }

static bool shared_shard_ok(int F, int N, int K, int batch, int ld) {
    if (F < 2 || N < 1 || K < 1 || batch < 1 || ld < 0) return false;
    if (ld > 0 && ((ld & 3) || (long)batch * N > ld || (batch > 1 && (N & 63)))) return false;
    return true;
}

// A shard that is ONE file in the plain layout (a rank's whole, short column range: at eight ranks a 160 s mixture leaves 20 s each)
// is latency-bound exactly like one mixture alone: it takes that path's split-K launches (W.H over the atoms, R.H^T over the columns,
// csrc/nmf.hip "One file alone") instead of 64-80 workgroup launches with 64-78-step chains.
static bool shared_single_file_layout(const NmfGeom& g, int batch, int ld) { return batch == 1 && (ld == 0 || ld == g.Np); }   // sizes the scratch
static bool shared_latency_shard(const NmfGeom& g, int batch, int ld) {                                                        // decides per call
#ifdef GCCNMF_EXPERIMENTS
    return shared_single_file_layout(g, batch, ld) && single_file_split(g, 1, g.Kp, gccnmf_tune_wh_splits) &&
           single_file_split(g, 1, g.Np, gccnmf_tune_rht_splits);
#else
    return false;           // the product's latency path is the direct one (sh.direct); the split-K launches are an experiment build's
#endif
}
static long shared_split_floats(const NmfGeom& g) { return GCCNMF_SPLITS * ((g.sV > g.sU ? g.sV : g.sU) + (long)g.Kp) + direct_floats(g, 1); }

static SharedShard make_shard(const float* V, float* H, float* ws, int F, int N, int K, int batch, int ld) {
    SharedShard sh;
    sh.g = make_geom(F, N, K);
    sh.direct = shared_single_file_layout(sh.g, batch, ld) && direct_path(sh.g, 1);
    sh.latency = sh.direct || shared_latency_shard(sh.g, batch, ld);
    if (ld > 0) {
        sh.g.ld = ld;
        sh.g.sV = sh.g.sH = N;              // the next file is the next column block
    }
    sh.V = V; sh.H = H; sh.batch = batch;
    sh.r_floats = shared_r_floats(sh.g, batch, ld);
    sh.R = ws;
    sh.Upart = sh.R + sh.r_floats;
    sh.rowsum_part = sh.Upart + (long)batch * sh.g.sU;
    sh.parts = sh.rowsum_parts = nullptr;
    if (shared_single_file_layout(sh.g, batch, ld)) {
        sh.g = make_geom(F, N, K);              // (ld == Np: the plain single-file geometry; strides are irrelevant for one file)
        sh.parts = sh.rowsum_part + (long)batch * sh.g.Kp;
        sh.rowsum_parts = sh.parts + GCCNMF_SPLITS * (sh.g.sV > sh.g.sU ? sh.g.sV : sh.g.sU);
        sh.d = direct_bufs(sh.g, sh.rowsum_parts + GCCNMF_SPLITS * (long)sh.g.Kp, 1);
    }
    return sh;
}

static int shared_begin(const SharedShard* sh, int n, const float* W, float* colsumW, float* hscale, int F, int K, hipStream_t s) {
    NmfGeom g = make_geom(F, 1, K);
    for (int i = 0; i < n; ++i) {    // R's padding (rows >= F, columns >= N) is a reduction operand of K2 and K4a: zero
        if (hipMemsetAsync(sh[i].R, 0, sizeof(float) * sh[i].r_floats, s) != hipSuccess) return GCCNMF_ERR_LAUNCH;
        // ... and so are the rows n >= N of Ht / Rt and the columns f >= F of Wt on the direct path
        if (sh[i].parts && hipMemsetAsync(sh[i].d.Wt, 0, sizeof(float) * direct_floats(sh[i].g, 1), s) != hipSuccess) return GCCNMF_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(nmf_prepare_kernel, dim3(g.Kp / 16), dim3(256), 0, s, W, colsumW, hscale, g.F, g.Fp, g.Kp);
    GCCNMF_CHECK_LAUNCH();
    return GCCNMF_OK;
}

// Side streams of the shared-dictionary iteration.  K1 .. K4a of different files (column blocks) are independent until the W update,
// so a rank's files run as up to GCCNMF_SHARED_STREAMS groups on separate streams (tuning key 8, default 3): the tail of one group's
// launch -- when its last workgroups no longer fill both slots of every CU -- overlaps the head of another group's, exactly as the
// per-file-dictionary engine does with its file groups.  Same kernels on the same data: bitwise the one-stream result.
struct SidePool {
    hipStream_t side[GCCNMF_SHARED_STREAMS - 1];
    hipEvent_t fork, join[GCCNMF_SHARED_STREAMS - 1];
    bool ok = false;
};
static SidePool* side_pool() {
    static SidePool pools[GCCNMF_MAX_DEVICES];
    static std::mutex mu;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= GCCNMF_MAX_DEVICES) return nullptr;
    std::lock_guard<std::mutex> lock(mu);
    SidePool& p = pools[dev];
    if (!p.ok) {
        if (hipEventCreateWithFlags(&p.fork, hipEventDisableTiming) != hipSuccess) return nullptr;
        for (int i = 0; i < GCCNMF_SHARED_STREAMS - 1; ++i) {
            if (hipStreamCreateWithFlags(&p.side[i], hipStreamNonBlocking) != hipSuccess) return nullptr;
            if (hipEventCreateWithFlags(&p.join[i], hipEventDisableTiming) != hipSuccess) return nullptr;
        }
        p.ok = true;
    }
    return &p;
}

// files [b0, b1) of a shard as a shard of their own (same matrices, offset bases)
static SharedShard sub_shard(const SharedShard& sh, int b0, int b1) {
    SharedShard u = sh;
    u.V = sh.V + b0 * sh.g.sV;
    u.H = sh.H + b0 * sh.g.sH;
    u.R = sh.R + b0 * sh.g.sV;
    u.Upart = sh.Upart + b0 * sh.g.sU;
    u.rowsum_part = sh.rowsum_part + (long)b0 * sh.g.Kp;
    u.batch = b1 - b0;
    return u;
}

// H update with the current W, then the per-file (V/WH).H^T and row sums of H (K1, K2, K3, K4a on stream s)
static int shared_gemms(const SharedShard& sh, const float* W, const float* colsumW, const float* hscale, float alpha, float eps,
                        bool beside_others, hipStream_t s) {
    const NmfGeom& g = sh.g;
    int rc;
    // bit 1 = another unit's launches run beside these on a side stream: the tile-layout cost model of gemm_dma.h prices a launch that
    // has the chip to itself, so such units keep full tiles (their partial rounds overlap the neighbours' kernels)
    const int xcd = 1 | (beside_others ? 2 : 0);
    if (sh.direct) {                            // one file alone: the direct-to-register kernels (csrc/direct.hip), W transposed per iteration
        if ((rc = gccnmf_transpose_launch(W, 0, g.Kp, sh.d.Wt, sh.d.sWt, sh.d.ldwt, g.F, g.Kp, 1, s))) return rc;
        if ((rc = direct_wh_div(g, sh.d, sh.V, W, sh.H, hscale, sh.R, false, 1, s))) return rc;
        if ((rc = direct_update_h(g, sh.d, W, sh.R, sh.H, hscale, colsumW, alpha, eps, 1, s))) return rc;
        if ((rc = direct_wh_div(g, sh.d, sh.V, W, sh.H, nullptr, sh.R, true, 1, s))) return rc;
        return direct_rht(g, sh.d, sh.R, sh.Upart, sh.rowsum_part, 1, s);
    }
#ifdef GCCNMF_EXPERIMENTS
    if (sh.latency) {                           // one file alone: the split-K launches of the round-3 latency path
        if ((rc = launch_wh_div_split(g, sh.V, W, sh.H, hscale, sh.parts, sh.R, s))) return rc;
        if ((rc = launch_update_h(g, W, 0, sh.R, sh.H, hscale, 0, colsumW, 0, alpha, eps, 1, 0, s))) return rc;
        if ((rc = launch_wh_div_split(g, sh.V, W, sh.H, nullptr, sh.parts, sh.R, s))) return rc;
        return launch_rht_split(g, sh.R, sh.H, sh.parts, sh.rowsum_parts, s);
    }
#endif
    // files are independent inside K1-K3 and per file inside K4a: keep every tile of a file on one XCD (its H / R panels are shared
    // through that XCD's L2), exactly as the per-file-dictionary path does
    if ((rc = launch_wh_div(g, sh.V, W, 0, sh.H, hscale, 0, sh.R, sh.batch, xcd, s))) return rc;
    if ((rc = launch_update_h(g, W, 0, sh.R, sh.H, hscale, 0, colsumW, 0, alpha, eps, sh.batch, xcd, s))) return rc;
    // (H now carries the previous normalisation; K3 below takes no scale, and step B rewrites hscale before anyone reads it again)
    if ((rc = launch_wh_div(g, sh.V, W, 0, sh.H, nullptr, 0, sh.R, sh.batch, xcd, s))) return rc;
    return launch_rht(g, sh.R, sh.H, sh.Upart, sh.rowsum_part, sh.batch, xcd, s);
}

// partial (+)= [sum_files Upart || sum_files rowsum_part], files in ascending order (deterministic)
static int shared_reduce(const SharedShard& sh, float* partial, int accumulate, hipStream_t s) {
    const NmfGeom& g = sh.g;
    // (one file alone: the "files" are the parts of its split R.H^T reduction, added in ascending order)
    const bool split = sh.latency && !sh.direct;
    const float* U = split ? sh.parts : sh.Upart;
    const float* rowsum = split ? sh.rowsum_parts : sh.rowsum_part;
    const int n = split ? gccnmf_tune_rht_splits : sh.batch;
    hipLaunchKernelGGL(nmf_reduce_files_kernel, dim3((unsigned)((g.sU + 255) / 256)), dim3(256), 0, s, U, g.sU, n, g.sU, partial, accumulate);
    GCCNMF_CHECK_LAUNCH();
    hipLaunchKernelGGL(nmf_reduce_files_kernel, dim3(gccnmf_ceil_div(g.Kp, 256)), dim3(256), 0, s, rowsum, (long)g.Kp, n, (long)g.Kp,
                       partial + g.sU, accumulate);
    GCCNMF_CHECK_LAUNCH();
    return GCCNMF_OK;
}

// Step A of every shard of a rank: the GEMMs of the shards' file groups dealt out over the main stream and the side streams, joined,
// then the file-order reductions on the main stream.
static int shared_step_a_all(const SharedShard* sh, int n, const float* W, const float* colsumW, const float* hscale, float* partial,
                             float alpha, float eps, hipStream_t s) {
    int rc;
    SharedShard units[GCCNMF_MAX_SHARDS * GCCNMF_SHARED_STREAMS];
    int nu = 0;
    const int want = gccnmf_tune_shared_groups;
    for (int i = 0; i < n; ++i) {
        // Groups pay only when ONE launch over the whole shard cannot fill the chip (fewer than 512 throughput tiles = two per CU:
        // a 160 s mixture has 313): then the groups' launches, each a fraction of a round, run side by side and the short kernels of
        // different stages overlap.  A shard that fills the chip by itself loses with groups in lock-step (64 files: 151 k frames/s
        // as one group, 144 k as two, 130 k as three -- profiles/r03b): one group.
        const long tiles = (long)sh[i].batch * gccnmf_ceil_div(sh[i].g.N, 64);
        int groups = tiles >= 512 ? 1 : (int)(tiles / 64);
        if (groups > sh[i].batch) groups = sh[i].batch;
        groups = groups < 1 ? 1 : (groups > want ? want : groups);
        for (int q = 0; q < groups; ++q) units[nu++] = sub_shard(sh[i], (int)((long)sh[i].batch * q / groups), (int)((long)sh[i].batch * (q + 1) / groups));
    }
    SidePool* pool = (nu > 1 && want > 1) ? side_pool() : nullptr;
    const int lanes = pool ? (want < nu ? want : nu) : 1;
    if (lanes > 1) {
        if (hipEventRecord(pool->fork, s) != hipSuccess) return GCCNMF_ERR_LAUNCH;
        for (int l = 1; l < lanes; ++l)
            if (hipStreamWaitEvent(pool->side[l - 1], pool->fork, 0) != hipSuccess) return GCCNMF_ERR_LAUNCH;
    }
    for (int u = 0; u < nu; ++u) {
        const int l = u % lanes;
        if ((rc = shared_gemms(units[u], W, colsumW, hscale, alpha, eps, lanes > 1, l ? pool->side[l - 1] : s))) return rc;
    }
    for (int l = 1; l < lanes; ++l) {
        if (hipEventRecord(pool->join[l - 1], pool->side[l - 1]) != hipSuccess) return GCCNMF_ERR_LAUNCH;
        if (hipStreamWaitEvent(s, pool->join[l - 1], 0) != hipSuccess) return GCCNMF_ERR_LAUNCH;
    }
    for (int i = 0; i < n; ++i)
        if ((rc = shared_reduce(sh[i], partial, i > 0, s))) return rc;
    return GCCNMF_OK;
}

static int shared_step_b(float* W, const float* partial, float* colsumW, float* hscale, int F, int K, hipStream_t s) {
    NmfGeom g = make_geom(F, 1, K);
    return launch_update_w(W, partial, partial + g.sU, colsumW, hscale, g.F, g.Fp, g.K, g.Kp, 0L, 0L, 0L, 0L, 1, s);
}

static int shared_finish(const SharedShard* sh, int n, float* hscale, int K, hipStream_t s) {
    for (int i = 0; i < n; ++i) {
        const NmfGeom& g = sh[i].g;
        hipLaunchKernelGGL(nmf_scale_h_kernel, dim3(sh[i].batch * g.K), dim3(256), 0, s, sh[i].H, hscale, 0L, g.K, g.sH, g.ld, g.Np);
        GCCNMF_CHECK_LAUNCH();
    }
    const int Kp = gccnmf_round_up(K, 64);
    hipLaunchKernelGGL(nmf_fill_kernel, dim3(gccnmf_ceil_div(Kp, 256)), dim3(256), 0, s, hscale, 1.f, (long)Kp);
    GCCNMF_CHECK_LAUNCH();
    return GCCNMF_OK;
}

// the four-call protocol's workspace: one default-layout shard's scratch | colsumW [Kp] | hscale [Kp]
static float* legacy_vec(const SharedShard& sh) {
    return sh.parts ? sh.rowsum_parts + GCCNMF_SPLITS * (long)sh.g.Kp + direct_floats(sh.g, 1) : sh.rowsum_part + (long)sh.batch * sh.g.Kp;
}

long gccnmf_klnmf_shared_workspace_floats(int F, int N, int K, int batch) {
    GCCNMF_ENTER();
    if (!shared_shard_ok(F, N, K, batch, 0)) return -1;
    NmfGeom g = make_geom(F, N, K);
    return (long)batch * (g.sV + g.sU + g.Kp) + (shared_single_file_layout(g, batch, 0) ? shared_split_floats(g) : 0) + 2L * g.Kp;
}

long gccnmf_klnmf_shared_shard_workspace_floats(int F, int N, int K, int batch, int ld) {
    GCCNMF_ENTER();
    if (!shared_shard_ok(F, N, K, batch, ld)) return -1;
    NmfGeom g = make_geom(F, N, K);
    return shared_r_floats(g, batch, ld) + (long)batch * (g.sU + g.Kp) + (shared_single_file_layout(g, batch, ld) ? shared_split_floats(g) : 0);
}

long gccnmf_klnmf_shared_partial_floats(int F, int K) {
    GCCNMF_ENTER();
    if (F < 2 || K < 1) return -1;
    NmfGeom g = make_geom(F, 1, K);
    return g.sU + g.Kp;
}

int gccnmf_klnmf_shared_begin(const float* W, float* workspace, int F, int N, int K, int batch, void* stream) {
    GCCNMF_ENTER();
    if (!W || !workspace || !shared_shard_ok(F, N, K, batch, 0)) return GCCNMF_ERR_ARG;
    SharedShard sh = make_shard(nullptr, nullptr, workspace, F, N, K, batch, 0);
    float* vec = legacy_vec(sh);
    return shared_begin(&sh, 1, W, vec, vec + sh.g.Kp, F, K, (hipStream_t)stream);
}

int gccnmf_klnmf_shared_step_a(const float* V, const float* W, float* H, float* workspace, float* partial, int F, int N,
                               int K, int batch, float sparsity_alpha, float epsilon, void* stream) {
    GCCNMF_ENTER();
    if (!V || !W || !H || !workspace || !partial || !shared_shard_ok(F, N, K, batch, 0)) return GCCNMF_ERR_ARG;
    SharedShard sh = make_shard(V, H, workspace, F, N, K, batch, 0);
    float* vec = legacy_vec(sh);
    return shared_step_a_all(&sh, 1, W, vec, vec + sh.g.Kp, partial, sparsity_alpha, epsilon, (hipStream_t)stream);
}

int gccnmf_klnmf_shared_step_b(float* W, float* workspace, const float* partial, int F, int N, int K, int batch, void* stream) {
    GCCNMF_ENTER();
    if (!W || !workspace || !partial || !shared_shard_ok(F, N, K, batch, 0)) return GCCNMF_ERR_ARG;
    SharedShard sh = make_shard(nullptr, nullptr, workspace, F, N, K, batch, 0);
    float* vec = legacy_vec(sh);
    return shared_step_b(W, partial, vec, vec + sh.g.Kp, F, K, (hipStream_t)stream);
}

int gccnmf_klnmf_shared_finish(float* H, float* workspace, int F, int N, int K, int batch, void* stream) {
    GCCNMF_ENTER();
    if (!H || !workspace || !shared_shard_ok(F, N, K, batch, 0)) return GCCNMF_ERR_ARG;
    SharedShard sh = make_shard(nullptr, H, workspace, F, N, K, batch, 0);
    float* vec = legacy_vec(sh);
    return shared_finish(&sh, 1, vec + sh.g.Kp, K, (hipStream_t)stream);
}

// The whole shared-dictionary training of one rank in ONE call: begin, `iterations` x (step A of every shard -> all-reduce of
// `partial` -> step B), finish -- every launch and the collective enqueued on `stream` from C, no host round trip per iteration.
// One training at a time per device: the side streams and events of shared_step_a_all are per-device library state.  A second call that
// arrives while one is still enqueueing is REJECTED (GCCNMF_ERR_UNSUPPORTED) instead of racing on them.
namespace {
std::atomic<int> shared_run_busy[GCCNMF_MAX_DEVICES];
struct SharedRunGuard {
    int dev = -1;
    bool held = false;
    SharedRunGuard() {
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= GCCNMF_MAX_DEVICES) { dev = -1; held = true; return; }
        int expected = 0;
        held = shared_run_busy[dev].compare_exchange_strong(expected, 1);
    }
    ~SharedRunGuard() {
        if (held && dev >= 0) shared_run_busy[dev].store(0);
    }
};
}  // namespace

int gccnmf_klnmf_shared_run(const gccnmf_shared_shard* shards, int nshards, float* W, float* partial, float* vec, int F, int K,
                            int iterations, float sparsity_alpha, float epsilon, gccnmf_allreduce_fn allreduce, void* allreduce_ctx,
                            void* stream) {
    GCCNMF_ENTER();
    if (nshards < 0 || nshards > GCCNMF_MAX_SHARDS || (nshards && !shards) || !W || !partial || !vec || F < 2 || K < 1 || iterations < 0)
        return GCCNMF_ERR_ARG;
    SharedRunGuard guard;
    if (!guard.held) return GCCNMF_ERR_UNSUPPORTED;
    hipStream_t s = (hipStream_t)stream;
    SharedShard sh[GCCNMF_MAX_SHARDS];
    for (int i = 0; i < nshards; ++i) {
        const gccnmf_shared_shard& d = shards[i];
        if (!d.V || !d.H || !d.workspace || !shared_shard_ok(F, d.N, K, d.batch, d.ld)) return GCCNMF_ERR_ARG;
        sh[i] = make_shard(d.V, d.H, d.workspace, F, d.N, K, d.batch, d.ld);
    }
    const NmfGeom g = make_geom(F, 1, K);
    float *colsumW = vec, *hscale = vec + g.Kp;
    const long np = g.sU + g.Kp;
    int rc;
    if ((rc = shared_begin(sh, nshards, W, colsumW, hscale, F, K, s))) return rc;
    for (int it = 0; it < iterations; ++it) {
        // a rank without columns (fewer files than ranks) contributes a zero partial and still follows every W update
        if (nshards == 0 && hipMemsetAsync(partial, 0, sizeof(float) * np, s) != hipSuccess) return GCCNMF_ERR_LAUNCH;
        if (nshards && (rc = shared_step_a_all(sh, nshards, W, colsumW, hscale, partial, sparsity_alpha, epsilon, s))) return rc;
        if (allreduce && allreduce(allreduce_ctx, partial, np, stream) != 0) return GCCNMF_ERR_COLLECTIVE;
        if ((rc = shared_step_b(W, partial, colsumW, hscale, F, K, s))) return rc;
    }
    return shared_finish(sh, nshards, hscale, K, s);
}

#ifdef GCCNMF_EXPERIMENTS
int gccnmf_debug_mfma_peak(float* scratch, int blocks, int iters, void* stream) {
    GCCNMF_ENTER();
    if (!scratch || blocks < 1 || iters < 1) return GCCNMF_ERR_ARG;
    hipLaunchKernelGGL(mfma_peak_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, scratch, iters);
    GCCNMF_CHECK_LAUNCH();
    return GCCNMF_OK;
}
#endif

int gccnmf_debug_gemm_plan(int M, int N, int batch, int xcd_affine, int concurrent, int narrow_capable, int* plan, int* items, int max_items) {
    GCCNMF_ENTER();
    if (M < 1 || N < 1 || batch < 1 || !plan || (max_items > 0 && !items)) return -1;
    GemmArgs a = {};
    a.M = M; a.N = N; a.batch = batch; a.xcd_affine = xcd_affine; a.concurrent = concurrent;
    const int grid = gemm_dma_plan(a, (narrow_capable & 1) != 0, 4, (narrow_capable & 2) != 0);      // bit 1: the whole-file lists of chained launches
    if (grid < 1) return -1;
    const int fields[8] = {a.lists, a.cw, a.cr, a.split, a.rag, a.tiles_m, a.tiles_n, grid};
    for (int i = 0; i < 8; ++i) plan[i] = fields[i];
    int n = 0;
    for (int list = 0; list < a.lists; ++list)
        for (int t = 0;; ++t) {
            int file, tm, col0, nw;
            if (!gemm_dma_item(a, list, t, file, tm, col0, nw)) break;      // the same decode the kernel runs
            if (n < max_items) {
                int* it = items + 6 * n;
                it[0] = list; it[1] = t; it[2] = file; it[3] = tm; it[4] = col0; it[5] = nw;
            }
            ++n;
        }
    return n;
}

int gccnmf_debug_gemm(const float* A, const float* B, float* C, int M, int N, int Kd, int lda, int ldb, int ldc,
                      int a_clamp, int b_clamp, int layout, int batch, long sA, long sB, long sC, const float* bscale,
                      float* rowsumB, void* stream) {
    GCCNMF_ENTER();
    GemmArgs a = {};
    a.A = A; a.sA = sA; a.lda = lda; a.a_clamp = a_clamp;
    a.B = B; a.sB = sB; a.ldb = ldb; a.b_clamp = b_clamp;
    const bool tail = layout & 4;
    a.M = tail ? M - 1 : M; a.N = N; a.Kd = Kd;
    if (layout & 16) {   // rank-1 reduction tail (non-KC operands only)
        if (layout & 3) return GCCNMF_ERR_ARG;
        a.Kd = Kd - 1;
        a.ktailA = A + (long)(Kd - 1) * lda; a.s_ktailA = sA;
        a.ktailB = B + (long)(Kd - 1) * ldb; a.s_ktailB = sB;
    }
    a.tail_row = M - 1;
    a.batch = batch; a.xcd_affine = 1;
    a.bscale = bscale; a.s_bscale = 0;
    a.rowsumB = rowsumB; a.s_rowsumB = N;
    a.C = C; a.sC = sC; a.ldc = ldc;
    hipStream_t s = (hipStream_t)stream;
#ifdef GCCNMF_EXPERIMENTS
    if (layout & 32) return (layout & 31) ? GCCNMF_ERR_ARG : gccnmf_launch_gemm_stream(a, s);      // the LDS-free throughput tile (direct.hip)
#else
    if (layout & 32) return GCCNMF_ERR_UNSUPPORTED;
#endif
    const bool wide = layout & 8;
    if (!wide && gccnmf_tune_dma) {       // the throughput tile's default staging path (tuning key 3)
        switch (layout & 3) {
            case 0: return tail ? GCCNMF_ERR_ARG : gccnmf_launch_gemm_dma<false, false, EPI_STORE, false>(a, s);
            case 1: return tail ? gccnmf_launch_gemm_dma<true, false, EPI_STORE, true>(a, s) : gccnmf_launch_gemm_dma<true, false, EPI_STORE, false>(a, s);
            case 3: return tail ? gccnmf_launch_gemm_dma<true, true, EPI_STORE, true>(a, s) : gccnmf_launch_gemm_dma<true, true, EPI_STORE, false>(a, s);
            default: return GCCNMF_ERR_UNSUPPORTED;
        }
    }
    switch (layout & 3) {
        case 0:
            if (tail) return GCCNMF_ERR_ARG;
            return wide ? gccnmf_launch_gemm<1, 4, false, false, EPI_STORE, false>(a, s)
                        : gccnmf_launch_gemm<4, 1, false, false, EPI_STORE, false>(a, s);
        case 1:
            if (wide) return tail ? gccnmf_launch_gemm<1, 4, true, false, EPI_STORE, true>(a, s)
                                  : gccnmf_launch_gemm<1, 4, true, false, EPI_STORE, false>(a, s);
            return tail ? gccnmf_launch_gemm<4, 1, true, false, EPI_STORE, true>(a, s)
                        : gccnmf_launch_gemm<4, 1, true, false, EPI_STORE, false>(a, s);
        case 3:
            if (wide) return tail ? gccnmf_launch_gemm<1, 4, true, true, EPI_STORE, true>(a, s)
                                  : gccnmf_launch_gemm<1, 4, true, true, EPI_STORE, false>(a, s);
            return tail ? gccnmf_launch_gemm<4, 1, true, true, EPI_STORE, true>(a, s)
                        : gccnmf_launch_gemm<4, 1, true, true, EPI_STORE, false>(a, s);
        default:
            return GCCNMF_ERR_UNSUPPORTED;   // (A non-KC, B KC) is not used by the path
    }
}

}  // extern "C"
