// Batched f32 GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32: exact-f32, k-ordered
// fmaf chain), with the KL-NMF / GCC-NMF element-wise work fused into the epilogue.
//
// One launch covers every mixture file of a batch.  A workgroup owns a (WM*128) x (WN*64)
// output tile; each 64-lane wave owns 128 x 64 of it = 4 x 2 MFMA tiles = 128 accumulator
// registers.  Operand tiles (BK = 16) are staged global -> registers -> LDS with the next
// tile's global loads in flight under the current tile's MFMAs (one register set, written
// to LDS after the barrier).  f32 MFMA is 16x slower than bf16 MFMA, so one 16-deep tile
// is 4096 matrix-pipe cycles per wave against 9 x 16-byte loads per lane: the loop is
// MFMA-bound by construction and LDS traffic is <10 % of the LDS rate.
//
// F = n_fft/2 + 1 is hostile to 32-row MFMA tiles (513 = 16*32 + 1), so the kernel can
// carry ONE extra output row (the Nyquist bin) on the otherwise idle VALU: TAIL = true
// computes out[tail_row][:] = A[tail_row][:] . B from the B tile that is in LDS anyway.
//
// Operand layouts (what "KC" = reduction-index-contiguous means):
//   A_KC : A(i,kk) = A[i*lda + kk]   (W in W.H, R in R.H^T)      else A(i,kk) = A[kk*lda + i]  (W^T)
//   B_KC : B(kk,j) = B[j*ldb + kk]   (H in R.H^T)                else B(kk,j) = B[kk*ldb + j]  (H, R)
// Reference operations these GEMMs replace: numpy.dot at gccNMF/gccNMFFunctions.py:76,77,150 and the
// einsum contractions at :92,132-133.
#pragma once
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

// build-time experiment switches (scripts/kbench.py A/B): GEMM_VARIANT bit 0 = s_setprio(1) around a tile's MFMAs,
// bit 1 = no sched_barrier pins between the fragment reads and the MFMA groups
#ifndef GEMM_VARIANT
#define GEMM_VARIANT 2   // measured: the compiler's own placement of the fragment reads is ~1.5 % faster than the pinned order
#endif
#if GEMM_VARIANT & 1
#define GEMM_PRIO(x) __builtin_amdgcn_s_setprio(x)
#else
#define GEMM_PRIO(x)
#endif
#if GEMM_VARIANT & 2
#define GEMM_PIN()
#else
#define GEMM_PIN() __builtin_amdgcn_sched_barrier(0)
#endif

// run-time tuning knobs (gccnmf_set_tuning); defined in nmf.hip
extern long long* gccnmf_trace_buf;
extern int gccnmf_trace_blocks;

enum GemmEpilogue {
    EPI_STORE = 0,   // C[row][col] = acc
    EPI_DIV = 1,     // C[row][col] = E0[row][col] / acc                      (R = V / (W.H),  :76/:77)
    EPI_UPDH = 2,    // C[row][col] = (C[row][col]*E1[row]) * (acc / (E2[row] + alpha + eps))   (H update, :76)
    EPI_PHASE = 3,   // Cx[ic][row][t] = acc * X[c][row][t] / |X|              (:150-151)
    EPI_UPDW = 4     // block-level: W = normalise(W * acc / rowsum(B)); the tile must own every row  (:77,:79-80)
};

#define GEMM_RAGGED_LMAX 31      // files per XCD list of a ragged launch

struct GemmArgs {
    const float* A;
    const float* B;
    long sA, sB;               // per-file strides in floats (0 = shared by the batch)
    int lda, ldb;
    int M, N, Kd;              // MFMA output rows, output columns, reduction length
    int a_clamp, b_clamp;      // KC operand: last addressable row; non-KC operand: last addressable float4 start column
    int tiles_m, tiles_n, batch, xcd_affine;
    int concurrent;            // 1 = this launch shares the chip with launches of another stream (file groups): keep the throughput tile,
                               //     its partial round runs beside the other group's kernels
    int file0;                 // first file of this launch (a launch over a sub-range of the batch: strides are applied to file0 + index)
    int kparts;                // ring kernel only: > 0 = the `batch` "files" are kparts balanced parts of ONE reduction of Kd (split-K, partial outputs sC apart)
    int exact_div;             // LDS-DMA kernel, EPI_DIV: 1 = IEEE division instead of v_rcp_f32 + one Newton step (tuning key 7)
    int ablate;                // timing experiments: 1 no global loads, 2 no LDS stores, 4 no k-loop barrier, 8 no tail row, 16 no epilogue (results invalid)
    const float* bscale;       // optional per-reduction-index scale applied to B while staging (non-KC B only)
    long s_bscale;
    int tail_row;              // TAIL: index of the extra VALU-computed output row
    const float* ktailA;       // optional rank-1 reduction tail: acc[row][col] += ktailA[row] * ktailB[col] (the one
    const float* ktailB;       //   reduction index beyond a multiple of 16, e.g. f = 512 of F = 513, kept off the matrix cores)
    long s_ktailA, s_ktailB;
    float* out_colsum;         // EPI_UPDW outputs: column sums of the new W, and the atom norms (the lazy H row scale)
    float* out_norm;
    long s_out;
    float* rowsumB;            // optional: rowsumB[j] = sum_kk B(kk, j) (KC B only), written by the tm == 0 blocks
    long s_rowsumB;
    float* C;
    long sC;
    int ldc;
    const float* E0;
    long sE0;
    const float* E1;
    long sE1;
    const float* E2;
    long sE2;
    const float2* X;
    long sX;
    float alpha, eps;
    int T, Tp, Fp, ldv;        // EPI_PHASE geometry
    long long* trace;          // debug: per-workgroup timeline, 8 x int64 per block (gccnmf_debug_set_trace)
    // LDS-DMA throughput tile (gemm_dma.h): the launch's work lists, set by gccnmf_launch_gemm_dma
    int lists;                 // 8 = one ordered item list per XCD (blockIdx & 7), 1 = one list
    int cw, cr;                // wide tiles / ragged narrow tiles per list (chunks of the file-major lists)
    int split;                 // the last `split` wide tiles of every list run as two narrow (512 x 32) halves
    int rag, wide_n;           // rag = 1: the last column tile of a file is a narrow item; wide_n = tiles_n - rag
    int whole_files;           // chained launches: list x holds WHOLE files (x, x + 8, ...), each file's items together, its ragged items (rag) behind its wide tiles
    // ragged batches (files of different lengths in one chained launch; gccnmf_klnmf_ragged): device tables
    const int* ragged_n;       // [file] this GEMM's ragged extent of the file: its output COLUMNS (K1 - K3: N_f), or -- ragged_kd -- its REDUCTION length (K4: N_f)
    const int* ragged_lists;   // [8][GEMM_RAGGED_LMAX + 1]: per list the number of files, then their indexes in list order
    int ragged_kd;
    int narrow_ok;             // the instantiation carries the narrow loop and the tuning allows narrow items: a file's ragged last column tile is one
    int wpl, prefetch;         // persistent grid: resident workgroups per list; 1 = the next item's first k-tile is requested before the epilogue
    unsigned* tickets;         // persistent grid: [0..7] next-item counters per list, [8] workgroups gone; nullptr = classic grid
    int trace_rows, trace_grid;   // rows of the trace buffer; items of the classic grid (the per-wave probe rows start behind them)
};

template <int EPI>
__device__ __forceinline__ bool gemm_col_valid(const GemmArgs& p, int col) {
    if (EPI == EPI_PHASE) {
        int ic = col / p.Tp;
        return col < p.N && (col - ic * p.Tp) < p.T;
    }
    return col < p.N;
}

template <int EPI>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& p, int file, int row, int col, float acc) {
    if (EPI == EPI_STORE) {
        p.C[file * p.sC + (long)row * p.ldc + col] = acc;
    } else if (EPI == EPI_DIV) {
        long idx = (long)row * p.ldc + col;
        p.C[file * p.sC + idx] = p.E0[file * p.sE0 + idx] / acc;
    } else if (EPI == EPI_UPDH) {
        long idx = file * p.sC + (long)row * p.ldc + col;
        float h = p.C[idx];
        if (p.E1) h *= p.E1[file * p.sE1 + row];
        float den = p.E2[file * p.sE2 + row] + p.alpha + p.eps;
        p.C[idx] = h * (acc / den);
    } else {  // EPI_PHASE
        int ic = col / p.Tp;
        int t = col - ic * p.Tp;
        int c = ic & 1;
        float2 x = p.X[file * p.sX + ((long)c * p.Fp + row) * p.Tp + t];
        float v = p.E0[file * p.sE0 + (long)row * p.ldv + c * p.T + t];
        float2 ph;
        if (v > 0.f) {
            ph.x = x.x / v;
            ph.y = x.y / v;
        } else {
            ph.x = 1.f;   // numpy.angle(0) == 0 -> exp(0j) == 1
            ph.y = 0.f;
        }
        float2 o;
        o.x = acc * ph.x;
        o.y = acc * ph.y;
        ((float2*)p.C)[file * p.sC + ((long)ic * p.Fp + row) * p.Tp + t] = o;
    }
}

// Epilogue of one wave's pair of 32x32 MFMA tiles (columns col and col+32, same 16 rows per lane:
// row(r) = row_base + (r&3) + 8*(r>>2)).  Every global LOAD the epilogue needs is issued up front from
// clamped (always in-bounds) addresses, then the arithmetic, then the predicated stores: a per-element
// load -> use -> store chain would serialise 128 memory round trips per lane (the in-place H update cannot
// be reordered by the compiler) and cost as much as the whole k-loop.
template <int EPI>
__device__ __forceinline__ void gemm_epilogue_pair(const GemmArgs& p, int file, int row_base, int col_a, const f32x16& acc_in_a,
                                                   const f32x16& acc_in_b, bool allow_b = true) {
    const int col_b = col_a + 32;
    const bool ok_a = gemm_col_valid<EPI>(p, col_a), ok_b = allow_b && gemm_col_valid<EPI>(p, col_b);      // allow_b = false: a narrow (32-column) item
    // (the accumulators are never written here: a second code path that modifies them next to the lean epilogues of
    // gemm_dma.h doubles their live ranges and spills)
    float acc_a[16], acc_b[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        acc_a[r] = acc_in_a[r];
        acc_b[r] = acc_in_b[r];
    }
    if (EPI == EPI_STORE || EPI == EPI_UPDH) {
        if (p.ktailA) {   // last reduction index as one fmaf per element, in chain order (it is the final k)
            const float* __restrict__ ta = p.ktailA + file * p.s_ktailA;
            const float* __restrict__ tb = p.ktailB + file * p.s_ktailB;
            const float ba = tb[min(col_a, p.N - 1)], bb = tb[min(col_b, p.N - 1)];
            float av[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) av[r] = ta[min(row_base + (r & 3) + 8 * (r >> 2), p.M - 1)];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc_a[r] = fmaf(av[r], ba, acc_a[r]);
                acc_b[r] = fmaf(av[r], bb, acc_b[r]);
            }
        }
    }
    if (EPI == EPI_STORE) {
        float* __restrict__ C = p.C + file * p.sC;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = row_base + (r & 3) + 8 * (r >> 2);
            if (row < p.M) {
                if (ok_a) C[(long)row * p.ldc + col_a] = acc_a[r];
                if (ok_b) C[(long)row * p.ldc + col_b] = acc_b[r];
            }
        }
    } else if (EPI == EPI_DIV) {
        const float* __restrict__ V = p.E0 + file * p.sE0;
        float* __restrict__ C = p.C + file * p.sC;
        const int ca = min(col_a, p.N - 1), cb = min(col_b, p.N - 1);
        float va[16], vb[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long ro = (long)min(row_base + (r & 3) + 8 * (r >> 2), p.M - 1) * p.ldc;
            va[r] = V[ro + ca];
            vb[r] = V[ro + cb];
        }
        // (the quotients outside the predicated stores: inside them every element's division sat behind a branch with a full
        // `s_waitcnt vmcnt(0)` -- the previous element's store -- in front of it; an element that is not stored may divide by zero, harmlessly)
        float qa[16], qb[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            qa[r] = va[r] / acc_a[r];
            qb[r] = vb[r] / acc_b[r];
            asm volatile("" : "+v"(qa[r]), "+v"(qb[r]));      // (keeps the division from being sunk back into the store's branch)
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = row_base + (r & 3) + 8 * (r >> 2);
            if (row < p.M) {
                if (ok_a) C[(long)row * p.ldc + col_a] = qa[r];
                if (ok_b) C[(long)row * p.ldc + col_b] = qb[r];
            }
        }
    } else if (EPI == EPI_UPDH) {
        float* C = p.C + file * p.sC;                      // read-modify-write: all reads first
        const float* __restrict__ E1 = p.E1 ? p.E1 + file * p.sE1 : nullptr;
        const float* __restrict__ E2 = p.E2 + file * p.sE2;
        const int ca = min(col_a, p.N - 1), cb = min(col_b, p.N - 1);
        float ha[16], hb[16], den[16], sc[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rc = min(row_base + (r & 3) + 8 * (r >> 2), p.M - 1);
            ha[r] = C[(long)rc * p.ldc + ca];
            hb[r] = C[(long)rc * p.ldc + cb];
            den[r] = E2[rc];
            sc[r] = E1 ? E1[rc] : 1.f;
        }
        float ua[16], ub[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float d = den[r] + p.alpha + p.eps;
            ua[r] = (ha[r] * sc[r]) * (acc_a[r] / d);
            ub[r] = (hb[r] * sc[r]) * (acc_b[r] / d);
            asm volatile("" : "+v"(ua[r]), "+v"(ub[r]));
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = row_base + (r & 3) + 8 * (r >> 2);
            if (row < p.M) {
                if (ok_a) C[(long)row * p.ldc + col_a] = ua[r];
                if (ok_b) C[(long)row * p.ldc + col_b] = ub[r];
            }
        }
    } else {  // EPI_PHASE: one tile at a time (X is two registers per element)
        const float2* __restrict__ X = p.X + file * p.sX;
        const float* __restrict__ V = p.E0 + file * p.sE0;
        float2* __restrict__ C = (float2*)p.C + file * p.sC;
        const int nic = p.N / p.Tp;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int col = half ? col_b : col_a;
            const bool ok = half ? ok_b : ok_a;
            const int ic = min(col / p.Tp, nic - 1);
            const int t = min(col - (col / p.Tp) * p.Tp, p.T - 1);
            const int c = ic & 1;
            float2 x[16];
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rc = min(row_base + (r & 3) + 8 * (r >> 2), p.M - 1);
                x[r] = X[((long)c * p.Fp + rc) * p.Tp + t];
                v[r] = V[(long)rc * p.ldv + c * p.T + t];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row_base + (r & 3) + 8 * (r >> 2);
                const float a = half ? acc_b[r] : acc_a[r];
                float2 o;
                if (v[r] > 0.f) {
                    o.x = a * (x[r].x / v[r]);
                    o.y = a * (x[r].y / v[r]);
                } else {           // numpy.angle(0) == 0 -> exp(0j) == 1
                    o.x = a;
                    o.y = 0.f;
                }
                if (ok && row < p.M) C[((long)ic * p.Fp + row) * p.Tp + t] = o;
            }
        }
    }
}

// ---- operand staging: global -> registers (float4 per lane) -> LDS ---------------------------------
// Free functions on array references (no lambdas: a by-reference lambda capture of the register arrays
// sent them to scratch memory in the non-KC instantiations -- 144 B/lane of private-memory traffic per tile).
// Per-thread element offsets of its float4 units inside the operand (loop invariant, 32-bit: one VGPR each);
// the k-tile origin is folded into the wave-uniform base pointer so the loads use the SGPR-base addressing form.
template <int BMN, int NT, int U, bool KC>
__device__ __forceinline__ void gemm_operand_offsets(int (&off)[U], int ld, int origin, int clamp, int tid) {
#pragma unroll
    for (int i = 0; i < U; ++i) {
        const int u = tid + NT * i;
        if (KC) {
            const int row = u >> 2, c4 = u & 3;
            off[i] = min(origin + row, clamp) * ld + 4 * c4;
        } else {
            const int kk = u / (BMN / 4), c4 = u - kk * (BMN / 4);
            off[i] = kk * ld + min(origin + 4 * c4, clamp);
        }
    }
}

template <int U>
__device__ __forceinline__ void gemm_load_operand(float4 (&r)[U], const float* __restrict__ tile_base, const int (&off)[U]) {
#pragma unroll
    for (int i = 0; i < U; ++i) r[i] = *(const float4*)(tile_base + off[i]);
}

template <int BMN, int NT, int U, bool KC, int LD>
__device__ __forceinline__ void gemm_store_operand(const float4 (&r)[U], float* __restrict__ s, int tid) {
#pragma unroll
    for (int i = 0; i < U; ++i) {
        const int u = tid + NT * i;
        if (KC) {
            const int row = u >> 2, c4 = u & 3;
            float* d = s + row * LD + 4 * c4;
            d[0] = r[i].x;
            d[1] = r[i].y;
            d[2] = r[i].z;
            d[3] = r[i].w;
        } else {
            const int kk = u / (BMN / 4), c4 = u - kk * (BMN / 4);
            // rebuilt from components: copying the array element as a whole demotes the register array to scratch
            *(float4*)(s + kk * BMN + 4 * c4) = make_float4(r[i].x, r[i].y, r[i].z, r[i].w);
        }
    }
}

// MFMA operand fragments of one k-pair: lane (l31, hh) supplies A[i = l31][k = hh] and B[k = hh][j = l31].
template <int BM, int BN, bool A_KC, bool B_KC, int LDA, int LDB, int TM>
__device__ __forceinline__ void gemm_read_frags(float (&a)[TM], float (&b)[2], const float* __restrict__ sA,
                                                const float* __restrict__ sB, int arow, int bcol, int kk) {
#pragma unroll
    for (int m = 0; m < TM; ++m) a[m] = A_KC ? sA[(arow + m * 32) * LDA + kk] : sA[kk * BM + arow + m * 32];
#pragma unroll
    for (int n = 0; n < 2; ++n) b[n] = B_KC ? sB[(bcol + n * 32) * LDB + kk] : sB[kk * BN + bcol + n * 32];
}

template <int TM>
__device__ __forceinline__ void gemm_mma8(f32x16 (&acc)[TM][2], const float (&a)[TM], const float (&b)[2]) {
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], b[n], acc[m][n], 0, 0, 0);
}

// EPI_UPDW: the W update, unit-L2 atom normalisation and the K-vectors of the next iteration, fused behind U = R.H^T
// (gccNMFFunctions.py:77,79-80).  The <4,1> workgroup owns all F rows of its 64 atoms (512 on the matrix cores + the
// VALU tail row), so the column norms are a workgroup-local reduction: lanes -> row halves (shuffle) -> waves (LDS).
//   Wt = W * (U / rowsumH);  norm = sqrt(sum_f Wt^2);  W = Wt / norm;  colsum = sum_f W;  hscale = norm
template <bool TAIL>
__device__ __forceinline__ void gemm_epilogue_update_w(const GemmArgs& p, int file, int col0, int tid, int wm, int l31, int hh,
                                                       f32x16 (&acc)[4][2], float tail_acc, float rowsum_acc, float* smem) {
    float* s_rs = smem;              // [64]     rowsum of H per atom
    float* s_tail = smem + 64;       // [4][64]  partial dot products of the tail row
    float* s_red = smem + 320;       // [5][64]  per-wave (+ tail row) partial column reductions
    float* s_norm = smem + 640;      // [64]
    float rs = rowsum_acc;
    rs += __shfl_xor(rs, 1);
    rs += __shfl_xor(rs, 2);
    if ((tid & 3) == 0) s_rs[tid >> 2] = rs;
    if (TAIL) s_tail[tid] = tail_acc;
    __syncthreads();
    const int ca = l31, cb = l31 + 32;
    const int ka = col0 + ca, kb = col0 + cb;
    const bool oka = ka < p.N, okb = kb < p.N;
    const int kac = min(ka, p.N - 1), kbc = min(kb, p.N - 1);
    const float rsa = s_rs[ca], rsb = s_rs[cb];
    float* W = p.C + file * p.sC;
    float ssa = 0.f, ssb = 0.f;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int rbase = wm * 128 + m * 32 + 4 * hh;
        float wa[16], wb[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long ro = (long)min(rbase + (r & 3) + 8 * (r >> 2), p.M - 1) * p.ldc;
            wa[r] = W[ro + kac];
            wb[r] = W[ro + kbc];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const bool valid = (rbase + (r & 3) + 8 * (r >> 2)) < p.M;
            const float ta = (valid && oka) ? wa[r] * (acc[m][0][r] / rsa) : 0.f;
            const float tb = (valid && okb) ? wb[r] * (acc[m][1][r] / rsb) : 0.f;
            acc[m][0][r] = ta;
            acc[m][1][r] = tb;
            ssa = fmaf(ta, ta, ssa);
            ssb = fmaf(tb, tb, ssb);
        }
    }
    ssa += __shfl_xor(ssa, 32);
    ssb += __shfl_xor(ssb, 32);
    if (hh == 0) {
        s_red[wm * 64 + ca] = ssa;
        s_red[wm * 64 + cb] = ssb;
    }
    float wt_tail = 0.f;
    const bool tail_ok = TAIL && tid < 64 && (col0 + tid) < p.N;
    if (tid < 64) {
        if (tail_ok) {
            const float u = (s_tail[tid] + s_tail[64 + tid]) + (s_tail[128 + tid] + s_tail[192 + tid]);
            wt_tail = W[(long)p.tail_row * p.ldc + col0 + tid] * (u / s_rs[tid]);
        }
        s_red[256 + tid] = wt_tail * wt_tail;
    }
    __syncthreads();
    if (tid < 64) s_norm[tid] = sqrtf(((s_red[tid] + s_red[64 + tid]) + (s_red[128 + tid] + s_red[192 + tid])) + s_red[256 + tid]);
    __syncthreads();
    const float na = s_norm[ca], nb = s_norm[cb];
    float csa = 0.f, csb = 0.f;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int rbase = wm * 128 + m * 32 + 4 * hh;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = rbase + (r & 3) + 8 * (r >> 2);
            if (row < p.M) {
                if (oka) {
                    const float wn = acc[m][0][r] / na;
                    W[(long)row * p.ldc + ka] = wn;
                    csa += wn;
                }
                if (okb) {
                    const float wn = acc[m][1][r] / nb;
                    W[(long)row * p.ldc + kb] = wn;
                    csb += wn;
                }
            }
        }
    }
    csa += __shfl_xor(csa, 32);
    csb += __shfl_xor(csb, 32);
    if (hh == 0) {
        s_red[wm * 64 + ca] = csa;
        s_red[wm * 64 + cb] = csb;
    }
    if (tid < 64) {
        float wn_tail = 0.f;
        if (tail_ok) {
            wn_tail = wt_tail / s_norm[tid];
            W[(long)p.tail_row * p.ldc + col0 + tid] = wn_tail;
        }
        s_red[256 + tid] = wn_tail;
    }
    __syncthreads();
    if (tid < 64 && (col0 + tid) < p.N) {
        p.out_colsum[file * p.s_out + col0 + tid] = ((s_red[tid] + s_red[64 + tid]) + (s_red[128 + tid] + s_red[192 + tid])) + s_red[256 + tid];
        p.out_norm[file * p.s_out + col0 + tid] = s_norm[tid];
    }
}

template <int WM, int WN, bool A_KC, bool B_KC, int EPI, bool TAIL, int TM = 4>
__global__ __launch_bounds__(WM* WN * 64, 2) void gccnmf_gemm_kernel(GemmArgs p) {
    constexpr int BK = 16;
    constexpr int WT = 32 * TM;             // output rows per wave (TM = 4: 128 x 64 per wave; TM = 1: the small-batch tile)
    constexpr int BM = WM * WT, BN = WN * 64;
    constexpr int NT = WM * WN * 64;
    constexpr int LDA = A_KC ? (BK + 1) : BM;
    constexpr int LDB = B_KC ? (BK + 1) : BN;
    constexpr int SA = A_KC ? BM * LDA : BK * BM;
    constexpr int SB = B_KC ? BN * LDB : BK * BN;
    constexpr int SBUF = SA + SB + BK;      // one staging buffer: A tile | B tile | tail row of A
    constexpr int UA = BM * 4 / NT;         // float4 units of the A tile per thread
    constexpr int UB = BN * 4 / NT;
    static_assert(UA >= 1 && UB >= 1 && UA * NT == BM * 4 && UB * NT == BN * 4, "tile/thread mismatch");
    static_assert(!TAIL || A_KC, "the VALU tail row needs a reduction-contiguous A");
    static_assert(NT % BN == 0 || BN % NT == 0, "tail mapping");
    static_assert(SBUF % 4 == 0 && SA % 4 == 0 && SB % 4 == 0, "16-byte aligned LDS carve");

    // two staging buffers: tile t is computed from buffer t&1 while tile t+1 is written into the other one
    __shared__ __attribute__((aligned(16))) float smem[2 * SBUF];

    // ---- which (file, tile) is this workgroup? ------------------------------------------
    const int tiles = p.tiles_m * p.tiles_n;
    int file, tile;
    if (p.xcd_affine) {
        // blocks b, b+8, b+16, ... land on XCD b%8 (observed dispatch order): keep every tile of
        // a file on one XCD so its W/H/R panels are shared through that XCD's L2.
        // XCD x owns the contiguous chunk [x * chunk, (x+1) * chunk) of the tile list (chunk = xcd_affine = ceil(total / 8)) taken in
        // the file order 0, 8, 16, ... | 1, 9, ... : balanced to one tile whatever the batch (25 files used to put 4 files = 80 tiles
        // on XCD 0 and 3 on the others), a file still sits on one XCD (or straddles two neighbours), and for a multiple of 8 files it
        // is exactly the map of rounds 1-2 (XCD x <- files x, x+8, ...), which is 3 % faster at 64 files than contiguous files per
        // XCD (K3 0.645 vs 0.668 ms, A/B on one box: profiles/r03_ab_xcd_map.txt)
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        const int idx = xcd * p.xcd_affine + slot;
        if (idx >= p.batch * tiles) return;
        // position q in the file order 0, 8, 16, ... | 1, 9, 17, ... | ... (files of one residue class mod 8 are neighbours)
        const int q = idx / tiles;
        tile = idx - q * tiles;
        const int n_full = p.batch >> 3, rem = p.batch & 7, big = rem * (n_full + 1);
        const int cls = q < big ? q / (n_full + 1) : rem + (q - big) / n_full;
        file = cls + 8 * (q < big ? q - cls * (n_full + 1) : (q - big) - (cls - rem) * n_full);
    } else {
        file = blockIdx.x / tiles;
        tile = blockIdx.x - file * tiles;
    }
    // the integer divisions run on the VALU; pin their (wave-uniform) results to SGPRs so that the operand / output base
    // pointers derived from them are scalar arithmetic instead of per-lane 64-bit VALU adds
    file = __builtin_amdgcn_readfirstlane(file);
    const int tm = __builtin_amdgcn_readfirstlane(tile / p.tiles_n), tn = __builtin_amdgcn_readfirstlane(tile) - tm * p.tiles_n;
    const int row0 = tm * BM, col0 = tn * BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave - wm * WN;
    const int l31 = lane & 31, hh = lane >> 5;
    const int arow = wm * WT + l31, bcol = wn * 64 + l31;

    const float* __restrict__ A = p.A + file * p.sA;
    const float* __restrict__ B = p.B + file * p.sB;
    const float* __restrict__ bscale = (!B_KC && p.bscale) ? p.bscale + file * p.s_bscale : nullptr;

    const bool wave_active = (row0 + wm * WT) < p.M;
    const bool do_tail = TAIL && (tm == 0) && !(p.ablate & 8);
    const bool do_rowsum = B_KC && (p.rowsumB != nullptr || EPI == EPI_UPDW) && (tm == 0);

    f32x16 acc[TM][2];
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    float tail_acc = 0.f, rowsum_acc = 0.f;

    int offA[UA], offB[UB];
    gemm_operand_offsets<BM, NT, UA, A_KC>(offA, p.lda, row0, p.a_clamp, tid);
    gemm_operand_offsets<BN, NT, UB, B_KC>(offB, p.ldb, col0, p.b_clamp, tid);
    float4 ra[UA], rb[UB];
    float4 rt = make_float4(0.f, 0.f, 0.f, 0.f);
    float rsc[UB];
#pragma unroll
    for (int i = 0; i < UB; ++i) rsc[i] = 1.f;

#define GEMM_LOAD_TILE(k0_)                                                                                     \
    do {                                                                                                        \
        gemm_load_operand<UA>(ra, A + (A_KC ? (long)(k0_) : (long)(k0_) * p.lda), offA);                        \
        gemm_load_operand<UB>(rb, B + (B_KC ? (long)(k0_) : (long)(k0_) * p.ldb), offB);                        \
        if (!B_KC) {                                                                                            \
            if (bscale) {                                                                                       \
                _Pragma("unroll") for (int i_ = 0; i_ < UB; ++i_) rsc[i_] = bscale[(k0_) + (tid + NT * i_) / (BN / 4)]; \
            }                                                                                                   \
        }                                                                                                       \
        if (TAIL) {                                                                                             \
            if (do_tail && tid < 4) rt = *(const float4*)(A + (long)p.tail_row * p.lda + (k0_) + 4 * tid);      \
        }                                                                                                       \
    } while (0)

#define GEMM_STORE_TILE(buf_)                                                                                   \
    do {                                                                                                        \
        float* sbuf_ = smem + (buf_) * SBUF;                                                                    \
        gemm_store_operand<BM, NT, UA, A_KC, LDA>(ra, sbuf_, tid);                                              \
        if (!B_KC) {                                                                                            \
            if (bscale) {   /* the lazy H row scale rides on the staging pass (the wait on it happens here) */  \
                _Pragma("unroll") for (int i_ = 0; i_ < UB; ++i_) {                                             \
                    rb[i_].x *= rsc[i_];                                                                        \
                    rb[i_].y *= rsc[i_];                                                                        \
                    rb[i_].z *= rsc[i_];                                                                        \
                    rb[i_].w *= rsc[i_];                                                                        \
                }                                                                                               \
            }                                                                                                   \
        }                                                                                                       \
        gemm_store_operand<BN, NT, UB, B_KC, LDB>(rb, sbuf_ + SA, tid);                                         \
        if (TAIL) {                                                                                             \
            if (do_tail && tid < 4) *(float4*)(sbuf_ + SA + SB + 4 * tid) = rt;                                 \
        }                                                                                                       \
    } while (0)

    const int nkt = (p.Kd + BK - 1) / BK;
    GEMM_LOAD_TILE(0);
    GEMM_STORE_TILE(0);
    if (nkt > 1) GEMM_LOAD_TILE(BK);
    __syncthreads();

    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        // tile kt+1 (in registers since the previous iteration) -> the other buffer; tile kt+2 -> registers.
        // One barrier per tile: buffer cur^1 was last read in iteration kt-1, which every wave has left.
        if (kt + 1 < nkt && !(p.ablate & 2)) GEMM_STORE_TILE(cur ^ 1);
        if (kt + 2 < nkt && !(p.ablate & 1)) GEMM_LOAD_TILE((kt + 2) * BK);

        const float* __restrict__ sA = smem + cur * SBUF;
        const float* __restrict__ sB = sA + SA;
        const float* __restrict__ sT = sB + SB;
        if (wave_active) {
            // software-pipelined fragments: the reads of k-pair p+1 are in flight under the 8 MFMAs of k-pair p
            float a0[TM], b0[2], a1[TM], b1[2];
            gemm_read_frags<BM, BN, A_KC, B_KC, LDA, LDB, TM>(a0, b0, sA, sB, arow, bcol, hh);
            GEMM_PRIO(1);
#pragma unroll
            for (int pp = 0; pp < BK / 2; pp += 2) {
                gemm_read_frags<BM, BN, A_KC, B_KC, LDA, LDB, TM>(a1, b1, sA, sB, arow, bcol, 2 * (pp + 1) + hh);
                GEMM_PIN();
                gemm_mma8<TM>(acc, a0, b0);
                GEMM_PIN();
                if (pp + 2 < BK / 2) gemm_read_frags<BM, BN, A_KC, B_KC, LDA, LDB, TM>(a0, b0, sA, sB, arow, bcol, 2 * (pp + 2) + hh);
                GEMM_PIN();
                gemm_mma8<TM>(acc, a1, b1);
                GEMM_PIN();
            }
            GEMM_PRIO(0);
        }
        if (TAIL) {
            if (do_tail) {
                // NT/BN thread groups split the 16 reduction steps of the extra output row
                constexpr int G = (NT >= BN) ? NT / BN : 1;
                constexpr int PER = BK / G;
                const int j = tid % BN, g = tid / BN;
#pragma unroll
                for (int e = 0; e < PER; ++e) {
                    const int kk = g * PER + e;
                    const float bv = B_KC ? sB[j * LDB + kk] : sB[kk * BN + j];
                    tail_acc = fmaf(sT[kk], bv, tail_acc);
                }
            }
        }
        if (B_KC) {
            if (do_rowsum) {
                constexpr int PERT = (NT >= BN) ? NT / BN : 1;   // threads per atom row
                constexpr int CNT = BK / PERT;
                const int j = tid / PERT, q = tid - j * PERT;
                if (j < BN) {
#pragma unroll
                    for (int e = 0; e < CNT; ++e) rowsum_acc += sB[j * LDB + q * CNT + e];
                }
            }
        }
        if (!(p.ablate & 4)) __syncthreads();   // ablate: timing experiments only
    }
#undef GEMM_LOAD_TILE
#undef GEMM_STORE_TILE

    // ---- epilogue: MFMA C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5) ---------
    if (EPI == EPI_UPDW) {
        static_assert(EPI != EPI_UPDW || (WM == 4 && WN == 1 && TM == 4 && B_KC), "the fused W update needs the tall tile that owns every row");
        if constexpr (TM == 4) gemm_epilogue_update_w<TAIL>(p, file, col0, tid, wm, l31, hh, acc, tail_acc, rowsum_acc, smem);
        return;
    }
    if (p.ablate & 16) {   // timing experiment: keep the accumulators alive, store one value per lane
        float keep = 0.f;
#pragma unroll
        for (int m = 0; m < TM; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) keep += acc[m][n][r];
        if (wave_active && keep == 123.456f) p.C[file * p.sC] = keep;
    } else if (wave_active) {
#pragma unroll
        for (int m = 0; m < TM; ++m)
            gemm_epilogue_pair<EPI>(p, file, row0 + wm * WT + m * 32 + 4 * hh, col0 + wn * 64 + l31, acc[m][0], acc[m][1]);
    }
    if (TAIL) {
        if (do_tail) {   // block-uniform
            constexpr int G = (NT >= BN) ? NT / BN : 1;
            smem[tid] = tail_acc;     // the staging buffers are free after the final barrier of the k loop
            __syncthreads();
            if (tid < BN) {
                float s = 0.f;
#pragma unroll
                for (int g = 0; g < G; ++g) s += smem[g * BN + tid];
                const int col = col0 + tid;
                if (gemm_col_valid<EPI>(p, col)) gemm_epilogue<EPI>(p, file, p.tail_row, col, s);
            }
        }
    }
    if (B_KC) {
        if (do_rowsum) {
            constexpr int PERT = (NT >= BN) ? NT / BN : 1;
            float s = rowsum_acc;
            if (PERT >= 2) s += __shfl_xor(s, 1);
            if (PERT >= 4) s += __shfl_xor(s, 2);
            const int j = tid / PERT, q = tid - j * PERT;
            if (q == 0 && j < BN && (col0 + j) < p.N) p.rowsumB[file * p.s_rowsumB + col0 + j] = s;
        }
    }
}

template <int WM, int WN, bool A_KC, bool B_KC, int EPI, bool TAIL, int TM = 4>
static int gccnmf_launch_gemm(GemmArgs a, hipStream_t stream) {
    constexpr int BM = WM * 32 * TM, BN = WN * 64, NT = WM * WN * 64;
    if (!a.A || !a.B || !a.C || a.M < 1 || a.N < 1 || a.Kd < 1 || a.batch < 1) return GCCNMF_ERR_ARG;
    if ((a.lda & 3) || (a.ldb & 3)) return GCCNMF_ERR_ARG;   // float4 staging
    a.ablate = gccnmf_tune_ablate;
    a.tiles_m = gccnmf_ceil_div(a.M, BM);
    a.tiles_n = gccnmf_ceil_div(a.N, BN);
    const int tiles = a.tiles_m * a.tiles_n;
    int grid;
    if (a.xcd_affine && a.batch >= 8) {
        a.xcd_affine = gccnmf_ceil_div(a.batch * tiles, 8);     // tiles per XCD
        grid = 8 * a.xcd_affine;
    } else {
        a.xcd_affine = 0;
        grid = a.batch * tiles;
    }
    hipLaunchKernelGGL((gccnmf_gemm_kernel<WM, WN, A_KC, B_KC, EPI, TAIL, TM>), dim3(grid), dim3(NT), 0, stream, a);
    GCCNMF_CHECK_LAUNCH();
    return GCCNMF_OK;
}
