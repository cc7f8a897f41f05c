// Latency-path GEMM (one mixture alone, a handful of files): interface of csrc/direct.hip.
//
//   C[m][n] (+ fused element-wise work) = sum_r A[r][m] * B[r][n]          r < Kd, m < M, n < N
//
// BOTH operands are reduction-major ("r-major": one reduction index per row, the output index contiguous).  That is the layout in
// which a 16x16x4 MFMA operand can be fetched from global memory straight into registers with fully coalesced 16-byte loads
// (a lane group of 16 reads 256 contiguous bytes of one reduction row), so the kernel needs no LDS staging, no LDS-DMA and no
// barrier in its main loop.  The KL-NMF driver keeps transposed copies (Wt, Ht, Rt) so that each of its four GEMMs has this form.
#pragma once
#include "common.h"
#include "../../include/gccnmf_hip.h"

enum DirectEpilogue {
    DEPI_STORE = 0,   // C = acc                                             (U = R.H^T; + row sums of B, + tail row)
    DEPI_DIV = 1,     // C = E0 / acc                                        (R = V / (W.H), gccNMFFunctions.py:76 inner)
    DEPI_DIVT = 2,    // Ct[n][m] = E0[m][n] / acc  (transposed output only; the tail row goes to C and Ct)    (:77 inner)
    DEPI_UPDH = 3     // C = (C*E1[m]) * ((acc + ktailA[m]*ktailB[n]) / (E2[m] + alpha + eps)), also stored transposed into Ct   (:76)
};

// the argument block is the public descriptor (include/gccnmf_hip.h: gccnmf_direct_gemm documents every field)
typedef gccnmf_direct_gemm DirectArgs;

// epi: DirectEpilogue.  tile: 0 = chosen by the cost model, 1.. = index into the tile table (experiments).  Returns a GCCNMF_* status.
int gccnmf_direct_launch(DirectArgs a, int epi, int tile, hipStream_t stream);

// K1 + K2 of one iteration in one launch for short dictionaries (Kd <= 128, M = F - 1 <= 512 a multiple of 64; bin M rides on the VALU):
//   H <- (scale*H) * (W^T . (V / (W . (scale*H)))) / (colsum + alpha + eps), R never written (workgroup = 64-frame column tile of a file,
//   its scaled H resident in registers, W in 64-bin chunks through LDS)
struct WhUpdhArgs {
    const float *W, *V, *scale, *colsum;
    float* H;                   // updated in place
    long sW, sH, sV, sVec;      // per-file strides (floats)
    int lda, ldb, ldv;          // row pitches of W, H, V
    int M, N, Kd, batch;
    float alpha, eps;
    int tiles_n, xc;            // filled in by the launcher
    long long* trace;
};
int gccnmf_wh_updh_launch(WhUpdhArgs a, hipStream_t stream);

// K3 + K4a of one iteration in one launch for short dictionaries (Kd <= 128, M = F - 1 <= 512 a multiple of 64; bin M on the VALU):
//   U = (V / (W . H)) . H^T and rowsumH = sum_n H, R never written (workgroup = 64-bin slab of a file, W rows resident in registers)
struct WhdivRhtArgs {
    const float *W, *H, *V;
    float *U, *rowsumH;
    long sW, sH, sV, sU, sVec;  // per-file strides (floats)
    int lda, ldb, ldv, ldu;     // row pitches of W, H, V, U
    int M, N, Kd, batch;
    int nslabs, xc;             // filled in by the launcher
    long long* trace;
};
int gccnmf_whdiv_rht_launch(WhdivRhtArgs a, hipStream_t stream);

// out[c][r] = in[r][c] for r < rows, c < cols (batched; strides in floats) -- the transposed copies the direct path starts from
int gccnmf_transpose_launch(const float* in, long s_in, int ld_in, float* out, long s_out, int ld_out, int rows, int cols, int batch,
                            hipStream_t stream);


// ---- the short-dictionary iteration as ONE chained launch (round 6) --------------------------------------------------------------
// K1 + K2 (column tiles) | K3 + K4a (bin slabs) | the one-pass W update of EVERY iteration of a gccnmf_klnmf call in one kernel: per XCD
// the items of list x = files x, x + 8, ... whole, stage by stage, iteration by iteration; the stages hand over per FILE through ready
// counters (chain_sync.h): a slab needs all column tiles of its file, a W-update group all slabs, the next iteration's column tiles
// all W-update groups.  Same item programs as the three plain launches: same bits.
struct UpdateWArgs {
    float* W;
    const float *U, *rowsumH;
    float *colsumW, *hscale;
    int F, K, Kp;
    long sW, sU, sVec, sRowsum;
};
struct ShortChainArgs {
    WhUpdhArgs a12;
    WhdivRhtArgs a34;
    UpdateWArgs aw;
    int per_file[3];             // items per file of the three stages: column tiles, slabs, atom groups
    int first[4];                // per list: stage s of an iteration serves positions [first[s], first[s + 1]) -- stage-major, whole files (a
                                 // two-group order with the groups half an iteration apart, one group's slabs beside the other's column tiles, was
                                 // measured slower: 0.382 against 0.376 ms per iteration at K = 128, LABBOOK R6.4)
    int it0, iterations, atoms_per_group, solo;
    unsigned* counters;          // [3][batch] ready counters (zeroed by the caller), then the error flag / XCC table block of the throughput chain
    unsigned* error;
    unsigned* xcc_seen;
    long long timeout;           // GemmSync.timeout of every stage
    long long* trace;
    int trace_rows, trace_it;
};
int gccnmf_short_chain_launch(ShortChainArgs a, hipStream_t stream);
