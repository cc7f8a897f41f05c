// Direct-to-LDS variant of the throughput GEMM tile (512 x 64 per workgroup, 128 x 64 per wave):
// operand tiles go global -> LDS by `global_load_lds_dwordx4` (LDS-DMA: no VGPR staging, no ds_write pass), and the
// reduction-contiguous operands are read back as ds_read_b128 fragments.
//
// LDS-DMA writes lane-linear (wave-uniform base + lane*16 B), so the LDS image of a reduction-contiguous operand is the
// unpadded [row][16] tile and bank conflicts are avoided by permuting which 16-byte chunk each lane FETCHES:
//     LDS chunk c' of row r holds global chunk  c = c' ^ ((r >> 2) & 3)
// (source-side XOR swizzle + the same swizzle on the read: for every ds_read_b128 lane group the 16 lanes then touch 16
// distinct 16-byte slots).  The k order inside a 16-deep tile is permuted to make one float4 feed four MFMA steps: step
// (q, e), q = 0..1, e = 0..3, multiplies k = 8q + e in lanes 0-31 with k = 8q + 4 + e in lanes 32-63; A and B use the same
// assignment, so only the (irrelevant, exact-f32) summation order inside the tile changes.
//
// The lazy H row scale of K1 is applied to the B fragments after the LDS read (fl(H*s) exactly as before).
// Epilogues, the VALU tail row and the row sums are those of gemm_mfma.h.
#pragma once
#include <type_traits>
#include "gemm_mfma.h"

typedef __attribute__((address_space(3))) void* gemm_lds_ptr;
typedef __attribute__((address_space(1))) const void* gemm_glb_ptr;

__device__ __forceinline__ void gemm_dma16(const float* src, float* lds_wave_base) {
    // lane l lands at lds_wave_base + 16*l bytes; lds_wave_base must be wave-uniform
    __builtin_amdgcn_global_load_lds((gemm_glb_ptr)src, (gemm_lds_ptr)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ int gemm_swz(int row) { return (row >> 2) & 3; }

// MFMA fragment reads as inline asm: the compiler neither sees them as LDS reads (so it does not drain the LDS-DMA queue
// in front of them) nor waits for them -- every use is preceded by a hand-placed s_waitcnt lgkmcnt + sched_barrier.
typedef float gemm_f32x4 __attribute__((ext_vector_type(4)));
template <int OFF>
__device__ __forceinline__ gemm_f32x4 gemm_lds_read_b128(unsigned byte_addr) {
    gemm_f32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(byte_addr), "n"(OFF));
    return v;
}
// two dwords, OFF0/OFF1 in dword units (< 256) from the same base
typedef float gemm_f32x2 __attribute__((ext_vector_type(2)));
template <int OFF0, int OFF1>
__device__ __forceinline__ gemm_f32x2 gemm_lds_read2_b32(unsigned byte_addr) {
    gemm_f32x2 v;
    asm volatile("ds_read2_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(v) : "v"(byte_addr), "n"(OFF0), "n"(OFF1));
    return v;
}

// all fragments of one k half (q) of a tile; every address is ONE per-operand base VGPR + an immediate offset
template <bool A_KC, bool B_KC, int BM, int BN, int TM>
__device__ __forceinline__ void gemm_dma_read_frags(gemm_f32x4 (&av)[TM], gemm_f32x4 (&bv)[2], unsigned baseA, unsigned baseB) {
    // KC:     base = lds + 4*(row0*16 + 4*(chunk ^ swz(row0))); rows row0 + 32*m share swz -> offset m*2048
    // non-KC: base = lds + 4*(4*chunk*B + row0);                element e of row m at offset 4*(e*B + 32*m)
    if (A_KC) {
        av[0] = gemm_lds_read_b128<0 * 2048>(baseA);
        av[1] = gemm_lds_read_b128<1 * 2048>(baseA);
        av[2] = gemm_lds_read_b128<2 * 2048>(baseA);
        av[3] = gemm_lds_read_b128<3 * 2048>(baseA);
    } else {
        // element e of rows m, m+1 are 32 dwords apart: one ds_read2_b32 each; the e stride (BM dwords) goes into the base
#define GEMM_RD_A(e_, f_)                                                                        \
    {                                                                                            \
        const gemm_f32x2 lo = gemm_lds_read2_b32<0, 32>(baseA + 4 * (e_) * BM);                  \
        const gemm_f32x2 hi = gemm_lds_read2_b32<64, 96>(baseA + 4 * (e_) * BM);                 \
        av[0].f_ = lo.x; av[1].f_ = lo.y; av[2].f_ = hi.x; av[3].f_ = hi.y;                      \
    }
        GEMM_RD_A(0, x) GEMM_RD_A(1, y) GEMM_RD_A(2, z) GEMM_RD_A(3, w)
#undef GEMM_RD_A
    }
    if (B_KC) {
        bv[0] = gemm_lds_read_b128<0>(baseB);
        bv[1] = gemm_lds_read_b128<2048>(baseB);
    } else {
#define GEMM_RD_B(e_, f_)                                                                        \
    {                                                                                            \
        const gemm_f32x2 v2 = gemm_lds_read2_b32<0, 32>(baseB + 4 * (e_) * BN);                  \
        bv[0].f_ = v2.x; bv[1].f_ = v2.y;                                                        \
    }
        GEMM_RD_B(0, x) GEMM_RD_B(1, y) GEMM_RD_B(2, z) GEMM_RD_B(3, w)
#undef GEMM_RD_B
    }
}

template <bool A_KC, bool B_KC, int EPI, bool TAIL>
__global__ __launch_bounds__(256, 2) void gccnmf_gemm_dma_kernel(GemmArgs p) {
    constexpr int BK = 16, TM = 4, BM = 512, BN = 64;
    constexpr int SA = BM * BK, SB = BN * BK;
    constexpr int SBUF = SA + SB + BK + BK;          // A | B | A tail-row chunk | B row-scale chunk
    __shared__ __attribute__((aligned(16))) float smem[2 * SBUF];

    const int tiles = p.tiles_m * p.tiles_n;
    int file, tile;
    if (p.xcd_affine) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        file = xcd + 8 * (slot / tiles);
        tile = slot % tiles;
        if (file >= p.batch) return;
    } else {
        file = blockIdx.x / tiles;
        tile = blockIdx.x - file * tiles;
    }
    const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
    const int row0 = tm * BM, col0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform in an SGPR: LDS-DMA bases derive from it
    const int wm = wave, wn = 0;
    const int l31 = lane & 31, hh = lane >> 5;
    const int arow = wm * 128 + l31, bcol = wn * 64 + l31;

    const float* __restrict__ A = p.A + file * p.sA;
    const float* __restrict__ B = p.B + file * p.sB;
    const float* __restrict__ bscale = (!B_KC && p.bscale) ? p.bscale + file * p.s_bscale : nullptr;
    const bool wave_active = (row0 + wm * 128) < p.M;
    const bool do_tail = TAIL && (tm == 0);
    const bool do_rowsum = B_KC && (p.rowsumB != nullptr || EPI == EPI_UPDW) && (tm == 0);

    // per-lane source offsets (elements) of this wave's DMA pieces: 8 of A, 1 of B
    int offA[8], offB;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int piece = wave * 8 + i;                                // 1 KB pieces of the A tile
        if (A_KC) {
            const int row = piece * 16 + (lane >> 2), c = (lane & 3) ^ gemm_swz(piece * 16 + (lane >> 2));
            offA[i] = min(row0 + row, p.a_clamp) * p.lda + 4 * c;
        } else {
            const int kk = piece >> 1, col = (piece & 1) * 256 + lane * 4;
            offA[i] = kk * p.lda + min(row0 + col, p.a_clamp);
        }
    }
    if (B_KC) {
        const int row = wave * 16 + (lane >> 2), c = (lane & 3) ^ gemm_swz(row);
        offB = min(col0 + row, p.b_clamp) * p.ldb + 4 * c;
    } else {
        const int kk = wave * 4 + (lane >> 4), col = (lane & 15) * 4;
        offB = kk * p.ldb + min(col0 + col, p.b_clamp);
    }

    f32x16 acc[TM][2];
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    float tail_acc = 0.f, rowsum_acc = 0.f;

#define GEMM_DMA_TILE(kt_, buf_)                                                                               \
    do {                                                                                                       \
        float* sb_ = smem + (buf_) * SBUF;                                                                     \
        const float* At_ = A + (A_KC ? (long)(kt_) * BK : (long)(kt_) * BK * p.lda);                           \
        const float* Bt_ = B + (B_KC ? (long)(kt_) * BK : (long)(kt_) * BK * p.ldb);                           \
        _Pragma("unroll") for (int i_ = 0; i_ < 8; ++i_) gemm_dma16(At_ + offA[i_], sb_ + (wave * 8 + i_) * 256); \
        gemm_dma16(Bt_ + offB, sb_ + SA + wave * 256);                                                         \
        if (TAIL) {                                                                                            \
            if (do_tail && wave == 0 && lane < 4) gemm_dma16(A + (long)p.tail_row * p.lda + (kt_) * BK + 4 * lane, sb_ + SA + SB); \
        }                                                                                                      \
        if (!B_KC) {                                                                                           \
            if (bscale && wave == 1 && lane < 4) gemm_dma16(bscale + (kt_) * BK + 4 * lane, sb_ + SA + SB + BK); \
        }                                                                                                      \
    } while (0)

    const int nkt = (p.Kd + BK - 1) / BK;
    GEMM_DMA_TILE(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // One k-tile.  CUR (which staging buffer holds the tile) is a COMPILE-TIME constant: with a run-time buffer index the
    // compiler cannot prove that the LDS-DMA destination (the other buffer) does not alias this step's ds_reads and
    // drains the DMA queue (s_waitcnt vmcnt(0)) before the first fragment read, serialising copy and compute.
    auto step = [&](auto cur_c, const int kt) {
        constexpr int CUR = decltype(cur_c)::value;
        const float* __restrict__ sA = smem + CUR * SBUF;
        const float* __restrict__ sB = sA + SA;
        const float* __restrict__ sT = sB + SB;
        const float* __restrict__ sS = sT + BK;

        // hipcc drains the LDS-DMA queue (s_waitcnt vmcnt(0)) in front of ANY compiler-visible ds_read that follows a
        // global_load_lds, which would serialise copy and compute.  So: the few reads the compiler sees (tail row, row
        // sums, row scale) go first, and the MFMA fragments are read with inline-asm ds_read (waits counted by hand).
        float4 t4 = make_float4(0.f, 0.f, 0.f, 0.f), tb4 = t4, rs4 = t4, sc4[2] = {t4, t4};
        if (TAIL) {
            if (do_tail) {
                const int j = tid & 63, g = tid >> 6;                  // 4 thread groups x 4 reduction steps
                t4 = *(const float4*)(sT + 4 * g);
                if (B_KC) {
                    tb4 = *(const float4*)(sB + j * 16 + 4 * (g ^ gemm_swz(j)));
                } else {
                    tb4.x = sB[(4 * g + 0) * BN + j];
                    tb4.y = sB[(4 * g + 1) * BN + j];
                    tb4.z = sB[(4 * g + 2) * BN + j];
                    tb4.w = sB[(4 * g + 3) * BN + j];
                    if (bscale) {
                        const float4 s4 = *(const float4*)(sS + 4 * g);
                        tb4.x *= s4.x;
                        tb4.y *= s4.y;
                        tb4.z *= s4.z;
                        tb4.w *= s4.w;
                    }
                }
            }
        }
        if (B_KC) {
            if (do_rowsum) rs4 = *(const float4*)(sB + (tid >> 2) * 16 + 4 * (tid & 3));   // 4 threads per atom row, one chunk each
        } else if (bscale) {
            sc4[0] = *(const float4*)(sS + 4 * hh);
            sc4[1] = *(const float4*)(sS + 4 * (2 + hh));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);

        const unsigned ldsA = (unsigned)(size_t)(gemm_lds_ptr)(smem + CUR * SBUF), ldsB = ldsA + 4 * SA;
        gemm_f32x4 av[2][TM], bv[2][2];
        auto read_frags = [&](const int q) {
            const int cq = 2 * q + hh;                                 // this lane half's 16-byte k chunk
            const unsigned bA = A_KC ? ldsA + 4 * (arow * 16 + 4 * (cq ^ gemm_swz(arow))) : ldsA + 4 * (4 * cq * BM + arow);
            const unsigned bB = B_KC ? ldsB + 4 * (bcol * 16 + 4 * (cq ^ gemm_swz(bcol))) : ldsB + 4 * (4 * cq * BN + bcol);
            gemm_dma_read_frags<A_KC, B_KC, BM, BN, TM>(av[q], bv[q], bA, bB);
        };
        auto mma32 = [&](const int q) {
            if (!B_KC) {
                if (bscale) {                                          // lazy H row scale: fl(H * s), as the staged form did
#pragma unroll
                    for (int n = 0; n < 2; ++n) {
                        bv[q][n].x *= sc4[q].x;
                        bv[q][n].y *= sc4[q].y;
                        bv[q][n].z *= sc4[q].z;
                        bv[q][n].w *= sc4[q].w;
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int m = 0; m < TM; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q][m][e], bv[q][n][e], acc[m][n], 0, 0, 0);
        };
        if (wave_active) read_frags(0);
        // tile kt+1 streams into the other buffer (last read in step kt-1, which every wave has left)
        if (kt + 1 < nkt && !(p.ablate & 1)) GEMM_DMA_TILE(kt + 1, CUR ^ 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (wave_active) {
            read_frags(1);                                             // in flight under the 32 MFMAs of the first half
            __builtin_amdgcn_sched_barrier(0);
            mma32(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            mma32(1);
        }
        if (TAIL) {
            if (do_tail) {
                tail_acc = fmaf(t4.x, tb4.x, tail_acc);
                tail_acc = fmaf(t4.y, tb4.y, tail_acc);
                tail_acc = fmaf(t4.z, tb4.z, tail_acc);
                tail_acc = fmaf(t4.w, tb4.w, tail_acc);
            }
        }
        if (B_KC) {
            if (do_rowsum) rowsum_acc += (rs4.x + rs4.y) + (rs4.z + rs4.w);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!(p.ablate & 4)) __syncthreads();
    };
    for (int kt = 0; kt < nkt; kt += 2) {
        step(std::integral_constant<int, 0>{}, kt);
        if (kt + 1 < nkt) step(std::integral_constant<int, 1>{}, kt + 1);
    }
#undef GEMM_DMA_TILE

    if (EPI == EPI_UPDW) {
        gemm_epilogue_update_w<TAIL>(p, file, col0, tid, wm, l31, hh, acc, tail_acc, rowsum_acc, smem);
        return;
    }
    if (wave_active) {
#pragma unroll
        for (int m = 0; m < TM; ++m)
            gemm_epilogue_pair<EPI>(p, file, row0 + wm * 128 + m * 32 + 4 * hh, col0 + wn * 64 + l31, acc[m][0], acc[m][1]);
    }
    if (TAIL) {
        if (do_tail) {
            smem[tid] = tail_acc;
            __syncthreads();
            if (tid < BN) {
                const float s = (smem[tid] + smem[BN + tid]) + (smem[2 * BN + tid] + smem[3 * BN + tid]);
                const int col = col0 + tid;
                if (gemm_col_valid<EPI>(p, col)) gemm_epilogue<EPI>(p, file, p.tail_row, col, s);
            }
        }
    }
    if (B_KC) {
        if (do_rowsum) {
            float s = rowsum_acc;
            s += __shfl_xor(s, 1);
            s += __shfl_xor(s, 2);
            const int j = tid >> 2;
            if ((tid & 3) == 0 && (col0 + j) < p.N) p.rowsumB[file * p.s_rowsumB + col0 + j] = s;
        }
    }
}

template <bool A_KC, bool B_KC, int EPI, bool TAIL>
static int gccnmf_launch_gemm_dma(GemmArgs a, hipStream_t stream) {
    if (!a.A || !a.B || !a.C || a.M < 1 || a.N < 1 || a.Kd < 1 || a.batch < 1) return GCCNMF_ERR_ARG;
    if ((a.lda & 3) || (a.ldb & 3)) return GCCNMF_ERR_ARG;
    a.ablate = gccnmf_tune_ablate;
    a.tiles_m = gccnmf_ceil_div(a.M, 512);
    a.tiles_n = gccnmf_ceil_div(a.N, 64);
    const int tiles = a.tiles_m * a.tiles_n;
    int grid;
    if (a.xcd_affine && a.batch >= 8) {
        a.xcd_affine = 1;
        grid = 8 * gccnmf_ceil_div(a.batch, 8) * tiles;
    } else {
        a.xcd_affine = 0;
        grid = a.batch * tiles;
    }
    hipLaunchKernelGGL((gccnmf_gemm_dma_kernel<A_KC, B_KC, EPI, TAIL>), dim3(grid), dim3(256), 0, stream, a);
    GCCNMF_CHECK_LAUNCH();
    return GCCNMF_OK;
}
