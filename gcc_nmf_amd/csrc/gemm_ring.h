// Small-tile GEMM for launches that cannot fill the chip with throughput tiles (one mixture alone, a handful of files):
// 128 x 64 per workgroup (32 x 64 per wave), operands streamed through an NSTAGE-deep LDS ring by LDS-DMA.
//
// Why a ring: with a few hundred workgroups of 16-80 k-tiles each, a launch lasts as long as ONE workgroup's dependent chain
// of k-tiles.  The register-staged small-batch kernel (gemm_mfma.h, TM = 1) prefetches exactly one k-tile ahead, so every
// k-tile waits for a global round trip (measured: 2870 cycles per k-tile for W^T.R on one file, against 1024 cycles of MFMA
// work).  Here `global_load_lds` needs no staging registers, NSTAGE-3 k-tiles are in flight, and the loop is software-pipelined:
// the fragments of k-tile kt+1 are read (dealt out between the MFMAs) while k-tile kt is multiplied.  The LDS image, the
// source-side XOR swizzle and the k permutation inside a 16-deep tile are those of gemm_dma.h (one float4 of a
// reduction-contiguous operand feeds four MFMA steps).
//
// Everything the NMF launches need rides along as in the other two kernels: the VALU tail row (F = 513 = 16*32 + 1), the lazy
// row scale of the B operand (K1), the row sums of B (K4a), the rank-1 reduction tail of the generic epilogue (K2).  The two
// per-reduction-index vectors (tail row of A, row scale of B) are copied into LDS ONCE per workgroup, so the loop carries no
// side pieces and no wave is special.  What each of these costs in the main loop was measured with build-time ablations
// (scripts/ktrace_single.py; one file, K = 1024, 32 k-tiles of W.H): MFMA-bound 14.9 us; + fragment reads 2.2; + tail-row
// branches in the loop 2.6 (now: two copies of the loop, chosen once per workgroup); + one v_mul in front of every MFMA for
// the row scale 4.8 (now: one packed-multiply burst per k-tile on the look-ahead fragments, under the tail of the MFMAs).
// With these the main loops run at 75-88 % of their MFMA time at the 2.2 GHz the part holds under this load (1160-1360 shader
// clocks per k-tile against 1024).  Two things that did NOT help (measured, removed again): a second wave group per workgroup taking
// alternate k-tiles (two waves per SIMD: same time per k-tile), and padding the reduction-strided LDS images against the 2-way
// bank conflict between the two lane halves of a fragment read (same).
#pragma once
#include <atomic>
#define GCCNMF_MAX_DEVICES 64
#include <type_traits>
#include "gemm_dma.h"

#ifndef RING_EARLY_FINISH
#define RING_EARLY_FINISH 1
#endif

template <int N>
__device__ __forceinline__ void gemm_wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory");
}
// two dwords 64*OFF0 and 64*OFF1 dwords from the base
template <int OFF0, int OFF1>
__device__ __forceinline__ gemm_f32x2 gemm_lds_read2st64_b32(unsigned byte_addr) {
    gemm_f32x2 v;
    asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(v) : "v"(byte_addr), "n"(OFF0), "n"(OFF1));
    return v;
}

template <bool A_KC, bool B_KC, int EPI, bool TAIL, int NSTAGE>
__global__ __launch_bounds__(256, 2) void gccnmf_gemm_ring_kernel(GemmArgs p) {
    constexpr int BK = 16, BM = 128, BN = 64;
    constexpr int SA = BM * BK, SB = BN * BK;
    constexpr int STG = SA + SB;                                   // floats per stage (12 KB)
    constexpr bool SCALE = !B_KC && (EPI == EPI_DIV || EPI == EPI_STORE);
    static_assert(!TAIL || A_KC, "the VALU tail row needs a reduction-contiguous A");
    static_assert(NSTAGE >= 4 && NSTAGE <= 12, "ring depth");
    extern __shared__ __attribute__((aligned(16))) float ring_smem[];     // ring | tail row of A [nkt*16] | row scale of B [nkt*16]

    const int tiles = p.tiles_m * p.tiles_n;
    int file = blockIdx.x / tiles, tile = blockIdx.x - file * tiles;
    file = __builtin_amdgcn_readfirstlane(file);
    const int tm = __builtin_amdgcn_readfirstlane(tile / p.tiles_n), tn = __builtin_amdgcn_readfirstlane(tile) - tm * p.tiles_n;
    const int row0 = tm * BM, col0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    const int arow = wave * 32 + l31, bcol = l31;

    if (p.trace && tid == 0) {      // per-workgroup timeline (gccnmf_debug_set_trace, scripts/ktrace_single.py)
        p.trace[8 * blockIdx.x + 0] = __builtin_amdgcn_s_memrealtime();
        p.trace[8 * blockIdx.x + 4] = (long long)((__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u) << 16 | (__builtin_amdgcn_s_getreg((31 << 11) | 4) & 0xffffu));
    }
    // split-K (kparts > 0): "file" is the part index; all parts read the same operands, each its own balanced range of k-tiles
    // [kb, kb + nkt) (the first (total % kparts) parts take one more), and writes its own partial output (C + part * sC)
    const bool split = p.kparts > 0;
    int kb = 0, nkt = (p.Kd + BK - 1) / BK;
    if (split) {
        const int q = nkt / p.kparts, r = nkt - q * p.kparts;
        kb = file * q + min(file, r);
        nkt = q + (file < r ? 1 : 0);
    }
    kb = __builtin_amdgcn_readfirstlane(kb);
    nkt = __builtin_amdgcn_readfirstlane(nkt);
    const float* __restrict__ A = p.A + (split ? 0 : file * p.sA) + (A_KC ? (long)kb * BK : (long)kb * BK * p.lda);
    const float* __restrict__ B = p.B + (split ? 0 : file * p.sB) + (B_KC ? (long)kb * BK : (long)kb * BK * p.ldb);
    const float* __restrict__ bscale = (SCALE && p.bscale) ? p.bscale + (split ? 0 : file * p.s_bscale) + kb * BK : nullptr;
    const bool side_wg = (tm == 0);                                // the row-0 workgroups carry the tail row and the row sums
    const bool do_rowsum = B_KC && (p.rowsumB != nullptr) && side_wg;
    const int nside = (nkt * BK + 255) & ~255;                     // floats per reduction-index vector in LDS (whole 1 KB pieces)

    // per-lane source byte offsets of this wave's pieces: 2 of A (pieces 2w, 2w+1 of 8), 1 of B (piece w of 4)
    unsigned offA[2], offB;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int piece = wave * 2 + i;
        if (A_KC) {
            const int row = piece * 16 + (lane >> 2), c = (lane & 3) ^ gemm_swz(row);
            offA[i] = 4u * (unsigned)(min(row0 + row, p.a_clamp) * p.lda + 4 * c);
        } else {
            const int kk = piece * 2 + (lane >> 5), col = (lane & 31) * 4;
            offA[i] = 4u * (unsigned)(kk * p.lda + min(row0 + col, p.a_clamp));
        }
    }
    if (B_KC) {
        const int row = wave * 16 + (lane >> 2), c = (lane & 3) ^ gemm_swz(row);
        offB = 4u * (unsigned)(min(col0 + row, p.b_clamp) * p.ldb + 4 * c);
    } else {
        const int kk = wave * 4 + (lane >> 4), col = (lane & 15) * 4;
        offB = 4u * (unsigned)(kk * p.ldb + min(col0 + col, p.b_clamp));
    }
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(gemm_lds_ptr)ring_smem);
    const unsigned lds_tailrow = lds0 + 4u * (unsigned)(NSTAGE * STG), lds_scale = lds_tailrow + 4u * (unsigned)nside;

    auto issue = [&](const int kt_raw, const int stage) {
        const int kt = min(kt_raw, nkt - 1);        // past the end: re-fetch the last tile (valid addresses, never read)
        const unsigned dst = lds0 + 4u * (unsigned)(stage * STG);
        const float* Ak = A + (A_KC ? (long)kt * BK : (long)kt * BK * p.lda);
        const float* Bk = B + (B_KC ? (long)kt * BK : (long)kt * BK * p.ldb);
        gemm_dma16(Ak, offA[0], dst + 4 * ((wave * 2 + 0) * 256));
        gemm_dma16(Ak, offA[1], dst + 4 * ((wave * 2 + 1) * 256));
        gemm_dma16(Bk, offB, dst + 4 * (SA + wave * 256));
    };
    // prologue: the two reduction-index vectors first (LDS-DMA too, 1 KB pieces dealt over the waves; older than every ring piece,
    // so any later vmcnt wait covers them; lanes past the end re-read the last float4), then NSTAGE-1 k-tiles in flight
    {
        const unsigned last = 4u * (unsigned)(nkt * BK - 4);
        if (TAIL) {
            if (side_wg) {
                const float* __restrict__ tail_src = A + (long)p.tail_row * p.lda;      // (A already points at this part's first k-tile)
                for (int pc = wave; pc * 256 < nkt * BK; pc += 4) gemm_dma16(tail_src, min(1024u * pc + 16u * lane, last), lds_tailrow + 1024u * pc);
            }
        }
        if (SCALE) {
            if (bscale != nullptr) {      // (no scale pending, K3: the loop copy without the multiplies runs, below)
                for (int pc = wave; pc * 256 < nkt * BK; pc += 4) gemm_dma16(bscale, min(1024u * pc + 16u * lane, last), lds_scale + 1024u * pc);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NSTAGE - 1; ++i) issue(i, i);

    // per-lane fragment byte offsets inside a stage
    const unsigned oA0 = A_KC ? 4 * (arow * 16 + 4 * ((0 + hh) ^ gemm_swz(arow))) : 4 * (4 * (0 + hh) * BM + arow);
    const unsigned oA1 = A_KC ? 4 * (arow * 16 + 4 * ((2 + hh) ^ gemm_swz(arow))) : 4 * (4 * (2 + hh) * BM + arow);
    const unsigned oB0 = 4 * SA + (B_KC ? 4 * (bcol * 16 + 4 * ((0 + hh) ^ gemm_swz(bcol))) : 4 * (4 * (0 + hh) * BN + bcol));
    const unsigned oB1 = 4 * SA + (B_KC ? 4 * (bcol * 16 + 4 * ((2 + hh) ^ gemm_swz(bcol))) : 4 * (4 * (2 + hh) * BN + bcol));
    const int tj = tid & 63, tg = tid >> 6;
    const unsigned oTB = 4 * SA + (B_KC ? 4 * (tj * 16 + 4 * (tg ^ gemm_swz(tj))) : 4 * (4 * tg * BN + tj));
    const unsigned oRS = 4 * SA + 4 * ((tid >> 2) * 16 + 4 * (tid & 3));

    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
    float tail_acc = 0.f, rowsum_acc = 0.f;

    // Fragment registers, two sets (k-tile kt is multiplied from one while k-tile kt+1 is read into the other).  Inline-asm reads:
    // see gemm_dma.h on why the compiler must not see them.
    struct Frags {
        gemm_f32x4 a4[2];                    // A_KC: one float4 per k half
        gemm_f32x2 a2[2][2];                 // !A_KC: (e, e+1) pairs per k half
        gemm_f32x4 b4[2][2];                 // B_KC: [k half][column tile]
        gemm_f32x2 b2[2][4];                 // !B_KC: [k half][e] = (col, col + 32)
        gemm_f32x4 sc[2], t4, tb4, ts4, rs4;
        gemm_f32x2 tbxy, tbzw;
    };
    Frags fr0, fr1;

    // The reads of one k-tile in six parts, dealt out between the MFMAs of the previous tile.  DT (compile time): this workgroup
    // carries the tail row / the row sums.
    auto read_part = [&](auto dt_c, auto sc_c, Frags& f, const unsigned sb, const int kt, const int part) {
        constexpr bool DT = decltype(dt_c)::value;
        constexpr bool SC = SCALE && decltype(sc_c)::value;      // a lazy row scale of B is pending (K1), not just possible
        const int q = part >> 1;
        if (part < 4) {
            const unsigned ao = sb + (q ? oA1 : oA0), bo = sb + (q ? oB1 : oB0);
            if ((part & 1) == 0) {
                if (A_KC) {
                    f.a4[q] = gemm_lds_read_b128<0>(ao);
                } else {
                    f.a2[q][0] = gemm_lds_read2st64_b32<0, 2>(ao);          // k rows e = 0, 1 (BM = 2 x 64 dwords apart)
                    f.a2[q][1] = gemm_lds_read2st64_b32<4, 6>(ao);          // e = 2, 3
                }
                if (B_KC) {
                    f.b4[q][0] = gemm_lds_read_b128<0>(bo);
                } else {
                    f.b2[q][0] = gemm_lds_read2_b32<0 * BN, 0 * BN + 32>(bo);
                    f.b2[q][1] = gemm_lds_read2_b32<1 * BN, 1 * BN + 32>(bo);
                }
            } else {
                if (B_KC) {
                    f.b4[q][1] = gemm_lds_read_b128<2048>(bo);
                } else {
                    f.b2[q][2] = gemm_lds_read2_b32<2 * BN, 2 * BN + 32>(bo);
                    f.b2[q][3] = gemm_lds_read2_b32<3 * BN, 3 * BN + 32>(bo);
                }
                if (SC) f.sc[q] = gemm_lds_read_b128<0>(lds_scale + 4u * (unsigned)(kt * BK + 4 * (2 * q + hh)));
            }
        } else if (part == 4) {
            if (TAIL && DT) {
                f.t4 = gemm_lds_read_b128<0>(lds_tailrow + 4u * (unsigned)(kt * BK + 4 * tg));
                if (B_KC) {
                    f.tb4 = gemm_lds_read_b128<0>(sb + oTB);
                } else {
                    f.tbxy = gemm_lds_read2_b32<0 * BN, 1 * BN>(sb + oTB);
                    f.tbzw = gemm_lds_read2_b32<2 * BN, 3 * BN>(sb + oTB);
                    if (SC) f.ts4 = gemm_lds_read_b128<0>(lds_scale + 4u * (unsigned)(kt * BK + 4 * tg));
                }
            }
        } else if (part == 5) {
            if (B_KC && DT) {
                if (do_rowsum) f.rs4 = gemm_lds_read_b128<0>(sb + oRS);
            }
        }
    };
    // "the fragments have landed": wait, tie every destination, apply the lazy row scale of B as one burst
    auto finish_reads = [&](auto dt_c, auto sc_c, Frags& f) {
        constexpr bool DT = decltype(dt_c)::value;
        constexpr bool SC = SCALE && decltype(sc_c)::value;
        gemm_wait_lds();
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            if (A_KC) gemm_tie(f.a4[q]);
            else { gemm_tie(f.a2[q][0]); gemm_tie(f.a2[q][1]); }
            if (B_KC) { gemm_tie(f.b4[q][0]); gemm_tie(f.b4[q][1]); }
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) gemm_tie(f.b2[q][e]);
            }
            if (SC) {
                gemm_tie(f.sc[q]);
#pragma unroll
                for (int e = 0; e < 4; ++e) f.b2[q][e] *= f.sc[q][e];              // fl(H * s): v_pk_mul_f32
            }
        }
        if (TAIL && DT) {
            gemm_tie(f.t4);
            if (B_KC) {
                gemm_tie(f.tb4);
            } else {
                gemm_tie(f.tbxy);
                gemm_tie(f.tbzw);
                f.tb4 = gemm_f32x4{f.tbxy.x, f.tbxy.y, f.tbzw.x, f.tbzw.y};
                if (SC) {
                    gemm_tie(f.ts4);
                    f.tb4 *= f.ts4;
                }
            }
        }
        if (B_KC && DT) {
            if (do_rowsum) gemm_tie(f.rs4);
        }
    };

    // One k-tile (software-pipelined over the ring):
    //   16 MFMAs of tile kt, with the reads of tile kt+1 (other register set) and the LDS-DMA pieces of tile kt+NSTAGE-1 dealt out
    //   between them (into the stage of tile kt-1, which every wave finished reading before the last barrier) | side FMAs |
    //   fragments of tile kt+1 landed + scaled | wait until tile kt+2 has landed | barrier
    auto step = [&](auto dt_c, auto sc_c, Frags& cur, Frags& nxt, const int kt, const int stage) {
        constexpr bool DT = decltype(dt_c)::value;
        const unsigned sbn = lds0 + 4u * (unsigned)(((stage + 1 == NSTAGE) ? 0 : stage + 1) * STG);
        const int refill = (stage == 0) ? NSTAGE - 1 : stage - 1;
        const int ktf = min(kt + NSTAGE - 1, nkt - 1);      // past the end: re-fetch the last tile (valid addresses, never read)
        const int ktn = min(kt + 1, nkt - 1);
        const unsigned dst = lds0 + 4u * (unsigned)(refill * STG);
        const float* Ak = A + (A_KC ? (long)ktf * BK : (long)ktf * BK * p.lda);
        const float* Bk = B + (B_KC ? (long)ktf * BK : (long)ktf * BK * p.ldb);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float av = A_KC ? cur.a4[q][e] : cur.a2[q][e >> 1][e & 1];
                const float bv0 = B_KC ? cur.b4[q][0][e] : cur.b2[q][e][0];
                const float bv1 = B_KC ? cur.b4[q][1][e] : cur.b2[q][e][1];
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv0, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv1, acc1, 0, 0, 0);
                const int slot = q * 4 + e;
                if (slot == 0) gemm_dma16(Ak, offA[0], dst + 4 * ((wave * 2 + 0) * 256));
                if (slot == 1) gemm_dma16(Ak, offA[1], dst + 4 * ((wave * 2 + 1) * 256));
                if (slot == 2) gemm_dma16(Bk, offB, dst + 4 * (SA + wave * 256));
                if (slot < 6) read_part(dt_c, sc_c, nxt, sbn, ktn, slot);        // tile kt+1 -> the other register set
                // its fragments have landed by now (the last read went out two MFMAs ago): tie them and apply the lazy scale while the
                // matrix pipe still has four MFMAs of this tile to issue, not in the gap before the barrier
                if (RING_EARLY_FINISH && slot == 6) finish_reads(dt_c, sc_c, nxt);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (TAIL && DT) {
            tail_acc = fmaf(cur.t4.x, cur.tb4.x, tail_acc);
            tail_acc = fmaf(cur.t4.y, cur.tb4.y, tail_acc);
            tail_acc = fmaf(cur.t4.z, cur.tb4.z, tail_acc);
            tail_acc = fmaf(cur.t4.w, cur.tb4.w, tail_acc);
        }
        if (B_KC && DT) {
            if (do_rowsum) rowsum_acc += (cur.rs4.x + cur.rs4.y) + (cur.rs4.z + cur.rs4.w);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!RING_EARLY_FINISH) finish_reads(dt_c, sc_c, nxt);
        __builtin_amdgcn_sched_barrier(0);
        // tile kt+2 has landed (pieces of this wave; the barrier extends it to everyone's), and every wave is past its reads of tile kt+1
        gemm_wait_vmcnt<3 * (NSTAGE - 3)>();
        asm volatile("s_barrier" ::: "memory");
    };

    auto main_loop = [&](auto dt_c, auto sc_c) {
        // tiles 0 and 1 landed and visible (the plain LDS stores of the prologue too); fragments of tile 0
        gemm_wait_vmcnt<3 * (NSTAGE - 3)>();
        asm volatile("s_barrier" ::: "memory");
#pragma unroll
        for (int part = 0; part < 6; ++part) read_part(dt_c, sc_c, fr0, lds0, 0, part);
        finish_reads(dt_c, sc_c, fr0);
        if (p.trace && tid == 0) {
            p.trace[8 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
            p.trace[8 * blockIdx.x + 5] = __builtin_amdgcn_s_memtime();
        }
        int stage = 0;
        for (int kt = 0; kt < nkt; kt += 2) {
            step(dt_c, sc_c, fr0, fr1, kt, stage);
            stage = (stage + 1 == NSTAGE) ? 0 : stage + 1;
            if (kt + 1 < nkt) {
                step(dt_c, sc_c, fr1, fr0, kt + 1, stage);
                stage = (stage + 1 == NSTAGE) ? 0 : stage + 1;
            }
        }
    };
    // loop copies, chosen once per workgroup: with / without the side work of the row-0 tiles, with / without the lazy-scale multiplies
    const bool dt = (TAIL || B_KC) && side_wg;
    if (SCALE && bscale != nullptr) {
        if (dt) main_loop(std::true_type{}, std::true_type{});
        else main_loop(std::false_type{}, std::true_type{});
    } else {
        if (dt) main_loop(std::true_type{}, std::false_type{});
        else main_loop(std::false_type{}, std::false_type{});
    }
    if (p.trace && tid == 0) {
        p.trace[8 * blockIdx.x + 2] = __builtin_amdgcn_s_memrealtime();
        p.trace[8 * blockIdx.x + 6] = __builtin_amdgcn_s_memtime();
    }
    // the redundant pieces issued past the last k-tile are still landing; the epilogue reuses the ring as scratch
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");

    // Epilogue scratch in the (now free) ring: the output tile as a [128][TP] image, then 3 x 128 per-row factors, then 256 floats for
    // the tail-row reduction.  Why an image: in the MFMA accumulator layout a lane owns 32 scattered dwords (32 predicated dword
    // stores, and 32 dword loads for the in-place H update); turned through LDS the tile is 8 float4 per lane, rows i * 16 + tid / 16,
    // columns 4 * (tid % 16).  Measured on one file (scripts/ktrace_single.py): epilogue of the H update 4.3 -> 3.3 us (with its
    // loads reordered into one round trip), of a plain store 1.3 -> 1.2 us -- what is left is memory round trips, not instructions.
    constexpr int TP = 68;                                        // pitch: 4 rows apart (the two lane halves) is 16 banks apart
    float* const s_tile = ring_smem;
    float* const s_rows = ring_smem + BM * TP;
    float* const s_red = s_rows + 3 * BM;
    static_assert(BM * TP + 3 * BM + 256 <= 4 * STG, "epilogue scratch must fit the ring");
    const int frow = tid >> 4, fc = 4 * (tid & 15), fcol = col0 + fc;
    auto tile_to_lds = [&]() {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int lr = wave * 32 + 4 * hh + (r & 3) + 8 * (r >> 2);
            s_tile[lr * TP + l31] = acc0[r];
            s_tile[lr * TP + 32 + l31] = acc1[r];
        }
        __syncthreads();
    };

    if constexpr (EPI == EPI_UPDH) {
        // H update, in place: every global load first -- the old tile of H (8 float4 per lane in the image layout), the per-row factors
        // (lazy scale, 1 / (column sum + alpha + eps), rank-1 tail column of A; one row per thread, published through LDS) and the
        // tail row of B -- all in flight together with the turn of the accumulators through LDS; one barrier; then arithmetic and stores.
        float* Cf = p.C + file * p.sC;
        gemm_f32x4 h4[8], kb4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 8; ++i) h4[i] = *(const gemm_f32x4*)(Cf + (long)min(row0 + i * 16 + frow, p.M - 1) * p.ldc + fcol);
        if (p.ktailA) kb4 = *(const gemm_f32x4*)(p.ktailB + file * p.s_ktailB + fcol);
        if (tid < BM) {
            const int row = min(row0 + tid, p.M - 1);
            s_rows[tid] = p.E1 ? p.E1[file * p.sE1 + row] : 1.f;
            s_rows[BM + tid] = 1.0f / (p.E2[file * p.sE2 + row] + p.alpha + p.eps);
            s_rows[2 * BM + tid] = p.ktailA ? p.ktailA[file * p.s_ktailA + row] : 0.f;
        }
        tile_to_lds();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int lr = i * 16 + frow, row = row0 + lr;
            const gemm_f32x4 a = *(const gemm_f32x4*)(s_tile + lr * TP + fc);
            const float sc = s_rows[lr], rd = s_rows[BM + lr], ta = s_rows[2 * BM + lr];
            // last reduction index, in chain order; columns >= N: H is zero there and stays zero
            const float ux = fmaf(ta, kb4.x, a.x), uy = fmaf(ta, kb4.y, a.y), uz = fmaf(ta, kb4.z, a.z), uw = fmaf(ta, kb4.w, a.w);
            const gemm_f32x4 o = {(h4[i].x * sc) * (ux * rd), fcol + 1 < p.N ? (h4[i].y * sc) * (uy * rd) : 0.f,
                                  fcol + 2 < p.N ? (h4[i].z * sc) * (uz * rd) : 0.f, fcol + 3 < p.N ? (h4[i].w * sc) * (uw * rd) : 0.f};
            if (row < p.M && fcol < p.N) *(gemm_f32x4*)(Cf + (long)row * p.ldc + fcol) = o;
        }
    } else if constexpr (EPI == EPI_STORE) {
        float* Cf = p.C + file * p.sC;
        if (((p.ldc & 3) | (int)(((size_t)Cf >> 2) & 3)) == 0) {      // float4-addressable output (every NMF / GCC-NMF launch): image layout
            tile_to_lds();
            gemm_f32x4 a[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = *(const gemm_f32x4*)(s_tile + (i * 16 + frow) * TP + fc);
            if (p.ktailA) {   // last reduction index as one fmaf per element, in chain order (it is the final k)
                const float* __restrict__ tb = p.ktailB + file * p.s_ktailB;
                const gemm_f32x4 b = {tb[min(fcol, p.N - 1)], tb[min(fcol + 1, p.N - 1)], tb[min(fcol + 2, p.N - 1)], tb[min(fcol + 3, p.N - 1)]};
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float ta = p.ktailA[file * p.s_ktailA + min(row0 + i * 16 + frow, p.M - 1)];
                    a[i] = gemm_f32x4{fmaf(ta, b.x, a[i].x), fmaf(ta, b.y, a[i].y), fmaf(ta, b.z, a[i].z), fmaf(ta, b.w, a[i].w)};
                }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int row = row0 + i * 16 + frow;
                float* q = Cf + (long)row * p.ldc + fcol;
                if (row < p.M) {
                    if (fcol + 3 < p.N) {
                        *(gemm_f32x4*)q = a[i];
                    } else {          // ragged last column tile: nothing beyond column N - 1 is written
                        if (fcol < p.N) q[0] = a[i].x;
                        if (fcol + 1 < p.N) q[1] = a[i].y;
                        if (fcol + 2 < p.N) q[2] = a[i].z;
                    }
                }
            }
        } else {
            gemm_epilogue_pair<EPI>(p, file, row0 + wave * 32 + 4 * hh, col0 + l31, acc0, acc1);
        }
    } else {
        gemm_epilogue_pair<EPI>(p, file, row0 + wave * 32 + 4 * hh, col0 + l31, acc0, acc1);
    }
    if (TAIL) {
        if (side_wg) {
            s_red[tid] = tail_acc;
            __syncthreads();
            if (tid < BN) {
                const float s = (s_red[tid] + s_red[BN + tid]) + (s_red[2 * BN + tid] + s_red[3 * BN + tid]);
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
    if (p.trace) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) p.trace[8 * blockIdx.x + 3] = __builtin_amdgcn_s_memrealtime();
    }
}

// Ring depth: 6 stages (72 KB + the two reduction-index vectors) when that still lets two workgroups share a CU or the launch
// has at most one workgroup per CU anyway; 5 stages otherwise.  (Deeper rings change nothing: with >= 3 tiles in flight the
// loop is not latency-bound -- measured with 10 stages.)
template <bool A_KC, bool B_KC, int EPI, bool TAIL, int NSTAGE>
static int gccnmf_launch_gemm_ring_n(const GemmArgs& a, size_t lds, hipStream_t stream) {
    // the attribute is per device AND per instantiation; the cache is keyed by the current device (several engines on several devices
    // may share this process) and atomic (a stale read only repeats the call)
    static std::atomic<size_t> configured[GCCNMF_MAX_DEVICES];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return GCCNMF_ERR_LAUNCH;
    if (dev < 0 || dev >= GCCNMF_MAX_DEVICES || lds > configured[dev].load(std::memory_order_relaxed)) {
        if (hipFuncSetAttribute((const void*)gccnmf_gemm_ring_kernel<A_KC, B_KC, EPI, TAIL, NSTAGE>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess)
            return GCCNMF_ERR_LAUNCH;
        if (dev >= 0 && dev < GCCNMF_MAX_DEVICES) configured[dev].store(lds, std::memory_order_relaxed);
    }
    hipLaunchKernelGGL((gccnmf_gemm_ring_kernel<A_KC, B_KC, EPI, TAIL, NSTAGE>), dim3(a.batch * a.tiles_m * a.tiles_n), dim3(256), lds, stream, a);
    GCCNMF_CHECK_LAUNCH();
    return GCCNMF_OK;
}

// reduction lengths the ring kernel takes (the two vectors must fit beside the ring)
static inline bool gccnmf_ring_supports(int Kd) { return Kd >= 1 && Kd <= 4096; }

template <bool A_KC, bool B_KC, int EPI, bool TAIL>
static int gccnmf_launch_gemm_ring(GemmArgs a, hipStream_t stream) {
    if (!a.A || !a.B || !a.C || a.M < 1 || a.N < 1 || a.Kd < 1 || a.batch < 1) return GCCNMF_ERR_ARG;
    if ((a.lda & 3) || (a.ldb & 3) || !gccnmf_ring_supports(a.Kd)) return GCCNMF_ERR_ARG;
    a.ablate = 0;
    a.tiles_m = gccnmf_ceil_div(a.M, 128);
    a.tiles_n = gccnmf_ceil_div(a.N, 64);
    a.xcd_affine = 0;
    const long grid = (long)a.batch * a.tiles_m * a.tiles_n;
    a.trace = (gccnmf_trace_buf && grid <= gccnmf_trace_blocks) ? gccnmf_trace_buf : nullptr;
    const size_t side = sizeof(float) * 2 * 256 * (size_t)gccnmf_ceil_div(16 * gccnmf_ceil_div(a.Kd, 16), 256);
    const size_t stage = sizeof(float) * (128 * 16 + 64 * 16);
    if (grid <= 256 || 6 * stage + side <= 80 * 1024) return gccnmf_launch_gemm_ring_n<A_KC, B_KC, EPI, TAIL, 6>(a, 6 * stage + side, stream);
    return gccnmf_launch_gemm_ring_n<A_KC, B_KC, EPI, TAIL, 5>(a, 5 * stage + side, stream);
}
