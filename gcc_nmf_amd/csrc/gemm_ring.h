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
#include <type_traits>
#include "gemm_dma.h"

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
    constexpr bool SCALE = !B_KC && (EPI == EPI_DIV || EPI == EPI_STORE || EPI == EPI_DIVFIX);
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
            if (bscale != nullptr) {
                for (int pc = wave; pc * 256 < nkt * BK; pc += 4) gemm_dma16(bscale, min(1024u * pc + 16u * lane, last), lds_scale + 1024u * pc);
            } else {
                for (int k = tid; k < nkt * BK; k += 256) ring_smem[NSTAGE * STG + nside + k] = 1.f;     // no lazy scale pending (K3)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
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
    auto read_part = [&](auto dt_c, Frags& f, const unsigned sb, const int kt, const int part) {
        constexpr bool DT = decltype(dt_c)::value;
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
                if (SCALE) f.sc[q] = gemm_lds_read_b128<0>(lds_scale + 4u * (unsigned)(kt * BK + 4 * (2 * q + hh)));
            }
        } else if (part == 4) {
            if (TAIL && DT) {
                f.t4 = gemm_lds_read_b128<0>(lds_tailrow + 4u * (unsigned)(kt * BK + 4 * tg));
                if (B_KC) {
                    f.tb4 = gemm_lds_read_b128<0>(sb + oTB);
                } else {
                    f.tbxy = gemm_lds_read2_b32<0 * BN, 1 * BN>(sb + oTB);
                    f.tbzw = gemm_lds_read2_b32<2 * BN, 3 * BN>(sb + oTB);
                    if (SCALE) f.ts4 = gemm_lds_read_b128<0>(lds_scale + 4u * (unsigned)(kt * BK + 4 * tg));
                }
            }
        } else if (part == 5) {
            if (B_KC && DT) {
                if (do_rowsum) f.rs4 = gemm_lds_read_b128<0>(sb + oRS);
            }
        }
    };
    // "the fragments have landed": wait, tie every destination, apply the lazy row scale of B as one burst
    auto finish_reads = [&](auto dt_c, Frags& f) {
        constexpr bool DT = decltype(dt_c)::value;
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
            if (SCALE) {
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
                if (SCALE) {
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
    auto step = [&](auto dt_c, Frags& cur, Frags& nxt, const int kt, const int stage) {
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
                if (slot < 6) read_part(dt_c, nxt, sbn, ktn, slot);              // tile kt+1 -> the other register set
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
        finish_reads(dt_c, nxt);
        __builtin_amdgcn_sched_barrier(0);
        // tile kt+2 has landed (pieces of this wave; the barrier extends it to everyone's), and every wave is past its reads of tile kt+1
        gemm_wait_vmcnt<3 * (NSTAGE - 3)>();
        asm volatile("s_barrier" ::: "memory");
    };

    auto main_loop = [&](auto dt_c) {
        // tiles 0 and 1 landed and visible (the plain LDS stores of the prologue too); fragments of tile 0
        gemm_wait_vmcnt<3 * (NSTAGE - 3)>();
        asm volatile("s_barrier" ::: "memory");
#pragma unroll
        for (int part = 0; part < 6; ++part) read_part(dt_c, fr0, lds0, 0, part);
        finish_reads(dt_c, fr0);
        if (p.trace && tid == 0) {
            p.trace[8 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
            p.trace[8 * blockIdx.x + 5] = __builtin_amdgcn_s_memtime();
        }
        int stage = 0;
        for (int kt = 0; kt < nkt; kt += 2) {
            step(dt_c, fr0, fr1, kt, stage);
            stage = (stage + 1 == NSTAGE) ? 0 : stage + 1;
            if (kt + 1 < nkt) {
                step(dt_c, fr1, fr0, kt + 1, stage);
                stage = (stage + 1 == NSTAGE) ? 0 : stage + 1;
            }
        }
    };
    if ((TAIL || B_KC) && side_wg) main_loop(std::true_type{});
    else main_loop(std::false_type{});
    if (p.trace && tid == 0) {
        p.trace[8 * blockIdx.x + 2] = __builtin_amdgcn_s_memrealtime();
        p.trace[8 * blockIdx.x + 6] = __builtin_amdgcn_s_memtime();
    }
    // the redundant pieces issued past the last k-tile are still landing; the epilogue reuses the ring as scratch
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");

    if constexpr (EPI == EPI_DIVFIX) {
        // Split-K with the combine folded into the launch (one file alone: W.H in kparts parts).  Every part stores its partial tile
        // (and its share of the VALU tail row) to C + part * sC, then bumps the tile's arrival counter; the part that arrives LAST
        // reads all kparts partials back -- its own included, so what is added does not depend on who is last -- adds them in
        // ascending part order, exactly as nmf_div_partials_kernel does, and writes C2 = E0 / sum.  Result: bit-identical to the
        // two-launch form, without the 5 us combine launch behind every W.H.  The counter is left at zero for the next launch.
        // Visibility across the 8 XCD-private L2s, fix_mode 1: agent-scope fences around the counter (release before the bump, acquire
        // before the read-back); fix_mode 2: the partials themselves are agent-scope relaxed atomic stores / loads (write-through,
        // L2-coherent reads), ordered by completion (vmcnt) before the bump.
        auto fix = [&](auto mode_c) {
            constexpr int MODE = decltype(mode_c)::value;
            const int nparts = p.kparts;
            const int rb = row0 + wave * 32 + 4 * hh;
            const int ca = col0 + l31, cb = ca + 32;
            const bool oka = ca < p.N, okb = cb < p.N;
            const int cac = min(ca, p.N - 1), cbc = min(cb, p.N - 1);
            const int tcol = min(col0 + tid, p.N - 1);
            float tail_s = 0.f;
            if (TAIL) {
                if (side_wg) {
                    ring_smem[tid] = tail_acc;
                    __syncthreads();
                    if (tid < BN) tail_s = (ring_smem[tid] + ring_smem[BN + tid]) + (ring_smem[2 * BN + tid] + ring_smem[3 * BN + tid]);
                }
            }
            const bool tail_lane = TAIL && side_wg && tid < BN && (col0 + tid) < p.N;
            const long tail_off = (long)p.tail_row * p.ldc + tcol;
            float* Pm = p.C + (long)file * p.sC;
            auto st = [&](float* q, float v) {
                if (MODE == 2) __hip_atomic_store(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else *q = v;
            };
            auto ld = [&](const float* q) -> float {
                if (MODE == 2) return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return *q;                 // after the acquire fence
            };
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rb + (r & 3) + 8 * (r >> 2);
                if (row < p.M) {
                    if (oka) st(Pm + (long)row * p.ldc + ca, acc0[r]);
                    if (okb) st(Pm + (long)row * p.ldc + cb, acc1[r]);
                }
            }
            if (tail_lane) st(Pm + tail_off, tail_s);
            // E0 (V) does not depend on the other parts: in flight while the counter is bumped
            float va[16], vb[16], vt = 1.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long ro = (long)min(rb + (r & 3) + 8 * (r >> 2), p.M - 1) * p.ldc;
                va[r] = p.E0[ro + cac];
                vb[r] = p.E0[ro + cbc];
            }
            if (tail_lane) vt = p.E0[tail_off];
            if (MODE == 1) __threadfence();
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            unsigned* s_flag = (unsigned*)ring_smem + 320;
            unsigned* counter = p.fix_counter + (tm * p.tiles_n + tn);
            if (tid == 0) *s_flag = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            if ((int)*s_flag != nparts - 1) return;
            if (MODE == 1) __threadfence();
            float pa[4][16], pb[4][16], pt[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                pt[q] = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) pa[q][r] = pb[q][r] = 0.f;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (q < nparts) {
                    const float* Pq = p.C + (long)q * p.sC;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const long ro = (long)min(rb + (r & 3) + 8 * (r >> 2), p.M - 1) * p.ldc;
                        pa[q][r] = ld(Pq + ro + cac);
                        pb[q][r] = ld(Pq + ro + cbc);
                    }
                    if (tail_lane) pt[q] = ld(Pq + tail_off);
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rb + (r & 3) + 8 * (r >> 2);
                const float da = ((pa[0][r] + pa[1][r]) + pa[2][r]) + pa[3][r];       // absent parts add +0.0f: exact
                const float db = ((pb[0][r] + pb[1][r]) + pb[2][r]) + pb[3][r];
                if (row < p.M) {
                    if (oka) p.C2[(long)row * p.ldc + ca] = va[r] / da;
                    if (okb) p.C2[(long)row * p.ldc + cb] = vb[r] / db;
                }
            }
            if (tail_lane) p.C2[tail_off] = vt / (((pt[0] + pt[1]) + pt[2]) + pt[3]);
            if (tid == 0) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        };
        if (p.fix_mode == 2) fix(std::integral_constant<int, 2>{});
        else fix(std::integral_constant<int, 1>{});
    } else if constexpr (EPI == EPI_UPDHFIX) {
        // The H update (K2) of one file alone as a split-K launch: 160 output tiles are 160 workgroups for 256 CUs, each a chain of 32
        // k-tiles; cut in kparts parts every CU is busy and the chain is a third as long.  Hand-over as in EPI_DIVFIX; the last part
        // adds the partial W^T.R tiles in ascending order, then the rank-1 reduction tail (f = F-1, still the final reduction index) and
        // the update itself exactly as EPI_UPDH does: C2 = (C2 * E1[row]) * (acc / (E2[row] + alpha + eps)) via the row reciprocal.
        auto fixh = [&](auto mode_c) {
            constexpr int MODE = decltype(mode_c)::value;
            const int nparts = p.kparts;
            float* s_sc = ring_smem, *s_rd = ring_smem + BM, *s_ta = ring_smem + 2 * BM;
            if (tid < BM) {
                const int row = min(row0 + tid, p.M - 1);
                s_sc[tid] = p.E1 ? p.E1[row] : 1.f;
                s_rd[tid] = 1.0f / (p.E2[row] + p.alpha + p.eps);
                s_ta[tid] = p.ktailA ? p.ktailA[row] : 0.f;
            }
            const int rb = row0 + wave * 32 + 4 * hh;
            const int ca = col0 + l31, cb = ca + 32;
            const bool oka = ca < p.N, okb = cb < p.N;
            const int cac = min(ca, p.N - 1), cbc = min(cb, p.N - 1);
            const float ba = p.ktailA ? p.ktailB[cac] : 0.f, bb = p.ktailA ? p.ktailB[cbc] : 0.f;
            float* Pm = p.C + (long)file * p.sC;
            auto st = [&](float* q, float v) {
                if (MODE == 2) __hip_atomic_store(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else *q = v;
            };
            auto ld = [&](const float* q) -> float {
                if (MODE == 2) return __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return *q;                 // after the acquire fence
            };
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rb + (r & 3) + 8 * (r >> 2);
                if (row < p.M) {
                    if (oka) st(Pm + (long)row * p.ldc + ca, acc0[r]);
                    if (okb) st(Pm + (long)row * p.ldc + cb, acc1[r]);
                }
            }
            // the tile of H this launch will rewrite is touched by nobody but the last part: its old values can be in flight now
            float ha[16], hb[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long ro = (long)min(rb + (r & 3) + 8 * (r >> 2), p.M - 1) * p.ldc;
                ha[r] = p.C2[ro + cac];
                hb[r] = p.C2[ro + cbc];
            }
            if (MODE == 1) __threadfence();
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            unsigned* s_flag = (unsigned*)ring_smem + 3 * BM + 16;
            unsigned* counter = p.fix_counter + (tm * p.tiles_n + tn);
            if (tid == 0) *s_flag = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            if ((int)*s_flag != nparts - 1) return;
            if (MODE == 1) __threadfence();
            float pa[4][16], pb[4][16];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int r = 0; r < 16; ++r) pa[q][r] = pb[q][r] = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (q < nparts) {
                    const float* Pq = p.C + (long)q * p.sC;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const long ro = (long)min(rb + (r & 3) + 8 * (r >> 2), p.M - 1) * p.ldc;
                        pa[q][r] = ld(Pq + ro + cac);
                        pb[q][r] = ld(Pq + ro + cbc);
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int lr = wave * 32 + 4 * hh + (r & 3) + 8 * (r >> 2), row = row0 + lr;
                const float sc = s_sc[lr], rd = s_rd[lr], ta = s_ta[lr];
                const float ua = fmaf(ta, ba, ((pa[0][r] + pa[1][r]) + pa[2][r]) + pa[3][r]);      // absent parts add +0.0f: exact
                const float ub = fmaf(ta, bb, ((pb[0][r] + pb[1][r]) + pb[2][r]) + pb[3][r]);
                if (row < p.M) {
                    if (oka) p.C2[(long)row * p.ldc + ca] = (ha[r] * sc) * (ua * rd);
                    if (okb) p.C2[(long)row * p.ldc + cb] = (hb[r] * sc) * (ub * rd);
                }
            }
            if (tid == 0) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        };
        if (p.fix_mode == 2) fixh(std::integral_constant<int, 2>{});
        else fixh(std::integral_constant<int, 1>{});
    } else if constexpr (EPI == EPI_UPDH) {
        // H update with the per-row factors (lazy scale, 1 / (column sum + alpha + eps), rank-1 tail column of A) put into LDS once per
        // workgroup: the generic epilogue loads them per element and divides per element (4.0 us of a 21 us launch for one file)
        float* s_sc = ring_smem, *s_rd = ring_smem + BM, *s_ta = ring_smem + 2 * BM;
        if (tid < BM) {
            const int row = min(row0 + tid, p.M - 1);
            s_sc[tid] = p.E1 ? p.E1[file * p.sE1 + row] : 1.f;
            s_rd[tid] = 1.0f / (p.E2[file * p.sE2 + row] + p.alpha + p.eps);
            s_ta[tid] = p.ktailA ? p.ktailA[file * p.s_ktailA + row] : 0.f;
        }
        __syncthreads();
        const int ca = col0 + l31, cb = ca + 32;
        const bool oka = ca < p.N, okb = cb < p.N;
        const float ba = p.ktailA ? p.ktailB[file * p.s_ktailB + min(ca, p.N - 1)] : 0.f;
        const float bb = p.ktailA ? p.ktailB[file * p.s_ktailB + min(cb, p.N - 1)] : 0.f;
        float* Cf = p.C + file * p.sC;
        float ha[16], hb[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {                     // every load first (in-place update: the compiler must not interleave)
            const long ro = (long)min(row0 + wave * 32 + 4 * hh + (r & 3) + 8 * (r >> 2), p.M - 1) * p.ldc;
            ha[r] = Cf[ro + min(ca, p.N - 1)];
            hb[r] = Cf[ro + min(cb, p.N - 1)];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int lr = wave * 32 + 4 * hh + (r & 3) + 8 * (r >> 2), row = row0 + lr;
            const float sc = s_sc[lr], rd = s_rd[lr], ta = s_ta[lr];
            const float ua = fmaf(ta, ba, acc0[r]), ub = fmaf(ta, bb, acc1[r]);     // last reduction index, in chain order
            if (row < p.M) {
                if (oka) Cf[(long)row * p.ldc + ca] = (ha[r] * sc) * (ua * rd);
                if (okb) Cf[(long)row * p.ldc + cb] = (hb[r] * sc) * (ub * rd);
            }
        }
    } else {
        gemm_epilogue_pair<EPI>(p, file, row0 + wave * 32 + 4 * hh, col0 + l31, acc0, acc1);
    }
    if constexpr (TAIL && EPI != EPI_DIVFIX) {
        if (side_wg) {
            ring_smem[tid] = tail_acc;
            __syncthreads();
            if (tid < BN) {
                const float s = (ring_smem[tid] + ring_smem[BN + tid]) + (ring_smem[2 * BN + tid] + ring_smem[3 * BN + tid]);
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
    static size_t configured = 0;      // per instantiation
    if (lds > configured) {
        if (hipFuncSetAttribute((const void*)gccnmf_gemm_ring_kernel<A_KC, B_KC, EPI, TAIL, NSTAGE>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess)
            return GCCNMF_ERR_LAUNCH;
        configured = lds;
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
