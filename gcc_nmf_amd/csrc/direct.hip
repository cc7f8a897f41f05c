// Latency-path GEMM for gfx950: operands straight from global memory into MFMA registers, the reduction split over the
// waves of a workgroup.  Used where a launch cannot fill the chip with throughput tiles: one mixture alone (the shape behind the
// reference's own function names: performKLNMF(V (513, 1244), 1024, 100, 0), gccNMF/gccNMFFunctions.py:69-83), a handful of files.
//
// Why this form.  A launch over one file is as long as ONE workgroup's dependent chain, so the way to make it short is (i) every
// SIMD of the chip busy, (ii) with equal work, (iii) without partial results crossing workgroups.  A 513 x 1244 x 1024 product is
// 640 blocks of 32 x 32 -- on 1024 SIMDs that is either 62 % of the chip (one block per wave) or a split of the reduction ACROSS
// workgroups (rounds 2-3: three partial products through HBM plus two combine launches per W.H, 22.6 us against an MFMA floor of
// 10).  Here the output is cut into exactly-one-round tiles of 16 x 16 blocks instead (32 x 80 for W.H: 16 x 16 = 256 workgroups),
// and the reduction is split INSIDE the workgroup: wave w multiplies the 16-deep chunks w, w+4, w+8 ... of the reduction into its own
// copy of the whole tile's accumulators, and the four copies are added in wave order through LDS at the end (deterministic).
// No operand is shared between waves, so there is nothing to stage: with both operands reduction-major (direct.h) lane (c, g) of a
// wave loads 16 bytes at row 4g + e, column 4c of the tile -- 256 contiguous bytes per lane group -- and the four floats are the
// B (or A) values of four 16 x 16 blocks whose columns interleave (block j owns columns 4c + j).  One chunk of a 32 x 80 tile is
// 12 load instructions for 40 MFMAs (v_mfma_f32_16x16x4_f32, 32 cycles each, 10 independent accumulators): no LDS, no LDS-DMA
// issue cost, no barrier, no fragment waits in the loop; the next chunk's loads are in flight under the current chunk's MFMAs.
//
// Numerics: exact f32 like every other GEMM of the path (k-ordered fmaf chains per wave, then ((p0 + p1) + p2) + p3).
#include <atomic>
#include <type_traits>
#include "direct.h"

#define DIRECT_MAX_DEVICES 64
extern long long* gccnmf_trace_buf;
extern int gccnmf_trace_blocks;

typedef float df32x4 __attribute__((ext_vector_type(4)));
typedef float df32x2 __attribute__((ext_vector_type(2)));

template <int W> struct DVec;
template <> struct DVec<1> { typedef float type; };
template <> struct DVec<2> { typedef df32x2 type; };
template <> struct DVec<4> { typedef df32x4 type; };
__device__ __forceinline__ float dget(float v, int) { return v; }
__device__ __forceinline__ float dget(df32x2 v, int j) { return v[j]; }
__device__ __forceinline__ float dget(df32x4 v, int j) { return v[j]; }

// Operand loads are buffer loads: wave-uniform descriptor (SGPRs) + constant per-lane byte offset (voffset) + the chunk's byte offset
// (soffset, an SGPR) -- no per-load 64-bit address arithmetic on the VALU, and loads the compiler counts in its vmcnt bookkeeping.
typedef unsigned du32x4 __attribute__((ext_vector_type(4)));
typedef unsigned du32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t direct_rsrc(const float* base, unsigned bytes) {
    const unsigned long long a = (unsigned long long)base;      // uniformity made provable: the halves go through readfirstlane
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
template <int W> struct DLoad;
template <> struct DLoad<1> {
    static __device__ __forceinline__ float ld(__amdgpu_buffer_rsrc_t r, unsigned v, unsigned s) {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, v, s, 0));
    }
};
template <> struct DLoad<2> {
    static __device__ __forceinline__ df32x2 ld(__amdgpu_buffer_rsrc_t r, unsigned v, unsigned s) {
        return __builtin_bit_cast(df32x2, __builtin_amdgcn_raw_buffer_load_b64(r, v, s, 0));
    }
};
template <> struct DLoad<4> {
    static __device__ __forceinline__ df32x4 ld(__amdgpu_buffer_rsrc_t r, unsigned v, unsigned s) {
        return __builtin_bit_cast(df32x4, __builtin_amdgcn_raw_buffer_load_b128(r, v, s, 0));
    }
};

// One 16-deep chunk of both operands in registers: step e (four MFMA steps per chunk) multiplies reduction index 4g + e of the chunk
// in lane group g = lane / 16.
template <int MB, int NB>
struct DirectFrag {
    static constexpr int G4 = NB / 4, REM = NB % 4;
    typename DVec<MB>::type a[4];
    df32x4 b4[4][G4 ? G4 : 1];
    typename DVec<REM ? REM : 1>::type br[4];
    df32x4 sc, tl;                                // lazy scale of B / tail-row values of A for reduction indexes 4g .. 4g + 3
};

template <int MB, int NB, int EPI, int NW, int NBUF>
__global__ __launch_bounds__(64 * NW, 1) void gccnmf_direct_kernel(const DirectArgs p) {
    static_assert(NBUF >= 2 && NBUF <= 4, "register sets of the operand pipeline (NBUF - 1 chunks in flight)");
    static_assert(MB == 1 || MB == 2 || MB == 4, "rows of a tile: 16, 32 or 64");
    static_assert(NB % 4 != 3, "column groups: float4s plus one float2 or float");
    constexpr int TR = 16 * MB, TC = 16 * NB;
    constexpr int G4 = NB / 4, REM = NB % 4;
    constexpr int P = TC + 4;                                     // pitch of a partial tile in LDS
    constexpr int PT = TR + 1;                                    // pitch of the transposed image
    constexpr int NT = 64 * NW;
    constexpr int C4 = TC / 4, ITEMS = (TR * C4 + NT - 1) / NT;   // float4 outputs per thread
    constexpr bool TRANSPOSED = EPI == DEPI_DIVT || EPI == DEPI_UPDH;
    extern __shared__ __attribute__((aligned(16))) float dsm[];
    float* const s_part = dsm;                                    // [NW][TR][P]
    float* const s_t = s_part + NW * TR * P;                      // [TC][PT]
    float* const s_side = s_t + TC * PT;                          // [2][NW][TC]

    // block -> (file, tm, tn): blocks b, b + 8, ... run on XCD b % 8 (observed, for speed only); XCD x owns the sub-grid
    // (x / xc, x % xc) of sm x sn tiles, so its L2 holds sm row panels of A and sn column panels of B
    const int per_file = 8 * p.sm * p.sn;
    const int file = blockIdx.x / per_file, rem = blockIdx.x - file * per_file;
    const int xcd = rem & 7, idx = rem >> 3;
    const int tm = (xcd / p.xc) * p.sm + idx / p.sn, tn = (xcd % p.xc) * p.sn + idx % p.sn;
    if (tm >= p.tiles_m || tn >= p.tiles_n) return;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, g = lane >> 4;
    const int m0 = tm * TR, n0 = tn * TC;
    if (p.trace && tid == 0) {      // per-workgroup timeline (gccnmf_debug_set_trace, scripts/ktrace_single.py): same slots as the ring kernel
        p.trace[8 * blockIdx.x + 0] = __builtin_amdgcn_s_memrealtime();
        p.trace[8 * blockIdx.x + 4] = (long long)((__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u) << 16 | (__builtin_amdgcn_s_getreg((31 << 11) | 4) & 0xffffu));
    }
    const float* __restrict__ A = p.A + file * p.sA;
    const float* __restrict__ B = p.B + file * p.sB;
    const int nchunks = (p.Kd + 15) >> 4;
    const int niter = nchunks > wave ? (nchunks - wave + NW - 1) / NW : 0;

    // per-lane byte offsets inside a chunk (columns past the pitch are clamped: they belong to outputs that are never stored)
    unsigned offA[4], offB[4][G4 ? G4 : 1], offR[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int r = 4 * g + e;
        offA[e] = 4u * (unsigned)(r * p.lda + min(m0 + MB * c, p.lda - MB));
#pragma unroll
        for (int q = 0; q < G4; ++q) offB[e][q] = 4u * (unsigned)(r * p.ldb + min(n0 + 64 * q + 4 * c, p.ldb - 4));
        offR[e] = REM ? 4u * (unsigned)(r * p.ldb + min(n0 + 64 * G4 + REM * c, p.ldb - REM)) : 0u;
    }
    const bool side = tm == 0 && (p.tailA != nullptr || p.rowsumB != nullptr) && EPI != DEPI_UPDH;
    // (no tail row: the loop still runs its FMAs on some valid memory, the result is dropped)
    const float* __restrict__ tailv = p.tailA ? p.tailA + file * p.s_tailA : B;
    const float* __restrict__ scalev = p.bscale ? p.bscale + file * p.s_bscale : B;

    df32x4 acc[MB][NB];
#pragma unroll
    for (int j = 0; j < MB; ++j)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[j][nb] = df32x4{0.f, 0.f, 0.f, 0.f};
    float tacc[NB], racc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) tacc[nb] = racc[nb] = 0.f;

    typedef DirectFrag<MB, NB> Frag;
    // every load is unconditional (a conditional one would make the compiler drain the queue at the join): past the end the last
    // chunk is fetched again and never used
    const unsigned chunkA = 64u * (unsigned)p.lda, chunkB = 64u * (unsigned)p.ldb;      // bytes per 16-row chunk
    const __amdgpu_buffer_rsrc_t rA = direct_rsrc(A, (unsigned)nchunks * chunkA), rB = direct_rsrc(B, (unsigned)nchunks * chunkB);
    const __amdgpu_buffer_rsrc_t rS = direct_rsrc(scalev, 64u * (unsigned)nchunks), rT = direct_rsrc(tailv, 64u * (unsigned)nchunks);
    auto load = [&](auto side_c, auto sc_c, Frag& f, int ch) {
        constexpr bool SIDE = decltype(side_c)::value, SC = decltype(sc_c)::value;
        const unsigned chu = (unsigned)__builtin_amdgcn_readfirstlane(min(ch, nchunks - 1));
        const unsigned sa = chu * chunkA, sb = chu * chunkB;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            f.a[e] = DLoad<MB>::ld(rA, offA[e], sa);
#pragma unroll
            for (int q = 0; q < G4; ++q) f.b4[e][q] = DLoad<4>::ld(rB, offB[e][q], sb);
            if (REM) f.br[e] = DLoad<REM ? REM : 1>::ld(rB, offR[e], sb);
        }
        if (SC) f.sc = DLoad<4>::ld(rS, 16u * (unsigned)g, 64u * chu);
        if (SIDE) f.tl = DLoad<4>::ld(rT, 16u * (unsigned)g, 64u * chu);
    };
    // The lazy scale of B per reduction index (K1: H carries the previous normalisation as the vector s) is applied to the A values
    // instead: s[r] * (A[r][m] * B[r][n]) summed over r is the same product, MB multiplies per step instead of NB, and a wave's own A
    // values are not shared with anybody (fl(W s) . H rather than W . fl(s H): one rounding per term either way).  Timeline of one
    // mixture (scripts/ktrace_single.py): the main loop of K1 took 12.8 us with the B-side multiplies, K3 -- the same loop without a
    // scale -- 10.6.  Row sums of B are taken only where they are asked for (the R.H^T launch), not in every side workgroup.
    constexpr bool ROWSUM = EPI == DEPI_STORE;
    auto compute = [&](auto side_c, auto sc_c, const Frag& f) {
        constexpr bool SIDE = decltype(side_c)::value, SC = decltype(sc_c)::value;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float bv[NB];
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) bv[nb] = nb < 4 * G4 ? f.b4[e][nb < 4 * G4 ? nb / 4 : 0][nb & 3] : dget(f.br[e], nb < 4 * G4 ? 0 : nb - 4 * G4);
#pragma unroll
            for (int j = 0; j < MB; ++j) {
                float av = dget(f.a[e], j);
                if (SC) av *= f.sc[e];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) acc[j][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[nb], acc[j][nb], 0, 0, 0);
            }
            if (SIDE) {
                const float tv = SC ? f.tl[e] * f.sc[e] : f.tl[e];
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    tacc[nb] = fmaf(tv, bv[nb], tacc[nb]);
                    if (ROWSUM) racc[nb] += bv[nb];
                }
            }
        }
    };
    auto main_loop = [&](auto side_c, auto sc_c) {
        // Operand pipeline: NBUF register sets, the loads of chunk i + NBUF - 1 are issued while chunk i is multiplied (what has to be
        // covered is a first touch of the operands per XCD -- every kernel of the iteration reads what the previous one wrote, so a line
        // comes from the memory side once per XCD, and the 32 workgroups of an XCD that share it wait for it together).
        // One scheduling region per chunk: the loads are dealt out between the chunk's MFMAs (one per MPL MFMAs), so the memory
        // pipeline takes them at its own pace while the matrix pipe stays busy.  Left alone, the scheduler sinks every load to just
        // before its first use to save registers, and the loop waits for a full memory round trip per chunk.
        constexpr int NLOADS = 4 * (1 + G4 + (REM ? 1 : 0)) + (decltype(side_c)::value ? 1 : 0) + (decltype(sc_c)::value ? 1 : 0);
        constexpr int NMFMA = 16 * MB * NB / 4;
        constexpr int MPL = NMFMA / NLOADS > 0 ? NMFMA / NLOADS : 1;
        auto interleave = [&]() {
#pragma unroll
            for (int i = 0; i < NLOADS; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, MPL, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            }
        };
        Frag f[NBUF];
#pragma unroll
        for (int j = 0; j < NBUF - 1; ++j) load(side_c, sc_c, f[j], wave + j * NW);
        if (p.trace && tid == 0) {
            p.trace[8 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
            p.trace[8 * blockIdx.x + 5] = __builtin_amdgcn_s_memtime();
        }
        int it = 0;
        for (; it + NBUF <= niter; it += NBUF) {
#pragma unroll
            for (int j = 0; j < NBUF; ++j) {
                __builtin_amdgcn_sched_barrier(0);
                load(side_c, sc_c, f[(j + NBUF - 1) % NBUF], wave + (it + j + NBUF - 1) * NW);
                compute(side_c, sc_c, f[j]);
                interleave();
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // fewer than NBUF chunks left: they are in flight or landed already (chunk it + j in set j)
#pragma unroll
        for (int j = 0; j < NBUF - 1; ++j)
            if (it + j < niter) compute(side_c, sc_c, f[j]);
    };
    // loop copies, chosen once per workgroup: with / without the side work of the row-0 tiles, with / without the lazy-scale multiplies
    if constexpr (EPI == DEPI_UPDH) {
        main_loop(std::false_type{}, std::false_type{});
    } else if constexpr (EPI == DEPI_DIV) {
        if (p.bscale != nullptr) {
            if (side) main_loop(std::true_type{}, std::true_type{});
            else main_loop(std::false_type{}, std::true_type{});
        } else {
            if (side) main_loop(std::true_type{}, std::false_type{});
            else main_loop(std::false_type{}, std::false_type{});
        }
    } else {
        if (side) main_loop(std::true_type{}, std::false_type{});
        else main_loop(std::false_type{}, std::false_type{});
    }

    if (p.trace && tid == 0) {
        p.trace[8 * blockIdx.x + 2] = __builtin_amdgcn_s_memrealtime();
        p.trace[8 * blockIdx.x + 6] = __builtin_amdgcn_s_memtime();
    }
    // ---- epilogue ----------------------------------------------------------------------------------------------------
    // what the element-wise part needs from global memory is requested before the partial tiles are exchanged
    const long fC = file * p.sC;
    df32x4 e0[ITEMS];                               // DIV / DIVT: V; UPDH: the old H
    df32x4 kb[ITEMS];
    float rsc[ITEMS], rrd[ITEMS], rka[ITEMS];
#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        const int id = tid + it * NT;
        const int row = id / C4, cq = id - row * C4;
        const int grow = min(m0 + row, p.M - 1), gcol = min(n0 + 4 * cq, ((p.N - 1) & ~3));
        e0[it] = df32x4{1.f, 1.f, 1.f, 1.f};
        kb[it] = df32x4{0.f, 0.f, 0.f, 0.f};
        rsc[it] = 1.f; rrd[it] = 1.f; rka[it] = 0.f;
        if (EPI == DEPI_DIV || EPI == DEPI_DIVT) e0[it] = *(const df32x4*)(p.E0 + file * p.sE0 + (long)grow * p.lde0 + gcol);
        if (EPI == DEPI_UPDH) {
            e0[it] = *(const df32x4*)(p.C + fC + (long)grow * p.ldc + gcol);
            if (p.E1) rsc[it] = p.E1[file * p.sE1 + grow];
            rrd[it] = 1.0f / (p.E2[file * p.sE2 + grow] + p.alpha + p.eps);
            if (p.ktailA) {
                rka[it] = p.ktailA[file * p.s_ktailA + grow];
                kb[it] = *(const df32x4*)(p.ktailB + file * p.s_ktailB + gcol);
            }
        }
    }

    float vtail = 1.f;                  // numerator of the tail-row element this thread will finish (requested now, used after the exchange)
    if ((EPI == DEPI_DIV || EPI == DEPI_DIVT) && side && p.tailA && tid < TC)
        vtail = p.E0[file * p.sE0 + (long)p.tail_row * p.lde0 + min(n0 + tid, p.N - 1)];
    // the wave's copy of the tile -> LDS: accumulator register i of block (j, nb) is row MB * (4g + i) + j, column 4c + (nb % 4) of
    // column group nb / 4 (one 16-byte store per group)
    {
        float* mine = s_part + wave * (TR * P);
#pragma unroll
        for (int j = 0; j < MB; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float* rowp = mine + (MB * (4 * g + i) + j) * P;
#pragma unroll
                for (int q = 0; q < G4; ++q)
                    *(df32x4*)(rowp + 64 * q + 4 * c) = df32x4{acc[j][4 * q][i], acc[j][4 * q + 1][i], acc[j][4 * q + 2][i], acc[j][4 * q + 3][i]};
                if (REM == 1) rowp[64 * G4 + c] = acc[j][4 * G4][i];
                if (REM == 2) *(df32x2*)(rowp + 64 * G4 + 2 * c) = df32x2{acc[j][4 * G4][i], acc[j][4 * G4 + 1][i]};
            }
    }
    if (side) {          // lane groups hold disjoint reduction indexes of the same columns
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            float t = tacc[nb], r = racc[nb];
            t += __shfl_xor(t, 16); t += __shfl_xor(t, 32);
            r += __shfl_xor(r, 16); r += __shfl_xor(r, 32);
            const int lc = nb < 4 * G4 ? 64 * (nb / 4) + 4 * c + (nb & 3) : 64 * G4 + REM * c + (nb - 4 * G4);
            if (g == 0) {
                s_side[wave * TC + lc] = t;
                s_side[(NW + wave) * TC + lc] = r;
            }
        }
    }
    __syncthreads();

#pragma unroll
    for (int it = 0; it < ITEMS; ++it) {
        const int id = tid + it * NT;
        if (id < TR * C4) {
            const int row = id / C4, cq = id - row * C4;
            const int grow = m0 + row, gcol = n0 + 4 * cq;
            df32x4 s = *(const df32x4*)(s_part + row * P + 4 * cq);
#pragma unroll
            for (int w = 1; w < NW; ++w) s += *(const df32x4*)(s_part + w * (TR * P) + row * P + 4 * cq);
            df32x4 o;
            if (EPI == DEPI_STORE) o = s;
            if (EPI == DEPI_DIV || EPI == DEPI_DIVT) o = df32x4{e0[it].x / s.x, e0[it].y / s.y, e0[it].z / s.z, e0[it].w / s.w};
            if (EPI == DEPI_UPDH) {
                const float sc = rsc[it], rd = rrd[it], ka = rka[it];
                o = df32x4{(e0[it].x * sc) * (fmaf(ka, kb[it].x, s.x) * rd), (e0[it].y * sc) * (fmaf(ka, kb[it].y, s.y) * rd),
                           (e0[it].z * sc) * (fmaf(ka, kb[it].z, s.z) * rd), (e0[it].w * sc) * (fmaf(ka, kb[it].w, s.w) * rd)};
            }
            // nothing outside the M x N corner is ever non-zero (the padding is a reduction operand of the next GEMM)
            const bool rv = grow < p.M;
            o.x = (rv && gcol < p.N) ? o.x : 0.f;
            o.y = (rv && gcol + 1 < p.N) ? o.y : 0.f;
            o.z = (rv && gcol + 2 < p.N) ? o.z : 0.f;
            o.w = (rv && gcol + 3 < p.N) ? o.w : 0.f;
            if (EPI != DEPI_DIVT && rv && gcol < p.N) *(df32x4*)(p.C + fC + (long)grow * p.ldc + gcol) = o;
            if (TRANSPOSED) {
                s_t[(4 * cq + 0) * PT + row] = o.x;
                s_t[(4 * cq + 1) * PT + row] = o.y;
                s_t[(4 * cq + 2) * PT + row] = o.z;
                s_t[(4 * cq + 3) * PT + row] = o.w;
            }
        }
    }
    if (TRANSPOSED) {
        __syncthreads();
        constexpr int R4 = TR / 4;
        float* Ct = p.Ct + file * p.sCt;
        for (int id = tid; id < TC * R4; id += NT) {
            const int ncol = id / R4, rq = id - ncol * R4;
            const int gn = n0 + ncol, gm = m0 + 4 * rq;
            if (gn < p.N && gm < p.M) {
                const float* src = s_t + ncol * PT + 4 * rq;
                *(df32x4*)(Ct + (long)gn * p.ldct + gm) = df32x4{src[0], src[1], src[2], src[3]};
            }
        }
    }
    if (side && tid < TC) {
        const int col = n0 + tid;
        float t = s_side[tid], r = s_side[NW * TC + tid];
#pragma unroll
        for (int w = 1; w < NW; ++w) {
            t += s_side[w * TC + tid];
            r += s_side[(NW + w) * TC + tid];
        }
        if (col < p.N) {
            if (p.tailA) {
                float o = t;
                if (EPI == DEPI_DIV || EPI == DEPI_DIVT) o = vtail / t;
                p.C[fC + (long)p.tail_row * p.ldc + col] = o;
                if (EPI == DEPI_DIVT) p.Ct[file * p.sCt + (long)col * p.ldct + p.tail_row] = o;
            }
            if (p.rowsumB) p.rowsumB[file * p.s_rowsumB + col] = r;
        }
    }
    if (p.trace) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) p.trace[8 * blockIdx.x + 3] = __builtin_amdgcn_s_memrealtime();
    }
}

// ---- launch --------------------------------------------------------------------------------------------------------------
struct DirectTile { int mb, nb; };
static const DirectTile direct_tiles[] = {{1, 1}, {1, 2}, {1, 5}, {2, 2}, {2, 4}, {2, 5}, {4, 4}, {4, 5}};
static const int direct_ntiles = sizeof(direct_tiles) / sizeof(direct_tiles[0]);

static size_t direct_lds_bytes(int mb, int nb, int nw) {
    const int TR = 16 * mb, TC = 16 * nb;
    const size_t need = sizeof(float) * ((size_t)nw * TR * (TC + 4) + (size_t)TC * (TR + 1) + 2 * (size_t)nw * TC);
    // at least 84 KB: never two of these workgroups on one CU (a launch is one round of one workgroup per CU, by construction)
    return need > 84 * 1024 ? need : 84 * 1024;
}

template <int MB, int NB, int EPI, int NBUF>
static int direct_launch_n(const DirectArgs& a, hipStream_t stream) {
    constexpr int NW = 4;
    static std::atomic<int> configured[DIRECT_MAX_DEVICES];
    const size_t lds = direct_lds_bytes(MB, NB, NW);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return GCCNMF_ERR_LAUNCH;
    if (dev < 0 || dev >= DIRECT_MAX_DEVICES || !configured[dev].load(std::memory_order_relaxed)) {
        if (hipFuncSetAttribute((const void*)gccnmf_direct_kernel<MB, NB, EPI, NW, NBUF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return GCCNMF_ERR_LAUNCH;
        if (dev >= 0 && dev < DIRECT_MAX_DEVICES) configured[dev].store(1, std::memory_order_relaxed);
    }
    hipLaunchKernelGGL((gccnmf_direct_kernel<MB, NB, EPI, NW, NBUF>), dim3(a.batch * 8 * a.sm * a.sn), dim3(64 * NW), lds, stream, a);
    GCCNMF_CHECK_LAUNCH();
    return GCCNMF_OK;
}

// register sets of the operand pipeline: tuning key 13 (0 = by tile: three sets, four for the smallest tiles)
template <int MB, int NB, int EPI>
static int direct_launch_t(const DirectArgs& a, hipStream_t stream) {
    // measured on one mixture (K = 1024; profiles/r04b_direct_bench_*.txt): W.H on 32 x 80 tiles 20.7 / 18.2 / 17.6 us with 2 / 3 / 4 sets,
    // the H update on 64 x 80 tiles 16.2 / 15.7 (a chunk there is twice as long); K = 128: R.H^T on 16 x 16 tiles 9.8 / 8.9 / 8.3
    const int depth = gccnmf_tune_direct_depth ? gccnmf_tune_direct_depth : (MB * NB <= 2 ? 4 : 3);
    if (depth == 2) return direct_launch_n<MB, NB, EPI, 2>(a, stream);
    if (MB * NB < 20 && depth == 4) return direct_launch_n<MB, NB, EPI, (MB * NB < 20 ? 4 : 3)>(a, stream);
    return direct_launch_n<MB, NB, EPI, 3>(a, stream);
}

template <int EPI>
static int direct_launch_e(const DirectArgs& a, int tile, hipStream_t s) {
    switch (tile) {
        case 0: return direct_launch_t<1, 1, EPI>(a, s);
        case 1: return direct_launch_t<1, 2, EPI>(a, s);
        case 2: return direct_launch_t<1, 5, EPI>(a, s);
        case 3: return direct_launch_t<2, 2, EPI>(a, s);
        case 4: return direct_launch_t<2, 4, EPI>(a, s);
        case 5: return direct_launch_t<2, 5, EPI>(a, s);
        case 6: return direct_launch_t<4, 4, EPI>(a, s);
        case 7: return direct_launch_t<4, 5, EPI>(a, s);
        default: return GCCNMF_ERR_ARG;
    }
}

// Tile choice: the launch should be whole rounds of one workgroup per CU (256), each as short as possible.  Cost of a candidate in
// matrix-pipe cycles of one wave: rounds x (chunks per wave x max(MFMA time of a chunk, time to fetch it) + a fixed prologue / epilogue).
static int direct_pick_tile(const DirectArgs& a) {
    const int nchunks = gccnmf_ceil_div(a.Kd, 16), per_wave = gccnmf_ceil_div(nchunks, 4);
    int best = -1;
    double best_cost = 0;
    for (int t = 0; t < direct_ntiles; ++t) {
        const int mb = direct_tiles[t].mb, nb = direct_tiles[t].nb;
        const long wgs = (long)a.batch * gccnmf_ceil_div(a.M, 16 * mb) * gccnmf_ceil_div(a.N, 16 * nb);
        const long rounds = (wgs + 255) / 256;
        const double mfma = 4.0 * mb * nb * (mb * nb == 1 ? 40 : 32);
        const double fetch = 4.0 * (16 * mb + 16 * nb) * 64.0 / 40.0;          // four waves' chunk bytes at ~40 B/clk/CU
        const double cost = rounds * (per_wave * (mfma > fetch ? mfma : fetch) + 4000.0);
        if (best < 0 || cost < best_cost * 0.999) {
            best = t;
            best_cost = cost;
        }
    }
    return best;
}

int gccnmf_direct_launch(DirectArgs a, int epi, int tile, hipStream_t stream) {
    if (!a.A || !a.B || a.M < 1 || a.N < 1 || a.Kd < 1 || a.batch < 1 || (a.lda & 3) || (a.ldb & 3)) return GCCNMF_ERR_ARG;
    if (epi != DEPI_DIVT && (!a.C || (a.ldc & 3))) return GCCNMF_ERR_ARG;
    if ((epi == DEPI_DIVT || epi == DEPI_UPDH) && (!a.Ct || (a.ldct & 3))) return GCCNMF_ERR_ARG;
    if ((epi == DEPI_DIV || epi == DEPI_DIVT) && (!a.E0 || (a.lde0 & 3))) return GCCNMF_ERR_ARG;
    if (epi == DEPI_UPDH && !a.E2) return GCCNMF_ERR_ARG;
    if (a.tailA && !a.C) return GCCNMF_ERR_ARG;
    if (a.bscale && epi != DEPI_DIV) return GCCNMF_ERR_ARG;            // the lazy scale exists in the K1 form only
    const int t = tile > 0 ? tile - 1 : direct_pick_tile(a);
    if (t < 0 || t >= direct_ntiles) return GCCNMF_ERR_ARG;
    const int TR = 16 * direct_tiles[t].mb, TC = 16 * direct_tiles[t].nb;
    a.tiles_m = gccnmf_ceil_div(a.M, TR);
    a.tiles_n = gccnmf_ceil_div(a.N, TC);
    // XCD grid (8 / xc) x xc: as little of A and B per XCD as possible, sub-grids that cover the tile grid with the fewest idle blocks
    long best_waste = -1, best_bytes = 0;
    for (int xc = 1; xc <= 8; xc *= 2) {
        const int xr = 8 / xc, sm = gccnmf_ceil_div(a.tiles_m, xr), sn = gccnmf_ceil_div(a.tiles_n, xc);
        const long waste = 8L * sm * sn - (long)a.tiles_m * a.tiles_n, bytes = (long)sm * TR + (long)sn * TC;
        if (best_waste < 0 || waste < best_waste || (waste == best_waste && bytes < best_bytes)) {
            best_waste = waste; best_bytes = bytes;
            a.xc = xc; a.sm = sm; a.sn = sn;
        }
    }
    const long grid = (long)a.batch * 8 * a.sm * a.sn;
    a.trace = (gccnmf_trace_buf && grid <= gccnmf_trace_blocks) ? gccnmf_trace_buf : nullptr;
    switch (epi) {
        case DEPI_STORE: return direct_launch_e<DEPI_STORE>(a, t, stream);
        case DEPI_DIV: return direct_launch_e<DEPI_DIV>(a, t, stream);
        case DEPI_DIVT: return direct_launch_e<DEPI_DIVT>(a, t, stream);
        case DEPI_UPDH: return direct_launch_e<DEPI_UPDH>(a, t, stream);
        default: return GCCNMF_ERR_ARG;
    }
}

extern "C" int gccnmf_gemm_direct(const gccnmf_direct_gemm* desc, int epilogue, int tile, void* stream) {
    GCCNMF_ENTER();
    if (!desc || tile < 0) return GCCNMF_ERR_ARG;
    return gccnmf_direct_launch(*desc, epilogue, tile, (hipStream_t)stream);
}

#include "gemm_mfma.h"
#ifdef GCCNMF_EXPERIMENTS
// ---- experiment: the THROUGHPUT tile without LDS (round 4) ------------------------------------------------------------------
// The same idea at batch scale, for GEMMs whose operands are both reduction-major already (the H update: A = W [f][k], B = R [f][n]):
// a 512 x 64 workgroup tile as in gemm_dma.h (4 waves x 128 x 64, v_mfma_f32_32x32x2_f32), but every wave fetches its own 128 rows
// of A (one 16-byte load per lane and k-step pair = four interleaved 32-row blocks) and the tile's 64 columns of B straight from
// L1 / L2 into registers: no LDS-DMA, no fragment reads, no barrier -- the four waves never synchronise.  B is fetched by all four
// waves of a workgroup (the second to fourth hit the CU's vector L1).  Reached through gccnmf_debug_gemm (layout bit 32) for timing.
template <int NBUF>
__global__ __launch_bounds__(256, 2) void gccnmf_gemm_stream_kernel(const GemmArgs p) {
    constexpr int BM = 512, BN = 64, RW = 128, CK = 8;         // chunk = 8 reduction rows = 4 MFMA steps of 2
    const int tiles = p.tiles_m * p.tiles_n;
    int file = blockIdx.x / tiles, tile = blockIdx.x - file * tiles;
    if (p.xcd_affine) {      // XCD x owns a contiguous eighth of the file-major tile list (blocks b, b + 8, ... run on XCD b % 8)
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, idx = xcd * p.xcd_affine + slot;
        if (idx >= p.batch * tiles) return;
        file = idx / tiles;
        tile = idx - file * tiles;
    }
    file = __builtin_amdgcn_readfirstlane(file);
    const int tm = __builtin_amdgcn_readfirstlane(tile / p.tiles_n), tn = __builtin_amdgcn_readfirstlane(tile) - tm * p.tiles_n;
    const int row0 = tm * BM, col0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row_w = row0 + wave * RW;
    if (row_w >= p.M) return;
    const float* __restrict__ A = p.A + file * p.sA;
    const float* __restrict__ B = p.B + file * p.sB;
    const int nchunks = (p.Kd + CK - 1) / CK;
    const unsigned chunkA = 4u * CK * (unsigned)p.lda, chunkB = 4u * CK * (unsigned)p.ldb;
    const __amdgpu_buffer_rsrc_t rA = direct_rsrc(A, (unsigned)nchunks * chunkA), rB = direct_rsrc(B, (unsigned)nchunks * chunkB);
    unsigned offA[4], offB0[4], offB1[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        offA[s] = 4u * (unsigned)((2 * s + hh) * p.lda + min(row_w + 4 * l31, p.a_clamp));
        offB0[s] = 4u * (unsigned)((2 * s + hh) * p.ldb + min(col0 + l31, p.b_clamp + 3));
        offB1[s] = 4u * (unsigned)((2 * s + hh) * p.ldb + min(col0 + 32 + l31, p.b_clamp + 3));
    }
    f32x16 acc[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    struct Frag {
        df32x4 a[4];
        float b0[4], b1[4];
    };
    auto load = [&](Frag& f, int ch) {
        const unsigned chu = (unsigned)__builtin_amdgcn_readfirstlane(min(ch, nchunks - 1));
        const unsigned sa = chu * chunkA, sb = chu * chunkB;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            f.a[s] = DLoad<4>::ld(rA, offA[s], sa);
            f.b0[s] = DLoad<1>::ld(rB, offB0[s], sb);
            f.b1[s] = DLoad<1>::ld(rB, offB1[s], sb);
        }
    };
    auto compute = [&](const Frag& f) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[s][m], f.b0[s], acc[m][0], 0, 0, 0);
                acc[m][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[s][m], f.b1[s], acc[m][1], 0, 0, 0);
            }
    };
    auto interleave = [&]() {      // 12 loads dealt out over the chunk's 32 MFMAs
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
    };
    Frag f[NBUF];
#pragma unroll
    for (int j = 0; j < NBUF - 1; ++j) load(f[j], j);
    int it = 0;
    for (; it + NBUF <= nchunks; it += NBUF) {
#pragma unroll
        for (int j = 0; j < NBUF; ++j) {
            __builtin_amdgcn_sched_barrier(0);
            load(f[(j + NBUF - 1) % NBUF], it + j + NBUF - 1);
            compute(f[j]);
            interleave();
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int j = 0; j < NBUF - 1; ++j)
        if (it + j < nchunks) compute(f[j]);
    // plain store: register r of block (m, n) is row row_w + 4 * ((r & 3) + 8 * (r >> 2) + 4 * hh) + m, column col0 + 32 n + l31
    float* C = p.C + file * p.sC;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = row_w + 4 * ((r & 3) + 8 * (r >> 2) + 4 * hh) + m;
            if (row < p.M) {
                if (col0 + l31 < p.N) C[(long)row * p.ldc + col0 + l31] = acc[m][0][r];
                if (col0 + 32 + l31 < p.N) C[(long)row * p.ldc + col0 + 32 + l31] = acc[m][1][r];
            }
        }
}

int gccnmf_launch_gemm_stream(GemmArgs a, hipStream_t stream) {
    if (!a.A || !a.B || !a.C || a.M < 1 || a.N < 1 || a.Kd < 1 || a.batch < 1 || (a.lda & 3)) return GCCNMF_ERR_ARG;
    a.tiles_m = gccnmf_ceil_div(a.M, 512);
    a.tiles_n = gccnmf_ceil_div(a.N, 64);
    const int tiles = a.tiles_m * a.tiles_n;
    int grid = a.batch * tiles;
    if (a.xcd_affine && a.batch >= 8) {
        a.xcd_affine = gccnmf_ceil_div(a.batch * tiles, 8);
        grid = 8 * a.xcd_affine;
    } else {
        a.xcd_affine = 0;
    }
    if (gccnmf_tune_direct_depth == 2) hipLaunchKernelGGL(gccnmf_gemm_stream_kernel<2>, dim3(grid), dim3(256), 0, stream, a);
    else if (gccnmf_tune_direct_depth == 4) hipLaunchKernelGGL(gccnmf_gemm_stream_kernel<4>, dim3(grid), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(gccnmf_gemm_stream_kernel<3>, dim3(grid), dim3(256), 0, stream, a);
    GCCNMF_CHECK_LAUNCH();
    return GCCNMF_OK;
}
#endif  // GCCNMF_EXPERIMENTS

// ---- K1 and K2 of one KL-NMF iteration in ONE launch, for short dictionaries (round 4) ----------------------------------------------
//   R = V / (W . (s*H))   then   H = (s*H) * (W^T . R) / (colsum W + alpha + eps)            (gccNMFFunctions.py:76)
// K2 reduces over ALL bins of a file, so here a workgroup owns a COLUMN tile (64 frames) and walks the bins in chunks of 64 -- the mirror
// image of the slab kernel below, with the roles of W and H exchanged:
//   * the tile's scaled H (128 atoms x 64 frames) stays in registers for the whole launch: lane (frame n, half hh) keeps
//     s[k] H[k][n] for its 64 atoms, laid out as the MFMA's B operand (64 VGPRs);
//   * a chunk of W [64 bins][128 atoms] goes through LDS once (double-buffered, register-staged in two halves under the matrix work, one
//     barrier per chunk) and is the operand of BOTH products: A of  D[f][n] = sum_k W[f][k] (sH)[k][n]  (a lane reads one float of its bin's
//     row) and B of the transposed second product  numer^T[n][k] += sum_f R[f][n] W[f][k]  (a lane reads one float of atom column k);
//   * register r of D holds bin (r&3) + 8 (r>>2) + 4 (lane>>5) for frame lane&31 -- read as an A operand (A[i = n][kk = lane>>5]) that IS
//     a pair of reduction rows (bins f, f + 4): after the divide the accumulator registers feed the second product as they are;
//   * numer^T (32 frames x 128 atoms per wave, 64 VGPRs) accumulates over all chunks; the two waves that share a frame half (bin groups
//     fg = 0 / 1 of every chunk) meet in LDS at the end (fixed order), where the tail bin's rank-1 term, the pending scale and the
//     denominator are applied and H is rewritten in place.  R never exists.
// LDS image of a chunk: row pitch 129 floats (odd: the 32 bins of a first-product read hit 32 banks) and the rows permuted -- bin
// b5 b4 b3 b2 b1 b0 sits in row b2 b5 b4 b3 b1 b0 -- so that (i) the 32 bins of a wave keep one value of row bit 4 (its lane half hh reads
// 16 atoms further: the other 32 banks) and (ii) the bin pair (f, f + 4) of a second-product read is 32 rows = 32 banks apart.
// Four waves: (bin group fg, frame half nh).  Requirements: M = F - 1 a multiple of 64 (bin M rides on the VALU), Kd <= 128.
__device__ __forceinline__ float direct_div_fast(float v, float d) {      // v / d: v_rcp_f32 + one Newton step through the exact residual
    const float r = __builtin_amdgcn_rcpf(d);
    const float q = v * r;
    return fmaf(fmaf(-d, q, v), r, q);
}

// LDS of one workgroup: two W chunks, the tail bin's W row, the tail bin's quotients
template <int KB>
struct WhUpdhLds {
    static constexpr int FLOATS = 2 * 64 * 129 + 32 * KB + 64;
};
// One work item: column tile tn of file `file`.  trace: this item's 8-slot timeline row, or nullptr.
template <int KB>
__device__ __forceinline__ void gccnmf_wh_updh_item(const WhUpdhArgs& p, const int file, const int tn, float* const smem, long long* const trace,
                                                    const int tid_in) {
    constexpr int PW = 129, NT = 16 * KB, NK = 32 * KB;          // LDS pitch | MFMA steps of the first product | padded atoms
    float (*Ws)[64][PW] = (float (*)[64][PW])smem;               // [2][64][PW]
    float* const s_wm = smem + 2 * 64 * PW;                      // [NK]
    float* const s_r = s_wm + NK;                                // [64]
    const int col0 = tn * 64;
    const int tid = tid_in, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fg = wave & 1, nh = wave >> 1;
    const float* __restrict__ W = p.W + file * p.sW;            // [M + 1][lda]
    float* __restrict__ H = p.H + file * p.sH;                  // [Kd][ldb], updated in place
    const float* __restrict__ V = p.V + file * p.sV;
    const float* __restrict__ sc = p.scale + file * p.sVec;
    const int nchunks = p.M >> 6;
    const int n = col0 + 32 * nh + l31;                          // this lane's frame
    if (trace && tid == 0) {
        trace[0] = __builtin_amdgcn_s_memrealtime();
        trace[4] = (long long)((__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u) << 16 | (__builtin_amdgcn_s_getreg((31 << 11) | 4) & 0xffffu));
    }
    // staging: the chunk is 64 rows x 8 KB float4; thread tid moves float4 q = tid + 256 i (row q / (8 KB), atoms 4 (q % (8 KB)) ..), i < 2 KB,
    // in two parts of KB float4 each; row (bin) b goes to LDS row slot(b)
    const __amdgpu_buffer_rsrc_t rW = direct_rsrc(W, 4u * (unsigned)(p.M + 1) * (unsigned)p.lda);
    auto stage_load = [&](df32x4 (&r)[KB], int chunk, int part) {
#pragma unroll
        for (int i = 0; i < KB; ++i) {
            const int q = tid + 256 * (part * KB + i), row = q / (8 * KB), c4 = q - row * (8 * KB);
            r[i] = DLoad<4>::ld(rW, 4u * (unsigned)(row * p.lda + 4 * c4), 4u * (unsigned)(64 * chunk * p.lda));
        }
    };
    auto stage_store = [&](const df32x4 (&r)[KB], int buf, int part) {
#pragma unroll
        for (int i = 0; i < KB; ++i) {
            const int q = tid + 256 * (part * KB + i), row = q / (8 * KB), c4 = q - row * (8 * KB);
            const int slot = (row & 3) | (((row >> 3) & 3) << 2) | ((row >> 5) << 4) | (((row >> 2) & 1) << 5);
            float* dst = &Ws[buf][slot][4 * c4];
            dst[0] = r[i].x; dst[1] = r[i].y; dst[2] = r[i].z; dst[3] = r[i].w;
        }
    };
    df32x4 h0[KB], h1[KB];
    stage_load(h0, 0, 0);
    stage_load(h1, 0, 1);
    if (tid < NK) s_wm[tid] = tid < p.Kd ? W[(long)p.M * p.lda + tid] : 0.f;
    // the scaled H tile, resident: hr[t] = s[k] H[k][n], k = (t & 15) + 16 hh + 32 (t >> 4)   (MFMA step t multiplies atoms (k, k + 16))
    float hr[NT];
    {
        const __amdgpu_buffer_rsrc_t rH = direct_rsrc(H, 4u * (unsigned)p.Kd * (unsigned)p.ldb);      // atoms >= Kd read as zero
        const __amdgpu_buffer_rsrc_t rS = direct_rsrc(sc, 4u * (unsigned)p.Kd);
        const unsigned ho = 4u * (unsigned)(16 * hh * p.ldb + min(n, p.ldb - 1));
#pragma unroll
        for (int t = 0; t < NT; ++t) hr[t] = DLoad<1>::ld(rH, ho, 4u * (unsigned)(((t & 15) + 32 * (t >> 4)) * p.ldb));
#pragma unroll
        for (int t = 0; t < NT; ++t) hr[t] *= DLoad<1>::ld(rS, 64u * (unsigned)hh, 4u * (unsigned)((t & 15) + 32 * (t >> 4)));
    }
    stage_store(h0, 0, 0);
    stage_store(h1, 0, 1);
    __syncthreads();
    // tail bin: r[n] = V[M][n] / sum_k W[M][k] (sH)[k][n]   (each lane half sums its atoms, the halves meet by a shuffle)
    {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < NT; ++i) t = fmaf(s_wm[(i & 15) + 16 * hh + 32 * (i >> 4)], hr[i], t);
        t += __shfl_xor(t, 32);
        if (fg == 0 && hh == 0) s_r[32 * nh + l31] = n < p.N ? V[(long)p.M * p.ldv + n] / t : 0.f;
    }
    f32x16 nt[KB];                                               // numer^T[32 nh + rows][32 ab + lane&31], partial over this wave's bins
#pragma unroll
    for (int ab = 0; ab < KB; ++ab)
#pragma unroll
        for (int r = 0; r < 16; ++r) nt[ab][r] = 0.f;
    const __amdgpu_buffer_rsrc_t rV = direct_rsrc(V, 4u * (unsigned)(p.M + 1) * (unsigned)p.ldv);
    const unsigned vo = 4u * (unsigned)(4 * hh * p.ldv + min(n, p.ldv - 1));
    const int aslot = (l31 & 3) | (((l31 >> 3) & 3) << 2) | (fg << 4) | (((l31 >> 2) & 1) << 5);      // LDS row of bin 32 fg + l31
    const float* arow0 = &Ws[0][aslot][16 * hh];
    const float* brow0 = &Ws[0][16 * fg + 32 * hh][l31];          // second product: bin (r, hh) of the wave's group sits in row r + 16 fg + 32 hh
    float av[8], bv[KB];
#pragma unroll
    for (int t = 0; t < 8; ++t) av[t] = arow0[t];
    for (int c = 0; c < nchunks; ++c) {
        const int cur = c & 1;
        const bool more = c + 1 < nchunks;
        const float* arow = arow0 + cur * 64 * PW;
        const float* brow = brow0 + cur * 64 * PW;
        // The chunk's instruction order below IS the schedule (scheduling barriers between the small groups keep it), as in the slab kernel.
        __builtin_amdgcn_s_setprio(2);                          // the first product is ONE dependent MFMA chain
        if (more) stage_load(h0, c + 1, 0);
        float v[16];                                            // V is read once, straight from HBM: requested a whole first product ahead
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = DLoad<1>::ld(rV, vo, 4u * (unsigned)((64 * c + 32 * fg + (r & 3) + 8 * (r >> 2)) * p.ldv));
        f32x16 d;
#pragma unroll
        for (int r = 0; r < 16; ++r) d[r] = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (t == NT / 2) {
                __builtin_amdgcn_sched_barrier(0);
                if (more) {
                    stage_store(h0, cur ^ 1, 0);
                    stage_load(h0, c + 1, 1);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            d = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t & 7], hr[t], d, 0, 0, 0);
            if (t + 8 < NT) av[t & 7] = arow[((t + 8) & 15) + 32 * ((t + 8) >> 4)];
            if (t >= NT - KB) bv[t - (NT - KB)] = brow[32 * (t - (NT - KB))];          // second product, bin pair r = 0: atom blocks 0 .. KB - 1
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(0);
        // R in place (0 beyond the last frame); bins are rows of d: (r&3) + 8 (r>>2) + 4 hh
#pragma unroll
        for (int r = 0; r < 16; ++r) d[r] = n < p.N ? direct_div_fast(v[r], d[r]) : 0.f;
        // second product (transposed): numer^T[n][32 ab + j] += sum_f R[f][n] W[f][32 ab + j]
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ab = 0; ab < KB; ++ab) {
                nt[ab] = __builtin_amdgcn_mfma_f32_32x32x2f32(d[r], bv[ab], nt[ab], 0, 0, 0);
                if (r + 1 < 16) bv[ab] = brow[(r + 1) * PW + 32 * ab];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (more) {
            stage_store(h0, cur ^ 1, 1);
            __builtin_amdgcn_sched_barrier(0);
            __syncthreads();
#pragma unroll
            for (int t = 0; t < 8; ++t) av[t] = arow0[(cur ^ 1) * 64 * PW + t];
        }
    }
    if (trace && tid == 0) trace[2] = __builtin_amdgcn_s_memrealtime();
    // The two bin groups of a frame half meet in LDS (the W chunks are done): fg = 1 hands its part over, fg = 0 adds it to its own, applies
    // the tail bin's rank-1 term, the pending scale and the denominator, and rewrites H.  What that step needs from global memory (the old H
    // values: four consecutive frames per lane and step) is requested before the exchange.
    // nt[ab][r]: frame col0 + 32 nh + (r&3) + 8 (r>>2) + 4 hh, atom 32 ab + l31.
    float* red = &Ws[0][0][0];                                   // [nh][ab][r][lane]
    df32x4 h4[KB][4];
    float s_at[KB], wm_at[KB], rd[KB];
    if (fg == 0) {
#pragma unroll
        for (int ab = 0; ab < KB; ++ab) {
            const int arow_ = min(32 * ab + l31, p.Kd - 1);
#pragma unroll
            for (int j = 0; j < 4; ++j) h4[ab][j] = *(const df32x4*)(H + (long)arow_ * p.ldb + col0 + 32 * nh + 8 * j + 4 * hh);
            s_at[ab] = sc[arow_];
            wm_at[ab] = W[(long)p.M * p.lda + arow_];
            rd[ab] = 1.0f / (p.colsum[file * p.sVec + arow_] + p.alpha + p.eps);
        }
    }
    __syncthreads();
    if (fg == 1) {
#pragma unroll
        for (int ab = 0; ab < KB; ++ab)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[((nh * KB + ab) * 16 + r) * 64 + lane] = nt[ab][r];
    }
    __syncthreads();
    if (fg == 0) {
#pragma unroll
        for (int ab = 0; ab < KB; ++ab) {
            const int atom = 32 * ab + l31;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int nl = 32 * nh + 8 * j + 4 * hh, nn = col0 + nl;          // first of the four frames
                df32x4 out;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float u = nt[ab][4 * j + q] + red[((nh * KB + ab) * 16 + 4 * j + q) * 64 + lane];
                    out[q] = nn + q < p.N ? (h4[ab][j][q] * s_at[ab]) * (fmaf(wm_at[ab], s_r[nl + q], u) * rd[ab]) : 0.f;
                }
                if (atom < p.Kd && nn < p.N) *(df32x4*)(H + (long)atom * p.ldb + nn) = out;
            }
        }
    }
    if (trace && tid == 0) trace[3] = __builtin_amdgcn_s_memrealtime();
}

template <int KB>
__global__ __launch_bounds__(256, 2) void gccnmf_wh_updh_kernel(const WhUpdhArgs p) {
    __shared__ __attribute__((aligned(16))) float smem[WhUpdhLds<KB>::FLOATS];
    const int tiles = p.tiles_n;
    int idx = blockIdx.x;
    if (p.xc) {          // XCD x owns a contiguous eighth of the file-major tile list (blocks b, b + 8, ... run on XCD b % 8)
        idx = (blockIdx.x & 7) * p.xc + (blockIdx.x >> 3);
        if (idx >= p.batch * tiles) return;
    }
    const int file = __builtin_amdgcn_readfirstlane(idx / tiles);
    const int tn = __builtin_amdgcn_readfirstlane(idx - file * tiles);
    gccnmf_wh_updh_item<KB>(p, file, tn, smem, p.trace ? p.trace + 8 * (long)blockIdx.x : nullptr, (int)threadIdx.x);
}


int gccnmf_wh_updh_launch(WhUpdhArgs a, hipStream_t stream) {
    if (!a.W || !a.H || !a.V || !a.colsum || !a.scale || a.batch < 1 || a.N < 1) return GCCNMF_ERR_ARG;
    if (a.M < 64 || a.M > 512 || (a.M & 63) || a.Kd < 1 || a.Kd > 128 || (a.lda & 3) || (a.ldb & 3)) return GCCNMF_ERR_UNSUPPORTED;
    a.tiles_n = gccnmf_ceil_div(a.N, 64);
    int grid = a.batch * a.tiles_n;
    a.xc = 0;
    if (a.batch >= 8) {
        a.xc = gccnmf_ceil_div(a.batch * a.tiles_n, 8);          // tiles per XCD
        grid = 8 * a.xc;
    }
    a.trace = (gccnmf_trace_buf && grid <= gccnmf_trace_blocks) ? gccnmf_trace_buf : nullptr;
    const int kb = gccnmf_ceil_div(a.Kd, 32);
    if (kb == 1) hipLaunchKernelGGL(gccnmf_wh_updh_kernel<1>, dim3(grid), dim3(256), 0, stream, a);
    else if (kb == 2) hipLaunchKernelGGL(gccnmf_wh_updh_kernel<2>, dim3(grid), dim3(256), 0, stream, a);
    else if (kb == 3) hipLaunchKernelGGL(gccnmf_wh_updh_kernel<3>, dim3(grid), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(gccnmf_wh_updh_kernel<4>, dim3(grid), dim3(256), 0, stream, a);
    GCCNMF_CHECK_LAUNCH();
    return GCCNMF_OK;
}

// ---- K3 and K4a of one KL-NMF iteration in ONE launch, for short dictionaries (round 4) ---------------------------------------------
//   R = V / (W . H)   then   U = R . H^T (+ row sums of H)                                   (gccNMFFunctions.py:77, :79)
// A workgroup owns a SLAB of 64 bins of one file for the whole launch and walks the file's column tiles (64 frames each):
//   * its rows of W (64 x Kd <= 128) stay in registers for the whole launch (a lane keeps W[f][its 64 atoms]: 64 VGPRs);
//   * the H tile [Kd][64] of the current frames goes through LDS once (double-buffered, register-staged) and is BOTH operand of the
//     first product (A = H^T: a lane reads one float of row k) and of the second (A = H: a lane reads four consecutive frames of atom row);
//   * the first product is computed TRANSPOSED, D'[n][f] = sum_k H[k][n] W[f][k], so that after the divide the accumulator registers are
//     directly the B operand of U^T[k][f] += sum_n H[k][n] R'[n][f] (register r of a 32x32 accumulator = frame pair (n, n + 4));
//   * U^T (Kd x 32 bins per wave) accumulates in registers across all column tiles: R is never written, U is written once.
// Four waves per workgroup: (bin group fg, frame half nh); the two frame halves of a bin group are added at the end (fixed order).
// The tail bin (row M of W and V) and the row sums of H ride along on the VALU: every slab recomputes r[n] = V[M][n] / (W[M] . H[:, n]) for
// the tile's frames (16 frames per wave, the atoms dealt over 4 lane groups) and accumulates U[M][k] and rowsumH[k] for ITS OWN 16 atoms
// -- no reduction across workgroups, no extra launch (needs 16 x slabs >= atoms).
// 64 files x 8 slabs = 512 workgroups: exactly two per CU, one launch instead of two, no 2.7 MB R round trip per file.
// LDS of one workgroup: two H tiles, the tail bin's W row, the tail sums
template <int KB>
struct WhdivRhtLds {
    static constexpr int FLOATS = 2 * 32 * KB * 68 + 32 * KB + 4 * 4 * 4 * 2;
};
// One work item: slab sl (64 bins) of file `file`.
template <int KB>
__device__ __forceinline__ void gccnmf_whdiv_rht_item(const WhdivRhtArgs& p, const int file, const int sl, float* const smem, long long* const trace,
                                                      const int tid_in) {
    constexpr int P = 68, NCH = 2 * KB, KR = 32 * KB;             // LDS pitch | chunks of 16 atoms | atom rows of the H tile
    float (*Hs)[KR][P] = (float (*)[KR][P])smem;                  // [2][KR][P] (16-byte aligned: smem is, P is a multiple of 4)
    float* const s_w = smem + 2 * KR * P;                         // [KR]
    float (*s_tail)[4][4][2] = (float (*)[4][4][2])(s_w + KR);    // [4][4][4][2]
    const int tid = tid_in, lane = tid & 63, l31 = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float* __restrict__ W = p.W + file * p.sW;
    const float* __restrict__ H = p.H + file * p.sH;
    const float* __restrict__ V = p.V + file * p.sV;
    float* __restrict__ U = p.U + file * p.sU;
    const int ntiles = (p.N + 63) >> 6;
    const __amdgpu_buffer_rsrc_t rH = direct_rsrc(H, 4u * (unsigned)p.Kd * (unsigned)p.ldb);          // atom rows >= Kd read as zero
    const __amdgpu_buffer_rsrc_t rV = direct_rsrc(V, 4u * (unsigned)(p.M + 1) * (unsigned)p.ldv);
    // staging: thread (row tid >> 4, float4 tid & 15) moves rows (tid >> 4) + 16 i of the tile, i < 2 KB, in two parts of KB rows each
    const unsigned st_off = 4u * (unsigned)((tid >> 4) * p.ldb + 4 * (tid & 15));
    auto stage_load = [&](df32x4 (&r)[KB], int tile, int part) {
#pragma unroll
        for (int i = 0; i < KB; ++i) r[i] = DLoad<4>::ld(rH, st_off, 4u * (unsigned)(16 * (part * KB + i) * p.ldb + 64 * tile));
    };
    auto stage_store = [&](const df32x4 (&r)[KB], int buf, int part) {
#pragma unroll
        for (int i = 0; i < KB; ++i) *(df32x4*)&Hs[buf][(tid >> 4) + 16 * (part * KB + i)][4 * (tid & 15)] = r[i];
    };
    if (trace && tid == 0) {
        trace[0] = __builtin_amdgcn_s_memrealtime();
        trace[4] = (long long)((__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u) << 16 | (__builtin_amdgcn_s_getreg((31 << 11) | 4) & 0xffffu));
    }
    df32x4 h0[KB], h1[KB];
    stage_load(h0, 0, 0);
    stage_load(h1, 0, 1);

    // ---- bins f0 .. f0 + 63; wave (fg, nh): bins f0 + 32 fg + l31, frames 32 nh .. 32 nh + 31 of every tile
    const int fg = wave & 1, nh = wave >> 1;
    const int f = 64 * sl + 32 * fg + l31;
    // W resident: w[c][t] = W[f][16 c + 8 hh + t]  (MFMA step t of chunk c multiplies atoms (16 c + t, 16 c + 8 + t))
    df32x4 w[NCH][2];
    {
        const __amdgpu_buffer_rsrc_t rW = direct_rsrc(W, 4u * (unsigned)(p.M + 1) * (unsigned)p.lda);
        const unsigned wo = 4u * (unsigned)(f * p.lda + 8 * hh);
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            w[c][0] = DLoad<4>::ld(rW, wo, 64u * c);
            w[c][1] = DLoad<4>::ld(rW, wo, 64u * c + 16u);
        }
        // atoms >= Kd: the H rows are zero (buffer bound), but W beyond the row's Kp columns is the next row: keep the products finite
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int t = 0; t < 8; ++t)
                if (16 * c + 8 * hh + t >= p.lda) w[c][t >> 2][t & 3] = 0.f;
    }
    f32x16 u[KB];
#pragma unroll
    for (int ab = 0; ab < KB; ++ab)
#pragma unroll
        for (int r = 0; r < 16; ++r) u[ab][r] = 0.f;
    stage_store(h0, 0, 0);
    stage_store(h1, 0, 1);
    // tail bin: this wave's 16 frames of a tile (frame nl), atoms dealt over the 4 lane groups kq; this slab's atoms 16 sl + 4 kq + i
    if (tid < KR) s_w[tid] = tid < p.Kd ? W[(long)p.M * p.lda + tid] : 0.f;
    const int nl = 16 * wave + (lane & 15), kq = lane >> 4;
    const int a0 = 16 * sl + 4 * kq;
    const bool tail_atoms = 16 * sl < KR;
    float um[4] = {0.f, 0.f, 0.f, 0.f}, rs[4] = {0.f, 0.f, 0.f, 0.f};
    const unsigned vo = 4u * (unsigned)(f * p.ldv + 32 * nh + 4 * hh);
    float av[8];
    df32x4 a4[KB];
    long long tt1 = 0, tt5 = 0, tt6 = 0, tt7 = 0;
    __syncthreads();
    for (int c = 0; c < ntiles; ++c) {
        const int cur = c & 1;
        const bool more = c + 1 < ntiles;
        // The tile's instruction order below IS the schedule (scheduling barriers between the small groups keep it):
        //   first product  (transposed): d[r] = sum_k H[k][n(r)] W[f][k],  n(r) = 64 c + 32 nh + (r&3) + 8 (r>>2) + 4 hh -- one MFMA, then the
        //     LDS read that refills ITS operand register for the next chunk (7 MFMAs = 450 cycles ahead of its use), four terms of the tail
        //     bin's dot product per chunk on the VALU;
        //   half way: the staged rows of the next H tile go to LDS, the second half and this tile's V values are requested;
        //   second product: U^T[32 ab + i][f] += sum_n H[32 ab + i][n] R'[n][f] -- four MFMAs on one atom block, then the read that refills
        //     that block's operand for the next four frames; the divides of the next four frames ride in the first group.
        { const long long now = __builtin_amdgcn_s_memrealtime(); tt1 = c == 8 ? now : tt1; }      // tile 8: start (scalar registers; written out at the end)
        __builtin_amdgcn_s_setprio(2);                          // the first product is ONE dependent MFMA chain: it yields every other slot anyway
        constexpr int REQ_CH = KB == 4 ? 1 : 0;                 // (always before the half-way store of the same rows)
        float vm = 0.f;
        df32x4 v[4];
        f32x16 d;
#pragma unroll
        for (int r = 0; r < 16; ++r) d[r] = 0.f;
        float tdot = 0.f, tw[4], th[4];
        if (c == 0) {
#pragma unroll
            for (int t = 0; t < 8; ++t) av[t] = Hs[0][8 * hh + t][32 * nh + l31];
        }
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            if (ch == REQ_CH) {
                // the next tile's first staged rows and the tail bin's V are requested one chunk into the tile, not at its top, where the dictionary has 97 .. 128 atoms: this
                // instantiation keeps nine values of the resident W operand in scratch, their reloads sit in the tile's first MFMA groups, and every scratch
                // reload is followed by s_waitcnt vmcnt(0) -- with the requests at the top that wait was for the memory side, once per tile
                __builtin_amdgcn_sched_barrier(0);
                if (more) stage_load(h0, c + 1, 0);
                vm = V[(long)p.M * p.ldv + min(64 * c + nl, p.ldv - 1)];
            }
            if (ch == NCH / 2) {
                __builtin_amdgcn_sched_barrier(0);
                if (more) {
                    stage_store(h0, cur ^ 1, 0);
                    stage_load(h0, c + 1, 1);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = DLoad<4>::ld(rV, vo, 4u * (unsigned)(64 * c + 8 * j));
            }
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                __builtin_amdgcn_sched_barrier(0);
                d = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], w[ch][t >> 2][t & 3], d, 0, 0, 0);
                if (ch + 1 < NCH) av[t] = Hs[cur][16 * (ch + 1) + 8 * hh + t][32 * nh + l31];
                if (t < 4) {                                   // tail bin: operands now, the multiply-add four groups later
                    tw[t] = s_w[kq * (KR / 4) + 4 * ch + t];
                    th[t] = Hs[cur][kq * (KR / 4) + 4 * ch + t][nl];
                } else {
                    tdot = fmaf(tw[t - 4], th[t - 4], tdot);
                }
                if (ch == NCH - 1 && t >= 8 - KB) a4[t - (8 - KB)] = *(const df32x4*)&Hs[cur][32 * (t - (8 - KB)) + l31][32 * nh + 4 * hh];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(0);                          // (the second product's four independent chains would otherwise starve the CU's other workgroup)
        { const long long now = __builtin_amdgcn_s_memrealtime(); tt5 = c == 8 ? now : tt5; }      // first product issued
        // tail bin: r[n] for this wave's 16 frames, then U[M][k] and rowsumH[k] of this slab's atoms (while the last MFMAs drain)
        {
            tdot += __shfl_xor(tdot, 16);
            tdot += __shfl_xor(tdot, 32);
            const float r = 64 * c + nl < p.N ? direct_div_fast(vm, tdot) : 0.f;
            if (tail_atoms) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float hv = Hs[cur][a0 + i][nl];
                    um[i] = fmaf(r, hv, um[i]);
                    rs[i] += hv;
                }
            }
        }
        auto divide4 = [&](int j) {                            // R' in place (0 beyond the last frame)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int n = 64 * c + 32 * nh + q + 8 * j + 4 * hh;
                d[4 * j + q] = n < p.N ? direct_div_fast(v[j][q], d[4 * j + q]) : 0.f;
            }
        };
        divide4(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int ab = 0; ab < KB; ++ab) {
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int q = 0; q < 4; ++q) u[ab] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[ab][q], d[4 * j + q], u[ab], 0, 0, 0);
                if (j + 1 < 4) a4[ab] = *(const df32x4*)&Hs[cur][32 * ab + l31][32 * nh + 8 * (j + 1) + 4 * hh];
                if (ab == 0 && j + 1 < 4) divide4(j + 1);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // the first operands of the next tile's first product (its rows 0 .. 15 were stored half way through this tile... by ALL threads
        // only after the next barrier: so they are read at the top of the next tile instead)
        { const long long now = __builtin_amdgcn_s_memrealtime(); tt6 = c == 8 ? now : tt6; }      // second product issued
        if (more) stage_store(h0, cur ^ 1, 1);
        __builtin_amdgcn_sched_barrier(0);
        if (more) {
            __syncthreads();
            { const long long now = __builtin_amdgcn_s_memrealtime(); tt7 = c == 8 ? now : tt7; }  // past the barrier
#pragma unroll
            for (int t = 0; t < 8; ++t) av[t] = Hs[cur ^ 1][8 * hh + t][32 * nh + l31];
        }
    }
    if (trace && tid == 0) {
        trace[2] = __builtin_amdgcn_s_memrealtime();
        trace[1] = tt1;
        trace[5] = tt5;
        trace[6] = tt6;
        trace[7] = tt7;
    }
    // the two frame halves of a bin group meet in LDS (the H tiles are done): nh = 1 writes, nh = 0 adds and stores U[f][atoms]
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) {
            um[i] += __shfl_xor(um[i], m);
            rs[i] += __shfl_xor(rs[i], m);
        }
        if ((lane & 15) == 0) {
            s_tail[wave][kq][i][0] = um[i];
            s_tail[wave][kq][i][1] = rs[i];
        }
    }
    __syncthreads();
    if (tid < 16 && tail_atoms) {                                // atom 16 sl + tid: the four waves' frames in wave order
        const int q = tid >> 2, i = tid & 3, atom = 16 * sl + tid;
        if (atom < p.ldu) {
            U[(long)p.M * p.ldu + atom] = (s_tail[0][q][i][0] + s_tail[1][q][i][0]) + (s_tail[2][q][i][0] + s_tail[3][q][i][0]);
            p.rowsumH[file * p.sVec + atom] = (s_tail[0][q][i][1] + s_tail[1][q][i][1]) + (s_tail[2][q][i][1] + s_tail[3][q][i][1]);
        }
    }
    float* red = &Hs[0][0][0];                                   // [fg][ab][r][lane]
    if (nh == 1) {
#pragma unroll
        for (int ab = 0; ab < KB; ++ab)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[((fg * KB + ab) * 16 + r) * 64 + lane] = u[ab][r];
    }
    __syncthreads();
    if (nh == 0) {
#pragma unroll
        for (int ab = 0; ab < KB; ++ab)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                df32x4 o;
#pragma unroll
                for (int q = 0; q < 4; ++q) o[q] = u[ab][4 * j + q] + red[((fg * KB + ab) * 16 + 4 * j + q) * 64 + lane];
                const int atom = 32 * ab + 8 * j + 4 * hh;
                if (atom < p.ldu) *(df32x4*)(U + (long)f * p.ldu + atom) = o;
            }
    }
    if (trace && tid == 0) trace[3] = __builtin_amdgcn_s_memrealtime();
}

template <int KB>
__global__ __launch_bounds__(256, 2) void gccnmf_whdiv_rht_kernel(const WhdivRhtArgs p) {
    __shared__ __attribute__((aligned(16))) float smem[WhdivRhtLds<KB>::FLOATS];
    const int per_file = p.nslabs;
    int idx = blockIdx.x;
    if (p.xc) {          // XCD x owns a contiguous eighth of the file-major workgroup list
        idx = (blockIdx.x & 7) * p.xc + (blockIdx.x >> 3);
        if (idx >= p.batch * per_file) return;
    }
    const int file = __builtin_amdgcn_readfirstlane(idx / per_file);
    const int sl = __builtin_amdgcn_readfirstlane(idx - file * per_file);
    gccnmf_whdiv_rht_item<KB>(p, file, sl, smem, p.trace ? p.trace + 8 * (long)blockIdx.x : nullptr, (int)threadIdx.x);
}


int gccnmf_whdiv_rht_launch(WhdivRhtArgs a, hipStream_t stream) {
    if (!a.W || !a.H || !a.V || !a.U || !a.rowsumH || a.batch < 1 || a.N < 1) return GCCNMF_ERR_ARG;
    if (a.M < 64 || a.M > 512 || (a.M & 63) || a.Kd < 1 || a.Kd > 128 || (a.lda & 3) || (a.ldb & 3) || (a.ldv & 3) || (a.ldu & 3))
        return GCCNMF_ERR_UNSUPPORTED;
    a.nslabs = a.M / 64;
    if (16 * a.nslabs < 32 * gccnmf_ceil_div(a.Kd, 32)) return GCCNMF_ERR_UNSUPPORTED;      // every atom's tail-bin sum needs a slab
    const int total = a.batch * a.nslabs;
    int grid = total;
    a.xc = 0;
    if (a.batch >= 8) {
        a.xc = gccnmf_ceil_div(total, 8);
        grid = 8 * a.xc;
    }
    a.trace = (gccnmf_trace_buf && grid <= gccnmf_trace_blocks) ? gccnmf_trace_buf : nullptr;
    const int kb = gccnmf_ceil_div(a.Kd, 32);
    if (kb == 1) hipLaunchKernelGGL(gccnmf_whdiv_rht_kernel<1>, dim3(grid), dim3(256), 0, stream, a);
    else if (kb == 2) hipLaunchKernelGGL(gccnmf_whdiv_rht_kernel<2>, dim3(grid), dim3(256), 0, stream, a);
    else if (kb == 3) hipLaunchKernelGGL(gccnmf_whdiv_rht_kernel<3>, dim3(grid), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(gccnmf_whdiv_rht_kernel<4>, dim3(grid), dim3(256), 0, stream, a);
    GCCNMF_CHECK_LAUNCH();
    return GCCNMF_OK;
}

// ---- the short-dictionary iteration as ONE chained launch (direct.h: ShortChainArgs) ----------------------------------------------------
#include "chain_sync.h"
#include "update_w.h"

template <int KB, int AT>
__global__ __launch_bounds__(256, 2) void gccnmf_short_chain_kernel(const ShortChainArgs c) {
    constexpr int LDS12 = WhUpdhLds<KB>::FLOATS, LDS34 = WhdivRhtLds<KB>::FLOATS;
    __shared__ __attribute__((aligned(16))) float smem[LDS12 > LDS34 ? LDS12 : LDS34];
    static_assert(5 * AT <= LDS12, "");
    const int list = (int)(blockIdx.x & 7);
    int pos = (int)(blockIdx.x >> 3), it = c.it0;
    if (c.iterations > 1) {
        const int q = pos / c.first[3];
        it += q;
        pos -= q * c.first[3];
    }
    const int stage = pos < c.first[1] ? 0 : pos < c.first[2] ? 1 : 2;
    const int r = pos - c.first[stage];
    const int per = c.per_file[stage];
    const int ql = r / per, sub = __builtin_amdgcn_readfirstlane(r - ql * per);
    const int file = __builtin_amdgcn_readfirstlane(list + 8 * ql);
    const int batch = c.a12.batch;
    if (file >= batch) return;
    const int tid = threadIdx.x;
    long long* const tr = (c.trace && it == c.trace_it && 8 * pos + list < c.trace_rows) ? c.trace + 8 * (long)(8 * pos + list) : nullptr;
    // stage s waits for the per-file counter of stage s - 1 (stage 0: for the W update of the PREVIOUS iteration) and signals its own
    GemmSync y = {};
    y.error = c.error;
    y.timeout = c.timeout;
    y.xcc_seen = c.xcc_seen;
    y.sig_cnt = c.counters + stage * batch;
    y.sig_stride = 1;
    y.wait_stride = 1;
    if (stage > 0 || it > 0) {
        y.wait_cnt = c.counters + ((stage + 2) % 3) * batch;
        y.wait_need = (unsigned)c.per_file[(stage + 2) % 3];
        y.wait_lag = stage == 0 ? 1 : 0;
    }
    if (tr && tid == 0) tr[7] = __builtin_amdgcn_s_memrealtime();
    gemm_sync_wait(y, file, 0, tid, it, list, 0);
    if (stage == 0) {
        gccnmf_wh_updh_item<KB>(c.a12, file, sub, smem, tr, tid);
    } else if (stage == 1) {
        gccnmf_whdiv_rht_item<KB>(c.a34, file, sub, smem, tr, tid);
    } else {
        if (tr && tid == 0) tr[0] = __builtin_amdgcn_s_memrealtime();
        nmf_update_w_onepass_item<AT, 1>(c.aw.W, c.aw.U, c.aw.rowsumH, c.aw.colsumW, c.aw.hscale, c.aw.F, c.aw.K, c.aw.Kp, c.aw.sW, c.aw.sU, c.aw.sVec,
                                         c.aw.sRowsum, 0, 0, nullptr, 0, 0, file, sub, smem, tid);
        if (tr && tid == 0) tr[3] = __builtin_amdgcn_s_memrealtime();
    }
    gemm_sync_signal(y, file, 0, 1, tid);
}

int gccnmf_short_chain_launch(ShortChainArgs a, hipStream_t stream) {
    if (!a.counters || !a.error || a.iterations < 1 || a.a12.batch < 8) return GCCNMF_ERR_ARG;
    if (a.a12.M < 64 || a.a12.M > 512 || (a.a12.M & 63) || a.a12.Kd < 1 || a.a12.Kd > 128) return GCCNMF_ERR_UNSUPPORTED;
    a.a12.tiles_n = gccnmf_ceil_div(a.a12.N, 64);
    a.a12.xc = 0; a.a12.trace = nullptr;
    a.a34.nslabs = a.a34.M / 64;
    a.a34.xc = 0; a.a34.trace = nullptr;
    if (16 * a.a34.nslabs < 32 * gccnmf_ceil_div(a.a34.Kd, 32)) return GCCNMF_ERR_UNSUPPORTED;
    if (a.atoms_per_group != 16 && a.atoms_per_group != 32) return GCCNMF_ERR_ARG;
    a.per_file[0] = a.a12.tiles_n;
    a.per_file[1] = a.a34.nslabs;
    a.per_file[2] = a.aw.Kp / a.atoms_per_group;
    const int longest = (a.a12.batch + 7) / 8;
    a.first[0] = 0;
    for (int i = 0; i < 3; ++i) a.first[i + 1] = a.first[i] + longest * a.per_file[i];
    if ((long)8 * a.first[3] * a.iterations > (1L << 30)) return GCCNMF_ERR_ARG;
    a.trace = gccnmf_trace_buf;
    a.trace_rows = gccnmf_trace_buf ? gccnmf_trace_blocks : 0;
    a.trace_it = a.it0 + (a.iterations > 4 ? a.iterations - 3 : a.iterations - 1);      // (timeline builds: a steady-state iteration, not the call's last)
    const int grid = 8 * a.first[3] * a.iterations;
    const unsigned pad = a.solo ? 24576u : 0u;                     // (lab: one workgroup per CU)
    const int kb = gccnmf_ceil_div(a.a12.Kd, 32);
#define GCCNMF_SHORT_CHAIN(KB_, AT_) hipLaunchKernelGGL((gccnmf_short_chain_kernel<KB_, AT_>), dim3(grid), dim3(256), pad, stream, a)
    if (a.atoms_per_group == 32) {
        if (kb == 1) GCCNMF_SHORT_CHAIN(1, 32); else if (kb == 2) GCCNMF_SHORT_CHAIN(2, 32); else if (kb == 3) GCCNMF_SHORT_CHAIN(3, 32); else GCCNMF_SHORT_CHAIN(4, 32);
    } else {
        if (kb == 1) GCCNMF_SHORT_CHAIN(1, 16); else if (kb == 2) GCCNMF_SHORT_CHAIN(2, 16); else if (kb == 3) GCCNMF_SHORT_CHAIN(3, 16); else GCCNMF_SHORT_CHAIN(4, 16);
    }
#undef GCCNMF_SHORT_CHAIN
    GCCNMF_CHECK_LAUNCH();
    return GCCNMF_OK;
}

// ---- transposed copies ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gccnmf_transpose_kernel(const float* __restrict__ in, long s_in, int ld_in, float* __restrict__ out,
                                                               long s_out, int ld_out, int rows, int cols) {
    __shared__ float t[32][33];
    const int b = blockIdx.z, r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float* src = in + b * s_in;
    float* dst = out + b * s_out;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + ty + 8 * i, c = c0 + tx;
        t[ty + 8 * i][tx] = (r < rows && c < cols) ? src[(long)r * ld_in + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + 8 * i, r = r0 + tx;
        if (c < cols && r < rows) dst[(long)c * ld_out + r] = t[tx][ty + 8 * i];
    }
}

int gccnmf_transpose_launch(const float* in, long s_in, int ld_in, float* out, long s_out, int ld_out, int rows, int cols, int batch,
                            hipStream_t stream) {
    if (!in || !out || rows < 1 || cols < 1 || batch < 1) return GCCNMF_ERR_ARG;
    hipLaunchKernelGGL(gccnmf_transpose_kernel, dim3(gccnmf_ceil_div(cols, 32), gccnmf_ceil_div(rows, 32), batch), dim3(256), 0, stream, in, s_in,
                       ld_in, out, s_out, ld_out, rows, cols);
    GCCNMF_CHECK_LAUNCH();
    return GCCNMF_OK;
}
