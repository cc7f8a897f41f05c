// The throughput GEMM tile (512 x 64 per workgroup, 128 x 64 per wave; narrow items 512 x 32) on LDS-DMA:
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
// The lazy H row scale of K1 is applied to the B fragments after the LDS read (fl(H*s), as the register-staged kernel does
// while staging).  Full tiles take the lean epilogues below, ragged ones the generic epilogues of gemm_mfma.h.
//
// Measurement tooling lives here too: gccnmf_debug_set_trace (per-workgroup timeline, scripts/ktrace.py) and the
// GEMM_DMA_PROBE build (per-phase cycle counts of the k-tile, scripts/ktrace.py --probe).
#pragma once
#include <mutex>
#include <type_traits>
#include "gemm_mfma.h"
#include "chain_sync.h"

// build-time timing experiments (results invalid): GEMM_DMA_ABL bit 1 = no fragment reads in the main loop, bit 2 = no split barrier;
// GEMM_DMA_SKIP_FROM = n: only the first n LDS-DMA pieces of a k-tile are issued
#ifndef GEMM_DMA_ABL
#define GEMM_DMA_ABL 0
#endif

typedef __attribute__((address_space(3))) void* gemm_lds_ptr;
typedef __attribute__((address_space(1))) const void* gemm_glb_ptr;

// One LDS-DMA piece: 16 B per lane from (wave-uniform base, SGPR pair) + (unsigned 32-bit per-lane byte offset); lane l lands at
// M0 + 16*l, M0 = LDS byte address of the piece (wave-uniform).  Inline asm: for __builtin_amdgcn_global_load_lds hipcc does
// not select this `saddr + voffset` form -- it adds base + offset on the VALU before every piece and keeps the offsets as
// 64-bit pairs.
// M0: the statement that reads it also writes it, first thing -- so it never depends on what the compiler left there.  It cannot be
// DECLARED as clobbered: M0 is a compiler-reserved register on AMDGPU, a "m0" clobber only draws `warning: inline asm clobber list
// contains reserved registers: m0 ... may not be preserved across the asm statement` and changes nothing (checked with hipcc 7.2; the
// CDNA HIP guide's section on inline asm says the same: "Write M0 in the SAME statement that reads it").  The compiler's own uses of M0
// (LDS-DMA builtins, ds_gws / s_sendmsg, indirect register indexing) re-materialise it immediately before each use for the same reason,
// and these kernels contain none of them; tests/test_host.py::test_lds_dma_statements_set_m0_themselves keeps it that way.
__device__ __forceinline__ void gemm_dma16(const float* wave_uniform_base, unsigned lane_byte_offset, unsigned lds_wave_byte_addr) {
    asm volatile("s_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1"
                 :
                 : "v"(lane_byte_offset), "s"(wave_uniform_base), "s"(lds_wave_byte_addr)
                 : "memory");
}

// A piece fetched by a few lanes only: exec is narrowed to a RUN-TIME wave-uniform lane mask (an SGPR pair) and restored inside the asm, so
// the main loop carries no v_cmp / s_and_saveexec for it, and a mask of 0 makes the piece a no-op without a branch around it -- the
// side chunks of a k-tile (tail row of A, row scale of B) belong to wave 0 only, and a taken scalar branch per k-tile in the other
// three waves is dearer than five scalar instructions in all four.
__device__ __forceinline__ void gemm_dma16_masked(const float* wave_uniform_base, unsigned lane_byte_offset, unsigned lds_wave_byte_addr,
                                                  unsigned long long wave_uniform_mask) {
    unsigned long long saved;
    asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %4\n\ts_mov_b32 m0, %3\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b64 exec, %0"
                 : "=&s"(saved)
                 : "v"(lane_byte_offset), "s"(wave_uniform_base), "s"(lds_wave_byte_addr), "s"(wave_uniform_mask)
                 : "memory");
}

__device__ __forceinline__ int gemm_swz(int row) { return (row >> 2) & 3; }

// The thread index through an asm statement the compiler cannot see through: values derived from it are recomputed where they are
// needed instead of being kept in registers as loop invariants.
__device__ __forceinline__ int gemm_opaque_tid() {
    int t;
    asm volatile("v_mov_b32 %0, %1" : "=v"(t) : "v"(threadIdx.x));
    return t;
}

// MFMA fragment reads as inline asm: the compiler neither sees them as LDS reads (so it does not drain the LDS-DMA queue
// in front of them) nor waits for them -- every use is preceded by gemm_wait_lds() + gemm_tie() on the destination.
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

// "The LDS reads issued so far have landed": an s_waitcnt followed by empty asm statements that take the destination
// registers as in/out operands (asm volatile statements keep their order), so the compiler can neither hoist a use above
// the wait nor copy a register before its data has arrived.  Nothing may touch a destination between the read and its tie.
__device__ __forceinline__ void gemm_wait_lds() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
template <typename T>
__device__ __forceinline__ void gemm_tie(T& v) { asm volatile("" : "+v"(v)); }

// Split workgroup barrier on an LDS counter (gfx950 has only the monolithic s_barrier): a wave ARRIVES (one atomic add, no
// return, exec narrowed to one lane inside the asm) as soon as its own part of the hand-over is done, and WAITS -- later
// in its instruction stream -- until all arrivals of that round are in.  Both are single asm statements (the poll loop
// included), so the compiler sees no control flow and its register allocation of the main loop is untouched.
__device__ __forceinline__ void gemm_barrier_arrive(unsigned counter_byte_addr) {
    unsigned long long saved;
    asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 1\n\tds_add_u32 %1, %2\n\ts_mov_b64 exec, %0"
                 : "=&s"(saved)
                 : "v"(counter_byte_addr), "v"(1u)
                 : "memory");
}
// The wait in two parts: the counter is READ a few MFMAs ahead (no wait), and CHECKED where the hand-over is needed -- in the
// usual case every wave had arrived by the time of the read and the check costs no LDS round trip; otherwise it polls.
__device__ __forceinline__ unsigned gemm_barrier_peek(unsigned counter_byte_addr) {
    unsigned seen;
    asm volatile("ds_read_b32 %0, %1" : "=v"(seen) : "v"(counter_byte_addr) : "memory");
    return seen;
}
__device__ __forceinline__ void gemm_barrier_wait(unsigned counter_byte_addr, unsigned target, unsigned seen) {
    unsigned seen_s;
    asm volatile("s_waitcnt lgkmcnt(0)\n1:\n\tv_readfirstlane_b32 %1, %0\n\ts_cmp_ge_u32 %1, %3\n\ts_cbranch_scc1 2f\n\t"
                 "s_sleep 1\n\tds_read_b32 %0, %2\n\ts_waitcnt lgkmcnt(0)\n\ts_branch 1b\n2:"
                 : "+v"(seen), "=&s"(seen_s)
                 : "v"(counter_byte_addr), "s"(target)
                 : "memory", "scc");
}

// MFMA fragments of one k half (4 consecutive MFMA steps e = 0..3) for NT 32-row tiles; every address is ONE base VGPR +
// an immediate offset.
template <bool KC, int NT, int ROWS>
struct GemmFrag;
// reduction-contiguous operand: base = buffer + 4*(row0*16 + 4*(chunk ^ swz(row0))); rows row0 + 32*m share swz -> offset m*2048
template <int NT, int ROWS>
struct GemmFrag<true, NT, ROWS> {
    gemm_f32x4 v[NT];
    __device__ __forceinline__ void read(unsigned base) {
        v[0] = gemm_lds_read_b128<0 * 2048>(base);
        v[1] = gemm_lds_read_b128<1 * 2048>(base);
        if constexpr (NT == 4) {
            v[2] = gemm_lds_read_b128<2 * 2048>(base);
            v[3] = gemm_lds_read_b128<3 * 2048>(base);
        }
    }
    __device__ __forceinline__ void tie() {
#pragma unroll
        for (int m = 0; m < NT; ++m) gemm_tie(v[m]);
    }
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int m = 0; m < NT; ++m) v[m] = gemm_f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __device__ __forceinline__ float get(int m, int e) const { return v[m][e]; }
    __device__ __forceinline__ void scale(const gemm_f32x4& s) {
#pragma unroll
        for (int m = 0; m < NT; ++m) v[m] *= s;
    }
};
// reduction-strided operand [k][ROWS]: base = buffer + 4*(4*chunk*ROWS + row0); element e of tile m at 4*(e*ROWS + 32*m):
// tiles m, m+1 come from one ds_read2_b32, the e stride goes into the base
template <int NT, int ROWS>
struct GemmFrag<false, NT, ROWS> {
    gemm_f32x2 v[4][NT / 2];
    __device__ __forceinline__ void read(unsigned base) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            v[e][0] = gemm_lds_read2_b32<0, 32>(base + 4 * e * ROWS);
            if constexpr (NT == 4) v[e][1] = gemm_lds_read2_b32<64, 96>(base + 4 * e * ROWS);
        }
    }
    __device__ __forceinline__ void tie() {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int h = 0; h < NT / 2; ++h) gemm_tie(v[e][h]);
    }
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int h = 0; h < NT / 2; ++h) v[e][h] = gemm_f32x2{0.f, 0.f};
    }
    __device__ __forceinline__ float get(int m, int e) const { return v[e][m >> 1][m & 1]; }
    __device__ __forceinline__ void scale(const gemm_f32x4& s) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int h = 0; h < NT / 2; ++h) v[e][h] *= s[e];
    }
};

// ---- lean epilogues -----------------------------------------------------------------------------------------------------
// Beside a co-resident workgroup that is in its main loop, every instruction of an epilogue costs several times its
// stand-alone issue time (measured with the per-workgroup timeline, scripts/ktrace.py: the W.H epilogue takes 13 us when
// the CU is otherwise idle and 36 us next to a main loop -- with or without its loads and stores -- and the main loop
// next to it drops to 65 % of the matrix-pipe rate).  So the epilogues of FULL tiles (all 32 rows < M, all 64 columns < N:
// every tile of the benchmark shape but the last column tile of a file) are written for instruction count:
//   * rows are addressed through a wave-uniform row pointer (SALU arithmetic, SGPR base of the global access) plus ONE
//     per-lane element offset: no per-element 64-bit VALU address arithmetic, clamps or predicates;
//   * x / d with a divisor that is constant per lane or per row multiplies by the correctly rounded 1/d (<= 1 ulp from
//     the IEEE quotient); V / (W.H) is v_rcp_f32 + one Newton correction through the exact fma residual (4 instructions
//     instead of the ~10 of the IEEE sequence; correctly rounded except for rare 1-ulp cases, no denormal/overflow fix-up).
// Ragged tiles take the generic gemm_epilogue_pair / gemm_epilogue_update_w of gemm_mfma.h.
__device__ __forceinline__ float gemm_div_fast(float v, float d) {
    const float r = __builtin_amdgcn_rcpf(d);
    const float q = v * r;
    return fmaf(fmaf(-d, q, v), r, q);
}

// Raw buffer accesses: address = descriptor base (SGPRs) + soffset (SGPR, the row) + voffset (VGPR, the lane) + immediate.
typedef int gemm_i32x4 __attribute__((ext_vector_type(4)));
__device__ float gemm_buffer_load(gemm_i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.f32");
__device__ gemm_f32x4 gemm_buffer_load4(gemm_i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4f32");
__device__ void gemm_buffer_store(float data, gemm_i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.store.f32");
// bytes: extent of the addressed matrix; accesses beyond it (the padding columns of a ragged last tile when the row pitch
// is not padded to the tile) read 0 / are dropped by the hardware range check
__device__ __forceinline__ gemm_i32x4 gemm_buffer_rsrc(const void* wave_uniform_base, long bytes = 0x7fffffffL) {
    const unsigned long long a = (unsigned long long)wave_uniform_base;
    gemm_i32x4 r;
    r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32) & 0xffff);     // stride 0
    r.z = __builtin_amdgcn_readfirstlane((int)(bytes < 0x7fffffffL ? bytes : 0x7fffffffL));   // num_records (bytes)
    r.w = 0x00020000;                                                            // gfx9 raw dword buffer
    return r;
}

// One tile pair's epilogue inputs, loaded by load() and consumed by finish(): two instances alternate so that the loads of
// pair m+1 are in flight while pair m is computed and stored (one exposed memory round trip per wave instead of four).
// Per-row factors (EPI_UPDH: scale / (column sum + alpha + eps); the rank-1 tail column) are NOT loaded here: the
// workgroup puts them into LDS once, before its main loop (s_rowvec), and finish() reads them as 16-byte groups.
template <int EPI, int BM = 512>
struct GemmEpiloguePair {
    static_assert(EPI == EPI_STORE || EPI == EPI_DIV || EPI == EPI_UPDH, "");
    float xa[16], xb[16];          // EPI_DIV: V; EPI_UPDH: old H

    // row_u: first row of the pair (wave-uniform); lanes hold rows row_u + 8g + 4hh + (0..3), g = 0..3, columns col_a, col_a + 32
    // cb: bytes of one file's V / C matrix (descriptor range)
    __device__ __forceinline__ void load(const GemmArgs& p, int file, int row_u, int hh, int col_a, long cb) {
        if (EPI == EPI_DIV || EPI == EPI_UPDH) {
            const int lo = 4 * (4 * hh * p.ldc + col_a);           // lane offset (bytes) from the wave-uniform row start
            const int ro = 4 * row_u * p.ldc, rstep = 4 * p.ldc;   // row offsets (bytes) inside one file: < 2^31 for any sane size
            const gemm_i32x4 X = gemm_buffer_rsrc(EPI == EPI_DIV ? p.E0 + file * p.sE0 : p.C + file * p.sC, cb);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int so = ro + ((r & 3) + 8 * (r >> 2)) * rstep;
                xa[r] = gemm_buffer_load(X, lo, so, 0);
                xb[r] = gemm_buffer_load(X, lo + 128, so, 0);
            }
        }
    }

    // s_rowvec: [0, BM) = tail column of A (0 without a rank-1 tail), [BM, 2 BM) = EPI_UPDH row factor, indexed by the row
    // inside the workgroup tile; tile_row = row_u - (first row of the workgroup tile)
    // oka / okb: this lane's column (col_a, col_a + 32) is inside N -- the only predicate of the lean path, two exec-mask
    // regions per pair (the last column tile of a file is ragged: N = 1244 = 19 x 64 + 28)
    __device__ __forceinline__ void finish(const GemmArgs& p, int file, int row_u, int tile_row, int hh, int col_a, float ba, float bb,
                                           const float* s_rowvec, const f32x16& acc_a, const f32x16& acc_b, bool oka, bool okb, long cb) const {
        const int lo = 4 * (4 * hh * p.ldc + col_a);
        const int ro = 4 * row_u * p.ldc, rstep = 4 * p.ldc;
        const gemm_i32x4 C = gemm_buffer_rsrc(p.C + file * p.sC, cb);
        // One column block at a time: its 16 values first -- independent chains the scheduler interleaves (an IEEE division is an
        // 11-instruction dependent chain: computed one element at a time in front of its store, the epilogue took 28 us instead of 17
        // beside a neighbour's main loop, profiles/r05e_ktrace_*; all 32 at once spill) -- then its 16 stores as one burst.  The division
        // mode is ONE wave-uniform branch around two unrolled loops, not a branch per element.
        auto column = [&](const f32x16& acc, const float (&x)[16], const float b, const int lane_off) {
            float u[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) u[r] = acc[r];
            if (EPI == EPI_STORE || EPI == EPI_UPDH) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const gemm_f32x4 tav = *(const gemm_f32x4*)(s_rowvec + tile_row + 8 * g + 4 * hh);
                    gemm_f32x4 gv = {1.f, 1.f, 1.f, 1.f};
                    if (EPI == EPI_UPDH) gv = *(const gemm_f32x4*)(s_rowvec + BM + tile_row + 8 * g + 4 * hh);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int r = 4 * g + i;
                        u[r] = fmaf(tav[i], b, u[r]);          // last reduction index as one fmaf per element, in chain order (the final k)
                        if (EPI == EPI_UPDH) u[r] = (x[r] * u[r]) * gv[i];
                    }
                }
            }
            if (EPI == EPI_DIV) {
                if (p.exact_div) {          // tuning key 7 (default): the IEEE quotient, as the reference's numpy.divide
#pragma unroll
                    for (int r = 0; r < 16; ++r) u[r] = x[r] / u[r];
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) u[r] = gemm_div_fast(x[r], u[r]);
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) gemm_buffer_store(u[r], C, lane_off, ro + ((r & 3) + 8 * (r >> 2)) * rstep, 0);
        };
        if (oka) column(acc_a, xa, ba, lo);
        if (okb) column(acc_b, xb, bb, lo + 128);
    }
};

// EPI_UPDW for full tiles (every wave entirely inside or entirely outside M, all 64 atoms < N): gemm_epilogue_update_w of
// gemm_mfma.h with row-pointer addressing and the two per-atom divisors (row sum of H, atom norm) turned into one IEEE
// reciprocal per lane each.
template <bool TAIL>
__device__ __forceinline__ void gemm_epilogue_update_w_full(const GemmArgs& p, int file, int col0, int tid, int wm, int l31, int hh,
                                                            bool wave_active, f32x16 (&acc)[4][2], float tail_acc, float rowsum_acc,
                                                            float* smem) {
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
    const float ira = 1.0f / s_rs[ca], irb = 1.0f / s_rs[cb];
    const gemm_i32x4 W = gemm_buffer_rsrc(p.C + file * p.sC);
    const int lo = 4 * (4 * hh * p.ldc + col0 + l31), rstep = 4 * p.ldc;
    float ssa = 0.f, ssb = 0.f;
    if (wave_active) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int ro = (wm * 128 + m * 32) * rstep;
            float wa[16], wb[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int so = ro + ((r & 3) + 8 * (r >> 2)) * rstep;
                wa[r] = gemm_buffer_load(W, lo, so, 0);
                wb[r] = gemm_buffer_load(W, lo + 128, so, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float ta = wa[r] * (acc[m][0][r] * ira), tb = wb[r] * (acc[m][1][r] * irb);
                acc[m][0][r] = ta;
                acc[m][1][r] = tb;
                ssa = fmaf(ta, ta, ssa);
                ssb = fmaf(tb, tb, ssb);
            }
        }
    }
    ssa += __shfl_xor(ssa, 32);
    ssb += __shfl_xor(ssb, 32);
    if (hh == 0) {
        s_red[wm * 64 + ca] = ssa;
        s_red[wm * 64 + cb] = ssb;
    }
    float wt_tail = 0.f;
    float* Wp = p.C + file * p.sC;
    if (tid < 64) {
        if (TAIL) {
            const float u = (s_tail[tid] + s_tail[64 + tid]) + (s_tail[128 + tid] + s_tail[192 + tid]);
            wt_tail = Wp[(long)p.tail_row * p.ldc + col0 + tid] * (u / s_rs[tid]);
        }
        s_red[256 + tid] = wt_tail * wt_tail;
    }
    __syncthreads();
    if (tid < 64) s_norm[tid] = sqrtf(((s_red[tid] + s_red[64 + tid]) + (s_red[128 + tid] + s_red[192 + tid])) + s_red[256 + tid]);
    __syncthreads();
    const float ina = 1.0f / s_norm[ca], inb = 1.0f / s_norm[cb];
    float csa = 0.f, csb = 0.f;
    if (wave_active) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int ro = (wm * 128 + m * 32) * rstep;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int so = ro + ((r & 3) + 8 * (r >> 2)) * rstep;
                const float na = acc[m][0][r] * ina, nb = acc[m][1][r] * inb;
                gemm_buffer_store(na, W, lo, so, 0);
                gemm_buffer_store(nb, W, lo + 128, so, 0);
                csa += na;
                csb += nb;
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
        if (TAIL) {
            wn_tail = wt_tail / s_norm[tid];
            Wp[(long)p.tail_row * p.ldc + col0 + tid] = wn_tail;
        }
        s_red[256 + tid] = wn_tail;
    }
    __syncthreads();
    if (tid < 64) {
        p.out_colsum[file * p.s_out + col0 + tid] = ((s_red[tid] + s_red[64 + tid]) + (s_red[128 + tid] + s_red[192 + tid])) + s_red[256 + tid];
        p.out_norm[file * p.s_out + col0 + tid] = s_norm[tid];
    }
}

// ---- the work list of a launch ---------------------------------------------------------------------------------------------
// A launch is an ORDERED list of work items per XCD (GemmArgs.lists = 8: XCD x = blockIdx & 7 serves list x; 1: one list), taken in
// order by the hardware dispatcher (workgroup b takes item b >> 3 of list b & 7; the experiment build can also hand them out to
// resident workgroups through a ticket counter).  An item is one output tile, computed by ONE workgroup in the fixed k order, so a
// file's bits depend neither on who takes an item nor on the form it has.  Three kinds, longest first:
//   wide    512 x 64   items 0 .. len - split - 1 of a list: its chunk [list * cw, ...) of the file-major list of wide tiles
//                      (files in the class order 0, 8, 16, ... | 1, 9, ... so that a file's tiles share one XCD's L2)
//   halves  512 x 32   the last `split` wide tiles of the chunk as two NARROW items each, left / right 32 columns (tuning key 9 = 2:
//                      every tile -- a test form; choosing `split` by a list-scheduling model lost to half-height tiles, LABBOOK R5.2)
//   ragged  512 x 32   the last column tile of a file when at most 32 of its 64 columns exist (N = 1244 = 19 x 64 + 28): half the
//                      matrix work of the padded tile it replaces (rag = 1; chunk [list * cr, ...) of the file-major list of these tiles)
// The short items come LAST: the dispatcher hands blocks out in order across the eight XCDs, so a slot freed early in the middle of a
// launch cannot take its XCD's next block while an earlier block waits elsewhere, and the co-resident pairs fall out of step -- the
// ragged tile in its natural place costs 10 % of the launch (LABBOOK R5.3b).
// A narrow item runs the same loop without the MFMAs of the right column block: same k order per element -> same bits.
__host__ __device__ __forceinline__ int gemm_dma_list_file(const GemmArgs& p, int q) {      // position q in the class order -> file
    if (p.lists != 8) return q;
    const int n_full = p.batch >> 3, rem = p.batch & 7, big = rem * (n_full + 1);
    const int cls = q < big ? q / (n_full + 1) : rem + (q - big) / n_full;
    return cls + 8 * (q < big ? q - cls * (n_full + 1) : (q - big) - (cls - rem) * n_full);
}
__host__ __device__ __forceinline__ bool gemm_dma_item(const GemmArgs& p, int list, int t, int& file, int& tm, int& col0, int& nw) {
    if (p.ragged_n) {
        // a ragged batch: the list names its files (balanced by the host over the eight lists); a file contributes tiles_m x its own column tiles
        // (K1 - K3), or the uniform tiles_m x tiles_n atom tiles with its own reduction length (K4).  Same order inside a file as below.
        const int* L = p.ragged_lists + list * (GEMM_RAGGED_LMAX + 1);
        const int cnt = L[0];
        int acc = 0;
        for (int k = 0; k < cnt; ++k) {
            const int f = L[1 + k];
            const int n = p.ragged_kd ? p.N : p.ragged_n[f];
            const int tn = (n + 63) >> 6, per = p.tiles_m * tn;
            if (t < acc + per) {
                const int w = t - acc;
                const int wide_n = tn - (gemm_dma_file_rag(p.narrow_ok, n) ? 1 : 0), wt = p.tiles_m * wide_n;
                if (w < wt) {
                    tm = w / wide_n;
                    col0 = (w - tm * wide_n) * 64;
                    nw = 2;
                } else {
                    tm = w - wt;
                    col0 = (tn - 1) * 64;
                    nw = 1;
                }
                file = f + p.file0;
                return true;
            }
            acc += per;
        }
        return false;
    }
    const int wt = p.tiles_m * p.wide_n;                      // wide tiles per file
    if (p.whole_files == 2) {
        // chained launches with agent-scope hand-over: the same file-major order (a file's wide tiles, then its ragged items), cut into equal eighths --
        // a file may straddle two lists, any batch size balances
        const int per = p.tiles_m * p.tiles_n;
        const long idx = (long)list * p.cw + t;
        if (t >= p.cw || idx >= (long)p.batch * per) return false;
        const int q = (int)(idx / per), w = (int)(idx - (long)q * per);
        if (w < wt) {
            tm = w / p.wide_n;
            col0 = (w - tm * p.wide_n) * 64;
            nw = 2;
        } else {
            tm = w - wt;
            col0 = (p.tiles_n - 1) * 64;
            nw = 1;
        }
        file = gemm_dma_list_file(p, q) + p.file0;
        return true;
    }
    if (p.whole_files) {
        // chained launches: every producer and consumer of a file must run on ONE XCD in every GEMM of the iteration, whatever the batch
        // size: list x holds the files x, x + 8, ... whole, each file's tm-major wide tiles followed by its ragged items.  (A plain
        // launch cuts the file-major list into equal eighths and keeps the short items at the end of each list: LABBOOK R5.3b.  A consumer
        // that needs the WHOLE file -- K4 after K3 -- would wait for an item at the very end of the producer's list.)
        const int per = p.tiles_m * p.tiles_n;
        const int ql = t / per, w = t - ql * per;
        const int fi = list + 8 * ql;
        if (fi >= p.batch) return false;
        if (w < wt) {
            tm = w / p.wide_n;
            col0 = (w - tm * p.wide_n) * 64;
            nw = 2;
        } else {
            tm = w - wt;
            col0 = (p.tiles_n - 1) * 64;
            nw = 1;
        }
        file = fi + p.file0;
        return true;
    }
    const int base_w = list * p.cw;
    const int len = min(max(p.batch * wt - base_w, 0), p.cw), s = min(p.split, len);
    int q;
    if (t < len + s) {
        int idx, half = 0;
        if (t < len - s) {
            idx = base_w + t;
            nw = 2;
        } else {
            const int j = t - (len - s);
            idx = base_w + (len - s) + (j >> 1);
            half = j & 1;
            nw = 1;
        }
        q = idx / wt;
        const int w = idx - q * wt;
        tm = w / p.wide_n;
        col0 = (w - tm * p.wide_n) * 64 + 32 * half;
    } else {
        const int r = t - (len + s), base_r = list * p.cr;
        const int lenr = min(max(p.batch * p.rag * p.tiles_m - base_r, 0), p.cr);
        if (r >= lenr) return false;
        const int idx = base_r + r;
        q = idx / p.tiles_m;
        tm = idx - q * p.tiles_m;
        col0 = (p.tiles_n - 1) * 64;
        nw = 1;
    }
    file = gemm_dma_list_file(p, q) + p.file0;
    return true;
}

// The throughput tile: 512 x 64 per workgroup, 128 x 64 per wave (TM = 4 MFMA row tiles x 2 column blocks).
// NARROW: the instantiation also carries the 512 x 32 form of the main loop (items with nw = 1).
// PERSISTENT grid (p.tickets != nullptr): 64 resident workgroups per list (two per CU) pull items through a ticket counter until the list
// is empty -- work-conserving whatever the items cost; the ticket of the NEXT item is taken at the start of the current one, and the
// next tile's first LDS-DMA pieces are issued BEFORE the current tile's epilogue (p.prefetch), so a workgroup's matrix pipe does not
// wait for a prologue between two tiles.  The last workgroup to leave resets the counters (the next launch on the stream finds zeros).
// LDS of one workgroup of the throughput tile: two staging buffers (A | B | A tail-row chunk | B row-scale chunk), the lean epilogue's row
// factors, the split barrier's counter, the persistent grid's ticket slots
template <int TM>
struct GemmDmaLds {
    static constexpr int SBUF = 128 * TM * 16 + 64 * 16 + 16 + 16;
};

// One workgroup of the throughput tile: items t, (persistent grid: further tickets) of list `list`.
template <bool A_KC, bool B_KC, int EPI, bool TAIL, int TM, bool NARROW>
__device__ __forceinline__ void gemm_dma_workgroup(const GemmArgs& p, float* const smem, float* const s_rowvec, unsigned* const s_arrivals_p,
                                                   int* const s_ticket, const int list, int t, const GemmSync& sync, const int sync_it = 0,
                                                   const bool trace_on = true) {
    static_assert(TM == 4 || TM == 2, "");
    static_assert(TM == 4 || EPI != EPI_UPDW, "the fused W update owns all rows of its atoms: full-height tiles only");
    static_assert(!NARROW || (EPI != EPI_UPDW && TM == 4), "narrow items: full-height tiles with an element-wise epilogue");
    constexpr int BK = 16, BM = 128 * TM, BN = 64, RW = 32 * TM;      // RW: rows per wave
    constexpr int NA = 2 * TM;                                         // 1 KB LDS-DMA pieces of the A tile per wave
    constexpr int SA = BM * BK, SB = BN * BK;
    constexpr int SBUF = SA + SB + BK + BK;          // A | B | A tail-row chunk | B row-scale chunk
    static_assert(SBUF == GemmDmaLds<TM>::SBUF, "");
    constexpr bool SCALE = !B_KC && (EPI == EPI_DIV || EPI == EPI_STORE);   // K1 (R = V / (W.(s*H))) carries a lazy row scale on its B operand
    unsigned& s_arrivals = *s_arrivals_p;            // split barrier of the main loop: 4 arrivals per k-tile

    // Per-lane values are RE-DERIVED per item from an opaque copy of the thread index (gemm_opaque_tid): as loop invariants of the item
    // loop they would stay live across the epilogue, whose 128 accumulators + two tile pairs of inputs leave no registers for them.
    int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform in an SGPR: LDS-DMA bases derive from it
    const int wm = wave, wn = 0;
    int l31 = lane & 31, hh = lane >> 5;
#ifdef GCCNMF_EXPERIMENTS
    const bool persistent = p.tickets != nullptr;
#else
    constexpr bool persistent = false;          // the resident-workgroup form is an experiment build (make EXPERIMENTS=1): measured slower, see LABBOOK.md
#endif

    // ---- per-item state (set by take_item) ----
    int nkt = (p.Kd + BK - 1) / BK;                 // (per item in a ragged K4: the file's own reduction length)
    int n_item = p.N;                               // columns of the item's file (per item in a ragged K1 - K3)
    int n_sync = p.N;                               // the file's column count in the GEMMs that hand over per column tile (what a consumer's wait is sized by)
    int file = 0, tm = 0, col0 = 0, nw = 2, row0 = 0;
    const float* __restrict__ A = nullptr;
    const float* __restrict__ B = nullptr;
    const float* __restrict__ bscale = nullptr;
    const float* __restrict__ tail_src = nullptr;
    const float* __restrict__ scale_src = nullptr;
    unsigned long long tail_mask = 0, scale_mask = 0;                 // lane masks of wave 0's side-chunk pieces (0 = no-op)
    bool wave_active = false, do_tail = false, do_rowsum = false;
    unsigned offA[NA], offB = 0;                                       // per-lane source offsets (bytes from the wave-uniform k-tile origin) of this wave's DMA pieces
    auto take_item = [&](const int ticket) -> bool {
        int f_, tm_, c_, nw_;
        if (!gemm_dma_item(p, list, ticket, f_, tm_, c_, nw_)) return false;
        // (the integer divisions of the decode run on the VALU: without the readfirstlanes every value derived from file / tile -- all
        // row pointers, descriptors and scalar offsets below -- stays in VGPRs and each buffer access becomes a waterfall loop)
        file = __builtin_amdgcn_readfirstlane(f_);
        tm = __builtin_amdgcn_readfirstlane(tm_);
        col0 = __builtin_amdgcn_readfirstlane(c_);
        nw = NARROW ? __builtin_amdgcn_readfirstlane(nw_) : 2;
        if (p.ragged_n) {
            const int nf = __builtin_amdgcn_readfirstlane(p.ragged_n[file]);
            n_sync = nf;
            if (p.ragged_kd) nkt = (nf + BK - 1) / BK;
            else n_item = nf;
        }
        row0 = tm * BM;
        A = p.A + file * p.sA;
        B = p.B + file * p.sB;
        bscale = (SCALE && p.bscale) ? p.bscale + file * p.s_bscale : nullptr;
        wave_active = (row0 + wm * RW) < p.M;
        do_tail = TAIL && (tm == 0);
        do_rowsum = B_KC && (p.rowsumB != nullptr || EPI == EPI_UPDW) && (tm == 0);
        tail_src = A + (long)p.tail_row * p.lda;
        tail_mask = (wave == 0 && do_tail) ? 0x0full : 0ull;
        scale_mask = (wave == 0 && bscale != nullptr) ? 0xf0ull : 0ull;
        scale_src = bscale != nullptr ? bscale : A;              // (never dereferenced under a zero mask; kept a valid address anyway)
        return true;
    };
    auto set_offsets = [&](const int lane) {                           // this wave's DMA source offsets for the current item
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int piece = wave * NA + i;                               // 1 KB pieces of the A tile
            if (A_KC) {
                const int row = piece * 16 + (lane >> 2), c = (lane & 3) ^ gemm_swz(piece * 16 + (lane >> 2));
                offA[i] = 4u * (unsigned)(min(row0 + row, p.a_clamp) * p.lda + 4 * c);
            } else {
                constexpr int PPR = BM / 256;                              // pieces per k row of the [16][BM] image
                const int kk = piece / PPR, col = (piece % PPR) * 256 + lane * 4;
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
    };

    f32x16 acc[TM][2];
    float tail_acc = 0.f, rowsum_acc = 0.f;

    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(gemm_lds_ptr)smem);
    // the two 64-byte side chunks (tail row of A, row scale of B) sit next to each other behind the B tile: ONE piece of wave 0,
    // lanes 0-3 fetch the tail chunk, lanes 4-7 the scale chunk -- both land at (tail chunk) + 16 * lane
    unsigned side_off = 0;

    // One 1 KB LDS-DMA piece of tile kt into staging buffer `buf`: 0 .. NA-1 = this wave's share of A, NA = of B, NA+1 = the tail row
    // chunk / the row-scale chunk.  (Dealt out between the MFMAs of the main loop: SPREAD below.)
    auto dma_piece = [&](const int piece, const int kt, const int buf) {
#ifdef GEMM_DMA_SKIP_FROM
        if (piece >= GEMM_DMA_SKIP_FROM) return;                    // timing experiment (results invalid): fewer pieces per k-tile
#endif
        if (piece < NA) {
            gemm_dma16(A + (A_KC ? (long)kt * BK : (long)kt * BK * p.lda), offA[piece], lds0 + 4 * (buf * SBUF + (wave * NA + piece) * 256));
        } else if (piece == NA) {
            gemm_dma16(B + (B_KC ? (long)kt * BK : (long)kt * BK * p.ldb), offB, lds0 + 4 * (buf * SBUF + SA + wave * 256));
        } else {
            if (TAIL || SCALE) {                                   // wave 0's lanes 0-3 / 4-7; every other wave (and a tile without the chunk): mask 0
                const unsigned dst = lds0 + 4 * (buf * SBUF + SA + SB);
                if (TAIL) gemm_dma16_masked(tail_src + kt * BK, side_off, dst, tail_mask);
                if (SCALE) gemm_dma16_masked(scale_src + kt * BK, side_off, dst, scale_mask);
            }
        }
    };
    constexpr int NPIECES = NA + 2;
    // how the pieces are dealt out over the 16 MFMA pairs of group 0 -- 0: all before the first MFMA; 1: one per pair;
    // 2: two per three pairs.  While every piece carried a 64-bit VALU address add this mattered (W.H 0.816 / 0.698 / 0.699 ms,
    // R.H^T 0.777 / 0.779 / 0.750); with the saddr form the three are within 2 % (0.664 / 0.654 / 0.654, 0.644 / 0.624 / 0.624).
#ifdef GEMM_DMA_SPREAD
    constexpr int SPREAD = GEMM_DMA_SPREAD;
#else
    constexpr int SPREAD = (A_KC && !B_KC) ? 1 : 2;
#endif

    // Fragment registers.  Group 0 = first k half (q = 0) of a tile + its row-scale chunk; group 1 = second half + the
    // operands of the VALU side work (tail row, row sums).  All LDS reads are inline asm (the compiler would drain the
    // LDS-DMA queue, s_waitcnt vmcnt(0), in front of any ds_read it can see after a global_load_lds); every group is
    // made visible by wait_group0/1(): gemm_wait_lds() + gemm_tie() of every destination, so no use can move above the wait.
    const gemm_f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    GemmFrag<A_KC, TM, BM> a0, a1;
    GemmFrag<B_KC, 2, BN> b0, b1;
    gemm_f32x4 sc0 = zero4, sc1 = zero4, t4 = zero4, tb4 = zero4, ts4 = zero4, rs4 = zero4;
    gemm_f32x2 tbxy = {0.f, 0.f}, tbzw = {0.f, 0.f};

    const unsigned arrivals_addr = (unsigned)(size_t)(gemm_lds_ptr)s_arrivals_p;
    // per-lane byte offsets inside a staging buffer (q = 0 / q = 1 chunk of this lane half)
    unsigned oA0 = 0, oA1 = 0, oB0 = 0, oB1 = 0, oTB = 0, oRS = 0;
    const int tg = wave;                                           // tail row: 4 thread groups (= waves) x 4 reduction steps each
    auto set_lane_constants = [&](const int tid_) {
        tid = tid_;
        lane = tid_ & 63;
        l31 = lane & 31;
        hh = lane >> 5;
        const int arow = wm * RW + l31, bcol = wn * 64 + l31, tj = lane;
        side_off = 16u * (unsigned)(lane & 3);
        oA0 = A_KC ? 4 * (arow * 16 + 4 * ((0 + hh) ^ gemm_swz(arow))) : 4 * (4 * (0 + hh) * BM + arow);
        oA1 = A_KC ? 4 * (arow * 16 + 4 * ((2 + hh) ^ gemm_swz(arow))) : 4 * (4 * (2 + hh) * BM + arow);
        oB0 = 4 * SA + (B_KC ? 4 * (bcol * 16 + 4 * ((0 + hh) ^ gemm_swz(bcol))) : 4 * (4 * (0 + hh) * BN + bcol));
        oB1 = 4 * SA + (B_KC ? 4 * (bcol * 16 + 4 * ((2 + hh) ^ gemm_swz(bcol))) : 4 * (4 * (2 + hh) * BN + bcol));
        oTB = 4 * SA + (B_KC ? 4 * (tj * 16 + 4 * (tg ^ gemm_swz(tj))) : 4 * (4 * tg * BN + tj));
        oRS = 4 * SA + 4 * ((tid_ >> 2) * 16 + 4 * (tid_ & 3));   // row sums: 4 threads per atom row, one chunk each
    };

    auto read_group0 = [&](const unsigned lds) {                    // lds = byte address of the staging buffer
        if (wave_active) {
            a0.read(lds + oA0);
            b0.read(lds + oB0);
        }
        if (SCALE) sc0 = gemm_lds_read_b128<4 * (SA + SB + BK)>(lds + 16 * hh);
    };
    auto read_group1 = [&](const unsigned lds) {
        if (wave_active) {
            a1.read(lds + oA1);
            b1.read(lds + oB1);
        }
        if (SCALE) sc1 = gemm_lds_read_b128<4 * (SA + SB + BK) + 32>(lds + 16 * hh);
        // (tail row / row sums: read and accumulated by every workgroup of an instantiation that has them -- a tile that does not need
        // them ignores the result; no scalar branch per k-tile)
        if (TAIL) {
            t4 = gemm_lds_read_b128<4 * (SA + SB)>(lds + 16 * tg);
            if (B_KC) {
                tb4 = gemm_lds_read_b128<0>(lds + oTB);
            } else {
                tbxy = gemm_lds_read2_b32<0 * BN, 1 * BN>(lds + oTB);      // rows 4g, 4g+1 of column j (BN dwords apart)
                tbzw = gemm_lds_read2_b32<2 * BN, 3 * BN>(lds + oTB);
                if (SCALE) ts4 = gemm_lds_read_b128<4 * (SA + SB + BK)>(lds + 16 * tg);
            }
        }
        if (B_KC) rs4 = gemm_lds_read_b128<0>(lds + oRS);
    };
    auto wait_group0 = [&]() {
        gemm_wait_lds();
        a0.tie();
        b0.tie();
        if (SCALE) {
            gemm_tie(sc0);
            b0.scale(sc0);                                         // lazy H row scale: fl(H * s), as the staged form does (1.0 without one)
        }
    };
    auto wait_group1 = [&]() {
        gemm_wait_lds();
        a1.tie();
        b1.tie();
        if (SCALE) {
            gemm_tie(sc1);
            b1.scale(sc1);
        }
        if (TAIL) {
            gemm_tie(t4);
            if (B_KC) {
                gemm_tie(tb4);
            } else {
                gemm_tie(tbxy);
                gemm_tie(tbzw);
                tb4 = gemm_f32x4{tbxy.x, tbxy.y, tbzw.x, tbzw.y};
                if (SCALE) {
                    gemm_tie(ts4);
                    tb4 *= ts4;
                }
            }
        }
        if (B_KC) gemm_tie(rs4);
    };
    // NW = column blocks of the item: 2 = the 512 x 64 tile, 1 = a narrow (512 x 32) item -- the right block's MFMAs are not issued
    auto mma = [&](auto nw_c, const GemmFrag<A_KC, TM, BM>& a, const GemmFrag<B_KC, 2, BN>& b, const int e, const int m) {
        constexpr int NW = decltype(nw_c)::value;
        acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.get(m, e), b.get(0, e), acc[m][0], 0, 0, 0);
        if constexpr (NW == 2) acc[m][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.get(m, e), b.get(1, e), acc[m][1], 0, 0, 0);
    };

    // One k-tile, software-pipelined ACROSS the workgroup barrier so that no wave has an MFMA-free stretch per tile (two
    // co-resident workgroups fall into lock-step -- the one behind gets the whole matrix pipe whenever the leader is busy
    // with something else and catches up -- so any such stretch is matrix-pipe idle time on the whole SIMD):
    //     issue reads of group 1 (tile t)                         | buffer CUR
    //     32 MFMAs of group 0, the 10 LDS-DMA pieces of tile t+1 dealt out one per two MFMAs   | -> buffer CUR^1
    //     wait group 1;  first 16 MFMAs of group 1
    //     wait LDS-DMA;  barrier  (tile t+1 visible; every wave is past its last LDS read of tile t)
    //     issue reads of group 0 (tile t+1)                       | buffer CUR^1
    //     last 16 MFMAs of group 1, tail-row / row-sum FMAs;  wait group 0
    // CUR is a compile-time constant so that all LDS offsets are immediates.  The piece loads of the step after the last
    // re-fetch the last tile (valid addresses, never read): cheaper than a branch around every piece.
#ifdef GEMM_DMA_PROBE
    // build-time instrumentation (make variant X=-DGEMM_DMA_PROBE, scripts/ktrace.py --probe): shader-clock cycles per
    // phase of the k-tile, summed over the main loops of all items of the workgroup, per wave
    unsigned long long probe[7] = {0, 0, 0, 0, 0, 0, 0};
#define GEMM_PROBE(i_) const unsigned long long pt##i_ = __builtin_amdgcn_s_memtime()
#define GEMM_PROBE_DECL(i_) unsigned long long pt##i_ = 0
#define GEMM_PROBE_SET(i_) pt##i_ = __builtin_amdgcn_s_memtime()
#else
#define GEMM_PROBE(i_)
#define GEMM_PROBE_DECL(i_)
#define GEMM_PROBE_SET(i_)
#endif
    auto step = [&](auto cur_c, auto nw_c, const int kt) {
        constexpr int CUR = decltype(cur_c)::value;
        const unsigned ldsC = lds0 + 4 * CUR * SBUF, ldsN = lds0 + 4 * (CUR ^ 1) * SBUF;
        const int ktn = min(kt + 1, nkt - 1);
        GEMM_PROBE(0);
#if !(GEMM_DMA_ABL & 1)
        read_group1(ldsC);
#endif
        __builtin_amdgcn_sched_barrier(0);
        if (wave_active) {
            if (SPREAD == 0) {
#pragma unroll
                for (int i = 0; i < NPIECES; ++i) dma_piece(i, ktn, CUR ^ 1);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int m = 0; m < TM; ++m) {
                    mma(nw_c, a0, b0, e, m);
                    const int slot = e * TM + m;
                    if (SPREAD == 1 && slot < NPIECES) {
                        dma_piece(slot, ktn, CUR ^ 1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (SPREAD == 2 && slot % 3 != 2 && (slot / 3) * 2 + slot % 3 < NPIECES) {
                        dma_piece((slot / 3) * 2 + slot % 3, ktn, CUR ^ 1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
        } else {
#pragma unroll
            for (int i = 0; i < NPIECES; ++i) dma_piece(i, ktn, CUR ^ 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        GEMM_PROBE(1);
        wait_group1();
        GEMM_PROBE(2);
        GEMM_PROBE_DECL(3);
        GEMM_PROBE_DECL(4);
        GEMM_PROBE_DECL(5);
        unsigned seen = 0;
#ifndef GEMM_DMA_ARRIVE_E
#define GEMM_DMA_ARRIVE_E 1        // MFMA groups (8 each) of the second k half issued before the arrival ...
#define GEMM_DMA_WAIT_E 3          // ... and before the wait + the reads of the next tile's group 0
#endif
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (e == GEMM_DMA_ARRIVE_E) {
                // this wave's part of the hand-over: its pieces of tile t+1 have landed, its last reads of tile t are done
                __builtin_amdgcn_sched_barrier(0);
                GEMM_PROBE_SET(3);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#if !(GEMM_DMA_ABL & 2)
                gemm_barrier_arrive(arrivals_addr);
#endif
                __builtin_amdgcn_sched_barrier(0);
            }
            if (e == GEMM_DMA_WAIT_E - 1) {
                __builtin_amdgcn_sched_barrier(0);
#if !(GEMM_DMA_ABL & 2)
                seen = gemm_barrier_peek(arrivals_addr);
#endif
                __builtin_amdgcn_sched_barrier(0);
            }
            if (e == GEMM_DMA_WAIT_E) {
                // every wave has arrived: tile t+1 is complete in LDS, and nobody reads tile t any more (its buffer is the
                // target of the next step's pieces)
                __builtin_amdgcn_sched_barrier(0);
                GEMM_PROBE_SET(4);
#if !(GEMM_DMA_ABL & 2)
                gemm_barrier_wait(arrivals_addr, 4u * (unsigned)(kt + 1), seen);
#endif
                GEMM_PROBE_SET(5);
#if !(GEMM_DMA_ABL & 1)
                read_group0(ldsN);
#endif
                __builtin_amdgcn_sched_barrier(0);
            }
            if (wave_active) {
#pragma unroll
                for (int m = 0; m < TM; ++m) mma(nw_c, a1, b1, e, m);
            }
        }
        if (TAIL) {
            tail_acc = fmaf(t4.x, tb4.x, tail_acc);
            tail_acc = fmaf(t4.y, tb4.y, tail_acc);
            tail_acc = fmaf(t4.z, tb4.z, tail_acc);
            tail_acc = fmaf(t4.w, tb4.w, tail_acc);
        }
        if (B_KC) rowsum_acc += (rs4.x + rs4.y) + (rs4.z + rs4.w);
        __builtin_amdgcn_sched_barrier(0);
        GEMM_PROBE(6);
        wait_group0();
#ifdef GEMM_DMA_PROBE
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long pt7 = __builtin_amdgcn_s_memtime();
        probe[0] += pt1 - pt0;      // group-1 read issue, 32 MFMAs, LDS-DMA pieces
        probe[1] += pt2 - pt1;      // wait group 1
        probe[2] += pt3 - pt2;      // MFMAs of the second half before the arrival
        probe[3] += pt4 - pt3;      // wait LDS-DMA, arrive, MFMAs up to the wait
        probe[4] += pt5 - pt4;      // split-barrier wait
        probe[5] += pt6 - pt5;      // group-0 read issue, remaining MFMAs, side FMAs
        probe[6] += pt7 - pt6;      // wait group 0
        __builtin_amdgcn_sched_barrier(0);
#endif
    };
    auto main_loop = [&](auto nw_c) {
        for (int kt = 0; kt < nkt; kt += 2) {
            step(std::integral_constant<int, 0>{}, nw_c, kt);
            if (kt + 1 < nkt) step(std::integral_constant<int, 1>{}, nw_c, kt + 1);
        }
    };

    if (SCALE) {
        if (!p.bscale && tid < 2 * BK) smem[(tid >> 4) * SBUF + SA + SB + BK + (tid & 15)] = 1.f;       // no row scale: both chunks stay 1
    }
    bool have = take_item(t);
    if (have) set_offsets(lane);
    bool prefetched = false;                           // the first k-tile of the current item is already on its way into buffer 0
    int it = 0;                                        // items done by this workgroup (ticket slot = it & 1)
    long long* trace_row = nullptr;
    while (have) {
        const int vb = p.lists == 8 ? 8 * t + list : t;                 // the item's index in the classic grid (= its trace row)
        trace_row = (trace_on && p.trace && vb < p.trace_rows) ? p.trace + 8 * (long)vb : nullptr;
        if (trace_row && tid == 0) {
            trace_row[0] = trace_row[1] = __builtin_amdgcn_s_memrealtime();
            trace_row[4] = (long long)((__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u) << 16 | (__builtin_amdgcn_s_getreg((31 << 11) | 4) & 0xffffu));
            trace_row[7] = (long long)blockIdx.x << 16 | (long long)it;
        }
#ifdef GCCNMF_EXPERIMENTS
        set_lane_constants(gemm_opaque_tid());
        if (it > 0) set_offsets(lane);
        // every fragment register starts an item defined: a conditional read followed by an unconditional tie would otherwise keep the
        // previous item's values alive across the epilogue (76 registers the epilogue does not have)
        a0.clear(); a1.clear(); b0.clear(); b1.clear();
        sc0 = sc1 = t4 = tb4 = ts4 = rs4 = zero4;
        tbxy = tbzw = gemm_f32x2{0.f, 0.f};
#else
        set_lane_constants(tid);
#endif
        gemm_sync_wait(sync, file, col0, tid, sync_it, list, n_sync);   // chained launch: this item's operands come from an earlier stage of the same launch
        if (trace_row && tid == 0) trace_row[1] = __builtin_amdgcn_s_memrealtime();      // [1] - [0] = time spent waiting for a producer
        // ---- prologue: tile 0 -> buffer 0 (unless the previous item's epilogue already sent it), group 0 of tile 0 into registers
        if (!prefetched) {
#pragma unroll
            for (int i = 0; i < NPIECES; ++i) dma_piece(i, 0, 0);
        }
#pragma unroll
        for (int m = 0; m < TM; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
        tail_acc = 0.f;
        rowsum_acc = 0.f;
        // row factors of the lean epilogue (visible after the prologue barrier; read only after the main loop)
        if constexpr (EPI == EPI_STORE || EPI == EPI_UPDH) {
#pragma unroll
            for (int i = 0; i < BM / 256; ++i) {
                const int lr = tid + 256 * i, row = min(row0 + lr, p.M - 1);
                s_rowvec[lr] = p.ktailA ? p.ktailA[file * p.s_ktailA + row] : 0.f;
                if (EPI == EPI_UPDH) {
                    const float rd = 1.0f / (p.E2[file * p.sE2 + row] + p.alpha + p.eps);
                    s_rowvec[BM + lr] = p.E1 ? p.E1[file * p.sE1 + row] * rd : rd;
                }
            }
        }
        if (tid == 0) {
            s_arrivals = 0;
            // the ticket of the NEXT item, taken now: its round trip is hidden by this item, and the epilogue can already fetch for it
            if (persistent) s_ticket[it & 1] = p.wpl + (int)__hip_atomic_fetch_add(p.tickets + list, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        read_group0(lds0);
        wait_group0();

        if constexpr (NARROW) {
            if (nw == 1) main_loop(std::integral_constant<int, 1>{});
            else main_loop(std::integral_constant<int, 2>{});
        } else {
            main_loop(std::integral_constant<int, 2>{});
        }
        __syncthreads();                      // the epilogues reuse the staging buffers
        if (trace_row && tid == 0) trace_row[2] = __builtin_amdgcn_s_memrealtime();

        // ---- the finished item's coordinates move aside; the next item's operands are set up (and requested) before the epilogue
        const int e_file = file, e_row0 = row0, e_col0 = col0, e_nw = nw, e_n = n_item;
        const bool e_active = wave_active, e_tail = do_tail, e_rowsum = do_rowsum;
        have = false;
        prefetched = false;
        if (persistent) {
            t = __builtin_amdgcn_readfirstlane(s_ticket[it & 1]);
            have = take_item(t);
            if (have && p.prefetch) {
                set_offsets(gemm_opaque_tid() & 63);           // (computed again at the top of the next item: not kept across the epilogue)
#pragma unroll
                for (int i = 0; i < NPIECES; ++i) dma_piece(i, 0, 0);
                prefetched = true;
            }
        }
        float* const scratch = smem + SBUF;             // epilogue scratch: buffer 1 (buffer 0 may be receiving the next item's first tile)

        if constexpr (EPI == EPI_UPDW) {
            // launch_rht_update_w guarantees M % 128 == 0 and N % 64 == 0 (a second, generic variant in this kernel would
            // double the live ranges of the accumulators and spill the main loop)
            if constexpr (TM == 4) gemm_epilogue_update_w_full<TAIL>(p, e_file, e_col0, tid, wm, l31, hh, e_active, acc, tail_acc, rowsum_acc, scratch);
        } else {
            if (e_active) {
                const int row_w = e_row0 + wm * RW;                           // wave-uniform
                bool lean = false;
                if constexpr (EPI == EPI_STORE || EPI == EPI_DIV || EPI == EPI_UPDH) {
                    lean = row_w + RW <= p.M;                               // all tile pairs of the wave have all their rows
                    if (lean) {
                        GemmEpiloguePair<EPI, BM> e0, e1;
                        const int ca = e_col0 + l31, tr = wm * RW;
                        const bool oka = ca < e_n, okb = (!NARROW || e_nw == 2) && ca + 32 < e_n;
                        const long cb = 4L * p.ldc * (p.M + (TAIL ? 1 : 0));
                        float ba = 0.f, bb = 0.f;
                        e0.load(p, e_file, row_w, hh, ca, cb);
                        e1.load(p, e_file, row_w + 32, hh, ca, cb);
                        if (EPI != EPI_DIV && p.ktailA) {
                            ba = p.ktailB[e_file * p.s_ktailB + min(ca, e_n - 1)];
                            bb = p.ktailB[e_file * p.s_ktailB + min(ca + 32, e_n - 1)];
                        }
                        e0.finish(p, e_file, row_w, tr, hh, ca, ba, bb, s_rowvec, acc[0][0], acc[0][1], oka, okb, cb);
                        if (trace_row && tid == 0) trace_row[5] = __builtin_amdgcn_s_memrealtime();
                        if constexpr (TM == 4) e0.load(p, e_file, row_w + 64, hh, ca, cb);
                        e1.finish(p, e_file, row_w + 32, tr + 32, hh, ca, ba, bb, s_rowvec, acc[1][0], acc[1][1], oka, okb, cb);
                        if constexpr (TM == 4) {
                            e1.load(p, e_file, row_w + 96, hh, ca, cb);
                            e0.finish(p, e_file, row_w + 64, tr + 64, hh, ca, ba, bb, s_rowvec, acc[2][0], acc[2][1], oka, okb, cb);
                            e1.finish(p, e_file, row_w + 96, tr + 96, hh, ca, ba, bb, s_rowvec, acc[3][0], acc[3][1], oka, okb, cb);
                        }
                        if (trace_row && tid == 0) trace_row[6] = __builtin_amdgcn_s_memrealtime();
                    }
                }
                if (!lean) {
                    GemmArgs pq = p;              // (the generic epilogues read the column count from the arguments: the file's own in a ragged batch)
                    pq.N = e_n;
#pragma unroll
                    for (int m = 0; m < TM; ++m)
                        gemm_epilogue_pair<EPI>(pq, e_file, row_w + m * 32 + 4 * hh, e_col0 + wn * 64 + l31, acc[m][0], acc[m][1], !NARROW || e_nw == 2);
                }
            }
            if (TAIL) {
                if (e_tail) {
                    scratch[tid] = tail_acc;
                    __syncthreads();
                    if (tid < (NARROW && e_nw == 1 ? 32 : BN)) {
                        const float s = (scratch[tid] + scratch[BN + tid]) + (scratch[2 * BN + tid] + scratch[3 * BN + tid]);
                        const int col = e_col0 + tid;
                        if (EPI == EPI_PHASE ? gemm_col_valid<EPI>(p, col) : col < e_n) gemm_epilogue<EPI>(p, e_file, p.tail_row, col, s);
                    }
                }
            }
            if (B_KC) {
                if (e_rowsum) {
                    float s = rowsum_acc;
                    s += __shfl_xor(s, 1);
                    s += __shfl_xor(s, 2);
                    const int j = tid >> 2;
                    if ((tid & 3) == 0 && (e_col0 + j) < e_n && (!NARROW || e_nw == 2 || j < 32)) p.rowsumB[e_file * p.s_rowsumB + e_col0 + j] = s;
                }
            }
        }
        if (trace_row) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // (also waits for a prefetched first tile: timeline builds only)
            __syncthreads();
            if (tid == 0) trace_row[3] = __builtin_amdgcn_s_memrealtime();
        }
        gemm_sync_signal(sync, e_file, e_col0, e_nw, tid);  // chained launch: this item's output is complete in the XCD's L2
        ++it;
        if (have) __syncthreads();            // s_rowvec, the scratch and s_arrivals are rewritten for the next item
    }
#ifdef GEMM_DMA_PROBE
    if (p.trace && lane == 0 && (long)p.trace_grid + 4L * blockIdx.x + 4 <= p.trace_rows) {
#pragma unroll
        for (int i = 0; i < 7; ++i) p.trace[8 * ((long)p.trace_grid + blockIdx.x * 4 + wave) + i] = (long long)probe[i];
        p.trace[8 * ((long)p.trace_grid + blockIdx.x * 4 + wave) + 7] = it;
    }
#endif
    if (persistent && tid == 0) {
        // the last workgroup to leave zeroes the counters: every other workgroup has taken its last ticket before it counted itself out
        const unsigned gone = __hip_atomic_fetch_add(p.tickets + 8, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (gone == gridDim.x - 1) {
#pragma unroll
            for (int i = 0; i < 9; ++i) __hip_atomic_store(p.tickets + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// The classic launch: one GEMM, workgroup b takes item b >> 3 of list b & 7 (one list: item b).
template <bool A_KC, bool B_KC, int EPI, bool TAIL, int TM, bool NARROW>
__global__ __launch_bounds__(256, 2) void gccnmf_gemm_dma_kernel(GemmArgs p) {
    __shared__ __attribute__((aligned(16))) float smem[2 * GemmDmaLds<TM>::SBUF];
    __shared__ __attribute__((aligned(16))) float s_rowvec[(EPI == EPI_STORE || EPI == EPI_UPDH) ? 2 * 128 * TM : 4];   // lean epilogue row factors
    __shared__ unsigned s_arrivals;
    __shared__ int s_ticket[2];
    const int list = p.lists == 8 ? (int)(blockIdx.x & 7) : 0;
    const int t = p.lists == 8 ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;      // first item: static (the counter starts behind the resident workgroups)
    GemmSync none = {};
    gemm_dma_workgroup<A_KC, B_KC, EPI, TAIL, TM, NARROW>(p, smem, s_rowvec, &s_arrivals, s_ticket, list, t, none);
}

// The chained launch of a KL-NMF iteration (or its first STAGES GEMMs): stage 0 = K1 (R = V / (W.(s*H))), 1 = K2 (H update), 2 = K3
// (R = V / (W.H)), 3 = K4 (R.H^T with the fused W update).  Workgroup b serves list b & 7; its position b >> 3 falls into one stage's
// item range [first[s], first[s + 1]).  K1 and K3 are the same tile program on different arguments.
struct GemmChain {
    int first[5];                        // per list: stage s serves positions [first[s], first[s + 1]) of an iteration
    int it0, iterations;                 // iterations of this launch: it0 .. it0 + iterations - 1 (position / first[4] selects one)
    int trace_it;                        // timeline builds: the iteration whose workgroups write trace rows
    GemmSync sync[4];
};
// TM1: height of K2's tiles in 128-row units -- 2 for dictionaries of at most 256 atoms (a full tile would leave two of its four waves idle)
template <bool TAIL, int STAGES, int TM1>
__global__ __launch_bounds__(256, 2) void gccnmf_gemm_chain_kernel(GemmArgs p0, GemmArgs p1, GemmArgs p2, GemmArgs p3, GemmChain ch) {
    static_assert(STAGES == 2 || STAGES == 4, "");
    __shared__ __attribute__((aligned(16))) float smem[2 * GemmDmaLds<4>::SBUF];
    __shared__ __attribute__((aligned(16))) float s_rowvec[2 * 512];
    __shared__ unsigned s_arrivals;
    __shared__ int s_ticket[2];
    const int list = (int)(blockIdx.x & 7);
    int pos = (int)(blockIdx.x >> 3), it = ch.it0;
    if (ch.iterations > 1) {
        const int q = pos / ch.first[4];
        it += q;
        pos -= q * ch.first[4];
    }
    const bool tr = it == ch.trace_it;
    if (pos < ch.first[1]) {
        gemm_dma_workgroup<true, false, EPI_DIV, TAIL, 4, true>(p0, smem, s_rowvec, &s_arrivals, s_ticket, list, pos, ch.sync[0], it, tr);
    } else if (pos < ch.first[2]) {
        gemm_dma_workgroup<false, false, EPI_UPDH, false, TM1, TM1 == 4>(p1, smem, s_rowvec, &s_arrivals, s_ticket, list, pos - ch.first[1], ch.sync[1], it, tr);
    } else if constexpr (STAGES == 4) {
        if (pos < ch.first[3]) {
            gemm_dma_workgroup<true, false, EPI_DIV, TAIL, 4, true>(p2, smem, s_rowvec, &s_arrivals, s_ticket, list, pos - ch.first[2], ch.sync[2], it, tr);
        } else {
            gemm_dma_workgroup<true, true, EPI_UPDW, TAIL, 4, false>(p3, smem, s_rowvec, &s_arrivals, s_ticket, list, pos - ch.first[3], ch.sync[3], it, tr);
        }
    }
}

// ---- host side: the plan of a launch -----------------------------------------------------------------------------------------
#ifdef GCCNMF_EXPERIMENTS
unsigned* gccnmf_ticket_block(hipStream_t stream);
#endif

// The work lists of one launch of TM-high tiles over the files [a.file0, a.file0 + a.batch).  narrow_capable: the kernel instantiation
// carries the 512 x 32 loop.  Returns the size of the classic grid (items of the longest list x lists), -1 on overflow.
//   key 9 = 0: wide tiles only | 1 (default): a file's ragged last column tile (at most 32 of its 64 columns exist) becomes a narrow item
//   when the launch shares the chip with another file group's launches (its early finish is used at once), or when the extra items do not
//   cost the launch another round of workgroup slots (51 files: 128 tiles per XCD fit two rounds of 64, 122 + 7 items do not) | 2 (tests):
//   every tile as two narrow halves
static int gemm_dma_plan(GemmArgs& a, bool narrow_capable, int TM, int whole_files = 0) {
    a.tiles_m = gccnmf_ceil_div(a.M, 128 * TM);
    a.tiles_n = gccnmf_ceil_div(a.N, 64);
    a.lists = (a.xcd_affine && a.batch >= 8) ? 8 : 1;
    const int policy = gccnmf_tune_tail_split;
    const bool narrow_ok = narrow_capable && TM == 4 && policy != 0;
    a.narrow_ok = narrow_ok ? 1 : 0;
    const long tiles = (long)a.batch * a.tiles_m * a.tiles_n;
    if (tiles > (1L << 28)) return -1;
    a.rag = (narrow_ok && a.tiles_n >= 2 && a.N - (a.tiles_n - 1) * 64 <= 32) ? 1 : 0;
    if (a.rag && policy == 1 && !a.concurrent) {
        const long slots = a.lists == 8 ? 64 : 512;
        const long plain = (tiles + a.lists - 1) / a.lists;
        const long with_rag = ((long)a.batch * a.tiles_m * (a.tiles_n - 1) + a.lists - 1) / a.lists + ((long)a.batch * a.tiles_m + a.lists - 1) / a.lists;
        if ((with_rag + slots - 1) / slots > (plain + slots - 1) / slots) a.rag = 0;
    }
    a.wide_n = a.tiles_n - a.rag;
    a.whole_files = 0;
    if (whole_files) {
        if (a.lists != 8 || policy > 1) return -1;
        a.rag = (narrow_ok && a.tiles_n >= 2 && a.N - (a.tiles_n - 1) * 64 <= 32) ? 1 : 0;      // (no round-count argument: nothing ends at a launch boundary)
        a.wide_n = a.tiles_n - a.rag;
        a.whole_files = whole_files;
        a.cw = whole_files == 2 ? (int)((tiles + 7) / 8) : ((a.batch + 7) / 8) * a.tiles_m * a.tiles_n;
        a.cr = a.split = 0;
        return a.lists * a.cw;
    }
    const long wide = (long)a.batch * a.tiles_m * a.wide_n, ragged = (long)a.batch * a.rag * a.tiles_m;
    a.cw = (int)((wide + a.lists - 1) / a.lists);
    a.cr = (int)((ragged + a.lists - 1) / a.lists);
    a.split = (narrow_ok && policy == 2) ? a.cw : 0;
    return a.lists * (a.cw + a.split + a.cr);
}

// One launch of TM-high tiles over the files [a.file0, a.file0 + a.batch)
template <bool A_KC, bool B_KC, int EPI, bool TAIL, int TM>
static int gccnmf_launch_gemm_dma_tm(GemmArgs a, hipStream_t stream) {
#ifdef GEMM_DMA_NO_NARROW           // A/B build: no instantiation carries the narrow loop
    constexpr bool NARROW = false;
#else
    constexpr bool NARROW = TM == 4 && (EPI == EPI_DIV || EPI == EPI_UPDH || EPI == EPI_STORE);
#endif
    const int classic_grid = gemm_dma_plan(a, NARROW, TM);
    if (classic_grid < 1) return GCCNMF_ERR_ARG;
    a.trace = gccnmf_trace_buf;
    a.trace_rows = gccnmf_trace_buf ? gccnmf_trace_blocks : 0;
    a.trace_grid = classic_grid;
    a.tickets = nullptr;
    a.prefetch = 0;
    a.wpl = 0;
    int grid = classic_grid;
#ifdef GCCNMF_EXPERIMENTS
    a.prefetch = gccnmf_tune_prefetch;
    if (gccnmf_tune_persistent && TM == 4 && classic_grid > 512) {
        a.tickets = gccnmf_ticket_block(stream);
        if (a.tickets) {
            grid = 512;
            a.wpl = 512 / a.lists;
        }
    }
#endif
    hipLaunchKernelGGL((gccnmf_gemm_dma_kernel<A_KC, B_KC, EPI, TAIL, TM, NARROW>), dim3(grid), dim3(256), 0, stream, a);
    GCCNMF_CHECK_LAUNCH();
    return GCCNMF_OK;
}

// A launch is whole rounds of 512 workgroups (two per CU) plus a partial one, and a partial round costs 0.135 ms however few
// workgroups it holds (a workgroup alone on its CU: profiles/r02k_files_sweep.txt).  When that pays, the files the partial round
// would take run as a second launch of HALF-HEIGHT tiles (TM = 2: 256 x 64 per workgroup, 64 x 64 per wave, three workgroups per CU)
// instead: twice the workgroups at half the length (80 files: 3 rounds + 64 tiles -> 76 files + 4 files as 160 half tiles, one per CU for
// a third of the time); a launch of less than one round may run entirely on half-height tiles (16 files: 640 of them, three per CU).
// Whole files only, same k order per output element, so a file's bits do not depend on which launch it rides in.  Outputs of at most 256
// rows (the H update of a dictionary of up to 256 atoms) ALWAYS take half-height tiles: a full tile would leave two of its four waves idle.
// Round 5 measured two alternatives on one box against this policy and kept it (profiles/r05b_files_sweep*.txt): narrow halves of the
// last tiles chosen by a list-scheduling model, and 512 resident workgroups pulling tiles by ticket -- LABBOOK.md.
template <bool A_KC, bool B_KC, int EPI, bool TAIL>
static int gccnmf_launch_gemm_dma(GemmArgs a, hipStream_t stream) {
    if (!a.A || !a.B || !a.C || a.M < 1 || a.N < 1 || a.Kd < 1 || a.batch < 1) return GCCNMF_ERR_ARG;
    if ((a.lda & 3) || (a.ldb & 3)) return GCCNMF_ERR_ARG;
    a.ablate = gccnmf_tune_ablate;
    a.exact_div = gccnmf_tune_exact_div;
    if constexpr (EPI != EPI_UPDW) {
        if ((a.M <= 256 && gccnmf_tune_tail_split != 0) || gccnmf_tune_tail_split == 3) return gccnmf_launch_gemm_dma_tm<A_KC, B_KC, EPI, TAIL, 2>(a, stream);
        const long tpf = (long)gccnmf_ceil_div(a.M, 512) * gccnmf_ceil_div(a.N, 64);     // throughput tiles per file
        const long total = tpf * a.batch, rounds = total / 512;
        // Decided by a cost model in units of one paired round of full tiles = 250 (measured at Kd = 1024: 0.25 ms; everything scales
        // with Kd alike): a partial round of <= 256 full tiles costs 135 (one workgroup alone per CU), a larger one a whole round;
        // half-height workgroups (130-170 VGPRs, 43 KB of LDS: three per CU, 768 per round) cost 85 up to one per CU, 135 up to two,
        // 190 for three; a second launch costs 10 (profiles/r03g_files_sweep.txt, r03k_files_sweep.txt).
        if (gccnmf_tune_tail_split == 1 && !gccnmf_trace_buf && !a.concurrent && a.M > 256) {
            // three forms, priced for a launch that has the chip to itself: all full tiles | whole rounds of full tiles + the rest
            // of the files half-height | everything half-height (768 per round: 32 files 0.389 -> 0.330 ms, 16 files 0.249 -> 0.192)
            auto full_cost = [](long tiles) { const long r = tiles % 512; return (tiles / 512) * 250 + (r == 0 ? 0 : r <= 256 ? 135 : 250); };
            auto half_cost = [](long halves) {
                const long r = halves % 768;
                return (halves / 768) * 190 + (r == 0 ? 0 : r <= 256 ? 85 : r <= 512 ? 135 : 190);
            };
            const long plain = full_cost(total), all_half = half_cost(2 * total);
            long split = 1L << 40;
            const int head = (int)(rounds * 512 / tpf);              // whole files that fit the whole rounds
            const int rest = a.batch - head;
            if (rounds >= 1 && head >= 8 && rest >= 1 && 2 * rest * tpf <= 768) split = full_cost(head * tpf) + half_cost(2 * rest * tpf) + 10;
            if (all_half < plain && all_half <= split) return gccnmf_launch_gemm_dma_tm<A_KC, B_KC, EPI, TAIL, 2>(a, stream);
            if (split < plain) {
                GemmArgs h = a, t = a;
                h.batch = head;
                t.batch = rest;
                t.file0 = a.file0 + head;
                int rc = gccnmf_launch_gemm_dma_tm<A_KC, B_KC, EPI, TAIL, 4>(h, stream);
                if (rc) return rc;
                return gccnmf_launch_gemm_dma_tm<A_KC, B_KC, EPI, TAIL, 2>(t, stream);
            }
        }
    }
    return gccnmf_launch_gemm_dma_tm<A_KC, B_KC, EPI, TAIL, 4>(a, stream);
}
