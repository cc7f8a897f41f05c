// Hand-over between the stages of a CHAINED launch (gemm_dma.h: the throughput tile's K1 | K2 | K3 | K4; direct.hip: the short-dictionary
// launches): ready counters in global memory, one producer-side signal and one consumer-side wait per work item.
#pragma once
#include <hip/hip_runtime.h>

// a file's last column tile is ragged -- at most 32 of its 64 columns exist -- and the launch may run it as a narrow item
__host__ __device__ __forceinline__ bool gemm_dma_file_rag(int narrow_ok, int n) {
    const int tn = (n + 63) >> 6;
    return narrow_ok && tn >= 2 && n - (tn - 1) * 64 <= 32;
}

// ---- chained launches: hand-over between the GEMMs of an iteration inside ONE launch ---------------------------------------------
// The four GEMMs of a KL-NMF iteration depend on each other tile by tile or file by file only (LABBOOK R5.5), and with the XCD-affine
// work lists every producer and consumer of a file run on the same XCD.  A chained launch holds the item lists of several GEMMs back
// to back (per XCD: stage 0's list, then stage 1's, ...); the hardware dispatcher hands workgroups out in order, so every producer
// is resident (or finished) before its consumer starts, and a consumer that finds its operands not ready spins on a counter:
//   producer   after its last store: s_waitcnt vmcnt(0) (its stores have reached the XCD's L2), workgroup barrier, ONE agent-scope
//              atomic add on the counter of what it produced -- no L2 write-back (`buffer_wbl2`): the consumer reads the same L2
//   consumer   thread 0 polls the counter past the L1 (agent-scope load); then buffer_inv sc1 (this CU's L1 may hold lines of the
//              operand's previous contents) and only then the first operand fetch
// A poll gives up after GEMM_SYNC_TIMEOUT (a dispatcher that is not in order, a workgroup on the wrong XCD): it raises *error and goes
// on, the call's outputs are then poisoned by the caller -- never a hang.
struct GemmSync {
    unsigned* wait_cnt;                  // consumer side: nullptr = the stage has no in-launch producer
    unsigned* sig_cnt;                   // producer side: nullptr = nothing waits for this stage inside the launch
    int wait_stride, sig_stride;         // counters per file
    int wait_per_tile, sig_per_tile;     // 1: counter (file, column tile), the producer adds the item's 32-column blocks (nw); 0: counter (file), + 1 per item
    unsigned wait_need;                  // producer counts PER ITERATION the consumer waits for: per column tile 2 x the producer's row tiles (32-column blocks), halved
                                         // for a file's ragged last tile when the producer runs it as ONE narrow item (prod_narrow); per file the producer's items
    int prod_narrow;
    int wait_scale_tiles;                // per-file counter of a ragged batch: wait_need is the count per COLUMN TILE of the file (x its own number of tiles)
    int wait_lag;                        // 0: the producer runs in the same iteration (waits for need x (it + 1)); 1: in the previous one (need x it: K4 -> K1)
    int wide;                            // 1: producer and consumer may sit on DIFFERENT XCDs (a file's tiles spread over the lists): the producer writes its
                                         //    XCD's L2 back (buffer_wbl2 sc1) before it signals -- the agent-scope release; the consumer's buffer_inv sc1 is the acquire
    long long timeout;                   // s_memrealtime ticks a consumer polls before it gives up (GEMM_SYNC_TIMEOUT; the lab build's fault injection: 0)
    unsigned* error;
    unsigned* xcc_seen;                  // [list]: bit x set by every workgroup of the list that ran on XCC x -- checked after the call (one bit per list)
};
#define GEMM_SYNC_TIMEOUT 2000000LL      // s_memrealtime ticks (100 MHz): 20 ms, a whole chained launch is ~1-3 ms

// Agent scope on both sides (the pair the compiler emits for agent-scope relaxed atomics): ~3 us per poll.  The cheaper XCD-local pair -- an
// atomic add without a scope, polls that only bypass the L1 (sc0) -- was tried and does NOT hand over reliably: consumers timed out (round 6).
__device__ __forceinline__ unsigned gemm_sync_peek(const unsigned* counter) {
    unsigned v;
    asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(counter) : "memory");
    return v;
}
// it: the iteration this workgroup belongs to, counted from the zeroing of the counters.
// n_file: the file's column count in the producer's GEMM (K1 - K3: N_f)
__device__ __forceinline__ void gemm_sync_wait(const GemmSync& y, int file, int col0, int tid, int it, int list, int n_file) {
    if (y.xcc_seen && tid == 0) {        // the hand-over relies on a list's workgroups sharing ONE XCD's L2: recorded here (no return, no wait), judged after the call
        const unsigned bit = 1u << (__builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u);      // HW_REG_XCC_ID
        asm volatile("global_atomic_or %0, %1, off sc1" : : "v"(y.xcc_seen + list), "v"(bit) : "memory");
    }
    if (y.wait_cnt) {                    // (kernel argument: wave-uniform)
        if (tid == 0) {
            const int tile = col0 >> 6;
            const unsigned* c = y.wait_cnt + (long)file * y.wait_stride + (y.wait_per_tile ? tile : 0);
            const int tn = (n_file + 63) >> 6;
            unsigned need = y.wait_need;
            if (y.wait_per_tile) {
                if (tile == tn - 1 && gemm_dma_file_rag(y.prod_narrow, n_file)) need >>= 1;
            } else if (y.wait_scale_tiles) {
                need *= (unsigned)tn;
            }
            need *= (unsigned)(it + 1 - y.wait_lag);
            const long long t0 = __builtin_amdgcn_s_memrealtime();
            while (gemm_sync_peek(c) < need) {
                __builtin_amdgcn_s_sleep(16);
                if ((long long)__builtin_amdgcn_s_memrealtime() - t0 > y.timeout) {
                    *y.error = 1u;
                    break;
                }
            }
        }
        __syncthreads();
        asm volatile("buffer_inv sc1" ::: "memory");
    }
}
// all waves: after the item's last store
__device__ __forceinline__ void gemm_sync_signal(const GemmSync& y, int file, int col0, int nw, int tid) {
    if (y.sig_cnt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            if (y.wide) asm volatile("buffer_wbl2 sc1\n\ts_waitcnt vmcnt(0)" ::: "memory");
            unsigned* c = y.sig_cnt + (long)file * y.sig_stride + (y.sig_per_tile ? (col0 >> 6) : 0);
            const unsigned n = y.sig_per_tile ? (unsigned)nw : 1u;
            asm volatile("global_atomic_add %0, %1, off sc1" : : "v"(c), "v"(n) : "memory");
        }
    }
}

