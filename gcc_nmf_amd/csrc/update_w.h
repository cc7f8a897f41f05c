// The W update + normalisation of one group of AT atoms of one file in ONE pass (gccNMFFunctions.py:77, :79-80): the device function behind
// nmf_update_w_onepass_kernel (nmf.hip) and the W-update stage of the short-dictionary chained launch (direct.hip).
//   smem: 5 * AT floats of LDS.  b: file, ch: atom group.  tid: the thread index (a caller that loops over items passes an opaque copy, so that
//   nothing derived from it becomes an invariant of that loop).
#pragma once
#include <hip/hip_runtime.h>

template <int AT, int NSPLIT>
__device__ __forceinline__ void nmf_update_w_onepass_item(float* __restrict__ W, const float* __restrict__ U, const float* __restrict__ rowsumH,
                                                          float* __restrict__ colsumW, float* __restrict__ hscale, int F, int K, int Kp, long sW,
                                                          long sU, long sVec, long sRowsum, long sSplitU, long sSplitR, float* __restrict__ Wt,
                                                          long sWt, int ldwt, const int b, const int ch, float* const smem, const int tid) {
    constexpr int nsplit = NSPLIT;
    constexpr int L4 = AT / 4;                 // float4 lanes per row segment
    constexpr int PH = 256 / L4;               // row phases per workgroup (PH / 4 per wave)
    constexpr int R = (64 * 9 + PH - 1) / PH;  // rows per thread for F <= 576
    float (*red)[AT] = (float (*)[AT])smem;          // [4][AT]
    float* s_norm = smem + 4 * AT;                    // [AT]
    const int c4 = tid % L4, q = tid / L4, wave = tid >> 6;
    const int k0 = ch * AT + 4 * c4;
    const bool v0 = k0 < K, v1 = k0 + 1 < K, v2 = k0 + 2 < K, v3 = k0 + 3 < K;      // padded atoms stay exactly zero
    float* Wb = W + b * sW;
    const float* Ub = U + b * sU;
    float4 rs = *(const float4*)(rowsumH + b * sRowsum + k0);
#pragma unroll
    for (int sp = 1; sp < nsplit; ++sp) {
        const float4 t = *(const float4*)(rowsumH + b * sRowsum + sp * sSplitR + k0);
        rs.x += t.x; rs.y += t.y; rs.z += t.z; rs.w += t.w;
    }
    // every load of the thread is issued before the first use: rows beyond F re-read row F-1 (clamped, always in bounds) and are
    // masked afterwards -- conditional loads would serialise into R dependent round trips
    float4 wt[R], uu[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const long i = (long)min(q + PH * r, F - 1) * Kp + k0;
        wt[r] = *(const float4*)(Wb + i);
        uu[r] = *(const float4*)(Ub + i);
    }
    float4 tt[NSPLIT > 1 ? NSPLIT - 1 : 1][R];          // the partials of a split-K launch: all in flight together, added in ascending order
#pragma unroll
    for (int sp = 1; sp < nsplit; ++sp)
#pragma unroll
        for (int r = 0; r < R; ++r) tt[sp - 1][r] = *(const float4*)(Ub + sp * sSplitU + (long)min(q + PH * r, F - 1) * Kp + k0);
#pragma unroll
    for (int sp = 1; sp < nsplit; ++sp)
#pragma unroll
        for (int r = 0; r < R; ++r) {
            uu[r].x += tt[sp - 1][r].x; uu[r].y += tt[sp - 1][r].y; uu[r].z += tt[sp - 1][r].z; uu[r].w += tt[sp - 1][r].w;
        }
    float4 ss = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const bool ok = q + PH * r < F;
        const float4 w = wt[r], u = uu[r];
        wt[r] = make_float4((ok && v0) ? w.x * (u.x / rs.x) : 0.f, (ok && v1) ? w.y * (u.y / rs.y) : 0.f, (ok && v2) ? w.z * (u.z / rs.z) : 0.f,
                            (ok && v3) ? w.w * (u.w / rs.w) : 0.f);
        ss.x = fmaf(wt[r].x, wt[r].x, ss.x);
        ss.y = fmaf(wt[r].y, wt[r].y, ss.y);
        ss.z = fmaf(wt[r].z, wt[r].z, ss.z);
        ss.w = fmaf(wt[r].w, wt[r].w, ss.w);
    }
    auto reduce_wave = [&](float4 v) {        // sum over the row phases of this wave (lane bits log2(L4) .. 5)
#pragma unroll
        for (int o = L4; o < 64; o <<= 1) {
            v.x += __shfl_xor(v.x, o);
            v.y += __shfl_xor(v.y, o);
            v.z += __shfl_xor(v.z, o);
            v.w += __shfl_xor(v.w, o);
        }
        return v;
    };
    ss = reduce_wave(ss);
    if ((tid & 63) < L4) *(float4*)&red[wave][4 * c4] = ss;
    __syncthreads();
    if (tid < AT) s_norm[tid] = sqrtf((red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]));
    __syncthreads();
    const float4 nm = *(const float4*)&s_norm[4 * c4];
    float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int f = q + PH * r;
        if (f < F) {
            const float4 wn = make_float4(v0 ? wt[r].x / nm.x : 0.f, v1 ? wt[r].y / nm.y : 0.f, v2 ? wt[r].z / nm.z : 0.f,
                                          v3 ? wt[r].w / nm.w : 0.f);
            *(float4*)(Wb + (long)f * Kp + k0) = wn;
            if (Wt) {            // the reduction-major copy the direct path's W.H reads (padded atoms: zero rows)
                float* t = Wt + b * sWt + (long)k0 * ldwt + f;
                t[0] = wn.x; t[ldwt] = wn.y; t[2 * (long)ldwt] = wn.z; t[3 * (long)ldwt] = wn.w;
            }
            cs.x += wn.x; cs.y += wn.y; cs.z += wn.z; cs.w += wn.w;
        }
    }
    cs = reduce_wave(cs);
    __syncthreads();
    if ((tid & 63) < L4) *(float4*)&red[wave][4 * c4] = cs;
    __syncthreads();
    if (tid < AT && ch * AT + tid < K) {
        const int k = ch * AT + tid;
        colsumW[b * sVec + k] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
        hscale[b * sVec + k] = s_norm[tid];
    }
}
