// GCC-PHAT localisation, GCC-NMF target scores / coefficient masks and masked reconstruction.
// Reference: gccNMF/gccNMFFunctions.py:85-151 and gccNMF/runGCCNMF.py:46.
//
// Every contraction here is a real GEMM on the f32 matrix cores (gemm_mfma.h):
//   angular spectrogram   A[tau,t]  = [cos;sin]^T (D x 2Fp) . [Re C; Im C] (2Fp x T)        (:88-92)
//   GCC-NMF scores        G_i[k,t]  = W^T (K x F) . P_i (F x T),  P_i = Re(C * exp(-j w tau_i))  (:127-133)
//   reconstruction        S[i,c]    = W (F x K) . (H_c * M_i) (K x T)  * X_c/|X_c|           (:147-151)
// The three targets / six (target, channel) pairs are concatenated along the GEMM's column axis,
// so each stage is ONE launch for the whole batch.
#include "gemm_ring.h"


// One-shot GEMMs of a launch that cannot fill the chip with 512 x 64 tiles (one mixture alone: 60 of them) take the small-tile ring
// kernel (128 x 64, gemm_ring.h): 240 workgroups instead of 60 -- reconstruction 191 -> ~40 us, scores 83 -> ~25 us, angular
// spectrogram (three 128 x 256 workgroups before) 148 -> ~40 us for one file.
static bool gcc_small_launch(int batch, int M, int N, int Kd) {
    if (!gccnmf_tune_ring || !gccnmf_ring_supports(Kd) || gccnmf_tune_tile_policy == 1) return false;
    if (gccnmf_tune_tile_policy == 2) return true;
    return (long)batch * gccnmf_ceil_div(M, 512) * gccnmf_ceil_div(N, 64) < 256;
}

// mean over time of the angular spectrogram, accumulated in double (runGCCNMF.py:46 works on float64).
// grid = batch * D, 256 threads.
__global__ __launch_bounds__(256) void ang_mean_kernel(const float* __restrict__ ang, int T, int D, int Dp, int Tp,
                                                       double* __restrict__ mean_ang) {
    __shared__ double red[256];
    const int b = blockIdx.x / D, d = blockIdx.x - b * D;
    const float* row = ang + ((long)b * Dp + d) * Tp;
    double s = 0.0;
    for (int t = threadIdx.x; t < T; t += 256) s += (double)row[t];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if (threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) mean_ang[(long)b * Dp + d] = red[0] / (double)T;
}

// Strict local maxima (scipy.signal.argrelmax, order 1: edges never qualify, NaN never compares greater),
// keep the S largest, ascending index order.  One 64-thread block per file; D <= 4096.
__global__ __launch_bounds__(64) void pick_peaks_kernel(const double* __restrict__ mean_ang, int D, int Dp, int S,
                                                        int* __restrict__ tdoa_idx, int* __restrict__ status) {
    __shared__ double v[4096];
    __shared__ unsigned char is_peak[4096];
    const int b = blockIdx.x;
    const double* m = mean_ang + (long)b * Dp;
    for (int i = threadIdx.x; i < D; i += 64) v[i] = m[i];
    __syncthreads();
    for (int i = threadIdx.x; i < D; i += 64) is_peak[i] = (i > 0 && i < D - 1 && v[i] > v[i - 1] && v[i] > v[i + 1]) ? 1 : 0;
    __syncthreads();
    // top-S peaks: S rounds of a 64-lane arg-max over the peaks still standing (largest value, smallest index on ties -- the order
    // the serial scan this replaces produced; that scan cost 33 us of a single-mixture run)
    __shared__ int s_found;
    if (threadIdx.x == 0) s_found = 0;
    __syncthreads();
    for (int s = 0; s < S; ++s) {
        double bv = 0.0;
        int bi = -1;
        for (int i = 1 + threadIdx.x; i < D - 1; i += 64)
            if (is_peak[i] == 1 && (bi < 0 || v[i] > bv)) {
                bv = v[i];
                bi = i;
            }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const double ov = __shfl_xor(bv, o);
            const int oi = __shfl_xor(bi, o);
            if (oi >= 0 && (bi < 0 || ov > bv || (ov == bv && oi < bi))) {
                bv = ov;
                bi = oi;
            }
        }
        if (bi < 0) break;                     // wave-uniform after the butterfly
        if (threadIdx.x == 0) {
            is_peak[bi] = 2;
            ++s_found;
        }
        __syncthreads();
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int* out = tdoa_idx + (long)b * S;
        int n = 0;
        for (int i = 1; i < D - 1 && n < S; ++i)
            if (is_peak[i] == 2) out[n++] = i;
        for (; n < S; ++n) out[n] = -1;
        status[b] = (s_found == S) ? 0 : 1;
    }
}

// P[f][i*Tp + t] = Re(C[f,t] * exp(-j 2 pi f tau_i)) = Cr*cos + Ci*sin, zero in every padded position.
// One thread = four consecutive frames (16-byte loads of Cr / Ci, one 16-byte store).  grid = (ceil(S*Tp/1024), Fp, batch)
__global__ __launch_bounds__(256) void gcc_steer_kernel(const float* __restrict__ CC, const float* __restrict__ trig,
                                                        const int* __restrict__ tdoa_idx, int F, int Fp, int T, int Tp, int D,
                                                        int Dp, int S, float* __restrict__ P) {
    const int col = 4 * (blockIdx.x * 256 + threadIdx.x);
    const int f = blockIdx.y, b = blockIdx.z;
    if (col >= S * Tp) return;
    const int i = col / Tp, t = col - i * Tp;                   // Tp is a multiple of 64: the four frames share the target
    float4 out = make_float4(0.f, 0.f, 0.f, 0.f);
    if (f < F && t < T) {
        int tau = tdoa_idx[(long)b * S + i];
        tau = tau < 0 ? 0 : (tau >= D ? D - 1 : tau);
        const long plane = (long)Fp * Tp;
        const float4 cr = *(const float4*)(CC + (long)b * 2 * plane + (long)f * Tp + t);
        const float4 ci = *(const float4*)(CC + (long)b * 2 * plane + plane + (long)f * Tp + t);
        const float c = trig[(long)f * Dp + tau], sn = trig[((long)Fp + f) * Dp + tau];
        out.x = cr.x * c + ci.x * sn;
        if (t + 1 < T) out.y = cr.y * c + ci.y * sn;
        if (t + 2 < T) out.z = cr.z * c + ci.z * sn;
        if (t + 3 < T) out.w = cr.w * c + ci.w * sn;
    }
    *(float4*)(P + ((long)b * Fp + f) * ((long)S * Tp) + col) = out;
}

// argmax over targets, first index wins ties, NaN ignored (numpy.nanargmax, gccNMFFunctions.py:138).
// One thread = four consecutive frames (S 16-byte loads, one 4-byte store).  grid = (ceil(Tp/1024), Kp, batch)
__global__ __launch_bounds__(256) void gcc_argmax_kernel(const float* __restrict__ scores, int K, int Kp, int T, int Tp, int S,
                                                         unsigned char* __restrict__ argmax) {
    const int t = 4 * (blockIdx.x * 256 + threadIdx.x);
    const int k = blockIdx.y, b = blockIdx.z;
    if (t >= Tp) return;
    unsigned char best[4] = {0, 0, 0, 0};
    if (k < K && t < T) {
        const float* row = scores + ((long)b * Kp + k) * ((long)S * Tp) + t;
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        bool have[4] = {false, false, false, false};
        for (int i = 0; i < S; ++i) {
            const float4 v4 = *(const float4*)(row + (long)i * Tp);
            const float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (v[j] != v[j]) continue;
                if (!have[j] || v[j] > bv[j]) {
                    bv[j] = v[j];
                    best[j] = (unsigned char)i;
                    have[j] = true;
                }
            }
        }
#pragma unroll
        for (int j = 1; j < 4; ++j)
            if (t + j >= T) best[j] = 0;                          // padded frames stay 0
    }
    *(uchar4*)(argmax + ((long)b * Kp + k) * Tp + t) = make_uchar4(best[0], best[1], best[2], best[3]);
}

// PHAT coherence X0*conj(X1)/|X0|/|X1| from an existing spectrogram (runGCCNMF.py:44); the pipeline gets
// it for free from the STFT epilogue, this standalone form serves the host-array API.
// grid = (ceil(T/256), F, batch)
__global__ __launch_bounds__(256) void gcc_coherence_kernel(const float2* __restrict__ X, int Fp, int T, int Tp,
                                                            float* __restrict__ CC) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int f = blockIdx.y, b = blockIdx.z;
    if (t >= T) return;
    const long plane = (long)Fp * Tp;
    const float2 xl = X[(long)b * 2 * plane + (long)f * Tp + t];
    const float2 xr = X[(long)b * 2 * plane + plane + (long)f * Tp + t];
    const float aL = hypotf(xl.x, xl.y), aR = hypotf(xr.x, xr.y);
    float re = xl.x * xr.x + xl.y * xr.y, im = xl.y * xr.x - xl.x * xr.y;
    if (aL > 0.f && aR > 0.f) {
        re = re / aL / aR;
        im = im / aL / aR;
    } else {
        re = 0.f;   // a bin that is exactly 0 in f32 carries no phase: it contributes nothing (NaN policy)
        im = 0.f;
    }
    float* Cb = CC + (long)b * 2 * plane + (long)f * Tp + t;
    Cb[0] = re;
    Cb[plane] = im;
}

// V = concatenate(abs(X), axis=-1) (runGCCNMF.py:40) from an existing spectrogram: V[f][c*T + t] = |X[c][f][t]|, the same hypotf as the
// STFT epilogue (fft.hip), which is where the pipeline gets V from; this standalone form serves the host-array API
// (getTargetSpectrogramEstimates called with a spectrogram that did not come from this process's STFT).
// grid = (ceil(T/256), F, batch * 2)
__global__ __launch_bounds__(256) void gcc_magnitude_kernel(const float2* __restrict__ X, int Fp, int T, int Tp, int Np,
                                                            float* __restrict__ V) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int f = blockIdx.y, b = blockIdx.z >> 1, c = blockIdx.z & 1;
    if (t >= T) return;
    const float2 x = X[((long)b * 2 + c) * Fp * Tp + (long)f * Tp + t];
    V[(long)b * Fp * Np + (long)f * Np + c * T + t] = hypotf(x.x, x.y);
}

// Hm[k][(i*2+c)*Tp + t] = H[k][c*T + t] if argmax[k][t] == i else 0 (H_c * M_i, gccNMFFunctions.py:150).
// One thread = four consecutive frames of one (target, channel) block: one 16-byte store (the write is 4/5 of this kernel's traffic:
// 1 GB per 64-file step), the arg-max as one 4-byte load; H_c starts at column c*T, which is only 4-byte aligned: scalar loads (L2).
// grid = (ceil(2*S*Tp/1024), Kp, batch)
__global__ __launch_bounds__(256) void gcc_masked_h_kernel(const float* __restrict__ H, const unsigned char* __restrict__ argmax,
                                                           const float* __restrict__ masks, int K, int Kp, int T, int Tp, int Np,
                                                           int S, float* __restrict__ Hm) {
    const int col = 4 * (blockIdx.x * 256 + threadIdx.x);
    const int k = blockIdx.y, b = blockIdx.z;
    const int ncol = 2 * S * Tp;
    if (col >= ncol) return;
    const int ic = col / Tp, t = col - ic * Tp;                 // Tp is a multiple of 64: the four frames share (target, channel)
    const int i = ic >> 1, c = ic & 1;
    float4 out = make_float4(0.f, 0.f, 0.f, 0.f);
    if (k < K && t < T) {
        const float* h = H + ((long)b * Kp + k) * Np + c * T + t;
        float4 m;
        if (masks) {   // arbitrary (soft) masks [batch][S][Kp][Tp]: coefficients * targetCoefficientMask
            m = *(const float4*)(masks + (((long)b * S + i) * Kp + k) * Tp + t);
        } else {
            const uchar4 a = *(const uchar4*)(argmax + ((long)b * Kp + k) * Tp + t);
            m = make_float4(a.x == i ? 1.f : 0.f, a.y == i ? 1.f : 0.f, a.z == i ? 1.f : 0.f, a.w == i ? 1.f : 0.f);
        }
        // (h * 1.0f and h * 0.0f are h and +-0: the product form IS the select for finite h, and NaN / Inf coefficients propagate as in
        // the reference's H * mask)
        out.x = h[0] * m.x;
        if (t + 1 < T) out.y = h[1] * m.y;
        if (t + 2 < T) out.z = h[2] * m.z;
        if (t + 3 < T) out.w = h[3] * m.w;
    }
    *(float4*)(Hm + ((long)b * Kp + k) * (long)ncol + col) = out;
}

// C [batch][M][ldc] = A . B with both operands stored [reduction][.] (A(i,kk) = A[kk*lda + i], B(kk,j) = B[kk*ldb + j]), plain store:
// the DFT-as-GEMM path of fft.hip (any n_fft).  Rows of A / B between Kd and the next multiple of 16 must exist and be zero;
// lda, ldb multiples of 4 and >= the tile overreach (operands padded to multiples of 64 columns).
int gccnmf_gemm_nn_store(const float* A, const float* B, float* C, int M, int N, int Kd, int lda, int ldb, int ldc, int batch, long sA,
                         long sB, long sC, hipStream_t s) {
    GemmArgs a = {};
    a.A = A; a.sA = sA; a.lda = lda; a.a_clamp = lda - 4;
    a.B = B; a.sB = sB; a.ldb = ldb; a.b_clamp = ldb - 4;
    a.M = M; a.N = N; a.Kd = Kd;
    a.batch = batch; a.xcd_affine = 0;
    a.C = C; a.sC = sC; a.ldc = ldc;
    if (gcc_small_launch(batch, a.M, a.N, a.Kd)) return gccnmf_launch_gemm_ring<false, false, EPI_STORE, false>(a, s);
    return (M > 128) ? gccnmf_launch_gemm<4, 1, false, false, EPI_STORE, false>(a, s) : gccnmf_launch_gemm<1, 4, false, false, EPI_STORE, false>(a, s);
}

extern "C" {

int gccnmf_angular_spectrogram(const float* CC, const float* trig, int F, int T, int D, int batch, float* ang,
                               double* mean_ang, void* stream) {
    GCCNMF_ENTER();
    if (!CC || !trig || !ang || F < 2 || T < 1 || D < 1 || batch < 1) return GCCNMF_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    GccNmfPitches p = gccnmf_make_pitches(F, T, 1);
    const int Dp = gccnmf_round_up(D, 64);
    GemmArgs a = {};
    a.A = trig; a.sA = 0; a.lda = Dp; a.a_clamp = Dp - 4;
    a.B = CC; a.sB = 2L * p.Fp * p.Tp; a.ldb = p.Tp; a.b_clamp = p.Tp - 4;
    a.M = D; a.N = T; a.Kd = 2 * p.Fp;
    a.batch = batch; a.xcd_affine = 0;
    a.C = ang; a.sC = (long)Dp * p.Tp; a.ldc = p.Tp;
    int rc;
    if (gcc_small_launch(batch, a.M, a.N, a.Kd)) rc = gccnmf_launch_gemm_ring<false, false, EPI_STORE, false>(a, s);
    else rc = (D > 128) ? gccnmf_launch_gemm<4, 1, false, false, EPI_STORE, false>(a, s)
                        : gccnmf_launch_gemm<1, 4, false, false, EPI_STORE, false>(a, s);
    if (rc) return rc;
    if (mean_ang) {
        hipLaunchKernelGGL(ang_mean_kernel, dim3(batch * D), dim3(256), 0, s, ang, T, D, Dp, p.Tp, mean_ang);
        GCCNMF_CHECK_LAUNCH();
    }
    return GCCNMF_OK;
}

int gccnmf_pick_tdoa_peaks(const double* mean_ang, int D, int Dp, int S, int batch, int* tdoa_idx, int* status,
                           void* stream) {
    GCCNMF_ENTER();
    if (!mean_ang || !tdoa_idx || !status || D < 3 || D > 4096 || Dp < D || S < 1 || batch < 1) return GCCNMF_ERR_ARG;
    hipLaunchKernelGGL(pick_peaks_kernel, dim3(batch), dim3(64), 0, (hipStream_t)stream, mean_ang, D, Dp, S, tdoa_idx, status);
    GCCNMF_CHECK_LAUNCH();
    return GCCNMF_OK;
}

long gccnmf_scores_workspace_floats(int F, int T, int S, int batch) {
    GCCNMF_ENTER();
    if (F < 2 || T < 1 || S < 1 || batch < 1) return -1;
    GccNmfPitches p = gccnmf_make_pitches(F, T, 1);
    return (long)batch * p.Fp * S * p.Tp;
}

int gccnmf_target_scores_masks(const float* CC, const float* trig, const int* tdoa_idx, const float* W, int F, int T,
                               int K, int D, int S, int batch, float* workspace, float* scores, unsigned char* argmax,
                               void* stream) {
    GCCNMF_ENTER();
    if (!CC || !trig || !tdoa_idx || !W || !workspace || !scores || F < 2 || T < 1 || K < 1 || D < 1 || S < 1 || S > 255 ||
        batch < 1)
        return GCCNMF_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    GccNmfPitches p = gccnmf_make_pitches(F, T, K);
    const int Dp = gccnmf_round_up(D, 64);
    const int ncol = S * p.Tp;
    float* P = workspace;
    hipLaunchKernelGGL(gcc_steer_kernel, dim3(gccnmf_ceil_div(ncol, 1024), p.Fp, batch), dim3(256), 0, s, CC, trig, tdoa_idx, F,
                       p.Fp, T, p.Tp, D, Dp, S, P);
    GCCNMF_CHECK_LAUNCH();
    GemmArgs a = {};
    a.A = W; a.sA = (long)p.Fp * p.Kp; a.lda = p.Kp; a.a_clamp = p.Kp - 4;
    a.B = P; a.sB = (long)p.Fp * ncol; a.ldb = ncol; a.b_clamp = ncol - 4;
    a.M = K; a.N = ncol; a.Kd = F;
    if ((F % 16) == 1) {
        a.Kd = F - 1;
        a.ktailA = W + (long)(F - 1) * p.Kp; a.s_ktailA = a.sA;
        a.ktailB = P + (long)(F - 1) * ncol; a.s_ktailB = a.sB;
    }
    a.batch = batch; a.xcd_affine = 1;
    a.C = scores; a.sC = (long)p.Kp * ncol; a.ldc = ncol;
    int rc;
    if (gcc_small_launch(batch, a.M, a.N, a.Kd)) rc = gccnmf_launch_gemm_ring<false, false, EPI_STORE, false>(a, s);
    else if (K > 128 && gccnmf_tune_dma && gccnmf_tune_ablate != 128) rc = gccnmf_launch_gemm_dma<false, false, EPI_STORE, false>(a, s);
    else rc = (K > 128) ? gccnmf_launch_gemm<4, 1, false, false, EPI_STORE, false>(a, s)
                        : gccnmf_launch_gemm<1, 4, false, false, EPI_STORE, false>(a, s);
    if (rc) return rc;
    if (argmax) {
        hipLaunchKernelGGL(gcc_argmax_kernel, dim3(gccnmf_ceil_div(p.Tp, 1024), p.Kp, batch), dim3(256), 0, s, scores, K, p.Kp, T,
                           p.Tp, S, argmax);
        GCCNMF_CHECK_LAUNCH();
    }
    return GCCNMF_OK;
}

int gccnmf_argmax_targets(const float* scores, int K, int T, int S, int batch, unsigned char* argmax, void* stream) {
    GCCNMF_ENTER();
    if (!scores || !argmax || K < 1 || T < 1 || S < 1 || S > 255 || batch < 1) return GCCNMF_ERR_ARG;
    GccNmfPitches p = gccnmf_make_pitches(2, T, K);
    hipLaunchKernelGGL(gcc_argmax_kernel, dim3(gccnmf_ceil_div(p.Tp, 1024), p.Kp, batch), dim3(256), 0, (hipStream_t)stream, scores,
                       K, p.Kp, T, p.Tp, S, argmax);
    GCCNMF_CHECK_LAUNCH();
    return GCCNMF_OK;
}

int gccnmf_coherence(const float* X, int F, int T, int batch, float* CC, void* stream) {
    GCCNMF_ENTER();
    if (!X || !CC || F < 2 || T < 1 || batch < 1) return GCCNMF_ERR_ARG;
    GccNmfPitches p = gccnmf_make_pitches(F, T, 1);
    hipLaunchKernelGGL(gcc_coherence_kernel, dim3(gccnmf_ceil_div(T, 256), F, batch), dim3(256), 0, (hipStream_t)stream,
                       (const float2*)X, p.Fp, T, p.Tp, CC);
    GCCNMF_CHECK_LAUNCH();
    return GCCNMF_OK;
}

int gccnmf_magnitude(const float* X, int F, int T, int batch, float* V, void* stream) {
    GCCNMF_ENTER();
    if (!X || !V || F < 2 || T < 1 || batch < 1) return GCCNMF_ERR_ARG;
    GccNmfPitches p = gccnmf_make_pitches(F, T, 1);
    hipLaunchKernelGGL(gcc_magnitude_kernel, dim3(gccnmf_ceil_div(T, 256), F, batch * 2), dim3(256), 0, (hipStream_t)stream,
                       (const float2*)X, p.Fp, T, p.Tp, p.Np, V);
    GCCNMF_CHECK_LAUNCH();
    return GCCNMF_OK;
}

long gccnmf_reconstruct_workspace_floats(int T, int K, int S, int batch) {
    GCCNMF_ENTER();
    if (T < 1 || K < 1 || S < 1 || batch < 1) return -1;
    GccNmfPitches p = gccnmf_make_pitches(2, T, K);
    return (long)batch * p.Kp * 2 * S * p.Tp;
}

int gccnmf_reconstruct(const float* W, const float* H, const unsigned char* argmax, const float* masks, const float* X,
                       const float* V, int F, int T, int K, int S, int batch, float* workspace, float* spec, void* stream) {
    GCCNMF_ENTER();
    if (!W || !H || (!argmax && !masks) || !X || !V || !workspace || !spec || F < 2 || T < 1 || K < 1 || S < 1 || batch < 1)
        return GCCNMF_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    GccNmfPitches p = gccnmf_make_pitches(F, T, K);
    const int ncol = 2 * S * p.Tp;
    float* Hm = workspace;
    hipLaunchKernelGGL(gcc_masked_h_kernel, dim3(gccnmf_ceil_div(ncol, 1024), p.Kp, batch), dim3(256), 0, s, H, argmax, masks, K, p.Kp,
                       T, p.Tp, p.Np, S, Hm);
    GCCNMF_CHECK_LAUNCH();
    const bool tail = (F % 128) == 1;
    GemmArgs a = {};
    a.A = W; a.sA = (long)p.Fp * p.Kp; a.lda = p.Kp; a.a_clamp = p.Fp - 1;
    a.B = Hm; a.sB = (long)p.Kp * ncol; a.ldb = ncol; a.b_clamp = ncol - 4;
    a.M = tail ? F - 1 : F; a.N = ncol; a.Kd = K;
    a.tail_row = F - 1;
    a.batch = batch; a.xcd_affine = 1;
    a.C = spec; a.sC = 2L * S * p.Fp * p.Tp;   // in float2 units (EPI_PHASE indexes complex elements)
    a.E0 = V; a.sE0 = (long)p.Fp * p.Np;
    a.X = (const float2*)X; a.sX = 2L * p.Fp * p.Tp;
    a.T = T; a.Tp = p.Tp; a.Fp = p.Fp; a.ldv = p.Np;
    if (gcc_small_launch(batch, a.M, a.N, a.Kd))
        return tail ? gccnmf_launch_gemm_ring<true, false, EPI_PHASE, true>(a, s) : gccnmf_launch_gemm_ring<true, false, EPI_PHASE, false>(a, s);
    const bool tall = a.M > 128;
    // batch scale: the LDS-DMA throughput tile (main loop of K1) with the generic phase epilogue
    if (tall && gccnmf_tune_dma && gccnmf_tune_ablate != 128)
        return tail ? gccnmf_launch_gemm_dma<true, false, EPI_PHASE, true>(a, s) : gccnmf_launch_gemm_dma<true, false, EPI_PHASE, false>(a, s);
    if (tall) return tail ? gccnmf_launch_gemm<4, 1, true, false, EPI_PHASE, true>(a, s)
                          : gccnmf_launch_gemm<4, 1, true, false, EPI_PHASE, false>(a, s);
    return tail ? gccnmf_launch_gemm<1, 4, true, false, EPI_PHASE, true>(a, s)
                : gccnmf_launch_gemm<1, 4, true, false, EPI_PHASE, false>(a, s);
}

}  // extern "C"
