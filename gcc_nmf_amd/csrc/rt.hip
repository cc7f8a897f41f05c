// Streaming (real-time) GCC-NMF frame processor on gfx950 -- SURVEY.md section 8f #1 and #4.
// Reference: gccNMF/realtime/gccNMFProcessor.py:201-270 (the Theano graph + processFrames) and
// gccNMF/realtime/utils.py:72-118 (OverlapAddProcessor), :34-70 (history ring).
//
// One block of `blockSize` new stereo samples per call; Tc = blockSize/hopSize analysis windows per call.
// Latency-bound, not throughput-bound: everything is sized so that a call is six short launches on one stream and the
// only host traffic is the new block in and the finished block out.
//   rt_shift      8-block input / output buffers move one block left, new samples appended          (utils.py:99-103)
//   rt_frames     sqrt-hamming window, both channels in one packed complex FFT, X and PHAT coherence (processor :202,:253)
//   rt_gccnmf     S[tau,k] = sum_f Re(C[f] e^{-j w tau}) W[f,k] on the matrix cores, arg-max over tau,
//                 soft / boxcar coefficient mask around the tracked target TDOA                      (:254,:259-265)
//   rt_tfmask     tfMask[f] = sum_k W[f,k] HMask[k] / sum_k W[f,k];  Y = tfMask * X                  (:267-269,:209)
//   rt_synth      packed inverse FFT, synthesis window, overlap-add, hand out the block two blocks old (:231, utils.py:113-116)
//   rt_localize   gccPHAT[tau] = nanmean_f G, history ring, target = argmax nanmean(last L columns)   (:214-222)
// NaN semantics follow the reference here (a zero-magnitude bin makes the frame's GCC-NMF scores NaN and its arg-max 0;
// nanmean skips it), because the localisation history depends on them.
#include "fft_core.h"

typedef float rt_f32x16 __attribute__((ext_vector_type(16)));

// ---- rt_shift: single workgroup, in-place left shift by B of both 8-block buffers ---------------------------
__global__ __launch_bounds__(1024) void rt_shift_kernel(float* __restrict__ in_ring, float* __restrict__ out_ring,
                                                        const float* __restrict__ block_in, int B, int ring) {
    // ring = 8*B samples per channel, 2 channels.  Chunks of 8192 values in ascending order, every thread loading everything it
    // will store before the barrier: a chunk reads at or above the positions it writes (never below), so no later chunk's stores can
    // reach what an earlier one still has to read -- any block size, one workgroup (the reference's buffers have no size limit,
    // realtime/utils.py:72-97).
    const int total = 2 * ring;
    for (int base = 0; base < total; base += 8192) {
        float vin[8], vout[8];
#pragma unroll
        for (int n = 0; n < 8; ++n) {
            const int i = base + threadIdx.x + 1024 * n;
            vin[n] = vout[n] = 0.f;
            if (i < total) {
                const int c = i / ring, s = i - c * ring;
                vin[n] = (s + B < ring) ? in_ring[c * ring + s + B] : block_in[c * B + (s + B - ring)];
                vout[n] = (s + B < ring) ? out_ring[c * ring + s + B] : 0.f;
            }
        }
        __syncthreads();
#pragma unroll
        for (int n = 0; n < 8; ++n) {
            const int i = base + threadIdx.x + 1024 * n;
            if (i < total) {
                in_ring[i] = vin[n];
                out_ring[i] = vout[n];
            }
        }
        __syncthreads();
    }
}

// ---- rt_frames: grid = Tc, one analysis window per workgroup ----------------------------------------------------
__global__ __launch_bounds__(FFT_NT) void rt_frames_kernel(const float* __restrict__ in_ring, int ring, int N, int logN, int start0,
                                                           int start_step, int Tc, const float* __restrict__ window,
                                                           const float2* __restrict__ twiddle, float2* __restrict__ X,
                                                           float2* __restrict__ C) {
    extern __shared__ __attribute__((aligned(16))) float2 rt_smem[];
    float2* z = rt_smem;
    float2* tw = rt_smem + N;
    const int t = blockIdx.x;
    const int F = N / 2 + 1;
    const int start = start0 + t * start_step;            // ring mode: utils.py:105; frames mode: t * N
    for (int i = threadIdx.x; i < N / 2; i += FFT_NT) tw[i] = twiddle[i];
    for (int n = threadIdx.x; n < N; n += FFT_NT) {
        const float w = window[n];
        z[bitrev(n, logN)] = make_float2(w * in_ring[start + n], w * in_ring[ring + start + n]);
    }
    __syncthreads();
    fft_stages<false, 1>(z, tw, N, logN, N);
    for (int f = threadIdx.x; f < F; f += FFT_NT) {
        const float2 zk = z[f], zn = z[(N - f) & (N - 1)];
        const float2 XL = make_float2(0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y));     // rfft, NOT conjugated (:202)
        const float2 XR = make_float2(0.5f * (zk.y + zn.y), -0.5f * (zk.x - zn.x));
        X[(long)f * Tc + t] = XL;
        X[((long)F + f) * Tc + t] = XR;
        const float aL = hypotf(XL.x, XL.y), aR = hypotf(XR.x, XR.y);
        float re = XL.x * XR.x + XL.y * XR.y, im = XL.y * XR.x - XL.x * XR.y;
        C[(long)f * Tc + t] = make_float2(re / aL / aR, im / aL / aR);                   // 0/0 -> NaN like the reference (:253)
    }
}

// ---- rt_gccnmf: grid = (Kp/32, Tc); 32 atoms per workgroup, the 4 waves split the reduction over f ------------------------------
// MFMA 32x32x2: A[i = tau][k = f] = G[f][tau] built on the fly from the steering tables, B[k = f][j = atom] = W[f][atom].
// Latency matters here, not throughput (33.7 MFLOP per frame): the first version ran 16 workgroups with a 129-step chain of
// dependent global loads (31 us).  Now each of 8 waves owns an eighth of the frequency rows (17 steps at n_fft = 512: ONE batch of
// loads, all issued before the first use, two TDOA tiles per pass sharing the W and C loads), the partial accumulators meet in
// LDS, and 32 atoms per workgroup double the number of CUs at work.
#define RT_G_WAVES 8
#define RT_G_CHUNK 17
__global__ __launch_bounds__(64 * RT_G_WAVES) void rt_gccnmf_kernel(const float2* __restrict__ C, const float* __restrict__ cosT,
                                                        const float* __restrict__ sinT, const float* __restrict__ W, int F, int K,
                                                        int Kp, int D, int Dp, int Tc, const float* __restrict__ target,
                                                        int target_mode, float* __restrict__ HMask, int* __restrict__ argmaxTDOA) {
    __shared__ float s_part[RT_G_WAVES - 1][32][64];      // partial accumulators (two TDOA tiles) of the other waves
    const int t = blockIdx.y, k0 = blockIdx.x * 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hh = lane >> 5;
    const int atom = k0 + l31;
    // frequency rows of this wave: an even number per wave so that the two lane halves (k = 0 / 1 of an MFMA step) pair up
    const int per = ((F + 2 * RT_G_WAVES - 1) / (2 * RT_G_WAVES)) * 2;
    const int f_lo = wave * per, f_hi = min(f_lo + per, F);
    const int steps = (f_hi - f_lo + 1) / 2;
    float best_val = -INFINITY;
    int best_idx = 0;
    // two TDOA tiles per pass (they share the W and C loads; D = 64 is one pass), 16 steps' loads in flight together
    for (int tt = 0; tt * 32 < Dp; tt += 2) {
        rt_f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
        const bool two = (tt + 1) * 32 < Dp;
        const int tau0 = tt * 32 + l31, tau1 = two ? tau0 + 32 : tau0;
        for (int p0 = 0; p0 < steps; p0 += RT_G_CHUNK) {
            // raw loads first, ALL of them (the scheduler otherwise sinks them next to their uses and the chunk degenerates into a
            // chain of one or two outstanding loads: measured 19 us for the kernel), then the arithmetic
            float2 cv[RT_G_CHUNK];
            float c0[RT_G_CHUNK], s0[RT_G_CHUNK], c1[RT_G_CHUNK], s1[RT_G_CHUNK], bv[RT_G_CHUNK];
#pragma unroll
            for (int u = 0; u < RT_G_CHUNK; ++u) {         // (clamped row, masked below)
                const int f = min(f_lo + 2 * (p0 + u) + hh, F - 1);
                cv[u] = C[f * Tc + t];
                c0[u] = cosT[f * Dp + tau0];
                s0[u] = sinT[f * Dp + tau0];
                c1[u] = cosT[f * Dp + tau1];
                s1[u] = sinT[f * Dp + tau1];
                bv[u] = W[f * Kp + atom];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < RT_G_CHUNK; ++u) {
                const bool ok = (p0 + u) < steps && (f_lo + 2 * (p0 + u) + hh) < f_hi;
                const float b = ok ? bv[u] : 0.f;
                const float a0 = cv[u].x * c0[u] + cv[u].y * s0[u], a1 = cv[u].x * c1[u] + cv[u].y * s1[u];
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(ok ? a0 : 0.f, b, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(ok ? a1 : 0.f, b, acc1, 0, 0, 0);
            }
        }
        if (wave > 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s_part[wave - 1][r][lane] = acc0[r];
                s_part[wave - 1][16 + r][lane] = acc1[r];
            }
        }
        __syncthreads();
        if (wave == 0) {
            // total of the waves' partial sums, then the arg-max over the 32 TDOAs of each tile for the lane's atom:
            // rows (r&3) + 8*(r>>2) + 4*hh, ascending within a lane, tiles in ascending order
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                if (half == 1 && !two) break;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float own = half ? acc1[r] : acc0[r];
                    float v = own;                              // fixed order: w0 + w1 + ... + w7
#pragma unroll
                    for (int w = 0; w < RT_G_WAVES - 1; ++w) v += s_part[w][16 * half + r][lane];
                    const int row = (tt + half) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                    if (row < D && (v > best_val || (v == best_val && row < best_idx))) {
                        best_val = v;
                        best_idx = row;
                    }
                }
            }
        }
        __syncthreads();
    }
    if (wave == 0) {
        const float ov = __shfl_xor(best_val, 32);        // the other row half
        const int oi = __shfl_xor(best_idx, 32);
        if (ov > best_val || (ov == best_val && oi < best_idx)) {
            best_val = ov;
            best_idx = oi;
        }
        if (hh == 0 && atom < K) {
            int i = best_idx;
            if (!(best_val > -INFINITY)) i = 0;           // every score NaN (or D == 0): numpy.argmax of an all-NaN column is 0
            const float tgt = target[0], eps = target[1], beta = target[2], nf = target[3];
            const float dist = fabsf((float)i - tgt);
            float m;
            if (target_mode == 0)
                m = dist < eps ? 1.f : 0.f;                                   // TARGET_MODE_BOXCAR (:263)
            else
                m = expf(-powf(dist / eps, beta)) / (1.f + nf) + nf;          // TARGET_MODE_WINDOW_FUNCTION (:265)
            HMask[(long)atom * Tc + t] = m;
            if (argmaxTDOA) argmaxTDOA[(long)atom * Tc + t] = i;
        }
    }
}

// ---- rt_tfmask: one wave per frequency row; grid = ceil(F/4) ------------------------------------------------------
__global__ __launch_bounds__(256) void rt_tfmask_kernel(const float* __restrict__ W, const float* __restrict__ HMask, int F, int K,
                                                        int Kp, int Tc, const float2* __restrict__ X, float2* __restrict__ Y, float* __restrict__ tfMask) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int f = blockIdx.x * 4 + wave;
    if (f >= F) return;
    const float* Wr = W + (long)f * Kp;
    float rec = 0.f;
    for (int k = lane; k < K; k += 64) rec += Wr[k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) rec += __shfl_xor(rec, o);
    for (int t = 0; t < Tc; ++t) {
        float s = 0.f;
        for (int k = lane; k < K; k += 64) s = fmaf(Wr[k], HMask[(long)k * Tc + t], s);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) {
            const float m = s / rec;
            tfMask[(long)f * Tc + t] = m;
            const float2 a = X[(long)f * Tc + t], b = X[((long)F + f) * Tc + t];
            Y[(long)f * Tc + t] = make_float2(m * a.x, m * a.y);
            Y[((long)F + f) * Tc + t] = make_float2(m * b.x, m * b.y);
        }
    }
}

// ---- rt_synth: single workgroup; frames in order (their output ranges overlap) ----------------------------------------
__global__ __launch_bounds__(FFT_NT) void rt_synth_kernel(const float2* __restrict__ Y, int N, int logN, int start0, int start_step,
                                                          int accumulate, int Tc, int ring, int B, int out_delay,
                                                          const float* __restrict__ window, const float2* __restrict__ twiddle,
                                                          float* __restrict__ out_ring, float* __restrict__ block_out) {
    extern __shared__ __attribute__((aligned(16))) float2 rt_smem[];
    float2* z = rt_smem;
    float2* tw = rt_smem + N;
    const int F = N / 2 + 1;
    const float invN = 1.f / (float)N;
    for (int i = threadIdx.x; i < N / 2; i += FFT_NT) tw[i] = twiddle[i];
    for (int t = 0; t < Tc; ++t) {
        __syncthreads();
        for (int f = threadIdx.x; f < F; f += FFT_NT) {
            float2 a = Y[(long)f * Tc + t], b = Y[((long)F + f) * Tc + t];
            if (f == 0 || f == N / 2) {      // numpy.fft.irfft ignores the imaginary part of DC and Nyquist
                a.y = 0.f;
                b.y = 0.f;
            }
            z[bitrev(f, logN)] = make_float2(a.x - b.y, a.y + b.x);
            if (f != 0 && f != N / 2) z[bitrev(N - f, logN)] = make_float2(a.x + b.y, b.x - a.y);
        }
        __syncthreads();
        fft_stages<true, 1>(z, tw, N, logN, N);
        const int start = start0 + t * start_step;
        for (int n = threadIdx.x; n < N; n += FFT_NT) {
            const float w = window[n];
            const float ya = w * (z[n].x * invN), yb = w * (z[n].y * invN);
            if (accumulate) {                    // overlap-add into the output buffer (utils.py:113-114)
                out_ring[start + n] += ya;
                out_ring[ring + start + n] += yb;
            } else {                             // frames mode: the processed frames themselves (:231)
                out_ring[start + n] = ya;
                out_ring[ring + start + n] = yb;
            }
        }
    }
    if (!accumulate) return;
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * B; i += FFT_NT) {
        const int c = i / B, s = i - c * B;
        block_out[i] = out_ring[c * ring + ring - (out_delay + 1) * B + s];           // utils.py:116 (out_delay = 2 there)
    }
}

// ---- any even window size (the reference takes every windowSize: numpy.fft.rfft / irfft, gccNMFProcessor.py:202,:231) -------------
// Off the powers of two the transform is evaluated as the sum it is, against a full-circle table (cos, sin)(2 pi k / N), k < N, computed
// in float64 on the host: a frame is 2 x (N/2 + 1) x N complex MACs (N = 1000: 4 MFLOP) -- nothing on this chip, and no mixed-radix
// machinery for a path whose defaults are powers of two.  The angle index f * n mod N advances by one addition and one compare.
// rt_frames_dft: grid = (ceil(F / 64), Tc), 256 threads = 64 frequencies x 4 segments of n (partial sums added in segment order).
__global__ __launch_bounds__(256) void rt_frames_dft_kernel(const float* __restrict__ in_ring, int ring, int N, int start0, int start_step,
                                                            int Tc, const float* __restrict__ window, const float2* __restrict__ table,
                                                            float2* __restrict__ X, float2* __restrict__ C) {
    extern __shared__ __attribute__((aligned(16))) float2 rt_smem[];
    float2* xw = rt_smem;                  // windowed samples (left, right)
    float2* tb = rt_smem + N;
    __shared__ float4 s_part[4][64];
    const int t = blockIdx.y, F = N / 2 + 1;
    const int start = start0 + t * start_step;
    for (int n = threadIdx.x; n < N; n += 256) {
        const float w = window[n];
        xw[n] = make_float2(w * in_ring[start + n], w * in_ring[ring + start + n]);
        tb[n] = table[n];
    }
    __syncthreads();
    const int fl = threadIdx.x & 63, seg = threadIdx.x >> 6;
    const int f = min(blockIdx.x * 64 + fl, F - 1);
    const int n0 = (int)(((long)N * seg) / 4), n1 = (int)(((long)N * (seg + 1)) / 4);
    int idx = (int)(((long)f * n0) % N);
    float reL = 0.f, imL = 0.f, reR = 0.f, imR = 0.f;
    for (int n = n0; n < n1; ++n) {
        const float2 c = tb[idx], x = xw[n];           // e^{-j theta} = cos - j sin (rfft, NOT conjugated, :202)
        reL = fmaf(x.x, c.x, reL); imL = fmaf(-x.x, c.y, imL);
        reR = fmaf(x.y, c.x, reR); imR = fmaf(-x.y, c.y, imR);
        idx += f;
        if (idx >= N) idx -= N;
    }
    s_part[seg][fl] = make_float4(reL, imL, reR, imR);
    __syncthreads();
    if (seg == 0 && blockIdx.x * 64 + fl < F) {
        float4 a = s_part[0][fl];
#pragma unroll
        for (int q = 1; q < 4; ++q) {
            const float4 b = s_part[q][fl];
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        const float2 XL = make_float2(a.x, a.y), XR = make_float2(a.z, a.w);
        X[(long)f * Tc + t] = XL;
        X[((long)F + f) * Tc + t] = XR;
        const float aL = hypotf(XL.x, XL.y), aR = hypotf(XR.x, XR.y);
        const float re = XL.x * XR.x + XL.y * XR.y, im = XL.y * XR.x - XL.x * XR.y;
        C[(long)f * Tc + t] = make_float2(re / aL / aR, im / aL / aR);                   // 0/0 -> NaN like the reference (:253)
    }
}

// rt_synth_dft: one thread per OUTPUT position of the buffer (both channels), frames in ascending order -- the reference's accumulation
// order (utils.py:113-114), and no two threads ever add into the same sample.  grid = ceil(positions / 256); s_lo = first position.
__global__ __launch_bounds__(256) void rt_synth_dft_kernel(const float2* __restrict__ Y, int N, int start0, int start_step, int accumulate,
                                                           int Tc, int ring, int B, int out_delay, int s_lo, const float* __restrict__ window,
                                                           const float2* __restrict__ table, float* __restrict__ out_ring,
                                                           float* __restrict__ block_out) {
    extern __shared__ __attribute__((aligned(16))) float2 rt_smem[];
    float2* tb = rt_smem;                  // [N]
    float2* ya = rt_smem + N;              // [F] left
    float2* yb = ya + (N / 2 + 1);         // [F] right
    const int F = N / 2 + 1, H = N / 2;
    const float invN = 1.f / (float)N;
    const int s = s_lo + blockIdx.x * 256 + threadIdx.x;
    const bool live = s < ring;
    for (int n = threadIdx.x; n < N; n += 256) tb[n] = table[n];
    float va = 0.f, vb = 0.f;
    if (live && accumulate) {
        va = out_ring[s];
        vb = out_ring[ring + s];
    }
    for (int t = 0; t < Tc; ++t) {
        __syncthreads();
        for (int f = threadIdx.x; f < F; f += 256) {
            float2 a = Y[(long)f * Tc + t], b = Y[((long)F + f) * Tc + t];
            if (f == 0 || f == H) {          // numpy.fft.irfft ignores the imaginary part of DC and Nyquist
                a.y = 0.f;
                b.y = 0.f;
            }
            ya[f] = a;
            yb[f] = b;
        }
        __syncthreads();
        const int n = s - (start0 + t * start_step);
        if (live && n >= 0 && n < N) {
            const float sign = (n & 1) ? -1.f : 1.f;
            float sa = 0.f, sb = 0.f;
            int idx = n;                                   // f * n mod N for f = 1
            for (int f = 1; f < H; ++f) {
                const float2 c = tb[idx], a = ya[f], b = yb[f];
                sa = fmaf(a.x, c.x, fmaf(-a.y, c.y, sa));
                sb = fmaf(b.x, c.x, fmaf(-b.y, c.y, sb));
                idx += n;
                if (idx >= N) idx -= N;
            }
            const float w = window[n];
            const float fa = w * ((ya[0].x + sign * ya[H].x + 2.f * sa) * invN), fb = w * ((yb[0].x + sign * yb[H].x + 2.f * sb) * invN);
            va = accumulate ? va + fa : fa;
            vb = accumulate ? vb + fb : fb;
        }
    }
    if (!live) return;
    out_ring[s] = va;
    out_ring[ring + s] = vb;
    const int h0 = ring - (out_delay + 1) * B;           // utils.py:116 (out_delay = 2 there)
    if (accumulate && s >= h0 && s < h0 + B) {
        block_out[s - h0] = va;
        block_out[B + s - h0] = vb;
    }
}

// ---- rt_localize: single workgroup of 1024 threads = Dq TDOAs x 1024/Dq frequency phases ---------------------------
// hist is a [D][Lh] float ring with write position hist_pos[0]; target[0] is updated for the NEXT block (:216-222).
__global__ __launch_bounds__(1024) void rt_localize_kernel(const float2* __restrict__ C, const float* __restrict__ cosT,
                                                           const float* __restrict__ sinT, int F, int D, int Dp, int Dq, int Tc,
                                                           float* __restrict__ hist, int Lh, int* __restrict__ hist_pos,
                                                           int loc_enabled, int loc_window, float* __restrict__ target,
                                                           float* __restrict__ gccphat_out) {
    __shared__ float s_sum[1024];
    __shared__ int s_cnt[1024];
    const int tau = threadIdx.x % Dq, g = threadIdx.x / Dq, G = 1024 / Dq;
    int pos = hist_pos[0];
    for (int t = 0; t < Tc; ++t) {
        float s = 0.f;
        int cnt = 0;
        if (tau < D)
            for (int f = g; f < F; f += G) {
                const float2 c = C[(long)f * Tc + t];
                const float v = c.x * cosT[(long)f * Dp + tau] + c.y * sinT[(long)f * Dp + tau];
                if (v == v) {
                    s += v;
                    ++cnt;
                }
            }
        s_sum[threadIdx.x] = s;
        s_cnt[threadIdx.x] = cnt;
        __syncthreads();
        if (g == 0 && tau < D) {
            float st = 0.f;
            int ct = 0;
            for (int j = 0; j < G; ++j) {
                st += s_sum[j * Dq + tau];
                ct += s_cnt[j * Dq + tau];
            }
            const float m = ct ? st / (float)ct : NAN;                                    // numpy.nanmean over f (:214)
            hist[(long)tau * Lh + pos] = m;
            if (gccphat_out) gccphat_out[(long)tau * Tc + t] = m;
        }
        __syncthreads();
        pos = (pos + 1) % Lh;
    }
    if (loc_enabled) {
        if (g == 0) {
            float m = NAN;
            if (tau < D) {
                float s = 0.f;
                int cnt = 0;
                for (int j = 1; j <= loc_window; ++j) {                                   // the last L columns
                    const float v = hist[(long)tau * Lh + ((pos - j) % Lh + Lh) % Lh];
                    if (v == v) {
                        s += v;
                        ++cnt;
                    }
                }
                m = cnt ? s / (float)cnt : NAN;
            }
            s_sum[tau] = m;
        }
        __syncthreads();
        // numpy.argmax: NaN counts as the maximum, first occurrence wins.  Tree reduction over (rank, index) pairs: a NaN ranks
        // above every number, equal ranks keep the smaller index (the serial scan this replaces cost ~6 us of the call).
        {
            const int i = threadIdx.x;
            float v = (i < D) ? s_sum[i] : -INFINITY;
            int idx = (i < D) ? i : 0x7fffffff;
            __syncthreads();
            for (int w = 512; w > 0; w >>= 1) {
                s_sum[i] = v;
                s_cnt[i] = idx;
                __syncthreads();
                if (i < w) {
                    const float ov = s_sum[i + w];
                    const int oi = s_cnt[i + w];
                    const bool vn = v != v, on = ov != ov;
                    const bool take = (on && !vn) || (on == vn && (on ? oi < idx : (ov > v || (ov == v && oi < idx))));
                    if (take) {
                        v = ov;
                        idx = oi;
                    }
                }
                __syncthreads();
            }
            if (i == 0) target[0] = (float)idx;
        }
    }
    if (threadIdx.x == 0) hist_pos[0] = pos;
}

// ---- per-frame coefficient inference (numHUpdates > 0): KL-NMF H updates with W fixed -------------------------------------------
// gccNMF/gccNMFFunctions.py:76 with W constant (sparsityAlpha = 0):  h <- h * (W^T (v / (W h))) / (W^T 1),  h0 = 1, per channel
// and frame, v = |X_c[:, t]|.  The reference's processor accepts numHUpdates and never uses it (gccNMFProcessor.py:168); with h = 1
// the mask below is exactly its tfMask (:267-269), so numHUpdates = 0 reproduces the reference and n > 0 is the low-latency
// notebook's "NMF coefficients are inferred frame-by-frame" (README.md:74).  Columns: col = 2 * t + c, ncol = 2 * Tc.
__global__ __launch_bounds__(256) void rt_fill_kernel(float* __restrict__ p, float v, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = v;
}

// r[f][col] = |X_c[f][t]| / sum_k W[f][k] h[k][col]; one wave per frequency row, grid = ceil(F/4)
// first != 0: h is still all ones (no fill pass: the first update reads no coefficients and WRITES them)
__global__ __launch_bounds__(256) void rt_wh_kernel(const float* __restrict__ W, const float* __restrict__ Hc, const float2* __restrict__ X,
                                                    float* __restrict__ Rv, int F, int K, int Kp, int Tc, int first) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int f = blockIdx.x * 4 + wave;
    if (f >= F) return;
    const int ncol = 2 * Tc;
    const float* Wr = W + (long)f * Kp;
    for (int t = 0; t < Tc; ++t) {                         // both channels of a frame in one pass over the row of W
        float s0 = 0.f, s1 = 0.f;
        if (first) {
            for (int k = lane; k < K; k += 64) s0 = fmaf(Wr[k], 1.f, s0);
            s1 = s0;
        } else {
#pragma unroll 4
            for (int k = lane; k < K; k += 64) {
                const float w = Wr[k];
                const float2 h = *(const float2*)(Hc + (long)k * ncol + 2 * t);
                s0 = fmaf(w, h.x, s0);
                s1 = fmaf(w, h.y, s1);
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            s0 += __shfl_xor(s0, o);
            s1 += __shfl_xor(s1, o);
        }
        if (lane == 0) {
            const float2 xl = X[(long)f * Tc + t], xr = X[((long)F + f) * Tc + t];
            *(float2*)(Rv + (long)f * ncol + 2 * t) = make_float2(hypotf(xl.x, xl.y) / s0, hypotf(xr.x, xr.y) / s1);
        }
    }
}

// h[k][col] *= (sum_f W[f][k] r[f][col]) / colsumW[k]; 16 atoms x 16 frequency phases per workgroup, grid = Kp/16 (the first version
// had 64 x 4: 16 workgroups with a 65-step chain of dependent loads, 17 us; now 64 workgroups, 17 independent loads per thread)
__global__ __launch_bounds__(256) void rt_hupdate_kernel(const float* __restrict__ W, const float* __restrict__ Rv,
                                                         const float* __restrict__ colsumW, float* __restrict__ Hc, int F, int K, int Kp,
                                                         int Tc, int first) {
    __shared__ float red[4][16];
    const int c = threadIdx.x & 15, q = threadIdx.x >> 4, wave = threadIdx.x >> 6;
    const int k = blockIdx.x * 16 + c;
    const int ncol = 2 * Tc;
    for (int col = 0; col < ncol; ++col) {
        float s = 0.f;
#pragma unroll 4
        for (int f = q; f < F; f += 16) s = fmaf(W[(long)f * Kp + k], Rv[(long)f * ncol + col], s);
        s += __shfl_xor(s, 16);                            // the 4 frequency phases of this wave
        s += __shfl_xor(s, 32);
        if ((threadIdx.x & 63) < 16) red[wave][c] = s;
        __syncthreads();
        if (threadIdx.x < 16 && k < K) {
            const float num = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
            const float h = first ? 1.f : Hc[(long)k * ncol + col];
            Hc[(long)k * ncol + col] = h * (num / colsumW[k]);
        }
        __syncthreads();
    }
}

// per-channel mask with inferred coefficients: m_c[f][t] = sum_k W h_c HMask / sum_k W h_c;  Y_c = m_c X_c;  tfMask [2][F][Tc]
__global__ __launch_bounds__(256) void rt_tfmask_h_kernel(const float* __restrict__ W, const float* __restrict__ HMask,
                                                          const float* __restrict__ Hc, int F, int K, int Kp, int Tc,
                                                          const float2* __restrict__ X, float2* __restrict__ Y, float* __restrict__ tfMask) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int f = blockIdx.x * 4 + wave;
    if (f >= F) return;
    const int ncol = 2 * Tc;
    const float* Wr = W + (long)f * Kp;
    for (int t = 0; t < Tc; ++t) {                         // both channels of a frame in one pass over the row of W
        float n0 = 0.f, d0 = 0.f, n1 = 0.f, d1 = 0.f;
#pragma unroll 4
        for (int k = lane; k < K; k += 64) {
            const float w = Wr[k], hm = HMask[(long)k * Tc + t];
            const float2 h = *(const float2*)(Hc + (long)k * ncol + 2 * t);
            const float wh0 = w * h.x, wh1 = w * h.y;
            d0 += wh0;
            d1 += wh1;
            n0 = fmaf(wh0, hm, n0);
            n1 = fmaf(wh1, hm, n1);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            n0 += __shfl_xor(n0, o);
            d0 += __shfl_xor(d0, o);
            n1 += __shfl_xor(n1, o);
            d1 += __shfl_xor(d1, o);
        }
        if (lane == 0) {
            const float m0 = n0 / d0, m1 = n1 / d1;
            tfMask[(long)f * Tc + t] = m0;
            tfMask[((long)F + f) * Tc + t] = m1;
            const float2 a = X[(long)f * Tc + t], b = X[((long)F + f) * Tc + t];
            Y[(long)f * Tc + t] = make_float2(m0 * a.x, m0 * a.y);
            Y[((long)F + f) * Tc + t] = make_float2(m1 * b.x, m1 * b.y);
        }
    }
}

extern "C" {

// Every buffer is passed explicitly (allocation lives with the host, gcc_nmf_amd/realtime.py).
//   frames_mode = 0: streaming.  block_in [2][B] -> 8-block input buffer -> Tc windows -> ... -> overlap-add -> block_out [2][B]
//   frames_mode = 1: the reference's GCCNMFProcessor.processFrames on its own: in_ring = windowed-sample frames [2][Tc][N]
//                    in, out_ring = processed frames [2][Tc][N] out, no shift / overlap-add (block_in, block_out unused).
int gccnmf_rt_process_block_ll(const float* block_in, float* block_out, float* in_ring, float* out_ring, float* X, float* Y, float* C,
                               float* HMask, int* argmaxTDOA, float* tfMask, float* hist, int* hist_pos, float* target,
                               float* gccphat, const float* W, const float* cosT, const float* sinT, const float* window,
                               const float* synthesis_window, const float* twiddle, const float* colsumW, float* Hcoef, float* Rv,
                               int windowSize, int hopSize, int blockSize, int K, int Kp, int D, int Dp, int numTDOAHistory,
                               int target_mode, int separation_enabled, int localization_enabled, int localization_window,
                               int frames_mode_bits, int numHUpdates, int out_delay_blocks, void* stream) {
    GCCNMF_ENTER();
    // frames_mode_bits: 1 = frames mode (above); 2 = leave the localisation kernel out of this call; 4 = ONLY the localisation kernel
    // (2 then 4 = the same work in two calls, so that a host can fetch block_out before the tracking update has run)
    const int frames_mode = frames_mode_bits & 1;
    const bool skip_localize = frames_mode_bits & 2, only_localize = frames_mode_bits & 4;
    // powers of two from 64 up: the radix-2 LDS transform (twiddle = N/2 values of exp(-2 pi j k / N)); every other even size: the direct
    // sums above (twiddle = the N-entry table (cos, sin)(2 pi k / N))
    const bool pow2 = windowSize >= 64 && (windowSize & (windowSize - 1)) == 0;
    const int logN = pow2 ? ilog2_exact(windowSize) : 0;
    if (!in_ring || !out_ring || !X || !Y || !C || !HMask || !tfMask || !hist || !hist_pos || !target || !W || !cosT || !sinT ||
        !window || !synthesis_window || !twiddle || (!frames_mode && (!block_in || !block_out)))
        return GCCNMF_ERR_ARG;
    if (numHUpdates < 0 || (numHUpdates > 0 && (!colsumW || !Hcoef || !Rv)) || out_delay_blocks < 1 || out_delay_blocks > 7) return GCCNMF_ERR_ARG;
    if (windowSize < 4 || windowSize > 4096 || (windowSize & 1) || hopSize < 1 || blockSize < hopSize || blockSize % hopSize || K < 1 || Kp % 64 || Kp < K || D < 1 ||
        D > 1024 || Dp % 32 || Dp < D || numTDOAHistory < 1 || localization_window < 1 || localization_window > numTDOAHistory)
        return GCCNMF_ERR_ARG;
    const int Tc = blockSize / hopSize, F = windowSize / 2 + 1;
    const int ring = frames_mode ? Tc * windowSize : 8 * blockSize;
    if (!frames_mode && ring < windowSize + (Tc - 1) * hopSize) return GCCNMF_ERR_UNSUPPORTED;      // the 8-block buffer must hold one block's windows
    const int start0 = frames_mode ? 0 : ring - windowSize - (Tc - 1) * hopSize;
    const int start_step = frames_mode ? windowSize : hopSize;
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = sizeof(float2) * (windowSize + windowSize / 2);
    int Dq = 64;
    while (Dq < D) Dq *= 2;                 // power of two so that 1024 % Dq == 0
    if (only_localize) {
        hipLaunchKernelGGL(rt_localize_kernel, dim3(1), dim3(1024), 0, s, (const float2*)C, cosT, sinT, F, D, Dp, Dq, Tc, hist,
                           numTDOAHistory, hist_pos, localization_enabled, localization_window, target, gccphat);
        GCCNMF_CHECK_LAUNCH();
        return GCCNMF_OK;
    }
    if (!frames_mode) {
        hipLaunchKernelGGL(rt_shift_kernel, dim3(1), dim3(1024), 0, s, in_ring, out_ring, block_in, blockSize, ring);
        GCCNMF_CHECK_LAUNCH();
    }
    const size_t lds_dft = sizeof(float2) * (2 * windowSize + 2);
    if (!pow2 && lds_dft + 4096 > 64 * 1024) {
        // windows above ~3800 samples: the table + frame image (+ the analysis kernel's 4 KB of static LDS) pass the 64 KB a launch gets by default
        if (hipFuncSetAttribute((const void*)rt_frames_dft_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dft) != hipSuccess ||
            hipFuncSetAttribute((const void*)rt_synth_dft_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dft) != hipSuccess)
            return GCCNMF_ERR_LAUNCH;
    }
    if (pow2) {
        hipLaunchKernelGGL(rt_frames_kernel, dim3(Tc), dim3(FFT_NT), lds, s, in_ring, ring, windowSize, logN, start0, start_step, Tc,
                           window, (const float2*)twiddle, (float2*)X, (float2*)C);
    } else {
        hipLaunchKernelGGL(rt_frames_dft_kernel, dim3(gccnmf_ceil_div(F, 64), Tc), dim3(256), lds_dft, s, in_ring, ring, windowSize, start0,
                           start_step, Tc, window, (const float2*)twiddle, (float2*)X, (float2*)C);
    }
    GCCNMF_CHECK_LAUNCH();
    if (separation_enabled) {
        hipLaunchKernelGGL(rt_gccnmf_kernel, dim3(Kp / 32, Tc), dim3(64 * RT_G_WAVES), 0, s, (const float2*)C, cosT, sinT, W, F, K, Kp, D, Dp, Tc,
                           target, target_mode, HMask, argmaxTDOA);
        GCCNMF_CHECK_LAUNCH();
        if (numHUpdates == 0) {
            hipLaunchKernelGGL(rt_tfmask_kernel, dim3(gccnmf_ceil_div(F, 4)), dim3(256), 0, s, W, HMask, F, K, Kp, Tc, (const float2*)X,
                               (float2*)Y, tfMask);
            GCCNMF_CHECK_LAUNCH();
        } else {
            for (int it = 0; it < numHUpdates; ++it) {      // h0 = 1 is implicit in the first update (no fill pass)
                hipLaunchKernelGGL(rt_wh_kernel, dim3(gccnmf_ceil_div(F, 4)), dim3(256), 0, s, W, Hcoef, (const float2*)X, Rv, F, K, Kp, Tc,
                                   it == 0 ? 1 : 0);
                GCCNMF_CHECK_LAUNCH();
                hipLaunchKernelGGL(rt_hupdate_kernel, dim3(Kp / 16), dim3(256), 0, s, W, Rv, colsumW, Hcoef, F, K, Kp, Tc, it == 0 ? 1 : 0);
                GCCNMF_CHECK_LAUNCH();
            }
            hipLaunchKernelGGL(rt_tfmask_h_kernel, dim3(gccnmf_ceil_div(F, 4)), dim3(256), 0, s, W, HMask, Hcoef, F, K, Kp, Tc,
                               (const float2*)X, (float2*)Y, tfMask);
            GCCNMF_CHECK_LAUNCH();
        }
    }
    if (pow2) {
        hipLaunchKernelGGL(rt_synth_kernel, dim3(1), dim3(FFT_NT), lds, s, (const float2*)(separation_enabled ? Y : X), windowSize, logN,
                           start0, start_step, frames_mode ? 0 : 1, Tc, ring, blockSize, out_delay_blocks, synthesis_window,
                           (const float2*)twiddle, out_ring, block_out);
    } else {
        // positions that receive a frame this call, plus the block handed out (it may lie in front of them)
        const int h0 = ring - (out_delay_blocks + 1) * blockSize;
        const int s_lo = frames_mode ? 0 : (start0 < h0 ? start0 : (h0 > 0 ? h0 : 0));
        hipLaunchKernelGGL(rt_synth_dft_kernel, dim3(gccnmf_ceil_div(ring - s_lo, 256)), dim3(256), lds_dft, s,
                           (const float2*)(separation_enabled ? Y : X), windowSize, start0, start_step, frames_mode ? 0 : 1, Tc, ring, blockSize,
                           out_delay_blocks, s_lo, synthesis_window, (const float2*)twiddle, out_ring, block_out);
    }
    GCCNMF_CHECK_LAUNCH();
    if (skip_localize) return GCCNMF_OK;
    hipLaunchKernelGGL(rt_localize_kernel, dim3(1), dim3(1024), 0, s, (const float2*)C, cosT, sinT, F, D, Dp, Dq, Tc, hist,
                       numTDOAHistory, hist_pos, localization_enabled, localization_window, target, gccphat);
    GCCNMF_CHECK_LAUNCH();
    return GCCNMF_OK;
}

// The reference's processor as it is: one window for analysis and synthesis (:186-187), no coefficient inference (numHUpdates is
// accepted and unused there, :168), the block two blocks old handed out (utils.py:116).
int gccnmf_rt_process_block(const float* block_in, float* block_out, float* in_ring, float* out_ring, float* X, float* Y, float* C,
                            float* HMask, int* argmaxTDOA, float* tfMask, float* hist, int* hist_pos, float* target,
                            float* gccphat, const float* W, const float* cosT, const float* sinT, const float* window,
                            const float* twiddle, int windowSize, int hopSize, int blockSize, int K, int Kp, int D, int Dp,
                            int numTDOAHistory, int target_mode, int separation_enabled, int localization_enabled,
                            int localization_window, int frames_mode, void* stream) {
    GCCNMF_ENTER();
    return gccnmf_rt_process_block_ll(block_in, block_out, in_ring, out_ring, X, Y, C, HMask, argmaxTDOA, tfMask, hist, hist_pos, target,
                                      gccphat, W, cosT, sinT, window, window, twiddle, nullptr, nullptr, nullptr, windowSize, hopSize,
                                      blockSize, K, Kp, D, Dp, numTDOAHistory, target_mode, separation_enabled, localization_enabled,
                                      localization_window, frames_mode, 0, 2, stream);
}

}  // extern "C"
