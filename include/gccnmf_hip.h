/* libgccnmf_hip.so -- C ABI of the MI355X (gfx950) GCC-NMF hot path.
 *
 * The reference (seanwood/gcc-nmf) is pure Python/NumPy and has no FFI; its "operator API" for
 * this path is the set of free functions in gccNMF/gccNMFFunctions.py + gccNMF/librosaSTFT.py.
 * Each entry point below states which of those functions (file:line in the reference checkout)
 * it computes.  gcc_nmf_amd/_hip.py is the ctypes binding; INTEGRATION.md shows the stub a
 * reference maintainer would add.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer (HBM), every call is asynchronous on `stream`
 *    (a hipStream_t passed as void*), every function returns a status code (0 = GCCNMF_OK);
 *    nothing throws across this boundary;
 *  - a "batch" is `batch` independent stereo mixture files of identical shape, laid out
 *    back to back with the per-file strides implied by the padded geometry below;
 *  - matrices are row-major float32 with padded pitches (gccnmf_pitches): padding is zero
 *    on entry and stays zero.  Complex data is interleaved (re, im) float32 pairs.
 *
 * Geometry (F = n_fft/2+1 bins, T frames, N = 2T NMF columns, K atoms, D TDOAs, S targets):
 *   Fp = round_up(F,16)  Kp = round_up(K,64)  Np = round_up(2T,64)  Tp = round_up(T,64)
 *   X  [batch][2][Fp][Tp] complex   complexMixtureSpectrogram   (gccNMFFunctions.py:61-67)
 *   V  [batch][Fp][Np]              concatenate(abs(X), -1)     (runGCCNMF.py:40)
 *   CC [batch][2][Fp][Tp]           Re / Im planes of the PHAT coherence (runGCCNMF.py:44)
 *   W  [batch][Fp][Kp]   H [batch][Kp][Np]                      (gccNMFFunctions.py:69-83)
 */
#ifndef GCCNMF_HIP_H
#define GCCNMF_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define GCCNMF_OK 0
#define GCCNMF_ERR_ARG 1
#define GCCNMF_ERR_LAUNCH 2
#define GCCNMF_ERR_UNSUPPORTED 3
#define GCCNMF_ERR_COLLECTIVE 4   /* the all-reduce hook of gccnmf_klnmf_shared_run (or RCCL behind gccnmf_rccl_*) failed */

/* flags for gccnmf_klnmf */
#define GCCNMF_FLAG_NO_XCD_AFFINITY 1   /* plain file-major block order instead of the XCD-affine map */
#define GCCNMF_FLAG_UNFUSED_W_UPDATE 2   /* R.H^T and the W update/normalisation as two launches (always so when F-1 > 512) */
#define GCCNMF_FLAG_CONCURRENT_GROUPS 4  /* another file group runs the same call on another stream: keep the throughput tile (a launch
                                          * that has the chip to itself may run its partial last round, or all of it, on half-height tiles) */
#define GCCNMF_FLAG_GROUPS(n) (GCCNMF_FLAG_CONCURRENT_GROUPS | ((n) << 8))   /* ... n equal groups in all (bits 8-15; 0 = two): launch forms that
                                          * are chosen by the size of a launch (tuning keys 16 / 17) are chosen for the groups together, so a
                                          * file's result does not depend on how the batch was split */

int gccnmf_version(void);

/* Tuning knobs: process-global atomics; every library call works on a snapshot taken at its entry.  Results stay valid (and, where a
 * knob only changes the launch form, bitwise the same) for every value of a PRODUCT key.  Keys marked X exist only in the lab build
 * (make EXPERIMENTS=1 -> libgccnmf_hip_exp.so): the product library rejects them with GCCNMF_ERR_ARG and carries none of the code they
 * select.  The authoritative table (defaults, ranges) is csrc/common.h.
 *   2  GEMM tile policy: 0 by launch size, 1 always the 512 x 64 throughput tile, 2 always the 128 x 64 small-batch tile (tests cover both)
 *   3  1 (default) = throughput tiles stage operands by LDS-DMA (csrc/gemm_dma.h), 0 = through registers (csrc/gemm_mfma.h)
 *   7  1 (default) = V / (W.H) is the IEEE quotient like numpy.divide, 0 = v_rcp_f32 + one Newton step (rare 1-ulp differences)
 *   8  file groups (1..4, default 3) of a shared-dictionary shard that cannot fill the chip, on library-owned streams
 *   9  throughput-tile launch forms: 1 (default) by the launcher's cost model -- full tiles | whole rounds of full tiles + the remaining files
 *      half-height | half-height throughout; outputs of at most 256 rows always half-height; a file's ragged last column tile (at most 32 of its
 *      64 columns exist: N = 1244) as a NARROW 512 x 32 item -- 0 = full tiles only, 2 = every tile as two narrow halves, 3 = half-height
 *      everywhere (tests).  Same k order per element: bitwise the same in every form.
 *  10  1 (default) = launches that cannot fill the chip (one mixture alone, up to key-12 files) take the direct-to-register kernels
 *      (csrc/direct.hip); 0 = the ring (csrc/gemm_ring.h) and throughput kernels.   12  largest batch on the direct path (1..8, default 4)
 *  16 / 17  short dictionaries (K <= 128, F = 64 n + 1 <= 513): K1 + K2 as one launch of column tiles / K3 + K4a as one launch of bin slabs
 *      (R never written): 0 never, 1 (default) by a cost model in rounds of 512 workgroups, 2 whenever the shape allows
 *  21  batch scale, K > 256: 1 (default) = the whole gccnmf_klnmf call -- every GEMM of every iteration -- as ONE chained launch whose
 *      workgroups hand tiles over through ready counters in their XCD's L2 (csrc/gemm_dma.h: GemmSync), where that wins: at least three files
 *      per XCD, balanced whole-file lists, no other file group beside it; 0 = never; forced forms for tests and A/B runs: 2 = K1 | K2, 4 = the
 *      four GEMMs of an iteration, 8 = every iteration of the call.  Bitwise the same factors in every form.
 *  23  lists of a chained launch: 1 (default) = whole files per XCD where they balance, else the tile list in equal eighths with agent-scope hand-over;
 *      0 = the plain launch's lists (batch a multiple of 8); 2 = always spread; 3 = always whole files
 *  24  iterations per chained launch (default 2048; a call of more iterations is several chained launches)
 *   X  1 ablations of the register-staged kernel (results INVALID), 4 ring kernel off, 5 / 6 parts of the round-3 single-file split-K, 11 / 13
 *      fixed tile / pipeline depth of the direct kernels, 14 short H updates off the ring kernel, 15 one FFT stage per LDS round trip, 18 / 19
 *      resident-workgroup grid and its prefetch, 20 the 16-atom W update, 22 chained launches with one workgroup per CU, 25 fault injection into
 *      the chained launches' hand-over (consumers give up at once)
 * gccnmf_klnmf_plan reports what a call would launch.  Unknown keys / values: GCCNMF_ERR_ARG. */
int gccnmf_set_tuning(int key, int value);

/* Padded geometry every other entry point assumes. */
int gccnmf_pitches(int F, int T, int K, int* Fp, int* Kp, int* Np, int* Tp);

/* Stereo STFT + magnitude + PHAT coherence, one launch for the whole batch.
 * Replaces computeComplexMixtureSpectrogram (gccNMFFunctions.py:61-67 -> librosaSTFT.py:126-181,
 * center=False, result conjugated), V = concatenate(abs(X)) (runGCCNMF.py:40) and
 * spectralCoherenceV (runGCCNMF.py:44).
 *   x        [batch][2][n_samples] float32 (file stride x_stride floats)
 *   window   [n_fft] float32 analysis window (numpy.hanning(n_fft) for the reference path)
 *   twiddle  [n_fft/2] complex: exp(-2j*pi*k/n_fft), computed in float64 on the host
 *   X, V, CC as in the geometry table; V or CC may be NULL to skip that output.
 * n_fft must be a power of two in [64, 4096] here -- the radix-2 kernel keeps eight frames per workgroup in LDS (four at
 * n_fft = 4096); other sizes GCCNMF_ERR_ARG.  The same holds for gccnmf_istft_ola.  Every other size goes through
 * gccnmf_stft_dft / gccnmf_istft_dft below (the Python stft / istft choose). */
int gccnmf_stft_stereo(const float* x, long x_stride, int n_samples, int n_fft, int hop, int T, int batch,
                       const float* window, const float* twiddle, float* X, float* V, float* CC, void* stream);

/* ANY n_fft (2..8192, e.g. 400, 1000, 1536): the reference's stft / istft accept every size (scipy.fftpack, librosaSTFT.py:162-179,
 * :276-279).  Off the power-of-two sizes the transform is computed as the real GEMM it is, on the matrix cores (csrc/fft.hip):
 *   gccnmf_stft_dft : x [nsig][n_samples] real signals (x_stride floats apart) -> X [nsig][Fp][Tp] complex (= conj(fft), center=False)
 *     basis [round_up(n_fft,16)][2*Fp]: basis[n][f] = w[n] cos(2 pi f n / n_fft), basis[n][Fp+f] = w[n] sin(..), zero elsewhere
 *   gccnmf_istft_dft: spec [nsig][Fp][Tp] complex -> y [nsig][L], L = n_fft + hop*(T-1) - (center ? n_fft : 0); n_fft even
 *     ibasis [2*Fp][round_up(n_fft,64)]: ibasis[k][n] = c_k w[n] cos(2 pi k n / n_fft) / n_fft, ibasis[Fp+k][n] = c_k w[n] sin(..) / n_fft,
 *     c_0 = c_{n_fft/2} = 1, else 2 (the real part of ifft([conj(S), S[-2:0:-1]]) times the synthesis window)
 *   both tables are evaluated in float64 by the caller; workspace: gccnmf_dft_workspace_floats(n_fft, T, nsig) floats. */
long gccnmf_dft_workspace_floats(int n_fft, int T, int nsig);
int gccnmf_stft_dft(const float* x, long x_stride, int n_samples, int n_fft, int hop, int T, int nsig, const float* basis,
                    float* workspace, float* X, void* stream);
int gccnmf_istft_dft(const float* spec, int nsig, int n_fft, int hop, int T, const float* ibasis, float gain, int center, float* workspace,
                     float* y, void* stream);

/* The same with the wav ingest fused in: pcm = interleaved int16 stereo frames [batch][n_samples][2]
 * exactly as they sit in a wav data chunk (frame_stride stereo frames between files); the /32768 conversion of
 * wavread -> pcm2float (gccNMF/wavfile.py:34-37, :57-89) and the de-interleave happen in the STFT's load. */
int gccnmf_stft_stereo_pcm16(const short* pcm, long frame_stride, int n_samples, int n_fft, int hop, int T, int batch,
                             const float* window, const float* twiddle, float* X, float* V, float* CC, void* stream);

/* wav egress on the device: y [groups][2][L] float32 -> pcm [groups][L][2] int16 with wavwrite's clip protection per
 * group (peak >= 1 -> rescale to 0.99) and float2pcm's clip + truncation (gccNMF/wavfile.py:39-48, :92-131).  One group
 * = one target of one file = one wavwrite call.  peak_scratch: `groups` uint32 of device scratch; on return it holds the bit
 * image of each group's peak |y|, and an image >= 0x7F800000 means the group contained NaN / Inf samples (NaN is written as 0,
 * +-Inf clips, no rescale; the reference's result is platform-defined there) -- the caller should treat it as an error. */
int gccnmf_pack_pcm16(const float* y, int groups, int L, unsigned int* peak_scratch, short* pcm, void* stream);

/* KL-NMF multiplicative updates, independent dictionary per file.
 * Replaces the iteration loop of performKLNMF (gccNMFFunctions.py:75-81); the MT19937 initial
 * W, H (:70-73) are drawn on the host and passed in.  W and H are updated in place.
 *   V [batch][Fp][Np], W [batch][Fp][Kp], H [batch][Kp][Np] with Np = round_up(N,64); N = 2T on the
 *   GCC-NMF path but any N >= 1 is accepted (performKLNMF is also called on arbitrary V).
 *   workspace: gccnmf_klnmf_workspace_floats(...) floats of scratch (R, R.H^T, K-vectors; for batch == 1 also the partial
 *   products of the split-K latency path).  Always size it with the batch of the call it is used with.
 *   batch >= 2: a file's result is bit-identical whatever batch it rides in (for equal launch-size decisions); batch == 1
 *   cuts the two long reductions in four parts added in a fixed order (deterministic, summation-order accuracy). */
long gccnmf_klnmf_workspace_floats(int F, int N, int K, int batch);
int gccnmf_klnmf(const float* V, float* W, float* H, float* workspace, int F, int N, int K, int batch,
                 int iterations, float sparsity_alpha, float epsilon, int flags, void* stream);

/* The same for mixtures of DIFFERENT lengths in one call (the reference separates a file of any length, gccNMF/runGCCNMF.py:30-36; a
 * batch of them shards "independent mixture files" whatever their durations): file b has N[b] <= Nmax columns (N: HOST array of `batch`
 * ints), every file's V / H block has the pitches of Nmax (gccnmf_pitches(F, Nmax / 2, K)), zero beyond its own columns.  One chained
 * launch (tuning key 21) whose work lists hold each file's own column tiles -- padding to the 64-column tile only -- dealt out to the
 * XCDs by length; a file's factors are bit for bit those of gccnmf_klnmf on a batch of files of its length.  GCCNMF_ERR_UNSUPPORTED where
 * the chained form does not exist (K <= 128 or not a multiple of 128, fewer than 8 files or 256 column tiles, more than 248 files,
 * F - 1 not a multiple of 128 up to 512): run the files of each length as a batch of their own then.
 * workspace: gccnmf_klnmf_ragged_workspace_floats(F, Nmax, K, batch). */
long gccnmf_klnmf_ragged_workspace_floats(int F, int Nmax, int K, int batch);
int gccnmf_klnmf_ragged(const float* V, float* W, float* H, float* workspace, int F, const int* N, int Nmax, int K, int batch,
                        int iterations, float sparsity_alpha, float epsilon, int flags, void* stream);

/* After a synchronisation: did the chained launches of the last gccnmf_klnmf / gccnmf_klnmf_ragged call on `workspace` hand over cleanly?
 * *status = 0: yes (or the call did not chain); bit 0: a consumer gave up waiting for its producer; bit 1: the workgroups of a work list ran
 * on more than one XCC.  In both cases the call has already turned W and H into NaN (never plausible-looking garbage); this is the explicit
 * check (a blocking 128-byte read).  F, N, K, batch as for the workspace size (N = Nmax for a ragged batch). */
int gccnmf_klnmf_chain_status(const float* workspace, int F, int N, int K, int batch, int* status);

/* Which launches gccnmf_klnmf uses for this problem under the current tuning: bit 0 = the direct latency kernels (a handful of files),
 * bit 1 = K1 + K2 as one launch of column tiles (tuning key 16), bit 2 = K3 + K4a as one launch of 64-bin slabs (key 17), bit 3 = the whole
 * call as one chained launch (key 21).  -1 on bad arguments.
 * (Benchmarks and tests name the kernel they time by this; the result of gccnmf_klnmf does not depend on it beyond round-off.) */
int gccnmf_klnmf_plan(int F, int N, int K, int batch, int flags);

/* One launch group of the iteration on its own (per-kernel tests and per-kernel timing in bench.py).
 * stage: 0 prepare (zero R, colsum W, scale = 1) | 1 R=V/(W.(s*H)) | 2 H update | 3 R=V/(W.H) |
 *        4 U=R.H^T + rowsum H | 5 W update + atom normalisation | 6 materialise H *= s */
int gccnmf_klnmf_stage(const float* V, float* W, float* H, float* workspace, int F, int N, int K, int batch,
                       float sparsity_alpha, float epsilon, int flags, int stage, void* stream);

/* Shared-dictionary KL-NMF building blocks (one W for all files / all ranks; the host all-reduces
 * `partial` across ranks between the two calls -- realtime/gccNMFPretraining.py:79-80 is the
 * reference use of performKLNMF on one big V).
 *   step A: H update with the current W, then partial[0:Fp*Kp] = sum_files (V/WH).H^T and
 *           partial[Fp*Kp : Fp*Kp+Kp] = sum_files rowsum(H)   (deterministic file-order sum)
 *   step B: W *= num/den, atom normalisation, and the compensating H rescale (applied lazily
 *           through `workspace`; gccnmf_klnmf_shared_finish materialises it). */
long gccnmf_klnmf_shared_workspace_floats(int F, int N, int K, int batch);
long gccnmf_klnmf_shared_partial_floats(int F, int K);
int gccnmf_klnmf_shared_begin(const float* W, float* workspace, int F, int N, int K, int batch, void* stream);
int gccnmf_klnmf_shared_step_a(const float* V, const float* W, float* H, float* workspace, float* partial, int F,
                               int N, int K, int batch, float sparsity_alpha, float epsilon, void* stream);
int gccnmf_klnmf_shared_step_b(float* W, float* workspace, const float* partial, int F, int N, int K, int batch,
                               void* stream);
int gccnmf_klnmf_shared_finish(float* H, float* workspace, int F, int N, int K, int batch, void* stream);

/* The same training as ONE call per rank: begin, `iterations` x (step A of every shard -> all-reduce -> step B), finish, all
 * enqueued on `stream` from C -- no host round trip per iteration (gccnmf_klnmf_shared_step(..., ncclComm_t)).
 * A rank's columns are given as up to GCCNMF_MAX_SHARDS shards of `batch` equally wide files each:
 *   ld == 0  whole padded matrices back to back: V [batch][Fp][Np], H [batch][Kp][Np]  (the layout of every other entry point)
 *   ld  > 0  column blocks of ONE matrix with row pitch ld: file b = columns [b*N, b*N + N) of V [Fp][ld] / H [Kp][ld]
 *            (N a multiple of 64 unless batch == 1; batch*N <= ld; columns beyond a ragged block's N up to the next multiple
 *            of 64 must be allocated and zero) -- the frame windows of one long mixture without gather / scatter.
 *   workspace: gccnmf_klnmf_shared_shard_workspace_floats(F, N, K, batch, ld) floats of scratch per shard.
 * nshards == 0 is a rank without columns (fewer files than ranks): it contributes zeros and still applies every W update.
 *   W [Fp][Kp] in/out (identical on every rank on entry -> identical on exit), partial: gccnmf_klnmf_shared_partial_floats
 *   floats, vec: 2*Kp floats of scratch.  The shards' partial sums are added in shard order (deterministic).
 * allreduce: sums `count` floats of `buf` (device memory) in place over all ranks, ordered on `stream`; returns 0 on success.
 *   NULL = single rank.  Not re-entrant per device: the file groups of a shard that cannot fill the chip run on side streams and
 *   events the library owns, one set per device (tuning key 8) -- one training at a time per device, like the reference's
 *   single-threaded caller.  ENFORCED: a call that arrives while another thread is still inside this function for the same device
 *   returns GCCNMF_ERR_UNSUPPORTED without launching anything.  gccnmf_rccl_allreduce below is the RCCL implementation; any other transport (MPI, a host callback that
 *   runs torch.distributed over gloo ...) has the same signature. */
#define GCCNMF_MAX_SHARDS 8
typedef struct gccnmf_shared_shard {
    const float* V;
    float* H;
    float* workspace;
    int N, batch, ld;
} gccnmf_shared_shard;
typedef int (*gccnmf_allreduce_fn)(void* ctx, float* buf, long count, void* stream);
long gccnmf_klnmf_shared_shard_workspace_floats(int F, int N, int K, int batch, int ld);
int gccnmf_klnmf_shared_run(const gccnmf_shared_shard* shards, int nshards, float* W, float* partial, float* vec, int F, int K,
                            int iterations, float sparsity_alpha, float epsilon, gccnmf_allreduce_fn allreduce, void* allreduce_ctx,
                            void* stream);

/* RCCL binding for the hook above (csrc/collective.hip).  librccl.so.1 is bound at run time with dlopen (the copy already
 * loaded into the process, e.g. PyTorch's, else the system one): the library loads without RCCL, these calls then return
 * GCCNMF_ERR_COLLECTIVE.  One process per GPU: rank 0 calls gccnmf_rccl_unique_id and hands the 128 bytes to every rank
 * (any out-of-band channel), every rank calls gccnmf_rccl_comm_init with the device it computes on current.
 *   gccnmf_rccl_allreduce(comm, buf, count, stream): ncclAllReduce(buf, buf, count, ncclFloat32, ncclSum, comm, stream) --
 *   pass it as `allreduce` with allreduce_ctx = comm.  The reference has no collectives; this is the exchange
 *   of the shared-dictionary W update (gccNMFFunctions.py:77 summed over every rank's columns). */
#define GCCNMF_RCCL_UNIQUE_ID_BYTES 128
int gccnmf_rccl_available(void);
int gccnmf_rccl_unique_id(char* id_bytes);
int gccnmf_rccl_comm_init(const char* id_bytes, int world_size, int rank, void** comm);
int gccnmf_rccl_comm_destroy(void* comm);
int gccnmf_rccl_allreduce(void* comm, float* buf, long count, void* stream);
/* the address of gccnmf_rccl_allreduce as a gccnmf_allreduce_fn (for hosts that cannot take a symbol's address, e.g. ctypes) */
gccnmf_allreduce_fn gccnmf_rccl_allreduce_hook(void);

/* GCC-PHAT angular spectrogram A[tau][t] = sum_f Re(C[f,t] exp(-2j pi f tau)) as a real GEMM
 * [cos;sin]^T . [Re C; Im C], plus its time mean.  Replaces getAngularSpectrogram
 * (gccNMFFunctions.py:85-92) and mean(.., axis=-1) (runGCCNMF.py:46).
 *   trig   [2][Fp][Dp] float32: cos(2 pi f tau) / sin(2 pi f tau), Dp = round_up(D,64), zero padded
 *   ang    [batch][Dp][Tp] float32 out,  mean_ang [batch][Dp] float64 out (may be NULL) */
int gccnmf_angular_spectrogram(const float* CC, const float* trig, int F, int T, int D, int batch, float* ang,
                               double* mean_ang, void* stream);

/* Peak picking on the mean angular spectrum: strict local maxima (edges excluded), keep the
 * S largest, ascending order.  Replaces estimateTargetTDOAIndexesFromAngularSpectrum
 * (gccNMFFunctions.py:94-116) for numSources > 0.
 *   tdoa_idx [batch][S] int32 out; status [batch] int32 out (0 ok, 1 = fewer than S peaks) */
int gccnmf_pick_tdoa_peaks(const double* mean_ang, int D, int Dp, int S, int batch, int* tdoa_idx, int* status,
                           void* stream);

/* Per-target GCC-NMF scores G_i[k,t] = Re sum_f W[f,k] C[f,t] exp(-2j pi f tau_i) and the
 * arg-max-over-targets coefficient mask (first index wins ties, NaN ignored).  Replaces
 * getTargetTDOAGCCNMFs (gccNMFFunctions.py:118-135) + getTargetCoefficientMasks (:137-143).
 *   tdoa_idx [batch][S] device int32 (from gccnmf_pick_tdoa_peaks or uploaded)
 *   scores   [batch][Kp][S*Tp] float32 out (target i occupies columns i*Tp .. i*Tp+T-1)
 *   argmax   [batch][Kp][Tp] uint8 out;  workspace: gccnmf_scores_workspace_floats floats */
long gccnmf_scores_workspace_floats(int F, int T, int S, int batch);
int gccnmf_target_scores_masks(const float* CC, const float* trig, const int* tdoa_idx, const float* W, int F,
                               int T, int K, int D, int S, int batch, float* workspace, float* scores,
                               unsigned char* argmax, void* stream);

/* The arg-max half of the call above on its own (host-array getTargetCoefficientMasks,
 * gccNMFFunctions.py:137-143): scores [batch][Kp][S*Tp] -> argmax [batch][Kp][Tp]. */
int gccnmf_argmax_targets(const float* scores, int K, int T, int S, int batch, unsigned char* argmax, void* stream);

/* PHAT coherence from an existing spectrogram X [batch][2][Fp][Tp] (runGCCNMF.py:44); the pipeline
 * gets CC from gccnmf_stft_stereo's epilogue instead. */
int gccnmf_coherence(const float* X, int F, int T, int batch, float* CC, void* stream);

/* V = concatenate(abs(X), axis=-1) (runGCCNMF.py:40) from an existing spectrogram X [batch][2][Fp][Tp] -> V [batch][Fp][Np], columns
 * c*T + t; the pipeline gets V from gccnmf_stft_stereo's epilogue instead (same hypotf). */
int gccnmf_magnitude(const float* X, int F, int T, int batch, float* V, void* stream);

/* Masked reconstruction S[i,c] = (W . (H_c * M_i)) * X_c/|X_c|.  Replaces
 * getTargetSpectrogramEstimates (gccNMFFunctions.py:145-151).
 *   the mask is either `argmax` [batch][Kp][Tp] uint8 (one-hot masks, the device pipeline) or, when
 *   `masks` != NULL, arbitrary float masks [batch][S][Kp][Tp] (the reference signature accepts any array)
 *   spec [batch][S*2][Fp][Tp] complex out (index i*2+c); workspace: gccnmf_reconstruct_workspace_floats */
long gccnmf_reconstruct_workspace_floats(int T, int K, int S, int batch);
int gccnmf_reconstruct(const float* W, const float* H, const unsigned char* argmax, const float* masks,
                       const float* X, const float* V, int F, int T, int K, int S, int batch, float* workspace,
                       float* spec, void* stream);

/* Inverse STFT + overlap-add + centre trim + gain for `nsig` spectrograms per file.  Replaces
 * getTargetSignalEstimates (gccNMFFunctions.py:153-163 -> librosaSTFT.py:241-286, center=True).
 *   spec   [batch][nsig][Fp][Tp] complex (nsig must be even: signals are inverse-transformed in pairs)
 *   window [n_fft] synthesis window; twiddle as for the forward transform
 *   frames [batch][nsig][T][n_fft] float32 scratch, or NULL: inverse transform and overlap-add fused in one pass (same accumulation
 *          order, no frame buffer; needs n_fft + 3*hop <= 2048, else GCCNMF_ERR_UNSUPPORTED)
 *   center != 0 trims n_fft/2 samples at both ends (the reference path), 0 keeps all n_fft + hop*(T-1)
 *   y      [batch][nsig][L] float32 out, L = n_fft + hop*(T-1) - (center ? n_fft : 0) */
int gccnmf_istft_ola(const float* spec, int nsig, int n_fft, int hop, int T, int batch, const float* window,
                     const float* twiddle, float gain, int center, float* frames, float* y, void* stream);

/* The overlap-add half of the call above on its own, for ONE file whose frame sequence is `halo` frames of `prev` [nsig][halo][n_fft]
 * (the previous time shard's last frames; NULL with halo = 0: this file's frames alone) followed by the T frames of `frames`
 * [nsig][T][n_fft] (windowed time frames) -- no concatenated copy of the two -> y [nsig][L] = samples first_sample .. first_sample+L-1
 * of the overlap-added stream, frames added in ascending order (librosaSTFT.py:275-281), times gain.  Used by the time-sharded
 * single-mixture mode, where a shard's first samples need the previous shard's last n_fft/hop - 1 frames. */
int gccnmf_ola_frames_halo(const float* prev, int halo, const float* frames, int nsig, int n_fft, int hop, int T, int first_sample, int L,
                           float gain, float* y, void* stream);

/* Streaming (real-time) GCC-NMF: one block of `blockSize` new stereo samples per call, Tc = blockSize/hopSize analysis
 * windows.  Replaces GCCNMFProcessor.processFrames (gccNMF/realtime/gccNMFProcessor.py:201-270, a Theano graph in the
 * reference) together with OverlapAddProcessor.processFrames (gccNMF/realtime/utils.py:99-116) and the gccPHAT history /
 * online localisation (:214-222, utils.py:34-70).  Six launches on `stream`, no host synchronisation.
 * Sizes: ANY even windowSize in [4, 4096] and any blockSize, like the reference (numpy.fft.rfft / irfft of any length,
 * gccNMFProcessor.py:202,:231; buffers of any size, realtime/utils.py:72-97).  Powers of two from 64 up run the radix-2 LDS transform
 * (`twiddle` = windowSize/2 complex values exp(-2j pi k / N), as for gccnmf_stft_stereo); every other even size is evaluated as the
 * direct sum against `twiddle` = the windowSize-entry table (cos, sin)(2 pi k / N) (csrc/rt.hip: rt_frames_dft / rt_synth_dft).
 * The only requirement left: 8*blockSize >= one block's windows (the reference's 8-block buffers) -> else GCCNMF_ERR_UNSUPPORTED.
 *   block_in / block_out [2][blockSize]            new samples in, the block two blocks old out (utils.py:116)
 *   in_ring / out_ring   [2][8*blockSize]          state: the reference's 8-block buffers
 *   X, Y [2][F][Tc] complex, C [F][Tc] complex     rfft (not conjugated), masked spectrogram, PHAT coherence
 *   HMask [Kp][Tc], argmaxTDOA [Kp][Tc] (or NULL), tfMask [F][Tc], gccphat [D][Tc] (or NULL)
 *   hist [D][numTDOAHistory] + hist_pos [1]        state: gccPHAT history ring
 *   target [4]                                     state: {targetTDOAIndex, epsilon, beta, noiseFloor}; index rewritten by
 *                                                  the localisation for the NEXT block
 *   W [F][Kp]; cosT / sinT [F][Dp] = Re / -Im of exp(-2j pi f tau) (Dp = round_up(D,32), zero padded), float32 grids as
 *   in :245-248; window [windowSize] = sqrt(hamming) (:186); twiddle as for gccnmf_stft_stereo
 *   target_mode 0 = boxcar (:263), 2 = window function (:265)
 *   frames_mode 1 = processFrames alone: in_ring holds windowed-sample frames [2][Tc][windowSize], out_ring receives the
 *   processed frames, no shift / overlap-add. */
int gccnmf_rt_process_block(const float* block_in, float* block_out, float* in_ring, float* out_ring, float* X, float* Y,
                            float* C, float* HMask, int* argmaxTDOA, float* tfMask, float* hist, int* hist_pos, float* target,
                            float* gccphat, const float* W, const float* cosT, const float* sinT, const float* window,
                            const float* twiddle, int windowSize, int hopSize, int blockSize, int K, int Kp, int D, int Dp,
                            int numTDOAHistory, int target_mode, int separation_enabled, int localization_enabled,
                            int localization_window, int frames_mode, void* stream);

/* Low-latency extension of the call above (BASELINE config 5; README.md:74-78 -- the notebooks that implemented it are not in the
 * reference checkout, so this part has no reference code to be pinned on):
 *   synthesis_window [windowSize]  separate synthesis window (asymmetric analysis / synthesis pairs: long analysis window for
 *                                  spectral resolution, short synthesis window at the end of the frame for latency)
 *   numHUpdates > 0                per-frame coefficient inference: that many KL-NMF H updates with W fixed
 *                                  (gccNMFFunctions.py:76), h0 = 1, per channel; the mask becomes W.(h*HMask) / W.h per channel,
 *                                  tfMask is then [2][F][Tc].  colsumW [Kp] = sum_f W, Hcoef [Kp][2*Tc], Rv [F][2*Tc] scratch.
 *                                  numHUpdates = 0 is exactly gccnmf_rt_process_block (h = 1).
 *   frames_mode bits               1 = frames mode; 2 = everything but the localisation kernel; 4 = only the localisation kernel
 *                                  (2 then 4 on the same stream = one call; lets a host fetch block_out before the tracking update)
 *   out_delay_blocks               which finished block is handed out: 2 = the reference (utils.py:116); 1 is complete when
 *                                  the synthesis window spans at most two hops */
int gccnmf_rt_process_block_ll(const float* block_in, float* block_out, float* in_ring, float* out_ring, float* X, float* Y, float* C,
                               float* HMask, int* argmaxTDOA, float* tfMask, float* hist, int* hist_pos, float* target,
                               float* gccphat, const float* W, const float* cosT, const float* sinT, const float* window,
                               const float* synthesis_window, const float* twiddle, const float* colsumW, float* Hcoef, float* Rv,
                               int windowSize, int hopSize, int blockSize, int K, int Kp, int D, int Dp, int numTDOAHistory,
                               int target_mode, int separation_enabled, int localization_enabled, int localization_window,
                               int frames_mode, int numHUpdates, int out_delay_blocks, void* stream);

#ifdef GCCNMF_EXPERIMENTS      /* libgccnmf_hip_exp.so (make EXPERIMENTS=1): measurement tooling of the lab build */
/* Debug: per-workgroup timeline of the LDS-DMA GEMM launches (device buffer of 8 x int64 per workgroup: s_memrealtime
 * [100 MHz] at entry (slots 0 and 1), after the main loop, after the epilogue; [4] = xcc_id<<16 | HW_ID[15:0]).
 * Recorded for launches of at most `blocks` workgroups while buf != NULL (scripts/kbench.py --trace). */
int gccnmf_debug_set_trace(long long* buf, int blocks);

/* Diagnostics: a pure v_mfma_f32_32x32x2_f32 loop (blocks x 4 waves x iters x 8 instructions, 2*32*32*2 flop each):
 * the matrix-pipe rate this box sustains, quoted next to the roofline fractions. */
int gccnmf_debug_mfma_peak(float* scratch, int blocks, int iters, void* stream);
#endif

/* Latency-path GEMM with the KL-NMF element-wise work fused (csrc/direct.hip): what gccnmf_klnmf runs for ONE mixture alone
 * (performKLNMF(V (513, 1244), 1024, 100, 0) is four of these + the W update per iteration), exported so that it can be tested
 * and timed in isolation.
 *     C[m][n] = sum_r (bscale[r] *) A[r][m] * B[r][n]        r < Kd, m < M, n < N      (bscale: epilogue 1 only, else NULL)
 * BOTH operands are reduction-major: one reduction index per row (pitches lda / ldb, multiples of 4), the output index contiguous
 * -- rows r < round_up(Kd, 16) must be addressable, and zero beyond Kd in at least one operand.  Every pointer is device memory;
 * per-file strides (s*) are in floats.  epilogue (numpy.dot / element-wise lines of gccNMFFunctions.py:76-77):
 *   0 STORE  C = acc;  optional rowsumB[n] = sum_r B[r][n];
 *   1 DIV    C = E0 / acc                                   (E0 has pitch lde0)
 *   2 DIVT   Ct[n][m] = E0[m][n] / acc                      (transposed output only)
 *   3 UPDH   C = (C * E1[m]) * ((acc + ktailA[m] * ktailB[n]) / (E2[m] + alpha + eps)), also stored transposed into Ct
 * tailA != NULL adds output row `tail_row` = sum_r tailA[r] * B[r][n], computed on the VALU (F = 513 = 16 * 32 + 1), with the
 * epilogue applied (it goes to C, and to Ct as well for DIVT).  Nothing outside the M x N corner (plus the tail row) is written,
 * except zeros inside a 4-float group that straddles it.  tile: 0 = chosen by the library, 1..8 = a fixed tile (experiments).
 * The trailing six fields are filled in by the library. */
typedef struct gccnmf_direct_gemm {
    const float* A;
    const float* B;
    long sA, sB;
    int lda, ldb;
    int M, N, Kd, batch;
    const float* bscale;
    long s_bscale;
    const float* tailA;
    long s_tailA;
    int tail_row;
    float* rowsumB;
    long s_rowsumB;
    float* C;
    long sC;
    int ldc;
    float* Ct;
    long sCt;
    int ldct;
    const float* E0;
    long sE0;
    int lde0;
    const float* E1;
    long sE1;
    const float* E2;
    long sE2;
    const float* ktailA;
    const float* ktailB;
    long s_ktailA, s_ktailB;
    float alpha, eps;
    int tiles_m, tiles_n, xc, sm, sn;
    long long* trace;          /* library-filled: per-workgroup timeline while gccnmf_debug_set_trace is armed */
} gccnmf_direct_gemm;
int gccnmf_gemm_direct(const gccnmf_direct_gemm* desc, int epilogue, int tile, void* stream);

/* Diagnostics (tests only, no device needed): the work lists a throughput-tile launch (csrc/gemm_dma.h) would use for an M x N output
 * over `batch` files under the current tuning -- the SAME tile decode the kernel runs, executed on the host.
 *   plan  [8]: lists, wide tiles per list, ragged narrow tiles per list, split, rag, tiles_m, tiles_n, items of the classic grid
 *   items [max_items][6]: list, ticket, file, row tile, first column, column blocks (2 = 512 x 64, 1 = a narrow 512 x 32 item)
 *   narrow_capable: bit 0 = the kernel instantiation carries the narrow loop, bit 1 = the whole-file lists of chained launches (list x =
 *   files x, x + 8, ..., each file's wide tiles followed by its ragged items; plan[1] = items of the longest list)
 * Returns the number of items (all lists), -1 on bad arguments. */
int gccnmf_debug_gemm_plan(int M, int N, int batch, int xcd_affine, int concurrent, int narrow_capable, int* plan, int* items, int max_items);

/* Diagnostics (tests only): run one MFMA GEMM configuration in isolation.
 * layout bits: 1 = A reduction-contiguous, 2 = B reduction-contiguous, 4 = VALU tail row, 8 = <1,4> wave grid,
 * 16 = last reduction index as a rank-1 epilogue term (non-KC operands). */
int gccnmf_debug_gemm(const float* A, const float* B, float* C, int M, int N, int Kd, int lda, int ldb, int ldc,
                      int a_clamp, int b_clamp, int layout, int batch, long sA, long sB, long sC,
                      const float* bscale, float* rowsumB, void* stream);

#ifdef __cplusplus
}
#endif
#endif
