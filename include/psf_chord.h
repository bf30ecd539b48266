/*
 * psf_chord.h — C ABI of libpsf_chord.so: PSF-Attn's chord-sparse batched matmul on MI355X (gfx950).
 *
 * This is the drop-in boundary for ONE path of RuslanKhalitov/SparseFactorization: the call
 *
 *     V = spmm(chord_indicies, W.reshape(B, N*L), N, N, V)          (+ V = V + res_conn)
 *
 * that PSFNet.forward executes n_W times per forward
 *   (SyntheticExperiments/psf.py:172-188, LRA/psf.py:224-240, Genome_Clf/psf.py:214-230,
 *    attention_block.py:158-174, LRA/attention_maps/pathfinder_inference.py:66-81, imdb_inference.py:45-59),
 * and the reference's own optional native statement of the same operator and its gradients
 *   (spmul/spmul_cuda.cu:31-59 forward_host, :114-159 backward_host; spmul/spmul.py:12-31 SparseMultiply).
 *
 * Operator.  With link offsets off[0..L) (chord pattern: off[0]=0, off[k]=2^(k-1); taken mod N):
 *
 *   fwd : out[b,p,c] = sum_{k<L} W[b,p,k] * V[b,(p+off[k]) mod N, c]   (+ res[b,p,c])        spmul_cuda.cu:24
 *   dV  : dV [b,q,c] = sum_{k<L} W[b,(q-off[k]) mod N,k] * dZ[b,(q-off[k]) mod N, c]         spmul_cuda.cu:79-80
 *   dW  : dW [b,p,k] = sum_{c<C} dZ[b,p,c] * V[b,(p+off[k]) mod N, c]                        spmul_cuda.cu:105-108
 *
 * Sums run k (resp. c) ascending from 0 with a rounded product followed by a rounded add (no FMA
 * contraction), i.e. the order a CPU scatter_add over the reference's row-major (i,k) index list produces
 * (SyntheticExperiments/psf.py:7-32), so fp32 results are reproducible bit for bit against the oracle.
 *
 * Conventions
 *   - every data pointer is a DEVICE pointer to a contiguous row-major array; `offsets` and the step
 *     pointer tables are HOST pointers, read before the call returns;
 *   - the caller owns every buffer; the library allocates nothing on the device, keeps no pointer and
 *     holds no state besides a thread-local error string and process-wide tuning knobs;
 *   - work is enqueued asynchronously on `stream` (a hipStream_t passed as void*, NULL = default stream)
 *     of the calling thread's current device; no call synchronises;
 *   - outputs are fully overwritten (no pre-zero needed) and must not alias the gathered input
 *     (out != V, dV != dZ): rows are gathered from other rows of the same array;
 *   - return value: 0 = success; < 0 = invalid argument (PSF_E_*); > 0 = hipError_t of the failed HIP call.
 *     No C++ exception crosses this boundary. psf_last_error() describes the last failure on this thread.
 *
 * Threads.  Every entry point may be called from several host threads at once, on the same or on different streams
 * and devices: a call reads its arguments, takes ONE snapshot of the tuning knobs at entry (a psf_set_tuning from another
 * thread changes the next call, never the middle of one), enqueues its kernels and returns; the error string is
 * thread-local; nothing else is written. What the library does NOT do for the caller: order two calls that touch the same
 * buffers (that is the streams' job), or make psf_set_tuning atomic across several knobs (set them before starting the
 * worker threads). The first launch of a kernel instance that needs more than 48 KB of LDS raises that function's limit
 * (hipFuncSetAttribute, idempotent; two threads racing there both succeed).
 *
 * Streams.  Calls on different streams of one device may run at the same time and share compute units; no entry point
 * needs the device, a CU or a cache to itself, and no kernel's result depends on what runs beside it. This is tested, not
 * assumed: the forward chain at BASELINE configs[1]'s full size and the fused backward step are bit-identical to their solo
 * runs (and the chain to the CPU oracle) while the producer MLP kernels — MFMA phases, the one kind of neighbour the round-4
 * investigation ever suspected (profiles/r04b_mixer_lds_wait.md) — run on a second stream over the same CUs, and the other way
 * round (tests/test_gpu_coresidence.py). Two streams buy no time here (both kernel families are bound by the CUs' issue
 * slots and L2 ports: the two take the sum of their times, profiles/r05y_concurrent_streams_lab.log); the point is that
 * a caller who does it gets the same bits.
 *
 * Reference binding this replaces: pybind11 module `spmul_cuda` {forward_host, backward_host}
 * (spmul/spmul_cuda.cu:163-166) and the Python call torch_sparse.spmm (SyntheticExperiments/psf.py:5,178).
 * See INTEGRATION.md for the ctypes stub a maintainer of the reference would add.
 */
#ifndef PSF_CHORD_H
#define PSF_CHORD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PSF_ABI_VERSION 2 /* 2 (round 5): the three "far" training entries are gone, psf_device_info is new, per-step mixer kernels take rows only */

#define PSF_MAX_LINKS 64 /* L <= 64: N = 2^63 would need 64 links */

enum {
  PSF_OK = 0,
  PSF_E_NULL = -1,     /* a required pointer is NULL */
  PSF_E_SHAPE = -2,    /* B, N, L or C out of range */
  PSF_E_ALIAS = -3,    /* output aliases a gathered input */
  PSF_E_ALIGN = -4,    /* pointer not aligned to its element size */
  PSF_E_OFFSET = -5,   /* an explicit offset is not representable */
  PSF_E_TUNING = -6,   /* unknown tuning key / value */
  PSF_E_UNSUPPORTED = -7 /* this entry point has no kernel for the shape (psf_chord_chain_bwd_f32): not an error */
};

/* ABI version of the loaded library (== PSF_ABI_VERSION it was built with). */
int psf_version(void);

/* Human-readable description of the last failure on the calling thread ("" if none). Never NULL. */
const char* psf_last_error(void);

/* Build description: target arch, compiler, kernel variants compiled in. Never NULL. */
const char* psf_build_info(void);

/*
 * The calling thread's current HIP device, as the library sees it: "pci=<domain:bus:device.function> xcds=<n> cus=<n>
 * name=<marketing name>" into buf (NUL-terminated, truncated to len). One rank per GPU prints this into its benchmark line
 * so that a multi-GPU record shows N distinct devices (bench.py: "devices"). Touches the GPU (hipGetDevice).
 */
int psf_device_info(char* buf, int32_t len);

/*
 * Chord link offsets — the integer pattern of get_chord_indices_assym (SyntheticExperiments/psf.py:7-32):
 * offsets_out[0] = 0, offsets_out[k] = 2^(k-1) mod N for 1 <= k < L. Host-only, no GPU needed.
 */
int psf_chord_offsets(int64_t N, int32_t L, int64_t* offsets_out);

/*
 * The full COO index list of get_chord_indices_assym(n_vec=N, n_link=L): rows_out[i*L+k] = i,
 * cols_out[i*L+k] = (i + off[k]) mod N. Both arrays hold N*L entries. Host-only. Duplicate links
 * (off[k] == off[j] mod N) are kept, as in the reference.
 */
int psf_chord_indices(int64_t N, int32_t L, int64_t* rows_out, int64_t* cols_out);

/*
 * Forward step  out = W (.) V  [+ res].
 *   W   [B,N,L]   link weights (the reference's `value` reshaped, spmm arg 2 / `F` of spmul_cuda.cu:32)
 *   V   [B,N,C]   or [N,C] broadcast over the batch when v_batch_stride == 0 (the unbatched
 *                 eye(N) first operand of pathfinder_inference.py:57,75-81); otherwise
 *                 v_batch_stride must be N*C (elements)
 *   res [B,N,C]   or NULL; fused `V = V + res_conn` of SyntheticExperiments/psf.py:187-188
 *   out [B,N,C]
 *   offsets       host array of L link offsets (any int64, reduced mod N here), or NULL for the
 *                 chord pattern of psf_chord_offsets (what every PSFNet call site uses);
 *                 an explicit array is the `offsets` argument of spmul/spmul.py:8-9,15
 */
int psf_chord_spmm_fwd_f32(const float* W, const float* V, const float* res, float* out,
                           int64_t B, int64_t N, int32_t L, int64_t C, int64_t v_batch_stride,
                           const int64_t* offsets, void* stream);
int psf_chord_spmm_fwd_f64(const double* W, const double* V, const double* res, double* out,
                           int64_t B, int64_t N, int32_t L, int64_t C, int64_t v_batch_stride,
                           const int64_t* offsets, void* stream);

/*
 * Backward step (spmul_cuda.cu:114-159 backward_host). Either output may be NULL to skip it.
 *   dZ [B,N,C]  gradient w.r.t. the step's output (also the gradient w.r.t. `res`, which is the identity)
 *   W  [B,N,L]  needed for dV;   V [B,N,C] or [N,C] (v_batch_stride == 0) needed for dW
 *   dW [B,N,L]
 *   dV [B,N,C]  always per batch element; when V was broadcast the caller sums dV over b
 */
int psf_chord_spmm_bwd_f32(const float* dZ, const float* W, const float* V, float* dW, float* dV,
                           int64_t B, int64_t N, int32_t L, int64_t C, int64_t v_batch_stride,
                           const int64_t* offsets, void* stream);
int psf_chord_spmm_bwd_f64(const double* dZ, const double* W, const double* V, double* dW, double* dV,
                           int64_t B, int64_t N, int32_t L, int64_t C, int64_t v_batch_stride,
                           const int64_t* offsets, void* stream);

/*
 * Whole forward chain of PSFNet.forward's hot loop (SyntheticExperiments/psf.py:172-188):
 *     X_0 = V0;   X_{m+1} = W_m (.) X_m  [+ V0 if use_residual]      m = 0 .. M-1
 *   W_steps   host table of M device pointers, W_steps[m] -> [B,N,L]
 *   out_steps host table of M device pointers, out_steps[m] -> [B,N,C] receives X_{m+1}.
 *             Training keeps all M (they are the saved inputs of the backward steps); inference may
 *             alternate two buffers. out_steps[m] must differ from the step's input
 *             (V0 for m = 0, out_steps[m-1] after) and, with use_residual, from V0.
 *   V0 [B,N,C] or [N,C] with v0_batch_stride == 0 (then use_residual must be 0)
 * Launches M dependent kernels on `stream` — or, for short sequences, a single kernel that keeps each
 * sequence's X in LDS across all M steps (a step buffer that a later step overwrites, as in two-buffer
 * inference, is then not written at all). The result is out_steps[M-1].
 */
int psf_chord_chain_fwd_f32(const float* const* W_steps, const float* V0, float* const* out_steps,
                            int32_t M, int32_t use_residual,
                            int64_t B, int64_t N, int32_t L, int64_t C, int64_t v0_batch_stride,
                            const int64_t* offsets, void* stream);
int psf_chord_chain_fwd_f64(const double* const* W_steps, const double* V0, double* const* out_steps,
                            int32_t M, int32_t use_residual,
                            int64_t B, int64_t N, int32_t L, int64_t C, int64_t v0_batch_stride,
                            const int64_t* offsets, void* stream);

/*
 * Whole backward chain of the same loop in one call (SyntheticExperiments/psf.py:172-188 differentiated;
 * spmul/spmul_cuda.cu:75-84,102-111 per step):
 *   - ONE launch for short sequences of narrow rows (N <= 1024, C = 4 or 8, 2 <= L <= 20, M <= 64: the synthetic tasks up to
 *     N = 1024); psf_chord_chain_bwd_supported says whether the shape is covered (1 / 0); dX_steps may be NULL then;
 *   - otherwise the M per-step launches of psf_chord_spmm_bwd_f32 and one psf_sum_tensors_f32 pass, issued by the library
 *     (no trip through the caller's FFI per step): needs dX_steps, a host table of M device buffers [B,N,C] that receive
 *     the gradient after each step (dX_steps[m] = gradient of X_m; dX_steps[0] is unused without the residual);
 *   - PSF_E_UNSUPPORTED (not an error) when neither applies — dX_steps NULL for an uncovered shape, a residual chain of more
 *     than 31 steps or B*N*C not a multiple of 4, knob "chain_bwd_fused" = 0 — and the caller runs the steps itself.
 *   dOut [B,N,C]  gradient of the chain's result X_M
 *   W_steps[m] -> W_m [B,N,L];   X_steps[m] -> X_m [B,N,C], the forward's input of step m: X_steps[0] is ignored (V0 is
 *   used), X_steps[m] = psf_chord_chain_fwd_f32's out_steps[m-1];   dW_steps[m] -> dW_m [B,N,L] (written)
 *   dV0 [B,N,C] (written): gradient of V0 through the chain and, with use_residual, through every step's "+ V0" as well,
 *   summed in the per-step path's order ((g_M + g_{M-1}) + ... + g_1) + g_0.
 * The one launch's dV0 and dW are bit-identical to the CPU oracle (products and sums rounded separately, links / channels
 * ascending); the per-step path's are those of psf_chord_spmm_bwd_f32.
 */
int psf_chord_chain_bwd_supported(int64_t N, int32_t L, int64_t C, int32_t M);
int psf_chord_chain_bwd_f32(const float* dOut, const float* const* W_steps, const float* V0, const float* const* X_steps,
                            float* const* dW_steps, float* dV0, float* const* dX_steps, int32_t M, int32_t use_residual,
                            int64_t B, int64_t N, int32_t L, int64_t C, const int64_t* offsets, void* stream);

/*
 * (Rounds 2-4 also exported training variants that kept a link-major side copy of W's far columns for the dV kernel —
 * "..._chain_fwd_far_f32", "..._spmm_bwd_far_f32", "..._bwd_far_first_link". A wash end to end
 * (profiles/r02n_far_copy.log, r03aj_farcopy_in_step_ab.log), off by default since round 2, removed in round 5.)
 */

/*
 * Producer side: weight / bias gradient of the token-wise Linear layers of MLPBlock
 * (SyntheticExperiments/psf.py:35-60) that produce W_m = fs[m](data) and V = g(data) — a reduction over
 * T = B*N tokens into an n x m tile, done on the f32 matrix core (exact f32) with a fixed-order reduction:
 *     dWt[j,i] = sum_t dY[t,j] * X[t,i]     (nn.Linear.weight.grad layout, [n = out_features, m = in_features])
 *     db[j]    = sum_t dY[t,j]              (db may be NULL)
 *   X [T,m], dY [T,n] contiguous; 1 <= m, n <= 128. `workspace` is caller-owned device scratch of at least
 *   psf_linear_wgrad_workspace(T, m, n) bytes (that function returns -1 for unsupported sizes).
 */
int64_t psf_linear_wgrad_workspace(int64_t T, int32_t m, int32_t n);
int psf_linear_wgrad_f32(const float* X, const float* dY, int64_t T, int32_t m, int32_t n, float* dWt, float* db,
                         void* workspace, int64_t workspace_bytes, void* stream);
/* Same with row strides (in floats; ldx >= m, ldy >= n): X and dY may be column slices of wider row-major arrays,
 * e.g. one MLP's hidden block inside the stacked hidden layer of all MLPs. */
int psf_linear_wgrad_strided_f32(const float* X, int64_t ldx, const float* dY, int64_t ldy, int64_t T, int32_t m, int32_t n,
                                 float* dWt, float* db, void* workspace, int64_t workspace_bytes, void* stream);

/*
 * out[i] = ((srcs[0][i] + srcs[1][i]) + srcs[2][i]) + ...   i < n, in that order (separately rounded additions).
 * The gradient of the chain's residual: dV_0 = dX_0 + sum over the steps of dX_m (V = V + res_conn,
 * SyntheticExperiments/psf.py:187-188, reaches V_0 from every step), summed once instead of accumulated per step.
 *   1 <= count <= 32; n a multiple of 4; every pointer 16-byte aligned; `out` may alias none of the sources.
 */
int psf_sum_tensors_f32(const float* const* srcs, int32_t count, int64_t n, float* out, void* stream);

/*
 * One Adam step over `count` tensors (the optimizer of the reference's training loop, optim.Adam(net.parameters(), lr),
 * SyntheticExperiments/psf_training.py:50-53; no weight decay, no amsgrad):
 *     m += (g - m)(1 - beta1);  v = beta2 v + (1 - beta2) g^2;  p -= lr / (1 - beta1^t) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps)
 *   host tables of `count` device pointers (params, grads, exp_avg, exp_avg_sq) and element counts; every pointer 4-byte
 *   aligned (16-byte alignment and counts that are multiples of 4 take the vector path). `step` = t >= 1 from the host,
 *   or, when `step_dev` is non-NULL, t is read from that device float (a captured step must see it advance). 4096
 *   elements per workgroup: the two 524 288-element tensors of a PSFNet at N = 16384 spread over 256 workgroups.
 */
int psf_adam_step_f32(float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                      const int64_t* numels, int32_t count, float lr, float beta1, float beta2, float eps, float step,
                      const float* step_dev, void* stream);

/*
 * Token embedding fused with the positional-embedding add — the first two lines of PSFNet.forward
 * (SyntheticExperiments/psf.py:152-163, LRA/psf.py:203-214):
 *     out[t,:] = table[idx[t],:] (+ pos[t mod N,:])        t < T = B*N
 *   idx [T] int64 token ids (ids outside [0,V) are clamped: never an out-of-bounds read), table [V,E],
 *   pos [N,E] or NULL, out [T,E]; E a multiple of 4; table, pos, out 16-byte aligned. One pass over the output.
 */
int psf_embed_tokens_f32(const int64_t* idx, const float* table, const float* pos, float* out, int64_t T, int64_t N,
                         int32_t V, int32_t E, void* stream);
/*
 * Rows of a narrow affine input layer — `init_linear` of the synthetic PSFNet (SyntheticExperiments/psf.py:153-154:
 * Linear(2, embedding_size) on [value, marker]):
 *     out[t,:] = x[t,0..K) W^T + bias          t < T
 *   x [T,K] with 1 <= K <= 3, W [E,K], bias [E] or NULL, out [T,E] 16-byte aligned, E a multiple of 4, at most 1024. One pass over the
 *   output (the K = 2 product is a 134 MB write at 1 M positions, not a GEMM). Arithmetic: x_0 w_e0, fused adds of x_1 w_e1
 *   and x_2 w_e2, one rounded add of the bias — the same as the mixer entry's PSF_MIXER_IN_AFFINE recipe, bit for bit.
 */
int psf_affine_rows_f32(const float* x, const float* W, const float* bias, float* out, int64_t T, int32_t K, int32_t E,
                        void* stream);
/*
 * Its gradient with respect to the table:  dTable[v,:] = sum over {t : idx[t] == v} of dOut[t,:]   (fully written;
 * the caller zeroes a padding row). No sort, no atomics, no host read-back: deterministic (a fixed order of
 * additions) and capturable in a HIP graph, unlike the sort-and-partition backward of nn.Embedding.
 *   1 <= V <= 512, 1 <= E <= 4096; `workspace`: at least psf_embed_tokens_bwd_workspace(T, V, E) bytes (-1 for
 *   unsupported sizes).
 */
int64_t psf_embed_tokens_bwd_workspace(int64_t T, int32_t V, int32_t E);
int psf_embed_tokens_bwd_f32(const int64_t* idx, const float* dOut, int64_t T, int32_t V, int32_t E, float* dTable,
                             void* workspace, int64_t workspace_bytes, void* stream);

/*
 * The FLATTEN head of PSFNet, `final = nn.Linear(n_vec * n_channels_V, n_class)` on V.view(B, -1)
 * (SyntheticExperiments/psf.py:129-134,189-190):  out[b,j] = bias[j] + sum_i X[b,i] * W[j,i]
 *   X [B,K] (K = N*C, a multiple of 4), W [J,K] (nn.Linear.weight), bias [J] or NULL, out [B,J]; 1 <= J <= 8;
 *   X and W 16-byte aligned. `workspace`: at least psf_flat_head_workspace(B, K, J) bytes (-1 for unsupported
 *   sizes). One read of X; per-chunk partial sums are added in a fixed order: bit-reproducible run to run FOR A GIVEN
 *   (B, K). The chunk length (4096 floats, or 1024 when 4096 would leave fewer than 1024 workgroups) depends on B and K, so
 *   the logits of one sample differ in their last bits between batch sizes that fall on different sides of that rule
 *   (evaluation at another batch size than training agrees to ~1e-6 relative, not to the bit).
 */
int64_t psf_flat_head_workspace(int32_t B, int64_t K, int32_t J);
/*
 * Its backward (what autograd computes for that nn.Linear):  dW[j,i] = sum_b dY[b,j] * X[b,i],  dX[b,i] = sum_j dY[b,j] * W[j,i]
 *   dY [B,J] contiguous; dX [B,K] and / or dW [J,K] (NULL = not wanted; dW needs X, dX needs W); 1 <= B <= 1024,
 *   1 <= J <= 16 (the first layer of CIFAR-10's non-linear head has 16), K a positive multiple of 4; X, W, dX, dW 16-byte aligned. One read of X, sums over b in ascending order
 *   (bit-reproducible). The bias gradient (the column sums of dY) is left to the caller.
 */
int psf_flat_head_bwd_f32(const float* dY, const float* X, const float* W, float* dX, float* dW, int32_t B, int64_t K, int32_t J,
                          void* stream);
int psf_flat_head_f32(const float* X, const float* W, const float* bias, float* out, int32_t B, int64_t K, int32_t J,
                      void* workspace, int64_t workspace_bytes, void* stream);

/*
 * Producer side, forward (inference): K two-layer token-wise MLPs sharing one input, fused in one launch —
 * g and fs[0..M) of PSFNet (MLPBlock = Linear, GELU, Linear; SyntheticExperiments/psf.py:35-60,110-126,165,175):
 *     Y[k][t,:] = GELU(X[t,:] * A[k]^T + a[k]) * B[k]^T + b[k]          (erf GELU as torch.nn.GELU(); erf is
 *                                                                          evaluated to 1.5e-7 absolute)
 *   X [T,E]; A[k] [h[k],E], a[k] [h[k]], B[k] [O[k],h[k]], b[k] [O[k]] (nn.Linear layouts); Y[k] [T,O[k]].
 *   A, a, B, b, Y, h, O are HOST tables of K entries (device pointers / sizes).
 *   Limits: E a multiple of 4, 4 <= E <= 64; 1 <= h[k] <= 128; 1 <= O[k] <= 32; 1 <= K <= 32; X 16-byte aligned.
 *   `workspace`: caller-owned, 16-byte-aligned device scratch of at least psf_mlp_fwd_workspace(E, K, h, O)
 *   bytes (the packed weight images; that function returns -1 for unsupported sizes).
 * X is read once, the hidden activations never reach memory; f32 throughout (f32 matrix core).
 */
int64_t psf_mlp_fwd_workspace(int32_t E, int32_t K, const int32_t* h, const int32_t* O);
int psf_mlp_fwd_f32(const float* X, int64_t T, int32_t E, int32_t K, const float* const* A, const float* const* a,
                    const float* const* B, const float* const* b, const int32_t* h, const int32_t* O,
                    float* const* Y, void* workspace, int64_t workspace_bytes, void* stream);

/*
 * The mixer with W produced INSIDE the chain step: V_M from `data` without any W_m in memory.
 * What PSFNet.forward does between the embedding and the head (SyntheticExperiments/psf.py:165-188, LRA/psf.py:214-240,
 * Genome_Clf/psf.py:204-230, attention_block.py:148-174; dropout2 between g and the loop must be inactive):
 *     V0 = g(X);   for m < M:  W_m = fs[m](X);  X_{m+1} = W_m (.) X_m [+ V0]            (X_0 = V0)
 * Each step is ONE kernel that computes its tile's rows of W_m on chip — `data` rows straight into matrix-core operand
 * order, the two layers of fs[m] on the bf16 matrix pipe at f32 accuracy (psf_mlp_fwd_f32's arithmetic), the W tile in LDS
 * — and then gathers and accumulates as psf_chord_spmm_fwd_f32 does (same order, uncontracted multiply and add). The
 * producer's W writes (4 M L bytes per token) and the chain's W reads never happen; a step reads the data row (4 E bytes)
 * instead of the W row (4 L bytes).
 *   X [B,N,E]; MLP 0 is g (E -> h[0] -> C), MLPs 1..M are fs[0..M) (E -> h[k] -> L): A, a, Bw, b, h are HOST tables of
 *   M + 1 entries laid out as for psf_mlp_fwd_f32 (nn.Linear layouts).
 *   V0 [B,N,C] receives g(X) (it is also the residual); out_steps as psf_chord_chain_fwd_f32 (may alternate two buffers);
 *   the result is out_steps[M-1]. Chord offsets only.
 *   Limits: E a multiple of 4, 4 <= E <= 32; 1 <= h[k] <= 128; C a multiple of 4, 4 <= C <= 32; 4 <= L <= 20; 1 <= M <= 31;
 *   N at least two tiles (tile = 256, 256, 128, 64 rows for C <= 4, 8, 16, 32); X, V0 and out_steps 16-byte aligned.
 *   Short sequences (N a multiple of 32, N * C / 4 <= 1024, C = 4 or 8 — BASELINE configs[0], Adding N = 128, among them) run
 *   as ONE launch in which V also stays on chip: a workgroup owns a sequence, keeps its data rows in registers and its V in
 *   LDS through all M steps (csrc/mixer_lds.h; knob "mixer_lds"); a step buffer that a later step overwrites is then not
 *   stored at all, as in psf_chord_chain_fwd_f32.
 *   psf_mixer_fwd_workspace returns the bytes of 16-byte-aligned device scratch needed (packed weight images), or -1 when
 *   the fused path does not cover the shape — the caller then uses psf_mlp_fwd_f32 + psf_chord_chain_fwd_f32.
 */
int64_t psf_mixer_fwd_workspace(int64_t N, int32_t E, int32_t M, const int32_t* h, int64_t C, int32_t L);
/* Which form a call with this shape takes: 0 = none (outside the limits), 1 = M + 1 step launches, 2 = the single LDS-resident launch. */
int32_t psf_mixer_fwd_plan(int64_t N, int32_t E, int32_t M, const int32_t* h, int64_t C, int32_t L);
int psf_mixer_fwd_f32(const float* X, int64_t B, int64_t N, int32_t E, int32_t M, const float* const* A,
                      const float* const* a, const float* const* Bw, const float* const* b, const int32_t* h, int64_t C,
                      int32_t L, int32_t use_residual, float* V0, float* const* out_steps, void* workspace,
                      int64_t workspace_bytes, void* stream);

/*
 * The same with `data` itself never in memory: X is replaced by the recipe PSFNet.forward computes it by, and every kernel
 * of the mixer evaluates the rows it needs from that recipe (a step then reads 8 bytes per position instead of 4 E).
 *   PSF_MIXER_IN_DATA    src = float X [B,N,E]                                        (psf_mixer_fwd_f32)
 *   PSF_MIXER_IN_AFFINE  src = float in [B,N,K], K <= 3;  X = in * weight^T + bias  (+ pos[p])
 *                        weight [E,K], bias [E] or NULL — `init_linear` of the Adding network,
 *                        SyntheticExperiments/psf.py:136-141,153-154; the K products are summed first, then the bias
 *   PSF_MIXER_IN_TOKENS  src = int64 tokens [B,N] in [0, K);  X = weight[token] (+ pos[p])
 *                        weight = the embedding table [K,E], pos [N,E] or NULL — psf.py:151-152,157-162, LRA/psf.py:204-209,
 *                        attention_block.py:150-152; one rounded add, as psf_embed_tokens_f32. Tokens are device data and
 *                        cannot be range-checked by the host: an index outside [0, K) is CLAMPED into the table (row 0
 *                        or row K-1, as psf_embed_tokens_f32 does) — never a read outside it; nn.Embedding would raise.
 *   weight and pos 16-byte aligned, src aligned to its element type. Everything else as psf_mixer_fwd_f32.
 */
enum { PSF_MIXER_IN_DATA = 0, PSF_MIXER_IN_AFFINE = 1, PSF_MIXER_IN_TOKENS = 2 };
typedef struct psf_mixer_input {
  int32_t kind; /* PSF_MIXER_IN_* */
  int32_t K;    /* AFFINE: inputs per position (1..3); TOKENS: vocabulary size; DATA: ignored */
  const void* src;
  const float* weight;
  const float* bias;
  const float* pos;
} psf_mixer_input;
int psf_mixer_fwd_in_f32(const psf_mixer_input* in, int64_t B, int64_t N, int32_t E, int32_t M, const float* const* A,
                         const float* const* a, const float* const* Bw, const float* const* b, const int32_t* h, int64_t C,
                         int32_t L, int32_t use_residual, float* V0, float* const* out_steps, void* workspace,
                         int64_t workspace_bytes, void* stream);

/*
 * Producer side, backward (training) of the same K MLPs, fused in one pass over the tokens. What autograd does
 * for MLPBlock (SyntheticExperiments/psf.py:35-60) with 4K GEMMs, K GELU-backward kernels and K-1 accumulations
 * of the input gradient:
 *     Hpre  = X * A[k]^T + a[k]            (recomputed; nothing of size [T, h] is saved by the forward)
 *     dHpre = (dY[k] * B[k]) .* GELU'(Hpre)
 *     dA[k] = dHpre^T * X,  da[k] = sum_t dHpre,  dB[k] = dY[k]^T * GELU(Hpre),  db[k] = sum_t dY[k]
 *     dX    = sum_k dHpre_k * A[k]                                               (dX may be NULL: not computed)
 *   Layouts as psf_mlp_fwd_f32; dY[k] [T,O[k]]; dA/da/dB/db[k] have the nn.Linear parameter shapes and are fully
 *   overwritten. Limits: E a multiple of 4, 4 <= E <= 32; 1 <= h[k] <= 128; 1 <= O[k] <= 32; 1 <= K <= 32; X and
 *   dX 16-byte aligned. `workspace`: at least psf_mlp_bwd_workspace(T, E, K, h, O) bytes, 16-byte aligned (packed
 *   weights + per-wave partial sums; -1 for unsupported sizes).
 * Weight gradients are reduced in a fixed order (no float atomics): results are bit-reproducible run to run.
 */
int64_t psf_mlp_bwd_workspace(int64_t T, int32_t E, int32_t K, const int32_t* h, const int32_t* O);
int psf_mlp_bwd_f32(const float* X, int64_t T, int32_t E, int32_t K, const float* const* A, const float* const* a,
                    const float* const* B, const int32_t* h, const int32_t* O, const float* const* dY, float* dX,
                    float* const* dA, float* const* da, float* const* dB, float* const* db, void* workspace,
                    int64_t workspace_bytes, void* stream);

/*
 * Producer side at the WIDE (LRA) sizes — the same K MLPs for E up to 1024 and outputs up to 128: reference ListOps
 * E = 512, h = 128, outputs 12 (x 11 link MLPs) and 128 (LRA/psf_training_config.py:2-30; MLPBlock LRA/psf.py:35-60, call
 * sites LRA/psf.py:214,227). The K first layers run as ONE stacked GEMM on the bf16 matrix pipe at f32 accuracy (every
 * f32 operand split exactly into three bf16 terms, six product terms), forward, input gradient and weight gradient
 * alike; the second layers, GELU and its derivative are fused around them.
 *   psf_mlp_wide_fwd_f32  Y[k] as psf_mlp_fwd_f32. It also fills `saved` (caller-owned, 256-byte aligned, at least
 *                         psf_mlp_wide_saved_bytes bytes): X as bf16 term planes and the pre-activations of the hidden
 *                         layers — what the backward needs, so that nothing is recomputed (at these widths recomputing
 *                         is a fourth 100-GFLOP GEMM; keeping is 4 x sum(h) bytes per token). Inference passes saved =
 *                         NULL and a workspace of psf_mlp_wide_fwd_workspace + psf_mlp_wide_saved_bytes bytes.
 *   psf_mlp_wide_bwd_f32  the gradients of psf_mlp_bwd_f32 from `saved` and dY[k]; dX may be NULL (not computed).
 *   Layouts as psf_mlp_fwd_f32 / psf_mlp_bwd_f32. Limits: E a multiple of 16, 16 <= E <= 1024; 1 <= h[k] <= 128;
 *   1 <= O[k] <= 128; 1 <= K <= 24; T * E and T * (sum of h[k] rounded up to 32) below 2^30; X and A[k] 16-byte aligned.
 *   Workspaces: caller-owned, 256-byte aligned, at least psf_mlp_wide_{fwd,bwd}_workspace bytes (-1: unsupported sizes).
 * Every reduction has a fixed order (no float atomics): results are bit-reproducible run to run.
 */
int64_t psf_mlp_wide_saved_bytes(int64_t T, int32_t E, int32_t K, const int32_t* h, const int32_t* O);
int64_t psf_mlp_wide_fwd_workspace(int64_t T, int32_t E, int32_t K, const int32_t* h, const int32_t* O);
int64_t psf_mlp_wide_bwd_workspace(int64_t T, int32_t E, int32_t K, const int32_t* h, const int32_t* O);
int psf_mlp_wide_fwd_f32(const float* X, int64_t T, int32_t E, int32_t K, const float* const* A, const float* const* a,
                         const float* const* B, const float* const* b, const int32_t* h, const int32_t* O,
                         float* const* Y, void* saved, int64_t saved_bytes, void* workspace, int64_t workspace_bytes,
                         void* stream);
int psf_mlp_wide_bwd_f32(const void* saved, int64_t saved_bytes, int64_t T, int32_t E, int32_t K,
                         const float* const* A, const float* const* B, const int32_t* h, const int32_t* O,
                         const float* const* dY, float* dX, float* const* dA, float* const* da, float* const* dB,
                         float* const* db, void* workspace, int64_t workspace_bytes, void* stream);

/*
 * A plain streaming kernel with the forward step's byte mix, for measuring what the memory system of a box gives that
 * mix (bench.py's second denominator; no reference counterpart). Per 16-byte output vector i it reads w[2i], w[2i+1], v[i]
 * and r[i] and writes out[i]: W : V : residual : out = 2 : 1 : 1 : 1 (the step kernel at L = 15, C = 8 moves 15 : 8 : 8 : 8).
 *   w [2 n_vec4] float4, v / r / out [n_vec4] float4, all 16-byte aligned; bytes moved = 80 n_vec4.
 */
int psf_stream_mix_f32(const float* w, const float* v, const float* r, float* out, int64_t n_vec4, void* stream);
/*
 * The same for the BACKWARD step's byte mix: per index i it reads w[2i], w[2i+1], v[i], z[i] and writes dw[2i], dw[2i+1],
 * dv[i] — W : V : dZ read, dW : dV written = 2 : 1 : 1, 2 : 1 (the fused backward step at L = 15, C = 8 moves 15 : 8 : 8,
 * 15 : 8). Bytes moved = 112 n_vec4.
 */
int psf_stream_mix_bwd_f32(const float* w, const float* v, const float* z, float* dw, float* dv, int64_t n_vec4,
                           void* stream);

/*
 * Process-wide tuning knobs (benchmark / test use; defaults are the shipped configuration).
 *   key "fwd_variant": 0 = auto, 1 = generic direct-gather kernel, 2 = LDS-window kernel
 *   key "bwd_variant": 0 = auto (LDS-window dV / dW kernels where they apply), 1 = generic kernels
 *   key "fwd_split"  : window kernels (forward, dV, dW) on shapes with a ragged last tile per sequence: 1 = by size
 *                      (default: one launch of the general (edge) instance below ~300 MB of algorithmic bytes, where the
 *                      ~2.7 us of a second launch outweigh predicating every tile; above that as 2); 2 = full tiles on
 *                      the predicate-free instance + the ragged tiles in a second launch; 0 = every tile on the edge
 *                      instance
 *   key "dw_variant" : dW: 0 = auto (chunk-looping kernel for rows of >= 32 channels whose channel groups split into
 *                      chunks of 8, else the whole-row window kernel), 1 = whole-row window kernel, 2 = chunk-looping
 *                      kernel or PSF_E_TUNING where it does not apply
 *   key "dw_tgs"     : chunk-looping dW, lanes per row chunk: 0 = auto (8; 16 when that spares a ragged tile), 4 = 8, 5 = 16
 *   key "dv_threads" : dV window kernel: 0 = auto (512 threads x 1 row for C <= 8, else 256 threads x 2 rows), 1 = 256 threads
 *   key "bwd_fused"  : a backward step that wants both dW and dV: 1 (default) = ONE fused kernel for rows of C = 4, 8, 16, 32 or
 *                      64 channels (128 up to N = 4096) and sequences of at least two tiles — the dZ window staged once serves both gradients:
 *                      the aligned instance (N and the far offsets multiples of the tile, 16-byte-aligned chunk-clean
 *                      buffers: row-block addresses on the scalar unit) or else the general one (any N — N = 2^k + 1 with a
 *                      CLS token —, any far offsets, W / dW at any alignment); 2 = the aligned instance or the two kernels;
 *                      0 = always the two kernels
 *   key "fwd_wide"   : rows of >= 64 channels: 0 = automatic (default): the forward step takes 32-channel chunks on
 *                      1024-thread workgroups (256-row tiles) for rows of 64..256 channels and N <= 4096, one workgroup per
 *                      whole row otherwise; the backward kernels one workgroup per whole row; 1 = the 1024-thread chunks
 *                      wherever they fit (forward and dV); 2 = 32-channel chunks on 256-thread workgroups; 4 (and 3, whose
 *                      512-thread instance for rows of 32 channels was removed in round 5) = one workgroup per whole row
 *                      always
 *   key "chain_fused": 1 (default) = psf_chord_chain_fwd_f32 runs short sequences (N <= 2112, or 4160: see "chain_cc"; L <= 20, C % 4 == 0) as ONE
 *                      launch with the sequence resident in LDS: always when at most two step results are kept (inference
 *                      with alternating buffers), and when every step is kept (training) for sequences of at most 65536
 *                      elements (131072 for N <= 1024; 524288 where the eight-channel instance of "chain_cc" runs);
 *                      0 = always M per-step launches; 2 = the single launch wherever it fits. The workgroups of a
 *                      sequence share an XCD (knob "xcd_remap").
 *   key "fwd_wg_limit": LDS-window forward kernel, workgroups per CU: 0 = auto (3 for rows of <= 8 channels on
 *                      launches of >= 4096 tiles, else unlimited), 1 = unlimited, 2..4 = that many
 *   key "chain_zigzag": per-step launches of psf_chord_chain_fwd_*: 1 = every XCD walks its tile range forwards on
 *                      even steps and backwards on odd ones, so a launch starts on the tiles whose inputs the
 *                      previous launch wrote last (default); 0 = always forwards
 *   key "chain_cc"   : fused chain: channel groups (of 4 channels) per workgroup: 0 = automatic (default): two when the rows allow
 *                      (N <= 1056; N <= 2048 on the one-workgroup-per-CU instance when the launch keeps >= 256 workgroups;
 *                      that instance with one group also runs 2113 <= N <= 4160, the LRA text task, under the same
 *                      condition when only the last result is kept), 1 = one and no instance beyond N = 2112, 2 = the
 *                      one-workgroup-per-CU instances wherever they fit
 *   key "mlp_variant": psf_mlp_fwd_f32: 0 = auto (default: the split-bf16 kernel for E <= 32, else the f32-MFMA
 *                      kernel), 1 = f32 MFMA with streamed weights, 2 = f32 MFMA with LDS-resident weights,
 *                      3 = split-bf16 (each f32 operand as three exact bf16 terms, six product terms: f32 accuracy)
 *   key "bwd_fused_wg_limit": fused backward step: 0 = automatic (default): three workgroups per CU on launches of >= 4096
 *                      tiles, as many as fit below (five of 256 threads at C = 8); n = at most n (by requesting more LDS).
 *                      Round-4 kernel, rotating operands: 42.3 (what fits) / 40.9 (three) us at Order B = 40, 12.6 / 13.5 on
 *                      2048 tiles
 *   key "fwd_rows"   : LDS-window forward kernel, rows per thread: 0 = automatic (default): four for rows of 16..64 channels on
 *                      256-thread workgroups from N = 4096 on where the four-row tile divides N (one far link fewer: 2-6 %
 *                      per step), two otherwise; 2 / 4 = that many where compiled
 *   key "bwd_fronts" : fused backward step (full tiles): 0 = automatic (default): from N = 8192 on every XCD walks the tiles of
 *                      a batch element as TWO interleaved fronts half a sequence apart (the rows the longest link joins are
 *                      then in flight together: Order shape 41.1 -> 39.2 us per step); 1 = one front; 2, 4, 8 = that many
 *                      (speed only: results do not depend on the order of workgroups)
 *   key "wide_fuse"  : psf_mlp_wide_fwd_f32: 1 = the second layers of the MLPs with <= 32 outputs run inside the first
 *                      layers' GEMM epilogue when every MLP has 97..128 hidden rows (default), 0 = always the separate kernel
 *   key "mixer_lds"  : psf_mixer_fwd_*: 1 = short sequences take the single-launch LDS-resident mixer (default), 0 = per-step kernels
 *   key "mixer_wg_limit": psf_mixer_fwd_f32's step kernel, workgroups per CU: 0 = as many as fit (default: three), n = at most n
 *   key "mixer_ablate": TIMING EXPERIMENTS ONLY, and only in a library built with -DPSF_MIXER_ABLATE_LAB (ignored otherwise):
 *                      bit 0 no MLP arithmetic, bit 1 no multiply-add chain, bit 2 no far rows, bit 3 no data rows in
 *                      psf_mixer_fwd_f32's step kernel (profiles/mixer_bench.py)
 *   key "bwd_ablate" : TIMING EXPERIMENTS ONLY, and only in a library built with -DPSF_BWD_ABLATE_LAB (ignored otherwise): parts
 *                      of the fused backward step left out (csrc/bwd_fused.h, ABL; profiles/bwd_ablate_bench.py)
 *   key "xcd_remap"  : 1 = keep a batch element's tiles on one XCD group (default), 0 = linear
 * psf_get_tuning returns the value (>= 0) or PSF_E_TUNING.
 */
int psf_set_tuning(const char* key, int32_t value);
int psf_get_tuning(const char* key);

/*
 * Name of the kernel variant the dispatcher would run for this forward shape (for profiles and bench
 * logs; e.g. "chord_fwd_win<f32,L=15,C=8>"). Writes a NUL-terminated string of at most cap-1 chars.
 */
int psf_describe_fwd(int64_t B, int64_t N, int32_t L, int64_t C, int32_t elem_bytes,
                     char* buf, int32_t cap);
/* The same for a whole f32 forward chain of M steps: names the single-launch LDS-resident kernel when the chain takes it. */
int psf_describe_chain_fwd(int64_t B, int64_t N, int32_t L, int64_t C, int32_t M, char* buf, int32_t cap);

#ifdef __cplusplus
}
#endif
#endif /* PSF_CHORD_H */
