/*
 * myfa.h -- C ABI of libmyfa_gfx950.so: row-softmax and attention-forward for
 * AMD Instinct MI355X (gfx950 / CDNA4).
 *
 * This is the drop-in boundary.  It replaces the C++/pybind layer of the
 * reference (csrc/api.cpp) for the softmax/attention path: each entry point
 * below names the reference interface it stands in for.  Signatures are plain
 * C: device pointers, sizes, a scale and a HIP stream handle (hipStream_t passed
 * as void*; NULL = the default stream).  No torch types.
 *
 * Conventions
 *   - Every function returns 0 on success and a nonzero MYFA_E_* code on
 *     failure; myfa_last_error() then returns a thread-local, human-readable
 *     message.  The library never calls exit() (the reference's cuBLAS/CUDA
 *     helpers do: csrc/my_flash_attn/cublas.hpp:6-12, util.hpp:10-20).
 *   - All pointers are device pointers on the current HIP device; tensors are
 *     dense row-major ("contiguous", the reference's CHECK_CONTIGUOUS,
 *     csrc/my_flash_attn/util.hpp:8).  The library owns nothing: the caller
 *     allocates inputs and outputs (the reference allocates outputs with
 *     torch::zeros, csrc/api.cpp:46,74,114; here the Python host layer does).
 *   - Launches are asynchronous on `stream`; nothing synchronises.  (The
 *     reference launches on the legacy default stream,
 *     csrc/my_flash_attn/softmax.cu:463-475.)
 *   - No entry point allocates or frees device memory.
 */
#ifndef MYFA_H_
#define MYFA_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: the entry points with a caller workspace or the library's own (…_ws, …_static_ws), the fused
 * softmax prologue (…_nd_ex) and 16-bit head dims up to 256 were added; nothing of version 1 changed.
 * 3: myfa_attn_fwd_bhnd_kernel added; nothing of version 2 changed. */
#define MYFA_ABI_VERSION 3

enum {
    MYFA_OK = 0,
    MYFA_E_INVALID = 1,     /* bad argument (null pointer, negative size, ...) */
    MYFA_E_UNSUPPORTED = 2, /* shape / dtype outside what the kernels cover    */
    MYFA_E_LAUNCH = 3,      /* HIP reported an error at launch                 */
    MYFA_E_DEVICE = 4       /* no usable gfx950 device                         */
};

/* dtype tags for the batched attention entry point */
enum { MYFA_F32 = 0, MYFA_F16 = 1, MYFA_BF16 = 2 };

/* Library / ABI identification (no reference counterpart; used by the host
 * layer to check that header and .so agree). */
int myfa_abi_version(void);

/* Thread-local message of the last failing call on this thread ("" if none). */
const char* myfa_last_error(void);

/* Fills in facts about HIP device `device`: CU count, LDS bytes per workgroup,
 * wavefront size, and the gcnArchName (NUL-terminated, truncated to arch_len).
 * Any out-pointer may be NULL. */
int myfa_device_info(int device, int* cu_count, int* lds_bytes, int* wave_size,
                     char* arch, int arch_len);

/*
 * Row softmax of a dense [h, w] float matrix: out[r, :] = softmax(in[r, :]).
 *
 * Replaces my_softmax(inp, kernel_no) -> launch_softmax_kernel<128> ->
 * softmax_kernel_1..5 (csrc/api.cpp:110-166, csrc/my_flash_attn/softmax.cu:20-487).
 * All five kernel numbers of the reference compute the same function; there is
 * one implementation here, so kernel_no is handled (validated) by the host
 * layer and does not cross this boundary.  `in` and `out` may not overlap
 * unless identical (in-place is allowed).  h == 0 or w == 0 is a no-op.
 */
int myfa_softmax_f32(const float* in, float* out, int64_t h, int64_t w, void* stream);

/*
 * The same for float32, float16 or bfloat16 tensors (dtype: MYFA_F32 / MYFA_F16 /
 * MYFA_BF16, the element type of both `in` and `out`); max, exponential and sum are
 * float32 whatever the element type, the result is rounded once on the way out.
 * The reference's my_softmax only takes float32 (csrc/api.cpp:116,118:
 * data_ptr<float>()); 16-bit rows halve the bytes per element of this
 * bandwidth-bound kernel.
 */
int myfa_softmax(const void* in, void* out, int64_t h, int64_t w, int dtype, void* stream);

/*
 * Softmax along any one dimension of a dense tensor, viewed as [outer, len, inner]:
 * out[o, :, i] = softmax(in[o, :, i]) -- torch.softmax(x, dim) with outer = product of the
 * sizes before dim, len = size of dim, inner = product of the sizes after it.  inner == 1
 * is myfa_softmax on an [outer, len] matrix.  Same element types and float32 arithmetic as
 * myfa_softmax; in-place is allowed; any size 0 is a no-op.
 * The reference has rows of a 2-D matrix only (csrc/api.cpp:110-166); this generalises
 * that entry point to the other dimensions.
 */
int myfa_softmax_nd(const void* in, void* out, int64_t outer, int64_t len, int64_t inner,
                    int dtype, void* stream);

/*
 * myfa_softmax_nd with a fused prologue: out[o, :, i] = softmax(scale * in[o, :, i] + mask[o % mask_outer, :, i]),
 * and with MYFA_SOFTMAX_CAUSAL element j of row o hidden (-inf) when j > o % len.
 *
 * For callers that keep the reference's composition of attention, matmul -> softmax -> matmul
 * (csrc/api.cpp:186-187: QK = my_matmul_cublas(Q, K); W = my_softmax(QK)): the logit scale, an
 * additive mask (padding, ALiBi, ...) and the causal mask cost no extra pass over the N x N logits.
 *   scale       multiplies every element first (1.0f: none)
 *   add_mask    float32, dense [mask_outer, len, inner], or NULL; outer must be a multiple of
 *               mask_outer (mask_outer = 1 broadcasts one [len, inner] mask over everything; for
 *               [B, H, N, N] logits softmaxed along the last dimension an [N, N] mask is mask_outer = N);
 *               -inf entries are allowed, a slice that is -inf throughout gives NaN as torch.softmax does
 *   flags       MYFA_SOFTMAX_CAUSAL needs inner == 1 and outer a multiple of len (the last dimension of
 *               [..., len, len] logits); otherwise MYFA_E_UNSUPPORTED.  Unknown bits are an error.
 * scale = 1, add_mask = NULL, flags = 0 computes what myfa_softmax_nd computes.
 */
#define MYFA_SOFTMAX_CAUSAL 1
int myfa_softmax_nd_ex(const void* in, void* out, int64_t outer, int64_t len, int64_t inner, int dtype,
                       float scale, const float* add_mask, int64_t mask_outer, int flags, void* stream);

/*
 * myfa_softmax_nd_ex with a caller workspace, for shapes that leave the chip empty when every slice
 * along the softmax dimension is one workgroup's job:
 *   long columns (inner > 1, more rows than one workgroup's registers hold) whose column blocks are
 *   few, e.g. [100000, 512] along dimension 0: the rows of a column block are split over workgroups;
 *   few long rows (inner == 1, outer < 256, len beyond 65536 elements), e.g.
 *   [4, 4194304] or [16, 131072] logits: each row is split over workgroups.
 * Two launches: the slices leave their (max, sum) pairs in the workspace and the workgroup that
 * finishes a column block / row last merges them (nobody waits for anybody); the second launch
 * re-reads and writes.  myfa_softmax_nd_ws_bytes(outer, len, inner, dtype) is the size that lets the
 * library do so (0: the shape is not split; a smaller workspace, NULL, or tensors that are not
 * 16-byte aligned fall back to what myfa_softmax_nd_ex does).  The library clears the part of the
 * workspace it needs cleared on `stream` itself; the contents before and after the call mean
 * nothing; it must stay untouched until the call's work on `stream` has finished, and two calls
 * that may overlap need a workspace each.  Same result as myfa_softmax_nd_ex up to float32 rounding
 * of the merge.  In-place is allowed.
 * No counterpart in the reference (its softmax is one workgroup per row, csrc/my_flash_attn/softmax.cu:459-487).
 */
int64_t myfa_softmax_nd_ws_bytes(int64_t outer, int64_t len, int64_t inner, int dtype);
int myfa_softmax_nd_ws(const void* in, void* out, int64_t outer, int64_t len, int64_t inner, int dtype,
                       float scale, const float* add_mask, int64_t mask_outer, int flags,
                       void* workspace, int64_t workspace_bytes, void* stream);

/*
 * 2-D single-head attention, float32, the reference's calling convention:
 *     o = softmax(scale * (q @ k_t)) @ v
 * q: [n, d]; k_t: [d, n] (K already transposed, as the reference's caller
 * passes it, tests/test_flash_attn.py:163); v: [n, d]; o: [n, d].
 *
 * Replaces my_attention(Q, K, V) = my_matmul_cublas -> my_softmax ->
 * my_matmul_cublas (csrc/api.cpp:168-192) with one fused kernel.  The
 * reference has no scale argument (logits are unscaled); pass scale = 1.0f for
 * its semantics.
 */
int myfa_attn_fwd_f32(const float* q, const float* k_t, const float* v, float* o,
                      int64_t n, int64_t d, float scale, void* stream);

/*
 * myfa_attn_fwd_f32 with a caller workspace, for the single-head shapes of the reference's tests
 * (tests/test_flash_attn.py:155: N up to 1025): one head has N/16 query groups, far fewer than the
 * chip has matrix pipes, so with a workspace the keys are also split over workgroups and the
 * partial results merged by a second small kernel.  myfa_attn_fwd_f32_ws_bytes(n, d) is the size
 * that lets the library do so (0: it would not split; any smaller workspace, or NULL, falls back
 * to what myfa_attn_fwd_f32 does).  The workspace is scratch: its contents before and after the
 * call mean nothing; it must stay untouched until the call's work on `stream` has finished.
 * Same result as myfa_attn_fwd_f32 up to float32 rounding of the merge.
 */
int64_t myfa_attn_fwd_f32_ws_bytes(int64_t n, int64_t d);
int myfa_attn_fwd_f32_ws(const float* q, const float* k_t, const float* v, float* o,
                         int64_t n, int64_t d, float scale,
                         void* workspace, int64_t workspace_bytes, void* stream);

/*
 * Batched multi-head attention forward, no mask, no dropout:
 *     o[b,h] = softmax(scale * q[b,h] @ k[b,h]^T) @ v[b,h]
 * q, k, v, o: dense [B, H, N, d] of `dtype` (MYFA_F32 / MYFA_F16 / MYFA_BF16);
 * k is NOT transposed.  Softmax statistics and both accumulations are float32.
 * If `lse` is non-NULL it receives the row log-sum-exp (natural log) of the
 * scaled logits as float32 [B, H, N].
 *
 * This is the fused forward the reference sketches but never wrote
 * (csrc/my_flash_attn/flash_attention.cu:23-69: kernel signature
 * (out, Q, K, V, seq_len, d_head), host wrapper allocating out, l, m); its
 * math is the reference's Python Algorithm-1 restatement
 * (tests/test_flash_attn.py:46-101) applied per (batch, head).
 * MYFA_F16 / MYFA_BF16 run on MFMA and require d a multiple of 8 in 8..256 (64 and 128 are the fast
 * builds; the others run on the next build up -- 64, 128 or 256 columns -- with the missing columns
 * read as zeros), 16-byte aligned tensors and scale > 0; MYFA_F32 accepts
 * 1 <= d <= 512 and N*d < 2^31 (myfa_attn_fwd_f32 has the same limits;
 * the reference's my_attention has none, but its test shapes end at N = 1025, d = 99).  Outside
 * these limits the call returns MYFA_E_UNSUPPORTED.
 * `o` with and without `lse` agree to the rounding of the element type, not bit for bit: without
 * the row statistic, 16-bit launches with d > 64 and scale * log2(e) * sqrt(d) <= 4 multiply q by
 * scale * log2(e) once, rounded to the element type, instead of scaling every logit in float32
 * (callers that feed a backward ask for `lse` and get the statistic of the unrounded logits).
 * The 16-bit launches with 64 < d <= 128 run as many workgroups as the device holds at once, each
 * working through a share of the (batch, head, query block) problems: the launcher reads the
 * device's compute-unit count (once per device).
 */
int myfa_attn_fwd_bhnd(const void* q, const void* k, const void* v, void* o, float* lse,
                       int64_t B, int64_t H, int64_t N, int64_t d,
                       float scale, int dtype, void* stream);

/*
 * The same with option bits.  MYFA_ATTN_CAUSAL: key j contributes to query i only
 * if j <= i (the mask that autoregressive callers of a fused forward need; the
 * reference has no masked variant -- its Python statement
 * tests/test_flash_attn.py:46-101 and my_attention csrc/api.cpp:168-192 are both
 * unmasked -- so this follows torch's scaled_dot_product_attention(is_causal=True),
 * which is also what the oracle for it is pinned against).  flags = 0 is exactly
 * myfa_attn_fwd_bhnd.  Unknown bits are an error.
 */
#define MYFA_ATTN_CAUSAL 1
int myfa_attn_fwd_bhnd_ex(const void* q, const void* k, const void* v, void* o, float* lse,
                          int64_t B, int64_t H, int64_t N, int64_t d,
                          float scale, int dtype, int flags, void* stream);

/*
 * Name of the device kernel that myfa_attn_fwd_bhnd_ex launches for head dim `d`, `dtype` and
 * `flags` (the function name a kernel trace of the call shows, without template arguments), or
 * NULL where the call would return an error.  For harnesses that label a measurement with the
 * kernel it timed; launches nothing.  No counterpart in the reference (its launcher's switch,
 * csrc/my_flash_attn/softmax.cu:459-487, is not queryable).
 */
const char* myfa_attn_fwd_bhnd_kernel(int64_t d, int dtype, int flags);

/*
 * Backward of myfa_attn_fwd_bhnd_ex for MYFA_F16 / MYFA_BF16, d in {64, 128, 192, 256}
 * (192 and 256: four-wave workgroups on 32-row tiles, dK and dV in separate launches), and for
 * MYFA_F32 with any d up to 512, the float32 forward's range (float32 matrix instructions, float32
 * throughout; above 256 columns the kernels take the output columns in groups and walk the other
 * side once per group;
 * N*d < 2^31 as in the float32 forward):
 * given the forward's inputs, its output `o`, its row log-sum-exp `lse` and the
 * gradient `dout` of the output, writes dq, dk, dv (same shape and dtype as q, k, v).
 * `delta_ws` is a float32 workspace of 2*B*H*N elements provided by the caller (it
 * receives rowsum(dout o o) and lse in base 2); the library allocates nothing.
 * flags as in the forward.
 * No counterpart in the reference, which has neither a fused forward nor a
 * backward (csrc/my_flash_attn/flash_attention.cu:23-69 is a stub); the math is the
 * derivative of the forward it sketches, checked against autograd of PyTorch's
 * scaled_dot_product_attention.
 */
int myfa_attn_bwd_bhnd(const void* q, const void* k, const void* v, const void* o, const float* lse,
                       const void* dout, void* dq, void* dk, void* dv, float* delta_ws,
                       int64_t B, int64_t H, int64_t N, int64_t d,
                       float scale, int dtype, int flags, void* stream);

/*
 * myfa_attn_bwd_bhnd without the workspace argument, for callers that cannot allocate device
 * memory (code under HIP graph capture, tight serving loops, harnesses that forbid it): the
 * 2*B*H*N float workspace lives in a device array inside the library (8 MiB, held by every GPU
 * the library runs on until the process ends), so the call allocates nothing and takes only
 * inputs and outputs.  No counterpart in the reference.
 * Limit: B*H*N <= MYFA_BWD_STATIC_WS_ROWS, otherwise MYFA_E_UNSUPPORTED (use myfa_attn_bwd_bhnd).
 * All calls on one GPU share that array: calls that overlap on one GPU (different streams)
 * race; calls on one stream, or ordered by events, are fine.  Results are those of
 * myfa_attn_bwd_bhnd bit for bit.  The array's device address is looked up on the first call
 * on each device (make that call outside any graph capture).
 */
#define MYFA_BWD_STATIC_WS_ROWS 1048576
int myfa_attn_bwd_bhnd_static_ws(const void* q, const void* k, const void* v, const void* o, const float* lse,
                                 const void* dout, void* dq, void* dk, void* dv,
                                 int64_t B, int64_t H, int64_t N, int64_t d,
                                 float scale, int dtype, int flags, void* stream);

/*
 * c[m, n] = a[m, k] . b[k, n], float32, dense row-major.  Replaces the reference's
 * own tiled kernel behind my_matmul / my_matmul_out (launch_matmul_kernel,
 * csrc/my_flash_attn/matmul.cu:22-62, called from csrc/api.cpp:13-66).  Every
 * element of c is written (the reference zero-fills first, api.cpp:46).
 * c must be device memory and may not overlap a or b: small outputs with a long k are computed
 * by two workgroups per block that add their halves into a c the launcher has zeroed first, so
 * an input that shares storage with c would be cleared before it is read (the reference's own
 * kernel reads A and B while it writes `out`, so aliasing is not defined there either,
 * csrc/api.cpp:13-38).  Overlapping ranges are refused with MYFA_E_INVALID.
 */
int myfa_matmul_f32(const float* a, const float* b, float* c, int64_t m, int64_t k, int64_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MYFA_H_ */
