// Which kernel runs a 16-bit attention forward (behind myfa_attn_fwd_bhnd[_ex]):
//   head dims 136..256            attn_fwd_mfma_wide_kernel   (attn_fwd_mfma256.hip)
//   head dims 72..128             attn_fwd_stream_kernel      (attn_fwd_stream.hip)
//   head dims 8..64, causal       attn_fwd_stream_kernel, one workgroup per CU
//   head dims 8..64, no mask      attn_fwd_mfma16_kernel      (attn_fwd_mfma16.hip)
// The tuning flavour of the library can force any build, including the two 32x32x16 ones of
// attn_fwd_mfma.hip, which only that flavour contains.
#include <math.h>

#include "myfa_common.h"
#include "tuning.h"

namespace myfa {

#ifdef MYFA_TUNING
int attn_fwd_mfma32(const void* q, const void* k, const void* v, void* o, float* lse,
                    int64_t heads, int64_t N, int64_t d, float scale, int dtype, bool causal, hipStream_t s);
#endif
int attn_fwd_mfma16(const void* q, const void* k, const void* v, void* o, float* lse,
                    int64_t heads, int64_t N, int64_t d, float scale, float thr, int spec, int pre, int xflags,
                    int dtype, bool causal, hipStream_t s);
int attn_fwd_stream(const void* q, const void* k, const void* v, void* o, float* lse,
                    int64_t heads, int64_t N, int64_t d, float scale, float thr, int spec, int pre, int wgs, int occ,
                    int dtype, bool causal, hipStream_t s);
int attn_fwd_mfma_wide(const void* q, const void* k, const void* v, void* o, float* lse,
                       int64_t heads, int64_t N, int64_t d, float scale, float thr, int dtype, bool causal, hipStream_t s);

// The kernel the dispatch below picks (kept beside it so that the two cannot drift apart).
const char* attn_fwd_mfma_kernel_name(int64_t d, [[maybe_unused]] int dtype, [[maybe_unused]] bool causal) {
    if (d < 8 || d > 256 || d % 8 != 0) return nullptr;
    if (d > 128) return "attn_fwd_mfma_wide_kernel";
    const int variant = tuning().attn_variant;
    if (variant == 3 || (variant < 0 && (d > 64 || causal))) return "attn_fwd_stream_kernel";
    if (variant == 2 || variant < 0) return "attn_fwd_mfma16_kernel";
    return variant == 0 && !causal && (d == 64 || d == 128) ? "attn_fwd_mfma_kernel" : "attn_fwd_mfma_pp_kernel";
}

// dtype: MYFA_F16 or MYFA_BF16; d: a multiple of 8 in 8..256; heads = B*H
int attn_fwd_mfma(const void* q, const void* k, const void* v, void* o, float* lse,
                  int64_t heads, int64_t N, int64_t d, float scale, int dtype, bool causal, hipStream_t s) {
    if (!(scale > 0.f)) return fail(MYFA_E_UNSUPPORTED, "attention (f16/bf16): scale must be > 0, got %g", (double)scale);
    if (N > 0x7fffffffLL / 256) return fail(MYFA_E_UNSUPPORTED, "attention: N=%lld too large", (long long)N);
    // The 16x16x32 build is the default: the chip holds a higher clock on that MFMA shape (random
    // operands; measured +6..12 % at d = 128, +2 % at d = 64 without a mask).  Under the causal mask
    // its tile loop is split in two, unmasked tiles in front of the workgroup's rows and the masked
    // tiles on the diagonal, so the first loop is the unmasked kernel's code: even with the 32x32x16
    // build at d = 128 (N = 4096; +1 % at N = 16384) and +32..38 % at d = 64.
    const float thr_classic = tuning().attn_thr < 0.f ? 8.f : tuning().attn_thr;
    if (d > 128) return attn_fwd_mfma_wide(q, k, v, o, lse, heads, N, d, scale, thr_classic, dtype, causal, s);
    const int variant = tuning().attn_variant;
    // Head dims above 64: the resident-workgroup build (one workgroup per CU working through a list
    // of query blocks as one unbroken stream of tiles: +4..5 % on random operands, +9.5 % in cycles
    // over one workgroup per query block, whose fill and drain leave the CU's matrix cores idle for
    // ~6 % of its life).  Up to 64 it still loses (register spills at 128 registers per lane).
    // ... except under the causal mask, where its one-workgroup-per-CU flavour wins by 8-14 %.
    if (variant == 3 || (variant < 0 && (d > 64 || causal)))
        return attn_fwd_stream(q, k, v, o, lse, heads, N, d, scale, tuning().attn_thr, tuning().attn_spec,
                               tuning().attn_pre, tuning().attn_wgs, tuning().attn_occ, dtype, causal, s);
    if (variant == 2 || variant < 0)
        return attn_fwd_mfma16(q, k, v, o, lse, heads, N, d, scale, tuning().attn_thr, tuning().attn_spec,
                               tuning().attn_pre, tuning().attn_xflags, dtype, causal, s);
#ifdef MYFA_TUNING
    return attn_fwd_mfma32(q, k, v, o, lse, heads, N, d, scale, dtype, causal, s);
#else
    return fail(MYFA_E_UNSUPPORTED, "attention (f16/bf16): no such kernel variant (%d)", variant);
#endif
}

}  // namespace myfa
