// extern "C" surface of libmyfa_gfx950.so (declared in include/myfa.h).
//
// Replaces the argument checking / dispatch half of the reference's binding
// layer (csrc/api.cpp:110-192); output allocation stays with the caller.
#include "myfa_common.h"
#include "tuning.h"

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>

namespace myfa {

namespace {
thread_local char g_err[512] = "";
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

thread_local hipError_t t_launch_err = hipSuccess;

int check_launch(const char* what) {
    const hipError_t e = t_launch_err;
    t_launch_err = hipSuccess;
    if (e != hipSuccess) return fail(MYFA_E_LAUNCH, "%s: launch failed: %s", what, hipGetErrorString(e));
    return MYFA_OK;
}

#ifdef MYFA_TUNING
Tuning& tuning() {
    static Tuning t{};
    return t;
}
#endif

// implemented in the kernel translation units
int softmax_f32(const float* in, float* out, int64_t h, int64_t w, hipStream_t s);
int softmax_rows(const void* in, void* out, int64_t h, int64_t w, int dtype, hipStream_t s);
int softmax_strided(const void* in, void* out, int64_t outer, int64_t len, int64_t inner, int dtype, hipStream_t s);
int softmax_strided_pro(const void* in, void* out, int64_t outer, int64_t len, int64_t inner, int dtype,
                        float scale, const float* mask, int64_t mask_outer, bool causal, hipStream_t s,
                        void* ws, size_t ws_bytes);
size_t softmax_strided_ws_bytes(int64_t outer, int64_t len, int64_t inner, int dtype);
int matmul_f32(const float* a, const float* b, float* c, int64_t m, int64_t k, int64_t n, hipStream_t s);
int attn_fwd_f32(const float* q, const float* k, const float* v, float* o, float* lse,
                 int64_t heads, int64_t N, int64_t d, int64_t k_key_stride, int64_t k_d_stride,
                 float scale, bool causal, hipStream_t s, float* ws, size_t ws_bytes);
size_t attn_fwd_f32_ws_bytes(int64_t heads, int64_t N, int64_t d);
int attn_fwd_mfma(const void* q, const void* k, const void* v, void* o, float* lse,
                  int64_t heads, int64_t N, int64_t d, float scale, int dtype, bool causal, hipStream_t s);
const char* attn_fwd_mfma_kernel_name(int64_t d, int dtype, bool causal);
int attn_bwd_mfma(const void* q, const void* k, const void* v, const void* o, const float* lse, const void* dout,
                  void* dq, void* dk, void* dv, float* delta, int64_t heads, int64_t N, int64_t d,
                  float scale, int dtype, bool causal, hipStream_t s);
int attn_bwd_mfma_static_ws(const void* q, const void* k, const void* v, const void* o, const float* lse, const void* dout,
                            void* dq, void* dk, void* dv, int64_t heads, int64_t N, int64_t d,
                            float scale, int dtype, bool causal, hipStream_t s);

}  // namespace myfa

using namespace myfa;

extern "C" {

int myfa_abi_version(void) { return MYFA_ABI_VERSION; }

const char* myfa_last_error(void) { return g_err; }

int myfa_device_info(int device, int* cu_count, int* lds_bytes, int* wave_size, char* arch, int arch_len) {
    hipDeviceProp_t p;
    const hipError_t e = hipGetDeviceProperties(&p, device);
    if (e != hipSuccess) return fail(MYFA_E_DEVICE, "device %d: %s", device, hipGetErrorString(e));
    if (cu_count) *cu_count = p.multiProcessorCount;
    if (lds_bytes) *lds_bytes = (int)p.sharedMemPerBlock;
    if (wave_size) *wave_size = p.warpSize;
    if (arch && arch_len > 0) {
        std::strncpy(arch, p.gcnArchName, (size_t)arch_len - 1);
        arch[arch_len - 1] = '\0';
    }
    return MYFA_OK;
}

int myfa_softmax_f32(const float* in, float* out, int64_t h, int64_t w, void* stream) {
    return softmax_f32(in, out, h, w, static_cast<hipStream_t>(stream));
}

int myfa_softmax(const void* in, void* out, int64_t h, int64_t w, int dtype, void* stream) {
    return softmax_rows(in, out, h, w, dtype, static_cast<hipStream_t>(stream));
}

int myfa_softmax_nd(const void* in, void* out, int64_t outer, int64_t len, int64_t inner, int dtype, void* stream) {
    return softmax_strided(in, out, outer, len, inner, dtype, static_cast<hipStream_t>(stream));
}

int myfa_softmax_nd_ex(const void* in, void* out, int64_t outer, int64_t len, int64_t inner, int dtype,
                       float scale, const float* add_mask, int64_t mask_outer, int flags, void* stream) {
    if ((flags & ~MYFA_SOFTMAX_CAUSAL) != 0) return fail(MYFA_E_INVALID, "softmax: unknown flag bits 0x%x", flags);
    return softmax_strided_pro(in, out, outer, len, inner, dtype, scale, add_mask, mask_outer,
                               (flags & MYFA_SOFTMAX_CAUSAL) != 0, static_cast<hipStream_t>(stream), nullptr, 0);
}

int64_t myfa_softmax_nd_ws_bytes(int64_t outer, int64_t len, int64_t inner, int dtype) {
    return (int64_t)softmax_strided_ws_bytes(outer, len, inner, dtype);
}

int myfa_softmax_nd_ws(const void* in, void* out, int64_t outer, int64_t len, int64_t inner, int dtype,
                       float scale, const float* add_mask, int64_t mask_outer, int flags,
                       void* workspace, int64_t workspace_bytes, void* stream) {
    if ((flags & ~MYFA_SOFTMAX_CAUSAL) != 0) return fail(MYFA_E_INVALID, "softmax: unknown flag bits 0x%x", flags);
    if (workspace_bytes < 0 || (workspace == nullptr && workspace_bytes > 0) || reinterpret_cast<uintptr_t>(workspace) % 16 != 0)
        return fail(MYFA_E_INVALID, "softmax: workspace must be 16-byte aligned with a non-negative size");
    return softmax_strided_pro(in, out, outer, len, inner, dtype, scale, add_mask, mask_outer,
                               (flags & MYFA_SOFTMAX_CAUSAL) != 0, static_cast<hipStream_t>(stream),
                               workspace, (size_t)workspace_bytes);
}

int myfa_attn_fwd_f32(const float* q, const float* k_t, const float* v, float* o,
                      int64_t n, int64_t d, float scale, void* stream) {
    if (n < 0 || d < 0) return fail(MYFA_E_INVALID, "attention: negative size n=%lld d=%lld", (long long)n, (long long)d);
    if (n == 0 || d == 0) return MYFA_OK;
    if (!q || !k_t || !v || !o) return fail(MYFA_E_INVALID, "attention: null pointer");
    // K_t is [d, n]: key stride 1, dim stride n
    return attn_fwd_f32(q, k_t, v, o, nullptr, 1, n, d, 1, n, scale, false, static_cast<hipStream_t>(stream), nullptr, 0);
}

int64_t myfa_attn_fwd_f32_ws_bytes(int64_t n, int64_t d) { return (int64_t)attn_fwd_f32_ws_bytes(1, n, d); }

int myfa_attn_fwd_f32_ws(const float* q, const float* k_t, const float* v, float* o,
                         int64_t n, int64_t d, float scale, void* workspace, int64_t workspace_bytes, void* stream) {
    if (n < 0 || d < 0) return fail(MYFA_E_INVALID, "attention: negative size n=%lld d=%lld", (long long)n, (long long)d);
    if (n == 0 || d == 0) return MYFA_OK;
    if (!q || !k_t || !v || !o) return fail(MYFA_E_INVALID, "attention: null pointer");
    if (workspace_bytes < 0 || (workspace == nullptr && workspace_bytes > 0) || reinterpret_cast<uintptr_t>(workspace) % 16 != 0)
        return fail(MYFA_E_INVALID, "attention: workspace must be 16-byte aligned with a non-negative size");
    return attn_fwd_f32(q, k_t, v, o, nullptr, 1, n, d, 1, n, scale, false, static_cast<hipStream_t>(stream),
                        static_cast<float*>(workspace), (size_t)workspace_bytes);
}

int myfa_attn_fwd_bhnd(const void* q, const void* k, const void* v, void* o, float* lse,
                       int64_t B, int64_t H, int64_t N, int64_t d, float scale, int dtype, void* stream) {
    return myfa_attn_fwd_bhnd_ex(q, k, v, o, lse, B, H, N, d, scale, dtype, 0, stream);
}

int myfa_attn_fwd_bhnd_ex(const void* q, const void* k, const void* v, void* o, float* lse,
                          int64_t B, int64_t H, int64_t N, int64_t d, float scale, int dtype, int flags, void* stream) {
    if ((flags & ~MYFA_ATTN_CAUSAL) != 0) return fail(MYFA_E_INVALID, "attention: unknown flag bits 0x%x", flags);
    const bool causal = (flags & MYFA_ATTN_CAUSAL) != 0;
    if (B < 0 || H < 0 || N < 0 || d < 0)
        return fail(MYFA_E_INVALID, "attention: negative size B=%lld H=%lld N=%lld d=%lld",
                    (long long)B, (long long)H, (long long)N, (long long)d);
    if (B == 0 || H == 0 || N == 0 || d == 0) return MYFA_OK;
    if (!q || !k || !v || !o) return fail(MYFA_E_INVALID, "attention: null pointer");
    const int64_t heads = B * H;
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (dtype) {
        case MYFA_F32:
            // K is [N, d]: key stride d, dim stride 1
            return attn_fwd_f32(static_cast<const float*>(q), static_cast<const float*>(k),
                                static_cast<const float*>(v), static_cast<float*>(o), lse,
                                heads, N, d, d, 1, scale, causal, s, nullptr, 0);
        case MYFA_F16:
        case MYFA_BF16:
            if ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k) |
                 reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(o)) % 16 != 0)
                return fail(MYFA_E_INVALID, "attention (f16/bf16): tensors must be 16-byte aligned");
            return attn_fwd_mfma(q, k, v, o, lse, heads, N, d, scale, dtype, causal, s);
        default:
            return fail(MYFA_E_INVALID, "attention: unknown dtype tag %d", dtype);
    }
}

const char* myfa_attn_fwd_bhnd_kernel(int64_t d, int dtype, int flags) {
    if ((flags & ~MYFA_ATTN_CAUSAL) != 0 || d <= 0) return nullptr;
    if (dtype == MYFA_F32) return d <= 512 ? "attn_fwd_f32_kernel" : nullptr;
    if (dtype != MYFA_F16 && dtype != MYFA_BF16) return nullptr;
    return attn_fwd_mfma_kernel_name(d, dtype, (flags & MYFA_ATTN_CAUSAL) != 0);
}

int myfa_attn_bwd_bhnd(const void* q, const void* k, const void* v, const void* o, const float* lse,
                       const void* dout, void* dq, void* dk, void* dv, float* delta_ws,
                       int64_t B, int64_t H, int64_t N, int64_t d, float scale, int dtype, int flags, void* stream) {
    if (B < 0 || H < 0 || N < 0 || d < 0)
        return fail(MYFA_E_INVALID, "attention backward: negative size B=%lld H=%lld N=%lld d=%lld",
                    (long long)B, (long long)H, (long long)N, (long long)d);
    if ((flags & ~MYFA_ATTN_CAUSAL) != 0) return fail(MYFA_E_INVALID, "attention backward: unknown flag bits 0x%x", flags);
    if (B == 0 || H == 0 || N == 0 || d == 0) return MYFA_OK;
    if (!q || !k || !v || !o || !lse || !dout || !dq || !dk || !dv || !delta_ws)
        return fail(MYFA_E_INVALID, "attention backward: null pointer");
    if (dtype != MYFA_F16 && dtype != MYFA_BF16 && dtype != MYFA_F32)
        return fail(MYFA_E_UNSUPPORTED, "attention backward: float16, bfloat16 or float32 (dtype tag %d)", dtype);
    if ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k) | reinterpret_cast<uintptr_t>(v) |
         reinterpret_cast<uintptr_t>(o) | reinterpret_cast<uintptr_t>(dout) | reinterpret_cast<uintptr_t>(dq) |
         reinterpret_cast<uintptr_t>(dk) | reinterpret_cast<uintptr_t>(dv)) % 16 != 0)
        return fail(MYFA_E_INVALID, "attention backward: tensors must be 16-byte aligned");
    return attn_bwd_mfma(q, k, v, o, lse, dout, dq, dk, dv, delta_ws, B * H, N, d, scale, dtype,
                         (flags & MYFA_ATTN_CAUSAL) != 0, static_cast<hipStream_t>(stream));
}

int myfa_attn_bwd_bhnd_static_ws(const void* q, const void* k, const void* v, const void* o, const float* lse,
                                 const void* dout, void* dq, void* dk, void* dv,
                                 int64_t B, int64_t H, int64_t N, int64_t d, float scale, int dtype, int flags, void* stream) {
    if (B < 0 || H < 0 || N < 0 || d < 0)
        return fail(MYFA_E_INVALID, "attention backward: negative size B=%lld H=%lld N=%lld d=%lld",
                    (long long)B, (long long)H, (long long)N, (long long)d);
    if ((flags & ~MYFA_ATTN_CAUSAL) != 0) return fail(MYFA_E_INVALID, "attention backward: unknown flag bits 0x%x", flags);
    if (B == 0 || H == 0 || N == 0 || d == 0) return MYFA_OK;
    if (!q || !k || !v || !o || !lse || !dout || !dq || !dk || !dv)
        return fail(MYFA_E_INVALID, "attention backward: null pointer");
    if (dtype != MYFA_F16 && dtype != MYFA_BF16 && dtype != MYFA_F32)
        return fail(MYFA_E_UNSUPPORTED, "attention backward: float16, bfloat16 or float32 (dtype tag %d)", dtype);
    if ((reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(k) | reinterpret_cast<uintptr_t>(v) |
         reinterpret_cast<uintptr_t>(o) | reinterpret_cast<uintptr_t>(dout) | reinterpret_cast<uintptr_t>(dq) |
         reinterpret_cast<uintptr_t>(dk) | reinterpret_cast<uintptr_t>(dv)) % 16 != 0)
        return fail(MYFA_E_INVALID, "attention backward: tensors must be 16-byte aligned");
    if (B > MYFA_BWD_STATIC_WS_ROWS || H > MYFA_BWD_STATIC_WS_ROWS || B * H > MYFA_BWD_STATIC_WS_ROWS)
        return fail(MYFA_E_UNSUPPORTED, "attention backward (static workspace): B*H*N exceeds %d rows", MYFA_BWD_STATIC_WS_ROWS);
    return attn_bwd_mfma_static_ws(q, k, v, o, lse, dout, dq, dk, dv, B * H, N, d, scale, dtype,
                                   (flags & MYFA_ATTN_CAUSAL) != 0, static_cast<hipStream_t>(stream));
}

int myfa_matmul_f32(const float* a, const float* b, float* c, int64_t m, int64_t k, int64_t n, void* stream) {
    return matmul_f32(a, b, c, m, k, n, static_cast<hipStream_t>(stream));
}

#ifdef MYFA_TUNING
// Exported by the tuning flavour only (not part of include/myfa.h).  Keys: attn_variant,
// attn_thr, attn_spec, attn_wgs, matmul_block, matmul_ksplit, softmax_tpr, softmax_v, softmax_map, softmax_nt, softmax_pad,
// softmax_split_wgs, reset (back to the defaults).  Returns 0, or 1 for an unknown key.
int myfa_tuning_set(const char* key, double value) {
    Tuning& t = tuning();
    const std::string k(key ? key : "");
    if (k == "reset") t = Tuning{};
    else if (k == "attn_variant") t.attn_variant = (int)value;
    else if (k == "attn_thr") t.attn_thr = (float)value;
    else if (k == "attn_spec") t.attn_spec = (int)value;
    else if (k == "attn_xflags") t.attn_xflags = (int)value;
    else if (k == "attn_wgs") t.attn_wgs = (int)value;
    else if (k == "attn_pre") t.attn_pre = (int)value;
    else if (k == "attn_occ") t.attn_occ = (int)value;
    else if (k == "attn_f32_wgs") t.attn_f32_wgs = (int)value;
    else if (k == "matmul_block") t.matmul_block = (int)value;
    else if (k == "matmul_ksplit") t.matmul_ksplit = (int)value;
    else if (k == "softmax_tpr") t.softmax_shape[0] = (int)value;
    else if (k == "softmax_v") t.softmax_shape[1] = (int)value;
    else if (k == "softmax_map") t.softmax_shape[2] = (int)value;
    else if (k == "softmax_nt") t.softmax_shape[3] = (int)value;
    else if (k == "softmax_pad") t.softmax_shape[4] = (int)value;
    else if (k == "softmax_split_wgs") t.softmax_split_wgs = (int)value;
    else if (k == "softmax_cols_big") t.softmax_cols_big = (int)value;
    else return fail(MYFA_E_INVALID, "tuning: unknown key '%s'", k.c_str());
    return MYFA_OK;
}
// attn_dbg: device buffer of 24 u64 per workgroup for the stamped attention build (NULL: off)
int myfa_tuning_set_ptr(const char* key, void* p) {
    const std::string k(key ? key : "");
    if (k == "attn_dbg") tuning().attn_dbg = static_cast<unsigned long long*>(p);
    else return fail(MYFA_E_INVALID, "tuning: unknown pointer key '%s'", k.c_str());
    return MYFA_OK;
}
#endif

}  // extern "C"
