/*
 * CPU oracle in plain C for the row-softmax / attention-forward hot path.
 *
 * TEST INFRASTRUCTURE ONLY: linked by tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py as the checker / the timed CPU baseline.  The
 * shipped GPU path never calls into this file.
 *
 * Restates, in scalar C, the algorithm the reference states in Python:
 *   oracle_softmax_*    <- tests/test_flash_attn.py:31-35 (_softmax) and the three
 *                          sweeps of csrc/my_flash_attn/softmax.cu:34-80
 *                          (row max, exp and sum, divide)
 *   oracle_attention_*  <- tests/test_flash_attn.py:37-44 (_attention_forward) =
 *                          csrc/api.cpp:186-189 (S = Q.K^T, P = softmax(S), O = P.V)
 *                          evaluated one query row at a time so S is never stored
 *
 * Two arithmetic flavours of each:
 *   *_f64acc : inputs widened to double, all arithmetic in double, result
 *              rounded to float once.  This is the checker.
 *   *_f32    : float arithmetic throughout, like the reference's CPU tensors.
 *              This is what bench.py times as the CPU baseline.
 *
 * Rows are independent, so the outer loop is an OpenMP parallel-for; the
 * thread count is whatever the caller sets (oracle_set_threads / OMP_NUM_THREADS).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int oracle_abi_version(void) { return 1; }

int oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void oracle_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ---------------------------------------------------------------- softmax */

void oracle_softmax_f64acc(const float* in, float* out, int64_t h, int64_t w) {
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < h; ++r) {
        const float* x = in + r * w;
        float* y = out + r * w;
        double mx = -INFINITY;
        for (int64_t c = 0; c < w; ++c) mx = x[c] > mx ? x[c] : mx;
        double sum = 0.0;
        for (int64_t c = 0; c < w; ++c) sum += exp((double)x[c] - mx);
        for (int64_t c = 0; c < w; ++c) y[c] = (float)(exp((double)x[c] - mx) / sum);
    }
}

void oracle_softmax_f32(const float* in, float* out, int64_t h, int64_t w) {
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < h; ++r) {
        const float* x = in + r * w;
        float* y = out + r * w;
        float mx = -INFINITY;
        for (int64_t c = 0; c < w; ++c) mx = x[c] > mx ? x[c] : mx;
        float sum = 0.0f;
        for (int64_t c = 0; c < w; ++c) { float e = expf(x[c] - mx); y[c] = e; sum += e; }
        for (int64_t c = 0; c < w; ++c) y[c] /= sum;
    }
}

/* -------------------------------------------------------------- attention */
/* One (batch, head): q, k, v, o are (n, d) row-major; k is NOT transposed. */

/* qpos (may be NULL): position of each query row in the sequence; when given, row i sees
 * keys 0 .. qpos[i] only (the causal mask of scaled_dot_product_attention(is_causal=True);
 * the reference itself has no masked variant). */
static void attn_head_f64acc(const float* q, const float* k, const float* v, float* o,
                             int64_t nq, int64_t n_all, int64_t d, double scale, const int64_t* qpos) {
#pragma omp parallel
    {
        double* s = (double*)malloc(sizeof(double) * (size_t)n_all);
        double* acc = (double*)malloc(sizeof(double) * (size_t)d);
#pragma omp for schedule(static)
        for (int64_t i = 0; i < nq; ++i) {
            const float* qi = q + i * d;
            const int64_t n = qpos ? (qpos[i] + 1 < n_all ? qpos[i] + 1 : n_all) : n_all;
            double mx = -INFINITY;
            for (int64_t j = 0; j < n; ++j) {
                const float* kj = k + j * d;
                double dot = 0.0;
                for (int64_t c = 0; c < d; ++c) dot += (double)qi[c] * (double)kj[c];
                dot *= scale;
                s[j] = dot;
                mx = dot > mx ? dot : mx;
            }
            double sum = 0.0;
            for (int64_t j = 0; j < n; ++j) { s[j] = exp(s[j] - mx); sum += s[j]; }
            for (int64_t c = 0; c < d; ++c) acc[c] = 0.0;
            for (int64_t j = 0; j < n; ++j) {
                const float* vj = v + j * d;
                const double p = s[j] / sum;
                for (int64_t c = 0; c < d; ++c) acc[c] += p * (double)vj[c];
            }
            for (int64_t c = 0; c < d; ++c) o[i * d + c] = (float)acc[c];
        }
        free(s);
        free(acc);
    }
}

static void attn_head_f32(const float* q, const float* k, const float* v, float* o,
                          int64_t nq, int64_t n, int64_t d, float scale) {
#pragma omp parallel
    {
        float* s = (float*)malloc(sizeof(float) * (size_t)n);
        float* acc = (float*)malloc(sizeof(float) * (size_t)d);
#pragma omp for schedule(static)
        for (int64_t i = 0; i < nq; ++i) {
            const float* qi = q + i * d;
            float mx = -INFINITY;
            for (int64_t j = 0; j < n; ++j) {
                const float* kj = k + j * d;
                float dot = 0.0f;
#pragma omp simd reduction(+ : dot)
                for (int64_t c = 0; c < d; ++c) dot += qi[c] * kj[c];
                dot *= scale;
                s[j] = dot;
                mx = dot > mx ? dot : mx;
            }
            float sum = 0.0f;
            for (int64_t j = 0; j < n; ++j) { s[j] = expf(s[j] - mx); sum += s[j]; }
            for (int64_t c = 0; c < d; ++c) acc[c] = 0.0f;
            for (int64_t j = 0; j < n; ++j) {
                const float* vj = v + j * d;
                const float p = s[j] / sum;
#pragma omp simd
                for (int64_t c = 0; c < d; ++c) acc[c] += p * vj[c];
            }
            memcpy(o + i * d, acc, sizeof(float) * (size_t)d);
        }
        free(s);
        free(acc);
    }
}

/* heads = B*H problems laid out back to back: q and o are (nq, d) per head, k and
 * v are (n, d) per head.  nq < n lets a test check a sample of query rows against
 * the full key set. */
void oracle_attention_f64acc(const float* q, const float* k, const float* v, float* o,
                             int64_t heads, int64_t nq, int64_t n, int64_t d, float scale) {
    for (int64_t h = 0; h < heads; ++h)
        attn_head_f64acc(q + h * nq * d, k + h * n * d, v + h * n * d, o + h * nq * d,
                         nq, n, d, (double)scale, NULL);
}

/* The same under the causal mask; qpos[nq] are the sequence positions of the query rows
 * (identical for every head). */
void oracle_attention_causal_f64acc(const float* q, const float* k, const float* v, float* o,
                                    const int64_t* qpos, int64_t heads, int64_t nq, int64_t n,
                                    int64_t d, float scale) {
    for (int64_t h = 0; h < heads; ++h)
        attn_head_f64acc(q + h * nq * d, k + h * n * d, v + h * n * d, o + h * nq * d,
                         nq, n, d, (double)scale, qpos);
}

void oracle_attention_f32(const float* q, const float* k, const float* v, float* o,
                          int64_t heads, int64_t nq, int64_t n, int64_t d, float scale) {
    for (int64_t h = 0; h < heads; ++h)
        attn_head_f32(q + h * nq * d, k + h * n * d, v + h * n * d, o + h * nq * d,
                      nq, n, d, scale);
}
