/* CPU ORACLE in plain C (test infrastructure, NOT product code).
 *
 * Scalar restatement of the reference's per-tensor calibration arithmetic, bit-compatible with what the
 * reference computes through numpy 2.2.x on float32 data.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; dipoorlet_amd/ never does.
 *
 *   dplo_minmax        forward_net.py:222-235   x.max(), x.min()            (NaN propagates)
 *   dplo_abs_hist      forward_net.py:268       np.histogram(|x|, bins, (0, dmax)) uniform-bin fast path
 *   dplo_octav_scale   forward_net.py:323-330   OCTAV fixed point, float32 with numpy's pairwise summation
 *   dplo_batch         the three of them over a list of tensors, OpenMP-parallel over tensors (cpu baseline)
 *
 * Pinned by tests/test_oracle_golden.py against the golden vectors the reference itself produced.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* numpy/_core/src/umath/loops_utils.h.src: pairwise sum, PW_BLOCKSIZE 128, 8 accumulators */
static float pairwise_sum_f32(const float* a, int64_t n) {
    if (n < 8) {
        float res = 0.f;
        for (int64_t i = 0; i < n; i++) res += a[i];
        return res;
    } else if (n <= 128) {
        float r[8];
        int64_t i;
        for (i = 0; i < 8; i++) r[i] = a[i];
        for (i = 8; i < n - (n % 8); i += 8) {
            r[0] += a[i + 0]; r[1] += a[i + 1]; r[2] += a[i + 2]; r[3] += a[i + 3];
            r[4] += a[i + 4]; r[5] += a[i + 5]; r[6] += a[i + 6]; r[7] += a[i + 7];
        }
        float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += a[i];
        return res;
    } else {
        int64_t n2 = n / 2;
        n2 -= n2 % 8;
        return pairwise_sum_f32(a, n2) + pairwise_sum_f32(a + n2, n - n2);
    }
}

void dplo_minmax(const float* x, int64_t n, float* out_min, float* out_max) {
    float mn = INFINITY, mx = -INFINITY;
    int nan = 0;
    for (int64_t i = 0; i < n; i++) {
        const float v = x[i];
        nan |= (v != v);
        mn = v < mn ? v : mn;
        mx = v > mx ? v : mx;
    }
    *out_min = nan ? NAN : mn;
    *out_max = nan ? NAN : mx;
}

static inline float edge_f32(int64_t i, float step, float first) {
    const float p = (float)i * step; /* two roundings (built with -ffp-contract=off), like numpy's y = i*step; y += start */
    return p + first;
}

/* returns 0 ok, 1 range not finite, 2 too many bins */
int dplo_abs_hist(const float* x, int64_t n, int bins, float dmax, int64_t* hist) {
    float first = 0.f, last = dmax;
    if (!isfinite(last) || last < first) return 1;
    if (first == last) { first = -0.5f; last = 0.5f; }
    const float delta = last - first;
    const float step = delta / (float)bins;
    if (!(step > 0.f)) return 2;
    for (int b = 1; b <= bins; b++) {
        const float e0 = edge_f32(b - 1, step, first), e1 = b == bins ? last : edge_f32(b, step, first);
        if (!(e1 > e0)) return 2;
    }
    memset(hist, 0, sizeof(int64_t) * (size_t)bins);
    for (int64_t k = 0; k < n; k++) {
        const float a = fabsf(x[k]);
        if (!(a >= first && a <= last)) continue; /* also drops NaN */
        const float q = (a - first) / delta;
        int64_t i = (int64_t)(q * (float)bins);
        if (i == bins) i -= 1;
        if (a < edge_f32(i, step, first)) i -= 1;
        if (i != bins - 1 && a >= edge_f32(i + 1, step, first)) i += 1;
        hist[i] += 1;
    }
    return 0;
}

float dplo_octav_scale(const float* x, int64_t n, int unsigned_div, float* scratch) {
    /* scratch: 2*n floats (|x| and the boolean-indexed copy numpy materialises) */
    float* a = scratch;
    float* sel = scratch + n;
    int64_t nz = 0;
    for (int64_t i = 0; i < n; i++) {
        a[i] = fabsf(x[i]);
        nz += a[i] > 0.f;
    }
    float s = pairwise_sum_f32(a, n) / (float)nz;
    const double c = 1.0 / 65536.0 / 3.0 / (double)unsigned_div;
    for (int it = 0; it < 20; it++) {
        int64_t gt = 0, le = 0;
        for (int64_t i = 0; i < n; i++) {
            if (a[i] > s) sel[gt++] = a[i];
            le += a[i] <= s;
        }
        const double denom = c * (double)le + (double)gt;
        const float s1 = pairwise_sum_f32(sel, gt) / (float)denom;
        if (fabsf(s1 - s) < 1e-6f) break;
        s = s1;
    }
    return s;
}

/* algo: 0 minmax, 1 hist (minmax + histogram with the tensor's own range), 2 mse (minmax + OCTAV).
 * Runs every tensor, OpenMP-parallel over tensors; results are written per tensor. Returns threads used. */
int dplo_batch(const float* const* ptrs, const int64_t* counts, int64_t n_tensors, int algo, int bins, int threads,
               float* out_min, float* out_max, float* out_s, int64_t* out_hist) {
    int used = 1;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel
    {
#pragma omp single
        used = omp_get_num_threads();
    }
#endif
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t t = 0; t < n_tensors; t++) {
        dplo_minmax(ptrs[t], counts[t], &out_min[t], &out_max[t]);
        if (algo == 1) {
            const float a = out_max[t], b = -out_min[t];
            const float dmax = b > a ? b : a;
            dplo_abs_hist(ptrs[t], counts[t], bins, dmax, out_hist + (int64_t)t * bins);
        } else if (algo == 2) {
            float* scratch = (float*)malloc(sizeof(float) * 2 * (size_t)counts[t]);
            out_s[t] = dplo_octav_scale(ptrs[t], counts[t], 1, scratch);
            free(scratch);
        }
    }
    return used;
}
