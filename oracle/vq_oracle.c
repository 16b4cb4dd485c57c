/* CPU oracle for the nearest-codeword assignment -- TEST INFRASTRUCTURE ONLY.
 *
 * Restates  vqvae/modules/vector_quantizers.py:37-44 (Standard / EMA:  d = (|z|^2 + |e|^2) - 2 z.e)
 * and       vqvae/modules/vector_quantizers.py:337-343 (Entropy:        d = (|z|^2 - 2 z.e) + |e|^2)
 * followed by torch.argmin (first minimum wins) in fp32, with ONE canonical fp32 evaluation
 * order so that "bit-exact indices" is a well-defined property:
 *
 *   sqnorm(v)  : 64 partial sums p[l] = fma-chain over v[l], v[l+64], ... ; then an xor-butterfly
 *                p[l] += p[l^off], off = 32,16,8,4,2,1   (the order a 64-lane wavefront reduces in)
 *   dot(z,e)   : one fp32 fma chain visiting k in the order 8j+{0,4,1,5,2,6,3,7}, j = 0..D/8-1
 *                (the order v_mfma_f32_32x32x2_f32 consumes k when each lane holds a float4)
 *   distance   : fl(fl(z2 + e2) - fl(2*dot))           (assoc = 0, Standard/EMA)
 *                fl(fl(z2 - fl(2*dot)) + e2)           (assoc = 1, Entropy)
 *
 * SURVEY Appendix C measured that re-ordering the GEMM accumulation does not change any index vs
 * torch on the tested data; tests/test_oracle_golden.py re-checks that against fixtures captured
 * from the reference itself.
 *
 * Build: gcc -O2 -ffp-contract=off -mfma -shared -fPIC (oracle/Makefile).
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>

static float sqnorm64(const float* v, int d) {
    float p[64];
    for (int l = 0; l < 64; ++l) {
        float acc = 0.0f;
        for (int k = l; k < d; k += 64) acc = fmaf(v[k], v[k], acc);
        p[l] = acc;
    }
    for (int off = 32; off >= 1; off >>= 1) {
        float t[64];
        for (int l = 0; l < 64; ++l) t[l] = p[l] + p[l ^ off];
        for (int l = 0; l < 64; ++l) p[l] = t[l];
    }
    return p[0];
}

static float dot_mfma_order(const float* a, const float* b, int d) {
    float acc = 0.0f;
    for (int j = 0; j < d; j += 8)
        for (int t = 0; t < 4; ++t) {
            acc = fmaf(a[j + t], b[j + t], acc);
            acc = fmaf(a[j + 4 + t], b[j + 4 + t], acc);
        }
    return acc;
}

/* z[n][d], e[k][d] row-major fp32; idx[n] int64; optional dmin[n]; d % 8 == 0. */
void vq_oracle_assign(const float* z, const float* e, int64_t n, int64_t k, int d, int assoc,
                      int64_t* idx, float* dmin, float* z2_out, float* e2_out) {
    for (int64_t j = 0; j < k; ++j) e2_out[j] = sqnorm64(e + j * d, d);
    for (int64_t i = 0; i < n; ++i) {
        const float* zi = z + i * d;
        const float z2 = sqnorm64(zi, d);
        z2_out[i] = z2;
        float best = INFINITY;
        int64_t bi = 0;
        for (int64_t j = 0; j < k; ++j) {
            const float ab2 = 2.0f * dot_mfma_order(zi, e + j * d, d);
            float dist;
            if (assoc == 0) { volatile float s = z2 + e2_out[j]; dist = s - ab2; }
            else            { volatile float s = z2 - ab2;       dist = s + e2_out[j]; }
            if (dist < best || j == 0) { best = dist; bi = j; }
        }
        idx[i] = bi;
        if (dmin) dmin[i] = best;
    }
}
