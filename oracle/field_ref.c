/*
 * ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the reference's field head,
 *   /root/reference/nerf/network_tcnn.py:13-32  (MLP: Linear+ReLU stack, bias=True)
 *   /root/reference/nerf/network_tcnn.py:94-112 (gaussian blob, common_forward)
 *   /root/reference/activation.py:5-18          (trunc_exp forward = exp)
 * on top of the hash-grid restatement in hashgrid_ref.c.
 *
 * half_mode = 0 : everything fp32 (the parity mode, tolerance 1e-4 relative).
 * half_mode = 1 : emulates torch.autocast(fp16) around the nn.Linear stack the way
 *                 the reference trains (nerf/utils.py:979): encoder output (fp32) and the
 *                 fp32 master weights/bias are rounded to fp16, products accumulate in fp32,
 *                 each layer's output is rounded to fp16; the blob and trunc_exp are evaluated
 *                 in fp32 on the fp16-rounded MLP output (h[..., 0] + gaussian(x) promotes to fp32,
 *                 activation.py:7 casts to fp32), while torch.sigmoid(h[..., 1:]) of the fp16 tensor
 *                 (network_tcnn.py:110) RETURNS fp16: the albedo is rounded to fp16 as well.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORACLE_API __attribute__((visibility("default")))

void ref_hashgrid_forward(const float *x, uint32_t n, const float *params, uint32_t n_levels,
                          uint32_t base_resolution, float per_level_scale, uint32_t log2_hashmap_size, float *out);

/* round-to-nearest-even float -> IEEE binary16 -> float */
static inline float round_half(float f) {
    uint32_t x; memcpy(&x, &f, 4);
    const uint32_t sign = x & 0x80000000u;
    uint32_t a = x & 0x7FFFFFFFu;
    float r;
    if (a >= 0x7F800000u) return f;                       /* inf / nan */
    if (a >= 0x477FF000u) {                                /* >= 65520 -> inf */
        uint32_t inf = sign | 0x7F800000u; memcpy(&r, &inf, 4); return r;
    }
    if (a < 0x38800000u) {                                 /* < 2^-14 : subnormal half, quantum 2^-24 */
        float af; memcpy(&af, &a, 4);
        const float q = 5.9604644775390625e-08f;           /* 2^-24 */
        float k = nearbyintf(af / q);                      /* default rounding mode = RNE */
        af = k * q;
        memcpy(&a, &af, 4);
        a |= sign; memcpy(&r, &a, 4); return r;
    }
    /* normal: keep 10 mantissa bits, RNE on the 13 dropped bits */
    const uint32_t lsb = (a >> 13) & 1u;
    a += 0x0FFFu + lsb;
    a &= 0xFFFFE000u;
    a |= sign; memcpy(&r, &a, 4); return r;
}

ORACLE_API float ref_round_half(float f) { return round_half(f); }

/* One point through the Linear/ReLU stack. W_l is [out_l][in_l] row-major (torch nn.Linear.weight). */
static void mlp_point(const float *in, uint32_t num_layers, uint32_t dim_in, uint32_t dim_hidden, uint32_t dim_out,
                      const float *const *W, const float *const *B, int half_mode, float *out /* dim_out */) {
    float a[256], b[256];
    const float *cur = in; uint32_t cur_dim = dim_in;
    float *bufs[2] = {a, b};
    if (half_mode) { for (uint32_t i = 0; i < dim_in; i++) a[i] = round_half(in[i]); cur = a; bufs[0] = b; bufs[1] = a; }
    for (uint32_t l = 0; l < num_layers; l++) {
        const uint32_t od = (l == num_layers - 1) ? dim_out : dim_hidden;
        float *dst = (l == num_layers - 1) ? out : bufs[l & 1];
        for (uint32_t o = 0; o < od; o++) {
            float acc = 0;
            const float *w = W[l] + (size_t)o * cur_dim;
            for (uint32_t i = 0; i < cur_dim; i++) acc += (half_mode ? round_half(w[i]) : w[i]) * cur[i];
            acc += half_mode ? round_half(B[l][o]) : B[l][o];
            if (half_mode) acc = round_half(acc);
            if (l != num_layers - 1 && acc < 0) acc = 0;   /* F.relu */
            dst[o] = acc;
        }
        cur = dst; cur_dim = od;
    }
}

/* network_tcnn.py:102-112.  x: [n,3] world coords in [-bound, bound].
 * wb: W0,b0,W1,b1,... concatenated as separate pointers. sigma: [n], albedo: [n,3]. feat (optional): [n, L*2] */
ORACLE_API void ref_field_density(const float *x, uint32_t n, float bound, const float *params, uint32_t n_levels,
                                  uint32_t base_resolution, float per_level_scale, uint32_t log2_hashmap_size,
                                  uint32_t num_layers, uint32_t dim_hidden, const float *const *W,
                                  const float *const *B, double blob_density, double blob_radius, int half_mode,
                                  float *sigma, float *albedo, float *raw /* optional [n,4] */) {
    const uint32_t dim_in = n_levels * 2;
    float *h01 = (float *)malloc((size_t)n * 3 * sizeof(float));
    float *feat = (float *)malloc((size_t)n * dim_in * sizeof(float));
    for (size_t i = 0; i < (size_t)n * 3; i++) h01[i] = (x[i] + bound) / (2 * bound); /* :106 */
    ref_hashgrid_forward(h01, n, params, n_levels, base_resolution, per_level_scale, log2_hashmap_size, feat);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)n; i++) {
        float h[4];
        mlp_point(feat + (size_t)i * dim_in, num_layers, dim_in, dim_hidden, 4, W, B, half_mode, h);
        const float *p = x + (size_t)i * 3;
        const float d = p[0] * p[0] + p[1] * p[1] + p[2] * p[2];               /* :97 */
        /* :98 - python evaluates 2*r**2 in double, torch then divides the fp32 tensor by that scalar */
        const float g = (float)blob_density * expf(-d / (float)(2 * blob_radius * blob_radius));
        sigma[i] = expf(h[0] + g);                                              /* :109 trunc_exp fwd */
        for (int c = 0; c < 3; c++) {                                           /* :110 */
            const float a = 1.0f / (1.0f + expf(-h[1 + c]));
            albedo[(size_t)i * 3 + c] = half_mode ? round_half(a) : a;          /* sigmoid(fp16) -> fp16 */
        }
        if (raw) memcpy(raw + (size_t)i * 4, h, sizeof h);
    }
    free(h01); free(feat);
}
