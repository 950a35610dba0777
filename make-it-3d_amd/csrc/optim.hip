// Adan (Adaptive Nesterov Momentum) as ONE elementwise pass per parameter tensor, for MI355X (gfx950).
// Replaces the ~12 elementwise passes (+ a host sync for the clip factor) of the reference's optimizer step
// (/root/reference/optimizer.py:100-249: `Adan.step` + `_single_tensor_adan`) - Part 6 of include/mi3d.h.
// The 48.8 MB hash table dominates: 6 streams read (param, grad, exp_avg, exp_avg_diff, exp_avg_sq, neg_pre_grad),
// 6 written = 585 MB per step = 0.1 ms at HBM speed, against ~1 ms for the per-op sequence.
//
// The global-norm clip factor never leaves the device: mi3d_sumsq_accumulate adds sum(g^2) of every gradient tensor
// into one device float, and the update kernels derive clip = min(max_grad_norm / (sqrt(sum) + eps), 1) from it.
#include <hip/hip_runtime.h>

#include "../../include/mi3d.h"

namespace {

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

__global__ __launch_bounds__(256) void k_sumsq(const float *__restrict__ x, size_t n, float *__restrict__ acc) {
    __shared__ float part[4];
    float s = 0.f;
    const size_t stride = (size_t)gridDim.x * blockDim.x * 4;
    for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
        if (i + 4 <= n) {
            const float4 v = *reinterpret_cast<const float4 *>(x + i);
            s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        } else {
            for (size_t j = i; j < n; ++j) s += x[j] * x[j];
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) unsafeAtomicAdd(acc, part[0] + part[1] + part[2] + part[3]);
}

struct AdanArgs {
    float b1, b2, b3, bc1, bc2, bc3_sqrt, lr, wd, eps, max_grad_norm, clip_eps;
    int no_prox, first_step;
};

// one element, in the operation order of _single_tensor_adan (optimizer.py:195-249)
__device__ __forceinline__ void adan_one(float &p, float &g, float &m, float &n, float &d, float &npg, float clip,
                                         const AdanArgs &a) {
    g *= clip;
    if (a.first_step) npg = -g;            // state['neg_pre_grad'] = grad * -clip  (optimizer.py:151-153)
    npg += g;                              // g_t - g_{t-1}
    m = m * a.b1 + g * (1.0f - a.b1);
    d = d * a.b2 + npg * (1.0f - a.b2);
    npg = npg * a.b2 + g;                  // g_t + b2 (g_t - g_{t-1})
    n = n * a.b3 + npg * npg * (1.0f - a.b3);
    const float denom = sqrtf(n) / a.bc3_sqrt + a.eps;
    const float step = a.lr / a.bc1, step_diff = a.lr * a.b2 / a.bc2;
    if (a.no_prox) {
        p *= 1.0f - a.lr * a.wd;
        p += -step * (m / denom);
        p += -step_diff * (d / denom);
    } else {
        p += -step * (m / denom);
        p += -step_diff * (d / denom);
        p /= 1.0f + a.lr * a.wd;
    }
    npg = -g;                              // neg_grad_or_diff.zero_().add_(grad, alpha=-1.0)
}

__global__ __launch_bounds__(256) void k_adan(float *__restrict__ p, float *__restrict__ g, float *__restrict__ m,
                                               float *__restrict__ n, float *__restrict__ d, float *__restrict__ npg,
                                               size_t count, const float *__restrict__ sumsq, AdanArgs a) {
    float clip = 1.0f;
    if (a.max_grad_norm > 0.f && sumsq != nullptr)
        clip = fminf(a.max_grad_norm / (sqrtf(*sumsq) + a.clip_eps), 1.0f);
    const size_t stride = (size_t)gridDim.x * blockDim.x * 4;
    for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < count; i += stride) {
        if (i + 4 <= count) {
            float4 P = *reinterpret_cast<float4 *>(p + i), G = *reinterpret_cast<float4 *>(g + i),
                   M = *reinterpret_cast<float4 *>(m + i), N = *reinterpret_cast<float4 *>(n + i),
                   D = *reinterpret_cast<float4 *>(d + i), Q = *reinterpret_cast<float4 *>(npg + i);
            adan_one(P.x, G.x, M.x, N.x, D.x, Q.x, clip, a);
            adan_one(P.y, G.y, M.y, N.y, D.y, Q.y, clip, a);
            adan_one(P.z, G.z, M.z, N.z, D.z, Q.z, clip, a);
            adan_one(P.w, G.w, M.w, N.w, D.w, Q.w, clip, a);
            *reinterpret_cast<float4 *>(p + i) = P; *reinterpret_cast<float4 *>(g + i) = G;
            *reinterpret_cast<float4 *>(m + i) = M; *reinterpret_cast<float4 *>(n + i) = N;
            *reinterpret_cast<float4 *>(d + i) = D; *reinterpret_cast<float4 *>(npg + i) = Q;
        } else {
            for (size_t j = i; j < count; ++j) adan_one(p[j], g[j], m[j], n[j], d[j], npg[j], clip, a);
        }
    }
}

inline int grid_for(size_t n) {
    const size_t wgs = (n + 1023) / 1024;
    return (int)(wgs < 2048 ? (wgs ? wgs : 1) : 2048);
}

}  // namespace

extern "C" {

int mi3d_sumsq_accumulate(const float *x, size_t n, float *acc, void *stream) {
    if (n == 0) return 0;
    if ((reinterpret_cast<uintptr_t>(x) & 15u) != 0) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(k_sumsq, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), x, n, acc);
    return (int)hipGetLastError();
}

int mi3d_adan_step(float *param, float *grad, float *exp_avg, float *exp_avg_sq, float *exp_avg_diff,
                   float *neg_pre_grad, size_t count, const float *grad_sumsq, float max_grad_norm, float clip_eps,
                   int first_step, float beta1, float beta2, float beta3, float bias_correction1,
                   float bias_correction2, float bias_correction3_sqrt, float lr, float weight_decay, float eps,
                   int no_prox, void *stream) {
    if (count == 0) return 0;
    const float *ptrs[6] = {param, grad, exp_avg, exp_avg_sq, exp_avg_diff, neg_pre_grad};
    for (const float *q : ptrs)
        if ((reinterpret_cast<uintptr_t>(q) & 15u) != 0) return (int)hipErrorInvalidValue;
    const AdanArgs a{beta1, beta2, beta3, bias_correction1, bias_correction2, bias_correction3_sqrt, lr, weight_decay,
                     eps, max_grad_norm, clip_eps, no_prox, first_step};
    hipLaunchKernelGGL(k_adan, dim3(grid_for(count)), dim3(256), 0, as_stream(stream), param, grad, exp_avg, exp_avg_sq,
                       exp_avg_diff, neg_pre_grad, count, grad_sumsq, a);
    return (int)hipGetLastError();
}

}  // extern "C"
