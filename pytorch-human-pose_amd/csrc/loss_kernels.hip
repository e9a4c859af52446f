// Training loss of the associative-embedding path: masked heatmap MSE and the AE push/pull
// grouping loss, each fused with its gradient.  Restates /root/reference/src/keypoints/loss.py:
//   HeatmapsLoss.forward :12-16, AEGroupingLoss.forward :20-61 (a python triple loop issuing one tiny device op per
//   visible joint in the reference, host-bound; here one workgroup per image).
// Both kernels are HBM/latency bound (no matrix work): the MSE reads pred+target+mask once and writes the gradient
// once (coalesced, 16 B per lane); the grouping loss gathers <= P*K scalars per image.
// Sums are accumulated in double in a fixed order, so results do not depend on the launch.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace {
__device__ inline double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;  // lane 0
}
__device__ inline double block_sum256(double v, double *sh)  // fixed order: wave shuffles, then waves 0..3
{
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    const double r = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
    return r;
}
}  // namespace

// grad[b,k,y,x] = 2 (pred - target) mask[b,y,x] / N ; partial[block] = sum (pred-target)^2 mask over the block's items.
// One block walks whole (b,k) planes: plane = blockIdx.x + i*gridDim.x, 4 pixels per lane per step (hw % 4 == 0).
__global__ __launch_bounds__(256) void masked_mse_kernel(const float *__restrict__ pred, int64_t pred_bs, const float *__restrict__ target,
                                                         const float *__restrict__ mask, int B, int K, int hw,
                                                         float *__restrict__ grad, int64_t grad_bs, float gscale,
                                                         double *__restrict__ partial)
{
    __shared__ double sh[4];
    double acc = 0.0;
    const int hw4 = hw >> 2;
    for (int plane = blockIdx.x; plane < B * K; plane += gridDim.x) {
        const int b = plane / K, k = plane % K;
        const float4 *p = reinterpret_cast<const float4 *>(pred + (size_t)b * pred_bs + (size_t)k * hw);
        const float4 *t = reinterpret_cast<const float4 *>(target + ((size_t)b * K + k) * hw);
        const float4 *m = reinterpret_cast<const float4 *>(mask + (size_t)b * hw);
        float4 *g = grad ? reinterpret_cast<float4 *>(grad + (size_t)b * grad_bs + (size_t)k * hw) : nullptr;
        for (int i = threadIdx.x; i < hw4; i += 256) {
            const float4 pv = p[i], tv = t[i], mv = m[i];
            const float dx = pv.x - tv.x, dy = pv.y - tv.y, dz = pv.z - tv.z, dw = pv.w - tv.w;
            acc += (double)(dx * dx * mv.x) + (double)(dy * dy * mv.y) + (double)(dz * dz * mv.z) + (double)(dw * dw * mv.w);
            if (g) g[i] = make_float4(gscale * dx * mv.x, gscale * dy * mv.y, gscale * dz * mv.z, gscale * dw * mv.w);
        }
    }
    const double s = block_sum256(acc, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void sum_partials_kernel(const double *__restrict__ partial, int n, double scale, float *__restrict__ out)
{
    __shared__ double sh[4];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) acc += partial[i];
    const double s = block_sum256(acc, sh);
    if (threadIdx.x == 0) *out = (float)(s * scale);
}

hipError_t launch_masked_mse(const float *pred, int64_t pred_bs, const float *target, const float *mask, int B, int K, int h, int w,
                             float *loss, float *grad, int64_t grad_bs, double *scratch, hipStream_t s)
{
    const int planes = B * K, blocks = planes < HH_LOSS_SCRATCH ? planes : HH_LOSS_SCRATCH;
    const double n = (double)B * K * h * w;
    hipLaunchKernelGGL(masked_mse_kernel, dim3(blocks), dim3(256), 0, s, pred, pred_bs, target, mask, B, K, h * w, grad, grad_bs,
                       (float)(2.0 / n), scratch);
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, s, scratch, blocks, 1.0 / n, loss);
    return hipGetLastError();
}

// One workgroup per image.  joints [B,P,K,3] int32 (x, y, vis), people beyond num_people[b] ignored.
// LDS: ref[P] (mean tag of the person), cnt[P] (visible joints), dref[P] (d push / d ref).
__global__ __launch_bounds__(256) void ae_grouping_kernel(const float *__restrict__ tags, int64_t tags_bs, const int32_t *__restrict__ joints,
                                                          const int32_t *__restrict__ num_people, int P, int K, int h, int w,
                                                          float *__restrict__ grad, int64_t grad_bs, float push_scale, float pull_scale,
                                                          double *__restrict__ per_image, int B)
{
    extern __shared__ double lds[];
    double *ref = lds, *dref = lds + P, *red = lds + 2 * P;  // red[4]
    int *cnt = reinterpret_cast<int *>(red + 4);
    const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int np = min(num_people[b], P);
    const float *T = tags + (size_t)b * tags_bs;
    const int32_t *J = joints + (size_t)b * P * K * 3;
    const size_t hw = (size_t)h * w;

    // ---- per person (one wave each): reference tag = mean over visible joints, pull term (loss.py:25-38)
    double pull_acc = 0.0;  // lane 0 of each wave
    for (int p = wave; p < np; p += 4) {
        double s = 0.0;
        int n = 0;
        for (int k = lane; k < K; k += 64) {
            const int32_t *j = J + ((size_t)p * K + k) * 3;
            if (j[2] > 0) { s += (double)T[k * hw + (size_t)j[1] * w + j[0]]; ++n; }
        }
        s = wave_sum(s);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) n += __shfl_down(n, o, 64);
        n = __shfl(n, 0, 64);
        const double m = __shfl(s, 0, 64) / (n > 0 ? n : 1);
        double q = 0.0;
        for (int k = lane; k < K; k += 64) {
            const int32_t *j = J + ((size_t)p * K + k) * 3;
            if (j[2] > 0) { const double d = (double)T[k * hw + (size_t)j[1] * w + j[0]] - m; q += d * d; }
        }
        q = wave_sum(q);
        if (lane == 0) { ref[p] = m; cnt[p] = n; if (n > 0) pull_acc += q / n; }
    }
    if (lane == 0) red[wave] = pull_acc;
    __syncthreads();
    const double pull_sum = red[0] + red[1] + red[2] + red[3];
    int nobj = 0;
    for (int p = 0; p < np; ++p) nobj += cnt[p] > 0;  // every thread: np is small
    __syncthreads();

    // ---- push over pairs of reference tags (loss.py:50-60) and d push / d ref
    double push = 0.0;
    if (nobj > 1) {
        const double c = 0.5 / ((double)(nobj - 1) * nobj);
        double acc = 0.0;
        for (int a = tid; a < np; a += 256) {
            double g = 0.0;
            if (cnt[a] > 0)
                for (int q = 0; q < np; ++q)
                    if (cnt[q] > 0) {
                        const double d = ref[a] - ref[q], e = exp(-d * d);
                        acc += e;
                        g += -4.0 * d * e;  // (a,q) and (q,a) both depend on ref[a]
                    }
            dref[a] = g * c;
        }
        const double tot = block_sum256(acc, red);
        push = (tot - nobj) * c;
    } else {
        for (int a = tid; a < np; a += 256) dref[a] = 0.0;
        __syncthreads();
    }
    if (tid == 0) {
        per_image[b] = push;                                   // loss.py:60
        per_image[B + b] = nobj > 0 ? pull_sum / nobj : 0.0;   // loss.py:45,48
    }
    if (!grad || nobj == 0) return;

    // ---- gradient, scattered onto the tag map (two people may share a pixel: atomic add)
    float *G = grad + (size_t)b * grad_bs;
    const double ps = (double)push_scale / B, ls = (double)pull_scale / B / nobj;
    for (int p = wave; p < np; p += 4) {
        const int n = cnt[p];
        if (n == 0) continue;
        const double m = ref[p], gp = ps * dref[p] / n;
        for (int k = lane; k < K; k += 64) {
            const int32_t *j = J + ((size_t)p * K + k) * 3;
            if (j[2] > 0) {
                const size_t o = k * hw + (size_t)j[1] * w + j[0];
                atomicAdd(G + o, (float)(gp + ls * 2.0 * ((double)T[o] - m) / n));
            }
        }
    }
}

__global__ void ae_finalize_kernel(const double *__restrict__ per_image, int B, float *__restrict__ out)
{
    if (threadIdx.x || blockIdx.x) return;
    double push = 0.0, pull = 0.0;
    for (int b = 0; b < B; ++b) { push += per_image[b]; pull += per_image[B + b]; }
    out[0] = (float)(push / B);  // loss.py:61
    out[1] = (float)(pull / B);
}

hipError_t launch_ae_grouping(const float *tags, int64_t tags_bs, const int32_t *joints, const int32_t *num_people, int B, int P, int K,
                              int h, int w, float *push_pull, float *grad, int64_t grad_bs, float push_scale, float pull_scale,
                              double *scratch, hipStream_t s)
{
    const size_t lds = (size_t)(2 * P + 4) * sizeof(double) + (size_t)P * sizeof(int);
    hipLaunchKernelGGL(ae_grouping_kernel, dim3(B), dim3(256), lds, s, tags, tags_bs, joints, num_people, P, K, h, w, grad, grad_bs,
                       push_scale, pull_scale, scratch, B);
    hipLaunchKernelGGL(ae_finalize_kernel, dim3(1), dim3(64), 0, s, scratch, B, push_pull);
    return hipGetLastError();
}
