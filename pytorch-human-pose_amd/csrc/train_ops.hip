// Building blocks of the training path (first half): convolution with *current* fp32 weights
// (forward, and the data gradient of a stride-1 conv as a forward conv with rotated, transposed weights) and train-mode
// BatchNorm forward / backward, all on NHWC bf16 activations.  The convolution itself is the conv_mfma family; what is
// new here is packing the weights ON THE DEVICE every step (the inference engine packs once on the host at finalize).
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace {
__device__ __forceinline__ bf16_raw f2bf_dev(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }
}

// packed[cg][chunk][tap][c8][co_in][j]  <-  W, see hh_pack_weights (engine.cpp) for the host twin.
//   mode 0: conv weights W[cout][cin][ks][ks]                      -> out channel o = co, in channel i = ci, tap (ky,kx)
//   mode 1: data gradient of a stride-1 conv: the forward conv of dL/dy with W'[ci][co][ks-1-ky][ks-1-kx]
//           (cout_eff = cin, cin_eff = cout)
//   mode 2: data gradient of a 3x3 stride-2 conv, one output-parity phase (py, px) as a 2x2 conv over dL/dy:
//           dX[2i+py, 2j+px] = sum_t W'[t] dY[i + ty, j + tx];  even parity uses the centre tap only (ty = 0 <-> ky = 1),
//           odd parity ty = 0 <-> ky = 2 and ty = 1 <-> ky = 0 (same in x).  `ks` is then 2 (the packed kernel size).
__device__ __forceinline__ void pack_weights_range(const float *__restrict__ W, int cout, int cin, int ks, int mode, int KC, int COUT_T,
                                                   bf16_raw *__restrict__ packed, size_t total, int py, int px, size_t first, size_t step)
{
    const int co_eff = mode ? cin : cout, ci_eff = mode ? cout : cin;
    const int cin_pad = (ci_eff + KC - 1) / KC * KC, nch = cin_pad / KC, taps = ks * ks, C8 = KC / 8;
    for (size_t o = first; o < total; o += step) {
        size_t r = o;
        const int j = r % 8; r /= 8;
        const int co_in = r % COUT_T; r /= COUT_T;
        const int c8 = r % C8; r /= C8;
        const int t = r % taps; r /= taps;
        const int ch = r % nch; r /= nch;
        const int cg = (int)r;
        const int oc = cg * COUT_T + co_in, ic = ch * KC + c8 * 8 + j, ky = t / ks, kx = t % ks;
        float v = 0.f;
        if (oc < co_eff && ic < ci_eff) {
            if (mode == 2) {
                const int sy = py ? (ky ? 0 : 2) : (ky ? -1 : 1), sx = px ? (kx ? 0 : 2) : (kx ? -1 : 1);  // source tap of the 3x3 kernel
                if (sy >= 0 && sx >= 0) v = W[(((size_t)ic * cin + oc) * 3 + sy) * 3 + sx];
            } else if (mode == 1) {
                v = W[(((size_t)ic * cin + oc) * ks + (ks - 1 - ky)) * ks + (ks - 1 - kx)];  // W[co = ic][ci = oc][rot]
            } else {
                v = W[(((size_t)oc * cin + ic) * ks + ky) * ks + kx];
            }
        }
        packed[o] = f2bf_dev(v);
    }
}
__global__ __launch_bounds__(256) void pack_weights_kernel(const float *__restrict__ W, int cout, int cin, int ks, int mode, int KC,
                                                           int COUT_T, bf16_raw *__restrict__ packed, size_t total, int py, int px)
{
    pack_weights_range(W, cout, cin, ks, mode, KC, COUT_T, packed, total, py, px, (size_t)blockIdx.x * 256 + threadIdx.x, (size_t)gridDim.x * 256);
}
// Every weight set of a training step in ONE launch (blockIdx.y = descriptor): ~700 tiny dependent launches per step cost more
// in launch gaps than the packing itself.
__global__ __launch_bounds__(256) void pack_weights_batch_kernel(const PackDesc *__restrict__ descs)
{
    const PackDesc d = descs[blockIdx.y];
    pack_weights_range(d.W, d.cout, d.cin, d.ks, d.mode, d.KC, d.COUT_T, d.packed, (size_t)d.total, d.py, d.px,
                       (size_t)blockIdx.x * 256 + threadIdx.x, (size_t)gridDim.x * 256);
}

hipError_t launch_pack_weights_batch(const PackDesc *descs_dev, int n, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(pack_weights_batch_kernel, dim3(32, n), dim3(256), 0, s, descs_dev);
    return hipGetLastError();
}

hipError_t launch_pack_weights(const float *W, int cout, int cin, int ks, int mode, int KC, int COUT_T, bf16_raw *packed, size_t total,
                               hipStream_t s, int py, int px)
{
    unsigned grid = (unsigned)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(pack_weights_kernel, dim3(grid), dim3(256), 0, s, W, cout, cin, ks, mode, KC, COUT_T, packed, total, py, px);
    return hipGetLastError();
}

// ------------------------------------------------------------------ train-mode BatchNorm (nn.BatchNorm2d, training=True)
// x [P, C] bf16 (P = B*H*W pixels, channel stride cs).  Statistics in fp32 with double partial sums in a fixed order.
// stats kernel: partial[block][c] = {sum x, sum x^2} over the block's pixels; finalize: mean, biased var -> invstd.
__global__ __launch_bounds__(256) void bn_partial_kernel(const bf16_raw *__restrict__ x, int cs, size_t P, int C, double *__restrict__ partial)
{
    // thread = (pixel lane pl = tid / C8, channel group g = tid % C8) over 8 channels; C % 8 == 0, C <= 2048
    const int C8 = C / 8, g = threadIdx.x % C8, pl = threadIdx.x / C8, npl = 256 / C8;
    double s[8] = {}, q[8] = {};
    if (pl < npl) {
        // UB pixels per trip, their loads issued together: one load in flight per thread is a memory round trip per 16 bytes
        constexpr int UB = 4;
        const size_t step = (size_t)gridDim.x * npl;
        auto add = [&](const uint4 &v) {
            const unsigned u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float a = __builtin_bit_cast(float, u[i] << 16), b = __builtin_bit_cast(float, u[i] & 0xffff0000u);
                s[2 * i] += a; q[2 * i] += (double)a * a; s[2 * i + 1] += b; q[2 * i + 1] += (double)b * b;
            }
        };
        size_t p = (size_t)blockIdx.x * npl + pl;
        for (; p + (UB - 1) * step < P; p += UB * step) {
            uint4 v[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) v[u] = *reinterpret_cast<const uint4 *>(x + (p + u * step) * cs + g * 8);
#pragma unroll
            for (int u = 0; u < UB; ++u) add(v[u]);
        }
        for (; p < P; p += step) add(*reinterpret_cast<const uint4 *>(x + p * cs + g * 8));
    }
    __shared__ double sh[2][256][8 + 1];
#pragma unroll
    for (int i = 0; i < 8; ++i) { sh[0][threadIdx.x][i] = s[i]; sh[1][threadIdx.x][i] = q[i]; }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {  // fixed order over the pixel lanes
        const int gg = c / 8, i = c % 8;
        double a = 0, b = 0;
        for (int l = 0; l < npl; ++l) { a += sh[0][l * C8 + gg][i]; b += sh[1][l * C8 + gg][i]; }
        partial[((size_t)blockIdx.x * C + c) * 2] = a;
        partial[((size_t)blockIdx.x * C + c) * 2 + 1] = b;
    }
}
// one wave per channel: lane l adds partials l, l+64, ... and a shuffle tree finishes (fixed order: deterministic)
__device__ __forceinline__ void sum_partials_wave(const double *__restrict__ partial, int nblocks, int C, int c, double &a, double &b)
{
    a = 0; b = 0;
    for (int i = threadIdx.x & 63; i < nblocks; i += 64) { a += partial[((size_t)i * C + c) * 2]; b += partial[((size_t)i * C + c) * 2 + 1]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_down(a, o, 64); b += __shfl_down(b, o, 64); }
}
__global__ __launch_bounds__(256) void bn_finalize_kernel(const double *__restrict__ partial, int nblocks, int C, double P, float eps,
                                                          float *__restrict__ mean, float *__restrict__ invstd)
{
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= C) return;
    double a, b;
    sum_partials_wave(partial, nblocks, C, c, a, b);
    if (threadIdx.x & 63) return;
    const double m = a / P, var = b / P - m * m;  // biased variance, as F.batch_norm normalises with
    mean[c] = (float)m;
    invstd[c] = (float)(1.0 / sqrt((var > 0 ? var : 0) + (double)eps));
}
// gamma * (x - mean) * invstd + beta: ONE function for the forward and for the backward passes that recompute the ReLU mask from x
// (HASY = false below), so that both evaluate the same instruction sequence
__device__ __forceinline__ float bn_affine(float x, float mean, float invstd, float gamma, float beta)
{
    return (x - mean) * invstd * gamma + beta;
}
// y = act(gamma * (x - mean) * invstd + beta (+ res))
__global__ __launch_bounds__(256) void bn_apply_kernel(const bf16_raw *__restrict__ x, int cs, size_t P, int C, const float *__restrict__ mean,
                                                       const float *__restrict__ invstd, const float *__restrict__ gamma,
                                                       const float *__restrict__ beta, const bf16_raw *__restrict__ res, int relu,
                                                       bf16_raw *__restrict__ y)
{
    const int C8 = C / 8;
    const size_t total = P * C8;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t p = i / C8;
        const int g = (int)(i % C8);
        const uint4 v = *reinterpret_cast<const uint4 *>(x + p * cs + g * 8);
        uint4 rv = make_uint4(0, 0, 0, 0);
        if (res) rv = *reinterpret_cast<const uint4 *>(res + p * cs + g * 8);
        const unsigned u[4] = {v.x, v.y, v.z, v.w}, ru[4] = {rv.x, rv.y, rv.z, rv.w};
        unsigned o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float f[2] = {__builtin_bit_cast(float, u[k] << 16), __builtin_bit_cast(float, u[k] & 0xffff0000u)};
            const float r[2] = {__builtin_bit_cast(float, ru[k] << 16), __builtin_bit_cast(float, ru[k] & 0xffff0000u)};
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int c = g * 8 + 2 * k + h;
                float t = bn_affine(f[h], mean[c], invstd[c], gamma[c], beta[c]) + r[h];
                f[h] = relu ? fmaxf(t, 0.f) : t;
            }
            o[k] = (unsigned)f2bf_dev(f[0]) | ((unsigned)f2bf_dev(f[1]) << 16);
        }
        *reinterpret_cast<uint4 *>(y + p * cs + g * 8) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

hipError_t launch_bn_train_forward(const bf16_raw *x, int cs, size_t P, int C, const float *gamma, const float *beta, float eps,
                                   const bf16_raw *res, int relu, bf16_raw *y, float *mean, float *invstd, double *scratch, hipStream_t s)
{
    const int nblocks = HH_BN_BLOCKS;
    hipLaunchKernelGGL(bn_partial_kernel, dim3(nblocks), dim3(256), 0, s, x, cs, P, C, scratch);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 3) / 4), dim3(256), 0, s, scratch, nblocks, C, (double)P, eps, mean, invstd);
    unsigned grid = (unsigned)((P * (C / 8) + 255) / 256);
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(bn_apply_kernel, dim3(grid), dim3(256), 0, s, x, cs, P, C, mean, invstd, gamma, beta, res, relu, y);
    return hipGetLastError();
}

// Backward of y = act(gamma * xhat + beta (+ res)), xhat = (x - mean) * invstd:
//   g = dy * (y > 0 if relu);  dbeta = sum g;  dgamma = sum g * xhat;
//   dx = gamma * invstd * (g - dbeta / P - xhat * dgamma / P);  dres = g (returned in place of dy when res was used)
// HASY = false (a BatchNorm without a residual input): y is not read -- without ReLU nothing needs it, with ReLU the mask y > 0 is
// recomputed from x (bn_affine(x) > 0: what the forward rounded to bf16 and clamped), one tensor pass less in both kernels
template <bool HASY>
__global__ __launch_bounds__(256) void bn_bwd_partial_kernel(const bf16_raw *__restrict__ x, const bf16_raw *__restrict__ y,
                                                             const bf16_raw *__restrict__ dy, int cs, size_t P, int C,
                                                             const float *__restrict__ mean, const float *__restrict__ invstd,
                                                             const float *__restrict__ gamma, const float *__restrict__ beta, int relu,
                                                             double *__restrict__ partial)
{
    const int C8 = C / 8, g = threadIdx.x % C8, pl = threadIdx.x / C8, npl = 256 / C8;
    double s[8] = {}, q[8] = {};
    if (pl < npl) {
        float mu[8], is[8], ga[8], be[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            mu[i] = mean[g * 8 + i]; is[i] = invstd[g * 8 + i];
            ga[i] = HASY ? 0.f : gamma[g * 8 + i]; be[i] = HASY ? 0.f : beta[g * 8 + i];
        }
        auto add = [&](const uint4 &xv, const uint4 &yv, const uint4 &dv) {
            const unsigned xu[4] = {xv.x, xv.y, xv.z, xv.w}, yu[4] = {yv.x, yv.y, yv.z, yv.w}, du[4] = {dv.x, dv.y, dv.z, dv.w};
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float xf = __builtin_bit_cast(float, h ? (xu[k] & 0xffff0000u) : (xu[k] << 16));
                    const float yf = HASY ? __builtin_bit_cast(float, h ? (yu[k] & 0xffff0000u) : (yu[k] << 16))
                                          : bn_affine(xf, mu[2 * k + h], is[2 * k + h], ga[2 * k + h], be[2 * k + h]);
                    float gf = __builtin_bit_cast(float, h ? (du[k] & 0xffff0000u) : (du[k] << 16));
                    if (relu && !(yf > 0.f)) gf = 0.f;
                    s[2 * k + h] += gf;
                    q[2 * k + h] += (double)gf * ((xf - mu[2 * k + h]) * is[2 * k + h]);
                }
        };
        constexpr int UB = 2;  // 3 tensors x UB pixels in flight per thread
        const size_t step = (size_t)gridDim.x * npl;
        size_t p = (size_t)blockIdx.x * npl + pl;
        for (; p + (UB - 1) * step < P; p += UB * step) {
            uint4 xv[UB], yv[UB], dv[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                xv[u] = *reinterpret_cast<const uint4 *>(x + (p + u * step) * cs + g * 8);
                yv[u] = HASY ? *reinterpret_cast<const uint4 *>(y + (p + u * step) * cs + g * 8) : make_uint4(0, 0, 0, 0);
                dv[u] = *reinterpret_cast<const uint4 *>(dy + (p + u * step) * cs + g * 8);
            }
#pragma unroll
            for (int u = 0; u < UB; ++u) add(xv[u], yv[u], dv[u]);
        }
        for (; p < P; p += step)
            add(*reinterpret_cast<const uint4 *>(x + p * cs + g * 8),
                HASY ? *reinterpret_cast<const uint4 *>(y + p * cs + g * 8) : make_uint4(0, 0, 0, 0),
                *reinterpret_cast<const uint4 *>(dy + p * cs + g * 8));
    }
    __shared__ double sh[2][256][8 + 1];
#pragma unroll
    for (int i = 0; i < 8; ++i) { sh[0][threadIdx.x][i] = s[i]; sh[1][threadIdx.x][i] = q[i]; }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        const int gg = c / 8, i = c % 8;
        double a = 0, b = 0;
        for (int l = 0; l < npl; ++l) { a += sh[0][l * C8 + gg][i]; b += sh[1][l * C8 + gg][i]; }
        partial[((size_t)blockIdx.x * C + c) * 2] = a;
        partial[((size_t)blockIdx.x * C + c) * 2 + 1] = b;
    }
}
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const double *__restrict__ partial, int nblocks, int C,
                                                              float *__restrict__ dgamma, float *__restrict__ dbeta)
{
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= C) return;
    double a, b;
    sum_partials_wave(partial, nblocks, C, c, a, b);
    if (threadIdx.x & 63) return;
    dbeta[c] = (float)a;
    dgamma[c] = (float)b;
}
template <bool HASY>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const bf16_raw *__restrict__ x, const bf16_raw *__restrict__ y,
                                                           const bf16_raw *__restrict__ dy, int cs, size_t P, int C,
                                                           const float *__restrict__ mean, const float *__restrict__ invstd,
                                                           const float *__restrict__ gamma, const float *__restrict__ beta,
                                                           const float *__restrict__ dgamma,
                                                           const float *__restrict__ dbeta, int relu, float invP,
                                                           bf16_raw *__restrict__ dx, bf16_raw *__restrict__ dres)
{
    const int C8 = C / 8;
    const size_t total = P * C8;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t p = i / C8;
        const int g = (int)(i % C8);
        const uint4 xv = *reinterpret_cast<const uint4 *>(x + p * cs + g * 8);
        const uint4 yv = HASY ? *reinterpret_cast<const uint4 *>(y + p * cs + g * 8) : make_uint4(0, 0, 0, 0);
        const uint4 dv = *reinterpret_cast<const uint4 *>(dy + p * cs + g * 8);
        const unsigned xu[4] = {xv.x, xv.y, xv.z, xv.w}, yu[4] = {yv.x, yv.y, yv.z, yv.w}, du[4] = {dv.x, dv.y, dv.z, dv.w};
        // the eight channels' parameters as 16-byte loads, unconditionally (behind a condition they come one float at a time)
        float mu[8], is[8], ga[8], be[8], dg[8], db[8];
        auto ld8 = [&](const float *src, float *dst) {
            const float4 a = *reinterpret_cast<const float4 *>(src + g * 8), b = *reinterpret_cast<const float4 *>(src + g * 8 + 4);
            dst[0] = a.x; dst[1] = a.y; dst[2] = a.z; dst[3] = a.w; dst[4] = b.x; dst[5] = b.y; dst[6] = b.z; dst[7] = b.w;
        };
        ld8(mean, mu); ld8(invstd, is); ld8(gamma, ga); ld8(dgamma, dg); ld8(dbeta, db);
        if constexpr (!HASY) ld8(beta, be);
        unsigned o[4], r[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float out[2], gr[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int c = 2 * k + h;
                const float xf = __builtin_bit_cast(float, h ? (xu[k] & 0xffff0000u) : (xu[k] << 16));
                const float yf = HASY ? __builtin_bit_cast(float, h ? (yu[k] & 0xffff0000u) : (yu[k] << 16))
                                      : bn_affine(xf, mu[c], is[c], ga[c], be[c]);
                float gf = __builtin_bit_cast(float, h ? (du[k] & 0xffff0000u) : (du[k] << 16));
                if (relu && !(yf > 0.f)) gf = 0.f;
                const float xh = (xf - mu[c]) * is[c];
                out[h] = ga[c] * is[c] * (gf - db[c] * invP - xh * dg[c] * invP);
                gr[h] = gf;
            }
            o[k] = (unsigned)f2bf_dev(out[0]) | ((unsigned)f2bf_dev(out[1]) << 16);
            r[k] = (unsigned)f2bf_dev(gr[0]) | ((unsigned)f2bf_dev(gr[1]) << 16);
        }
        *reinterpret_cast<uint4 *>(dx + p * cs + g * 8) = make_uint4(o[0], o[1], o[2], o[3]);
        if (dres) *reinterpret_cast<uint4 *>(dres + p * cs + g * 8) = make_uint4(r[0], r[1], r[2], r[3]);
    }
}

// y == nullptr: a BatchNorm without a residual input (dres must be nullptr, beta is then needed for the mask); otherwise beta is unused
hipError_t launch_bn_train_backward(const bf16_raw *x, const bf16_raw *y, const bf16_raw *dy, int cs, size_t P, int C, const float *mean,
                                    const float *invstd, const float *gamma, const float *beta, int relu, bf16_raw *dx, bf16_raw *dres,
                                    float *dgamma, float *dbeta, double *scratch, hipStream_t s)
{
    const int nblocks = HH_BN_BLOCKS;
    unsigned grid = (unsigned)((P * (C / 8) + 255) / 256);
    if (grid > 8192) grid = 8192;
    const float invP = (float)(1.0 / (double)P);
    if (y) {
        hipLaunchKernelGGL(bn_bwd_partial_kernel<true>, dim3(nblocks), dim3(256), 0, s, x, y, dy, cs, P, C, mean, invstd, gamma, beta, relu, scratch);
        hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 3) / 4), dim3(256), 0, s, scratch, nblocks, C, dgamma, dbeta);
        hipLaunchKernelGGL(bn_bwd_apply_kernel<true>, dim3(grid), dim3(256), 0, s, x, y, dy, cs, P, C, mean, invstd, gamma, beta, dgamma, dbeta, relu, invP, dx, dres);
    } else {
        hipLaunchKernelGGL(bn_bwd_partial_kernel<false>, dim3(nblocks), dim3(256), 0, s, x, y, dy, cs, P, C, mean, invstd, gamma, beta, relu, scratch);
        hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 3) / 4), dim3(256), 0, s, scratch, nblocks, C, dgamma, dbeta);
        hipLaunchKernelGGL(bn_bwd_apply_kernel<false>, dim3(grid), dim3(256), 0, s, x, y, dy, cs, P, C, mean, invstd, gamma, beta, dgamma, dbeta, relu, invP, dx, dres);
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------ SyncBatchNorm (base/model.py:42-44): the same passes,
// split around the one exchange step.  stats -> sums[2C] doubles {sum, sum of squares} of THIS rank's pixels; the caller
// all-reduces them (with the pixel count) over the ranks; normalize / backward_apply then use the global sums.
__global__ __launch_bounds__(256) void bn_finalize_sums_kernel(const double *__restrict__ partial, int nblocks, int C, double *__restrict__ sums,
                                                               float *__restrict__ fa, float *__restrict__ fb)
{
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= C) return;
    double a, b;
    sum_partials_wave(partial, nblocks, C, c, a, b);
    if (threadIdx.x & 63) return;
    sums[2 * c] = a;
    sums[2 * c + 1] = b;
    if (fa) { fa[c] = (float)a; fb[c] = (float)b; }
}
__global__ void bn_stats_from_sums_kernel(const double *__restrict__ sums, int C, double count, float eps, float *__restrict__ mean,
                                          float *__restrict__ invstd)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const double m = sums[2 * c] / count, var = sums[2 * c + 1] / count - m * m;
    mean[c] = (float)m;
    invstd[c] = (float)(1.0 / sqrt((var > 0 ? var : 0) + (double)eps));
}
__global__ void bn_sums_to_f32_kernel(const double *__restrict__ sums, int C, float *__restrict__ a, float *__restrict__ b)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    a[c] = (float)sums[2 * c];
    b[c] = (float)sums[2 * c + 1];
}

hipError_t launch_bn_train_stats(const bf16_raw *x, int cs, size_t P, int C, double *sums, double *scratch, hipStream_t s)
{
    hipLaunchKernelGGL(bn_partial_kernel, dim3(HH_BN_BLOCKS), dim3(256), 0, s, x, cs, P, C, scratch);
    hipLaunchKernelGGL(bn_finalize_sums_kernel, dim3((C + 3) / 4), dim3(256), 0, s, scratch, HH_BN_BLOCKS, C, sums, (float *)nullptr, (float *)nullptr);
    return hipGetLastError();
}
hipError_t launch_bn_train_normalize(const bf16_raw *x, int cs, size_t P, int C, const double *sums, double count, const float *gamma,
                                     const float *beta, float eps, const bf16_raw *res, int relu, bf16_raw *y, float *mean, float *invstd,
                                     hipStream_t s)
{
    hipLaunchKernelGGL(bn_stats_from_sums_kernel, dim3((C + 255) / 256), dim3(256), 0, s, sums, C, count, eps, mean, invstd);
    unsigned grid = (unsigned)((P * (C / 8) + 255) / 256);
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(bn_apply_kernel, dim3(grid), dim3(256), 0, s, x, cs, P, C, mean, invstd, gamma, beta, res, relu, y);
    return hipGetLastError();
}
hipError_t launch_bn_train_backward_stats(const bf16_raw *x, const bf16_raw *y, const bf16_raw *dy, int cs, size_t P, int C, const float *mean,
                                          const float *invstd, int relu, double *sums, float *dgamma, float *dbeta, double *scratch,
                                          hipStream_t s)
{
    hipLaunchKernelGGL(bn_bwd_partial_kernel<true>, dim3(HH_BN_BLOCKS), dim3(256), 0, s, x, y, dy, cs, P, C, mean, invstd, (const float *)nullptr, (const float *)nullptr, relu, scratch);
    // this rank's sums are also its dbeta / dgamma (the parameter gradients are averaged by DDP like every other one)
    hipLaunchKernelGGL(bn_finalize_sums_kernel, dim3((C + 3) / 4), dim3(256), 0, s, scratch, HH_BN_BLOCKS, C, sums, dbeta, dgamma);
    return hipGetLastError();
}
hipError_t launch_bn_train_backward_apply(const bf16_raw *x, const bf16_raw *y, const bf16_raw *dy, int cs, size_t P, int C, const float *mean,
                                          const float *invstd, const float *gamma, int relu, const double *sums, double count, bf16_raw *dx,
                                          bf16_raw *dres, double *scratch, hipStream_t s)
{
    float *ga = reinterpret_cast<float *>(scratch), *gb = ga + C;  // global sum g, sum g * xhat as floats
    hipLaunchKernelGGL(bn_sums_to_f32_kernel, dim3((C + 255) / 256), dim3(256), 0, s, sums, C, ga, gb);
    unsigned grid = (unsigned)((P * (C / 8) + 255) / 256);
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(bn_bwd_apply_kernel<true>, dim3(grid), dim3(256), 0, s, x, y, dy, cs, P, C, mean, invstd, gamma, (const float *)nullptr, gb, ga, relu, (float)(1.0 / count), dx, dres);
    return hipGetLastError();
}
