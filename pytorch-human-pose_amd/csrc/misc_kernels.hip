// HBM-bound helper kernels of the forward path: the fused "nearest-upsample + n-way sum + ReLU" of the
// exchange (fusion) layers, flip TTA, classifier tail, preprocessing.
#include "kernels.h"

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pack2(float a, float b)
{
    f32x2 f = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2));
}
__device__ __forceinline__ float lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float hi(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }

// FusionLayer.forward's low->high terms and the sum (hrnet.py:200-205,214-229): the 1x1
// conv + BN ran at low resolution; here the nearest upsample is an index shift on read and the
// upsampled tensors are never materialised.  One thread = 8 channels (16 B) of one pixel.
__global__ __launch_bounds__(256) void upadd_kernel(const UpAddParams p)
{
    const int c8n = p.C / 8;
    const size_t total = (size_t)p.B * p.H * p.W * c8n;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c8 = (int)(i % c8n);
        size_t pix = i / c8n;
        const int x = (int)(pix % p.W);
        const int y = (int)((pix / p.W) % p.H);
        const int b = (int)(pix / ((size_t)p.W * p.H));
        const uint4 bv = *reinterpret_cast<const uint4 *>(p.base + pix * p.base_cs + p.base_coff + c8 * 8);
        float v[8] = {lo(bv.x), hi(bv.x), lo(bv.y), hi(bv.y), lo(bv.z), hi(bv.z), lo(bv.w), hi(bv.w)};
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (j < p.nup) {
                const int sh = p.up_shift[j];
                const size_t up = ((size_t)b * (p.H >> sh) + (y >> sh)) * (p.W >> sh) + (x >> sh);
                const uint4 u = *reinterpret_cast<const uint4 *>(p.up[j] + up * p.up_cs[j] + c8 * 8);
                v[0] += lo(u.x); v[1] += hi(u.x); v[2] += lo(u.y); v[3] += hi(u.y);
                v[4] += lo(u.z); v[5] += hi(u.z); v[6] += lo(u.w); v[7] += hi(u.w);
            }
        if (p.relu)
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = fmaxf(v[k], 0.f);
        *reinterpret_cast<uint4 *>(p.out + pix * p.out_cs + p.out_coff + c8 * 8) =
            make_uint4(pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7]));
    }
}

hipError_t launch_upadd(const UpAddParams &p, hipStream_t s)
{
    const size_t total = (size_t)p.B * p.H * p.W * (p.C / 8);
    unsigned grid = (unsigned)((total + 255) / 256);
    if (grid > 16384) grid = 16384;
    hipLaunchKernelGGL(upadd_kernel, dim3(grid), dim3(256), 0, s, p);
    return hipGetLastError();
}

// ---- backward of the fusion sum (training): g = dy * (out > 0) at the output resolution (the gradient of every same-resolution
// term), and for a term that was nearest-upsampled by 2^s the sum of g over its 2^s x 2^s block (fp32 sum, one bf16 rounding).
__global__ __launch_bounds__(256) void upadd_mask_kernel(const bf16_raw *__restrict__ dy, const bf16_raw *__restrict__ out, bf16_raw *__restrict__ g, size_t n8)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
        const uint4 d = reinterpret_cast<const uint4 *>(dy)[i], o = reinterpret_cast<const uint4 *>(out)[i];
        auto m = [](unsigned dv, unsigned ov) {  // bf16 pairs: keep dy where out > 0 (out is a ReLU output: > 0 <=> bits != 0 and sign clear)
            const unsigned lo = ((ov & 0xffffu) != 0u && !(ov & 0x8000u)) ? 0xffffu : 0u, hi_ = ((ov >> 16) != 0u && !(ov & 0x80000000u)) ? 0xffff0000u : 0u;
            return dv & (lo | hi_);
        };
        reinterpret_cast<uint4 *>(g)[i] = make_uint4(m(d.x, o.x), m(d.y, o.y), m(d.z, o.z), m(d.w, o.w));
    }
}
__global__ __launch_bounds__(256) void upadd_blocksum_kernel(const bf16_raw *__restrict__ g, bf16_raw *__restrict__ dup, int B, int h, int w, int C, int sh)
{
    const int c8n = C / 8, W = w << sh;
    const size_t total = (size_t)B * h * w * c8n;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c8 = (int)(i % c8n);
        const size_t pix = i / c8n;
        const int x = (int)(pix % w), y = (int)((pix / w) % h), b = (int)(pix / ((size_t)w * h));
        float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int dy_ = 0; dy_ < (1 << sh); ++dy_)
            for (int dx_ = 0; dx_ < (1 << sh); ++dx_) {
                const size_t src = ((size_t)b * (h << sh) + ((y << sh) + dy_)) * W + ((x << sh) + dx_);
                const uint4 u = *reinterpret_cast<const uint4 *>(g + src * C + c8 * 8);
                v[0] += lo(u.x); v[1] += hi(u.x); v[2] += lo(u.y); v[3] += hi(u.y);
                v[4] += lo(u.z); v[5] += hi(u.z); v[6] += lo(u.w); v[7] += hi(u.w);
            }
        *reinterpret_cast<uint4 *>(dup + pix * C + c8 * 8) = make_uint4(pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7]));
    }
}
hipError_t launch_upadd_backward(const bf16_raw *dy, const bf16_raw *out, int relu, int B, int H, int W, int C, bf16_raw *g, bf16_raw *const *dup,
                                 const int *up_shift, int nup, hipStream_t s)
{
    const size_t n8 = (size_t)B * H * W * C / 8;
    const bf16_raw *gsrc = dy;
    if (relu) {
        unsigned grid = (unsigned)((n8 + 255) / 256);
        if (grid > 16384) grid = 16384;
        hipLaunchKernelGGL(upadd_mask_kernel, dim3(grid), dim3(256), 0, s, dy, out, g, n8);
        gsrc = g;
    }
    for (int j = 0; j < nup; ++j) {
        const int sh = up_shift[j], h = H >> sh, w = W >> sh;
        const size_t total = (size_t)B * h * w * (C / 8);
        unsigned grid = (unsigned)((total + 255) / 256);
        if (grid > 16384) grid = 16384;
        hipLaunchKernelGGL(upadd_blocksum_kernel, dim3(grid), dim3(256), 0, s, gsrc, dup[j], B, h, w, C, sh);
    }
    return hipGetLastError();
}

// ---- the same fusion sum on e4m3 tensors (fp8 path): every operand carries its tensor scale, the sum is formed in fp32 and
// requantised with the output scale.  One thread = 16 channels (16 B) of one pixel.
__device__ __forceinline__ void fp8x16_fma(const uint4 q, float s, float v[16])
{
    const unsigned w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const auto lo2 = __builtin_amdgcn_cvt_pk_f32_fp8((int)w[d], false), hi2 = __builtin_amdgcn_cvt_pk_f32_fp8((int)w[d], true);
        v[4 * d + 0] = __builtin_fmaf(lo2[0], s, v[4 * d + 0]); v[4 * d + 1] = __builtin_fmaf(lo2[1], s, v[4 * d + 1]);
        v[4 * d + 2] = __builtin_fmaf(hi2[0], s, v[4 * d + 2]); v[4 * d + 3] = __builtin_fmaf(hi2[1], s, v[4 * d + 3]);
    }
}
__device__ __forceinline__ void bf16x16_add(const bf16_raw *src, float v[16])
{
    const uint4 a = reinterpret_cast<const uint4 *>(src)[0], b = reinterpret_cast<const uint4 *>(src)[1];
    const unsigned w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (int d = 0; d < 8; ++d) {
        v[2 * d + 0] += __builtin_bit_cast(float, w[d] << 16);
        v[2 * d + 1] += __builtin_bit_cast(float, w[d] & 0xffff0000u);
    }
}
__device__ __forceinline__ unsigned pack_bf16_pair(float a, float b)
{
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
    f32x2_ f = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2_));
}
__global__ __launch_bounds__(256) void upadd_fp8_kernel(const UpAddFp8Params p)
{
    const int cgn = p.C / 16;
    const size_t total = (size_t)p.B * p.H * p.W * cgn;
    float amax = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int cg = (int)(i % cgn);
        size_t pix = i / cgn;
        const int x = (int)(pix % p.W);
        const int y = (int)((pix / p.W) % p.H);
        const int b = (int)(pix / ((size_t)p.W * p.H));
        float v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = 0.f;
        if (p.base16) bf16x16_add(p.base16 + pix * p.base16_cs + cg * 16, v);
        else fp8x16_fma(*reinterpret_cast<const uint4 *>(p.base + pix * p.base_cs + cg * 16), p.base_scale, v);
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (j < p.nup) {
                const int sh = p.up_shift[j];
                const size_t up = ((size_t)b * (p.H >> sh) + (y >> sh)) * (p.W >> sh) + (x >> sh);
                if (p.up16[j]) bf16x16_add(p.up16[j] + up * p.up16_cs[j] + cg * 16, v);
                else fp8x16_fma(*reinterpret_cast<const uint4 *>(p.up[j] + up * p.up_cs[j] + cg * 16), p.up_scale[j], v);
            }
        unsigned w[4], wb[8];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            float t[4], f4[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float f = v[4 * d + k];
                if (p.relu) f = fmaxf(f, 0.f);
                amax = fmaxf(amax, fabsf(f));
                f4[k] = f;
                t[k] = __builtin_amdgcn_fmed3f(f * p.out_inv_scale, -448.f, 448.f);
            }
            int q = 0;
            q = __builtin_amdgcn_cvt_pk_fp8_f32(t[0], t[1], q, false);
            q = __builtin_amdgcn_cvt_pk_fp8_f32(t[2], t[3], q, true);
            w[d] = (unsigned)q;
            wb[2 * d] = pack_bf16_pair(f4[0], f4[1]);
            wb[2 * d + 1] = pack_bf16_pair(f4[2], f4[3]);
        }
        if (p.out) *reinterpret_cast<uint4 *>(p.out + pix * p.out_cs + cg * 16) = make_uint4(w[0], w[1], w[2], w[3]);
        if (p.out16) {
            uint4 *dst = reinterpret_cast<uint4 *>(p.out16 + pix * p.out16_cs + cg * 16);
            dst[0] = make_uint4(wb[0], wb[1], wb[2], wb[3]);
            dst[1] = make_uint4(wb[4], wb[5], wb[6], wb[7]);
        }
    }
    if (p.absmax) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off));
        if ((threadIdx.x & 63) == 0 && amax > 0.f) atomicMax(p.absmax, __float_as_uint(amax));
    }
}
hipError_t launch_upadd_fp8(const UpAddFp8Params &p, hipStream_t s)
{
    const size_t total = (size_t)p.B * p.H * p.W * (p.C / 16);
    unsigned grid = (unsigned)((total + 255) / 256);
    if (grid > 16384) grid = 16384;
    hipLaunchKernelGGL(upadd_fp8_kernel, dim3(grid), dim3(256), 0, s, p);
    return hipGetLastError();
}

// bf16 representation of a tensor -> its e4m3 one (fp8 plans: a tensor written by a bf16-kernel op and read by an fp8 conv)
__global__ __launch_bounds__(256) void quant_fp8_kernel(const bf16_raw *__restrict__ in, int in_cs, unsigned char *__restrict__ out, int out_cs,
                                                        size_t npix, int cgn, float inv_scale, unsigned *absmax)
{
    const size_t total = npix * cgn;
    float amax = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int cg = (int)(i % cgn);
        const size_t pix = i / cgn;
        float v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = 0.f;
        bf16x16_add(in + pix * in_cs + cg * 16, v);
        unsigned w[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            float t[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                amax = fmaxf(amax, fabsf(v[4 * d + k]));
                t[k] = __builtin_amdgcn_fmed3f(v[4 * d + k] * inv_scale, -448.f, 448.f);
            }
            int q = 0;
            q = __builtin_amdgcn_cvt_pk_fp8_f32(t[0], t[1], q, false);
            q = __builtin_amdgcn_cvt_pk_fp8_f32(t[2], t[3], q, true);
            w[d] = (unsigned)q;
        }
        *reinterpret_cast<uint4 *>(out + pix * out_cs + cg * 16) = make_uint4(w[0], w[1], w[2], w[3]);
    }
    if (absmax) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off));
        if ((threadIdx.x & 63) == 0 && amax > 0.f) atomicMax(absmax, __float_as_uint(amax));
    }
}
hipError_t launch_quant_fp8(const bf16_raw *in, int in_cs, unsigned char *out, int out_cs, size_t npix, int C, float inv_scale, unsigned *absmax,
                            hipStream_t s)
{
    const size_t total = npix * (C / 16);
    unsigned grid = (unsigned)((total + 255) / 256);
    if (grid > 16384) grid = 16384;
    hipLaunchKernelGGL(quant_fp8_kernel, dim3(grid), dim3(256), 0, s, in, in_cs, out, out_cs, npix, C / 16, inv_scale, absmax);
    return hipGetLastError();
}

// keypoints/model.py:86: torch.flip(x, [3])
__global__ __launch_bounds__(256) void flip_images_kernel(const float *__restrict__ in, float *__restrict__ out, size_t rows,
                                                          int W)
{
    const size_t total = rows * W;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t r = i / W;
        const int x = (int)(i % W);
        out[i] = in[r * W + (W - 1 - x)];
    }
}

hipError_t launch_flip_images(const float *in, float *out, int B, int C, int H, int W, hipStream_t s)
{
    const size_t rows = (size_t)B * C * H;
    unsigned grid = (unsigned)((rows * W + 255) / 256);
    if (grid > 16384) grid = 16384;
    hipLaunchKernelGGL(flip_images_kernel, dim3(grid), dim3(256), 0, s, in, out, rows, W);
    return hipGetLastError();
}

// keypoints/model.py:87-93: heatmaps averaged with the un-flipped, joint-permuted second pass;
// tags of the second pass un-flipped and permuted.  (a + b) / 2 in fp32 as torch does.
__global__ __launch_bounds__(256) void flip_merge_kernel(float *hm, int64_t hm_bs, const float *hmf, int64_t hmf_bs,
                                                         const float *tf, int64_t tf_bs, float *to, int64_t to_bs,
                                                         const FlipPerm perm, int B, int K, int h, int w)
{
    const size_t plane = (size_t)h * w, total = (size_t)B * K * plane;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int x = (int)(i % w);
        const int y = (int)((i / w) % h);
        const int k = (int)((i / plane) % K);
        const int b = (int)(i / (plane * K));
        const size_t src = (size_t)perm.v[k] * plane + (size_t)y * w + (w - 1 - x);
        const size_t dst = (size_t)k * plane + (size_t)y * w + x;
        if (hm) hm[b * hm_bs + dst] = (hm[b * hm_bs + dst] + hmf[b * hmf_bs + src]) / 2.0f;
        if (to) to[b * to_bs + dst] = tf[b * tf_bs + src];
    }
}

hipError_t launch_flip_merge(float *hm, int64_t hm_bs, const float *hmf, int64_t hmf_bs, const float *tf, int64_t tf_bs,
                             float *to, int64_t to_bs, const int32_t *perm_host, int B, int K, int h, int w, hipStream_t s)
{
    FlipPerm perm;  // K <= 64 joints: the permutation travels in the kernel arguments, no device buffer to share between streams
    for (int k = 0; k < 64; ++k) perm.v[k] = (unsigned char)(k < K ? perm_host[k] : 0);
    const size_t total = (size_t)B * K * h * w;
    unsigned grid = (unsigned)((total + 255) / 256);
    if (grid > 16384) grid = 16384;
    hipLaunchKernelGGL(flip_merge_kernel, dim3(grid), dim3(256), 0, s, hm, hm_bs, hmf, hmf_bs, tf, tf_bs, to, to_bs, perm,
                       B, K, h, w);
    return hipGetLastError();
}

// F.avg_pool2d over the whole map (classification/architectures/hrnet.py:57): one thread = 8 channels of one image
__global__ __launch_bounds__(256) void avgpool_kernel(const bf16_raw *__restrict__ in, int in_cs, float *__restrict__ out, int B,
                                                      int HW, int C)
{
    const int c8n = C / 8, i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * c8n) return;
    const int b = i / c8n, c8 = i % c8n;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int p = 0; p < HW; ++p) {
        const uint4 v = *reinterpret_cast<const uint4 *>(in + ((size_t)b * HW + p) * in_cs + c8 * 8);
        acc[0] += lo(v.x); acc[1] += hi(v.x); acc[2] += lo(v.y); acc[3] += hi(v.y);
        acc[4] += lo(v.z); acc[5] += hi(v.z); acc[6] += lo(v.w); acc[7] += hi(v.w);
    }
    for (int k = 0; k < 8; ++k) out[(size_t)b * C + c8 * 8 + k] = acc[k] / (float)HW;
}
hipError_t launch_avgpool(const bf16_raw *in, int in_cs, float *out, int B, int HW, int C, hipStream_t s)
{
    hipLaunchKernelGGL(avgpool_kernel, dim3((B * (C / 8) + 255) / 256), dim3(256), 0, s, in, in_cs, out, B, HW, C);
    return hipGetLastError();
}

// nn.Linear (classification/architectures/hrnet.py:46,60): one wave per output feature, fp32
__global__ __launch_bounds__(256) void linear_kernel(const float *__restrict__ x, const float *__restrict__ w,
                                                     const float *__restrict__ bias, float *__restrict__ y, int K, int N)
{
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6), b = blockIdx.y, lane = threadIdx.x & 63;
    if (n >= N) return;
    float acc = 0.f;
    for (int k = lane; k < K; k += 64) acc += x[(size_t)b * K + k] * w[(size_t)n * K + k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if (lane == 0) y[(size_t)b * N + n] = acc + bias[n];
}
hipError_t launch_linear(const float *x, const float *w, const float *bias, float *y, int B, int K, int N, hipStream_t s)
{
    hipLaunchKernelGGL(linear_kernel, dim3((N + 3) / 4, B), dim3(256), 0, s, x, w, bias, y, K, N);
    return hipGetLastError();
}

// InferenceKeypointsModel.prepare_input on the GPU (keypoints/model.py:70-76 + base/transforms/utils.py:89-97): the resize-align
// warp as cv2.warpAffine computes it for 8-bit images (OpenCV 4.9 imgwarp.cpp, INTER_LINEAR, BORDER_CONSTANT 0; restated
// independently in oracle/transforms.py), then ToTensor (/255) and Normalize -> fp32 NCHW.
//   `inv` is the destination -> source matrix cv::warpAffine builds for itself (hh_invert_affine).  Source coordinates are fixed
//   point with AB_BITS = 10: adelta = round(M0*x*1024), X0 = round((M1*y + M2)*1024) + 16, X = (X0 + adelta) >> 5; pixel = X >> 5,
//   fraction = X & 31 (INTER_BITS = 5); weights = the int16 table entries (32-fy)(32-fx)*32 ... that sum to 32768 (entry (0,0):
//   32767 + 1 on the bottom-right tap, as the table builder leaves it); value = (sum + 16384) >> 15; taps outside the image are 0.
//   Every double product / sum is rounded on its own (no contraction), round = nearest-even as cvRound.
__device__ __forceinline__ void warp_pixel_u8(const unsigned char *__restrict__ img, int h, int w, const double inv[6], int x, int y, int out[3])
{
    auto sat_i = [](double v) -> int { return v >= 2147483647.0 ? 2147483647 : v <= -2147483648.0 ? (int)(-2147483647 - 1) : (int)v; };
    const int adelta = sat_i(rint(__dmul_rn(__dmul_rn(inv[0], (double)x), 1024.0)));
    const int bdelta = sat_i(rint(__dmul_rn(__dmul_rn(inv[3], (double)x), 1024.0)));
    const int X0 = sat_i(rint(__dmul_rn(__dadd_rn(__dmul_rn(inv[1], (double)y), inv[2]), 1024.0))) + 16;
    const int Y0 = sat_i(rint(__dmul_rn(__dadd_rn(__dmul_rn(inv[4], (double)y), inv[5]), 1024.0))) + 16;
    const int X = (X0 + adelta) >> 5, Y = (Y0 + bdelta) >> 5;
    const int sx = min(max(X >> 5, -32768), 32767), sy = min(max(Y >> 5, -32768), 32767);
    const int fx = X & 31, fy = Y & 31;
    int w00 = (32 - fy) * (32 - fx) * 32, w01 = (32 - fy) * fx * 32, w10 = fy * (32 - fx) * 32, w11 = fy * fx * 32;
    if ((fx | fy) == 0) { w00 = 32767; w11 = 1; }
    const bool y0 = sy >= 0 && sy < h, y1 = sy + 1 >= 0 && sy + 1 < h, x0 = sx >= 0 && sx < w, x1 = sx + 1 >= 0 && sx + 1 < w;
    const unsigned char *r0 = img + ((size_t)(y0 ? sy : 0) * w) * 3, *r1 = img + ((size_t)(y1 ? sy + 1 : 0) * w) * 3;
    const int c0 = (x0 ? sx : 0) * 3, c1 = (x1 ? sx + 1 : 0) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int p00 = (y0 && x0) ? r0[c0 + c] : 0, p01 = (y0 && x1) ? r0[c1 + c] : 0;
        const int p10 = (y1 && x0) ? r1[c0 + c] : 0, p11 = (y1 && x1) ? r1[c1 + c] : 0;
        out[c] = (p00 * w00 + p01 * w01 + p10 * w10 + p11 * w11 + 16384) >> 15;
    }
}
__device__ __forceinline__ void preprocess_pixel(const unsigned char *__restrict__ img, int h, int w, const double inv[6],
                                                 float *__restrict__ out, int H, int W, int i, const float mean[3], const float stdv[3])
{
    int v[3];
    warp_pixel_u8(img, h, w, inv, i % W, i / W, v);
#pragma unroll
    for (int c = 0; c < 3; ++c) out[(size_t)c * H * W + i] = ((float)v[c] / 255.0f - mean[c]) / stdv[c];
}
// resize_align_multi_scale's image itself (uint8 HWC), for callers that want the warped pixels (base/transforms/utils.py:89-97)
__global__ __launch_bounds__(256) void warp_affine_u8_kernel(const unsigned char *__restrict__ img, int h, int w, double i00, double i01,
                                                             double i02, double i10, double i11, double i12, unsigned char *__restrict__ out,
                                                             int H, int W)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= H * W) return;
    const double inv[6] = {i00, i01, i02, i10, i11, i12};
    int v[3];
    warp_pixel_u8(img, h, w, inv, i % W, i / W, v);
#pragma unroll
    for (int c = 0; c < 3; ++c) out[(size_t)i * 3 + c] = (unsigned char)v[c];
}
hipError_t launch_warp_affine_u8(const unsigned char *img, int h, int w, const double inv[6], unsigned char *out, int H, int W, hipStream_t s)
{
    hipLaunchKernelGGL(warp_affine_u8_kernel, dim3((H * W + 255) / 256), dim3(256), 0, s, img, h, w, inv[0], inv[1], inv[2], inv[3], inv[4],
                       inv[5], out, H, W);
    return hipGetLastError();
}
__global__ __launch_bounds__(256) void preprocess_kernel(const unsigned char *__restrict__ img, int h, int w, double i00, double i01,
                                                         double i02, double i10, double i11, double i12, float *__restrict__ out,
                                                         int H, int W, float m0, float m1, float m2, float s0, float s1, float s2)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= H * W) return;
    const double inv[6] = {i00, i01, i02, i10, i11, i12};
    const float mean[3] = {m0, m1, m2}, stdv[3] = {s0, s1, s2};
    preprocess_pixel(img, h, w, inv, out, H, W, i, mean, stdv);
}
// a batch of raw images of any sizes into one [n,3,H,W] tensor: blockIdx.y = image, its descriptor read from device memory
__global__ __launch_bounds__(256) void preprocess_batch_kernel(const unsigned char *__restrict__ base, const HHImageDesc *__restrict__ descs,
                                                               float *__restrict__ out, int H, int W, float m0, float m1, float m2,
                                                               float s0, float s1, float s2)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= H * W) return;
    const HHImageDesc d = descs[blockIdx.y];
    const float mean[3] = {m0, m1, m2}, stdv[3] = {s0, s1, s2};
    preprocess_pixel(base + d.offset, d.h, d.w, d.inv, out + (size_t)blockIdx.y * 3 * H * W, H, W, i, mean, stdv);
}
hipError_t launch_preprocess_batch(const unsigned char *base, const HHImageDesc *descs, int n, float *out, int H, int W,
                                   const float mean[3], const float stdv[3], hipStream_t s)
{
    hipLaunchKernelGGL(preprocess_batch_kernel, dim3((H * W + 255) / 256, n), dim3(256), 0, s, base, descs, out, H, W, mean[0], mean[1],
                       mean[2], stdv[0], stdv[1], stdv[2]);
    return hipGetLastError();
}
hipError_t launch_preprocess(const unsigned char *img, int h, int w, const double inv[6], float *out, int H, int W,
                             const float mean[3], const float stdv[3], hipStream_t s)
{
    hipLaunchKernelGGL(preprocess_kernel, dim3((H * W + 255) / 256), dim3(256), 0, s, img, h, w, inv[0], inv[1], inv[2], inv[3], inv[4],
                       inv[5], out, H, W, mean[0], mean[1], mean[2], stdv[0], stdv[1], stdv[2]);
    return hipGetLastError();
}


// HH_POISON_LDS=1 (tests, hh_net::enqueue): LDS is not cleared between kernels -- a workgroup finds what the last workgroup on its CU
// left there.  This kernel leaves bf16 / fp32 NaN patterns (0xFF bytes) in all 160 KB of every CU's LDS, so that a kernel whose
// result depends on LDS bytes it never wrote (a halo nobody staged, padding taps multiplied by zero weights) shows NaNs in its
// output instead of depending on which kernels ran before it.  One workgroup needs a whole CU's LDS, so workgroups only land on
// CUs whose LDS is free; 8 per CU, each lingering ~2 us, so that the dispatcher reaches every CU.
__global__ __launch_bounds__(256) void lds_poison_kernel(unsigned *sink)
{
    extern __shared__ uint4 lds_all[];
    const uint4 nan4 = make_uint4(0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu);
    for (int i = threadIdx.x; i < 160 * 1024 / 16; i += 256) lds_all[i] = nan4;
    __syncthreads();
    __builtin_amdgcn_s_sleep(127);
    __builtin_amdgcn_s_sleep(127);
    if (lds_all[threadIdx.x].x != 0xffffffffu) *sink = 1;  // (keeps the stores alive)
}
hipError_t launch_lds_poison(int num_cus, hipStream_t s)
{
    static unsigned *sink[64] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (!sink[dev & 63]) {
        if ((e = hipFuncSetAttribute((const void *)lds_poison_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
        if ((e = hipMalloc((void **)&sink[dev & 63], 64)) != hipSuccess) return e;
    }
    hipLaunchKernelGGL(lds_poison_kernel, dim3(8 * num_cus), dim3(256), 160 * 1024, s, sink[dev & 63]);
    return hipGetLastError();
}
