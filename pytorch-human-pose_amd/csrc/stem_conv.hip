// First stem convolution (/root/reference/src/keypoints/architectures/hrnet.py:354-356,379-381):
//   conv3x3 stride 2 pad 1 (3 -> 64) + BN + ReLU, reading the fp32 NCHW images directly.
// Why a dedicated kernel: with 3 input channels the layer is pure bandwidth (100 MB of fp32 in, 268 MB of bf16 NHWC out
// at B=32, 512x512).  Through the generic path it cost a layout pass (NCHW fp32 -> NHWC16 bf16, 368 MB) plus a conv
// that staged 16-channel pixels of which 13 were padding (536 MB, 334 us); here the image is read once.
// Mapping: K = 27 taps (c, ky, kx) padded to 32 = two 16-wide MFMA k-steps; each lane owns one output pixel and gathers
// its taps from a bf16 LDS patch into the B-operand layout; A = the 64 x 32 weight matrix (BN scale folded) in registers.
#include "kernels.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef short i16x2 __attribute__((ext_vector_type(2)));

namespace {
constexpr int TH = 8, TW = 32;                  // output tile
constexpr int PH = 2 * TH + 1, PW = 2 * TW + 1;  // 17 x 65 input patch per channel
constexpr int PLANE = PH * PW;                  // 1105
constexpr int NVAL = 3 * PLANE;                 // 3315 values (+1 zero slot for the padded taps)

__device__ __forceinline__ unsigned pack_relu_bf16x2(float a, float b)
{
    f32x2 f = {a, b};
    const i16x2 v = __builtin_bit_cast(i16x2, __builtin_convertvector(f, bf16x2));
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(v, i16x2{0, 0}));
}
}  // namespace

__global__ __launch_bounds__(256) void stem_conv_kernel(const StemParams p)
{
    __shared__ unsigned short patch[NVAL + 1];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
#ifndef HH_NO_CLK
    // 8192 workgroups hammering one address would dominate this short kernel: only the first and the last 256 dispatched
    // workgroups stamp (dispatch order follows blockIdx closely enough for a start/end probe)
    if (p.clk && tid == 0 && blockIdx.x < 256) atomicMin(p.clk, wall_clock64());
#endif
    const int Ho = p.H >> 1, Wo = p.W >> 1;
    const int tiles_x = (Wo + TW - 1) / TW, tiles_y = (Ho + TH - 1) / TH;
    int bid = blockIdx.x;
    const int tx = bid % tiles_x; bid /= tiles_x;
    const int ty = bid % tiles_y;
    const int b = bid / tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW, iy0 = 2 * oy0 - 1, ix0 = 2 * ox0 - 1;

    // ---- weights: A fragments [cout tile][k-step] of this lane (rows = couts, 8 consecutive taps per lane), bias
    u32x4 a[2][2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) a[ct][kk] = reinterpret_cast<const u32x4 *>(p.w)[((ct * 2 + kk) * 2 + h) * 32 + r];

    // ---- patch: fp32 NCHW -> bf16 LDS [c][row][col], zero outside the image (conv padding)
    const float *img = p.images + (size_t)b * 3 * p.H * p.W;
    constexpr int NLD = (NVAL + 255) / 256;  // 13 loads per thread, all issued before the first one is touched
    float v[NLD];
    unsigned okmask = 0;  // the loads are not touched (nor predicated) here: zero padding is applied when they go to LDS
#pragma unroll
    for (int it = 0; it < NLD; ++it) {
        const int i = tid + 256 * it;
        const int c = i / PLANE, rem = i % PLANE, py = rem / PW, px = rem % PW;
        const int iy = iy0 + py, ix = ix0 + px;
        const bool ok = (i < NVAL) & ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
        const int o = ok ? (c * p.H + iy) * p.W + ix : 0;
        v[it] = img[o];
        okmask |= ok ? (1u << it) : 0u;
    }
#pragma unroll
    for (int it = 0; it < NLD; ++it) {
        const int i = tid + 256 * it;
        const float x = (okmask >> it) & 1u ? v[it] : 0.f;
        patch[i < NVAL ? i : NVAL] = __builtin_bit_cast(unsigned short, (__bf16)x);  // idle slots of the last round hit the zero slot
    }
    // tap t = kk*16 + 8h + j  ->  (c, ky, kx) = (t / 9, (t % 9) / 3, t % 3); taps >= 27 read the zero slot
    int off[2][8];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int t = kk * 16 + 8 * h + j;
            off[kk][j] = t < 27 ? (t / 9) * PLANE + ((t % 9) / 3) * PW + (t % 3) : -1;
        }
    float4 bv[2][4];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int g = 0; g < 4; ++g) bv[ct][g] = *reinterpret_cast<const float4 *>(p.bias + ct * 32 + 8 * g + 4 * h);
    __syncthreads();

    float amax = 0.f;  // fp8 calibration: max y of this lane
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int row = wave * 2 + q;  // output row inside the tile
        const int base = (2 * row) * PW + 2 * r;
        u32x4 bf[2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            unsigned short v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = patch[off[kk][j] >= 0 ? base + off[kk][j] : NVAL];
            bf[kk] = u32x4{(unsigned)v[0] | ((unsigned)v[1] << 16), (unsigned)v[2] | ((unsigned)v[3] << 16),
                           (unsigned)v[4] | ((unsigned)v[5] << 16), (unsigned)v[6] | ((unsigned)v[7] << 16)};
        }
        const int oy = oy0 + row, ox = ox0 + r;
        const bool valid = (oy < Ho) & (ox < Wo);
        bf16_raw *dst = p.out + (((size_t)b * Ho + (valid ? oy : 0)) * Wo + (valid ? ox : 0)) * p.out_cs + 8 * h;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            f32x16 acc;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                acc[4 * g + 0] = bv[ct][g].x; acc[4 * g + 1] = bv[ct][g].y; acc[4 * g + 2] = bv[ct][g].z; acc[4 * g + 3] = bv[ct][g].w;
            }
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[ct][kk]), __builtin_bit_cast(bf16x8, bf[kk]), acc, 0, 0, 0);
            if (p.out_fp8) {  // fp8 path: ReLU, scale, e4m3; two half-wave exchanges give every lane 16 contiguous couts (16h..)
                unsigned x[2], z[2];
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    float t[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float f = fmaxf(acc[8 * m + i], 0.f);
                        amax = fmaxf(amax, valid ? f : 0.f);
                        t[i] = fminf(f * p.out_inv_scale, 448.f);
                    }
                    int q = 0, q2 = 0;
                    q = __builtin_amdgcn_cvt_pk_fp8_f32(t[0], t[1], q, false); q = __builtin_amdgcn_cvt_pk_fp8_f32(t[2], t[3], q, true);
                    q2 = __builtin_amdgcn_cvt_pk_fp8_f32(t[4], t[5], q2, false); q2 = __builtin_amdgcn_cvt_pk_fp8_f32(t[6], t[7], q2, true);
                    x[m] = (unsigned)q; z[m] = (unsigned)q2;
                }
                auto s0 = __builtin_amdgcn_permlane32_swap(x[0], z[0], false, false);
                auto s1 = __builtin_amdgcn_permlane32_swap(x[1], z[1], false, false);
                auto a0 = __builtin_amdgcn_permlane32_swap(s0[0], s1[0], false, false);
                auto a1 = __builtin_amdgcn_permlane32_swap(s0[1], s1[1], false, false);
                unsigned char *d8 = p.out_fp8 + (((size_t)b * Ho + (valid ? oy : 0)) * Wo + (valid ? ox : 0)) * p.out_cs + ct * 32 + 16 * h;
                if (valid) *reinterpret_cast<u32x4 *>(d8) = u32x4{a0[0], a1[0], a0[1], a1[1]};
                continue;
            }
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const unsigned x0 = pack_relu_bf16x2(acc[8 * m + 0], acc[8 * m + 1]), x1 = pack_relu_bf16x2(acc[8 * m + 2], acc[8 * m + 3]);
                const unsigned y0 = pack_relu_bf16x2(acc[8 * m + 4], acc[8 * m + 5]), y1 = pack_relu_bf16x2(acc[8 * m + 6], acc[8 * m + 7]);
                auto s0 = __builtin_amdgcn_permlane32_swap(x0, y0, false, false);
                auto s1 = __builtin_amdgcn_permlane32_swap(x1, y1, false, false);
                if (valid) *reinterpret_cast<u32x4 *>(dst + ct * 32 + m * 16) = u32x4{s0[0], s1[0], s0[1], s1[1]};
            }
        }
    }
    if (p.absmax) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off));
        if (lane == 0 && amax > 0.f) atomicMax(p.absmax, __float_as_uint(amax));
    }
#ifndef HH_NO_CLK
    if (p.clk && tid == 0 && blockIdx.x + 256 >= gridDim.x) atomicMax(p.clk + 1, wall_clock64());
#endif
}

hipError_t stem_conv_launch(const StemParams &p, hipStream_t s)
{
    const int Ho = p.H >> 1, Wo = p.W >> 1;
    const int grid = p.B * ((Ho + TH - 1) / TH) * ((Wo + TW - 1) / TW);
    HH_LAUNCH(stem_conv_kernel, dim3(grid), dim3(256), 0, s, p);
    return hipGetLastError();
}
