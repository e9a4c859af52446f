// Junction between two stage-0 Bottlenecks (/root/reference/src/keypoints/architectures/hrnet.py:29-74):
//   y  = relu(bn3(conv3_1x1(t2)) + residual)          residual = previous y (256 ch), or bn_d(downsample_1x1(x)) for unit 0
//   t1 = relu(bn1'(conv1'_1x1(y)))                    first conv of the NEXT unit (optional)
// in ONE pass over the pixels.  Both convs are 1x1, so a pixel's 256 outputs never leave the wave that made them: the
// packed bf16 accumulators of the first GEMM (what goes to HBM as y) ARE the B fragments of the second -- the C layout
// after v_permlane32_swap is 8 consecutive channels per lane, exactly one 16-channel k-step per (cout tile, half).
// Why: stage 0 moves 256-channel tensors at 128x128 and is HBM bound; unfused, y is written by conv3 and read again by
// the next conv1 (268 MB at B=32, three times), and unit 0 round-trips the downsample result (2 x 268 MB).
// Pixel fragments of t2 / x are loaded from HBM straight into the MFMA B layout (16 B per lane), weights sit in LDS.
#include "kernels.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef short i16x2 __attribute__((ext_vector_type(2)));

namespace {
__device__ __forceinline__ unsigned pack_relu_bf16x2(float a, float b)
{
    f32x2 f = {a, b};
    const i16x2 v = __builtin_bit_cast(i16x2, __builtin_convertvector(f, bf16x2));
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(v, i16x2{0, 0}));  // ReLU on the packed pair
}
__device__ __forceinline__ float bf16_lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float bf16_hi(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }

// acc (32 couts of this lane's pixel, MFMA C layout) -> for m = 0,1 the 16 bytes of couts 16m+8h..+7 (ReLU, bf16)
__device__ __forceinline__ void pack_rows16(const f32x16 &acc, u32x4 out[2])
{
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        unsigned x0 = pack_relu_bf16x2(acc[8 * m + 0], acc[8 * m + 1]), x1 = pack_relu_bf16x2(acc[8 * m + 2], acc[8 * m + 3]);
        unsigned y0 = pack_relu_bf16x2(acc[8 * m + 4], acc[8 * m + 5]), y1 = pack_relu_bf16x2(acc[8 * m + 6], acc[8 * m + 7]);
        auto s0 = __builtin_amdgcn_permlane32_swap(x0, y0, false, false);
        auto s1 = __builtin_amdgcn_permlane32_swap(x1, y1, false, false);
        out[m] = u32x4{s0[0], s1[0], s0[1], s1[1]};
    }
}
// inverse for the residual: 16 bytes (couts 16m+8h..+7) per m -> added onto the accumulator in C layout
__device__ __forceinline__ void add_rows16(f32x16 &acc, const u32x4 v[2])
{
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        auto s0 = __builtin_amdgcn_permlane32_swap(v[m][0], v[m][2], false, false);
        auto s1 = __builtin_amdgcn_permlane32_swap(v[m][1], v[m][3], false, false);
        acc[8 * m + 0] += bf16_lo(s0[0]); acc[8 * m + 1] += bf16_hi(s0[0]);
        acc[8 * m + 2] += bf16_lo(s1[0]); acc[8 * m + 3] += bf16_hi(s1[0]);
        acc[8 * m + 4] += bf16_lo(s0[1]); acc[8 * m + 5] += bf16_hi(s0[1]);
        acc[8 * m + 6] += bf16_lo(s1[1]); acc[8 * m + 7] += bf16_hi(s1[1]);
    }
}
#ifndef JUNC_PREFETCH
#define JUNC_PREFETCH 1
#endif
constexpr int W3_BYTES = 256 * 64 * 2;  // conv3 / downsample: packed [cout group 4][chunk 2][c8 4][64][8] (conv_mfma family, NT = 2)
constexpr int W1_BYTES = 64 * 256 * 2;  // next conv1: packed [chunk 8][c8 4][64][8]
}  // namespace

// MODE 0: residual = p.res (256 ch).   MODE 1: unit 0 -- residual = downsample conv of x, folded in as 4 more k-steps.
// MODE 2 / 3 ("pair"): the y of the PREVIOUS unit is not read from HBM but made again, per pixel, from that unit's t2 (p.t2a,
// weights p.w3a) -- MODE 2: the previous unit is unit 0 (+ downsample of x), MODE 3: a plain unit on top of p.res -- and used,
// rounded to bf16 exactly as it would have been stored, as the residual of this unit.  The previous junction then does not
// store its y at all (p.y == nullptr there): stage 0 moves a quarter less through HBM for two more 1x1 GEMMs per pair.
// Workgroups: eight waves per CU where the registers allow it -- MODE 0 (64 KB of weights) and MODE 3 without a next conv1 (64 KB)
// as two workgroups of 256 threads, MODE 1 (96 KB) as one of 512 (81 -> 58 us); MODE 2 (128 KB, 241 registers) and MODE 3 with a
// next conv1 stay at one workgroup of 256 threads: with 512 they spill (149 -> 185 us).
template <int MODE>
__global__ __launch_bounds__(MODE == 1 ? 512 : 256, (MODE == 0 || MODE == 3) ? 2 : 1) void junction_kernel(const JuncParams p)
{
    constexpr int NT = MODE == 1 ? 512 : 256;
    constexpr bool HAS_DS = MODE == 1 || MODE == 2, PAIR = MODE >= 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *lds_w3 = smem;                                           // this unit's conv3
    char *lds_wd = smem + W3_BYTES;                                // only when HAS_DS
    char *lds_w3a = smem + (HAS_DS ? 2 : 1) * W3_BYTES;            // only when PAIR: the previous unit's conv3
    char *lds_w1 = smem + ((HAS_DS ? 2 : 1) + (PAIR ? 1 : 0)) * W3_BYTES;
    float *lds_b = reinterpret_cast<float *>(lds_w1 + (p.w1 ? W1_BYTES : 0));  // [256] y shift, [64] t1 shift, [256] previous y shift (PAIR)

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
#ifndef HH_NO_CLK
    if (p.clk && tid == 0) atomicMin(p.clk, wall_clock64());
#endif
    for (int u = tid; u < W3_BYTES / 16; u += NT) {
        reinterpret_cast<u32x4 *>(lds_w3)[u] = reinterpret_cast<const u32x4 *>(p.w3)[u];
        if (HAS_DS) reinterpret_cast<u32x4 *>(lds_wd)[u] = reinterpret_cast<const u32x4 *>(p.wd)[u];
        if (PAIR) reinterpret_cast<u32x4 *>(lds_w3a)[u] = reinterpret_cast<const u32x4 *>(p.w3a)[u];
        if (p.w1) reinterpret_cast<u32x4 *>(lds_w1)[u] = reinterpret_cast<const u32x4 *>(p.w1)[u];
    }
    // the downsample shift belongs to the unit the downsample conv belongs to: this one (MODE 1) or the previous one (MODE 2)
    if (tid < 256) {
        lds_b[tid] = p.b3[tid] + (MODE == 1 ? p.bd[tid] : 0.f);
        if (tid < 64) lds_b[256 + tid] = p.w1 ? p.b1[tid] : 0.f;
        if (PAIR) lds_b[320 + tid] = p.b3a[tid] + (MODE == 2 ? p.bd[tid] : 0.f);
    }
    __syncthreads();

    constexpr int GP = NT / 2;  // pixels per workgroup step: 32 per wave
    const int ngroups = (p.npix + GP - 1) / GP;
    // pixel fragments (B operands) of a group.  MODE 2 loads them one group AHEAD (round 4): its workgroup is alone on the CU, one wave per
    // SIMD, and every group began with a full HBM round trip in front of its first MFMA that nobody covered: 152 -> 125 us.  (The
    // other modes run two waves per SIMD, which cover each other: 1-2 us SLOWER with the 16-48 more registers, MODE 3 spills.)
    constexpr bool PF = JUNC_PREFETCH && MODE == 2;
    struct Frags { u32x4 bt[4], bx[HAS_DS ? 4 : 1], bta[PAIR ? 4 : 1]; };
    auto load_frags = [&](int g, Frags &f) {
        const int pixn = g * GP + wave * 32 + r;
        const size_t px = pixn < p.npix ? pixn : 0;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            f.bt[s] = *reinterpret_cast<const u32x4 *>(p.t2 + px * p.t2_cs + s * 16 + h * 8);
            if (HAS_DS) f.bx[s] = *reinterpret_cast<const u32x4 *>(p.x + px * p.x_cs + s * 16 + h * 8);
            if (PAIR) f.bta[s] = *reinterpret_cast<const u32x4 *>(p.t2a + px * p.t2a_cs + s * 16 + h * 8);
        }
    };
    Frags fnext;
    if (PF && (int)blockIdx.x < ngroups) load_frags(blockIdx.x, fnext);
    for (int g = blockIdx.x; g < ngroups; g += gridDim.x) {
        const int pix = g * GP + wave * 32 + r;
        const bool valid = pix < p.npix;
        const size_t px = valid ? pix : 0;
        Frags fcur;
        if (PF) fcur = fnext;
        else load_frags(g, fcur);
        const u32x4 (&bt)[4] = fcur.bt;
        const u32x4 (&bx)[HAS_DS ? 4 : 1] = fcur.bx;
        const u32x4 (&bta)[PAIR ? 4 : 1] = fcur.bta;
        {
            const int gn = g + (int)gridDim.x;
            if (PF && gn < ngroups) load_frags(gn, fnext);
        }
        // ---- GEMM 1 in two halves of 128 output channels (64 accumulator registers live at a time):
        //      y[256] = W3 t2 (+ Wd x) + shift (+ residual), ReLU, bf16; y leaves for HBM and stays in registers (yf) as the
        //      B operand of the next GEMM
        u32x4 yf[8][2];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            u32x4 rv[4][2];  // the residual of this unit: from memory (MODE 0), or the previous unit's y made here (PAIR)
            if (MODE == 0 || MODE == 3)
#pragma unroll
                for (int mm = 0; mm < 4; ++mm)
#pragma unroll
                    for (int q = 0; q < 2; ++q)
                        rv[mm][q] = *reinterpret_cast<const u32x4 *>(p.res + px * p.res_cs + (half * 4 + mm) * 32 + q * 16 + h * 8);
            f32x16 acc[4];
            auto init_acc = [&](int boff) {
#pragma unroll
                for (int mm = 0; mm < 4; ++mm)
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        const float4 bv = *reinterpret_cast<const float4 *>(lds_b + boff + (half * 4 + mm) * 32 + 8 * gq + 4 * h);
                        acc[mm][4 * gq + 0] = bv.x; acc[mm][4 * gq + 1] = bv.y; acc[mm][4 * gq + 2] = bv.z; acc[mm][4 * gq + 3] = bv.w;
                    }
            };
            if (PAIR) {  // the previous unit's y for these 128 channels: W3a t2a (+ Wd x) + shift (+ its residual), ReLU, bf16
                init_acc(320);
#pragma unroll
                for (int s = 0; s < 4; ++s) {
#pragma unroll
                    for (int mm = 0; mm < 4; ++mm) {
                        const int m = half * 4 + mm;
                        const int unit = (((m >> 1) * 2 + (s >> 1)) * 4 + (s & 1) * 2 + h) * 64 + (m & 1) * 32 + r;
                        const u32x4 a = *reinterpret_cast<const u32x4 *>(lds_w3a + unit * 16);
                        acc[mm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, bta[s]), acc[mm], 0, 0, 0);
                        if (MODE == 2) {
                            const u32x4 ad = *reinterpret_cast<const u32x4 *>(lds_wd + unit * 16);
                            acc[mm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ad), __builtin_bit_cast(bf16x8, bx[s]), acc[mm], 0, 0, 0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int mm = 0; mm < 4; ++mm) {
                    if (MODE == 3) add_rows16(acc[mm], rv[mm]);
                    pack_rows16(acc[mm], rv[mm]);  // (rv now holds what the previous junction would have stored)
                }
            }
            init_acc(0);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int mm = 0; mm < 4; ++mm) {
                    const int m = half * 4 + mm;
                    const int unit = (((m >> 1) * 2 + (s >> 1)) * 4 + (s & 1) * 2 + h) * 64 + (m & 1) * 32 + r;
                    const u32x4 a = *reinterpret_cast<const u32x4 *>(lds_w3 + unit * 16);
                    acc[mm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, bt[s]), acc[mm], 0, 0, 0);
                    if (MODE == 1) {
                        const u32x4 ad = *reinterpret_cast<const u32x4 *>(lds_wd + unit * 16);
                        acc[mm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ad), __builtin_bit_cast(bf16x8, bx[s]), acc[mm], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);  // keep the weight-fragment reads of later k-steps from piling up in registers
            }
#pragma unroll
            for (int mm = 0; mm < 4; ++mm) {
                const int m = half * 4 + mm;
                if (MODE != 1) add_rows16(acc[mm], rv[mm]);
                pack_rows16(acc[mm], yf[m]);
                if (valid && p.y) {
                    bf16_raw *dst = p.y + (size_t)pix * p.y_cs + m * 32 + h * 8;
                    *reinterpret_cast<u32x4 *>(dst) = yf[m][0];
                    *reinterpret_cast<u32x4 *>(dst + 16) = yf[m][1];
                }
            }
        }
        if (!p.w1) continue;
        // ---- GEMM 2: t1[64] = W1' y + shift, ReLU
        f32x16 acc2[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const float4 bv = *reinterpret_cast<const float4 *>(lds_b + 256 + t * 32 + 8 * gq + 4 * h);
                acc2[t][4 * gq + 0] = bv.x; acc2[t][4 * gq + 1] = bv.y; acc2[t][4 * gq + 2] = bv.z; acc2[t][4 * gq + 3] = bv.w;
            }
#pragma unroll
        for (int s = 0; s < 16; ++s) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int unit = ((s >> 1) * 4 + (s & 1) * 2 + h) * 64 + t * 32 + r;
                const u32x4 a = *reinterpret_cast<const u32x4 *>(lds_w1 + unit * 16);
                acc2[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, yf[s >> 1][s & 1]),
                                                                 acc2[t], 0, 0, 0);
            }
            if (s & 1) __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            u32x4 o[2];
            pack_rows16(acc2[t], o);
            if (valid) {
                bf16_raw *dst = p.t1 + (size_t)pix * p.t1_cs + t * 32 + h * 8;
                *reinterpret_cast<u32x4 *>(dst) = o[0];
                *reinterpret_cast<u32x4 *>(dst + 16) = o[1];
            }
        }
    }
#ifndef HH_NO_CLK
    if (p.clk && tid == 0) atomicMax(p.clk + 1, wall_clock64());
#endif
}

static size_t junc_lds(int mode, bool w1 = true)
{
    const bool ds = mode == 1 || mode == 2, pair = mode >= 2;
    return ((ds ? 2 : 1) + (pair ? 1 : 0)) * W3_BYTES + (w1 ? W1_BYTES : 0) + 576 * 4;
}

hipError_t junction_init()
{
    const void *fns[4] = {reinterpret_cast<const void *>(junction_kernel<0>), reinterpret_cast<const void *>(junction_kernel<1>),
                          reinterpret_cast<const void *>(junction_kernel<2>), reinterpret_cast<const void *>(junction_kernel<3>)};
    for (int m = 0; m < 4; ++m) {
        hipError_t e = hipFuncSetAttribute(fns[m], hipFuncAttributeMaxDynamicSharedMemorySize, (int)junc_lds(m));
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t junction_launch(const JuncParams &p, int num_cus, hipStream_t s)
{
    const bool ds = p.x != nullptr, pair = p.t2a != nullptr;
    const int mode = pair ? (ds ? 2 : 3) : (ds ? 1 : 0);
    const int nthr = mode == 1 ? 512 : 256;
    const int ngroups = (p.npix + nthr / 2 - 1) / (nthr / 2);
    const size_t lds = junc_lds(mode, p.w1 != nullptr);
    const int per_cu = lds <= 80 * 1024 ? 2 : 1;
    const int grid = ngroups < num_cus * per_cu ? ngroups : num_cus * per_cu;
    switch (mode) {
    case 0: HH_LAUNCH(junction_kernel<0>, dim3(grid), dim3(256), lds, s, p); break;
    case 1: HH_LAUNCH(junction_kernel<1>, dim3(grid), dim3(512), lds, s, p); break;
    case 2: HH_LAUNCH(junction_kernel<2>, dim3(grid), dim3(256), lds, s, p); break;
    default: HH_LAUNCH(junction_kernel<3>, dim3(grid), dim3(256), lds, s, p); break;
    }
    return hipGetLastError();
}
