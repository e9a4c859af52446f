// Fused BasicBlock on the fp8 path (e4m3 activations / weights, v_mfma_f32_32x32x64_f8f6f4):
//   out = relu(bn2(conv2(relu(bn1(conv1(x))))) + x)     -- /root/reference/src/keypoints/architectures/hrnet.py:108-124
// for the highest-resolution branch and the deconv head of an fp8 handle (C = 48 for HigherHRNet-W48, C = 64).  Layer by
// layer those are the dominant launches of BASELINE.json configs[4] (conv_fp8_kernel<3,1,48,..>: 113 us each at 0.15 of the
// fp8 MFMA peak: one K chunk per workgroup, so every workgroup is load patch + weights -> 28 MFMAs -> store).  Here one
// persistent 8-wave workgroup per CU keeps BOTH weight sets in LDS (57 KB at 1 B per weight), walks 8x32-pixel tiles, keeps
// the 10x34 intermediate tile in LDS as e4m3 and prefetches the next tile's patch under the MFMAs.
//
// Quantisation points are the layer-by-layer path's: mid = e4m3(relu(acc1 * mult1 + shift1) / s_mid),
// out = e4m3(relu(acc2 * mult2 + shift2 + q_x * s_in) / s_out), mult1 = s_in * s_w1[co], mult2 = s_mid * s_w2[co].
// Wave roles as basicblock_fused_c64.hip: cout tile = wave & 1 (couts padded to 64), conv1 column tiles / conv2 rows by wave >> 1.
// K is enumerated in 16-byte pieces (tap, 16-channel group) exactly as conv_fp8.hip packs the weights.
#include "kernels.h"

#include <utility>

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

namespace {
template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>)
{
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F &&f)
{
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}
__device__ __forceinline__ unsigned pack_fp8x4(float a, float b, float c, float d)
{
    a = fminf(a, 448.f); b = fminf(b, 448.f); c = fminf(c, 448.f); d = fminf(d, 448.f);  // inputs are ReLU outputs (>= 0)
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
    return (unsigned)w;
}
__device__ __forceinline__ i32x8 frag(const u32x4 &lo, const u32x4 &hi)
{
    return i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
}
// 16 fp32 (MFMA C layout: couts 8g + 4h + i) -> scaled e4m3, exchanged so that the lane holds couts 16h .. 16h+15 of its pixel
__device__ __forceinline__ u32x4 pack_tile_fp8(const float y[16], float inv)
{
    unsigned x[2], z[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        x[m] = pack_fp8x4(y[8 * m + 0] * inv, y[8 * m + 1] * inv, y[8 * m + 2] * inv, y[8 * m + 3] * inv);
        z[m] = pack_fp8x4(y[8 * m + 4] * inv, y[8 * m + 5] * inv, y[8 * m + 6] * inv, y[8 * m + 7] * inv);
    }
    auto s0 = __builtin_amdgcn_permlane32_swap(x[0], z[0], false, false);
    auto s1 = __builtin_amdgcn_permlane32_swap(x[1], z[1], false, false);
    auto a0 = __builtin_amdgcn_permlane32_swap(s0[0], s1[0], false, false);
    auto a1 = __builtin_amdgcn_permlane32_swap(s0[1], s1[1], false, false);
    return u32x4{a0[0], a1[0], a0[1], a1[1]};
}

constexpr int TH = 8, TW = 32;
constexpr int MH = TH + 2, MW = TW + 2;
constexpr int IH = TH + 4, IW = TW + 4;
constexpr int MPIX = MH * MW, MT = (MPIX + 31) / 32;  // 340 -> 11 column tiles
constexpr int NTHR = 512;
constexpr int COUT_T = 64;

template <int C>
struct Geo {
    static constexpr int G = C / 16;
    static constexpr int PS = Fp8ConvConfig::pixel_stride(C);
    static constexpr int NPIECE = 9 * G, NSTEP = (NPIECE + 3) / 4;
    static constexpr int P_UNITS = IH * IW * G;
    static constexpr int NPL = (P_UNITS + NTHR - 1) / NTHR;
    static constexpr int PATCH_BYTES = (NPL * NTHR + G - 1) / G * PS;  // + a pad that absorbs the idle units of the last round
    static constexpr int MID_BYTES = MT * 32 * PS;
    static constexpr int W_UNITS = NSTEP * 4 * COUT_T;
    static constexpr int W_BYTES = W_UNITS * 16;
    static constexpr int LDS = PATCH_BYTES + MID_BYTES + 2 * W_BYTES + 4 * COUT_T * 4;
};
}  // namespace

// T16: the trunk's bf16 representation is read as the residual (p.res16) and written beside the e4m3 output (p.out16);
// a compile-time switch, so that the registers of the other residual path do not exist (the kernel sits at the 256-register limit)
template <int C, bool T16>
__global__ __launch_bounds__(NTHR, 1) void bb_fp8_kernel(const Fp8BBParams p)
{
    using GE = Geo<C>;
    constexpr int G = GE::G, PS = GE::PS, NPIECE = GE::NPIECE, NSTEP = GE::NSTEP, NPL = GE::NPL;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *lds_p = smem;
    char *lds_m = smem + GE::PATCH_BYTES;
    char *lds_w1 = lds_m + GE::MID_BYTES;
    char *lds_w2 = lds_w1 + GE::W_BYTES;
    float *lds_c = reinterpret_cast<float *>(lds_w2 + GE::W_BYTES);  // [mult1 | bias1 | mult2 | bias2][64]

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int r = lane & 31, h = lane >> 5;
    const int ct = wave & 1, part = wave >> 1;

    int pl_yx[NPL];  // prefetch unit i: (py << 8) | px of its pixel (py = 255: an idle unit)
    int pl_part[NPL];
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const int u = tid + NTHR * i, pix = u / G;
        pl_part[i] = u % G;
        pl_yx[i] = u < GE::P_UNITS ? (((pix / IW) << 8) | (pix % IW)) : (255 << 8);
    }
    const int q0 = part * 3;
    int paddr[3], maddr[3], myx[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int pidx = (q0 + q) * 32 + r;
        const int pc = pidx < MPIX ? pidx : MPIX - 1;
        const int my = pc / MW, mx = pc % MW;
        paddr[q] = (my * IW + mx) * PS;
        maddr[q] = (pidx < MT * 32 ? pidx : 0) * PS + ct * 32 + 16 * h;
        myx[q] = (my << 8) | mx;
    }
    // byte offset of piece pc = (tap, group) relative to a pixel of an image with `rowpx` pixels per row
    auto piece_off = [](int pc, int rowpx) constexpr {
        const int q = pc < NPIECE ? pc : 0;
        const int tap = q / G, g = q % G;
        return ((tap / 3) * rowpx + (tap % 3)) * PS + g * 16;
    };

    const int tiles_per_img = p.tiles_x * p.tiles_y;
    u32x4 preg[NPL];
    unsigned pf_mask = 0;
    const unsigned char *pf_base = p.in;
    int pf_iy0 = 0, pf_ix0 = 0;
    bool pf_more = true;
    auto band = [&](int i) { return ((p.ntiles & 7) == 0 && (gridDim.x & 7) == 0) ? (i & 7) * (p.ntiles >> 3) + (i >> 3) : i; };
    auto pf_setup = [&](int ti) {
        const int t = band(ti);
        const int b = t / tiles_per_img, tt = t % tiles_per_img;
        pf_iy0 = (tt / p.tiles_x) * TH - 2; pf_ix0 = (tt % p.tiles_x) * TW - 2;
        pf_base = p.in + ((ptrdiff_t)b * p.H * p.W + (ptrdiff_t)pf_iy0 * p.W + pf_ix0) * p.in_cs;
        pf_mask = 0;
    };
    auto pf_load = [&](auto ic) {
        constexpr int i = decltype(ic)::value;
        const int py = pl_yx[i] >> 8, px = pl_yx[i] & 255;
        const int iy = pf_iy0 + py, ix = pf_ix0 + px;
        const bool ok = pf_more & ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
        preg[i] = *reinterpret_cast<const u32x4 *>(ok ? pf_base + (py * p.W + px) * p.in_cs + pl_part[i] * 16 : p.in);
        pf_mask |= ok ? (1u << i) : 0u;
    };
    auto write_patch_unit = [&](auto ic) {
        constexpr int i = decltype(ic)::value;
        const int u = tid + NTHR * i;
        *reinterpret_cast<u32x4 *>(lds_p + (u / G) * PS + (u % G) * 16) = (pf_mask >> i) & 1u ? preg[i] : u32x4{0u, 0u, 0u, 0u};
    };
    auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

    int t = blockIdx.x;
    pf_setup(t);
    static_for<NPL>(pf_load);
    for (int u = tid; u < GE::W_UNITS; u += NTHR) {
        reinterpret_cast<u32x4 *>(lds_w1)[u] = reinterpret_cast<const u32x4 *>(p.w1)[u];
        reinterpret_cast<u32x4 *>(lds_w2)[u] = reinterpret_cast<const u32x4 *>(p.w2)[u];
    }
    if (tid < COUT_T) {
        lds_c[tid] = p.mult1[tid]; lds_c[COUT_T + tid] = p.bias1[tid];
        lds_c[2 * COUT_T + tid] = p.mult2[tid]; lds_c[3 * COUT_T + tid] = p.bias2[tid];
    }
    static_for<NPL>(write_patch_unit);
    __syncthreads();

    float amax_mid = 0.f, amax_out = 0.f;
    for (; t < p.ntiles; t += gridDim.x) {
        const int tb = band(t);
        const int b = tb / tiles_per_img, tt = tb % tiles_per_img;
        const int oy0 = (tt / p.tiles_x) * TH, ox0 = (tt % p.tiles_x) * TW;
        const int tn = t + gridDim.x;
        pf_more = tn < p.ntiles;
        pf_setup(pf_more ? tn : t);

        unsigned resq[2][4];  // residual: e4m3 x at the lane's two output pixels, couts ct*32 + 8g + 4h .. +3
        u32x2 resw[2][4];     // ... or, with a bf16 trunk (p.res16), the same couts as bf16 pairs straight from HBM
        // ================= conv1 + bn1 + relu -> intermediate tile (LDS, e4m3) =================
        auto conv1_phase = [&](auto nqc) {
            constexpr int NQ = decltype(nqc)::value;
            f32x16 acc[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[q][i] = 0.f;
            u32x4 fa[2][2], fb[2][NQ][2];
            auto ld = [&](auto stc, int buf) {
                constexpr int st = decltype(stc)::value;
                constexpr int o00 = piece_off(4 * st, IW), o01 = piece_off(4 * st + 1, IW), o10 = piece_off(4 * st + 2, IW), o11 = piece_off(4 * st + 3, IW);
                const int o0 = h ? o10 : o00, o1 = h ? o11 : o01;
                const int unit = ((st * 2 + h) * 2) * COUT_T + ct * 32 + r;
                fa[buf][0] = *reinterpret_cast<const u32x4 *>(lds_w1 + unit * 16);
                fa[buf][1] = *reinterpret_cast<const u32x4 *>(lds_w1 + (unit + COUT_T) * 16);
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    fb[buf][q][0] = *reinterpret_cast<const u32x4 *>(lds_p + paddr[q] + o0);
                    fb[buf][q][1] = *reinterpret_cast<const u32x4 *>(lds_p + paddr[q] + o1);
                }
            };
            ld(std::integral_constant<int, 0>{}, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * (NQ + 1), 0);
            static_for<NSTEP>([&](auto ic) {
                constexpr int st = decltype(ic)::value;
                if constexpr (st + 1 < NSTEP) {
                    ld(std::integral_constant<int, st + 1>{}, (st + 1) & 1);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2 * (NQ + 1), 0);
                }
                if constexpr (st < NPL) pf_load(ic);  // the next tile's patch: one load per k-step, consumed during conv2
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                    acc[q] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(frag(fa[st & 1][0], fa[st & 1][1]), frag(fb[st & 1][q][0], fb[st & 1][q][1]),
                                                                             acc[q], 0, 0, 0, 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x8, NQ, 0);
            });
            // residual operands out of the patch centre before the patch buffer is recycled
            if constexpr (!T16)
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    // (couts past C are padding: their pixel bytes are the stride pad, not data)
                    resq[q][g] = (ct * 32 + 8 * g < C) ? *reinterpret_cast<const unsigned *>(lds_p + ((part * 2 + q + 2) * IW + r + 2) * PS + ct * 32 + 8 * g + 4 * h) : 0u;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int gy = oy0 - 1 + (myx[q] >> 8), gx = ox0 - 1 + (myx[q] & 255);
                const bool outside = ((unsigned)gy >= (unsigned)p.H) | ((unsigned)gx >= (unsigned)p.W);  // conv2 zero-pads the feature map
                float y[16];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 mu = *reinterpret_cast<const float4 *>(lds_c + ct * 32 + 8 * g + 4 * h);
                    const float4 bs = *reinterpret_cast<const float4 *>(lds_c + COUT_T + ct * 32 + 8 * g + 4 * h);
                    y[4 * g + 0] = fmaxf(__builtin_fmaf(acc[q][4 * g + 0], mu.x, bs.x), 0.f); y[4 * g + 1] = fmaxf(__builtin_fmaf(acc[q][4 * g + 1], mu.y, bs.y), 0.f);
                    y[4 * g + 2] = fmaxf(__builtin_fmaf(acc[q][4 * g + 2], mu.z, bs.z), 0.f); y[4 * g + 3] = fmaxf(__builtin_fmaf(acc[q][4 * g + 3], mu.w, bs.w), 0.f);
                }
                if (p.amax_mid && !outside && (q0 + q) * 32 + r < MPIX)
#pragma unroll
                    for (int i = 0; i < 16; ++i) amax_mid = fmaxf(amax_mid, y[i]);
                const u32x4 o = pack_tile_fp8(y, p.mid_inv_scale);
                *reinterpret_cast<u32x4 *>(lds_m + maddr[q]) = outside ? u32x4{0u, 0u, 0u, 0u} : o;
            }
        };
        if (part < 3) conv1_phase(std::integral_constant<int, 3>{});
        else conv1_phase(std::integral_constant<int, 2>{});
        lds_barrier();  // intermediate tile complete; every wave is done with the patch

        // ================= conv2 + bn2 + residual + relu -> HBM (e4m3) =================
        {
            f32x16 acc2[2];
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc2[q][i] = 0.f;
            u32x4 fa[2][2], fb[2][2][2];
            const int mbase = ((part * 2) * MW + r) * PS;
            auto ld = [&](auto stc, int buf) {
                constexpr int st = decltype(stc)::value;
                constexpr int o00 = piece_off(4 * st, MW), o01 = piece_off(4 * st + 1, MW), o10 = piece_off(4 * st + 2, MW), o11 = piece_off(4 * st + 3, MW);
                const int o0 = h ? o10 : o00, o1 = h ? o11 : o01;
                const int unit = ((st * 2 + h) * 2) * COUT_T + ct * 32 + r;
                fa[buf][0] = *reinterpret_cast<const u32x4 *>(lds_w2 + unit * 16);
                fa[buf][1] = *reinterpret_cast<const u32x4 *>(lds_w2 + (unit + COUT_T) * 16);
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    fb[buf][q][0] = *reinterpret_cast<const u32x4 *>(lds_m + mbase + q * MW * PS + o0);
                    fb[buf][q][1] = *reinterpret_cast<const u32x4 *>(lds_m + mbase + q * MW * PS + o1);
                }
            };
            if constexpr (T16) {  // in flight under conv2's MFMAs, consumed in its epilogue
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int oy = oy0 + part * 2 + q, ox = ox0 + r;
                    const bool ok = (oy < p.H) & (ox < p.W);
                    const bf16_raw *src = p.res16 + (ok ? (((ptrdiff_t)b * p.H + oy) * p.W + ox) * p.res16_cs : 0);
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const bool okc = ok & (ct * 32 + 8 * g < C);
                        const u32x2 v = *reinterpret_cast<const u32x2 *>(src + (okc ? ct * 32 + 8 * g + 4 * h : 0));
                        resw[q][g] = okc ? v : u32x2{0u, 0u};
                    }
                }
            }
            ld(std::integral_constant<int, 0>{}, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
            static_for<NSTEP>([&](auto ic) {
                constexpr int st = decltype(ic)::value;
                if constexpr (st + 1 < NSTEP) {
                    ld(std::integral_constant<int, st + 1>{}, (st + 1) & 1);
                    __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
                }
                if constexpr (st >= NSTEP - NPL) {  // the next patch goes to LDS (the patch buffer is free during conv2)
                    write_patch_unit(std::integral_constant<int, st - (NSTEP - NPL)>{});
                    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                }
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    acc2[q] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(frag(fa[st & 1][0], fa[st & 1][1]), frag(fb[st & 1][q][0], fb[st & 1][q][1]),
                                                                              acc2[q], 0, 0, 0, 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x8, 2, 0);
            });
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int oy = oy0 + part * 2 + q, ox = ox0 + r;
                float y[16];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 mu = *reinterpret_cast<const float4 *>(lds_c + 2 * COUT_T + ct * 32 + 8 * g + 4 * h);
                    const float4 bs = *reinterpret_cast<const float4 *>(lds_c + 3 * COUT_T + ct * 32 + 8 * g + 4 * h);
                    if constexpr (T16) {
                        y[4 * g + 0] = fmaxf(__builtin_bit_cast(float, resw[q][g][0] << 16) + __builtin_fmaf(acc2[q][4 * g + 0], mu.x, bs.x), 0.f);
                        y[4 * g + 1] = fmaxf(__builtin_bit_cast(float, resw[q][g][0] & 0xffff0000u) + __builtin_fmaf(acc2[q][4 * g + 1], mu.y, bs.y), 0.f);
                        y[4 * g + 2] = fmaxf(__builtin_bit_cast(float, resw[q][g][1] << 16) + __builtin_fmaf(acc2[q][4 * g + 2], mu.z, bs.z), 0.f);
                        y[4 * g + 3] = fmaxf(__builtin_bit_cast(float, resw[q][g][1] & 0xffff0000u) + __builtin_fmaf(acc2[q][4 * g + 3], mu.w, bs.w), 0.f);
                    } else {
                        const f32x2 lo = __builtin_amdgcn_cvt_pk_f32_fp8((int)resq[q][g], false), hi = __builtin_amdgcn_cvt_pk_f32_fp8((int)resq[q][g], true);
                        y[4 * g + 0] = fmaxf(__builtin_fmaf(lo[0], p.res_scale, __builtin_fmaf(acc2[q][4 * g + 0], mu.x, bs.x)), 0.f);
                        y[4 * g + 1] = fmaxf(__builtin_fmaf(lo[1], p.res_scale, __builtin_fmaf(acc2[q][4 * g + 1], mu.y, bs.y)), 0.f);
                        y[4 * g + 2] = fmaxf(__builtin_fmaf(hi[0], p.res_scale, __builtin_fmaf(acc2[q][4 * g + 2], mu.z, bs.z)), 0.f);
                        y[4 * g + 3] = fmaxf(__builtin_fmaf(hi[1], p.res_scale, __builtin_fmaf(acc2[q][4 * g + 3], mu.w, bs.w)), 0.f);
                    }
                }
                const bool valid = (oy < p.H) & (ox < p.W);
                if (p.amax_out && valid)
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                        if (ct * 32 + 8 * (i >> 2) < C) amax_out = fmaxf(amax_out, y[i]);
                if constexpr (T16) {  // the block output as bf16 too (the next block's residual): couts 16h .. 16h+15 after the half exchange
                    unsigned gd[4][2];
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        f32x2 f0 = {y[4 * g + 0], y[4 * g + 1]}, f1 = {y[4 * g + 2], y[4 * g + 3]};
                        gd[g][0] = __builtin_bit_cast(unsigned, __builtin_convertvector(f0, bf16x2));
                        gd[g][1] = __builtin_bit_cast(unsigned, __builtin_convertvector(f1, bf16x2));
                    }
                    u32x4 o0, o1;
#pragma unroll
                    for (int d = 0; d < 2; ++d) {
                        auto u02 = __builtin_amdgcn_permlane32_swap(gd[0][d], gd[2][d], false, false);
                        auto u13 = __builtin_amdgcn_permlane32_swap(gd[1][d], gd[3][d], false, false);
                        o0[d] = u02[0]; o0[2 + d] = u02[1];
                        o1[d] = u13[0]; o1[2 + d] = u13[1];
                    }
                    const int c16 = ct * 32 + 16 * h;
                    if (valid && c16 < C) {
                        u32x4 *dst = reinterpret_cast<u32x4 *>(p.out16 + (((ptrdiff_t)b * p.H + oy) * p.W + ox) * p.out16_cs + c16);
                        dst[0] = o0;
                        dst[1] = o1;
                    }
                }
                const u32x4 o = pack_tile_fp8(y, p.out_inv_scale);
                const int c0 = ct * 32 + 16 * h;
                if (valid && c0 < C) *reinterpret_cast<u32x4 *>(p.out + (((ptrdiff_t)b * p.H + oy) * p.W + ox) * p.out_cs + c0) = o;
            }
        }
        lds_barrier();  // every wave is done with the intermediate tile; the next patch is visible
    }
    if (p.amax_mid) {  // calibration: tensor maxima of the intermediate and of the output (values are ReLU outputs: >= 0)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            amax_mid = fmaxf(amax_mid, __shfl_xor(amax_mid, off));
            amax_out = fmaxf(amax_out, __shfl_xor(amax_out, off));
        }
        if (lane == 0) {
            if (amax_mid > 0.f) atomicMax(p.amax_mid, __float_as_uint(amax_mid));
            if (amax_out > 0.f) atomicMax(p.amax_out, __float_as_uint(amax_out));
        }
    }
}

template <int C, bool T16>
static hipError_t launch_c(Fp8BBParams p, int num_cus, hipStream_t s)
{
    static bool inited[64] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (!inited[dev & 63]) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(bb_fp8_kernel<C, T16>), hipFuncAttributeMaxDynamicSharedMemorySize, Geo<C>::LDS);
        if (e != hipSuccess) return e;
        inited[dev & 63] = true;
    }
    p.tiles_x = (p.W + TW - 1) / TW;
    p.tiles_y = (p.H + TH - 1) / TH;
    p.ntiles = p.B * p.tiles_x * p.tiles_y;
    const int grid = p.ntiles < num_cus ? p.ntiles : num_cus;
    HH_LAUNCH((bb_fp8_kernel<C, T16>), dim3(grid), dim3(NTHR), Geo<C>::LDS, s, p);
    return hipGetLastError();
}

bool bb_fp8_supported(int C) { return C == 48 || C == 64; }

hipError_t bb_fp8_launch(int C, const Fp8BBParams &p, int num_cus, hipStream_t s)
{
    const bool t16 = p.res16 && p.out16;
    if ((p.res16 != nullptr) != (p.out16 != nullptr)) return hipErrorInvalidValue;  // both representations of the trunk or neither
    if (C == 48) return t16 ? launch_c<48, true>(p, num_cus, s) : launch_c<48, false>(p, num_cus, s);
    if (C == 64) return t16 ? launch_c<64, true>(p, num_cus, s) : launch_c<64, false>(p, num_cus, s);
    return hipErrorInvalidValue;
}
