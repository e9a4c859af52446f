// Implicit-GEMM convolution on gfx950 matrix cores (v_mfma_f32_32x32x16_bf16), NHWC bf16.
//
// Replaces the ATen conv2d / conv_transpose2d + batch_norm + relu + add chain the
// reference dispatches per layer (hrnet.py:38-45,83-100,190,202,254,265,354-356;
// higher_hrnet.py:21-29,38,52 -- SURVEY.md §2a K1-K7,K9).  BatchNorm is folded into the
// weights/bias at load time; bias, residual add, ReLU and the bf16 (or fp32 NCHW) store
// are fused into the epilogue.
//
// Mapping (one 256-thread workgroup = 4 waves):
//   * output tile  = TH x TW pixels of one image  x  COUT_T output channels
//   * MFMA roles   : A = weights   (rows = 32 output channels, k = 16 input channels)
//                    B = pixels    (cols = 32 output pixels,  k = 16 input channels)
//                    D[cout][pixel]: lane = pixel, 16 regs = 4 groups of 4 consecutive couts
//                    -> every lane stores 8-byte runs of channels of ITS pixel (NHWC friendly).
//   * K loop       : input-channel chunks of KC; per chunk the (TH-1)*S+KS x (TW-1)*S+KS
//                    input patch (all taps share it) and the chunk's weights are staged in LDS.
//   * LDS layout   : patch pixel stride = 2*KC + 16 bytes -> odd number of 16-B slots, so the
//                    32 lanes of a ds_read_b128 B-fragment (consecutive pixels) hit distinct
//                    bank groups; weights are [tap][KC/8][COUT_T][8] so an A-fragment read is
//                    512 contiguous bytes per half-wave.
#include "kernels.h"

#include <utility>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));  // native vector: HIP's uint4 struct kept staging arrays in scratch

__device__ __forceinline__ unsigned pack_bf16x2(float a, float b)
{
    f32x2 f = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2));
}
__device__ __forceinline__ float bf16_lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float bf16_hi(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }

// Compile-time loop: indices are constant expressions in the front end, so per-thread staging arrays are
// promoted to registers (a "#pragma unroll" loop left them in scratch: guide rule 20).
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

template <int KS, int S, int KC, int NT, int WC, int PT, int TW>
__global__ __launch_bounds__(256, 2) void conv_mfma_kernel(const ConvParams p)
{
    constexpr int RPT = 32 / TW;  // image rows covered by one 32-pixel MFMA column tile
    constexpr int WP = 4 / WC;    // waves along pixels
    constexpr int TH = WP * PT * RPT;
    constexpr int PH = (TH - 1) * S + KS, PW = (TW - 1) * S + KS;
    constexpr int PS = KC * 2 + 16;  // bytes per staged pixel
    constexpr int C8 = KC / 8;
    constexpr int COUT_T = 32 * NT * WC;
    constexpr int PATCH_BYTES = (PH * PW * PS + 15) & ~15;
    constexpr int P_UNITS = PH * PW * C8;           // 16-byte units of one input patch chunk
    constexpr int W_UNITS = KS * KS * C8 * COUT_T;  // 16-byte units of one weight chunk
    constexpr int NPL = (P_UNITS + 255) / 256, NWL = (W_UNITS + 255) / 256;
    constexpr int OS = COUT_T * 2 + 16;             // bytes per pixel of the staged output tile
    constexpr int O8 = COUT_T / 8;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *lds_p = smem;
    char *lds_w = smem + PATCH_BYTES;

    int bid = blockIdx.x;
    const int cg = bid % p.ncg; bid /= p.ncg;
    const int tx = bid % p.tiles_x; bid /= p.tiles_x;
    const int ty = bid % p.tiles_y;
    const int b = bid / p.tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int iy0 = oy0 * S - p.pad_y, ix0 = ox0 * S - p.pad_x;

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int r = lane & 31, h = lane >> 5;
    const int wp = wave / WC, wc = wave % WC;
    const int dy = r / TW, dx = r % TW;

    const int nchunks = p.cin / KC;
    const bf16_raw *in_b = p.in + (size_t)b * p.Hin * p.Win * p.in_cs + p.in_coff;
    const u32x4 *w_cg = reinterpret_cast<const u32x4 *>(p.w) + (size_t)cg * nchunks * W_UNITS;

    // ---- register-staged loads: every global load of a chunk is issued before any LDS write, and the
    //      next chunk's loads are issued before the current chunk's MFMAs (latency hides under compute).
    u32x4 preg[NPL], wreg[NWL];
#define ISSUE_LOADS(chunk_)                                                                                        \
    {                                                                                                              \
        const int ch_ = (chunk_);                                                                                  \
        static_for<NPL>([&](auto ic) {                                                                             \
            constexpr int i = decltype(ic)::value;                                                                 \
            const int u = tid + 256 * i;                                                                           \
            const int pix = u / C8, part = u % C8;                                                                 \
            const int iy = iy0 + pix / PW, ix = ix0 + pix % PW;                                                    \
            const bool ok = u < P_UNITS && iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win;                         \
            const u32x4 *src = reinterpret_cast<const u32x4 *>(                                                    \
                in_b + ((size_t)(ok ? iy : 0) * p.Win + (ok ? ix : 0)) * p.in_cs + ch_ * KC + part * 8);           \
            const u32x4 v = *src; /* always-valid address; zero outside the image = conv padding */               \
            preg[i] = ok ? v : u32x4{0u, 0u, 0u, 0u};                                                                \
        });                                                                                                        \
        const u32x4 *wsrc = w_cg + (size_t)ch_ * W_UNITS;                                                          \
        static_for<NWL>([&](auto ic) {                                                                             \
            constexpr int i = decltype(ic)::value;                                                                 \
            const int u = tid + 256 * i;                                                                           \
            wreg[i] = wsrc[u < W_UNITS ? u : 0];                                                                   \
        });                                                                                                        \
    }
#define WRITE_LDS()                                                                                                \
    {                                                                                                              \
        static_for<NPL>([&](auto ic) {                                                                             \
            constexpr int i = decltype(ic)::value;                                                                 \
            const int u = tid + 256 * i;                                                                           \
            if (u < P_UNITS) *reinterpret_cast<u32x4 *>(lds_p + (u / C8) * PS + (u % C8) * 16) = preg[i];          \
        });                                                                                                        \
        static_for<NWL>([&](auto ic) {                                                                             \
            constexpr int i = decltype(ic)::value;                                                                 \
            const int u = tid + 256 * i;                                                                           \
            if (u < W_UNITS) reinterpret_cast<u32x4 *>(lds_w)[u] = wreg[i];                                        \
        });                                                                                                        \
    }

    ISSUE_LOADS(0);

    // ---- accumulators start at bias (+ residual).  All of these loads are issued back to back (clamped
    //      addresses instead of branches) so they overlap the patch loads instead of serialising on vmcnt(0).
    f32x16 acc[NT][PT];
    {
        float4 bs[NT][4];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                bs[nt][g] = *reinterpret_cast<const float4 *>(p.bias + cg * COUT_T + (wc * NT + nt) * 32 + 8 * g + 4 * h);
        uint2 rv[PT][NT][4];
        if (p.res) {
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
                const int oy = oy0 + (wp * PT + pt) * RPT + dy, ox = ox0 + dx;
                const bool valid = oy < p.Ho && ox < p.Wo;
                const size_t pix = valid ? ((size_t)b * p.Hob + (oy * p.osy + p.ooy)) * p.Wob + (ox * p.osx + p.oox) : 0;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int c0 = cg * COUT_T + (wc * NT + nt) * 32 + 8 * g + 4 * h;
                        const bool ok = valid && c0 < p.cout_store;
                        const uint2 v = *reinterpret_cast<const uint2 *>(p.res + pix * p.res_cs + p.res_coff + (ok ? c0 : 0));
                        rv[pt][nt][g] = ok ? v : make_uint2(0u, 0u);
                    }
            }
        } else {
#pragma unroll
            for (int pt = 0; pt < PT; ++pt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int g = 0; g < 4; ++g) rv[pt][nt][g] = make_uint2(0u, 0u);
        }
#pragma unroll
        for (int pt = 0; pt < PT; ++pt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    acc[nt][pt][4 * g + 0] = bs[nt][g].x + bf16_lo(rv[pt][nt][g].x);
                    acc[nt][pt][4 * g + 1] = bs[nt][g].y + bf16_hi(rv[pt][nt][g].x);
                    acc[nt][pt][4 * g + 2] = bs[nt][g].z + bf16_lo(rv[pt][nt][g].y);
                    acc[nt][pt][4 * g + 3] = bs[nt][g].w + bf16_hi(rv[pt][nt][g].y);
                }
    }

    for (int chunk = 0; chunk < nchunks; ++chunk) {
        WRITE_LDS();
        __syncthreads();
        if (chunk + 1 < nchunks) ISSUE_LOADS(chunk + 1);

        // ---- MFMA over taps x 16-channel k-steps; the LDS fragment reads of step s+1 are issued before the
        //      MFMAs of step s (sched_group_barrier pins that order), so ds_read latency hides under MFMA issue.
        {
            constexpr int NSTEP = KS * KS * (KC / 16);
            u32x4 fa[2][NT], fb[2][PT];
            auto ldf = [&](int st, int buf) {
                const int tap = st / (KC / 16), kk = st % (KC / 16), ky = tap / KS, kx = tap % KS;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const int unit = ((tap * C8 + kk * 2 + h) * COUT_T) + (wc * NT + nt) * 32 + r;
                    fa[buf][nt] = *reinterpret_cast<const u32x4 *>(lds_w + unit * 16);
                }
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) {
                    const int row = (wp * PT + pt) * RPT + dy;
                    const int addr = ((row * S + ky) * PW + dx * S + kx) * PS + kk * 32 + h * 16;
                    fb[buf][pt] = *reinterpret_cast<const u32x4 *>(lds_p + addr);
                }
            };
            ldf(0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, NT + PT, 0);
            static_for<NSTEP>([&](auto ic) {
                constexpr int st = decltype(ic)::value;
                if (st + 1 < NSTEP) {
                    ldf(st + 1, (st + 1) & 1);
                    __builtin_amdgcn_sched_group_barrier(0x100, NT + PT, 0);
                }
#pragma unroll
                for (int pt = 0; pt < PT; ++pt)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)
                        acc[nt][pt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[st & 1][nt]),
                                                                              __builtin_bit_cast(bf16x8, fb[st & 1][pt]),
                                                                              acc[nt][pt], 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x8, NT * PT, 0);
            });
        }
        __syncthreads();  // every wave is done reading this chunk (LDS is rewritten next)
    }

    // ---- epilogue: (ReLU) -> fp32 NCHW directly, bf16 NHWC through an LDS transpose so that every
    //      lane stores 16 contiguous bytes and a tile row leaves as one contiguous run.
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        const int lp = ((wp * PT + pt) * RPT + dy) * TW + dx;  // pixel index inside the tile
        const int oy = oy0 + (wp * PT + pt) * RPT + dy, ox = ox0 + dx;
        const bool valid = oy < p.Ho && ox < p.Wo;
        const int Y = oy * p.osy + p.ooy, X = ox * p.osx + p.oox;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v0 = acc[nt][pt][4 * g + 0], v1 = acc[nt][pt][4 * g + 1];
                float v2 = acc[nt][pt][4 * g + 2], v3 = acc[nt][pt][4 * g + 3];
                if (p.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
                const int cl = (wc * NT + nt) * 32 + 8 * g + 4 * h;  // channel inside the WG's cout tile
                if (p.out) {
                    uint2 o; o.x = pack_bf16x2(v0, v1); o.y = pack_bf16x2(v2, v3);
                    *reinterpret_cast<uint2 *>(smem + lp * OS + cl * 2) = o;
                }
                if (p.out_f32 && valid) {
                    const int c0 = cg * COUT_T + cl;
                    const size_t plane = (size_t)p.Hob * p.Wob;
                    float *o = p.out_f32 + ((size_t)b * p.cout_real + c0) * plane + (size_t)Y * p.Wob + X;
                    if (c0 + 0 < p.cout_real) o[0] = v0;
                    if (c0 + 1 < p.cout_real) o[plane] = v1;
                    if (c0 + 2 < p.cout_real) o[2 * plane] = v2;
                    if (c0 + 3 < p.cout_real) o[3 * plane] = v3;
                }
            }
    }
    if (p.out) {
        __syncthreads();
        constexpr int O_UNITS = TH * TW * O8;
#pragma unroll
        for (int i = 0; i < (O_UNITS + 255) / 256; ++i) {
            const int u = tid + 256 * i;
            const int lp = u / O8, part = u % O8;
            const int oy = oy0 + lp / TW, ox = ox0 + lp % TW;
            const int c = cg * COUT_T + part * 8;
            if (u < O_UNITS && oy < p.Ho && ox < p.Wo && c < p.cout_store) {
                const size_t pix = ((size_t)b * p.Hob + (oy * p.osy + p.ooy)) * p.Wob + (ox * p.osx + p.oox);
                *reinterpret_cast<uint4 *>(p.out + pix * p.out_cs + p.out_coff + c) =
                    *reinterpret_cast<const uint4 *>(smem + lp * OS + part * 16);
            }
        }
    }
}

#undef ISSUE_LOADS
#undef WRITE_LDS

// ---------------------------------------------------------------------------------------
// Instantiation table. {KS, S, KC, NT, WC, PT, TW}
#define CONV_CONFIGS(X)                                                                             \
    X(3, 1, 32, 1, 1, 2, 32) /* 0: 3x3 s1, Cout tile 32,  8x32 px  (C=32 branches, deconv head) */ \
    X(3, 1, 32, 2, 1, 2, 32) /* 1: 3x3 s1, Cout tile 64,  8x32 px  (C=64/128 branches)          */ \
    X(3, 1, 32, 2, 1, 1, 16) /* 2: 3x3 s1, Cout tile 64,  8x16 px  (16x16 maps)                 */ \
    X(3, 1, 16, 1, 1, 2, 32) /* 3: 3x3 s1, KC 16 fallback (Cin % 32 != 0, e.g. W48)             */ \
    X(3, 1, 16, 2, 1, 1, 16) /* 4: same, narrow maps                                            */ \
    X(3, 2, 16, 2, 1, 1, 32) /* 5: 3x3 s2, Cout tile 64,  4x32 px                               */ \
    X(3, 2, 16, 1, 1, 1, 32) /* 6: 3x3 s2, Cout tile 32                                         */ \
    X(3, 2, 16, 2, 1, 1, 16) /* 7: 3x3 s2, narrow maps                                          */ \
    X(3, 2, 16, 1, 1, 1, 16) /* 8                                                               */ \
    X(1, 1, 32, 2, 1, 2, 32) /* 9: 1x1, Cout tile 64, 8x32 px                                   */ \
    X(1, 1, 32, 1, 1, 4, 32) /* 10: 1x1, Cout tile 32, 16x32 px                                 */ \
    X(1, 1, 32, 2, 1, 1, 16) /* 11: 1x1 narrow maps                                             */ \
    X(1, 1, 32, 1, 1, 1, 16) /* 12                                                              */ \
    X(1, 1, 16, 2, 1, 2, 32) /* 13: 1x1 KC 16 fallback                                          */ \
    X(1, 1, 16, 1, 1, 2, 32) /* 14                                                              */ \
    X(2, 1, 16, 1, 1, 4, 32) /* 15: 2x2 phase of the 4x4 s2 transposed conv, Cout tile 32       */ \
    X(2, 1, 16, 2, 1, 2, 32) /* 16: same, Cout tile 64 (W48: C=48 -> 64)                        */

#define CFG_ROW(ks, s, kc, nt, wc, pt, tw) {ks, s, kc, nt, wc, pt, tw},
static const ConvConfig g_configs[] = {CONV_CONFIGS(CFG_ROW)};
#undef CFG_ROW

typedef void (*conv_fn)(const ConvParams);
#define CFG_FN(ks, s, kc, nt, wc, pt, tw) conv_mfma_kernel<ks, s, kc, nt, wc, pt, tw>,
static const conv_fn g_fns[] = {CONV_CONFIGS(CFG_FN)};
#undef CFG_FN

int conv_num_configs() { return (int)(sizeof(g_configs) / sizeof(g_configs[0])); }
const ConvConfig &conv_config(int i) { return g_configs[i]; }

hipError_t conv_init()
{
    for (int i = 0; i < conv_num_configs(); ++i) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(g_fns[i]),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)g_configs[i].lds_bytes());
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t conv_launch(int cfg_index, const ConvParams &p, hipStream_t stream)
{
    const ConvConfig &c = g_configs[cfg_index];
    const unsigned grid = (unsigned)p.B * p.tiles_y * p.tiles_x * p.ncg;
    hipLaunchKernelGGL(g_fns[cfg_index], dim3(grid), dim3(256), c.lds_bytes(), stream, p);
    return hipGetLastError();
}
