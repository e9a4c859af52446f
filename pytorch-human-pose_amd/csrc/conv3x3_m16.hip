// 3x3 stride-1 convolution on v_mfma_f32_16x16x32_bf16 -- an EXPERIMENT of round 3 (PlanSwitches::conv_m16, HH_CONV_M16=1), not
// the default path: the same layer, tile (8 x 32 pixels x 64 output channels per 4-wave workgroup), LDS weight image and K order as
// conv_mfma_kernel<3,1,32,2,1,2,32> (/root/reference/src/keypoints/architectures/hrnet.py:83-100 with BatchNorm folded), on the other
// bf16 MFMA shape.  Why: the forward runs at the board's power limit with the matrix pipes a quarter busy ("The
// power wall"), and MI355X_MICROARCH.md (DVFS give-back, item 7) measures the 16x16x32 shape 1.12-1.15x faster than 32x32x16 at
// equal cycles when the clock is held down; this kernel is there to measure that on a layer of this net.
//
//   MFMA roles: A = weights (16 output channels x 32 input channels: lane l holds cout l & 15, cin 8 (l >> 4) .. +7 -- one 16-byte
//               unit of the [tap][cin / 8][64 couts][8] weight image the 32x32x16 kernels stage, no repacking),
//               B = pixels  (32 input channels x 16 pixels: lane l holds pixel l & 15, cin 8 (l >> 4) .. +7),
//               D[cout][pixel]: lane l = pixel l & 15, registers = couts 4 (l >> 4) .. +3  -> 8-byte bf16 stores.
//   A K step of 32 is one tap of a 32-channel chunk, so a (kx) step is 3 ky x 4 cout tiles x 2 rows x 2 pixel halves = 48 MFMAs of
//   16 cycles: the same 768 cycles and the same twenty ds_read_b128 as the 24 MFMAs of the 32x32x16 form.
//   Patch pixel stride 96 bytes (the 32x32x16 kernels use 80): with 16 pixels x 4 channel groups per read, 96 puts the sixteen
//   lanes of every ds_read_b128 group on sixteen different 16-byte bank quads; 80 is 2-way.
#include "kernels.h"

#include <utility>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef short i16x2 __attribute__((ext_vector_type(2)));

namespace {
__device__ __forceinline__ unsigned pack2(float a, float b, i16x2 floor)
{
    f32x2 f = {a, b};
    const i16x2 v = __builtin_bit_cast(i16x2, __builtin_convertvector(f, bf16x2));
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(v, floor));
}
__device__ __forceinline__ float lo16(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float hi16(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }
template <typename F, int... I>
__device__ __forceinline__ void sfor_impl(F &&f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void sfor(F &&f) { sfor_impl(f, std::make_integer_sequence<int, N>{}); }

constexpr int KC = 32, C8 = KC / 8, TW = 32, PT = 2, TH = 4 * PT, PH = TH + 2, PW = TW + 2, PS = 96, COUT_T = 64;
constexpr int PATCH_BYTES = PH * PW * PS;          // 32,640
constexpr int P_UNITS = PH * PW * C8;              // 1360 sixteen-byte units of one patch chunk
constexpr int W_UNITS = 9 * C8 * COUT_T;           // 2304 of one weight chunk
constexpr int NPL = (P_UNITS + 255) / 256, NWL = (W_UNITS + 255) / 256, NL = NPL + NWL;  // 6 + 9
constexpr int DUMP_OFF = PATCH_BYTES + W_UNITS * 16;  // 256 x 16 bytes nobody reads
constexpr int NROW = PT + 2, NSTEPS = 3 * NROW;    // (kx, patch row) steps of a chunk
constexpr int LPR = (NL + NSTEPS - 1) / NSTEPS;    // next-chunk loads per step
}  // namespace

size_t conv3x3_m16_lds_bytes() { return DUMP_OFF + 256 * 16; }

__global__ __launch_bounds__(256, 2) void conv3x3_m16_kernel(const ConvParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *lds_p = smem, *lds_w = smem + PATCH_BYTES;

    int bid = blockIdx.x;
    // the cout groups of a tile share an XCD, tiles go to the XCDs in contiguous bands (as conv_mfma_kernel)
    const int SF = p.ncg;
    int cg = 0;
    if (SF > 1) {
        cg = (bid >> 3) % SF;
        bid = (bid & 7) | ((bid / (8 * SF)) << 3);
        if (bid >= p.B * p.tiles_y * p.tiles_x) return;
    }
    {
        const int ntiles = p.B * p.tiles_y * p.tiles_x;
        if ((ntiles & 7) == 0) bid = (bid & 7) * (ntiles >> 3) + (bid >> 3);
    }
    const int tx = bid % p.tiles_x; bid /= p.tiles_x;
    const int ty = bid % p.tiles_y;
    const int b = bid / p.tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW, iy0 = oy0 - 1, ix0 = ox0 - 1;

    const int tid = threadIdx.x, wp = tid >> 6, lane = tid & 63, q = lane & 15, g = lane >> 4;
    const int nchunks = p.cin / KC;
    const u32x4 *w_cg = reinterpret_cast<const u32x4 *>(p.w) + (size_t)cg * nchunks * W_UNITS;

    // ---- staging: register-staged chunk loads, zero padding applied when the registers go to LDS
    const bf16_raw *psrc[NPL];
    unsigned pmask = 0;
    {
        const bf16_raw *in_b = p.in + (size_t)b * p.Hin * p.Win * p.in_cs + p.in_coff;
        sfor<NPL>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const int u = tid + 256 * i, pix = u / C8, part = u % C8;
            const int iy = iy0 + pix / PW, ix = ix0 + pix % PW;
            const bool ok = u < P_UNITS && iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win;
            psrc[i] = ok ? in_b + ((size_t)iy * p.Win + ix) * p.in_cs + part * 8 : in_b;
            pmask |= ok ? (1u << i) : 0u;
        });
    }
    u32x4 preg[NPL], wreg[NWL];
    auto load_unit = [&](auto jc, int chunk) {
        constexpr int j = decltype(jc)::value;
        if constexpr (j < NPL) preg[j] = *reinterpret_cast<const u32x4 *>(psrc[j] + chunk * KC);
        else if constexpr (j < NL) {
            const int u = tid + 256 * (j - NPL);
            wreg[j - NPL] = w_cg[(size_t)chunk * W_UNITS + (u < W_UNITS ? u : 0)];
        }
    };
    auto write_lds = [&]() {
        sfor<NPL>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const int u = tid + 256 * i;
            // (a unit past the patch goes to the thread's dump slot: no branch, so no load is left unwaited at the barrier)
            const int dst = u < P_UNITS ? (u / C8) * PS + (u % C8) * 16 : DUMP_OFF + tid * 16;
            *reinterpret_cast<u32x4 *>(smem + dst) = (pmask >> i) & 1u ? preg[i] : u32x4{0u, 0u, 0u, 0u};
        });
        sfor<NWL>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const int u = tid + 256 * i;
            static_assert(W_UNITS % 256 == 0, "whole rounds of weight units");
            reinterpret_cast<u32x4 *>(lds_w)[u] = wreg[i];
        });
    };
    sfor<NL>([&](auto jc) { load_unit(jc, 0); });

    // ---- accumulators: [cout tile of 16][output row of the wave][pixel half], couts 16 ct + 4 g .. +3 of pixel (row, 16 ph + q)
    f32x4 acc[4][PT][2];
    {
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
            const float4 bv = *reinterpret_cast<const float4 *>(p.bias + cg * COUT_T + 16 * ct + 4 * g);
#pragma unroll
            for (int r = 0; r < PT; ++r)
#pragma unroll
                for (int ph = 0; ph < 2; ++ph) acc[ct][r][ph] = f32x4{bv.x, bv.y, bv.z, bv.w};
        }
        if (p.res) {
#pragma unroll
            for (int r = 0; r < PT; ++r)
#pragma unroll
                for (int ph = 0; ph < 2; ++ph) {
                    const int oy = oy0 + wp * PT + r, ox = ox0 + 16 * ph + q;
                    const bool valid = oy < p.Ho && ox < p.Wo;
                    const size_t pix = valid ? ((size_t)b * p.Hob + oy) * p.Wob + ox : 0;
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct) {
                        const int c0 = cg * COUT_T + 16 * ct + 4 * g;
                        const bool ok = valid && c0 < p.cout_store;
                        u32x2 v = *reinterpret_cast<const u32x2 *>(p.res + pix * p.res_cs + p.res_coff + (ok ? c0 : 0));
                        v = ok ? v : u32x2{0u, 0u};  // (+0.0f: the sums stay what they were)
                        acc[ct][r][ph][0] += lo16(v[0]); acc[ct][r][ph][1] += hi16(v[0]);
                        acc[ct][r][ph][2] += lo16(v[1]); acc[ct][r][ph][3] += hi16(v[1]);
                    }
                }
        }
    }

    // ---- one 32-channel chunk: 3 (kx) x 4 (patch row) steps.  The weight fragments of (kx, ky) live in fa[ky][ct]; ky = 2's are
    //      read at the head of a kx (first used at row 2), ky = 0's and ky = 1's for the NEXT kx as soon as this kx is done with them
    //      (rows 2 and 3), so a fragment is in flight for at least a row step (8-16 MFMAs) before its first use.
    const int a_lane = (g * COUT_T + q) * 16, b_lane = q * PS + g * 16;
    auto mfma_chunk = [&](auto more_c, int chunk) {
        constexpr bool more = decltype(more_c)::value;
        u32x4 fa[3][4], fb[2][2];
        auto lda = [&](int kx, int ky) {
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) fa[ky][ct] = *reinterpret_cast<const u32x4 *>(lds_w + ((ky * 3 + kx) * C8 * COUT_T + 16 * ct) * 16 + a_lane);
        };
        auto ldb = [&](int kx, int i, int buf) {
#pragma unroll
            for (int ph = 0; ph < 2; ++ph) fb[buf][ph] = *reinterpret_cast<const u32x4 *>(lds_p + ((wp * PT + i) * PW + 16 * ph + kx) * PS + b_lane);
        };
        lda(0, 0); lda(0, 1); lda(0, 2);
        ldb(0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 14, 0);
        sfor<NSTEPS>([&](auto sc) {
            constexpr int s = decltype(sc)::value, kx = s / NROW, i = s % NROW;
            // reads of the next step's pixel fragments, and of the weight fragments this kx has finished with
            constexpr bool nb = s + 1 < NSTEPS;
            if constexpr (nb) ldb((s + 1) / NROW, (s + 1) % NROW, (s + 1) & 1);
            constexpr bool a2 = i == 0 && kx > 0;                 // ky = 2 of this kx (its registers were busy until the last step)
            constexpr bool a0 = i == PT && kx + 1 < 3;            // ky = 0 of the next kx
            constexpr bool a1 = i == PT + 1 && kx + 1 < 3;        // ky = 1 of the next kx
            constexpr int lo = i - (PT - 1) > 0 ? i - (PT - 1) : 0, hi = i < 2 ? i : 2;  // ky range with 0 <= i - ky < PT
            if constexpr (a2) lda(kx, 2);
            // (ky = 0 is last used at row PT - 1, ky = 1 at row PT: their reloads go behind this step's MFMAs in program order where the
            //  step still uses them)
            if constexpr (a0) lda(kx + 1, 0);
            if constexpr ((nb ? 2 : 0) + (a2 ? 4 : 0) + (a0 ? 4 : 0) > 0) __builtin_amdgcn_sched_group_barrier(0x100, (nb ? 2 : 0) + (a2 ? 4 : 0) + (a0 ? 4 : 0), 0);
            if constexpr (more) {
                sfor<LPR>([&](auto lc) { load_unit(std::integral_constant<int, s * LPR + decltype(lc)::value>{}, chunk + 1); });
                constexpr int nld = (s + 1) * LPR <= NL ? LPR : (s * LPR < NL ? NL - s * LPR : 0);
                if constexpr (nld > 0) __builtin_amdgcn_sched_group_barrier(0x020, nld, 0);
            }
            sfor<3>([&](auto kyc) {
                constexpr int ky = decltype(kyc)::value, r = i - ky;
                if constexpr (r >= 0 && r < PT)
#pragma unroll
                    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                        for (int ph = 0; ph < 2; ++ph)
                            acc[ct][r][ph] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[ky][ct]),
                                                                                     __builtin_bit_cast(bf16x8, fb[s & 1][ph]), acc[ct][r][ph], 0, 0, 0);
            });
            __builtin_amdgcn_sched_group_barrier(0x8, 8 * (hi - lo + 1), 0);
            if constexpr (a1) {  // row PT + 1 only multiplies with ky = 2: ky = 1's registers are free behind this step's reads
                lda(kx + 1, 1);
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
            }
        });
    };
    for (int chunk = 0; chunk + 1 < nchunks; ++chunk) {
        write_lds();
        __syncthreads();
        mfma_chunk(std::true_type{}, chunk);
        __syncthreads();
    }
    write_lds();
    __syncthreads();
    mfma_chunk(std::false_type{}, nchunks - 1);

    // ---- epilogue: (ReLU) -> bf16 NHWC, 8 bytes (couts 16 ct + 4 g .. +3) per lane and tile
    const short fl = p.relu ? (short)0 : (short)-32768;
    const i16x2 floor = {fl, fl};
#pragma unroll
    for (int r = 0; r < PT; ++r)
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
            const int oy = oy0 + wp * PT + r, ox = ox0 + 16 * ph + q;
            const bool valid = oy < p.Ho && ox < p.Wo;
            const size_t pix = ((size_t)b * p.Hob + oy) * p.Wob + ox;
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                const int c0 = cg * COUT_T + 16 * ct + 4 * g;
                if (valid && c0 < p.cout_store)
                    *reinterpret_cast<u32x2 *>(p.out + pix * p.out_cs + p.out_coff + c0) =
                        u32x2{pack2(acc[ct][r][ph][0], acc[ct][r][ph][1], floor), pack2(acc[ct][r][ph][2], acc[ct][r][ph][3], floor)};
            }
        }
}

hipError_t conv3x3_m16_init()
{
    return hipFuncSetAttribute(reinterpret_cast<const void *>(conv3x3_m16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)conv3x3_m16_lds_bytes());
}

// Supported: 3x3 stride 1 pad 1, one input tensor, bf16 NHWC output, cin a multiple of 32, couts packed in groups of 64
// (KC = 32, NT = 2 weight image), cout_store a multiple of 4.
bool conv3x3_m16_supported(const ConvParams &p)
{
    return p.out && !p.out_f32 && p.nphase <= 1 && p.nch0 == 0 && p.cin % KC == 0 && p.osy == 1 && p.osx == 1 && p.ooy == 0 && p.oox == 0 &&
           p.pad_y == 1 && p.pad_x == 1 && p.cout_store % 4 == 0;
}

hipError_t conv3x3_m16_launch(ConvParams p, hipStream_t stream)
{
    p.tiles_x = (p.Wo + TW - 1) / TW;
    p.tiles_y = (p.Ho + TH - 1) / TH;
    const unsigned tiles = (unsigned)p.B * p.tiles_y * p.tiles_x, sf = (unsigned)p.ncg;
    const unsigned grid = sf > 1 ? (tiles + 7) / 8 * 8 * sf : tiles;
    HH_LAUNCH(conv3x3_m16_kernel, dim3(grid), dim3(256), conv3x3_m16_lds_bytes(), stream, p);
    return hipGetLastError();
}
