// Implicit-GEMM convolution on gfx950's fp8 matrix path (v_mfma_f32_32x32x64_f8f6f4, OCP e4m3 operands, fp32
// accumulate) -- the fp8 configuration of BASELINE.json (configs[4]); the reference has no fp8 precedent (its inference is
// fp32, keypoints/model.py:79-83), so this path is anchored to the fp32 oracle by tolerance (tests/test_gpu_parity.py).
//
// Numerics: activations live in HBM as e4m3 bytes with one scale per tensor (real = q * s_t), weights as e4m3 with one
// scale per output channel (BN folded first).  acc = sum q_x q_w is exact in fp32 up to accumulation rounding;
//     y = acc * (s_in * s_w[co]) + shift[co] (+ q_res * s_res);  ReLU;  q_y = e4m3(y / s_out).
// The MFMA's own block scales are not used (the unscaled form of the instruction is emitted).
//
// Mapping: as conv_mfma.hip (A = weights, rows = 32 couts; B = pixels, one pixel per lane; 4 waves = 4 pixel groups), but
// one MFMA contracts K = 64: lane half h holds 32 consecutive k bytes.  K is enumerated in 16-byte PIECES = (tap, 16-channel
// group) so that channel counts that are multiples of 16 but not of 64 (48, 96: HigherHRNet-W48) waste at most 3 pieces
// per chunk: k-step s covers pieces 4s..4s+3, lane half h takes pieces 4s+2h and 4s+2h+1 (two ds_read_b128 each for A and
// B).  The weight image is packed in exactly that order, so A and B always meet on the same k.
#include "kernels.h"

#include <utility>

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
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

// four fp32 -> four e4m3 bytes (round to nearest even, clamped to the finite range +-448)
__device__ __forceinline__ unsigned pack_fp8x4(float a, float b, float c, float d)
{
    a = __builtin_amdgcn_fmed3f(a, -448.f, 448.f); b = __builtin_amdgcn_fmed3f(b, -448.f, 448.f);
    c = __builtin_amdgcn_fmed3f(c, -448.f, 448.f); d = __builtin_amdgcn_fmed3f(d, -448.f, 448.f);
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, w, true);
    return (unsigned)w;
}
__device__ __forceinline__ i32x8 frag(const u32x4 &lo, const u32x4 &hi)
{
    return i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
}
}  // namespace

template <int KS, int S, int KC, int NT, int PT, int TW>
__global__ __launch_bounds__(256, 2) void conv_fp8_kernel(const Fp8ConvParams p)
{
    constexpr int RPT = 32 / TW;
    constexpr int TH = 4 * PT * RPT;
    constexpr int PH = (TH - 1) * S + KS, PW = (TW - 1) * S + KS;
    constexpr int G = KC / 16;                                  // 16-byte channel groups per staged pixel
    constexpr int PS = Fp8ConvConfig::pixel_stride(KC);         // odd number of 16-byte slots: conflict-free ds_read_b128
    constexpr int COUT_T = 32 * NT;
    constexpr int NPIECE = KS * KS * G, NSTEP = (NPIECE + 3) / 4;
    constexpr int PATCH_BYTES = (PH * PW * PS + 15) & ~15;
    constexpr int P_UNITS = PH * PW * G;
    constexpr int W_UNITS = NSTEP * 4 * COUT_T;                 // [step][h][i][cout] 16-byte units of one chunk
    constexpr int NPL = (P_UNITS + 255) / 256, NWL = (W_UNITS + 255) / 256, NL = NPL + NWL;
    constexpr int LPS = (NL + NSTEP - 1) / NSTEP;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *lds_p = smem;
    char *lds_w = smem + PATCH_BYTES;

    int bid = blockIdx.x;
    // XCD-aware order, as conv_mfma.hip: the SF = ncg * nphase variants of a tile (cout groups, transposed-conv phases) are
    // blocks {t, t+8, ..} of a group of 8*SF (one XCD under round-robin dispatch), tiles go to the XCDs in contiguous bands
    const int nph = p.nphase > 1 ? p.nphase : 1, SF = p.ncg * nph;
    int sub = 0;
    if (SF > 1) {
        sub = (bid >> 3) % SF;
        bid = (bid & 7) | ((bid / (8 * SF)) << 3);
        if (bid >= p.B * p.tiles_y * p.tiles_x) return;
    }
    {
        const int ntiles = p.B * p.tiles_y * p.tiles_x;
        if ((ntiles & 7) == 0) bid = (bid & 7) * (ntiles >> 3) + (bid >> 3);
    }
    const int cg = sub % p.ncg, ph = sub / p.ncg;
    const int pad_y = nph > 1 ? ((ph >> 1) ? 0 : 1) : p.pad_y, pad_x = nph > 1 ? ((ph & 1) ? 0 : 1) : p.pad_x;
    const int ooy = nph > 1 ? (ph >> 1) : p.ooy, oox = nph > 1 ? (ph & 1) : p.oox;
    const int tx = bid % p.tiles_x; bid /= p.tiles_x;
    const int ty = bid % p.tiles_y;
    const int b = bid / p.tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int iy0 = oy0 * S - pad_y, ix0 = ox0 * S - pad_x;

    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int r = lane & 31, h = lane >> 5;
    const int dy = r / TW, dx = r % TW;

    const int nchunks = p.cin / KC;
    const u32x4 *w_cg = reinterpret_cast<const u32x4 *>(p.w + (size_t)ph * p.phase_stride) + (size_t)cg * nchunks * W_UNITS;

    const unsigned char *psrc[NPL];
    unsigned pmask = 0;
    {
        const unsigned char *in_b = p.in + (size_t)b * p.Hin * p.Win * p.in_cs + p.in_coff;
        static_for<NPL>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const int u = tid + 256 * i;
            const int pix = u / G, part = u % G;
            const int iy = iy0 + pix / PW, ix = ix0 + pix % PW;
            const bool ok = u < P_UNITS && iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win;
            psrc[i] = ok ? in_b + ((size_t)iy * p.Win + ix) * p.in_cs + part * 16 : in_b;
            pmask |= ok ? (1u << i) : 0u;
        });
    }
    u32x4 preg[NPL], wreg[NWL];
    auto load_unit = [&](auto jc, int chunk) {
        constexpr int j = decltype(jc)::value;
        if constexpr (j < NPL) {
            preg[j] = *reinterpret_cast<const u32x4 *>(psrc[j] + ((pmask >> j) & 1u ? chunk * KC : 0));
        } else if constexpr (j < NL) {
            constexpr int i = j - NPL;
            const int u = tid + 256 * i;
            wreg[i] = w_cg[(size_t)chunk * W_UNITS + (u < W_UNITS ? u : 0)];
        }
    };
    auto write_lds = [&]() {
        static_for<NPL>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const int u = tid + 256 * i;
            if (u < P_UNITS)
                *reinterpret_cast<u32x4 *>(lds_p + (u / G) * PS + (u % G) * 16) = (pmask >> i) & 1u ? preg[i] : u32x4{0u, 0u, 0u, 0u};
        });
        static_for<NWL>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const int u = tid + 256 * i;
            if (u < W_UNITS) reinterpret_cast<u32x4 *>(lds_w)[u] = wreg[i];
        });
    };
    static_for<NL>([&](auto jc) { load_unit(jc, 0); });

    // the residual (fp8, 16 bytes = couts 16h..16h+15 of the lane's pixel and cout tile) is fetched now and consumed in
    // the epilogue
    // one register array for both residual forms (the kernel sits near the register limit): bf16 = couts 16h .. 16h+7 and
    // 16h+8 .. 16h+15 of the lane's pixel and cout tile in rw[..][0 / 1]; e4m3 = couts 16h .. 16h+15 in rw[..][0]
    u32x4 rw[PT][NT][2];
    if (p.res16) {
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
            const int oy = oy0 + (wave * PT + pt) * RPT + dy, ox = ox0 + dx;
            const bool valid = oy < p.Ho && ox < p.Wo;
            const size_t pix = valid ? ((size_t)b * p.Hob + (oy * p.osy + ooy)) * p.Wob + (ox * p.osx + oox) : 0;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int c0 = cg * COUT_T + nt * 32 + 16 * h;
                const bool ok = valid && c0 < p.cout_store;
                const u32x4 *src = reinterpret_cast<const u32x4 *>(p.res16 + pix * p.res16_cs + p.res16_coff + (ok ? c0 : 0));
                const u32x4 v0 = src[0], v1 = src[1];
                rw[pt][nt][0] = ok ? v0 : u32x4{0u, 0u, 0u, 0u};
                rw[pt][nt][1] = ok ? v1 : u32x4{0u, 0u, 0u, 0u};
            }
        }
    } else if (p.res) {
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
            const int oy = oy0 + (wave * PT + pt) * RPT + dy, ox = ox0 + dx;
            const bool valid = oy < p.Ho && ox < p.Wo;
            const size_t pix = valid ? ((size_t)b * p.Hob + (oy * p.osy + ooy)) * p.Wob + (ox * p.osx + oox) : 0;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int c0 = cg * COUT_T + nt * 32 + 16 * h;
                const bool ok = valid && c0 < p.cout_store;
                const u32x4 v = *reinterpret_cast<const u32x4 *>(p.res + pix * p.res_cs + p.res_coff + (ok ? c0 : 0));
                rw[pt][nt][0] = ok ? v : u32x4{0u, 0u, 0u, 0u};
            }
        }
    }

    f32x16 acc[NT][PT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int pt = 0; pt < PT; ++pt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[nt][pt][i] = 0.f;

    // byte offset (from the lane's patch base) of piece pc = (tap, group); pieces past the real ones re-read piece 0
    // (their weights are zero)
    auto piece_off = [](int pc) constexpr {
        const int q = pc < NPIECE ? pc : 0;
        const int tap = q / G, g = q % G;
        return ((tap / KS) * PW + (tap % KS)) * PS + g * 16;
    };
    int pbase[PT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) pbase[pt] = (((wave * PT + pt) * RPT + dy) * S * PW + dx * S) * PS;

    auto mfma_chunk = [&](auto more_c, int chunk) {
        constexpr bool more = decltype(more_c)::value;
        u32x4 fa[2][NT][2], fb[2][PT][2];
        auto ldf = [&](auto stc, int buf) {
            constexpr int st = decltype(stc)::value;
            constexpr int o00 = piece_off(4 * st), o01 = piece_off(4 * st + 1), o10 = piece_off(4 * st + 2), o11 = piece_off(4 * st + 3);
            const int o0 = h ? o10 : o00, o1 = h ? o11 : o01;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int unit = ((st * 2 + h) * 2) * COUT_T + nt * 32 + r;
                fa[buf][nt][0] = *reinterpret_cast<const u32x4 *>(lds_w + unit * 16);
                fa[buf][nt][1] = *reinterpret_cast<const u32x4 *>(lds_w + (unit + COUT_T) * 16);
            }
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
                fb[buf][pt][0] = *reinterpret_cast<const u32x4 *>(lds_p + pbase[pt] + o0);
                fb[buf][pt][1] = *reinterpret_cast<const u32x4 *>(lds_p + pbase[pt] + o1);
            }
        };
        ldf(std::integral_constant<int, 0>{}, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 2 * (NT + PT), 0);
        static_for<NSTEP>([&](auto ic) {
            constexpr int st = decltype(ic)::value;
            if constexpr (st + 1 < NSTEP) {
                ldf(std::integral_constant<int, st + 1>{}, (st + 1) & 1);
                __builtin_amdgcn_sched_group_barrier(0x100, 2 * (NT + PT), 0);
            }
            if constexpr (more)
                static_for<LPS>([&](auto lc) { load_unit(std::integral_constant<int, st * LPS + decltype(lc)::value>{}, chunk + 1); });
#pragma unroll
            for (int pt = 0; pt < PT; ++pt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[nt][pt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(frag(fa[st & 1][nt][0], fa[st & 1][nt][1]),
                                                                                  frag(fb[st & 1][pt][0], fb[st & 1][pt][1]),
                                                                                  acc[nt][pt], 0, 0, 0, 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x8, NT * PT, 0);
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

    // ---- epilogue: y = acc * mult[co] + shift[co] (+ residual), ReLU, -> e4m3 NHWC (16 contiguous bytes per lane after
    //      two half-wave exchanges) and / or fp32 NCHW.  Calibration runs also reduce max |y| of the tensor.
    float amax = 0.f;
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        const int oy = oy0 + (wave * PT + pt) * RPT + dy, ox = ox0 + dx;
        const bool valid = oy < p.Ho && ox < p.Wo;
        const int Y = oy * p.osy + ooy, X = ox * p.osx + oox;
        const size_t pix = ((size_t)b * p.Hob + Y) * p.Wob + X;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            float y[16];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c0 = cg * COUT_T + nt * 32 + 8 * g + 4 * h;
                const float4 mu = *reinterpret_cast<const float4 *>(p.mult + c0), bs = *reinterpret_cast<const float4 *>(p.bias + c0);
                y[4 * g + 0] = __builtin_fmaf(acc[nt][pt][4 * g + 0], mu.x, bs.x); y[4 * g + 1] = __builtin_fmaf(acc[nt][pt][4 * g + 1], mu.y, bs.y);
                y[4 * g + 2] = __builtin_fmaf(acc[nt][pt][4 * g + 2], mu.z, bs.z); y[4 * g + 3] = __builtin_fmaf(acc[nt][pt][4 * g + 3], mu.w, bs.w);
            }
            if (p.res16) {
                // 2 x 16 bytes = couts 16h .. 16h+15 as bf16 -> the accumulator layout (group g = couts 8g + 4h .. +3 = two dwords):
                // the exchange of the bf16 store path below, backwards: (g0, g2) = swap(dwords 0-1, 2-3), (g1, g3) = swap(4-5, 6-7)
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    auto u02 = __builtin_amdgcn_permlane32_swap(rw[pt][nt][0][d], rw[pt][nt][0][2 + d], false, false);
                    auto u13 = __builtin_amdgcn_permlane32_swap(rw[pt][nt][1][d], rw[pt][nt][1][2 + d], false, false);
                    const unsigned g4[4] = {u02[0], u13[0], u02[1], u13[1]};  // dword d of groups 0, 1, 2, 3
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        y[4 * g + 2 * d + 0] += __builtin_bit_cast(float, g4[g] << 16);
                        y[4 * g + 2 * d + 1] += __builtin_bit_cast(float, g4[g] & 0xffff0000u);
                    }
                }
            } else if (p.res) {
                // 16 bytes = couts 16h .. 16h+15  ->  the accumulator layout (couts 8g + 4h + i): the two exchanges of the
                // store path, backwards
                auto t0 = __builtin_amdgcn_permlane32_swap(rw[pt][nt][0][0], rw[pt][nt][0][2], false, false);
                auto t1 = __builtin_amdgcn_permlane32_swap(rw[pt][nt][0][1], rw[pt][nt][0][3], false, false);
                // lane half 0: (t0[0], t1[0]) = couts 0..7, (t0[1], t1[1]) = 16..23; half 1: 8..15 and 24..31
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    auto u = __builtin_amdgcn_permlane32_swap(t0[m], t1[m], false, false);  // -> couts 16m+4h.., 16m+8+4h..
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const f32x2 lo = __builtin_amdgcn_cvt_pk_f32_fp8((int)u[q], false), hi = __builtin_amdgcn_cvt_pk_f32_fp8((int)u[q], true);
                        y[8 * m + 4 * q + 0] = __builtin_fmaf(lo[0], p.res_scale, y[8 * m + 4 * q + 0]);
                        y[8 * m + 4 * q + 1] = __builtin_fmaf(lo[1], p.res_scale, y[8 * m + 4 * q + 1]);
                        y[8 * m + 4 * q + 2] = __builtin_fmaf(hi[0], p.res_scale, y[8 * m + 4 * q + 2]);
                        y[8 * m + 4 * q + 3] = __builtin_fmaf(hi[1], p.res_scale, y[8 * m + 4 * q + 3]);
                    }
                }
            }
            if (p.relu)
#pragma unroll
                for (int i = 0; i < 16; ++i) y[i] = fmaxf(y[i], 0.f);
            if (p.absmax) {
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (valid && cg * COUT_T + nt * 32 + 8 * g + 4 * h + i < p.cout_real) amax = fmaxf(amax, fabsf(y[4 * g + i]));
            }
            if (p.out) {
                unsigned x[2], z[2];
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const float s = p.out_inv_scale;
                    x[m] = pack_fp8x4(y[8 * m + 0] * s, y[8 * m + 1] * s, y[8 * m + 2] * s, y[8 * m + 3] * s);
                    z[m] = pack_fp8x4(y[8 * m + 4] * s, y[8 * m + 5] * s, y[8 * m + 6] * s, y[8 * m + 7] * s);
                }
                auto s0 = __builtin_amdgcn_permlane32_swap(x[0], z[0], false, false);  // half 0: couts 0..7, half 1: 8..15
                auto s1 = __builtin_amdgcn_permlane32_swap(x[1], z[1], false, false);  // half 0: 16..23,   half 1: 24..31
                auto a0 = __builtin_amdgcn_permlane32_swap(s0[0], s1[0], false, false);
                auto a1 = __builtin_amdgcn_permlane32_swap(s0[1], s1[1], false, false);
                const int c0 = cg * COUT_T + nt * 32 + 16 * h;
                if (valid && c0 < p.cout_store)
                    *reinterpret_cast<u32x4 *>(p.out + pix * p.out_cs + p.out_coff + c0) = u32x4{a0[0], a1[0], a0[1], a1[1]};
            }
            if (p.out16) {  // the same values as bf16: group g (couts 8g + 4h .. +3) = two dwords; (g0, g2) and (g1, g3) exchanged between the
                            // lane halves leave couts 16h .. 16h+15 contiguous in the lane: two 16-byte stores
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
                const int c0 = cg * COUT_T + nt * 32 + 16 * h;
                if (valid && c0 < p.cout_store) {
                    u32x4 *dst = reinterpret_cast<u32x4 *>(p.out16 + pix * p.out16_cs + p.out16_coff + c0);
                    dst[0] = o0;
                    dst[1] = o1;
                }
            }
            if (p.out_f32 && valid) {
                const size_t plane = (size_t)p.Hob * p.Wob;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c0 = cg * COUT_T + nt * 32 + 8 * g + 4 * h;
                    float *o = p.out_f32 + ((size_t)b * p.cout_real + c0) * plane + (size_t)Y * p.Wob + X;
                    if (c0 + 0 < p.cout_real) o[0] = y[4 * g + 0];
                    if (c0 + 1 < p.cout_real) o[plane] = y[4 * g + 1];
                    if (c0 + 2 < p.cout_real) o[2 * plane] = y[4 * g + 2];
                    if (c0 + 3 < p.cout_real) o[3 * plane] = y[4 * g + 3];
                }
            }
        }
    }
    if (p.absmax) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off));
        if (lane == 0 && amax > 0.f) atomicMax(p.absmax, __float_as_uint(amax));  // non-negative floats order like their bits
    }
}

// ---------------------------------------------------------------------------------------
// Instantiation table {KS, S, KC, NT, PT, TW}.  KC = input channels per K chunk (bytes per staged pixel).
#define FP8_CONFIGS(X)                                                                            \
    X(3, 1, 64, 2, 2, 32) X(3, 1, 64, 2, 1, 16) X(3, 1, 64, 1, 2, 32) X(3, 1, 64, 1, 1, 16)       \
    X(3, 1, 48, 2, 2, 32) X(3, 1, 48, 2, 1, 16) X(3, 1, 48, 1, 2, 32) X(3, 1, 48, 1, 1, 16)       \
    X(3, 1, 32, 2, 2, 32) X(3, 1, 32, 2, 1, 16) X(3, 1, 32, 1, 2, 32) X(3, 1, 32, 1, 1, 16)       \
    X(3, 1, 16, 2, 2, 32) X(3, 1, 16, 2, 1, 16) X(3, 1, 16, 1, 2, 32) X(3, 1, 16, 1, 1, 16)       \
    X(3, 2, 64, 2, 1, 32) X(3, 2, 64, 2, 1, 16) X(3, 2, 64, 1, 1, 32) X(3, 2, 64, 1, 1, 16)       \
    X(3, 2, 48, 2, 1, 32) X(3, 2, 48, 2, 1, 16) X(3, 2, 48, 1, 1, 32) X(3, 2, 48, 1, 1, 16)       \
    X(3, 2, 32, 2, 1, 32) X(3, 2, 32, 2, 1, 16) X(3, 2, 32, 1, 1, 32) X(3, 2, 32, 1, 1, 16)       \
    X(3, 2, 16, 2, 1, 32) X(3, 2, 16, 2, 1, 16) X(3, 2, 16, 1, 1, 32) X(3, 2, 16, 1, 1, 16)       \
    X(1, 1, 64, 2, 2, 32) X(1, 1, 64, 2, 1, 16) X(1, 1, 64, 1, 4, 32) X(1, 1, 64, 1, 1, 16)       \
    X(1, 1, 48, 2, 2, 32) X(1, 1, 48, 2, 1, 16) X(1, 1, 48, 1, 4, 32) X(1, 1, 48, 1, 1, 16)       \
    X(1, 1, 32, 2, 2, 32) X(1, 1, 32, 2, 1, 16) X(1, 1, 32, 1, 4, 32) X(1, 1, 32, 1, 1, 16)       \
    X(1, 1, 16, 2, 2, 32) X(1, 1, 16, 2, 1, 16) X(1, 1, 16, 1, 4, 32) X(1, 1, 16, 1, 1, 16)       \
    X(2, 1, 64, 2, 2, 32) X(2, 1, 64, 1, 4, 32) X(2, 1, 48, 2, 2, 32) X(2, 1, 48, 1, 4, 32)       \
    X(2, 1, 32, 2, 2, 32) X(2, 1, 32, 1, 4, 32) X(2, 1, 16, 2, 2, 32) X(2, 1, 16, 1, 4, 32)

#define CFG_ROW(ks, s, kc, nt, pt, tw) {ks, s, kc, nt, pt, tw},
static const Fp8ConvConfig g_configs[] = {FP8_CONFIGS(CFG_ROW)};
#undef CFG_ROW
typedef void (*conv_fn)(const Fp8ConvParams);
#define CFG_FN(ks, s, kc, nt, pt, tw) conv_fp8_kernel<ks, s, kc, nt, pt, tw>,
static const conv_fn g_fns[] = {FP8_CONFIGS(CFG_FN)};
#undef CFG_FN

int conv_fp8_num_configs() { return (int)(sizeof(g_configs) / sizeof(g_configs[0])); }
const Fp8ConvConfig &conv_fp8_config(int i) { return g_configs[i]; }

hipError_t conv_fp8_init()
{
    for (int i = 0; i < conv_fp8_num_configs(); ++i) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(g_fns[i]), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)g_configs[i].lds_bytes());
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t conv_fp8_launch(int cfg_index, const Fp8ConvParams &p, hipStream_t stream)
{
    const Fp8ConvConfig &c = g_configs[cfg_index];
    const unsigned tiles = (unsigned)p.B * p.tiles_y * p.tiles_x, sf = (unsigned)p.ncg * (p.nphase > 1 ? p.nphase : 1);
    const unsigned grid = sf > 1 ? (tiles + 7) / 8 * 8 * sf : tiles;
    HH_LAUNCH(g_fns[cfg_index], dim3(grid), dim3(256), c.lds_bytes(), stream, p);
    return hipGetLastError();
}
