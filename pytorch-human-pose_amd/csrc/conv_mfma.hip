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

typedef short i16x2 __attribute__((ext_vector_type(2)));
// Round a pair to bf16 and clamp it from below as signed 16-bit integers: floor = {0,0} is ReLU (every negative bf16,
// -0 included, is a negative int16; non-negative ones keep their bits), floor = {-32768,-32768} is the identity.
// One v_pk_max_i16 per pair instead of two canonicalise + two v_max_f32 on the fp32 values.
__device__ __forceinline__ unsigned pack_bf16x2(float a, float b, i16x2 floor)
{
    f32x2 f = {a, b};
    const i16x2 v = __builtin_bit_cast(i16x2, __builtin_convertvector(f, bf16x2));
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(v, floor));
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

#ifdef HH_CONV_DEBUG  // phase stamps of every workgroup (wave 0), read by tools/probes/conv_probe.hip only
__device__ long long g_conv_dbg[8192 * 8];
#define CONV_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x < 8192) g_conv_dbg[blockIdx.x * 8 + (i)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define CONV_STAMP(i)
#endif
// DB = 1 (round 3): the K loop runs on TWO LDS buffers.  While the MFMAs of chunk c read buffer c & 1, the staging registers
// that hold chunk c + 1 (loaded during chunk c - 1) are written to the other buffer between them and refilled with the loads
// of chunk c + 2: ONE LDS-only barrier per chunk, and no phase in which the matrix pipe waits for ds_write / vmcnt -- in the
// single-buffer form a wave's 15 ds_write_b128 per chunk (~780 clk of the CU's store path) and two barriers stand between the
// MFMA runs, and a layer of 128 / 256 channels has one wave per SIMD, so nothing else runs meanwhile.  With KC = 16 the two
// buffers take what one KC = 32 buffer took, so the launches keep sharing CUs with the other branch lanes.
template <int KS, int S, int KC, int NT, int WC, int PT, int TW, int DB>
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
    constexpr int NPL = (P_UNITS + 255) / 256, NWL = (W_UNITS + 255) / 256, NL = NPL + NWL;
    constexpr int NSTEP = KS * KS * (KC / 16);      // MFMA k-steps per chunk
    constexpr int LPS = (NL + NSTEP - 1) / NSTEP;   // prefetch loads issued per k-step
    // DB: two buffers of (patch, weights, 16 bytes per thread that staging units without an LDS destination are written to)
    constexpr int DUMP_OFF = PATCH_BYTES + W_UNITS * 16;
    constexpr int BUF_BYTES = DUMP_OFF + (DB ? 256 * 16 : 0);

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *lds_p = smem;
    char *lds_w = smem + PATCH_BYTES;

#ifdef HH_CONV_DEBUG
    if (threadIdx.x == 0 && blockIdx.x < 8192) g_conv_dbg[blockIdx.x * 8 + 7] = (long long)wall_clock64();  // very first instruction
#endif
    int bid = blockIdx.x;
    // XCD-aware block order.  Workgroups are dealt to the 8 XCDs round-robin, and the SF = ncg * nphase workgroups that
    // read the same input patch (the cout groups of a tile; the four phases of a fused transposed conv) should share an
    // XCD so that the patch comes from HBM / MALL once and from that XCD's L2 afterwards: blocks {t, t+8, ..., t+8(SF-1)}
    // of every group of 8*SF are the SF variants of one tile.
    const int nph = p.nphase > 1 ? p.nphase : 1, SF = p.ncg * nph;
    int sub = 0;
    if (SF > 1) {
        sub = (bid >> 3) % SF;
        bid = (bid & 7) | ((bid / (8 * SF)) << 3);
        if (bid >= p.B * p.tiles_y * p.tiles_x) return;  // padding of the last group (before any barrier)
    }
    {   // ... and the tiles themselves go to the XCDs in contiguous bands (tile index low bits = XCD so far): neighbouring tiles
        // then share the halo rows / straddled 128-byte lines of their patches in one L2 instead of fetching them once per XCD
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
#ifndef HH_NO_CLK
    // (grids reach 16384 workgroups: only the first / last 256 dispatched stamp, or the same-address atomics would show up
    // in the very duration they measure)
    CONV_STAMP(0);
    if (p.clk && tid == 0 && blockIdx.x < 256) atomicMin(p.clk, wall_clock64());
#endif
    const int wave = tid >> 6, lane = tid & 63;
    const int r = lane & 31, h = lane >> 5;
    const int wp = wave / WC, wc = wave % WC;
    const int dy = r / TW, dx = r % TW;

    const int nchunks = p.cin / KC;
    const u32x4 *w_cg = reinterpret_cast<const u32x4 *>(p.w + (size_t)ph * p.phase_stride) + (size_t)cg * nchunks * W_UNITS;

    // ---- chunk-invariant geometry of this thread's staging units: source pointer of chunk 0 and an
    //      "inside the image" bit per unit (outside = conv zero padding, applied when the unit is written to LDS)
    const bf16_raw *psrc[NPL];
    unsigned pmask = 0;
    {
        const bf16_raw *in_b = p.in + (size_t)b * p.Hin * p.Win * p.in_cs + p.in_coff;
        static_for<NPL>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const int u = tid + 256 * i;
            const int pix = u / C8, part = u % C8;
            const int iy = iy0 + pix / PW, ix = ix0 + pix % PW;
            const bool ok = u < P_UNITS && iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win;
            psrc[i] = ok ? in_b + ((size_t)iy * p.Win + ix) * p.in_cs + part * 8 : in_b;
            pmask |= ok ? (1u << i) : 0u;
        });
    }
    // ---- register staging.  load_unit(j, chunk) issues the j-th 16-byte load of a chunk (patch units first, then
    //      weight units) and never looks at the result, so no s_waitcnt lands next to it.
    u32x4 preg[NPL], wreg[NWL];
    // element offset of chunk `chunk` from the chunk-0 pointers (wave-uniform): its channels, and for a conv over several
    // concatenated inputs the distance to the tensor the chunk lives in
    auto chunk_off = [&](int chunk) -> ptrdiff_t {  // (selects, no branches: it is evaluated at the head of every chunk)
        const bool first = (p.nch0 == 0) | (chunk < p.nch0), second = chunk < p.nch0 + p.nch1;
        const ptrdiff_t base = first ? (ptrdiff_t)0 : (second ? p.src_delta1 : p.src_delta2);
        const int idx = first ? chunk : (second ? chunk - p.nch0 : chunk - p.nch0 - p.nch1);
        return base + (ptrdiff_t)idx * KC;
    };
    // (`coff` = chunk_off(chunk), evaluated ONCE per chunk by the caller: inside the select it used to sit in, its branches were
    // emitted per unit -- three s_cbranch per patch load in the middle of the MFMA steps.  A unit outside the image reads from
    // its image's first pixel, where the offset is as valid as anywhere; its value is never used.)
    auto load_unit = [&](auto jc, int chunk, ptrdiff_t coff) {
        constexpr int j = decltype(jc)::value;
        if constexpr (j < NPL) {
            preg[j] = *reinterpret_cast<const u32x4 *>(psrc[j] + coff);
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
                *reinterpret_cast<u32x4 *>(lds_p + (u / C8) * PS + (u % C8) * 16) = (pmask >> i) & 1u ? preg[i] : u32x4{0u, 0u, 0u, 0u};
        });
        static_for<NWL>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const int u = tid + 256 * i;
            if (u < W_UNITS) reinterpret_cast<u32x4 *>(lds_w)[u] = wreg[i];
        });
    };

    // DB: one staging unit -> LDS buffer at byte offset `boff`.  Branch-free and without touching the data (a select on the
    // loaded value would pull the wait for the load to wherever the compiler puts the select): every unit has a chunk-invariant
    // destination inside a buffer -- its place in the patch / weight image, or, for a unit past the end or outside the image,
    // the thread's dump slot; the zeros of the out-of-image units (the conv's padding) are written once, below, into both buffers.
    int wdst[DB ? NL : 1];
    if constexpr (DB) {
        static_for<NL>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            const int u = tid + 256 * (j < NPL ? j : j - NPL);
            if constexpr (j < NPL) wdst[j] = (pmask >> j) & 1u ? (u / C8) * PS + (u % C8) * 16 : DUMP_OFF + tid * 16;
            else wdst[j] = u < W_UNITS ? PATCH_BYTES + u * 16 : DUMP_OFF + tid * 16;
        });
        static_for<NPL>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const int u = tid + 256 * i;
            if (u < P_UNITS && !((pmask >> i) & 1u)) {
                *reinterpret_cast<u32x4 *>(smem + (u / C8) * PS + (u % C8) * 16) = u32x4{0u, 0u, 0u, 0u};
                *reinterpret_cast<u32x4 *>(smem + BUF_BYTES + (u / C8) * PS + (u % C8) * 16) = u32x4{0u, 0u, 0u, 0u};
            }
        });
    }
    auto write_unit = [&](auto jc, int boff) {
        constexpr int j = decltype(jc)::value;
        if constexpr (j < NPL) *reinterpret_cast<u32x4 *>(smem + boff + wdst[j]) = preg[j];
        else if constexpr (j < NL) *reinterpret_cast<u32x4 *>(smem + boff + wdst[j]) = wreg[j - NPL];
    };

    static_for<NL>([&](auto jc) { load_unit(jc, 0, 0); });

    // ---- accumulators start at bias (+ residual).  The residual is read 16 bytes per lane (couts 16m+8h..+7 of the
    //      lane's pixel) and the two half-waves exchange halves with v_permlane32_swap into the MFMA C layout.
    f32x16 acc[NT][PT];
    {
        float4 bs[NT][4];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                bs[nt][g] = *reinterpret_cast<const float4 *>(p.bias + cg * COUT_T + (wc * NT + nt) * 32 + 8 * g + 4 * h);
        u32x4 rv[PT][NT][2];
        if (p.res) {
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
                const int oy = oy0 + (wp * PT + pt) * RPT + dy, ox = ox0 + dx;
                const bool valid = oy < p.Ho && ox < p.Wo;
                const size_t pix = valid ? ((size_t)b * p.Hob + (oy * p.osy + ooy)) * p.Wob + (ox * p.osx + oox) : 0;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        const int c0 = cg * COUT_T + (wc * NT + nt) * 32 + 16 * m + 8 * h;
                        const bool ok = valid && c0 < p.cout_store;
                        const u32x4 v = *reinterpret_cast<const u32x4 *>(p.res + pix * p.res_cs + p.res_coff + (ok ? c0 : 0));
                        rv[pt][nt][m] = ok ? v : u32x4{0u, 0u, 0u, 0u};
                    }
            }
        }
#pragma unroll
        for (int pt = 0; pt < PT; ++pt)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    unsigned x0 = 0, x1 = 0, y0 = 0, y1 = 0;
                    if (p.res) {
                        auto s0 = __builtin_amdgcn_permlane32_swap(rv[pt][nt][m][0], rv[pt][nt][m][2], false, false);
                        auto s1 = __builtin_amdgcn_permlane32_swap(rv[pt][nt][m][1], rv[pt][nt][m][3], false, false);
                        x0 = s0[0]; y0 = s0[1]; x1 = s1[0]; y1 = s1[1];
                    }
                    const float4 ba = bs[nt][2 * m], bb = bs[nt][2 * m + 1];
                    acc[nt][pt][8 * m + 0] = ba.x + bf16_lo(x0); acc[nt][pt][8 * m + 1] = ba.y + bf16_hi(x0);
                    acc[nt][pt][8 * m + 2] = ba.z + bf16_lo(x1); acc[nt][pt][8 * m + 3] = ba.w + bf16_hi(x1);
                    acc[nt][pt][8 * m + 4] = bb.x + bf16_lo(y0); acc[nt][pt][8 * m + 5] = bb.y + bf16_hi(y0);
                    acc[nt][pt][8 * m + 6] = bb.z + bf16_lo(y1); acc[nt][pt][8 * m + 7] = bb.w + bf16_hi(y1);
                }
    }

    // ---- MFMA over taps x 16-channel k-steps of one chunk.  LDS fragment reads run one k-step ahead
    //      (sched_group_barrier pins the order) and, when `more_c`, the next chunk's global loads are spread over the
    //      k-steps, LPS per step: issued in a burst they back-pressure the CU's load path (~10 B/cycle) and the MFMAs wait
    //      behind them.  `more_c` is a compile-time flag (the last chunk is a separate copy of the body): a run-time
    //      branch inside a k-step splits the scheduling region and the LDS reads / loads no longer interleave with the MFMAs.
    // DB: `roff` = byte offset of the buffer this chunk reads, `woff` = of the buffer chunk + 1 is written to (when `more`);
    // `more2`: chunk + 2 exists and is loaded into the staging registers as they are written out.  The units are dealt to the
    // LAST steps of the chunk: a register then has (almost) a whole chunk between its load and its LDS write, in the first chunk too.
    auto mfma_chunk = [&](auto more_c, auto more2_c, int chunk, int roff, int woff) {
        constexpr bool more = decltype(more_c)::value, more2 = decltype(more2_c)::value;
        const int nxt = chunk + (DB ? 2 : 1);  // the chunk this one loads
        const ptrdiff_t ncoff = (DB ? more2 : more) ? chunk_off(nxt) : 0;
        // number of staging units among slots [q0, q0 + n) of NSLOT (DB: the units sit in the LAST NL slots)
        constexpr auto stage_count = [](int q0, int n, int nslot) {
            int c = 0;
            for (int q = q0; q < q0 + n; ++q) {
                const int j = DB ? q - (nslot - NL) : q;
                c += j >= 0 && j < NL;
            }
            return c;
        };
        // staging work of slot q of NSLOT (slots are dealt LPX per step)
        auto stage_slot = [&](auto qc, auto nslot_c) {
            constexpr int q = decltype(qc)::value, NSLOT = decltype(nslot_c)::value;
            if constexpr (DB) {
                constexpr int j = q - (NSLOT - NL);
                if constexpr (more && j >= 0) {
                    write_unit(std::integral_constant<int, j>{}, woff);
                    if constexpr (more2) load_unit(std::integral_constant<int, j>{}, nxt, ncoff);
                }
            } else {
                if constexpr (more) load_unit(qc, nxt, ncoff);
            }
        };
        if constexpr (KS == 3 && S == 1 && TW == 32) {
            // LDS read bandwidth (8 clk per ds_read_b128, 128 B/clk/CU) is as scarce as MFMA issue here, so a pixel-row
            // fragment is read once per (kx, k-step) and used for every output row it feeds (patch row i = out row + ky):
            // per (kx, kk) NT*3 weight fragments + PT+2 rows feed NT*PT*3 MFMAs, instead of one read per MFMA operand pair.
            constexpr int NC = 3 * (KC / 16), NR = PT + 2, NS = NC * NR;
            constexpr int LPR = (NL + NS - 1) / NS;
            u32x4 fa[2][3][NT], fb[2];
            auto lda = [&](int c, int buf) {
                const int kx = c / (KC / 16), kk = c % (KC / 16);
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        const int unit = (((ky * 3 + kx) * C8 + kk * 2 + h) * COUT_T) + (wc * NT + nt) * 32 + r;
                        fa[buf][ky][nt] = *reinterpret_cast<const u32x4 *>(lds_w + roff + unit * 16);
                    }
            };
            auto ldb = [&](int s, int buf) {
                const int c = s / NR, i = s % NR, kx = c / (KC / 16), kk = c % (KC / 16);
                fb[buf] = *reinterpret_cast<const u32x4 *>(lds_p + roff + ((wp * PT + i) * PW + dx + kx) * PS + kk * 32 + h * 16);
            };
            lda(0, 0);
            ldb(0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 3 * NT + 1, 0);
            static_for<NS>([&](auto sc) {
                constexpr int s = decltype(sc)::value, c = s / NR, i = s % NR;
                constexpr int nread = (s + 1 < NS ? 1 : 0) + ((i == 0 && c + 1 < NC) ? 3 * NT : 0);
                if constexpr (s + 1 < NS) ldb(s + 1, (s + 1) & 1);
                if constexpr (i == 0 && c + 1 < NC) lda(c + 1, (c + 1) & 1);  // next combo's weights, a whole combo ahead
                if constexpr (nread > 0) __builtin_amdgcn_sched_group_barrier(0x100, nread, 0);
                static_for<LPR>([&](auto lc) { stage_slot(std::integral_constant<int, s * LPR + decltype(lc)::value>{}, std::integral_constant<int, NS * LPR>{}); });
                {   // pin the step's staging between its LDS reads and its MFMAs: left alone, the scheduler gathers the loads of
                    // several steps into one burst
                    constexpr int nst = stage_count(s * LPR, LPR, NS * LPR);
                    if constexpr (DB && more && nst > 0) __builtin_amdgcn_sched_group_barrier(0x200, nst, 0);
                    if constexpr ((DB ? more2 : more) && nst > 0) __builtin_amdgcn_sched_group_barrier(0x020, nst, 0);
                }
                constexpr int lo = i - (PT - 1) > 0 ? i - (PT - 1) : 0, hi = i < 2 ? i : 2;  // ky range with 0 <= i - ky < PT
                static_for<3>([&](auto kyc) {
                    constexpr int ky = decltype(kyc)::value, pt = i - ky;
                    if constexpr (pt >= 0 && pt < PT)
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
                            acc[nt][pt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[c & 1][ky][nt]),
                                                                                  __builtin_bit_cast(bf16x8, fb[s & 1]), acc[nt][pt], 0, 0, 0);
                });
                __builtin_amdgcn_sched_group_barrier(0x8, NT * (hi - lo + 1), 0);
            });
            return;
        }
        u32x4 fa[2][NT], fb[2][PT];
        auto ldf = [&](int st, int buf) {
            const int tap = st / (KC / 16), kk = st % (KC / 16), ky = tap / KS, kx = tap % KS;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int unit = ((tap * C8 + kk * 2 + h) * COUT_T) + (wc * NT + nt) * 32 + r;
                fa[buf][nt] = *reinterpret_cast<const u32x4 *>(lds_w + roff + unit * 16);
            }
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
                const int row = (wp * PT + pt) * RPT + dy;
                const int addr = ((row * S + ky) * PW + dx * S + kx) * PS + kk * 32 + h * 16;
                fb[buf][pt] = *reinterpret_cast<const u32x4 *>(lds_p + roff + addr);
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
            static_for<LPS>([&](auto lc) { stage_slot(std::integral_constant<int, st * LPS + decltype(lc)::value>{}, std::integral_constant<int, NSTEP * LPS>{}); });
            {
                constexpr int nst = stage_count(st * LPS, LPS, NSTEP * LPS);
                if constexpr (DB && more && nst > 0) __builtin_amdgcn_sched_group_barrier(0x200, nst, 0);
                if constexpr ((DB ? more2 : more) && nst > 0) __builtin_amdgcn_sched_group_barrier(0x020, nst, 0);
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
    };
    CONV_STAMP(1);
    if constexpr (DB) {
        // LDS-only barrier: __syncthreads() would also drain vmcnt, i.e. the loads of the chunk after next
        auto lds_barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
        // (the host only picks this instantiation for layers of >= 2 chunks: no run-time case split around the loads below, which
        // would cost a vmcnt(0) at the join -- the first MFMAs would wait for chunk 1)
        static_for<NL>([&](auto jc) { write_unit(jc, 0); });
        {
            const ptrdiff_t c1 = chunk_off(1);
            static_for<NL>([&](auto jc) { load_unit(jc, 1, c1); });
        }
        lds_barrier();
        CONV_STAMP(2);
        int chunk = 0;
        for (; chunk + 2 < nchunks; ++chunk) {
            mfma_chunk(std::true_type{}, std::true_type{}, chunk, (chunk & 1) * BUF_BYTES, ((chunk + 1) & 1) * BUF_BYTES);
            lds_barrier();  // buffer (chunk + 1) & 1 is complete, and every wave is done reading buffer chunk & 1
        }
        mfma_chunk(std::true_type{}, std::false_type{}, chunk, (chunk & 1) * BUF_BYTES, ((chunk + 1) & 1) * BUF_BYTES);
        lds_barrier();
        ++chunk;
        CONV_STAMP(4);
        mfma_chunk(std::false_type{}, std::false_type{}, chunk, (chunk & 1) * BUF_BYTES, 0);
    } else {
        for (int chunk = 0; chunk + 1 < nchunks; ++chunk) {
            write_lds();
            __syncthreads();
            if (chunk == 0) CONV_STAMP(2);
            mfma_chunk(std::true_type{}, std::false_type{}, chunk, 0, 0);
            __syncthreads();  // every wave is done reading this chunk (LDS is rewritten next)
            if (chunk == 0) CONV_STAMP(3);
        }
        write_lds();
        __syncthreads();
        CONV_STAMP(4);
        mfma_chunk(std::false_type{}, std::false_type{}, nchunks - 1, 0, 0);
    }
    CONV_STAMP(5);

    // ---- epilogue: (ReLU) -> fp32 NCHW directly, or bf16 NHWC with the half-waves paired by v_permlane32_swap so
    //      that every lane stores 16 contiguous bytes (couts 16m+8h..+7 of its pixel) straight from registers.
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        const int oy = oy0 + (wp * PT + pt) * RPT + dy, ox = ox0 + dx;
        const bool valid = oy < p.Ho && ox < p.Wo;
        const int Y = oy * p.osy + ooy, X = ox * p.osx + oox;
        const size_t pix = ((size_t)b * p.Hob + Y) * p.Wob + X;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            if (p.relu && p.out_f32)  // (no layer of the net has both; the bf16 path clamps while packing)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[nt][pt][i] = fmaxf(acc[nt][pt][i], 0.f);
            if (p.out) {
                const short fl = p.relu ? (short)0 : (short)-32768;
                const i16x2 floor = {fl, fl};
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const unsigned x0 = pack_bf16x2(acc[nt][pt][8 * m + 0], acc[nt][pt][8 * m + 1], floor);
                    const unsigned x1 = pack_bf16x2(acc[nt][pt][8 * m + 2], acc[nt][pt][8 * m + 3], floor);
                    const unsigned y0 = pack_bf16x2(acc[nt][pt][8 * m + 4], acc[nt][pt][8 * m + 5], floor);
                    const unsigned y1 = pack_bf16x2(acc[nt][pt][8 * m + 6], acc[nt][pt][8 * m + 7], floor);
                    auto s0 = __builtin_amdgcn_permlane32_swap(x0, y0, false, false);
                    auto s1 = __builtin_amdgcn_permlane32_swap(x1, y1, false, false);
                    const int c0 = cg * COUT_T + (wc * NT + nt) * 32 + 16 * m + 8 * h;
                    if (valid && c0 < p.cout_store)
                        *reinterpret_cast<u32x4 *>(p.out + pix * p.out_cs + p.out_coff + c0) = u32x4{s0[0], s1[0], s0[1], s1[1]};
                }
            }
            if (p.out_f32 && valid) {
                const size_t plane = (size_t)p.Hob * p.Wob;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c0 = cg * COUT_T + (wc * NT + nt) * 32 + 8 * g + 4 * h;
                    float *o = p.out_f32 + ((size_t)b * p.cout_real + c0) * plane + (size_t)Y * p.Wob + X;
                    if (c0 + 0 < p.cout_real) o[0] = acc[nt][pt][4 * g + 0];
                    if (c0 + 1 < p.cout_real) o[plane] = acc[nt][pt][4 * g + 1];
                    if (c0 + 2 < p.cout_real) o[2 * plane] = acc[nt][pt][4 * g + 2];
                    if (c0 + 3 < p.cout_real) o[3 * plane] = acc[nt][pt][4 * g + 3];
                }
            }
        }
    }
#ifndef HH_NO_CLK
    CONV_STAMP(6);
    if (p.clk && tid == 0 && blockIdx.x + 256 >= gridDim.x) atomicMax(p.clk + 1, wall_clock64());
#endif
}

// ---------------------------------------------------------------------------------------
// Instantiation table. {KS, S, KC, NT, WC, PT, TW, DB}
#define CONV_CONFIGS(X)                                                                             \
    X(3, 1, 32, 1, 1, 2, 32, 0) /* 0: 3x3 s1, Cout tile 32,  8x32 px  (C=32 branches, deconv head) */ \
    X(3, 1, 32, 2, 1, 2, 32, 0) /* 1: 3x3 s1, Cout tile 64,  8x32 px  (C=64/128 branches)          */ \
    X(3, 1, 32, 2, 1, 1, 16, 0) /* 2: 3x3 s1, Cout tile 64,  8x16 px  (16x16 maps)                 */ \
    X(3, 1, 16, 1, 1, 2, 32, 0) /* 3: 3x3 s1, KC 16 fallback (Cin % 32 != 0, e.g. W48)             */ \
    X(3, 1, 16, 2, 1, 1, 16, 0) /* 4: same, narrow maps                                            */ \
    X(3, 2, 16, 2, 1, 1, 32, 0) /* 5: 3x3 s2, Cout tile 64,  4x32 px                               */ \
    X(3, 2, 16, 1, 1, 1, 32, 0) /* 6: 3x3 s2, Cout tile 32                                         */ \
    X(3, 2, 16, 2, 1, 1, 16, 0) /* 7: 3x3 s2, narrow maps                                          */ \
    X(3, 2, 16, 1, 1, 1, 16, 0) /* 8                                                               */ \
    X(1, 1, 32, 2, 1, 2, 32, 0) /* 9: 1x1, Cout tile 64, 8x32 px                                   */ \
    X(1, 1, 32, 1, 1, 4, 32, 0) /* 10: 1x1, Cout tile 32, 16x32 px                                 */ \
    X(1, 1, 32, 2, 1, 1, 16, 0) /* 11: 1x1 narrow maps                                             */ \
    X(1, 1, 32, 1, 1, 1, 16, 0) /* 12                                                              */ \
    X(1, 1, 16, 2, 1, 2, 32, 0) /* 13: 1x1 KC 16 fallback                                          */ \
    X(1, 1, 16, 1, 1, 2, 32, 0) /* 14                                                              */ \
    X(2, 1, 16, 1, 1, 4, 32, 0) /* 15: 2x2 phase of the 4x4 s2 transposed conv, Cout tile 32       */ \
    X(2, 1, 16, 2, 1, 2, 32, 0) /* 16: same, Cout tile 64 (W48: C=48 -> 64)                        */ \
    X(3, 1, 16, 2, 1, 2, 32, 1) /* 17: 3x3 s1, two LDS buffers (>= 128 input channels)            */ \
    X(3, 1, 16, 2, 1, 1, 16, 1) /* 18: same, narrow maps                                          */

#define CFG_ROW(ks, s, kc, nt, wc, pt, tw, db) {ks, s, kc, nt, wc, pt, tw, db},
static const ConvConfig g_configs[] = {CONV_CONFIGS(CFG_ROW)};
#undef CFG_ROW

typedef void (*conv_fn)(const ConvParams);
#define CFG_FN(ks, s, kc, nt, wc, pt, tw, db) conv_mfma_kernel<ks, s, kc, nt, wc, pt, tw, db>,
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
    const unsigned tiles = (unsigned)p.B * p.tiles_y * p.tiles_x, sf = (unsigned)p.ncg * (p.nphase > 1 ? p.nphase : 1);
    const unsigned grid = sf > 1 ? (tiles + 7) / 8 * 8 * sf : tiles;  // groups of 8 tiles x sf variants (see the kernel)
    HH_LAUNCH(g_fns[cfg_index], dim3(grid), dim3(256), c.lds_bytes(), stream, p);
    return hipGetLastError();
}
