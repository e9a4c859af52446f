// Weight gradient of a convolution on NHWC bf16 activations (training path):
//   dW[co][ci][ky][kx] = sum over (b, y, x) of dY[b, y, x, co] * X[b, y*S + ky - pad, x*S + kx - pad, ci]   (zero padded)
// As MFMA work this is a GEMM whose contraction runs over PIXELS: per tap, D[co][ci] += A[co][16 px] * B[16 px][ci].
// Both operands therefore need 8 consecutive pixels of ONE channel per lane, while NHWC keeps the channels of one pixel
// together -- the tiles are staged row-major ([pixel][channel], exactly as they come from HBM) and read back with
// gfx950's transposing LDS read (ds_read_b64_tr_b16: a 16-lane group reads a 4-row x 16-column block and each lane
// receives one column), so no separate transpose pass exists.
// One workgroup = 4 waves = the 2x2 (co, ci) tiles of 32 of one 64x64 channel block; every wave keeps KS*KS accumulators
// (one per tap) and the workgroup walks pixel tiles persistently; partial sums go to a workspace that a second kernel
// reduces in a fixed order (deterministic, no atomics).
#include "kernels.h"

#include <utility>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short i16x4 __attribute__((ext_vector_type(4)));
typedef short i16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int RS = 144;  // LDS bytes per staged pixel: 64 channels + 16 pad
constexpr int TH = 4;

template <typename F, int... I>
__device__ __forceinline__ void sfor_impl(F &&f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void sfor(F &&f) { sfor_impl(f, std::make_integer_sequence<int, N>{}); }

__device__ __forceinline__ i16x4 tr_read(const char *lds, int byte_off)
{
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) i16x4 *)(lds + byte_off));
}
}  // namespace

// NTG = taps per workgroup: the 9 taps of a 3x3 kernel are split over two workgroups (5 + 4, blockIdx.z) that stage the same
// tiles (the second read comes from L2) and write disjoint parts of the worker's partial sums.  That doubles the workgroups
// (one per CU was all a 64-channel layer had: every extra pixel-range worker costs a 9 x 64 x 64 partial set the reduction
// reads back) and takes the accumulators from 144 to 80 registers.
// SMALLC (cin, cout <= 32, the highest-resolution branch and the deconv head): the 64 x 64 channel block would leave three
// of the four waves multiplying padding, so there every wave takes the one real 32 x 32 block and the waves split the
// TAPS instead (wave w: taps w, w+4, w+8).
template <int KS, int S, int TW, int NTG, bool SMALLC>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(const WgradParams p)
{
    constexpr int PH = (TH - 1) * S + KS, PW = (TW - 1) * S + KS, NTAP = KS * KS;
    const int tap0 = SMALLC ? (int)(threadIdx.x >> 6) : blockIdx.z * NTG;  // first tap of this wave ...
    constexpr int TSTEP = SMALLC ? 4 : 1;                                   // ... and the stride to its next ones
    constexpr int Y_UNITS = TH * TW * 8, X_UNITS = PH * PW * 8;          // 16-byte units (8 per 64-channel pixel)
    constexpr int NYL = (Y_UNITS + 255) / 256, NXL = (X_UNITS + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *ldsY = smem, *ldsX = smem + TH * TW * RS;

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int tco = SMALLC ? 0 : wave >> 1, tci = SMALLC ? 0 : wave & 1;
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3, hh = g >> 1;
    const int ncib = (p.cin + 63) / 64;
    const int co0 = (blockIdx.y / ncib) * 64, ci0 = (blockIdx.y % ncib) * 64;  // channel block of this workgroup
    const int pad_y = p.pad_y, pad_x = p.pad_x;
    const int tiles_x = (p.Wo + TW - 1) / TW, tiles_y = (p.Ho + TH - 1) / TH, ntiles = p.B * tiles_y * tiles_x;

    f32x16 acc[NTG];
#pragma unroll
    for (int t = 0; t < NTG; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    // X-patch offset (pixels) of this workgroup's taps; a group short of NTG taps repeats its last one (computed, not stored:
    // no branch inside the pipelined loop)
    int tapoff[NTG];
#pragma unroll
    for (int t = 0; t < NTG; ++t) {
        const int tap = min(tap0 + t * TSTEP, NTAP - 1);
        tapoff[t] = (tap / KS) * PW + tap % KS;
    }

    // byte offsets of this lane's transposed reads inside a tile (row part added per k-step)
    const int ycol = (tco * 32 + 16 * (g & 1) + 4 * pp) * 2, xcol = (tci * 32 + 16 * (g & 1) + 4 * pp) * 2;

    // ---- staging registers of a tile's dY tile and X patch; the
    //      zero padding (image border, channels beyond the count) is applied when the registers go to LDS, so that no
    //      load result is touched next to its issue
    u32x4 yreg[NYL], xreg[NXL];
    unsigned ymask = 0, xmask = 0;
    auto issue_loads = [&](int t) {
        int r = t;
        const int tx = r % tiles_x; r /= tiles_x;
        const int ty = r % tiles_y;
        const int b = r / tiles_y;
        const int oy0 = ty * TH, ox0 = tx * TW, iy0 = oy0 * S - pad_y, ix0 = ox0 * S - pad_x;
        ymask = 0; xmask = 0;
        sfor<NYL>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const int u = tid + 256 * i, px = u >> 3, part = u & 7;
            const int oy = oy0 + px / TW, ox = ox0 + px % TW, c = co0 + part * 8;
            const bool ok = (t < ntiles) & (u < Y_UNITS) & (oy < p.Ho) & (ox < p.Wo) & (c < p.cout);
            const size_t off = ok ? (((size_t)b * p.Ho + oy) * p.Wo + ox) * p.cout + c : 0;
            yreg[i] = *reinterpret_cast<const u32x4 *>(p.dy + off);
            ymask |= ok ? (1u << i) : 0u;
        });
        sfor<NXL>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const int u = tid + 256 * i, px = u >> 3, part = u & 7;
            const int iy = iy0 + px / PW, ix = ix0 + px % PW, c = ci0 + part * 8;
            const bool ok = (t < ntiles) & (u < X_UNITS) & ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W) & (c < p.cin);
            const size_t off = ok ? (((size_t)b * p.H + iy) * p.W + ix) * p.cin + c : 0;
            xreg[i] = *reinterpret_cast<const u32x4 *>(p.x + off);
            xmask |= ok ? (1u << i) : 0u;
        });
    };
    if ((int)blockIdx.x < ntiles) issue_loads(blockIdx.x);
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        __syncthreads();  // the previous tile's reads are done
        sfor<NYL>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const int u = tid + 256 * i;
            if (u < Y_UNITS) *reinterpret_cast<u32x4 *>(ldsY + (u >> 3) * RS + (u & 7) * 16) = (ymask >> i) & 1u ? yreg[i] : u32x4{0u, 0u, 0u, 0u};
        });
        sfor<NXL>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const int u = tid + 256 * i;
            if (u < X_UNITS) *reinterpret_cast<u32x4 *>(ldsX + (u >> 3) * RS + (u & 7) * 16) = (xmask >> i) & 1u ? xreg[i] : u32x4{0u, 0u, 0u, 0u};
        });
        __syncthreads();
        // the next tile's loads fly under this tile's MFMAs (the staging registers are free once they are in LDS; with the
        // taps split over two workgroups the 44 of them fit beside 80 accumulator registers)
        issue_loads(t + gridDim.x);
        // ---- contraction over the tile's pixels, 16 per MFMA k-step (one half row of TW = 32, or a row of TW = 16)
        // Flat software pipeline over (k-step, tap): the transposed reads of step i+1 are issued before the MFMA of step i
        // (sched_group_barrier pins that order), so an MFMA never waits on the LDS latency of its own operands.
        constexpr int NKS = TH * (TW / 16), NSTEP = NKS * NTG;
        i16x8 afrag[2], bfrag[2];
        auto ld_a = [&](int ks, int buf) {
            const int y = ks / (TW / 16), hx = ks % (TW / 16);
            const int prow = y * TW + hx * 16 + 8 * hh + q;  // dY pixel row of this lane's first transposed read
            const i16x4 a0 = tr_read(ldsY, prow * RS + ycol), a1 = tr_read(ldsY, (prow + 4) * RS + ycol);
            afrag[buf] = i16x8{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
        };
        auto ld_b = [&](int step, int buf) {
            const int ks = step / NTG, tap = step % NTG;
            const int y = ks / (TW / 16), hx = ks % (TW / 16);
            const int xrow = (y * S) * PW + (hx * 16 + 8 * hh + q) * S + tapoff[tap];
            const i16x4 b0 = tr_read(ldsX, xrow * RS + xcol), b1 = tr_read(ldsX, (xrow + 4 * S) * RS + xcol);
            bfrag[buf] = i16x8{b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
        };
        ld_a(0, 0);
        ld_b(0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
        sfor<NSTEP>([&](auto sc) {
            constexpr int step = decltype(sc)::value, ks = step / NTG, tap = step % NTG;
            constexpr bool next_a = step + 1 < NSTEP && (step + 1) % NTG == 0;
            if constexpr (step + 1 < NSTEP) ld_b(step + 1, (step + 1) & 1);
            if constexpr (next_a) ld_a(ks + 1, (ks + 1) & 1);
            if constexpr (step + 1 < NSTEP) __builtin_amdgcn_sched_group_barrier(0x100, next_a ? 4 : 2, 0);
            acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, afrag[ks & 1]), __builtin_bit_cast(bf16x8, bfrag[step & 1]),
                                                               acc[tap], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
        });
    }
    // ---- partial sums of this workgroup: part[blockIdx.x][tap][coutp][cinp], D layout: lane = ci column, regs = co rows
    const int r32 = lane & 31, h = lane >> 5;
    const int coutp = SMALLC ? 32 : (p.cout + 63) / 64 * 64, cinp = SMALLC ? 32 : ncib * 64;  // (small layers: a quarter of the partial traffic)
    float *part = p.partial + (size_t)blockIdx.x * NTAP * coutp * cinp;
#pragma unroll
    for (int t = 0; t < NTG; ++t)
        if (tap0 + t * TSTEP < NTAP)
#pragma unroll
            for (int gg = 0; gg < 4; ++gg)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int co = co0 + tco * 32 + 8 * gg + 4 * h + i, ci = ci0 + tci * 32 + r32;
                    part[((size_t)(tap0 + t * TSTEP) * coutp + co) * cinp + ci] = acc[t][4 * gg + i];
                }
}

// dW[co][ci][tap] = sum over workgroups (fixed order) of part[wg][tap][co][ci].  One launch: thread (x, g) of a 64 x 16 block adds workers
// g, g + 16, ... for four consecutive elements (16-byte loads, two accumulator sets, all loads of a thread in flight together), the
// sixteen partial rows meet in LDS and are added in the order g = 0..15, and the sum goes to its place in the OIHW gradient.  (Until
// round 3 this was two launches with the rows in a global staging buffer: the same additions in the same order, ~7 us per layer more.)
constexpr int RG = 16;
__global__ __launch_bounds__(1024) void wgrad_reduce_kernel(const float *__restrict__ part, int nwg, size_t nel4, int ntap, int cout, int cin,
                                                            int coutp, int cinp, float *__restrict__ dw)
{
    __shared__ float4 rows[RG][64];
    const int tx = threadIdx.x, g = threadIdx.y;
    const size_t i = (size_t)blockIdx.x * 64 + tx;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
    if (i < nel4) {
        int w = g;
        for (; w + RG < nwg; w += 2 * RG) {
            const float4 u = reinterpret_cast<const float4 *>(part)[(size_t)w * nel4 + i];
            const float4 v = reinterpret_cast<const float4 *>(part)[(size_t)(w + RG) * nel4 + i];
            a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
            b.x += v.x; b.y += v.y; b.z += v.z; b.w += v.w;
        }
        if (w < nwg) {
            const float4 u = reinterpret_cast<const float4 *>(part)[(size_t)w * nel4 + i];
            a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
        }
    }
    rows[g][tx] = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    __syncthreads();
    const int t = g * 64 + tx;  // 0..1023; the first 256 finish one element each
    if (t >= 256) return;
    const size_t o = (size_t)blockIdx.x * 256 + t;  // element in [tap][coutp][cinp] order
    if (o >= nel4 * 4) return;
    const int ci = (int)(o % cinp), co = (int)((o / cinp) % coutp), tap = (int)(o / cinp / coutp);
    if (ci >= cin || co >= cout) return;
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < RG; ++r) sum += reinterpret_cast<const float *>(&rows[r][t >> 2])[t & 3];
    dw[((size_t)co * cin + ci) * ntap + tap] = sum;
}

template <int KS, int S, int TW, bool SMALLC = false>
static hipError_t launch_one(const WgradParams &p, int nwg, hipStream_t s)
{
    constexpr int PH = (TH - 1) * S + KS, PW = (TW - 1) * S + KS;
    constexpr int NTG = SMALLC ? (KS * KS + 3) / 4 : (KS == 3 ? 5 : KS * KS), NGRP = SMALLC ? 1 : (KS * KS + NTG - 1) / NTG;
    const size_t lds = (size_t)(TH * TW + PH * PW) * RS;
    auto fn = conv_wgrad_kernel<KS, S, TW, NTG, SMALLC>;
    static bool configured = false;  // once per instantiation
    if (!configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        configured = true;
    }
    const int ncob = (p.cout + 63) / 64, ncib = (p.cin + 63) / 64;
    hipLaunchKernelGGL(fn, dim3(nwg, ncob * ncib, NGRP), dim3(256), lds, s, p);
    return hipGetLastError();
}

// Persistent pixel-tile workers per 64x64 channel block: enough workgroups to fill the chip (>= 512 over all channel
// blocks), but no more partial-sum sets than that, since every one is 9 x 64 x 64 floats the reduction has to read back.
int conv_wgrad_num_workers(int B, int Ho, int Wo, int stride, int cin, int cout)
{
    const int TW = (stride == 2 || Wo <= 16) ? 16 : 32;
    const int ntiles = B * ((Ho + TH - 1) / TH) * ((Wo + TW - 1) / TW);
    const int nblocks = ((cin + 63) / 64) * ((cout + 63) / 64);
    int want = (HH_WGRAD_WORKERS * 2 + nblocks - 1) / nblocks;
    if (want < HH_WGRAD_WORKERS / 4) want = HH_WGRAD_WORKERS / 4;  // many channel blocks already fill the chip; every worker costs a partial set
    if (want > HH_WGRAD_WORKERS * 2) want = HH_WGRAD_WORKERS * 2;
    return ntiles < want ? ntiles : want;
}

hipError_t conv_wgrad_launch(const WgradParams &p, int ks, int stride, float *dw, hipStream_t s)
{
    const int nwg = conv_wgrad_num_workers(p.B, p.Ho, p.Wo, stride, p.cin, p.cout);
    hipError_t e = hipErrorInvalidValue;
    if (ks == 3 && stride == 1 && p.Wo <= 16) e = launch_one<3, 1, 16>(p, nwg, s);  // narrow maps: no half-empty tiles
    else if (ks == 3 && stride == 1 && p.cin <= 32 && p.cout <= 32) e = launch_one<3, 1, 32, true>(p, nwg, s);
    else if (ks == 3 && stride == 1) e = launch_one<3, 1, 32>(p, nwg, s);
    else if (ks == 1 && stride == 1) e = launch_one<1, 1, 32>(p, nwg, s);
    else if (ks == 3 && stride == 2) e = launch_one<3, 2, 16>(p, nwg, s);
    else if (ks == 2 && stride == 1) e = launch_one<2, 1, 32>(p, nwg, s);
    if (e != hipSuccess) return e;
    const bool smallc = ks == 3 && stride == 1 && p.Wo > 16 && p.cin <= 32 && p.cout <= 32;  // the SMALLC instance ran
    const int coutp = smallc ? 32 : (p.cout + 63) / 64 * 64, cinp = smallc ? 32 : (p.cin + 63) / 64 * 64, ntap = ks * ks;
    const size_t nel = (size_t)ntap * coutp * cinp, nel4 = nel / 4;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((nel4 + 63) / 64)), dim3(64, RG), 0, s, p.partial, nwg, nel4, ntap, p.cout, p.cin, coutp,
                       cinp, dw);
    return hipGetLastError();
}
