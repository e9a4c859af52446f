// EXPERIMENTAL -- built only with `make EXPERIMENTAL=1`, reachable only from tools/conv_bench.py (cfg >= 200).
// Measured on MI355X (B=32): bit-compatible with conv_mfma.hip (max |diff| 0.002-0.004 = accumulation order) but
// 5-50 % SLOWER (C=64@64^2: 26.8 vs 17.7 us; C=128@32^2: 18.8 vs 15.6; C=256@16^2: 18.6 vs 15.2), with one
// workgroup per CU and a 3-4 deep LDS ring.  Both kernels sit at ~8-13 B/cycle/CU of staged bytes, i.e. the
// per-CU fill rate and not the staging mechanism bounds these layers.  Kept for the next round.
//
// 3x3 stride-1 convolution for the 64/128/256-channel HRNet branches with LDS-DMA staging.
//
// Same math and MFMA roles as conv_mfma.hip (A = weights, B = pixels, v_mfma_f32_32x32x16_bf16, folded BN,
// residual + ReLU epilogue), but the operands travel HBM/L2 -> LDS with `global_load_lds_dwordx4`
// (no VGPR staging, no ds_write, no address VALU in the loop) into a DOUBLE-BUFFERED LDS image, one
// barrier per 16-channel K chunk:
//     issue DMA(chunk c+1 -> buf[(c+1)&1])  ||  MFMA(chunk c from buf[c&1])  ->  __syncthreads()
// An LDS-DMA write is lane-linear (wave-uniform base + lane*16), so the patch has no per-pixel padding;
// bank conflicts of the B-fragment ds_read_b128 are removed by an XOR swizzle applied on the per-lane
// SOURCE address and on the read: the 16-byte half c of pixel p lives in slot c ^ ((p >> 3) & 1).
// Zero padding = out-of-image lanes point their source at a 16-byte zero line.
#include "../kernels.h"

#include <utility>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {
__device__ __forceinline__ unsigned pack_bf16x2(float a, float b)
{
    f32x2 f = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2));
}
__device__ __forceinline__ float bf16_lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float bf16_hi(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }

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

constexpr int KC = 16, NT = 2, COUT_T = 64;
constexpr int W_UNITS = 9 * 2 * COUT_T;  // 16-byte units of one weight chunk: [tap][2][64]

template <int PT, int TW>
struct Geo {
    static constexpr int RPT = 32 / TW, TH = 4 * PT * RPT, PH = TH + 2, PW = TW + 2;
    static constexpr int P_UNITS = PH * PW * 2, T_UNITS = P_UNITS + W_UNITS;
    static constexpr int NI = (T_UNITS + 63) / 64;   // wave-level DMA instructions per chunk
    static constexpr int NIW = (NI + 3) / 4;         // per wave
    static constexpr int BUF_BYTES = NIW * 4 * 1024; // one LDS buffer: every wave issues exactly NIW instructions
    static constexpr int NBUF = (150 * 1024) / BUF_BYTES >= 4 ? 4 : 3;  // LDS ring: NBUF-1 chunks in flight
};
}  // namespace

template <int PT, int TW>
__global__ __launch_bounds__(256, 1) void conv3x3_dma_kernel(const ConvParams p)
{
    using G = Geo<PT, TW>;
    constexpr int RPT = G::RPT, TH = G::TH, PW = G::PW;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    int bid = blockIdx.x;
    const int cg = bid % p.ncg; bid /= p.ncg;
    const int tx = bid % p.tiles_x; bid /= p.tiles_x;
    const int ty = bid % p.tiles_y;
    const int b = bid / p.tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW;

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int r = lane & 31, h = lane >> 5;
    const int dy = r / TW, dx = r % TW;
    const int nchunks = p.cin / KC;

    // ---- per-lane DMA sources of chunk 0 (chunk-invariant geometry; chunk c adds c*16 channels / one weight chunk)
    const bf16_raw *in_b = p.in + (size_t)b * p.Hin * p.Win * p.in_cs + p.in_coff;
    const bf16_raw *w_cg = p.w + (size_t)cg * nchunks * W_UNITS * 8;
    const bf16_raw *src[G::NIW];
    unsigned kind = 0;  // 2 bits per instruction: 0 = zero line, 1 = patch (advance 16 channels per chunk), 2 = weights
    static_for<G::NIW>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        const int u = (i * 4 + wave) * 64 + lane;
        src[i] = p.zero;
        if (u < G::P_UNITS) {
            const int pidx = u >> 1, s = u & 1, c = s ^ ((pidx >> 3) & 1);
            const int iy = oy0 - 1 + pidx / PW, ix = ox0 - 1 + pidx % PW;
            if (iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win) {
                src[i] = in_b + ((size_t)iy * p.Win + ix) * p.in_cs + c * 8;
                kind |= 1u << (2 * i);
            }
        } else if (u < G::T_UNITS) {
            src[i] = w_cg + (size_t)(u - G::P_UNITS) * 8;
            kind |= 2u << (2 * i);
        }
    });
    auto dma = [&](auto ic, int chunk, int buf) {
        constexpr int i = decltype(ic)::value;
        {  // every wave issues all NIW instructions (surplus ones copy the zero line into slack): uniform vmcnt counts
            const unsigned kd = (kind >> (2 * i)) & 3u;
            const bf16_raw *g = src[i] + (kd == 1u ? chunk * KC : kd == 2u ? chunk * (W_UNITS * 8) : 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                             (__attribute__((address_space(3))) void *)(smem + buf * G::BUF_BYTES + (i * 4 + wave) * 1024),
                                             16, 0, 0);
        }
    };
    constexpr int D = G::NBUF - 1;  // chunks in flight
    for (int c = 0; c < D && c < nchunks; ++c) static_for<G::NIW>([&](auto ic) { dma(ic, c, c); });

    // ---- accumulators = bias (+ residual), as conv_mfma.hip
    f32x16 acc[NT][PT];
    {
        float4 bs[NT][4];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) bs[nt][g] = *reinterpret_cast<const float4 *>(p.bias + cg * COUT_T + nt * 32 + 8 * g + 4 * h);
        u32x4 rv[PT][NT][2];
        if (p.res) {
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
                const int oy = oy0 + (wave * PT + pt) * RPT + dy, ox = ox0 + dx;
                const bool valid = oy < p.Ho && ox < p.Wo;
                const size_t pix = valid ? ((size_t)b * p.Hob + oy) * p.Wob + ox : 0;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        const int c0 = cg * COUT_T + nt * 32 + 16 * m + 8 * h;
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
    // pixel index (inside the patch) of this lane's column in each of its PT column tiles, for tap (0,0)
    int pbase[PT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) pbase[pt] = ((wave * PT + pt) * RPT + dy) * PW + dx;

    int ring = 0;  // chunk % NBUF
    for (int chunk = 0; chunk < nchunks; ++chunk) {
        // chunk `chunk` of this wave has landed once at most `ahead` younger chunks' DMAs are outstanding (vmcnt is
        // in order); the barrier then covers every other wave's part and frees the buffer consumed last iteration.
        const int ahead = nchunks - 1 - chunk < D - 1 ? nchunks - 1 - chunk : D - 1;
        if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * G::NIW) : "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G::NIW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const char *buf = smem + ring * G::BUF_BYTES;
        const char *wbuf = buf + G::P_UNITS * 16;
        const bool more = chunk + D < nchunks;
        const int nring = ring + D >= G::NBUF ? ring + D - G::NBUF : ring + D;
        u32x4 fa[2][NT], fb[2][PT];
        auto ldf = [&](int tap, int sel) {
            const int ky = tap / 3, kx = tap % 3;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                fa[sel][nt] = *reinterpret_cast<const u32x4 *>(wbuf + (((tap * 2 + h) * COUT_T) + nt * 32 + r) * 16);
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) {
                const int pidx = pbase[pt] + ky * PW + kx;
                fb[sel][pt] = *reinterpret_cast<const u32x4 *>(buf + pidx * 32 + ((h ^ ((pidx >> 3) & 1)) << 4));
            }
        };
        ldf(0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, NT + PT, 0);
        static_for<9>([&](auto ic) {
            constexpr int st = decltype(ic)::value;
            if (st + 1 < 9) {
                ldf(st + 1, (st + 1) & 1);
                __builtin_amdgcn_sched_group_barrier(0x100, NT + PT, 0);
            }
            if (more) {  // DMA of chunk+D into the buffer released by the barrier above, spread over the k-steps
                if constexpr (st < G::NIW) dma(ic, chunk + D, nring);
                if constexpr (st + 9 < G::NIW) dma(std::integral_constant<int, st + 9>{}, chunk + D, nring);
            }
#pragma unroll
            for (int pt = 0; pt < PT; ++pt)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt)
                    acc[nt][pt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[st & 1][nt]),
                                                                          __builtin_bit_cast(bf16x8, fb[st & 1][pt]), acc[nt][pt], 0, 0, 0);
            __builtin_amdgcn_sched_group_barrier(0x8, NT * PT, 0);
        });
        ring = ring + 1 == G::NBUF ? 0 : ring + 1;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's LDS reads of the chunk are done
    }

    // ---- epilogue: (ReLU) -> bf16 NHWC, half-waves paired so every lane stores 16 contiguous bytes
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
        const int oy = oy0 + (wave * PT + pt) * RPT + dy, ox = ox0 + dx;
        const bool valid = oy < p.Ho && ox < p.Wo;
        const size_t pix = ((size_t)b * p.Hob + oy) * p.Wob + ox;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            if (p.relu)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[nt][pt][i] = fmaxf(acc[nt][pt][i], 0.f);
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const unsigned x0 = pack_bf16x2(acc[nt][pt][8 * m + 0], acc[nt][pt][8 * m + 1]);
                const unsigned x1 = pack_bf16x2(acc[nt][pt][8 * m + 2], acc[nt][pt][8 * m + 3]);
                const unsigned y0 = pack_bf16x2(acc[nt][pt][8 * m + 4], acc[nt][pt][8 * m + 5]);
                const unsigned y1 = pack_bf16x2(acc[nt][pt][8 * m + 6], acc[nt][pt][8 * m + 7]);
                auto s0 = __builtin_amdgcn_permlane32_swap(x0, y0, false, false);
                auto s1 = __builtin_amdgcn_permlane32_swap(x1, y1, false, false);
                const int c0 = cg * COUT_T + nt * 32 + 16 * m + 8 * h;
                if (valid && c0 < p.cout_store)
                    *reinterpret_cast<u32x4 *>(p.out + pix * p.out_cs + p.out_coff + c0) = u32x4{s0[0], s1[0], s0[1], s1[1]};
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
struct DmaVariant { int PT, TW; };
static const DmaVariant g_dma_variants[] = {{4, 32}, {2, 32}, {2, 16}, {1, 16}};
typedef void (*dma_fn)(const ConvParams);
static const dma_fn g_dma_fns[] = {conv3x3_dma_kernel<4, 32>, conv3x3_dma_kernel<2, 32>, conv3x3_dma_kernel<2, 16>,
                                   conv3x3_dma_kernel<1, 16>};
static size_t dma_lds(int v)
{
    switch (v) {
    case 0: return Geo<4, 32>::NBUF * Geo<4, 32>::BUF_BYTES;
    case 1: return Geo<2, 32>::NBUF * Geo<2, 32>::BUF_BYTES;
    case 2: return Geo<2, 16>::NBUF * Geo<2, 16>::BUF_BYTES;
    default: return Geo<1, 16>::NBUF * Geo<1, 16>::BUF_BYTES;
    }
}
int conv_dma_num_variants() { return 4; }
void conv_dma_variant(int v, int *PT, int *TW) { *PT = g_dma_variants[v].PT; *TW = g_dma_variants[v].TW; }

hipError_t conv_dma_init()
{
    for (int v = 0; v < 4; ++v) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(g_dma_fns[v]), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)dma_lds(v));
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// p.tiles_x / tiles_y / ncg are filled here; requires KS=3, stride 1, pad 1, cin % 16 == 0, cout padded to 64
hipError_t conv_dma_launch(int v, ConvParams p, hipStream_t s)
{
    const int PT = g_dma_variants[v].PT, TW = g_dma_variants[v].TW, TH = 4 * PT * (32 / TW);
    p.tiles_x = (p.Wo + TW - 1) / TW;
    p.tiles_y = (p.Ho + TH - 1) / TH;
    const unsigned grid = (unsigned)p.B * p.tiles_y * p.tiles_x * p.ncg;
    hipLaunchKernelGGL(g_dma_fns[v], dim3(grid), dim3(256), dma_lds(v), s, p);
    return hipGetLastError();
}
