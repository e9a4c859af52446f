// Fused BasicBlock for the 64-channel branch:   out = relu(bn2(conv2(relu(bn1(conv1(x))))) + x)
// -- /root/reference/src/keypoints/architectures/hrnet.py:108-124 -- in ONE persistent kernel.
//
// Why: layer by layer a 64-channel block is two one-round launches of ~17 us each (512 workgroups that all wait for their
// patch + weights, compute ~5 us of MFMAs, store, drain) with ~4 us between them, the intermediate crossing HBM / L2.
// Fused, the 10x34 intermediate tile stays in LDS and the weight / patch loads of the NEXT phase run under the MFMAs of
// the current one.  Unlike the 32-channel kernel (basicblock_fused.hip) the two 73.7 KB weight sets do not fit beside the
// tiles, so they are streamed through ONE 36.9 KB LDS buffer in 32-input-channel chunks (4 chunks per tile: conv1 c0, c1,
// conv2 c0, c1), each fetched into registers during the previous chunk's MFMAs and written to LDS between two barriers.
//
// Workgroup = 512 threads (8 waves, two per SIMD), one per CU, persistent over 8x32-pixel output tiles:
//   input patch 12x36 px, conv1 output (= conv2 input) 10x34 px flattened into 11 MFMA column tiles of 32 pixels.
//   wave w: cout tile ct = w & 1 (32 of the 64 output channels), part = w >> 1:
//     conv1: column tiles 3*part .. 3*part+2 (part 3: two)        -> 3 accumulators, one A fragment feeds 3 MFMAs
//     conv2: output rows 2*part, 2*part+1                         -> 2 accumulators; a mid-row fragment is read once per
//            (kx, k-step) and used for every output row it feeds (as in basicblock_fused.hip)
//   MFMA roles as in conv_mfma.hip: A = weights (32 couts x 16 cin), B = pixels.
#include "kernels.h"

#include <utility>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {
typedef short i16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_relu_bf16x2(float a, float b)
{
    f32x2 f = {a, b};
    const i16x2 v = __builtin_bit_cast(i16x2, __builtin_convertvector(f, bf16x2));
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(v, i16x2{0, 0}));
}
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
// 32 couts of one pixel (MFMA C layout) -> for m = 0,1 the 16 bytes (bf16, ReLU applied) of couts 16m+8h .. +7
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

constexpr int C = 64;
constexpr int TH = 8, TW = 32;
constexpr int MH = TH + 2, MW = TW + 2;
constexpr int IH = TH + 4, IW = TW + 4;
constexpr int PS = C * 2 + 16;               // 144 bytes per staged pixel: 9 sixteen-byte slots (odd)
constexpr int MPIX = MH * MW;                // 340 mid pixels -> 11 column tiles of 32 (12 idle lanes)
constexpr int MT = (MPIX + 31) / 32;
constexpr int NTHR = 512;
constexpr int P_UNITS = IH * IW * (C / 8);   // 3456 sixteen-byte units
constexpr int NPL = (P_UNITS + NTHR - 1) / NTHR;    // 7 prefetch loads per thread
constexpr int PATCH_BYTES = NPL * NTHR / (C / 8) * PS;  // 64512: the patch + a pad that absorbs the idle units of the last round
constexpr int MID_BYTES = MT * 32 * PS;      // 50688
constexpr int W_UNITS = 9 * 4 * C;           // 2304 sixteen-byte units of one 32-input-channel weight chunk
constexpr int W_BYTES = W_UNITS * 16;        // 36864
constexpr int NWL = (W_UNITS + NTHR - 1) / NTHR;    // 5 (the last round is half idle)
}  // namespace

size_t bb64_lds_bytes() { return PATCH_BYTES + MID_BYTES + W_BYTES + 2 * C * 4; }

__global__ __launch_bounds__(NTHR, 1) void bb64_fused_kernel(const BBParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *lds_p = smem;
    char *lds_m = smem + PATCH_BYTES;
    char *lds_w = lds_m + MID_BYTES;
    float *lds_b = reinterpret_cast<float *>(lds_w + W_BYTES);  // [2][64] folded BN shifts

    const int tid = threadIdx.x;
#ifndef HH_NO_CLK
    if (p.clk && tid == 0) atomicMin(p.clk, wall_clock64());
#endif
    const int wave = tid >> 6, lane = tid & 63;
    const int r = lane & 31, h = lane >> 5;
    const int ct = wave & 1, part = wave >> 1;

    // identity A fragments (rows = couts of this cout tile, k = cin of the 32-channel chunk ct): residual via the matrix pipe
    u32x4 ident[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int j = r - 16 * kk - 8 * h;
        const unsigned one = (j & 1) ? 0x3f800000u : 0x00003f80u;
        const bool on = j >= 0 && j < 8;
        ident[kk] = u32x4{on && (j >> 1) == 0 ? one : 0u, on && (j >> 1) == 1 ? one : 0u, on && (j >> 1) == 2 ? one : 0u,
                          on && (j >> 1) == 3 ? one : 0u};
    }

    // ---- tile-invariant geometry
    int pl_yx[NPL];  // prefetch unit i: (py << 8) | px inside the patch (py = 255: an idle unit of the last round)
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        // unit u -> (pixel, 16-byte part): 16 consecutive lanes take the SAME part of 16 consecutive pixels (conflict-free
        // ds_write_b128 at the 144-byte pixel stride); a group of 128 units = 16 pixels x 8 parts
        const int u = tid + NTHR * i, pix = (u >> 7) * 16 + (u & 15), py = pix / IW, px = pix % IW;
        pl_yx[i] = u < P_UNITS ? ((py << 8) | px) : (255 << 8);
    }
    const int prt8 = ((tid >> 4) & 7) * 8;  // (u >> 4) & 7 does not depend on the round: NTHR is a multiple of 128
    const int q0 = part * 3;
    int paddr[3], maddr[3], myx[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int pidx = (q0 + q) * 32 + r;
        const int pc = pidx < MPIX ? pidx : MPIX - 1;
        const int my = pc / MW, mx = pc % MW;
        paddr[q] = (my * IW + mx) * PS + h * 16;
        maddr[q] = (pidx < MT * 32 ? pidx : 0) * PS + ct * 64 + h * 16;
        myx[q] = (my << 8) | mx;
    }

    const int tiles_per_img = p.tiles_x * p.tiles_y;
    u32x4 preg[NPL], wreg[NWL];
    unsigned pf_mask = 0;
    const bf16_raw *pf_base = p.in;
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
        preg[i] = *reinterpret_cast<const u32x4 *>(ok ? pf_base + (py * p.W + px) * p.in_cs + prt8 : p.in);
        pf_mask |= ok ? (1u << i) : 0u;
    };
    auto write_patch_unit = [&](auto ic) {
        constexpr int i = decltype(ic)::value;
        const int u = tid + NTHR * i;
        *reinterpret_cast<u32x4 *>(lds_p + ((u >> 7) * 16 + (u & 15)) * PS + ((u >> 4) & 7) * 16) = (pf_mask >> i) & 1u ? preg[i] : u32x4{0u, 0u, 0u, 0u};
    };
    // weight chunk `c` of conv `which` (0 / 1): global -> registers (one 16-byte unit per call), registers -> LDS
    auto w_load = [&](auto ic, const bf16_raw *wsrc, int c) {
        constexpr int i = decltype(ic)::value;
        const int u = tid + NTHR * i;
        wreg[i] = reinterpret_cast<const u32x4 *>(wsrc)[(size_t)c * W_UNITS + (u < W_UNITS ? u : 0)];
    };
    auto w_write = [&]() {
        static_for<NWL>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const int u = tid + NTHR * i;
            if (u < W_UNITS) reinterpret_cast<u32x4 *>(lds_w)[u] = wreg[i];
        });
    };
    auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

    int t = blockIdx.x;
    pf_setup(t);
    static_for<NPL>(pf_load);
    static_for<NWL>([&](auto ic) { w_load(ic, p.w1, 0); });
    if (tid < C) { lds_b[tid] = p.b1[tid]; lds_b[C + tid] = p.b2[tid]; }
    static_for<NPL>(write_patch_unit);
    w_write();
    __syncthreads();

    f32x16 acc2[2];
    bool prev = false;
    bf16_raw *prev_out = p.out;
    int prev_oy0 = 0, prev_ox0 = 0;
    auto store_rows = [&](int q) {
        const int oy = prev_oy0 + part * 2 + q, ox = prev_ox0 + r;
        u32x4 o[2];
        pack_rows16(acc2[q], o);
        bf16_raw *dst = prev_out + ((ptrdiff_t)(part * 2 + q) * p.W + r) * p.out_cs + ct * 32 + 8 * h;
        if (!(prev & (oy < p.H) & (ox < p.W))) dst = p.trash + 8 * h;
        *reinterpret_cast<u32x4 *>(dst) = o[0];
        *reinterpret_cast<u32x4 *>(dst + 16) = o[1];
    };

    for (; t < p.ntiles; t += gridDim.x) {
        const int tb = band(t);
        const int b = tb / tiles_per_img, tt = tb % tiles_per_img;
        const int oy0 = (tt / p.tiles_x) * TH, ox0 = (tt % p.tiles_x) * TW;
        const int tn = t + gridDim.x;
        pf_more = tn < p.ntiles;
        pf_setup(pf_more ? tn : t);

        // ================= conv1 + bn1 + relu -> mid tile (LDS, bf16) =================
        auto conv1_phase = [&](auto nqc) {
            constexpr int NQ = decltype(nqc)::value;
            f32x16 acc[NQ];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 bv = *reinterpret_cast<const float4 *>(lds_b + ct * 32 + 8 * g + 4 * h);
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    acc[q][4 * g + 0] = bv.x; acc[q][4 * g + 1] = bv.y; acc[q][4 * g + 2] = bv.z; acc[q][4 * g + 3] = bv.w;
                }
            }
            auto chunk = [&](auto cc) {
                constexpr int c = decltype(cc)::value;
                u32x4 fa[2], fb[2][NQ];
                auto ld1 = [&](int st, int buf) {
                    const int tap = st >> 1, kk = st & 1, ky = tap / 3, kx = tap % 3;
                    fa[buf] = *reinterpret_cast<const u32x4 *>(lds_w + ((tap * 4 + kk * 2 + h) * C + ct * 32 + r) * 16);
#pragma unroll
                    for (int q = 0; q < NQ; ++q)
                        fb[buf][q] = *reinterpret_cast<const u32x4 *>(lds_p + paddr[q] + (ky * IW + kx) * PS + c * 64 + kk * 32);
                };
                ld1(0, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, NQ + 1, 0);
                static_for<18>([&](auto ic) {
                    constexpr int st = decltype(ic)::value;
                    if (st + 1 < 18) {
                        ld1(st + 1, (st + 1) & 1);
                        __builtin_amdgcn_sched_group_barrier(0x100, NQ + 1, 0);
                    }
                    // global loads of the next phase, one per k-step: chunk 0 fetches conv1's second weight chunk, chunk 1 the
                    // first chunk of conv2 (the next tile's patch is fetched during conv2, where fewer registers are live)
                    if constexpr (c == 0) {
                        if constexpr (st < NWL) w_load(ic, p.w1, 1);
                        if constexpr (st >= 12 && st < 14) store_rows(st - 12);  // the previous tile's rows leave meanwhile
                    } else {
                        if constexpr (st < NWL) w_load(ic, p.w2, 0);
                    }
#pragma unroll
                    for (int q = 0; q < NQ; ++q)
                        acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[st & 1]),
                                                                         __builtin_bit_cast(bf16x8, fb[st & 1][q]), acc[q], 0, 0, 0);
                    if constexpr (c == 0 && st >= 12 && st < 14) {
#pragma unroll
                        for (int q = 0; q < NQ; ++q) {
                            __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x2, 36 / NQ, 0);
                        }
                        __builtin_amdgcn_sched_group_barrier(0x40, 2, 0);
                    } else {
                        __builtin_amdgcn_sched_group_barrier(0x8, NQ, 0);
                    }
                });
            };
            chunk(std::integral_constant<int, 0>{});
            lds_barrier();  // every wave is done with weight chunk 0
            w_write();
            lds_barrier();
            chunk(std::integral_constant<int, 1>{});
            // conv2's accumulators start as bn2 shift + residual: the centre of the input patch times an identity A fragment
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 bv = *reinterpret_cast<const float4 *>(lds_b + C + ct * 32 + 8 * g + 4 * h);
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    acc2[q][4 * g + 0] = bv.x; acc2[q][4 * g + 1] = bv.y; acc2[q][4 * g + 2] = bv.z; acc2[q][4 * g + 3] = bv.w;
                }
            }
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const u32x4 x = *reinterpret_cast<const u32x4 *>(lds_p + ((part * 2 + q + 2) * IW + r + 2) * PS + ct * 64 + kk * 32 + h * 16);
                    acc2[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ident[kk]), __builtin_bit_cast(bf16x8, x),
                                                                      acc2[q], 0, 0, 0);
                }
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int gy = oy0 - 1 + (myx[q] >> 8), gx = ox0 - 1 + (myx[q] & 255);
                const bool outside = ((unsigned)gy >= (unsigned)p.H) | ((unsigned)gx >= (unsigned)p.W);
                u32x4 o[2];
                pack_rows16(acc[q], o);
                *reinterpret_cast<u32x4 *>(lds_m + maddr[q]) = outside ? u32x4{0u, 0u, 0u, 0u} : o[0];
                *reinterpret_cast<u32x4 *>(lds_m + maddr[q] + 32) = outside ? u32x4{0u, 0u, 0u, 0u} : o[1];
            }
        };
        if (part < 3) conv1_phase(std::integral_constant<int, 3>{});
        else conv1_phase(std::integral_constant<int, 2>{});
        lds_barrier();  // mid tile complete; every wave is done with the patch and with conv1's weights
        w_write();      // conv2 chunk 0
        lds_barrier();

        // ================= conv2 + bn2 (+ residual already in acc2) =================
        auto conv2_chunk = [&](auto cc) {
            constexpr int c = decltype(cc)::value;
            u32x4 fa[2][3], fb[2];
            auto lda = [&](int cb, int buf) {  // cb = kx * 2 + kk
                const int kx = cb >> 1, kk = cb & 1;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
                    fa[buf][ky] = *reinterpret_cast<const u32x4 *>(lds_w + (((ky * 3 + kx) * 4 + kk * 2 + h) * C + ct * 32 + r) * 16);
            };
            auto ldb = [&](int s, int buf) {  // s = cb * 4 + i
                const int cb = s >> 2, i = s & 3, kx = cb >> 1, kk = cb & 1;
                fb[buf] = *reinterpret_cast<const u32x4 *>(lds_m + ((part * 2 + i) * MW + r + kx) * PS + c * 64 + kk * 32 + h * 16);
            };
            lda(0, 0);
            ldb(0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
            static_for<24>([&](auto sc) {
                constexpr int s = decltype(sc)::value, cb = s >> 2, i = s & 3;
                constexpr int nread = (s + 1 < 24 ? 1 : 0) + ((i == 0 && cb + 1 < 6) ? 3 : 0);
                if constexpr (s + 1 < 24) ldb(s + 1, (s + 1) & 1);
                if constexpr (i == 0 && cb + 1 < 6) lda(cb + 1, (cb + 1) & 1);
                if constexpr (nread > 0) __builtin_amdgcn_sched_group_barrier(0x100, nread, 0);
                if constexpr (c == 0) {  // conv2's second weight chunk and the next tile's patch are fetched
                    if constexpr (s < NWL) w_load(sc, p.w2, 1);
                    else if constexpr (s - NWL < NPL) pf_load(std::integral_constant<int, s - NWL>{});
                } else {  // the next tile starts with conv1's first chunk; its patch goes to LDS (the patch buffer is free)
                    if constexpr (s < NWL) w_load(sc, p.w1, 0);
                    if constexpr (s >= 12 && s - 12 < NPL) {
                        write_patch_unit(std::integral_constant<int, s - 12>{});
                        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                    }
                }
                constexpr int nm = (i == 0 || i == 3) ? 1 : 2;
                static_for<3>([&](auto kyc) {
                    constexpr int ky = decltype(kyc)::value, j = i - ky;
                    if constexpr (j >= 0 && j < 2)
                        acc2[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[cb & 1][ky]),
                                                                          __builtin_bit_cast(bf16x8, fb[s & 1]), acc2[j], 0, 0, 0);
                });
                __builtin_amdgcn_sched_group_barrier(0x8, nm, 0);
            });
        };
        conv2_chunk(std::integral_constant<int, 0>{});
        lds_barrier();
        w_write();  // conv2 chunk 1
        lds_barrier();
        conv2_chunk(std::integral_constant<int, 1>{});
        lds_barrier();  // every wave is done with the mid tile and with the weights; the next patch is visible
        w_write();      // conv1 chunk 0 of the next tile
        lds_barrier();
        prev = true;
        prev_oy0 = oy0; prev_ox0 = ox0;
        prev_out = p.out + (((ptrdiff_t)b * p.H + oy0) * p.W + ox0) * p.out_cs;
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) store_rows(q);
#ifndef HH_NO_CLK
    if (p.clk && tid == 0) atomicMax(p.clk + 1, wall_clock64());
#endif
}

static bf16_raw *g_trash64[64] = {};

hipError_t bb64_fused_init()
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (!g_trash64[dev & 63]) {
        e = hipMalloc((void **)&g_trash64[dev & 63], 256);
        if (e != hipSuccess) return e;
    }
    return hipFuncSetAttribute(reinterpret_cast<const void *>(bb64_fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)bb64_lds_bytes());
}

hipError_t bb64_fused_launch(BBParams p, int num_cus, hipStream_t s)
{
    p.tiles_x = (p.W + TW - 1) / TW;
    p.tiles_y = (p.H + TH - 1) / TH;
    p.ntiles = p.B * p.tiles_x * p.tiles_y;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || !g_trash64[dev & 63]) return hipErrorNotInitialized;
    p.trash = g_trash64[dev & 63];
    const int grid = p.ntiles < num_cus ? p.ntiles : num_cus;
    HH_LAUNCH(bb64_fused_kernel, dim3(grid), dim3(NTHR), bb64_lds_bytes(), s, p);
    return hipGetLastError();
}
