// EXPERIMENT (not in the default build; `make EXPERIMENTAL=1`, HH_BB32=thin): measured slower than basicblock_fused.hip --
// forward 5.67 vs 5.09 ms with the branch lanes, 6.54 vs 5.99 ms serial (tools/probes/ab_thin.sh).  Kept as a record of the
// idea (weights in registers, half a CU per workgroup) and of what it costs: 21 spilled registers at the 256 limit, and the
// per-phase weight stream from L1 / L2.
//
// "Thin" fused BasicBlock for the 32-channel branches:   out = relu(bn2(conv2(relu(bn1(conv1(x))))) + x)
// -- /root/reference/src/keypoints/architectures/hrnet.py:108-124 --, the same math as basicblock_fused.hip in a workgroup
// that takes HALF a CU: 256 threads, 60 KB of LDS, so that two of them -- or one of them and a 64 KB convolution workgroup
// of another resolution branch -- share a CU and fill each other's barrier / epilogue / load phases.  (The 150 KB, 8-wave
// kernels own their CU: with the branch lanes running side by side they serialise)
//
// What makes it fit: the weights live in REGISTERS, not in LDS.  With 32 output channels a conv has one cout tile, so every
// wave needs all 18 A fragments (9 taps x 2 k-steps, 4 VGPRs each = 72 VGPRs) and nothing else of the weights: each fragment
// is loaded once per phase from global (L1 / L2 resident: 18.4 KB per conv) straight in MFMA layout.  The register set is
// recycled in place: right after conv1's last MFMA with fragment s has issued, conv2's fragment s is loaded over it (and during
// conv2 the next tile's conv1 fragment), so no phase ever waits for weights.  LDS holds the 20x20 input patch and the 18x18
// intermediate tile only, and serves B fragments only (1 ds_read_b128 per MFMA instead of 1.33-1.5).
//   tile 16x16 output pixels; intermediate 18x18 = 324 px flattened into 11 column tiles of 32 (wave w: tiles 3w..3w+2, the
//   last wave two); conv2: 8 column tiles of 2 rows x 16 px, two per wave.
#include "../kernels.h"

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

constexpr int TH = 16, TW = 16;
constexpr int MH = TH + 2, MW = TW + 2;
constexpr int IH = TH + 4, IW = TW + 4;
constexpr int PS = 80;                       // bytes per staged pixel: 32 bf16 + 16 pad (odd number of 16-B slots)
constexpr int MPIX = MH * MW;                // 324 -> 11 column tiles (28 idle lanes)
constexpr int MT = (MPIX + 31) / 32;
constexpr int NTHR = 256;
constexpr int P_UNITS = IH * IW * 4;         // 1600 sixteen-byte units
constexpr int NPL = (P_UNITS + NTHR - 1) / NTHR;   // 7 prefetch loads per thread
constexpr int PATCH_BYTES = NPL * NTHR / 4 * PS;   // 35840: the patch + a pad for the idle units of the last round
constexpr int MID_BYTES = MT * 32 * PS;      // 28160
constexpr int NFRAG = 18;                    // A fragments of a conv: 9 taps x 2 k-steps
}  // namespace

size_t bb_thin_lds_bytes() { return PATCH_BYTES + MID_BYTES + 64 * 4; }

__global__ __launch_bounds__(NTHR, 2) void bb_thin_kernel(const BBParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *lds_p = smem;
    char *lds_m = smem + PATCH_BYTES;
    float *lds_b = reinterpret_cast<float *>(lds_m + MID_BYTES);  // [2][32] folded BN shifts

    const int tid = threadIdx.x;
#ifndef HH_NO_CLK
    if (p.clk && tid == 0) atomicMin(p.clk, wall_clock64());
#endif
    const int wave = tid >> 6, lane = tid & 63;
    const int r = lane & 31, h = lane >> 5;

    // (buffer loads: a wave-uniform descriptor + ONE 32-bit lane offset + a constant per fragment; with flat loads the compiler
    // keeps a 64-bit pointer per fragment and spills)
    const auto rs_w1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_raw *>(p.w1), 0, NFRAG * 1024, 0x00020000);
    const auto rs_w2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_raw *>(p.w2), 0, NFRAG * 1024, 0x00020000);
    const auto rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_raw *>(p.in), 0, 0x7fffffff, 0x00020000);
    const int wofs = (h * 32 + r) * 16;  // bytes; fragment s adds s * 1024
    auto wfrag = [&](const decltype(rs_w1) &rs, int s) { return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, wofs, s * 1024, 0)); };

    const int prt8 = ((tid >> 4) & 3) * 8;
    const int q0 = wave * 3;
    int paddr[3], maddr[3], myx[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int pidx = (q0 + q) * 32 + r;
        const int pc = pidx < MPIX ? pidx : MPIX - 1;
        const int my = pc / MW, mx = pc % MW;
        paddr[q] = (my * IW + mx) * PS + h * 16;
        maddr[q] = (pidx < MT * 32 ? pidx : 0) * PS + h * 16;
        myx[q] = (my << 8) | mx;
    }
    // conv2: this wave's two output column tiles (2 rows x 16 px each): lane -> (row, col) inside the tile
    const int orow = wave * 4 + (r >> 4), ocol = r & 15;  // column tile q covers rows wave*4 + 2q, +1
    const int oaddr = (orow * MW + ocol) * PS + h * 16;   // + q * 2 * MW * PS

    const int tiles_per_img = p.tiles_x * p.tiles_y;
    u32x4 preg[NPL];
    unsigned pf_mask = 0;
    unsigned pf_base = 0;  // byte offset of the patch origin from p.in (may wrap below zero: only in-image units are loaded)
    int pf_iy0 = 0, pf_ix0 = 0;
    bool pf_more = true;
    auto pf_setup = [&](int t) {
        const int b = t / tiles_per_img, tt = t % tiles_per_img;
        pf_iy0 = (tt / p.tiles_x) * TH - 2; pf_ix0 = (tt % p.tiles_x) * TW - 2;
        pf_base = (unsigned)((((ptrdiff_t)b * p.H * p.W + (ptrdiff_t)pf_iy0 * p.W + pf_ix0) * p.in_cs) * 2);
        pf_mask = 0;
    };
    auto pf_load = [&](auto ic) {
        constexpr int i = decltype(ic)::value;
        // unit u -> (pixel, 16-byte part): 16 consecutive lanes take the SAME part of 16 consecutive pixels (conflict-free
        // ds_write_b128 at the 80-byte pixel stride); recomputed per load: seven registers are worth more than seven divisions
        const int u = tid + NTHR * i, pix = (u >> 6) * 16 + (u & 15), py = pix / IW, px = pix % IW;
        const int iy = pf_iy0 + py, ix = pf_ix0 + px;
        const bool ok = pf_more & (u < P_UNITS) & ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
        preg[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, ok ? (int)(pf_base + (unsigned)(((py * p.W + px) * p.in_cs + prt8) * 2)) : 0, 0, 0));
        pf_mask |= ok ? (1u << i) : 0u;
    };
    auto write_patch_unit = [&](auto ic) {
        constexpr int i = decltype(ic)::value;
        const int u = tid + NTHR * i;
        *reinterpret_cast<u32x4 *>(lds_p + ((u >> 6) * 16 + (u & 15)) * PS + ((u >> 4) & 3) * 16) = (pf_mask >> i) & 1u ? preg[i] : u32x4{0u, 0u, 0u, 0u};
    };
    auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

    int t = blockIdx.x;
    pf_setup(t);
    static_for<NPL>(pf_load);
    u32x4 wA[NFRAG];  // the A fragments of the conv that runs next (conv1 here), recycled in place from then on
    static_for<NFRAG>([&](auto sc) { wA[decltype(sc)::value] = wfrag(rs_w1, decltype(sc)::value); });
    if (tid < 32) { lds_b[tid] = p.b1[tid]; lds_b[32 + tid] = p.b2[tid]; }
    static_for<NPL>(write_patch_unit);
    __syncthreads();

    for (; t < p.ntiles; t += gridDim.x) {
        const int b = t / tiles_per_img, tt = t % tiles_per_img;
        const int oy0 = (tt / p.tiles_x) * TH, ox0 = (tt % p.tiles_x) * TW;
        const int tn = t + gridDim.x;
        pf_more = tn < p.ntiles;
        pf_setup(pf_more ? tn : t);

        f32x16 acc2[2];
        // ================= conv1 + bn1 + relu -> intermediate tile (LDS, bf16) =================
        auto conv1_phase = [&](auto nqc) {
            constexpr int NQ = decltype(nqc)::value;
            f32x16 acc[NQ];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 bv = *reinterpret_cast<const float4 *>(lds_b + 8 * g + 4 * h);
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    acc[q][4 * g + 0] = bv.x; acc[q][4 * g + 1] = bv.y; acc[q][4 * g + 2] = bv.z; acc[q][4 * g + 3] = bv.w;
                }
            }
            u32x4 fb[2][NQ];
            auto ld1 = [&](int st, int buf) {
                const int tap = st >> 1, kk = st & 1, ky = tap / 3, kx = tap % 3;
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                    fb[buf][q] = *reinterpret_cast<const u32x4 *>(lds_p + paddr[q] + (ky * IW + kx) * PS + kk * 32);
            };
            ld1(0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, NQ, 0);
            static_for<NFRAG>([&](auto ic) {
                constexpr int st = decltype(ic)::value;
                if (st + 1 < NFRAG) {
                    ld1(st + 1, (st + 1) & 1);
                    __builtin_amdgcn_sched_group_barrier(0x100, NQ, 0);
                }
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                    acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wA[st]), __builtin_bit_cast(bf16x8, fb[st & 1][q]), acc[q], 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x8, NQ, 0);
                wA[st] = wfrag(rs_w2, st);  // conv1 is done with fragment st: conv2's takes its registers
                __builtin_amdgcn_sched_group_barrier(0x20, 1, 0);  // (pinned BEHIND the MFMAs: hoisted, the 18 loads would hold 72 more registers)
            });
            // conv2's accumulators start as bn2 shift + residual: the patch centre read straight in the MFMA C layout (the lane's
            // pixel, couts 8g + 4h .. +3: one ds_read_b64 per group) -- no identity fragments to keep in registers
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const char *xc = lds_p + ((orow + 2 * q + 2) * IW + ocol + 2) * PS + h * 8;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 bv = *reinterpret_cast<const float4 *>(lds_b + 32 + 8 * g + 4 * h);
                    const uint2 xv = *reinterpret_cast<const uint2 *>(xc + g * 16);
                    acc2[q][4 * g + 0] = bv.x + __builtin_bit_cast(float, xv.x << 16); acc2[q][4 * g + 1] = bv.y + __builtin_bit_cast(float, xv.x & 0xffff0000u);
                    acc2[q][4 * g + 2] = bv.z + __builtin_bit_cast(float, xv.y << 16); acc2[q][4 * g + 3] = bv.w + __builtin_bit_cast(float, xv.y & 0xffff0000u);
                }
            }
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int gy = oy0 - 1 + (myx[q] >> 8), gx = ox0 - 1 + (myx[q] & 255);
                const bool outside = ((unsigned)gy >= (unsigned)p.H) | ((unsigned)gx >= (unsigned)p.W);  // conv2 zero-pads the feature map
                u32x4 o[2];
                pack_rows16(acc[q], o);
                *reinterpret_cast<u32x4 *>(lds_m + maddr[q]) = outside ? u32x4{0u, 0u, 0u, 0u} : o[0];
                *reinterpret_cast<u32x4 *>(lds_m + maddr[q] + 32) = outside ? u32x4{0u, 0u, 0u, 0u} : o[1];
            }
        };
        if (wave < 3) conv1_phase(std::integral_constant<int, 3>{});
        else conv1_phase(std::integral_constant<int, 2>{});
        lds_barrier();  // intermediate tile complete; every wave is done with the patch

        // ================= conv2 + bn2 (+ residual already in acc2) =================
        {
            u32x4 fb[2][2];
            auto ld2 = [&](int st, int buf) {
                const int tap = st >> 1, kk = st & 1, ky = tap / 3, kx = tap % 3;
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    fb[buf][q] = *reinterpret_cast<const u32x4 *>(lds_m + oaddr + (2 * q * MW) * PS + (ky * MW + kx) * PS + kk * 32);
            };
            ld2(0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            static_for<NFRAG>([&](auto ic) {
                constexpr int st = decltype(ic)::value;
                if (st + 1 < NFRAG) {
                    ld2(st + 1, (st + 1) & 1);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                }
                // the next tile's patch: fetched during the first k-steps (conv1's accumulators are dead, registers are free) and
                // written over the old patch -- free during conv2 -- in the last ones; the CU's other workgroup covers the wait
                if constexpr (st >= NFRAG - NPL) {
                    write_patch_unit(std::integral_constant<int, st - (NFRAG - NPL)>{});
                    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                }
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    acc2[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wA[st]), __builtin_bit_cast(bf16x8, fb[st & 1][q]), acc2[q], 0, 0, 0);
                __builtin_amdgcn_sched_group_barrier(0x8, 2, 0);
                wA[st] = wfrag(rs_w1, st);  // the next tile's conv1 fragment
                if constexpr (st < NPL) pf_load(ic);
                __builtin_amdgcn_sched_group_barrier(0x20, st < NPL ? 2 : 1, 0);
            });
        }
        // ---- epilogue: ReLU, bf16, 16 contiguous bytes per lane straight to HBM
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int oy = oy0 + orow + 2 * q, ox = ox0 + ocol;
            u32x4 o[2];
            pack_rows16(acc2[q], o);
            if ((oy < p.H) & (ox < p.W)) {
                bf16_raw *dst = p.out + (((ptrdiff_t)b * p.H + oy) * p.W + ox) * p.out_cs + 8 * h;
                *reinterpret_cast<u32x4 *>(dst) = o[0];
                *reinterpret_cast<u32x4 *>(dst + 16) = o[1];
            }
        }
        lds_barrier();  // every wave is done with the intermediate tile; the next patch is visible
    }
#ifndef HH_NO_CLK
    if (p.clk && tid == 0) atomicMax(p.clk + 1, wall_clock64());
#endif
}

hipError_t bb_thin_init()
{
    return hipFuncSetAttribute(reinterpret_cast<const void *>(bb_thin_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bb_thin_lds_bytes());
}

hipError_t bb_thin_launch(BBParams p, int num_cus, hipStream_t s)
{
    p.tiles_x = (p.W + TW - 1) / TW;
    p.tiles_y = (p.H + TH - 1) / TH;
    p.ntiles = p.B * p.tiles_x * p.tiles_y;
    const int grid = p.ntiles < 2 * num_cus ? p.ntiles : 2 * num_cus;  // two workgroups per CU
    HH_LAUNCH(bb_thin_kernel, dim3(grid), dim3(NTHR), bb_thin_lds_bytes(), s, p);
    return hipGetLastError();
}
