// Fused BasicBlock for the 128-channel branch:   out = relu(bn2(conv2(relu(bn1(conv1(x))))) + x)
// -- /root/reference/src/keypoints/architectures/hrnet.py:108-124 -- in ONE kernel.
//
// At 128 channels the layer is bound by streaming weights: a 3x3 conv has 295 KB of them and only 128 pixels x 128 couts of
// work per CU at batch 32 (32x32 maps), so layer by layer every launch is a cold start (patch + first weight chunk) followed
// by four chunks of 64 KB staged per 2.3 k cycles of MFMAs, ~17-18 us where the MFMAs need ~5.  Fused, ONE launch streams
// both convs' weights back to back through a double-buffered LDS ring while the 10x18 intermediate tile stays in LDS:
//
//   workgroup = 512 threads (8 waves), one per CU, output tile 8x16 px (256 tiles at batch 32: one per CU)
//   K chunks of 16 input channels: 8 per conv.  LDS: two buffers of {36.9 KB weight chunk [tap][2][128 couts][8] +
//   11.5 KB slice of the 12x20 input patch (16 channels)} + the 192-slot intermediate tile (272 B per pixel) = 150 KB.
//   Phase p computes chunk p from buffer p&1 while the chunk p+1 it loaded one phase earlier goes registers -> buffer (p+1)&1
//   (ds_writes between the MFMAs) and the global loads of chunk p+2 are issued: one LDS-only barrier per phase, no exposed
//   load or write after the first chunk.
//   wave w: cout tile ct = w & 3, part = w >> 2; conv1: 3 of the 6 intermediate column tiles, conv2: 2 of the 4 output ones.
//   The residual is re-read from global (L2) into the accumulator layout before conv2.
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

constexpr int C = 128;
constexpr int TH = 8, TW = 16;
constexpr int MH = TH + 2, MW = TW + 2;
constexpr int IH = TH + 4, IW = TW + 4;
constexpr int KC = 16;                        // input channels per chunk
constexpr int NCH = C / KC;                   // 8 chunks per conv
constexpr int PSM = C * 2 + 16;               // 272 bytes per intermediate pixel (17 sixteen-byte slots)
constexpr int PSP = KC * 2 + 16;              // 48 bytes per patch-slice pixel (3 slots)
constexpr int MPIX = MH * MW;                 // 180 -> 6 column tiles
constexpr int MT = (MPIX + 31) / 32;
constexpr int NTHR = 512;
constexpr int W_UNITS = 9 * 2 * C;            // 2304 sixteen-byte units of a weight chunk
constexpr int P_UNITS = IH * IW * 2;          // 480 units of a patch slice
constexpr int NWL = (W_UNITS + NTHR - 1) / NTHR;   // 5
constexpr int W_BYTES = W_UNITS * 16;         // 36864
constexpr int P_BYTES = IH * IW * PSP;        // 11520
constexpr int BUF_BYTES = W_BYTES + P_BYTES;  // 48384
constexpr int MID_BYTES = MT * 32 * PSM;      // 52224
}  // namespace

size_t bb128_lds_bytes() { return MID_BYTES + 2 * BUF_BYTES + 2 * C * 4; }

__global__ __launch_bounds__(NTHR, 1) void bb128_fused_kernel(const BBParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *lds_m = smem;
    char *lds_buf = smem + MID_BYTES;  // [2][weights | patch slice]
    float *lds_b = reinterpret_cast<float *>(lds_buf + 2 * BUF_BYTES);

    const int tid = threadIdx.x;
#ifndef HH_NO_CLK
    if (p.clk && tid == 0) atomicMin(p.clk, wall_clock64());
#endif
    const int wave = tid >> 6, lane = tid & 63;
    const int r = lane & 31, h = lane >> 5;
    const int ct = wave & 3, part = wave >> 2;

    if (tid < C) { lds_b[tid] = p.b1[tid]; lds_b[C + tid] = p.b2[tid]; }

    // ---- tile-invariant geometry
    // this thread's unit of a patch slice (threads 0..479): pixel tid >> 1, 16-byte part tid & 1
    const int pu_pix = tid >> 1, pu_part = tid & 1, pu_py = pu_pix / IW, pu_px = pu_pix % IW;
    const bool pu_on = tid < P_UNITS;
    int paddr[3], maddr[3], myx[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int pidx = (part * 3 + q) * 32 + r;
        const int pc = pidx < MPIX ? pidx : MPIX - 1;
        const int my = pc / MW, mx = pc % MW;
        paddr[q] = W_BYTES + (my * IW + mx) * PSP + h * 16;
        maddr[q] = pidx * PSM + ct * 64 + h * 16;
        myx[q] = (my << 8) | mx;
    }
    int oaddr[2], oyx[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int oidx = (part * 2 + q) * 32 + r, oy = oidx / TW, ox = oidx % TW;
        oaddr[q] = (oy * MW + ox) * PSM + h * 16;
        oyx[q] = (oy << 8) | ox;
    }
    auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    const auto rs_w1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_raw *>(p.w1), 0, NCH * W_BYTES, 0x00020000);
    const auto rs_w2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_raw *>(p.w2), 0, NCH * W_BYTES, 0x00020000);
    const auto rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_raw *>(p.in), 0, 0x7fffffff, 0x00020000);
    const unsigned woff = (unsigned)tid * 16u;  // byte offset of this thread's unit inside a round of 512 units
    const unsigned woff_last = (unsigned)(tid + (NWL - 1) * NTHR < W_UNITS ? (tid + (NWL - 1) * NTHR) * 16 : 0);  // the half-idle last round

    const int tiles_per_img = p.tiles_x * p.tiles_y;
    for (int t = blockIdx.x; t < p.ntiles; t += gridDim.x) {
        const int b = t / tiles_per_img, tt = t % tiles_per_img;
        const int oy0 = (tt / p.tiles_x) * TH, ox0 = (tt % p.tiles_x) * TW;
        // patch-slice source of this thread: pixel (oy0 - 2 + pu_py, ox0 - 2 + pu_px), zero outside the image
        const int siy = oy0 - 2 + pu_py, six = ox0 - 2 + pu_px;
        const bool pu_ok = pu_on & ((unsigned)siy < (unsigned)p.H) & ((unsigned)six < (unsigned)p.W);
        const unsigned pu_off = pu_ok ? (unsigned)(((((size_t)b * p.H + siy) * p.W + six) * p.in_cs + pu_part * 8) * 2) : 0u;  // bytes from p.in

        // chunk index cidx = 0..15: conv1 chunks 0..7 (weights w1 + a patch slice), conv2 chunks 8..15 (weights w2)
        u32x4 wreg[2][NWL], preg[2];
        // Buffer loads (wave-uniform descriptor + ONE 32-bit per-thread offset + a scalar offset per chunk): with flat loads the
        // compiler kept a 64-bit pointer per (chunk, load), spilled them, and every scratch reload waits on vmcnt(0), i.e. on the
        // very prefetch it sits next to.
        auto load_chunk = [&](auto setc, int cidx) {
            constexpr int set = decltype(setc)::value;
            const int soff = (cidx & (NCH - 1)) * W_BYTES;
            static_for<NWL>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                const auto rs = cidx < NCH ? rs_w1 : rs_w2;
                if constexpr ((i + 1) * NTHR <= W_UNITS) wreg[set][i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)woff, soff + i * (NTHR * 16), 0));
                else wreg[set][i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)woff_last, soff, 0));
            });
            preg[set] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, (int)pu_off, cidx < NCH ? cidx * (KC * 2) : 0, 0));
        };
        auto write_chunk_unit = [&](auto setc, auto ic, int cidx) {  // one 16-byte unit of chunk cidx: registers -> buffer cidx & 1
            constexpr int set = decltype(setc)::value, i = decltype(ic)::value;
            char *dst = lds_buf + (cidx & 1) * BUF_BYTES;
            if constexpr (i < NWL) {
                const int u = tid + NTHR * i;
                if (u < W_UNITS) reinterpret_cast<u32x4 *>(dst)[u] = wreg[set][i];
            } else {
                if (pu_on & (cidx < NCH)) *reinterpret_cast<u32x4 *>(dst + W_BYTES + pu_pix * PSP + pu_part * 16) = pu_ok ? preg[set] : u32x4{0u, 0u, 0u, 0u};
            }
        };
        load_chunk(std::integral_constant<int, 0>{}, 0);
        load_chunk(std::integral_constant<int, 1>{}, 1);
        static_for<NWL + 1>([&](auto ic) { write_chunk_unit(std::integral_constant<int, 0>{}, ic, 0); });
        __syncthreads();  // chunk 0 (and the biases) visible; also separates this tile from the previous one's LDS reads

        // one K chunk of MFMAs for NQ column tiles; `base[q]` = byte offset of the lane's pixel in the operand image,
        // `taps(ky, kx)` = byte offset of a tap, `chan` = byte offset of the chunk's channels inside a pixel
        // While it runs: chunk cidx+1 goes registers (set (cidx+1)&1) -> LDS, then chunk cidx+2 is fetched into set cidx&1.
#define BB128_PHASE(NQ, CIDX, ACC, BIMG, BASE, ROWSTRIDE, PSTRIDE, CHAN)                                                     \
    do {                                                                                                                      \
        constexpr int cidx_ = (CIDX);                                                                                         \
        const char *wimg_ = lds_buf + (cidx_ & 1) * BUF_BYTES;                                                                \
        const char *bimg_ = (BIMG);                                                                                           \
        u32x4 fa_[2], fb_[2][NQ];                                                                                             \
        auto ld_ = [&](int tap, int buf) {                                                                                    \
            const int ky = tap / 3, kx = tap % 3;                                                                             \
            fa_[buf] = *reinterpret_cast<const u32x4 *>(wimg_ + ((tap * 2 + h) * C + ct * 32 + r) * 16);                      \
            _Pragma("unroll") for (int q = 0; q < NQ; ++q)                                                                    \
                fb_[buf][q] = *reinterpret_cast<const u32x4 *>(bimg_ + BASE[q] + (ky * (ROWSTRIDE) + kx) * (PSTRIDE) + (CHAN)); \
        };                                                                                                                    \
        ld_(0, 0);                                                                                                            \
        __builtin_amdgcn_sched_group_barrier(0x100, NQ + 1, 0);                                                               \
        static_for<9>([&](auto ic) {                                                                                          \
            constexpr int st = decltype(ic)::value;                                                                           \
            if (st + 1 < 9) {                                                                                                 \
                ld_(st + 1, (st + 1) & 1);                                                                                    \
                __builtin_amdgcn_sched_group_barrier(0x100, NQ + 1, 0);                                                       \
            }                                                                                                                 \
            if constexpr (cidx_ + 1 < 2 * NCH && st < NWL + 1) {                                                              \
                write_chunk_unit(std::integral_constant<int, (cidx_ + 1) & 1>{}, ic, cidx_ + 1);                              \
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                                                            \
            }                                                                                                                 \
            _Pragma("unroll") for (int q = 0; q < NQ; ++q)                                                                    \
                ACC[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa_[st & 1]),                     \
                                                                 __builtin_bit_cast(bf16x8, fb_[st & 1][q]), ACC[q], 0, 0, 0); \
            __builtin_amdgcn_sched_group_barrier(0x8, NQ, 0);                                                                 \
        });                                                                                                                   \
    } while (0)
        // One K chunk of MFMAs for NQ column tiles: BASE[q] = byte offset of the lane's pixel in the operand image BIMG, taps
        // at (ky * ROWSTRIDE + kx) * PSTRIDE, the chunk's channels at CHAN inside a pixel.  While it runs, chunk CIDX+1 goes
        // registers (set (CIDX+1)&1) -> LDS; chunk CIDX+2 was requested just before the phase (set CIDX&1 is free by then).

        // ================= conv1 + bn1 + relu -> intermediate tile (LDS, bf16) =================
        {
            f32x16 acc[3];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 bv = *reinterpret_cast<const float4 *>(lds_b + ct * 32 + 8 * g + 4 * h);
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    acc[q][4 * g + 0] = bv.x; acc[q][4 * g + 1] = bv.y; acc[q][4 * g + 2] = bv.z; acc[q][4 * g + 3] = bv.w;
                }
            }
            static_for<NCH>([&](auto cc) {
                constexpr int c = decltype(cc)::value;
                // register set c & 1 is free (its chunk c went to LDS a phase ago): fetch chunk c+2 now, a whole phase ahead of its write
                if constexpr (c + 2 < 2 * NCH) load_chunk(std::integral_constant<int, c & 1>{}, c + 2);
                BB128_PHASE(3, c, acc, lds_buf + (c & 1) * BUF_BYTES, paddr, IW, PSP, 0);
                if constexpr (c + 1 < NCH) lds_barrier();
            });
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int gy = oy0 - 1 + (myx[q] >> 8), gx = ox0 - 1 + (myx[q] & 255);
                const bool outside = ((unsigned)gy >= (unsigned)p.H) | ((unsigned)gx >= (unsigned)p.W);  // conv2 zero-pads the feature map
                u32x4 o[2];
                pack_rows16(acc[q], o);
                *reinterpret_cast<u32x4 *>(lds_m + maddr[q]) = outside ? u32x4{0u, 0u, 0u, 0u} : o[0];
                *reinterpret_cast<u32x4 *>(lds_m + maddr[q] + 32) = outside ? u32x4{0u, 0u, 0u, 0u} : o[1];
            }
        }
        // conv2's accumulators start as bn2 shift + residual (x re-read from global / L2: 16 bytes = couts 16m+8h..+7 of the
        // lane's pixel, exchanged into the MFMA C layout)
        f32x16 acc2[2];
        {
            u32x4 rv[2][2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int oy = oy0 + (oyx[q] >> 8), ox = ox0 + (oyx[q] & 255);
                const bool valid = (oy < p.H) & (ox < p.W);
                const bf16_raw *src = p.in + (((ptrdiff_t)b * p.H + (valid ? oy : 0)) * p.W + (valid ? ox : 0)) * p.in_cs + ct * 32 + 8 * h;
#pragma unroll
                for (int m = 0; m < 2; ++m) rv[q][m] = *reinterpret_cast<const u32x4 *>(src + 16 * m);
            }
            lds_barrier();  // the intermediate tile is complete; chunk 8 (written during phase 7) is visible
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    auto s0 = __builtin_amdgcn_permlane32_swap(rv[q][m][0], rv[q][m][2], false, false);
                    auto s1 = __builtin_amdgcn_permlane32_swap(rv[q][m][1], rv[q][m][3], false, false);
                    const float4 ba = *reinterpret_cast<const float4 *>(lds_b + C + ct * 32 + 16 * m + 4 * h);
                    const float4 bb = *reinterpret_cast<const float4 *>(lds_b + C + ct * 32 + 16 * m + 8 + 4 * h);
                    acc2[q][8 * m + 0] = ba.x + bf16_lo(s0[0]); acc2[q][8 * m + 1] = ba.y + bf16_hi(s0[0]);
                    acc2[q][8 * m + 2] = ba.z + bf16_lo(s1[0]); acc2[q][8 * m + 3] = ba.w + bf16_hi(s1[0]);
                    acc2[q][8 * m + 4] = bb.x + bf16_lo(s0[1]); acc2[q][8 * m + 5] = bb.y + bf16_hi(s0[1]);
                    acc2[q][8 * m + 6] = bb.z + bf16_lo(s1[1]); acc2[q][8 * m + 7] = bb.w + bf16_hi(s1[1]);
                }
        }
        // ================= conv2 + bn2 (+ residual already in acc2) =================
        static_for<NCH>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            if constexpr (NCH + c + 2 < 2 * NCH) load_chunk(std::integral_constant<int, c & 1>{}, NCH + c + 2);
            BB128_PHASE(2, NCH + c, acc2, lds_m, oaddr, MW, PSM, c * 32);
            if constexpr (c + 1 < NCH) lds_barrier();
        });
        // ---- epilogue: ReLU, bf16, 16 contiguous bytes per lane straight to HBM
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int oy = oy0 + (oyx[q] >> 8), ox = ox0 + (oyx[q] & 255);
            u32x4 o[2];
            pack_rows16(acc2[q], o);
            if ((oy < p.H) & (ox < p.W)) {
                bf16_raw *dst = p.out + (((ptrdiff_t)b * p.H + oy) * p.W + ox) * p.out_cs + ct * 32 + 8 * h;
                *reinterpret_cast<u32x4 *>(dst) = o[0];
                *reinterpret_cast<u32x4 *>(dst + 16) = o[1];
            }
        }
        __syncthreads();  // the next tile's prologue overwrites buffer 0 and the intermediate tile
    }
#ifndef HH_NO_CLK
    if (p.clk && tid == 0) atomicMax(p.clk + 1, wall_clock64());
#endif
}

hipError_t bb128_fused_init()
{
    return hipFuncSetAttribute(reinterpret_cast<const void *>(bb128_fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)bb128_lds_bytes());
}

hipError_t bb128_fused_launch(BBParams p, int num_cus, hipStream_t s)
{
    p.tiles_x = (p.W + TW - 1) / TW;
    p.tiles_y = (p.H + TH - 1) / TH;
    p.ntiles = p.B * p.tiles_x * p.tiles_y;
    const int grid = p.ntiles < num_cus ? p.ntiles : num_cus;
    HH_LAUNCH(bb128_fused_kernel, dim3(grid), dim3(NTHR), bb128_lds_bytes(), s, p);
    return hipGetLastError();
}
