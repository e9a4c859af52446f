// The whole stem in one kernel (/root/reference/src/keypoints/architectures/hrnet.py:354-358,378-384):
//   conv3x3 s2 (3 -> 64) + BN + ReLU  ->  conv3x3 s2 (64 -> 64) + BN + ReLU,   fp32 NCHW images in, bf16 NHWC at 1/4 resolution out.
// Why: run as two launches the 64-channel half-resolution intermediate (268 MB at B = 32, 512x512) is written by stem_conv.hip
// (105 us, HBM-bound) and read back by a stride-2 conv_mfma launch (119 us); fused it lives in LDS and the stem moves 100 MB in
// and 67 MB out.
//
// One workgroup (256 threads, persistent over tiles) = a 2 x 32 tile of the final map:
//   patch   (4*2+3) x (4*32+3) x 3 input values as bf16 in LDS (fp32 global -> registers during the previous tile's conv1 ->
//           LDS during its conv2; outside the image the buffer load returns 0 = conv1's padding);
//   conv1   the 5 x 65 intermediate pixels the tile needs, flattened into 11 column tiles of 32 lanes (3, 3, 3, 2 per wave);
//           each lane gathers its pixel's 27 taps (K padded to 32 = two k-steps) from the patch as in stem_conv.hip;
//           BN shift + ReLU, bf16, into LDS in two column-parity planes (even / odd intermediate columns) so that conv2's
//           stride-2 reads (column 2r + kx for lane r) touch consecutive pixels of one plane: conflict-free ds_read_b128.
//           Intermediate pixels outside the half-resolution image are conv2's zero padding;
//   conv2   wave w = (output row w >> 1, 32-cout tile w & 1): 36 k-steps, its 36 weight fragments stay in 144 VGPRs for the life
//           of the workgroup (no weight reads), pixel fragments by hand-pinned asm reads three steps ahead; ReLU, 16-byte stores.
#include "kernels.h"

#include <utility>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef short i16x2 __attribute__((ext_vector_type(2)));

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
__device__ __forceinline__ unsigned pack_relu_bf16x2(float a, float b)
{
    f32x2 f = {a, b};
    const i16x2 v = __builtin_bit_cast(i16x2, __builtin_convertvector(f, bf16x2));
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(v, i16x2{0, 0}));
}
// 32 couts of one pixel (lanes (r,0): couts 8g..8g+3, lanes (r,1): 8g+4..8g+7 in acc[4g..4g+3]) -> for m = 0,1 the 16 bytes
// (bf16, ReLU applied) of couts 16m+8h .. 16m+8h+7 of this lane's pixel
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
// LDS fragment reads pinned by hand (see basicblock_fused_pc.hip)
template <int OFF>
__device__ __forceinline__ u32x4 lds_read_async(int addr)
{
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
    return v;
}
template <int N>
__device__ __forceinline__ void lds_wait(u32x4 &v)
{
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(v) : "n"(N));
}

constexpr int T2H = 2, T2W = 32;                            // tile of the final (1/4 resolution) map
constexpr int MR = 2 * T2H + 1, MC = 2 * T2W + 1;           // 5 x 65 intermediate (1/2 resolution) pixels
constexpr int MPIX = MR * MC, NQ = (MPIX + 31) / 32;        // 325 -> 11 column tiles
constexpr int PRW = 4 * T2H + 3, PCW = 4 * T2W + 3;         // 11 x 131 input patch per channel
constexpr int PLANE = PRW * PCW, NVAL = 3 * PLANE;          // 1441, 4323 (+1 zero slot for the padded taps)
constexpr int NLD = (NVAL + 255) / 256;                     // 17 loads per thread
constexpr int PS2 = 144, HALF = (MC + 1) / 2;               // intermediate pixel stride (128 B + 16: odd number of 16-B slots), 33 pixels per parity plane row
constexpr int MID_BYTES = MR * 2 * HALF * PS2;              // 47,520
constexpr int PATCH_BYTES = ((NVAL + 1) * 2 + 15) / 16 * 16;
constexpr int OFF_PATCH = MID_BYTES, OFF_BIAS = OFF_PATCH + PATCH_BYTES, LDS_BYTES = OFF_BIAS + 2 * 64 * 4;
constexpr int RD = 3, NFB = RD + 1;
}  // namespace

__global__ __launch_bounds__(256, 1) void stem_fused_kernel(const StemFusedParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned short *patch = reinterpret_cast<unsigned short *>(smem + OFF_PATCH);
    const int tid = threadIdx.x;
#ifndef HH_NO_CLK
    if (p.clk && tid == 0) atomicMin(p.clk, wall_clock64());
#endif
    const int lds0 = (int)(size_t)(__attribute__((address_space(3))) char *)smem;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int H1 = p.H >> 1, W1 = p.W >> 1, H2 = p.H >> 2, W2 = p.W >> 2;
    // one buffer descriptor per IMAGE (32-bit offsets inside it): no limit on the batch
    const size_t img_elems = (size_t)3 * p.H * p.W, out_elems = (size_t)H2 * W2 * p.out_cs;
    constexpr unsigned OOB = 0x80000000u;

    // ---- weights: conv1's four A fragments [cout tile][k-step] as stem_conv.hip, conv2's 36 fragments of this wave's cout tile
    const int row2 = wave >> 1, ct2 = wave & 1;
    u32x4 a1[2][2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) a1[ct][kk] = reinterpret_cast<const u32x4 *>(p.w1)[((ct * 2 + kk) * 2 + h) * 32 + r];
    u32x4 w2r[36];
    {
        const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_raw *>(p.w2), 0, 9 * 8 * 64 * 16, 0x00020000);
        static_for<36>([&](auto fc) {  // packed [tap][cin / 8][64 couts][8]: fragment (tap, kk) = rows (kk*2 + h) of that tap
            constexpr int f = decltype(fc)::value, tap = f >> 2, kk = f & 3;
            w2r[f] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, (((tap * 8 + kk * 2 + h) * 64) + ct2 * 32 + r) * 16, 0, 0));
        });
    }
    if (tid < 64) {
        reinterpret_cast<float *>(smem + OFF_BIAS)[tid] = p.b1[tid];
        reinterpret_cast<float *>(smem + OFF_BIAS)[64 + tid] = p.b2[tid];
    }
    if (tid == 0) patch[NVAL] = 0;  // the zero slot of the padded taps 27..31
    auto bias_acc = [&](int off) {
        f32x16 b0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 bv = *reinterpret_cast<const float4 *>(smem + OFF_BIAS + (off + 8 * q + 4 * h) * 4);
            b0[4 * q + 0] = bv.x; b0[4 * q + 1] = bv.y; b0[4 * q + 2] = bv.z; b0[4 * q + 3] = bv.w;
        }
        return b0;
    };

    // ---- tiles
    const int tiles_x = (W2 + T2W - 1) / T2W, tiles_y = (H2 + T2H - 1) / T2H, ntiles = p.B * tiles_x * tiles_y;
    struct Geom { int b, oy0, ox0; };
    auto geom = [&](int t) {
        const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y;
        return Geom{t / (tiles_x * tiles_y), ty * T2H, tx * T2W};
    };
    // patch staging: value i = tid + 256 it -> (c, py, px); input pixel (4 oy0 - 3 + py, 4 ox0 - 3 + px)
    float pv[NLD];
    auto patch_load = [&](int t) {
        const bool on = t < ntiles;
        const Geom g = geom(on ? t : 0);
        const auto rs_img = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.images + (size_t)g.b * img_elems), 0, (int)(img_elems * 4), 0x00020000);
        const int iy0 = 4 * g.oy0 - 3, ix0 = 4 * g.ox0 - 3;
#pragma unroll
        for (int it = 0; it < NLD; ++it) {
            const int i = tid + 256 * it;
            const int c = i / PLANE, rem = i - c * PLANE, py = rem / PCW, px = rem - py * PCW;
            const int iy = iy0 + py, ix = ix0 + px;
            const bool ok = on & (i < NVAL) & ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);
            const unsigned voff = ok ? (unsigned)(((c * p.H + iy) * p.W + ix) * 4) : OOB;  // outside: 0 = conv1's padding
            pv[it] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_img, (int)voff, 0, 0));
        }
    };
    auto patch_write = [&]() {
#pragma unroll
        for (int it = 0; it < NLD; ++it) {
            const int i = tid + 256 * it;
            if (it + 1 < NLD || i < NVAL) patch[i] = __builtin_bit_cast(unsigned short, (__bf16)pv[it]);
        }
    };
    // conv1 gather: tap t = kk*16 + 8h + j -> (c, ky, kx) = (t / 9, (t % 9) / 3, t % 3); taps >= 27 read the zero slot
    int off[2][8];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int t = kk * 16 + 8 * h + j;
            off[kk][j] = t < 27 ? (t / 9) * PLANE + ((t % 9) / 3) * PCW + (t % 3) : -1;
        }

    int t = blockIdx.x;
    patch_load(t);
    patch_write();
    __syncthreads();

    for (; t < ntiles; t += gridDim.x) {
        const Geom g = geom(t);
        patch_load(t + (int)gridDim.x);  // the next tile's input: in flight during conv1, to LDS during conv2

        // ================= conv1 + bn1 + relu -> intermediate tile (LDS, bf16, column-parity planes) =================
        // (one wave per SIMD: nothing hides a latency for it, so all gathers of the wave's column tiles are issued before the first MFMA)
        constexpr int NQW = (NQ + 3) / 4;
        u32x4 bf[NQW][2];
#pragma unroll
        for (int qi = 0; qi < NQW; ++qi) {
            const int q = wave + 4 * qi;
            const int m = q * 32 + r, mc = m < MPIX ? m : MPIX - 1;  // (q >= NQ: clamped, never written)
            const int my = mc / MC, mx = mc - my * MC;
            const int base = (2 * my) * PCW + 2 * mx;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                unsigned short v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = patch[off[kk][j] >= 0 ? base + off[kk][j] : NVAL];
                bf[qi][kk] = u32x4{(unsigned)v[0] | ((unsigned)v[1] << 16), (unsigned)v[2] | ((unsigned)v[3] << 16),
                                   (unsigned)v[4] | ((unsigned)v[5] << 16), (unsigned)v[6] | ((unsigned)v[7] << 16)};
            }
        }
        const f32x16 bias1[2] = {bias_acc(0), bias_acc(32)};
#pragma unroll
        for (int qi = 0; qi < NQW; ++qi) {
            const int q = wave + 4 * qi;
            if (q >= NQ) break;  // (wave-uniform)
            const int m = q * 32 + r, mc = m < MPIX ? m : MPIX - 1;
            const int my = mc / MC, mx = mc - my * MC;
            const int gy = 2 * g.oy0 - 1 + my, gx = 2 * g.ox0 - 1 + mx;
            const bool inside = ((unsigned)gy < (unsigned)H1) & ((unsigned)gx < (unsigned)W1);  // else conv2's zero padding
            char *dst = smem + ((my * 2 + (mx & 1)) * HALF + (mx >> 1)) * PS2 + h * 16;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                f32x16 acc = bias1[ct];
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a1[ct][kk]), __builtin_bit_cast(bf16x8, bf[qi][kk]), acc, 0, 0, 0);
                u32x4 o[2];
                pack_rows16(acc, o);
                if (m < MPIX) {
                    *reinterpret_cast<u32x4 *>(dst + ct * 64) = inside ? o[0] : u32x4{0u, 0u, 0u, 0u};
                    *reinterpret_cast<u32x4 *>(dst + ct * 64 + 32) = inside ? o[1] : u32x4{0u, 0u, 0u, 0u};
                }
            }
        }
        __syncthreads();  // intermediate tile complete; the patch is free

        patch_write();

        // ================= conv2 + bn2 + relu: wave = (output row, cout tile) =================
        {
            f32x16 acc = bias_acc(64 + ct2 * 32);
            const int ma = lds0 + (row2 * 4 * HALF + r) * PS2 + h * 16;  // intermediate row 2 row2 (+ ky), plane row pair, pixel r (+ kx >> 1)
            u32x4 fb[NFB];
            auto ldb = [&](auto sc, int buf) {  // step s = tap * 4 + kk: intermediate pixel (2 row2 + ky, 2 r + kx), channels 16 kk + 8 h ..
                constexpr int s = decltype(sc)::value, tap = s >> 2, kk = s & 3, ky = tap / 3, kx = tap % 3;
                fb[buf] = lds_read_async<((ky * 2 + (kx & 1)) * HALF + (kx >> 1)) * PS2 + kk * 32>(ma);
            };
            static_for<RD>([&](auto sc) { ldb(sc, decltype(sc)::value); });
            static_for<36>([&](auto sc) {
                constexpr int s = decltype(sc)::value;
                if constexpr (s + RD < 36) ldb(std::integral_constant<int, s + RD>{}, (s + RD) % NFB);
                lds_wait<(35 - s < RD ? 35 - s : RD)>(fb[s % NFB]);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, w2r[s]), __builtin_bit_cast(bf16x8, fb[s % NFB]), acc, 0, 0, 0);
            });
            u32x4 o[2];
            pack_rows16(acc, o);
            const int oy = g.oy0 + row2, ox = g.ox0 + r;
            const bool ok = (oy < H2) & (ox < W2);
            const auto rs_out = __builtin_amdgcn_make_buffer_rsrc(p.out + (size_t)g.b * out_elems, 0, (int)(out_elems * 2), 0x00020000);
            const unsigned voff = ok ? (unsigned)(((oy * W2 + ox) * p.out_cs + ct2 * 32 + 8 * h) * 2) : OOB;
            __builtin_amdgcn_raw_buffer_store_b128(o[0], rs_out, (int)voff, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(o[1], rs_out, (int)voff, 32, 0);
        }
        __syncthreads();  // the intermediate tile is free, the next patch is visible
    }
#ifndef HH_NO_CLK
    if (p.clk && tid == 0) atomicMax(p.clk + 1, wall_clock64());
#endif
}

hipError_t stem_fused_init()
{
    return hipFuncSetAttribute(reinterpret_cast<const void *>(stem_fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
}

bool stem_fused_supported(const StemFusedParams &p)
{
    return p.H % 4 == 0 && p.W % 4 == 0 && (size_t)3 * p.H * p.W * 4 < 0x7fffffffull && (size_t)(p.H / 4) * (p.W / 4) * p.out_cs * 2 < 0x7fffffffull;
}

hipError_t stem_fused_launch(const StemFusedParams &p, int num_cus, hipStream_t s)
{
    const int H2 = p.H >> 2, W2 = p.W >> 2;
    const int ntiles = p.B * ((H2 + T2H - 1) / T2H) * ((W2 + T2W - 1) / T2W);
    const int grid = ntiles < num_cus ? ntiles : num_cus;
    HH_LAUNCH(stem_fused_kernel, dim3(grid), dim3(256), LDS_BYTES, s, p);
    return hipGetLastError();
}
