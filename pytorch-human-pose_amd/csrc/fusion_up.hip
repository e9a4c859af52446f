// One output of a FusionLayer's low -> high half in ONE launch (hrnet.py:200-205,214-229):
//     out = [relu]( x_i + sum_{j > i} nearest_up_{2^(j-i)}( bn_ij( conv1x1_ij( x_j ) ) ) )
// Layer by layer that is up to three 1x1 conv launches at the low resolutions (12-15 us each: a handful of MFMAs behind a
// launch, a weight stage and a round trip of their outputs through HBM) and the upsample-add pass over the output (upadd_kernel).
// Here a persistent workgroup keeps the (small) weight sets of all sources in LDS and walks 32x32 tiles of the output:
//   phase 1  the 1x1 convs of the tile's 16x16 / 8x8 / 4x4 source pixels on MFMA (v_mfma_f32_32x32x16_bf16, A = weights, B = pixels:
//            a lane owns channels of ITS pixel); pixel fragments come straight from HBM / L2 in MFMA layout (NHWC: 16 bytes =
//            8 channels of one pixel), all loads of an item in flight before its first MFMA; results rounded to bf16 into LDS;
//   phase 2  every output pixel adds its base value and the sources' values at (y >> s, x >> s), applies ReLU, stores 16 bytes.
// The arithmetic is the layer-by-layer path's, operation for operation -- accumulators start at the folded BN shift, k runs
// over 16-channel steps in ascending order, one bf16 rounding per source (what conv_mfma stores), fp32 sum base + u_0 + u_1 + u_2
// in source order, ReLU, one bf16 rounding -- so the outputs are BIT-IDENTICAL to conv_mfma + upadd_kernel (HH_NO_FUSE_UP=1 keeps that
// plan; test_fusion_up_kernel_is_bit_identical).
#include "kernels.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

namespace {
__device__ __forceinline__ unsigned pack_bf16(float a, float b)
{
    f32x2 f = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2));
}
__device__ __forceinline__ float lo16(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float hi16(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }
constexpr int T = 32;  // output tile edge
}  // namespace

__global__ __launch_bounds__(256, 2) void fusion_up_kernel(const FuseUpParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
    // ---- LDS: [weights of source 0 | 1 | 2][u tile of source 0 | 1 | 2] (offsets from the launcher)
    for (int j = 0; j < p.nsrc; ++j) {
        const u32x4 *src = reinterpret_cast<const u32x4 *>(p.w[j]);
        u32x4 *dst = reinterpret_cast<u32x4 *>(smem + p.w_off[j]);
        for (int i = tid; i < p.w_units[j]; i += 256) dst[i] = src[i];
    }
    __syncthreads();
    const int c8n = p.C / 8, ctn = (p.C + 31) / 32;  // 16-byte channel groups / 32-cout tiles of the output
    const int tiles_x = (p.W + T - 1) / T, tiles_y = (p.H + T - 1) / T, ntiles = p.B * tiles_y * tiles_x;
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, b = t / (tiles_x * tiles_y);
        const int y0 = ty * T, x0 = tx * T;
        // ================= phase 1: u_j = bf16(shift_j + W_j x_j) on the tile's source pixels =================
        for (int j = 0; j < p.nsrc; ++j) {
            const int sh = p.shift[j], ts = T >> sh, np = ts * ts;  // source pixels of the tile: ts x ts
            const int Hs = p.H >> sh, Ws = p.W >> sh, sy0 = y0 >> sh, sx0 = x0 >> sh;
            const int npt = (np + 31) / 32, nks = p.cin[j] / 16;
            const int KC = p.KC[j], NT = p.NT[j], C8 = KC / 8, COUT_T = 32 * NT, nch = p.cin[j] / KC;
            const char *lw = smem + p.w_off[j];
            bf16_raw *lu = reinterpret_cast<bf16_raw *>(smem + p.u_off[j]);
            const bf16_raw *xin = p.src[j] + (size_t)b * Hs * Ws * p.src_cs[j];
            for (int item = wave; item < npt * ctn; item += 4) {  // (pixel tile, cout tile) -> one accumulator
                const int pt = item % npt, ct = item / npt;
                const int pix = pt * 32 + r, py = pix / ts, px = pix % ts;
                const bool inside = pix < np && sy0 + py < Hs && sx0 + px < Ws;
                const bf16_raw *xp = xin + ((size_t)(inside ? sy0 + py : 0) * Ws + (inside ? sx0 + px : 0)) * p.src_cs[j] + 8 * h;
                f32x16 acc;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 bs = *reinterpret_cast<const float4 *>(p.bias[j] + ct * 32 + 8 * g + 4 * h);
                    acc[4 * g + 0] = bs.x; acc[4 * g + 1] = bs.y; acc[4 * g + 2] = bs.z; acc[4 * g + 3] = bs.w;
                }
                const int cg = ct / NT, nt = ct % NT;
                // k-steps in ascending channel order, 8 loads in flight at a time
                for (int k0 = 0; k0 < nks; k0 += 8) {
                    u32x4 fb[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q)
                        if (k0 + q < nks) fb[q] = *reinterpret_cast<const u32x4 *>(xp + (k0 + q) * 16);
#pragma unroll
                    for (int q = 0; q < 8; ++q)
                        if (k0 + q < nks) {
                            const int ks = k0 + q, chunk = (ks * 16) / KC, kk = ((ks * 16) % KC) / 16;
                            const int unit = ((cg * nch + chunk) * C8 + kk * 2 + h) * COUT_T + nt * 32 + r;
                            const u32x4 fa = *reinterpret_cast<const u32x4 *>(lw + (size_t)unit * 16);
                            const u32x4 fbq = inside ? fb[q] : u32x4{0u, 0u, 0u, 0u};
                            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa), __builtin_bit_cast(bf16x8, fbq), acc, 0, 0, 0);
                        }
                }
                // lane (r, h) holds couts ct*32 + 8g + 4h + i of pixel `pix`: 8 bytes per g into the u tile [pixel][C]
                if (pix < np)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int c = ct * 32 + 8 * g + 4 * h;
                        if (c < p.C)
                            *reinterpret_cast<u32x2 *>(lu + (size_t)pix * p.C + c) =
                                u32x2{pack_bf16(acc[4 * g + 0], acc[4 * g + 1]), pack_bf16(acc[4 * g + 2], acc[4 * g + 3])};
                    }
            }
        }
        __syncthreads();
        // ================= phase 2: out = [relu](base + u_0 + u_1 + u_2), 16 bytes per thread and trip =================
        constexpr int UB = 4;  // base loads in flight per thread (one at a time is a memory round trip per 16 bytes)
        for (int i0 = tid; i0 < T * T * c8n; i0 += 256 * UB) {
            u32x4 bv[UB];
            size_t gp[UB];
            bool ok[UB];
#pragma unroll
            for (int q = 0; q < UB; ++q) {
                const int i = i0 + 256 * q, pix = i / c8n, ly = pix / T, lx = pix % T;
                ok[q] = i < T * T * c8n && y0 + ly < p.H && x0 + lx < p.W;
                gp[q] = ok[q] ? ((size_t)b * p.H + y0 + ly) * p.W + x0 + lx : 0;
                bv[q] = *reinterpret_cast<const u32x4 *>(p.base + gp[q] * p.base_cs + (ok[q] ? (i % c8n) * 8 : 0));
            }
#pragma unroll
            for (int q = 0; q < UB; ++q) {
                if (!ok[q]) continue;
                const int i = i0 + 256 * q, c8 = i % c8n, pix = i / c8n, ly = pix / T, lx = pix % T;
                float v[8] = {lo16(bv[q][0]), hi16(bv[q][0]), lo16(bv[q][1]), hi16(bv[q][1]), lo16(bv[q][2]), hi16(bv[q][2]), lo16(bv[q][3]), hi16(bv[q][3])};
                for (int j = 0; j < p.nsrc; ++j) {
                    const int sh = p.shift[j], ts = T >> sh;
                    const bf16_raw *lu = reinterpret_cast<const bf16_raw *>(smem + p.u_off[j]);
                    const u32x4 u = *reinterpret_cast<const u32x4 *>(lu + (size_t)((ly >> sh) * ts + (lx >> sh)) * p.C + c8 * 8);
                    v[0] += lo16(u[0]); v[1] += hi16(u[0]); v[2] += lo16(u[1]); v[3] += hi16(u[1]);
                    v[4] += lo16(u[2]); v[5] += hi16(u[2]); v[6] += lo16(u[3]); v[7] += hi16(u[3]);
                }
                if (p.relu)
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = fmaxf(v[k], 0.f);
                *reinterpret_cast<u32x4 *>(p.out + gp[q] * p.out_cs + c8 * 8) =
                    u32x4{pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7])};
            }
        }
        __syncthreads();  // the u tiles are rewritten by the next tile's phase 1
    }
}

// LDS bytes of a launch (weights of every source + their u tiles); <= 0: the shape does not fit this kernel
size_t fusion_up_lds_bytes(const FuseUpParams &p, int w_off[3], int u_off[3])
{
    size_t off = 0;
    for (int j = 0; j < p.nsrc; ++j) { w_off[j] = (int)off; off += (size_t)p.w_units[j] * 16; }
    for (int j = 0; j < p.nsrc; ++j) {
        const int ts = T >> p.shift[j];
        u_off[j] = (int)off;
        off += ((size_t)ts * ts * p.C * 2 + 15) & ~(size_t)15;
    }
    return off;
}
bool fusion_up_fits(int C, int nsrc, const int *cin_pad, const int *coutp, const int *shift)
{
    if (C % 8 || nsrc < 1 || nsrc > 3) return false;
    size_t bytes = 0;
    for (int j = 0; j < nsrc; ++j) {
        if (shift[j] < 1 || shift[j] > 3 || cin_pad[j] % 16) return false;
        bytes += (size_t)coutp[j] * cin_pad[j] * 2 + (((size_t)(T >> shift[j]) * (T >> shift[j]) * C * 2 + 15) & ~(size_t)15);
    }
    return bytes <= 150 * 1024;
}

hipError_t fusion_up_launch(FuseUpParams p, int num_cus, hipStream_t s)
{
    static bool inited[64] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (!inited[dev & 63]) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(fusion_up_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        if (e != hipSuccess) return e;
        inited[dev & 63] = true;
    }
    const size_t lds = fusion_up_lds_bytes(p, p.w_off, p.u_off);
    if (lds > 150 * 1024) return hipErrorInvalidValue;
    const int ntiles = p.B * ((p.H + T - 1) / T) * ((p.W + T - 1) / T);
    const int per_cu = lds * 2 <= 150 * 1024 ? 2 : 1;  // workgroups that fit a CU's LDS side by side (launch bounds: at most 2)
    const int grid = ntiles < num_cus * per_cu ? ntiles : num_cus * per_cu;
    hipLaunchKernelGGL(fusion_up_kernel, dim3(grid), dim3(256), lds, s, p);
    return hipGetLastError();
}
