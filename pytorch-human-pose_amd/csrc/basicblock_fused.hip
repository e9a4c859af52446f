// Fused BasicBlock for 32-channel branches (the highest-resolution HRNet branch and the
// HigherHRNet deconv head):   out = relu(bn2(conv2(relu(bn1(conv1(x))))) + x)
// -- /root/reference/src/keypoints/architectures/hrnet.py:108-124 -- in ONE kernel.
//
// Why: at C=32 a 3x3 conv has 144 FLOP per activation byte; run layer by layer the block moves
// x -> mid -> out through HBM (5 tensor passes, ~160 MB at 32x128x128x32) and is bandwidth/latency
// bound.  Fused, the 18x34 intermediate tile lives in LDS as bf16 and only x (once, + halo) and
// out cross HBM (2 passes).
//
// Shape of one workgroup (256 threads = 4 waves, one per SIMD; persistent over tiles):
//   output tile 16x32 px, mid tile 18x34 px (flattened into 20 MFMA column tiles of 32 pixels: 5 per
//   wave), input patch 20x36 px.  Both 3x3x32x32 weight sets stay in LDS for the life of the
//   workgroup.  MFMA roles as in conv_mfma.hip: A = weights (32 couts x 16 cin), B = pixels.
// Per tile (stamped with s_memtime in a -DHH_STAMP build, see tools/bb_bench.py):
//   * the next tile's 57.6 KB patch is prefetched into registers ONE 16-byte load per conv1 k-step
//     (a burst of 12 loads per thread back-pressures the CU's load path, ~10 B/cycle, and stalls the MFMAs)
//   * LDS fragment reads run one k-step ahead of the MFMAs (sched_group_barrier pins the order)
//   * the residual is two extra MFMAs per column tile with an identity A fragment (exact: x * 1.0 in fp32)
//     instead of 16 LDS reads + 16 converts + 16 adds per lane
//   * epilogues pair the two half-waves with v_permlane32_swap so every lane writes 16 contiguous bytes.
#include "kernels.h"

#include <utility>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {
typedef short i16x2 __attribute__((ext_vector_type(2)));
// ReLU after the rounding, on the packed pair: as signed 16-bit integers every negative bf16 (sign bit set, -0 included)
// is below zero and every non-negative one keeps its bits, so one v_pk_max_i16 replaces two canonicalise + two v_max_f32.
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

// 32 couts of one pixel: lanes (r,0) hold couts 8g..8g+3, lanes (r,1) couts 8g+4..8g+7 in acc[4g..4g+3].
// Returns for m = 0,1 the 16 bytes (bf16, ReLU applied) of couts 16m+8h .. 16m+8h+7 of this lane's pixel.
__device__ __forceinline__ void pack_rows16(const f32x16 &acc, u32x4 out[2])
{
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        unsigned x0 = pack_relu_bf16x2(acc[8 * m + 0], acc[8 * m + 1]), x1 = pack_relu_bf16x2(acc[8 * m + 2], acc[8 * m + 3]);
        unsigned y0 = pack_relu_bf16x2(acc[8 * m + 4], acc[8 * m + 5]), y1 = pack_relu_bf16x2(acc[8 * m + 6], acc[8 * m + 7]);
        // lanes 0-31: X = couts 16m..+3, Y = 16m+8..+11; lanes 32-63: X = 16m+4..+7, Y = 16m+12..+15.
        // swap X[32..63] <-> Y[0..31]: lanes 0-31 end with (X,Y) = couts 16m..16m+7, lanes 32-63 with 16m+8..16m+15
        auto s0 = __builtin_amdgcn_permlane32_swap(x0, y0, false, false);
        auto s1 = __builtin_amdgcn_permlane32_swap(x1, y1, false, false);
        out[m] = u32x4{s0[0], s1[0], s0[1], s1[1]};
    }
}

constexpr int TH = 16, TW = 32;          // output tile
constexpr int MH = TH + 2, MW = TW + 2;  // conv1 output (= conv2 input) tile
constexpr int IH = TH + 4, IW = TW + 4;  // input patch
constexpr int PS = 80;                   // bytes per staged pixel: 32 bf16 + 16 pad (odd number of 16-B slots)
constexpr int MPIX = MH * MW;            // 612 mid pixels -> 20 column tiles of 32 (28 idle lanes)
constexpr int NTHR = 512;                // 8 waves: two per SIMD, so one wave's VALU / LDS / store work hides behind the other's MFMAs
constexpr int P_UNITS = IH * IW * 4;                // 2880 16-byte units
constexpr int NPL = (P_UNITS + NTHR - 1) / NTHR;    // 6 prefetch loads per thread
constexpr int PATCH_BYTES = NPL * NTHR / 4 * PS;    // 61440: 57600 of patch + a pad that absorbs the idle units of the last load round
constexpr int MID_BYTES = 20 * 32 * PS;             // 51200
constexpr int W_BYTES = 9 * 4 * 32 * 16;            // 18432 per conv
constexpr int W_UNITS = W_BYTES / 16;               // 1152
}  // namespace

#ifdef HH_STAMP
#define STAMP(i) do { if (p.stamps && blockIdx.x == 0 && tid == 0) { const int tile_ix_ = (t - (int)blockIdx.x) / (int)gridDim.x; if (tile_ix_ < 8) p.stamps[tile_ix_ * 8 + (i)] = __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define STAMP(i)
#endif

size_t bb_fused_lds_bytes() { return PATCH_BYTES + MID_BYTES + 2 * W_BYTES + 256; }

__global__ __launch_bounds__(NTHR, 1) void bb_fused_kernel(const BBParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char *lds_p = smem;
    char *lds_m = smem + PATCH_BYTES;
    char *lds_w1 = lds_m + MID_BYTES;
    char *lds_w2 = lds_w1 + W_BYTES;
    float *lds_b = reinterpret_cast<float *>(lds_w2 + W_BYTES);  // [2][32] folded BN shifts (re-read at every accumulator init:
                                                                 // 32 live VGPRs would push the 256-register budget into scratch)
    const int tid = threadIdx.x;
#ifndef HH_NO_CLK
    if (p.clk && tid == 0) atomicMin(p.clk, wall_clock64());
#endif
#ifdef HH_STAMP  // in-kernel clock of workgroup 0: d(s_memtime) / d(s_memrealtime) x 100 MHz (MI355X_MICROARCH.md, DVFS give-back item 6)
    if (p.stamps && blockIdx.x == 0 && tid == 0) { p.stamps[64] = __builtin_amdgcn_s_memtime(); p.stamps[65] = __builtin_amdgcn_s_memrealtime(); }
#endif
    const int wave = tid >> 6, lane = tid & 63;
    const int r = lane & 31, h = lane >> 5;

    // identity A fragments (rows = couts, k = cin): frag kk has A[r][k] = 1 where 16*kk + k == r
    u32x4 ident[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int j = r - 16 * kk - 8 * h;  // element index inside this lane's 8-wide k slice
        const unsigned one = (j & 1) ? 0x3f800000u : 0x00003f80u;  // bf16 1.0 in the high / low half of a dword
        const bool on = j >= 0 && j < 8;
        ident[kk] = u32x4{on && (j >> 1) == 0 ? one : 0u, on && (j >> 1) == 1 ? one : 0u, on && (j >> 1) == 2 ? one : 0u,
                          on && (j >> 1) == 3 ? one : 0u};
    }

    // ---- tile-invariant per-thread geometry
    int pl_off[NPL], pl_yx[NPL];  // prefetch unit i: element offset from the tile's patch origin, (py << 8) | px
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        // unit u -> (pixel, 16-byte part): 16 consecutive lanes take the SAME part of 16 consecutive pixels, so that a
        // quarter-wave of ds_write_b128 (16 lanes x 16 B at the 80-byte pixel stride) touches every bank once; with the four
        // parts of a pixel on neighbouring lanes the pixels 0 and 3 of a quarter collided (2-way conflict on every patch write)
        const int u = tid + NTHR * i, pix = (u >> 6) * 16 + (u & 15), part = (u >> 4) & 3, py = pix / IW, px = pix % IW;
        pl_off[i] = (py * p.W + px) * p.in_cs + part * 8;
        pl_yx[i] = u < P_UNITS ? ((py << 8) | px) : (255 << 8);  // py = 255 -> never inside
    }
    // conv1 column tiles: waves 0-3 own 3 each (tiles 3w..3w+2), waves 4-7 own 2 each (12+2(w-4)..): waves w and w+4 share a
    // SIMD, so every SIMD carries 5 of the 20 tiles.  conv2: wave w owns output rows 2w, 2w+1.
    const int q0 = wave < 4 ? wave * 3 : 12 + (wave - 4) * 2;
    int paddr[3], maddr[3], myx[3];  // patch read base, mid write base, (my << 8) | mx   (entry 2 unused by waves 4-7)
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int pidx = (q0 + q) * 32 + r;
        const int pc = pidx < MPIX ? pidx : MPIX - 1;  // idle lanes read a valid pixel and own a real (unused) mid slot
        const int my = pc / MW, mx = pc % MW;
        paddr[q] = (my * IW + mx) * PS + h * 16;
        maddr[q] = pidx * PS + h * 16;
        myx[q] = (my << 8) | mx;
    }

    const int tiles_per_img = p.tiles_x * p.tiles_y;
    u32x4 preg[NPL];
    unsigned pf_mask = 0;            // bit i: prefetched unit i lies inside the image (else it is conv padding = 0)
    const bf16_raw *pf_base = p.in;  // patch origin of the tile being prefetched
    int pf_iy0 = 0, pf_ix0 = 0;
    bool pf_more = true;  // false: there is no next tile, the loads degenerate to re-reading p.in[0..7]
    // XCD-aware tile order: workgroup w (on XCD w % 8, the grid is a multiple of 8) walks tiles w, w + grid, ...; mapping tile
    // id i to (i % 8) * (ntiles / 8) + i / 8 gives every XCD a contiguous band of tiles (whole images at 128x128), so the halo
    // rows / straddled lines of neighbouring patches are fetched into one L2 once
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
        const int iy = pf_iy0 + (pl_yx[i] >> 8), ix = pf_ix0 + (pl_yx[i] & 255);
        const bool ok = pf_more & ((unsigned)iy < (unsigned)p.H) & ((unsigned)ix < (unsigned)p.W);  // '&': no short-circuit control flow
        // branch-free (a branch inside a k-step splits the scheduling region and un-interleaves the MFMAs);
        // the load result is not touched here (no s_waitcnt inside the MFMA loop); padding is applied when it goes to LDS
        preg[i] = *reinterpret_cast<const u32x4 *>(ok ? pf_base + pl_off[i] : p.in);
        pf_mask |= ok ? (1u << i) : 0u;
    };
    // one 16-byte unit of the next patch: registers -> LDS (zero outside the image = conv1 padding)
    auto write_patch_unit = [&](auto ic) {
        constexpr int i = decltype(ic)::value;
        const int u = tid + NTHR * i;  // units >= P_UNITS land in the pad behind the patch
        *reinterpret_cast<u32x4 *>(lds_p + ((u >> 6) * 16 + (u & 15)) * PS + ((u >> 4) & 3) * 16) = (pf_mask >> i) & 1u ? preg[i] : u32x4{0u, 0u, 0u, 0u};
    };
    // Workgroup barrier that waits for LDS traffic only: __syncthreads() also drains vmcnt, i.e. it would stall every
    // wave until the next tile's prefetch loads (issued during conv1, consumed during conv2) and the previous tile's
    // output stores have completed.
    auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

    int t = blockIdx.x;
    pf_setup(t);
    static_for<NPL>(pf_load);  // the first patch is in flight while the weights arrive (one exposed latency, not two)
    // ---- both weight sets: loaded once per workgroup
    for (int u = tid; u < W_UNITS; u += NTHR) {
        reinterpret_cast<u32x4 *>(lds_w1)[u] = reinterpret_cast<const u32x4 *>(p.w1)[u];
        reinterpret_cast<u32x4 *>(lds_w2)[u] = reinterpret_cast<const u32x4 *>(p.w2)[u];
    }
    if (tid < 32) { lds_b[tid] = p.b1[tid]; lds_b[32 + tid] = p.b2[tid]; }
    static_for<NPL>(write_patch_unit);
    __syncthreads();

    f32x16 acc2[2];              // conv2 accumulators; the finished tile is stored during the NEXT tile's conv1
    bool prev = false;
    bf16_raw *prev_out = p.out;  // + (oy0 * W + ox0) * out_cs of the finished tile
    int prev_oy0 = 0, prev_ox0 = 0;
    auto store_rows = [&](int q) {  // ReLU, bf16, 16 contiguous bytes per lane straight to HBM
        const int oy = prev_oy0 + wave * 2 + q, ox = prev_ox0 + r;
        u32x4 o[2];
        pack_rows16(acc2[q], o);
        bf16_raw *dst = prev_out + ((ptrdiff_t)(wave * 2 + q) * p.W + r) * p.out_cs + 8 * h;
        if (!(prev & (oy < p.H) & (ox < p.W))) dst = p.trash + 8 * h;  // select, not a branch: lanes outside the image (and the
        *reinterpret_cast<u32x4 *>(dst) = o[0];                         // first tile, which has no predecessor) hit a dummy line
        *reinterpret_cast<u32x4 *>(dst + 16) = o[1];
    };

    for (; t < p.ntiles; t += gridDim.x) {
        const int tb = band(t);
        const int b = tb / tiles_per_img, tt = tb % tiles_per_img;
        const int oy0 = (tt / p.tiles_x) * TH, ox0 = (tt % p.tiles_x) * TW;
        const int tn = t + gridDim.x;
        pf_more = tn < p.ntiles;
        pf_setup(pf_more ? tn : t);
        STAMP(0);

        // ================= conv1 + bn1 + relu -> mid tile (LDS, bf16) =================
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
            u32x4 fa[2], fb[2][NQ];
            auto ld1 = [&](int st, int buf) {
                const int tap = st >> 1, kk = st & 1, ky = tap / 3, kx = tap % 3;
                fa[buf] = *reinterpret_cast<const u32x4 *>(lds_w1 + ((tap * 4 + kk * 2 + h) * 32 + r) * 16);
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                    fb[buf][q] = *reinterpret_cast<const u32x4 *>(lds_p + paddr[q] + (ky * IW + kx) * PS + kk * 32);
            };
            ld1(0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, NQ + 1, 0);
            static_for<18>([&](auto ic) {
                constexpr int st = decltype(ic)::value;
                if (st + 1 < 18) {
                    ld1(st + 1, (st + 1) & 1);
                    __builtin_amdgcn_sched_group_barrier(0x100, NQ + 1, 0);
                }
                if constexpr (st < NPL) pf_load(ic);                     // one prefetch load per k-step
                if constexpr (st >= 12 && st < 14) store_rows(st - 12);  // the previous tile's rows leave while the MFMAs run
#pragma unroll
                for (int q = 0; q < NQ; ++q)
                    acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[st & 1]),
                                                                     __builtin_bit_cast(bf16x8, fb[st & 1][q]), acc[q], 0, 0, 0);
                if constexpr (st >= 12 && st < 14) {
                    // VALU work only overlaps this wave's own MFMAs when it sits BETWEEN them (in-order issue)
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
            STAMP(1);
            // conv2's accumulators start as bn2 shift + residual: the centre of the input patch times an identity A
            // fragment (exact: x * 1.0 in fp32), issued now because the patch buffer is recycled during conv2
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 bv = *reinterpret_cast<const float4 *>(lds_b + 32 + 8 * g + 4 * h);
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    acc2[q][4 * g + 0] = bv.x; acc2[q][4 * g + 1] = bv.y; acc2[q][4 * g + 2] = bv.z; acc2[q][4 * g + 3] = bv.w;
                }
            }
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const u32x4 x = *reinterpret_cast<const u32x4 *>(lds_p + ((wave * 2 + q + 2) * IW + r + 2) * PS + kk * 32 + h * 16);
                    acc2[q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ident[kk]), __builtin_bit_cast(bf16x8, x),
                                                                      acc2[q], 0, 0, 0);
                }
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int gy = oy0 - 1 + (myx[q] >> 8), gx = ox0 - 1 + (myx[q] & 255);
                // conv2 zero-pads the *feature map*: mid pixels outside the image are 0, not conv1(padding)
                const bool outside = ((unsigned)gy >= (unsigned)p.H) | ((unsigned)gx >= (unsigned)p.W);
                u32x4 o[2];
                pack_rows16(acc[q], o);
                *reinterpret_cast<u32x4 *>(lds_m + maddr[q]) = o[0];
                *reinterpret_cast<u32x4 *>(lds_m + maddr[q] + 32) = o[1];
                if (outside) {  // only tiles on the image border have such lanes
                    *reinterpret_cast<u32x4 *>(lds_m + maddr[q]) = u32x4{0u, 0u, 0u, 0u};
                    *reinterpret_cast<u32x4 *>(lds_m + maddr[q] + 32) = u32x4{0u, 0u, 0u, 0u};
                }
            }
        };
        if (wave < 4) conv1_phase(std::integral_constant<int, 3>{});
        else conv1_phase(std::integral_constant<int, 2>{});
        STAMP(2);
        lds_barrier();  // mid tile visible; every wave is done reading the patch
        STAMP(3);

        // ================= conv2 + bn2 (+ residual already in acc2); the next patch goes to LDS meanwhile =================
        // LDS read bandwidth (128 B/clk/CU, 8 clk per ds_read_b128) bounds this kernel, so every pixel fragment is read
        // once per (kx, k-half) and used for all the output rows it feeds: mid row i = out row j + ky.  Per (kx, kk):
        // 3 weight fragments (ky = 0..2) + 4 mid rows -> 6 MFMAs (was 9 reads per 6 MFMAs).
        {
            u32x4 fa[2][3], fb[2];
            auto lda = [&](int c, int buf) {  // c = kx * 2 + kk
                const int kx = c >> 1, kk = c & 1;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
                    fa[buf][ky] = *reinterpret_cast<const u32x4 *>(lds_w2 + (((ky * 3 + kx) * 4 + kk * 2 + h) * 32 + r) * 16);
            };
            auto ldb = [&](int s, int buf) {  // s = c * 4 + i
                const int c = s >> 2, i = s & 3, kx = c >> 1, kk = c & 1;
                fb[buf] = *reinterpret_cast<const u32x4 *>(lds_m + ((wave * 2 + i) * MW + r + kx) * PS + kk * 32 + h * 16);
            };
            lda(0, 0);
            ldb(0, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
            static_for<24>([&](auto sc) {
                constexpr int s = decltype(sc)::value, c = s >> 2, i = s & 3;
                constexpr int nread = (s + 1 < 24 ? 1 : 0) + ((i == 0 && c + 1 < 6) ? 3 : 0);
                if constexpr (s + 1 < 24) ldb(s + 1, (s + 1) & 1);
                if constexpr (i == 0 && c + 1 < 6) lda(c + 1, (c + 1) & 1);  // next combo's weights, a whole combo ahead
                if constexpr (nread > 0) __builtin_amdgcn_sched_group_barrier(0x100, nread, 0);
                if constexpr (s >= 2 && s - 2 < NPL) {
                    write_patch_unit(std::integral_constant<int, s - 2>{});
                    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                }
                constexpr int nm = (i == 0 || i == 3) ? 1 : 2;
                static_for<3>([&](auto kyc) {
                    constexpr int ky = decltype(kyc)::value, j = i - ky;
                    if constexpr (j >= 0 && j < 2)
                        acc2[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[c & 1][ky]),
                                                                          __builtin_bit_cast(bf16x8, fb[s & 1]), acc2[j], 0, 0, 0);
                });
                __builtin_amdgcn_sched_group_barrier(0x8, nm, 0);
            });
        }
        STAMP(4);
        lds_barrier();  // every wave is done with the mid tile; the next patch is visible
        STAMP(5);
        prev = true;
        prev_oy0 = oy0; prev_ox0 = ox0;
        prev_out = p.out + (((ptrdiff_t)b * p.H + oy0) * p.W + ox0) * p.out_cs;
        STAMP(6);
        STAMP(7);
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) store_rows(q);
#ifndef HH_NO_CLK
    if (p.clk && tid == 0) atomicMax(p.clk + 1, wall_clock64());
#endif
#ifdef HH_STAMP
    if (p.stamps && blockIdx.x == 0 && tid == 0) { p.stamps[66] = __builtin_amdgcn_s_memtime(); p.stamps[67] = __builtin_amdgcn_s_memrealtime(); }
#endif
}

static bf16_raw *g_trash_dev[64] = {};  // per device: 64 B every lane may scribble on (stores of lanes outside the image)

hipError_t bb_fused_init()
{
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (!g_trash_dev[dev & 63]) {
        e = hipMalloc((void **)&g_trash_dev[dev & 63], 256);
        if (e != hipSuccess) return e;
    }
    return hipFuncSetAttribute(reinterpret_cast<const void *>(bb_fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)bb_fused_lds_bytes());
}

hipError_t bb_fused_launch(BBParams p, int num_cus, hipStream_t s)
{
    p.tiles_x = (p.W + TW - 1) / TW;
    p.tiles_y = (p.H + TH - 1) / TH;
    p.ntiles = p.B * p.tiles_x * p.tiles_y;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || !g_trash_dev[dev & 63]) return hipErrorNotInitialized;
    p.trash = g_trash_dev[dev & 63];
    const int grid = p.ntiles < num_cus ? p.ntiles : num_cus;
    HH_LAUNCH(bb_fused_kernel, dim3(grid), dim3(NTHR), bb_fused_lds_bytes(), s, p);
    return hipGetLastError();
}
