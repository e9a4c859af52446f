// Fused 32-channel BasicBlock, single-wave form (round 4, experimental: HH_BB32=sw):   out = relu(bn2(conv2(relu(bn1(conv1(x))))) + x)
// -- /root/reference/src/keypoints/architectures/hrnet.py:108-124.  The tile, the LDS images, the fragment order and so the results are
// those of basicblock_fused_pc.hip (bit-identical), but ONE wavefront per SIMD runs BOTH convolutions of its row band: a 256-thread
// workgroup whose waves hold both weight sets (144 registers of the 512 a lone wave has) and whose pack / store / staging work is
// placed between its OWN MFMAs.  Why: in the producer / consumer form the two waves of a SIMD run their MFMA phases back to back at 43
// and 60 cycles per MFMA (the pipe's rate is 32): vector and matrix work of SIMD PARTNERS overlap far less than the issue rules suggest
// (profiles/design_notes_r01_r03.md, "Power, not pipes"), while vector work between a wave's own MFMAs does (MI355X_MICROARCH.md: up
// to five single-issue instructions hide per 32x32x16 MFMA).
#include "kernels.h"

#include <utility>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

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

// 32 couts of one pixel: lanes (r,0) hold couts 8g..8g+3, lanes (r,1) couts 8g+4..8g+7 in acc[4g..4g+3].
// Returns for m = 0,1 the 16 bytes (bf16, ReLU applied) of couts 16m+8h .. 16m+8h+7 of this lane's pixel.
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

// LDS fragment reads whose place in the instruction stream and whose wait are fixed by hand.  Left to the compiler, the
// reads of the software pipeline below end up right in front of their MFMAs (it renames the rotating fragment registers and
// waits lgkmcnt(0)), which exposes a full LDS round trip per step.  The read is an asm statement (volatile: the statements keep
// their order); its result may only be used through lds_wait<N>(), which waits until at most N younger LDS operations are
// outstanding (LDS operations complete in order; compiler-issued ones in between only make the wait conservative).
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

// MFMA with the weight fragment (A operand) held in ACCUMULATION registers: a lone wave has 256 + 256 registers, and only the AGPR half
// can hold what no vector instruction ever touches -- the two convolutions' 36 weight fragments (144 registers).  The compiler's own
// MFMA selection keeps A / B in the VGPR half (or, told otherwise, copies each fragment over before every use: 289 v_accvgpr_read per
// tile), so the instruction is written out.  acc += w x frag (mfma_w) or acc = c0 + w x frag (mfma_w0).
__device__ __forceinline__ void mfma_w(f32x16 &acc, const u32x4 &w, const u32x4 &frag)
{
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(w), "v"(frag));
}
__device__ __forceinline__ void mfma_w0(f32x16 &acc, const u32x4 &w, const u32x4 &frag, const f32x16 &c0)
{
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=v"(acc) : "a"(w), "v"(frag), "v"(c0));
}

#ifndef BBPC_STORE_AUX
#define BBPC_STORE_AUX 0  // cache policy bits of the output stores.  Experiment: 2 (nt) makes the block itself faster when its output is never
                          // read (128x128: 28.5 -> 25.4 us in tools/bb_compare.py) and the forward SLOWER (4.67 -> 4.72 ms): the next launch reads it
#endif
constexpr int TH = 14, TW = 32;          // output tile
constexpr int MH = TH + 2, MW = TW + 2;  // conv1 output (= conv2 input) tile: 16 x 34
constexpr int IH = TH + 4, IW = TW + 4;  // input patch: 18 x 36
constexpr int PRS = 38;                  // patch row stride in pixels (see the header: conflict-free edge tile)
constexpr int NTHR = 256;
constexpr int PATCH_BYTES = IH * PRS * 64;  // 43,776
constexpr int MID_BYTES = MH * MW * 64;     // 34,816
static_assert(IH * IW * 4 == 2592, "16-byte units of a patch");
constexpr int NPL = 6;                        // staging rounds of three patch rows (432 units): a thread moves units tid and tid + 256
// LDS fragment reads run RD steps (1-3 MFMAs each) ahead of the MFMAs that use them

constexpr int RDC = 4, NFBC = RDC + 1;  // consumer (more registers to spare)
constexpr int RP = MH / 4;                  // mid rows per producer wave
static_assert(MH % 4 == 0 && 2 * MH == 32, "4 producer bands; the two extra mid columns make exactly one 32-pixel column tile");
constexpr int OFF_MID = 2 * PATCH_BYTES, OFF_BIAS = OFF_MID + 2 * MID_BYTES;
constexpr int LDS_BYTES = OFF_BIAS + 256;
static_assert(LDS_BYTES <= 160 * 1024, "LDS");
}  // namespace

#ifdef HH_STAMP  // phase stamps of workgroup 0, iteration 2 (steady state): 8 slots per wave
#define PSTAMP(i) do { if (p.stamps && blockIdx.x == 0 && it == 2 && lane == 0) p.stamps[wave * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PSTAMP(i)
#endif

__global__ __launch_bounds__(NTHR, 1) void bbsw_kernel(const BBParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
#ifndef HH_NO_CLK
    if (p.clk && tid == 0) atomicMin(p.clk, wall_clock64());
    // workgroup 0 also leaves its core-cycle and wall-tick counts: their ratio is the clock the chip held during this launch
    const unsigned long long clk_c0 = p.clk && blockIdx.x == 0 ? __builtin_amdgcn_s_memtime() : 0ull;
    const unsigned long long clk_r0 = p.clk && blockIdx.x == 0 ? __builtin_amdgcn_s_memrealtime() : 0ull;
#endif
#ifdef HH_STAMP  // in-kernel clock of workgroup 0: d(s_memtime) / d(s_memrealtime) x 100 MHz (MI355X_MICROARCH.md, DVFS give-back item 6)
    if (p.stamps && blockIdx.x == 0 && tid == 0) { p.stamps[64] = __builtin_amdgcn_s_memtime(); p.stamps[65] = __builtin_amdgcn_s_memrealtime(); }
#endif
    const int lds0 = (int)(size_t)(__attribute__((address_space(3))) char *)smem;  // LDS byte address of smem[0], for the asm reads
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int r = lane & 31, h = lane >> 5;
    const int wj = wave;  // 4 waves: mid rows 4 wj .. 4 wj + 3 of conv1, output rows c0 .. of conv2

    const size_t in_bytes = (((size_t)p.B * p.H * p.W - 1) * p.in_cs + 32) * 2, out_bytes = (((size_t)p.B * p.H * p.W - 1) * p.out_cs + 32) * 2;
    const auto rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_raw *>(p.in), 0, (int)in_bytes, 0x00020000);
    const auto rs_out = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)out_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;  // a byte offset past every tensor here: the load returns 0, the store is dropped

    // ---- both convolutions' weight fragments (A operand: 32 couts x 16 cin per (tap, k half)), resident in registers
    u32x4 wreg1[18], wreg2[18];
    {
        const auto rs_w1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_raw *>(p.w1), 0, 18432, 0x00020000);
        const auto rs_w2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_raw *>(p.w2), 0, 18432, 0x00020000);
        static_for<18>([&](auto fc) {
            constexpr int f = decltype(fc)::value, tap = f >> 1, kk = f & 1;
            wreg1[f] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w1, ((tap * 4 + kk * 2 + h) * 32 + r) * 16, 0, 0));
            wreg2[f] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w2, ((tap * 4 + kk * 2 + h) * 32 + r) * 16, 0, 0));
        });
    }
    if (tid < 32) {
        reinterpret_cast<float *>(smem + OFF_BIAS)[tid] = p.b1[tid];
        reinterpret_cast<float *>(smem + OFF_BIAS)[32 + tid] = p.b2[tid];
    }

    // ---- tiles of this workgroup, XCD-aware order as in basicblock_fused.hip
    const int tiles_per_img = p.tiles_x * p.tiles_y;
    const int nloc = (p.ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    auto band = [&](int i) { return ((p.ntiles & 7) == 0 && (gridDim.x & 7) == 0) ? (i & 7) * (p.ntiles >> 3) + (i >> 3) : i; };
    // Rows.  Plain layout: a tile belongs to one image (b, first output row oy0).  TALL layout (round 4, p.VH = H + 2): the batch is
    // one image of B * (H + 2) rows -- two rows of zeros between consecutive images, what both 3x3 convolutions see as padding -- and
    // the tiles run through it without regard to the image borders, so only the very last tile row is partly empty (128 rows = 9.14
    // tiles of 14: a tenth of the plain layout's tiles were the 2-row remainders of the images).  A tile then touches at most two
    // images: b is the image of its first output row, oy0 that row's index inside it, and a row index y = oy0 + d that reaches VH
    // belongs to image b + 1, row y - VH (rowmap); rows H, H + 1 are the gap.  Plain layout: VH = 2^30, never reached.
    struct Geom { int b, oy0, ox0; };
    auto geom = [&](int k) {  // k-th tile of this workgroup
        const int tb = band((int)blockIdx.x + k * (int)gridDim.x);
        const int u = tb / tiles_per_img, tt = tb % tiles_per_img;
        const int oy = (tt / p.tiles_x) * TH, bq = oy / p.VH;
        return Geom{u + bq, oy - bq * p.VH, (tt % p.tiles_x) * TW};
    };
    // row y = oy0 + d of the tile of image b -> flat row (image * H + row) of the tensor, or -1 outside every image
    auto rowmap = [&](int b, int y) {
        const bool wrap = y >= p.VH;
        const int ya = wrap ? y - p.VH : y, bb = wrap ? b + 1 : b;
        return (((unsigned)ya < (unsigned)p.H) & (bb < p.B)) ? bb * p.H + ya : -1;
    };

    // ---- patch prefetch: global -> registers (issued early in an iteration) -> LDS (late in the same iteration).
    // Round i of 6 moves patch rows 3i..3i+2 (432 16-byte units; unit = (row 3i + u / 144, pixel (u % 144) >> 2, part u & 3)); a thread
    // moves units u = tid and (tid < 176) tid + 256, so its units of the six rounds differ only by a row step.
    u32x4 preg[NPL];  // six of the twelve staging items at a time: items 0..5 ride under P, items 6..11 under C
    static_assert(NPL * 3 == IH && IW * 4 * 3 <= 2 * NTHR, "six rounds of three patch rows, two units per thread");
    int pu_row[2], pu_cu[2], pu_px[2], pu_key[2], pu_lbase[2];
    bool pu_act[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int u = tid + k * NTHR;
        pu_row[k] = u / (IW * 4); pu_cu[k] = u - pu_row[k] * (IW * 4); pu_px[k] = pu_cu[k] >> 2;
        pu_act[k] = u < IW * 4 * 3;
        pu_key[k] = (pu_cu[k] ^ (pu_px[k] >> 2)) & 3;  // part ^ x key; the row key is XORed in per round
        pu_lbase[k] = (pu_row[k] * PRS + pu_px[k]) * 64;
    }
    const int pf_rowstep = 3 * p.W * p.in_cs * 2;
    const int pf_gapstep = (p.VH - p.H) * p.W * p.in_cs * 2;  // (tall layout) what a flat row index skips at an image border
    unsigned pf_vbase[2] = {0, 0};  // byte offset of the thread's round-0 units
    int pf_y[2] = {0, 0};           // image row of those units
    bool pf_xok[2] = {false, false}, pf_next = false;
    auto pf_setup = [&](int k) {
        const bool on = k < nloc;
        const Geom g = geom(on ? k : 0);
        pf_next = g.b + 1 < p.B;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int ix = g.ox0 - 2 + pu_px[q];
            pf_y[q] = g.oy0 - 2 + pu_row[q];
            pf_xok[q] = on & pu_act[q] & ((unsigned)ix < (unsigned)p.W);
            pf_vbase[q] = (unsigned)(((g.b * p.H + pf_y[q]) * p.W + ix) * p.in_cs * 2 + (pu_cu[q] & 3) * 16);
        }
    };
    auto pf_load1 = [&](auto ic, auto qc) {
        constexpr int i = decltype(ic)::value, q = decltype(qc)::value;
        const int yy = pf_y[q] + 3 * i;
        const bool wrap = yy >= p.VH;  // (tall layout) the row belongs to the next image
        const bool ok = pf_xok[q] & ((unsigned)(wrap ? yy - p.VH : yy) < (unsigned)p.H) & (!wrap | pf_next);
        const unsigned voff = ok ? pf_vbase[q] + (unsigned)(i * pf_rowstep) - (wrap ? (unsigned)pf_gapstep : 0u) : OOB;  // outside the image: zero = conv1's padding
        preg[(2 * i + q) % NPL] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, (int)voff, 0, 0));
    };
    auto pf_write1 = [&](auto ic, auto qc, int patch_off) {
        constexpr int i = decltype(ic)::value, q = decltype(qc)::value;
        if (pu_act[q])
            *reinterpret_cast<u32x4 *>(smem + patch_off + pu_lbase[q] + 3 * i * PRS * 64 + (((pu_key[q] ^ ((pu_row[q] + 3 * i) >> 1)) & 3) << 4)) = preg[(2 * i + q) % NPL];
    };
    // item n = 0 .. 11 of the staging work: (round n >> 1, unit n & 1)
    auto pf_load = [&](auto nc) { constexpr int n = decltype(nc)::value; pf_load1(std::integral_constant<int, (n >> 1)>{}, std::integral_constant<int, (n & 1)>{}); };
    auto pf_write = [&](auto nc, int patch_off) { constexpr int n = decltype(nc)::value; pf_write1(std::integral_constant<int, (n >> 1)>{}, std::integral_constant<int, (n & 1)>{}, patch_off); };
    constexpr int NPI = 2 * NPL;  // staging items per thread and tile
    auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

    pf_setup(0);
    static_for<NPL>(pf_load);
    static_for<NPL>([&](auto ic) { pf_write(ic, 0); });
    static_for<NPL>([&](auto ic) { pf_load(std::integral_constant<int, NPL + decltype(ic)::value>{}); });
    static_for<NPL>([&](auto ic) { pf_write(std::integral_constant<int, NPL + decltype(ic)::value>{}, 0); });
    __syncthreads();

    // ---- per-lane LDS read bases (buffer 0; the buffer offset is added per iteration)
    // patch, main column tiles: pixel (4 wj + i, r + kx), part kk*2 + h; the row key (2 wj + (i >> 1)) & 3 is XORed in per read
    int pa0[3][2], ma0[3][2];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int x = r + kx;
            pa0[kx][kk] = (4 * wj * PRS + x) * 64 + ((((kk * 2 + h) ^ (x >> 2)) & 3) << 4);
            ma0[kx][kk] = x * 64 + ((((kk * 2 + h) ^ (x >> 2)) & 3) << 4);  // + consumer row base below
        }
    const int c0 = wj < 2 ? 4 * wj : 8 + 3 * (wj - 2);  // consumer bands: rows 0-3, 4-7, 8-10, 11-13
    const int kb[3] = {((2 * wj) & 3) << 4, ((2 * wj + 1) & 3) << 4, ((2 * wj + 2) & 3) << 4};

    // One wave runs, per iteration `it`:  P(it) = conv1 of tile it (18 x 36 patch -> its 4 mid rows; wave 3 also the two extra mid
    // columns), then C(it-1) = conv2 of tile it-1 (mid tile -> its 3-4 output rows).  The vector work rides INSIDE the other phase's MFMA
    // loop: the previous tile's outputs (accumulators kept across the barrier) are packed and stored between the MFMAs of P, this tile's
    // mid rows and the next patch are packed / written to LDS between the MFMAs of C.  One LDS-only barrier per iteration: mid tile
    // and patch are double buffered exactly as in the producer / consumer form.
    auto bias_acc = [&](int off) {
        f32x16 b0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 bv = *reinterpret_cast<const float4 *>(smem + OFF_BIAS + (off + 8 * q + 4 * h) * 4);
            b0[4 * q + 0] = bv.x; b0[4 * q + 1] = bv.y; b0[4 * q + 2] = bv.z; b0[4 * q + 3] = bv.w;
        }
        return b0;
    };
    auto wave_loop = [&](auto edgec, auto rcc) {
        constexpr bool EDGE = decltype(edgec)::value;  // wave 3 also owns the column tile of the two extra mid columns
        constexpr int RC = decltype(rcc)::value;       // output rows of this wave (waves 0, 1: 4; waves 2, 3: 3)
        constexpr int NA1 = RP + (EDGE ? 1 : 0);
        f32x16 acc1[NA1], acc2[RC];
        // lane r of the edge tile = (mid row r >> 1, mid column 32 + (r & 1)); (x >> 2) & 3 == 0 for patch columns 32..35
        const int mrow = r >> 1, mcol = MW - 2 + (r & 1);

        // ---- slices of the vector work (one output / mid row each)
        auto finish_row = [&](const Geom &g, auto jc) {  // output row j of the tile whose conv2 ran in the previous iteration
            constexpr int j = decltype(jc)::value;
            const int ox = g.ox0 + r;
            const int fr = rowmap(g.b, g.oy0 + c0 + j);
            u32x4 o[2];
            pack_rows16(acc2[j], o);
            const bool ok = (fr >= 0) & (ox < p.W);
            const unsigned voff = ok ? (unsigned)((fr * p.W + ox) * p.out_cs * 2 + 16 * h) : OOB;
            __builtin_amdgcn_raw_buffer_store_b128(o[0], rs_out, (int)voff, 0, BBPC_STORE_AUX);
            __builtin_amdgcn_raw_buffer_store_b128(o[1], rs_out, (int)voff, 32, BBPC_STORE_AUX);
        };
        auto mid_row = [&](const Geom &g, int mcur, auto jc) {  // mid row 4 wj + j of this iteration's tile -> LDS (bf16, ReLU)
            constexpr int j = decltype(jc)::value;
            // Mid pixels outside the image are conv2's zero padding, not conv1(padding): whole rows (wave-uniform), column -1 (lane 0 of
            // the left-most tiles) and, in ragged widths only, columns >= W.
            const int m = 4 * wj + j;
            u32x4 o[2];
            pack_rows16(acc1[j], o);
            if (rowmap(g.b, g.oy0 - 1 + m) < 0) o[0] = o[1] = u32x4{0u, 0u, 0u, 0u};
            else if (g.ox0 + TW - 1 > p.W) {  // (wave-uniform: some main column ox0 - 1 + r is >= W)
                const bool outside = g.ox0 - 1 + r >= p.W;
                o[0] = outside ? u32x4{0u, 0u, 0u, 0u} : o[0];
                o[1] = outside ? u32x4{0u, 0u, 0u, 0u} : o[1];
            }
#pragma unroll
            for (int mm = 0; mm < 2; ++mm)
                *reinterpret_cast<u32x4 *>(smem + mcur + (m * MW + r) * 64 + ((((2 * mm + h) ^ (r >> 2)) & 3) << 4)) = o[mm];
            if (g.ox0 == 0 && r == 0) {  // column -1
#pragma unroll
                for (int mm = 0; mm < 2; ++mm)
                    *reinterpret_cast<u32x4 *>(smem + mcur + (m * MW) * 64 + ((2 * mm + h) << 4)) = u32x4{0u, 0u, 0u, 0u};
            }
        };

        for (int it = 0; it <= nloc; ++it) {
            pf_setup(it + 1);
            PSTAMP(0);
            const int pcur = (it & 1) * PATCH_BYTES, pnext = ((it + 1) & 1) * PATCH_BYTES;
            const Geom g = geom(it < nloc ? it : 0);       // tile of P
            const Geom gc = geom(it >= 1 ? it - 1 : 0);    // tile of C
            const Geom gf = geom(it >= 2 ? it - 2 : 0);    // tile whose outputs are still in acc2
            const int mcur = OFF_MID + (it & 1) * MID_BYTES;
            // residual of the tile about to be convolved by C, in flight under P (loaded as the output is stored: 16 bytes per lane)
            u32x4 res[RC][2];
            if (it >= 1) {
                const int ox = gc.ox0 + r;
#pragma unroll
                for (int j = 0; j < RC; ++j) {
                    const int fr = rowmap(gc.b, gc.oy0 + c0 + j);
                    const bool ok = (fr >= 0) & (ox < p.W);
                    const unsigned voff = ok ? (unsigned)((fr * p.W + ox) * p.in_cs * 2 + 16 * h) : OOB;
#pragma unroll
                    for (int m = 0; m < 2; ++m)
                        res[j][m] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, (int)voff, 32 * m, 0));
                }
            }
            // ---- P(it): conv1, with the stores of tile it-2 and the next patch's loads between its MFMAs
            if (it < nloc) {
                const f32x16 b0 = bias_acc(0);  // the C operand of every accumulator's first MFMA (no copies)
                int ea[3][2];
                if constexpr (EDGE) {
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int kk = 0; kk < 2; ++kk)
                            ea[ky][kk] = pcur + ((mrow + ky) * PRS + mcol) * 64 + ((((kk * 2 + h) ^ ((mrow + ky) >> 1)) & 3) << 4);
                }
                // wave 3 runs the edge tile FIRST (18 MFMAs into one accumulator, packed and written while the main rows' MFMAs run)
                constexpr int NR = RP + 2, NE = EDGE ? 18 : 0, NM = 6 * NR, NS = NE + NM;
                u32x4 fb[NFBC];
                auto ldb = [&](auto sc, int buf) {
                    constexpr int s = decltype(sc)::value;
                    if constexpr (s >= NE) {
                        constexpr int c = (s - NE) / NR, i = (s - NE) % NR, kx = c >> 1, kk = c & 1;
                        fb[buf] = lds_read_async<i * PRS * 64>(lds0 + pcur + (pa0[kx][kk] ^ kb[i >> 1]));
                    } else {
                        constexpr int tap = s >> 1, kk = s & 1, ky = tap / 3, kx = tap % 3;
                        fb[buf] = lds_read_async<kx * 64>(lds0 + ea[ky][kk]);
                    }
                };
                auto edge_out = [&]() {
                    if constexpr (!EDGE) return;
                    const int gxe = g.ox0 - 1 + mcol;
                    const bool outside = (rowmap(g.b, g.oy0 - 1 + mrow) < 0) | ((unsigned)gxe >= (unsigned)p.W);
                    u32x4 o[2];
                    pack_rows16(acc1[NA1 - 1], o);
#pragma unroll
                    for (int mm = 0; mm < 2; ++mm)
                        *reinterpret_cast<u32x4 *>(smem + mcur + (mrow * MW + mcol) * 64 + ((2 * mm + h) << 4)) = outside ? u32x4{0u, 0u, 0u, 0u} : o[mm];
                };
                static_for<RDC>([&](auto sc) { ldb(sc, decltype(sc)::value); });
                static_for<NS>([&](auto sc) {
                    constexpr int s = decltype(sc)::value;
                    if constexpr (s + RDC < NS) ldb(std::integral_constant<int, s + RDC>{}, (s + RDC) % NFBC);
                    if constexpr (s < NPL) pf_load(sc);  // first half of the next patch: in flight under this loop
                    if constexpr (s >= NS - NPL) {       // ... written at its end, the second half then fetched into the same registers
                        pf_write(std::integral_constant<int, s - (NS - NPL)>{}, pnext);
                        pf_load(std::integral_constant<int, NPL + s - (NS - NPL)>{});
                    }
                    // the previous-but-one tile's output rows: one per slice, spread over the MFMA steps
                    if constexpr (s >= NE + 4 && (s - NE - 4) % 8 == 0 && (s - NE - 4) / 8 < RC) {
                        if (it >= 2) finish_row(gf, std::integral_constant<int, (s - NE - 4) / 8>{});
                    }
                    lds_wait<(NS - 1 - s < RDC ? NS - 1 - s : RDC)>(fb[s % NFBC]);
                    if constexpr (EDGE && s == NE + 2) edge_out();
                    if constexpr (s >= NE) {
                        constexpr int c = (s - NE) / NR, i = (s - NE) % NR, kx = c >> 1, kk = c & 1;
                        constexpr int nm = (i == 0 || i == NR - 1) ? 1 : ((i == 1 || i == NR - 2) ? 2 : 3);
                        static_for<3>([&](auto kyc) {
                            constexpr int ky = decltype(kyc)::value, j = i - ky;
                            if constexpr (j >= 0 && j < RP) {
                                if constexpr (c == 0 && ky == 0) mfma_w0(acc1[j], wreg1[(ky * 3 + kx) * 2 + kk], fb[s % NFBC], b0);
                                else mfma_w(acc1[j], wreg1[(ky * 3 + kx) * 2 + kk], fb[s % NFBC]);
                            }
                        });
                        (void)nm;
                    } else {
                        if constexpr (s == 0) mfma_w0(acc1[NA1 - 1], wreg1[s], fb[s % NFBC], b0);
                        else mfma_w(acc1[NA1 - 1], wreg1[s], fb[s % NFBC]);
                    }
                });
            } else if (it >= 2) {
                static_for<RC>([&](auto jc) { finish_row(gf, jc); });
            }
            PSTAMP(1);
            // ---- C(it-1): conv2, with this tile's mid rows and the next patch going to LDS between its MFMAs
            if (it >= 1) {
                const int mread = OFF_MID + ((it - 1) & 1) * MID_BYTES + c0 * MW * 64;
                {
                    const f32x16 b0 = bias_acc(32);
#pragma unroll
                    for (int j = 0; j < RC; ++j)
#pragma unroll
                        for (int m = 0; m < 2; ++m) {
                            // lanes h = 0 hold channels 16m .. +7, lanes h = 1 channels 16m + 8 .. +15; swap (h = 0: dwords 2, 3) with
                            // (h = 1: dwords 0, 1): then dwords 0, 1 are q = 2m and dwords 2, 3 are q = 2m + 1 in both halves
                            auto s0 = __builtin_amdgcn_permlane32_swap(res[j][m][0], res[j][m][2], false, false);
                            auto s1 = __builtin_amdgcn_permlane32_swap(res[j][m][1], res[j][m][3], false, false);
                            const unsigned d[4] = {s0[0], s1[0], s0[1], s1[1]};
#pragma unroll
                            for (int t = 0; t < 4; ++t) {
                                acc2[j][8 * m + 2 * t + 0] = b0[8 * m + 2 * t + 0] + __uint_as_float(d[t] << 16);
                                acc2[j][8 * m + 2 * t + 1] = b0[8 * m + 2 * t + 1] + __uint_as_float(d[t] & 0xffff0000u);
                            }
                        }
                }
                PSTAMP(2);
                constexpr int NR = RC + 2, NS = 6 * NR;
                u32x4 fb[NFBC];
                auto ldb = [&](auto sc, int buf) {
                    constexpr int s = decltype(sc)::value, c = s / NR, i = s % NR, kx = c >> 1, kk = c & 1;
                    fb[buf] = lds_read_async<i * MW * 64>(lds0 + mread + ma0[kx][kk]);
                };
                static_for<RDC>([&](auto sc) { ldb(sc, decltype(sc)::value); });
                static_for<NS>([&](auto sc) {
                    constexpr int s = decltype(sc)::value, c = s / NR, i = s % NR, kx = c >> 1, kk = c & 1;
                    if constexpr (s + RDC < NS) ldb(std::integral_constant<int, s + RDC>{}, (s + RDC) % NFBC);
                    // this tile's mid rows (accumulators of P) -> LDS, one row per slice; then the next patch
                    if constexpr (s >= 2 && (s - 2) % 4 == 0 && (s - 2) / 4 < RP) {
                        if (it < nloc) mid_row(g, mcur, std::integral_constant<int, (s - 2) / 4>{});
                    }
                    if constexpr (s >= NS - NPL - 1 && s - (NS - NPL - 1) < NPL) {
                        if (it < nloc) pf_write(std::integral_constant<int, NPL + s - (NS - NPL - 1)>{}, pnext);
                    }
                    lds_wait<(NS - 1 - s < RDC ? NS - 1 - s : RDC)>(fb[s % NFBC]);
                    constexpr int nm = (i == 0 || i == NR - 1) ? 1 : ((i == 1 || i == NR - 2) ? 2 : 3);
                    static_for<3>([&](auto kyc) {
                        constexpr int ky = decltype(kyc)::value, j = i - ky;
                        if constexpr (j >= 0 && j < RC) mfma_w(acc2[j], wreg2[(ky * 3 + kx) * 2 + kk], fb[s % NFBC]);
                    });
                    (void)nm;
                });
            } else {
                static_for<RP>([&](auto jc) { mid_row(g, mcur, jc); });
                static_for<NPL>([&](auto ic) { pf_write(std::integral_constant<int, NPL + decltype(ic)::value>{}, pnext); });
            }
            PSTAMP(3);
            lds_barrier();
            PSTAMP(4);
        }
        static_for<RC>([&](auto jc) { finish_row(geom(nloc - 1), jc); });
    };
    if (wj < 2) wave_loop(std::false_type{}, std::integral_constant<int, 4>{});
    else if (wj == 2) wave_loop(std::false_type{}, std::integral_constant<int, 3>{});
    else wave_loop(std::true_type{}, std::integral_constant<int, 3>{});
#ifndef HH_NO_CLK
    if (p.clk && tid == 0) atomicMax(p.clk + 1, wall_clock64());
    if (p.clk && blockIdx.x == 0 && tid == 0) {
        p.clk[2] = __builtin_amdgcn_s_memtime() - clk_c0;
        p.clk[3] = __builtin_amdgcn_s_memrealtime() - clk_r0;
    }
#endif
#ifdef HH_STAMP
    if (p.stamps && blockIdx.x == 0 && tid == 0) { p.stamps[66] = __builtin_amdgcn_s_memtime(); p.stamps[67] = __builtin_amdgcn_s_memrealtime(); }
#endif
}

hipError_t bbsw_init()
{
    return hipFuncSetAttribute(reinterpret_cast<const void *>(bbsw_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
}

// 32-bit buffer offsets: both tensors must stay below 2 GB (the engine falls back to the tile form otherwise)
bool bbsw_supported(const BBParams &p)
{
    return (size_t)p.B * p.H * p.W * (size_t)(p.in_cs > p.out_cs ? p.in_cs : p.out_cs) * 2 < 0x7fffffffull;
}

hipError_t bbsw_launch(BBParams p, int num_cus, hipStream_t s)
{
    p.tiles_x = (p.W + TW - 1) / TW;
    p.tiles_y = (p.H + TH - 1) / TH;
    p.ntiles = p.B * p.tiles_x * p.tiles_y;
    p.VH = 1 << 30;
    if (p.tall != 0) {
        // the batch as one tall image (see the kernel): fewer tiles whenever H is not a multiple of the tile height.  The tile rows are
        // rounded up until the tile count is a multiple of 8 (the XCD-contiguous tile order needs that; the extra tiles lie behind
        // the last image and move nothing)
        int ty = (p.B * (p.H + 2) - 2 + TH - 1) / TH;
        while ((ty * p.tiles_x) & 7) ++ty;
        if (ty * p.tiles_x < p.ntiles || p.tall > 1) { p.tiles_y = ty; p.ntiles = ty * p.tiles_x; p.VH = p.H + 2; }
    }
    if (!bbsw_supported(p)) return hipErrorInvalidValue;
    const int grid = p.ntiles < num_cus ? p.ntiles : num_cus;
    HH_LAUNCH(bbsw_kernel, dim3(grid), dim3(NTHR), LDS_BYTES, s, p);
    return hipGetLastError();
}
