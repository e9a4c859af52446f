// Fused 32-channel BasicBlock, producer / consumer form:   out = relu(bn2(conv2(relu(bn1(conv1(x))))) + x)
// -- /root/reference/src/keypoints/architectures/hrnet.py:108-124 -- in ONE kernel, like basicblock_fused.hip (the "tile
// form", kept behind HH_BB32=tile), but organised around the LDS and the vector-instruction budget instead of around the tile:
//
//   * waves 0-3 only ever run conv1 and waves 4-7 only conv2, so each wave keeps ITS conv's 18 weight fragments in 72 VGPRs
//     for the life of the (persistent) workgroup: no weight reads (the tile form re-reads them for every tile);
//   * a wave owns a band of rows, and a pixel fragment (one row, one kx shift, one k half) feeds the three output rows it is a
//     tap of (ky = 0..2): 6 reads per 12 MFMAs in a 4-row band.  Together 0.45 ds_read_b128 per MFMA instead of 1.3;
//   * the two groups work one tile apart (conv1 of tile t+1 beside conv2 of tile t) with ONE workgroup barrier per tile, and
//     the two waves of a SIMD (w and w+4) run in antiphase inside it: the producer issues its MFMAs first and packs / writes
//     the mid rows second; the consumer first packs / stores the tile of the iteration before (its accumulators survive the
//     barrier) and then issues its MFMAs;
//   * LDS fragment reads are asm statements with hand-counted waits (lds_read_async / lds_wait): left to the compiler the
//     software pipeline collapses (it renames the rotating fragment registers and waits right behind the reads).
//
// Tile: 14x32 outputs, 16x34 mid pixels, 18x36 input patch; patch and mid tile are double buffered (157 KB of LDS), pixels
// are 64 bytes (no padding): the 16-byte part index is XOR-swizzled by (x >> 2) & 3 (and by (row >> 1) & 3 in the
// patch, whose rows are 38 pixels apart), which makes every ds_read_b128 below conflict-free, the column tile of the
// two extra mid columns (lanes = 16 rows x 2 columns) included.
// The residual enters as the accumulators' initial value (shift + x, fp32) from global memory: the lines are L2-warm, the
// patch was fetched one tile earlier.
//
// Measured (tools/bb_compare.py, 1 s of back-to-back launches, B = 32): 128x128 26.9 us vs 27.4 us for the tile form, 256x256
// 90 vs 94 us; in the forward 5.03 vs 5.12 ms (4 lanes), 5.75 vs 5.95 ms (one lane).  Halving the LDS reads bought this little
// because neither form is bound by the LDS or by issue slots: with every load and store compiled out (-DBBPC_NOLOAD -DBBPC_NOSTORE
// -DBBPC_NORES) the same instruction stream runs the 256x256 case in 65 us at an in-kernel clock of 2.40 GHz; with its 268 MB
// of HBM traffic it takes 8 % more cycles and the chip holds only 1.9-2.0 GHz (d s_memtime / d s_memrealtime, -DHH_STAMP
// build) -- the block is bound by what the chip can power: HBM streaming at ~3 TB/s beside ~0.85 PFLOP/s of MFMA.
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

#ifndef BBPC_STORE_AUX
#define BBPC_STORE_AUX 0  // cache policy bits of the output stores.  Experiment: 2 (nt) makes the block itself faster when its output is never
                          // read (128x128: 28.5 -> 25.4 us in tools/bb_compare.py) and the forward SLOWER (4.67 -> 4.72 ms): the next launch reads it
#endif
constexpr int TH = 14, TW = 32;          // output tile
constexpr int MH = TH + 2, MW = TW + 2;  // conv1 output (= conv2 input) tile: 16 x 34
constexpr int IH = TH + 4, IW = TW + 4;  // input patch: 18 x 36
constexpr int PRS = 38;                  // patch row stride in pixels (see the header: conflict-free edge tile)
constexpr int NTHR = 512;
constexpr int PATCH_BYTES = IH * PRS * 64;  // 43,776
constexpr int MID_BYTES = MH * MW * 64;     // 34,816
constexpr int P_UNITS = IH * IW * 4;        // 2592 16-byte units
constexpr int NPL = (P_UNITS + NTHR - 1) / NTHR;  // 6 (the last round: 32 threads)
// LDS fragment reads run RD steps (1-3 MFMAs each) ahead of the MFMAs that use them
#ifndef BBPC_RD
#define BBPC_RD 2
#endif
#ifndef BBPC_RDC
#define BBPC_RDC 4
#endif
constexpr int RD = BBPC_RD, NFB = RD + 1;     // producer
constexpr int RDC = BBPC_RDC, NFBC = RDC + 1;  // consumer (more registers to spare)
constexpr int RP = MH / 4;                  // mid rows per producer wave
static_assert(MH % 4 == 0 && 2 * MH == 32, "4 producer bands; the two extra mid columns make exactly one 32-pixel column tile");
constexpr int OFF_MID = 2 * PATCH_BYTES, OFF_BIAS = OFF_MID + 2 * MID_BYTES;
constexpr int OFF_FINW = OFF_BIAS + 256, OFF_FINB = OFF_FINW + 2048;  // (FIN) the 1x1 head's B fragments [2][2][32][8] bf16 and its bias [32]
constexpr int LDS_BYTES = OFF_BIAS + 256, LDS_BYTES_FIN = OFF_FINB + 128;
static_assert(LDS_BYTES_FIN <= 160 * 1024, "LDS");
}  // namespace

#ifdef HH_STAMP  // phase stamps of workgroup 0, iteration 2 (steady state): 8 slots per wave
#define PSTAMP(i) do { if (p.stamps && blockIdx.x == 0 && it == 2 && lane == 0) p.stamps[wave * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PSTAMP(i)
#endif

// FIN: the block is the last one of DeconvHeatmapsHead and the head's final 1x1 convolution (32 -> K channels with bias, fp32 NCHW
// result; higher_hrnet.py:38-44) runs in the consumer's epilogue: the packed bf16 rows of the block output are, as they stand, the A
// operand (32 pixels x 16 channels per k half) of two more MFMAs against the head's weights (B: 16 channels x 32 couts, K real), so
// the block output is never stored (-134 MB written, -134 MB read at B = 32 @ 512^2) and the head's launch disappears.  The result
// tile is D[pixel 8q + 4h + t][cout r] in d[4q + t]: a lane stores four float4 per row, 16 consecutive bytes of one channel plane.
template <bool FIN>
__device__ __forceinline__ void bbpc_body(const BBParams &p)
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
    const bool producer = wave < 4;
    const int wj = wave & 3;

    const size_t in_bytes = (((size_t)p.B * p.H * p.W - 1) * p.in_cs + 32) * 2, out_bytes = (((size_t)p.B * p.H * p.W - 1) * p.out_cs + 32) * 2;
    const auto rs_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_raw *>(p.in), 0, (int)in_bytes, 0x00020000);
    const auto rs_out = FIN ? __builtin_amdgcn_make_buffer_rsrc(p.fin_out, 0, (int)((size_t)p.B * p.fin_K * p.H * p.W * 4), 0x00020000)
                            : __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)out_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;  // a byte offset past every tensor here: the load returns 0, the store is dropped

    // ---- this wave's weight fragments (A operand: 32 couts x 16 cin per (tap, k half)), resident in registers
    u32x4 wreg[18];
    {
        const auto rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_raw *>(producer ? p.w1 : p.w2), 0, 18432, 0x00020000);
        static_for<18>([&](auto fc) {
            constexpr int f = decltype(fc)::value, tap = f >> 1, kk = f & 1;
            wreg[f] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_w, ((tap * 4 + kk * 2 + h) * 32 + r) * 16, 0, 0));
        });
    }
    if (tid < 32) {
        reinterpret_cast<float *>(smem + OFF_BIAS)[tid] = p.b1[tid];
        reinterpret_cast<float *>(smem + OFF_BIAS)[32 + tid] = p.b2[tid];
    }
    if constexpr (FIN) {
        if (tid >= 64 && tid < 192) reinterpret_cast<u32x4 *>(smem + OFF_FINW)[tid - 64] = reinterpret_cast<const u32x4 *>(p.fin_w)[tid - 64];
        if (tid >= 192 && tid < 224) reinterpret_cast<float *>(smem + OFF_FINB)[tid - 192] = p.fin_b[tid - 192];
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
    // Round i of 6 moves patch rows 3i..3i+2 (432 16-byte units: threads 0..431; unit = (row 3i + tid / 144, pixel (tid % 144) >> 2,
    // part tid & 3)), so a thread's six units differ only by a row step: one address each side per thread, a scalar step per
    // round -- the vector-instruction budget of this kernel is as tight as its MFMA budget.
    u32x4 preg[NPL];
    static_assert(NPL * 3 == IH && IW * 4 * 3 <= NTHR, "six rounds of three patch rows");
    const int pu_row = tid / (IW * 4), pu_cu = tid - pu_row * (IW * 4), pu_px = pu_cu >> 2;
    const bool pu_act = tid < IW * 4 * 3;
    const int pu_key = (pu_cu ^ (pu_px >> 2)) & 3;                 // part ^ x key; the row key is XORed in per round
    const int pu_lbase = (pu_row * PRS + pu_px) * 64;
    const int pf_rowstep = 3 * p.W * p.in_cs * 2;
    const int pf_gapstep = (p.VH - p.H) * p.W * p.in_cs * 2;  // (tall layout) what a flat row index skips at an image border
    unsigned pf_vbase = 0;  // byte offset of this thread's round-0 unit
    int pf_y = 0;           // image row of that unit
    bool pf_xok = false, pf_next = false;
    auto pf_setup = [&](int k) {
        const bool on = k < nloc;
        const Geom g = geom(on ? k : 0);
        const int ix = g.ox0 - 2 + pu_px;
        pf_y = g.oy0 - 2 + pu_row;
        pf_xok = on & pu_act & ((unsigned)ix < (unsigned)p.W);
        pf_next = g.b + 1 < p.B;
        pf_vbase = (unsigned)(((g.b * p.H + pf_y) * p.W + ix) * p.in_cs * 2 + (pu_cu & 3) * 16);
    };
    auto pf_load = [&](auto ic) {
        constexpr int i = decltype(ic)::value;
        const int yy = pf_y + 3 * i;
        const bool wrap = yy >= p.VH;  // (tall layout) the row belongs to the next image
        const bool ok = pf_xok & ((unsigned)(wrap ? yy - p.VH : yy) < (unsigned)p.H) & (!wrap | pf_next);
        const unsigned voff = ok ? pf_vbase + (unsigned)(i * pf_rowstep) - (wrap ? (unsigned)pf_gapstep : 0u) : OOB;  // outside the image: zero = conv1's padding
#ifdef BBPC_NOLOAD  // timing experiment: no patch traffic (results are wrong)
        preg[i] = u32x4{voff, 0u, 0u, 0u};
#else
        preg[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, (int)voff, 0, 0));
#endif
    };
    auto pf_write = [&](auto ic, int patch_off) {
        constexpr int i = decltype(ic)::value;
        if (pu_act)
            *reinterpret_cast<u32x4 *>(smem + patch_off + pu_lbase + 3 * i * PRS * 64 + (((pu_key ^ ((pu_row + 3 * i) >> 1)) & 3) << 4)) = preg[i];
    };
    auto lds_barrier = [&]() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };

    pf_setup(0);
    static_for<NPL>(pf_load);
    static_for<NPL>([&](auto ic) { pf_write(ic, 0); });
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

    // The two roles run separate copies of the tile loop (one barrier per iteration each, nloc + 1 iterations both), so
    // that neither carries the other's addresses and accumulators in its register budget.
    //
    // Inside an iteration the two waves of a SIMD are in ANTIPHASE: the producer runs its MFMAs first and its epilogue
    // (pack, ReLU, LDS writes) second; the consumer first finishes the tile of the iteration before (residual, pack, stores:
    // its accumulators survive the barrier) and only then starts its MFMAs.  Each one's vector / LDS / store work then sits
    // under the other's MFMAs instead of both groups converting and storing side by side at the end of the iteration.
    auto bias_acc = [&](int off) {
        f32x16 b0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 bv = *reinterpret_cast<const float4 *>(smem + OFF_BIAS + (off + 8 * q + 4 * h) * 4);
            b0[4 * q + 0] = bv.x; b0[4 * q + 1] = bv.y; b0[4 * q + 2] = bv.z; b0[4 * q + 3] = bv.w;
        }
        return b0;
    };
    if (producer) {
        auto produce = [&](auto edgec, int it) {
            constexpr bool EDGE = decltype(edgec)::value;  // wave 3 also owns the column tile of the two extra mid columns
            const int pcur = (it & 1) * PATCH_BYTES, pnext = ((it + 1) & 1) * PATCH_BYTES;
            const Geom g = geom(it);
            const int mcur = OFF_MID + (it & 1) * MID_BYTES;
            f32x16 acc[RP + (EDGE ? 1 : 0)];
            const f32x16 b0 = bias_acc(0);  // the C operand of every accumulator's first MFMA (no copies)
            // lane r of the edge tile = (mid row r >> 1, mid column 32 + (r & 1)); (x >> 2) & 3 == 0 for patch columns 32..35
            const int mrow = r >> 1, mcol = MW - 2 + (r & 1);
            int ea[3][2];
            if constexpr (EDGE) {
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kk = 0; kk < 2; ++kk)
                        ea[ky][kk] = pcur + ((mrow + ky) * PRS + mcol) * 64 + ((((kk * 2 + h) ^ ((mrow + ky) >> 1)) & 3) << 4);
            }
            // wave 3 runs the edge tile FIRST (18 MFMAs into one accumulator, packed and written while the main rows' MFMAs run):
            // five live accumulators next to the weights and the prefetch registers do not fit in 256 VGPRs
            constexpr int NR = RP + 2, NE = EDGE ? 18 : 0, NM = 6 * NR, NS = NE + NM;
            u32x4 fb[NFB];
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
                pack_rows16(acc[RP + (EDGE ? 0 : -1)], o);
#pragma unroll
                for (int mm = 0; mm < 2; ++mm)
                    *reinterpret_cast<u32x4 *>(smem + mcur + (mrow * MW + mcol) * 64 + ((2 * mm + h) << 4)) = outside ? u32x4{0u, 0u, 0u, 0u} : o[mm];
            };
            static_for<RD>([&](auto sc) { ldb(sc, decltype(sc)::value); });
            static_for<NS>([&](auto sc) {
                constexpr int s = decltype(sc)::value;
                if constexpr (s + RD < NS) {
                    ldb(std::integral_constant<int, s + RD>{}, (s + RD) % NFB);
                }
                if constexpr (s < NPL) pf_load(sc);
                lds_wait<(NS - 1 - s < RD ? NS - 1 - s : RD)>(fb[s % NFB]);
                if constexpr (EDGE && s == NE + 2) edge_out();
                if constexpr (s >= NE) {
                    constexpr int c = (s - NE) / NR, i = (s - NE) % NR, kx = c >> 1, kk = c & 1;
                    constexpr int nm = (i == 0 || i == NR - 1) ? 1 : ((i == 1 || i == NR - 2) ? 2 : 3);
                    static_for<3>([&](auto kyc) {
                        constexpr int ky = decltype(kyc)::value, j = i - ky;
                        if constexpr (j >= 0 && j < RP)
                            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wreg[(ky * 3 + kx) * 2 + kk]),
                                                                             __builtin_bit_cast(bf16x8, fb[s % NFB]),
                                                                             (c == 0 && ky == 0) ? b0 : acc[j], 0, 0, 0);
                    });
                    __builtin_amdgcn_sched_group_barrier(0x8, nm, 0);
                } else {
                    constexpr int ie = RP + (EDGE ? 0 : -1);  // (only instantiated with EDGE)
                    acc[ie] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wreg[s]), __builtin_bit_cast(bf16x8, fb[s % NFB]),
                                                                      s == 0 ? b0 : acc[ie], 0, 0, 0);
                    __builtin_amdgcn_sched_group_barrier(0x8, 1, 0);
                }
            });
            PSTAMP(1);
            // mid rows -> LDS (bf16, ReLU).  Mid pixels outside the image are conv2's zero padding, not conv1(padding): whole rows
            // (wave-uniform), column -1 (lane 0 of the left-most tiles) and, in ragged widths only, columns >= W.
            const bool ragged = g.ox0 + TW - 1 > p.W;  // wave-uniform: some main column ox0 - 1 + r is >= W
#pragma unroll
            for (int j = 0; j < RP; ++j) {
                const int m = 4 * wj + j;
                u32x4 o[2];
                pack_rows16(acc[j], o);
                if (rowmap(g.b, g.oy0 - 1 + m) < 0) o[0] = o[1] = u32x4{0u, 0u, 0u, 0u};
                else if (ragged) {
                    const bool outside = g.ox0 - 1 + r >= p.W;
                    o[0] = outside ? u32x4{0u, 0u, 0u, 0u} : o[0];
                    o[1] = outside ? u32x4{0u, 0u, 0u, 0u} : o[1];
                }
#pragma unroll
                for (int mm = 0; mm < 2; ++mm)
                    *reinterpret_cast<u32x4 *>(smem + mcur + (m * MW + r) * 64 + ((((2 * mm + h) ^ (r >> 2)) & 3) << 4)) = o[mm];
            }
            if (g.ox0 == 0 && r == 0) {  // column -1
#pragma unroll
                for (int j = 0; j < RP; ++j)
#pragma unroll
                    for (int mm = 0; mm < 2; ++mm)
                        *reinterpret_cast<u32x4 *>(smem + mcur + ((4 * wj + j) * MW) * 64 + ((2 * mm + h) << 4)) = u32x4{0u, 0u, 0u, 0u};
            }
            static_for<NPL>([&](auto ic) { pf_write(ic, pnext); });  // the next patch (loads issued at the top of the MFMA loop)
        };
        for (int it = 0; it <= nloc; ++it) {
            pf_setup(it + 1);
            PSTAMP(0);
            if (it < nloc) {
                if (wj == 3) produce(std::true_type{}, it);
                else produce(std::false_type{}, it);
            }  // (last iteration: nothing to produce and nothing to prefetch)
            PSTAMP(3);
            lds_barrier();
            PSTAMP(4);
        }
    } else {
        // One instantiation of the whole consumer loop per band height (waves 4, 5: 4 rows; waves 6, 7: 3 rows).
        auto consumer_loop = [&](auto rcc) {
            constexpr int RC = decltype(rcc)::value;
            f32x16 acc[RC];
            // finish the tile whose MFMAs ran in the previous iteration: ReLU, bf16, 16-byte stores (the residual went into the
            // accumulators' initial value)
            auto finish = [&](int k) {
                const Geom g = geom(k);
                const int ox = g.ox0 + r;
#pragma unroll
                for (int j = 0; j < RC; ++j) {
                    const int fr = rowmap(g.b, g.oy0 + c0 + j);
                    u32x4 o[2];
                    pack_rows16(acc[j], o);
                    const bool ok = (fr >= 0) & (ox < p.W);
#ifdef BBPC_NOSTORE
                    const unsigned voff = (ok && o[0][0] == 0x12345678u) ? (unsigned)((fr * p.W + ox) * p.out_cs * 2 + 16 * h) : OOB;
#else
                    const unsigned voff = ok ? (unsigned)((fr * p.W + ox) * p.out_cs * 2 + 16 * h) : OOB;
#endif
                    __builtin_amdgcn_raw_buffer_store_b128(o[0], rs_out, (int)voff, 0, BBPC_STORE_AUX);
                    __builtin_amdgcn_raw_buffer_store_b128(o[1], rs_out, (int)voff, 32, BBPC_STORE_AUX);
                }
            };
            for (int it = 0; it <= nloc; ++it) {
                pf_setup(it + 1);
                PSTAMP(0);
                // residual of the tile about to be convolved, in flight (like the next patch) under the stores of finish(); then the
                // accumulators start as shift + residual.  Loaded as the output is stored -- 16 bytes per lane, channels 16m + 8h .. +7 of
                // pixel (oy, ox0 + r): a wave instruction touches 32 half lines; 8-byte pieces in accumulator order touch 64 lines each and
                // kept the CU's address path busy for ~1000 cycles per wave and tile -- and brought into accumulator order (couts
                // 8q + 4h .. +3) by the inverse of pack_rows16's lane swap.
                u32x4 res[RC][2];
                const Geom g = geom(it >= 1 ? it - 1 : 0);
                if (it >= 1) {
                    const int ox = g.ox0 + r;
#pragma unroll
                    for (int j = 0; j < RC; ++j) {
                        const int fr = rowmap(g.b, g.oy0 + c0 + j);
                        const bool ok = (fr >= 0) & (ox < p.W);
                        const unsigned voff = ok ? (unsigned)((fr * p.W + ox) * p.in_cs * 2 + 16 * h) : OOB;
#pragma unroll
                        for (int m = 0; m < 2; ++m)
#ifdef BBPC_NORES
                            res[j][m] = u32x4{voff, 0u, 0u, 0u};
#else
                            res[j][m] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_in, (int)voff, 32 * m, 0));
#endif
                    }
                }
                static_for<NPL>(pf_load);  // the next patch: written to LDS late in the MFMA loop
                if constexpr (!FIN) { if (it >= 2) finish(it - 2); }
                PSTAMP(1);
                const int pnext = ((it + 1) & 1) * PATCH_BYTES;
                if (it >= 1) {
                    const int mcur = OFF_MID + ((it - 1) & 1) * MID_BYTES + c0 * MW * 64;
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
                                    acc[j][8 * m + 2 * t + 0] = b0[8 * m + 2 * t + 0] + __uint_as_float(d[t] << 16);
                                    acc[j][8 * m + 2 * t + 1] = b0[8 * m + 2 * t + 1] + __uint_as_float(d[t] & 0xffff0000u);
                                }
                            }
                    }
                    PSTAMP(2);
                    constexpr int NR = RC + 2, NS = 6 * NR;
                    u32x4 fb[NFBC];
                    auto ldb = [&](auto sc, int buf) {
                        constexpr int s = decltype(sc)::value, c = s / NR, i = s % NR, kx = c >> 1, kk = c & 1;
                        fb[buf] = lds_read_async<i * MW * 64>(lds0 + mcur + ma0[kx][kk]);
                    };
                    static_for<RDC>([&](auto sc) { ldb(sc, decltype(sc)::value); });
                            static_for<NS>([&](auto sc) {
                        constexpr int s = decltype(sc)::value, c = s / NR, i = s % NR, kx = c >> 1, kk = c & 1;
                        if constexpr (s + RDC < NS) {
                            ldb(std::integral_constant<int, s + RDC>{}, (s + RDC) % NFBC);
                                }
                        if constexpr (s >= NS - 12 && s - (NS - 12) < NPL) {
                            pf_write(std::integral_constant<int, s - (NS - 12)>{}, pnext);
                            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                        }
                        if constexpr (s == 2) PSTAMP(5);
                        if constexpr (s == NS / 2) PSTAMP(6);
                        if constexpr (s == 3 * NS / 4) PSTAMP(7);
                        lds_wait<(NS - 1 - s < RDC ? NS - 1 - s : RDC)>(fb[s % NFBC]);
                        constexpr int nm = (i == 0 || i == NR - 1) ? 1 : ((i == 1 || i == NR - 2) ? 2 : 3);
                        static_for<3>([&](auto kyc) {
                            constexpr int ky = decltype(kyc)::value, j = i - ky;
                            if constexpr (j >= 0 && j < RC)
                                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wreg[(ky * 3 + kx) * 2 + kk]),
                                                                                 __builtin_bit_cast(bf16x8, fb[s % NFBC]), acc[j], 0, 0, 0);
                        });
                        __builtin_amdgcn_sched_group_barrier(0x8, nm, 0);
                    });
                    if constexpr (FIN) {
                        // the head: block output rows (ReLU, bf16: what the unfused head reads back) x head weights, then bias, then
                        // out -- in THIS iteration, not carried over the barrier like the bf16 tile of the plain block: fp32 result
                        // tiles cannot be packed to half their registers, and 64 of them beside the next tile's residual and patch
                        // loads do not fit.  The fragment, residual and prefetch registers of the loop above are free here.
                        // (asm reads: as plain loads these loop invariants are hoisted out of the tile loop and their 8 registers spill)
                        u32x4 fw0 = lds_read_async<0>(lds0 + OFF_FINW + (h * 32 + r) * 16), fw1 = lds_read_async<1024>(lds0 + OFF_FINW + (h * 32 + r) * 16);
                        const float fb = reinterpret_cast<const float *>(smem + OFF_FINB)[r];
                        lds_wait<1>(fw0);
                        lds_wait<0>(fw1);
                        // this lane's channel plane r, pixels ox0 + 4h .. (+ 8q per store); lanes of the padding couts store nowhere
                        const bool lane_ok = r < p.fin_K;
                        const unsigned lane_off = (unsigned)((r * p.H * p.W + g.ox0 + 4 * h) * 4);
#pragma unroll
                        for (int j = 0; j < RC; ++j) {
                            u32x4 o[2];
                            pack_rows16(acc[j], o);
                            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // (an inline constant: no registers)
                            f32x16 d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, o[0]), __builtin_bit_cast(bf16x8, fw0), zero, 0, 0, 0);
                            d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, o[1]), __builtin_bit_cast(bf16x8, fw1), d, 0, 0, 0);
                            const int y = g.oy0 + c0 + j;
                            const bool wrap = y >= p.VH;
                            const int ya = wrap ? y - p.VH : y, bb = wrap ? g.b + 1 : g.b;
                            const bool row_ok = ((unsigned)ya < (unsigned)p.H) & (bb < p.B);
                            const unsigned row_off = (unsigned)(((bb * p.fin_K * p.H + ya) * p.W) * 4);
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const bool ok = lane_ok & row_ok & (g.ox0 + 8 * q + 4 * h < p.W);  // (W % 4 == 0: the four pixels are inside together)
#ifdef BBPC_NOSTORE
                                const unsigned voff = (ok && d[0] == 1.2345f) ? row_off + lane_off : OOB;
#else
                                const unsigned voff = ok ? row_off + lane_off : OOB;
#endif
                                __builtin_amdgcn_raw_buffer_store_b128(u32x4{__float_as_uint(d[4 * q] + fb), __float_as_uint(d[4 * q + 1] + fb),
                                                                             __float_as_uint(d[4 * q + 2] + fb), __float_as_uint(d[4 * q + 3] + fb)},
                                                                       rs_out, (int)voff, 32 * q, BBPC_STORE_AUX);
                            }
                        }
                    }
                } else {
                    static_for<NPL>([&](auto ic) { pf_write(ic, pnext); });
                }
                PSTAMP(3);
                lds_barrier();
                PSTAMP(4);
            }
            if constexpr (!FIN) finish(nloc - 1);
        };
        if (wj < 2) consumer_loop(std::integral_constant<int, 4>{});
        else consumer_loop(std::integral_constant<int, 3>{});
    }
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

__global__ __launch_bounds__(NTHR, 1) void bbpc_kernel(const BBParams p) { bbpc_body<false>(p); }
__global__ __launch_bounds__(NTHR, 1) void bbpc_final_kernel(const BBParams p) { bbpc_body<true>(p); }

hipError_t bbpc_init()
{
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(bbpc_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void *>(bbpc_final_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES_FIN);
}

// 32-bit buffer offsets: both tensors must stay below 2 GB (the engine falls back to the tile form otherwise)
bool bbpc_supported(const BBParams &p)
{
    return (size_t)p.B * p.H * p.W * (size_t)(p.in_cs > p.out_cs ? p.in_cs : p.out_cs) * 2 < 0x7fffffffull;
}

// the block with the head's 1x1 convolution in its epilogue (p.fin_*): fp32 NCHW result below 2 GB, widths in whole float4s
bool bbpc_final_supported(const BBParams &p)
{
    return bbpc_supported(p) && p.fin_K >= 1 && p.fin_K <= 32 && p.W % 4 == 0 && (size_t)p.B * p.fin_K * p.H * p.W * 4 < 0x7fffffffull;
}

hipError_t bbpc_launch(BBParams p, int num_cus, hipStream_t s)
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
    const int grid = p.ntiles < num_cus ? p.ntiles : num_cus;
    if (p.fin_out) {
        if (!bbpc_final_supported(p)) return hipErrorInvalidValue;
        HH_LAUNCH(bbpc_final_kernel, dim3(grid), dim3(NTHR), LDS_BYTES_FIN, s, p);
    } else {
        if (!bbpc_supported(p)) return hipErrorInvalidValue;
        HH_LAUNCH(bbpc_kernel, dim3(grid), dim3(NTHR), LDS_BYTES, s, p);
    }
    return hipGetLastError();
}
