// Decode front end, round 4: stage average + x2 resize + 5x5 NMS + candidate selection in ONE pass over the net's heatmap
// outputs (results.py:225-234, grouping.py:80-83 and the first half of top_k, grouping.py:147-153), for the default path of
// hh_decode (mode 0, sub-threshold work skipped).  It replaces stage_average_kernel + nms_tile_topk_kernel there: the averaged
// half-resolution map is never written to HBM (round 3 wrote its 143 MB per batch of 32 and read it back three times).
//
// One workgroup = one 128x128 full-resolution REGION of one (image, joint) map:
//   1. its 36x36 quarter-res and 68x68 half-res source patches (2-sample halos, clamped at the image border) are loaded once;
//   2. the stage average of the patch is formed in LDS, separably and in bilerp()'s own order of operations, so every value has
//      the bits stage_average_kernel would have stored;
//   3. every 4x4-pixel cell gets an upper bound (the maximum of the 4x4 averaged samples its pixels interpolate, plus the slack
//      of the three roundings) -- the refine scans prune with it -- and a 16x16 SUB-TILE whose cells all stay at or below
//      det_thr is done: nothing in it can survive match_by_tag's `score > det_thr` filter (grouping.py:98-102);
//   4. the remaining sub-tiles (a tenth of the map at ten people per image) are taken one per wavefront: 20x20 full-resolution
//      values from the LDS patch, separable 5x5 maximum, peaks above det_thr appended to the region's list (at most
//      max_num_people per sub-tile: a map's top-k cannot hold more of them), exact cell maxima for the refine scans;
//   5. the region's best max_num_people candidates go to its slots of `cand_key`; topk_merge_kernel ranks the regions' lists.
// Bit-exactness: a full-resolution value is fma(fma(A00, wx0, A01 * wx1), wy0, fma(A10, wx0, A11 * wx1) * wy1) on averaged
// samples A = (fma(fma(q00, ..), ..) + h) / 2, exactly the expressions of stage_average_kernel and bilerp(); equal values are
// ordered by ascending pixel index through the key, as everywhere in this decoder.
#include "decode_dev.h"

namespace {

constexpr int RG = 128;            // region edge (full-res pixels)
constexpr int ST = 16;             // sub-tile edge
constexpr int NS = RG / ST;        // sub-tiles per region row: 8 x 8 = one ballot
constexpr int HP = RG / 2 + 4;     // half-res patch edge (68)
constexpr int AS = HP + 2;         // LDS row stride of the half-res patch; patch column j sits at [j + 1], so that the cell windows
                                   // (patch columns 2cx+1 .. 2cx+4) are two aligned 8-byte reads
constexpr int VW = ST + 4;         // a sub-tile's neighbourhood edge (20)
constexpr int HRR = ST / 2 + 4;    // half-res rows under it (12)
constexpr int WR = 18;             // patch rows per wavefront (4 x 18 >= 68)
constexpr int NQ = WR / 2 + 2;     // quarter-res rows under them (11)
constexpr int CAP = 32 * HH_MAX_PEOPLE + HH_MAX_PEOPLE;  // candidate list: one batch of 32 sub-tiles + a pruned earlier batch
static_assert(NS * NS == 64, "the sub-tile activity mask is one 64-lane ballot");

// Exact x2 bilinear resize (torch: src_index(n, 0.5f, D)): destination index D = 2g reads sources (g-1, g) with weights (0.25, 0.75),
// D = 2g+1 reads (g, g+1) with (0.75, 0.25); D = 0 reads (0, 1) with (1, 0) and the upper index is clamped to n-1.  Below the taps are
// always the STATIC pair of the parity -- so that every index is a compile-time register / a fixed LDS offset -- and the two border
// cases are folded into the weights: D = 0 takes (0, 1) on (-1, 0) and a clamped upper tap takes (1, 0) on (g, g+1).  With finite
// samples both return the one sample that counts, as torch's fma(x0, 1, x1 * 0) and fma(xg, 0.75, xg * 0.25) do (0.25 xg is exact, so
// that sum is xg); the out-of-range tap is some finite in-buffer value times zero.  (Non-finite heatmaps: border pixels may differ.)
__device__ __forceinline__ float wlo(int D, int n) { return (D & 1) ? (((D >> 1) + 1 > n - 1) ? 1.f : 0.75f) : (D <= 0 ? 0.f : 0.25f); }
__device__ __forceinline__ float whi(int D, int n) { return (D & 1) ? (((D >> 1) + 1 > n - 1) ? 0.f : 0.25f) : (D <= 0 ? 1.f : 0.75f); }

// orders the LDS traffic of ONE wavefront (its ds instructions execute in issue order; the compiler must not move them)
__device__ __forceinline__ void wave_lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

#ifdef HH_PEAKS_DEBUG  // phase stamps of a sample of workgroup iterations, read by tools/probes/peaks_probe.hip only
__device__ long long g_peaks_dbg[4096 * 8];
#define PK_STAMP(i) do { if (threadIdx.x == 0 && dbg_slot < 4096) g_peaks_dbg[dbg_slot * 8 + (i)] = (long long)__builtin_readcyclecounter(); } while (0)
#define PK_CLEAR(i) do { if (threadIdx.x == 0 && dbg_slot < 4096) g_peaks_dbg[dbg_slot * 8 + (i)] = 0; } while (0)
#else
#define PK_STAMP(i)
#define PK_CLEAR(i)
#endif

}  // namespace

#ifndef PEAKS_WPS
#define PEAKS_WPS 4  // workgroups per CU = waves per SIMD
#endif
// Persistent workgroups.  The (image, joint, region) list is cut into HH_PEAKS_PARTS contiguous parts (fewer when the list or the
// grid does not divide); workgroup x works on part x % parts, i.e. on an XCD of its own kind (workgroups are dealt to the 8 XCDs
// round-robin: the halo lines neighbouring regions share go through one L2).  A workgroup's first unit is its slot in the part,
// every further one a ticket from the part's counter `ctr[part]` (zero at launch; topk_merge_kernel, the next launch, clears it
// again): regions with people in them take 2-3 times as long as empty ones, and a fixed stride gave every workgroup the same
// region position of each map -- the corner ones finished in half the time of the centre ones.  64 parts, not 8: a ticket costs
// what its counter's contention costs (128 workgroups on one counter: 3.5 us each, 16: hidden behind the region's loads).
__global__ __launch_bounds__(256, PEAKS_WPS) void peaks_region_kernel(const DecodeSrc src, int M, int nrx, int nreg, int nunits, float thr,
                                                                      u64 *__restrict__ cand_key, unsigned short *__restrict__ cellub,
                                                                      unsigned short *__restrict__ supmax, int *__restrict__ ctr)
{
    __shared__ float avgp[HP][AS];
    __shared__ __attribute__((aligned(16))) float hrs[4][HRR][VW];   // per wave: a sub-tile's half-res rows interpolated along x
    __shared__ __attribute__((aligned(16))) float rms[4][VW][VW];    // per wave: 5-wide row maxima (columns 0 .. 15 used)
    __shared__ u64 clist[CAP];
    __shared__ u64 tmpk[HH_MAX_PEOPLE];
    __shared__ u64 s_part[4];
    __shared__ int ccount;
    __shared__ u64 wbest[2][4];
    __shared__ int wpos[2][4];

    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = src.H, W = src.W, hh = H >> 1, wh = W >> 1, hq = H >> 2, wq = W >> 2;
    __shared__ int s_next;
    const int G = gridDim.x;
    const int nparts = (nunits % HH_PEAKS_PARTS == 0 && G % HH_PEAKS_PARTS == 0) ? HH_PEAKS_PARTS : (nunits % 8 == 0 && G % 8 == 0) ? 8 : 1;
    const int part = blockIdx.x % nparts, nwg = G / nparts;  // this workgroup's part, workgroups on it
    const int ubase = part * (nunits / nparts), usize = nunits / nparts;
    int uidx = blockIdx.x / nparts;  // position in the part

    // ---- source samples of one region -> registers.  Wavefront w forms patch rows 18w .. 18w+17 of the lane's column: their 18
    // half-res samples and the 11 quarter-res rows under them at the column's two source columns; the patch's last four columns
    // (64 .. 67) are dealt one element per thread (element e = row e >> 2, column 64 + (e & 3)) with their own four taps.
    float hv[WR], q0[NQ], q1[NQ], hx[2], qx[2][4];
    auto issue = [&](int un) {
        // (integer division runs on the vector pipe: readfirstlane puts the wave-uniform results, and with them the row pointers,
        // back into scalar registers: every load below is scalar base + one shared vector offset)
        const int map = __builtin_amdgcn_readfirstlane(un / nreg), reg = un - map * nreg;
        const int b = __builtin_amdgcn_readfirstlane(map / src.K), k = map - b * src.K;
        const int ry = __builtin_amdgcn_readfirstlane(reg / nrx), rx = reg - ry * nrx;
        const int R0 = (ry * RG) >> 1, C0 = (rx * RG) >> 1;
        const float *__restrict__ q = src.hm_q + (size_t)b * src.hm_q_bs + (size_t)k * hq * wq;
        const float *__restrict__ h = src.hm_h + (size_t)b * src.hm_h_bs + (size_t)k * hh * wh;
        const int cc = min(max(C0 - 2 + lane, 0), wh - 1);
        // source columns of half-res column cc (proper, clamped taps: these index global memory)
        // (the pair is adjacent except at the right border, where both are column wq-1: one 8-byte load from min(t0, wq-2))
        const int g = cc >> 1, t0 = (cc & 1) ? g : max(g - 1, 0);
        const bool dup = t0 >= wq - 1;
        const int tp = min(t0, wq - 2);
        const int rb = R0 - 2 + WR * wv;  // (even) first half-res row of this wavefront
#pragma unroll
        for (int i = 0; i < WR; ++i) hv[i] = ldg(h + (size_t)min(max(rb + i, 0), hh - 1) * wh, cc);
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const float *__restrict__ qr = q + (size_t)min(max((rb >> 1) - 1 + i, 0), hq - 1) * wq;
            const float2 pr = *reinterpret_cast<const float2 *>(reinterpret_cast<const char *>(qr) + (unsigned)(tp << 2));
            q0[i] = dup ? pr.y : pr.x; q1[i] = pr.y;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int e = tid + 256 * j;
            if (e < HP * 4) {
                const int rc = min(max(R0 - 2 + (e >> 2), 0), hh - 1), xc = min(max(C0 - 2 + 64 + (e & 3), 0), wh - 1);
                const int gy = rc >> 1, y0 = (rc & 1) ? gy : max(gy - 1, 0), y1 = (rc & 1) ? min(gy + 1, hq - 1) : (rc <= 0 ? min(1, hq - 1) : gy);
                const int gx = xc >> 1, x0 = (xc & 1) ? gx : max(gx - 1, 0), x1 = (xc & 1) ? min(gx + 1, wq - 1) : (xc <= 0 ? min(1, wq - 1) : gx);
                hx[j] = ldg(h, rc * wh + xc);
                qx[j][0] = ldg(q, y0 * wq + x0); qx[j][1] = ldg(q, y0 * wq + x1);
                qx[j][2] = ldg(q, y1 * wq + x0); qx[j][3] = ldg(q, y1 * wq + x1);
            }
        }
    };
#ifdef HH_PEAKS_DEBUG
    int dbg_iter = 0;
#endif
    while (uidx < usize) {
        const int unit = ubase + uidx;
#ifdef HH_PEAKS_DEBUG
        const int dbg_slot = dbg_iter < 4 && blockIdx.x < 1024 ? (int)blockIdx.x * 4 + dbg_iter : 4096;
        ++dbg_iter;
#endif
        PK_STAMP(0);
        issue(unit);
        // the next unit's ticket rides behind this region's loads: the compiler waits for a returning atomic where it stands (its
        // wave-aggregated form ends in a readfirstlane), and here that wait is the one for the samples
        int ticket = 0;
        if (tid == 0) ticket = nwg + atomicAdd(ctr + part, 1);
        const int map = __builtin_amdgcn_readfirstlane(unit / nreg), reg = unit - map * nreg;
        const int ry0 = __builtin_amdgcn_readfirstlane(reg / nrx);
        const int Y0 = ry0 * RG, X0 = (reg - ry0 * nrx) * RG;  // region origin, full res
        const int R0 = Y0 >> 1, C0 = X0 >> 1, Qy0 = Y0 >> 2, Qx0 = X0 >> 2;
        if (tid == 0) ccount = 0;

        // ---- 1. stage average of the patch (results.py:225-226) in bilerp()'s order: quarter-res rows along x, then along y, + the
        // 1/2-res sample, / 2.  Row rb + i of the wavefront (rb even) reads the static pair (i >> 1) + (i & 1), + 1 of its 11
        // quarter-res rows; rows / columns outside the image hold finite values that are only ever multiplied by zero or enter
        // the cell bounds (which they can only loosen).
        {
            const int rb = R0 - 2 + WR * wv;
            const int cc = min(max(C0 - 2 + lane, 0), wh - 1);
            const float cw0 = (cc & 1) ? 0.75f : (cc <= 0 ? 1.f : 0.25f), cw1 = (cc & 1) ? 0.25f : (cc <= 0 ? 0.f : 0.75f);
            float qh[NQ];
#pragma unroll
            for (int i = 0; i < NQ; ++i) qh[i] = __builtin_fmaf(q0[i], cw0, q1[i] * cw1);
#pragma unroll
            for (int i = 0; i < WR; ++i) {
                const int a = (i >> 1) + (i & 1);
                const float w0 = (i & 1) ? 0.75f : (rb + i <= 0 ? 1.f : 0.25f), w1 = (i & 1) ? 0.25f : (rb + i <= 0 ? 0.f : 0.75f);
                if (WR * wv + i < HP) avgp[WR * wv + i][lane + 1] = (__builtin_fmaf(qh[a], w0, qh[a + 1] * w1) + hv[i]) / 2.0f;
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int e = tid + 256 * j;
                if (e < HP * 4) {
                    const int rc = min(max(R0 - 2 + (e >> 2), 0), hh - 1), xc = min(max(C0 - 2 + 64 + (e & 3), 0), wh - 1);
                    const float wx0 = (xc & 1) ? 0.75f : (xc <= 0 ? 1.f : 0.25f), wx1 = (xc & 1) ? 0.25f : (xc <= 0 ? 0.f : 0.75f);
                    const float wy0 = (rc & 1) ? 0.75f : (rc <= 0 ? 1.f : 0.25f), wy1 = (rc & 1) ? 0.25f : (rc <= 0 ? 0.f : 0.75f);
                    const float a = __builtin_fmaf(qx[j][0], wx0, qx[j][1] * wx1);
                    const float d = __builtin_fmaf(qx[j][2], wx0, qx[j][3] * wx1);
                    avgp[e >> 2][64 + (e & 3) + 1] = (__builtin_fmaf(a, wy0, d * wy1) + hx[j]) / 2.0f;
                }
            }
        }
        PK_STAMP(1);
        lds_barrier();
        PK_STAMP(2);

        // ---- 2. cell bounds and the sub-tiles that can hold a pixel above det_thr.  Cell (cy, cx) of the region = full-res rows
        // 4cy .. 4cy+3, which interpolate half-res rows 2cy-1 .. 2cy+2 = patch rows 2cy+1 .. 2cy+4.  Thread = (cell column cx, the four
        // cell rows of sub-tile row cyg): ten patch rows, the maximum of a sub-tile = its four rows here x four neighbouring lanes.
        float ub[4];
        const int cx = tid & 31, cyg = tid >> 5;
        {
            float rmx[10];
            float2 lo[10], hi[10];
#pragma unroll
            for (int a = 0; a < 10; ++a) {
                lo[a] = *reinterpret_cast<const float2 *>(&avgp[8 * cyg + 1 + a][2 * cx + 2]);
                hi[a] = *reinterpret_cast<const float2 *>(&avgp[8 * cyg + 1 + a][2 * cx + 4]);
            }
            __builtin_amdgcn_sched_barrier(0);  // (left alone the scheduler waits for every row where it is read: ten LDS round trips)
#pragma unroll
            for (int a = 0; a < 10; ++a) rmx[a] = fmaxf(fmaxf(lo[a].x, lo[a].y), fmaxf(hi[a].x, hi[a].y));
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float m = fmaxf(fmaxf(rmx[2 * j], rmx[2 * j + 1]), fmaxf(rmx[2 * j + 2], rmx[2 * j + 3]));
                ub[j] = m + 4e-7f * fabsf(m);  // a pixel is a convex combination of those samples, rounded three times
            }
            float sm = fmaxf(fmaxf(ub[0], ub[1]), fmaxf(ub[2], ub[3]));
            sm = fmaxf(sm, __shfl_xor(sm, 1));
            sm = fmaxf(sm, __shfl_xor(sm, 2));
            {   // the largest bound of every 8x8-cell super (2 x 2 sub-tiles: lanes cx .. cx+7 of both cell-row groups of this wavefront):
                // what the refine scans open a super on.  (Bounds, not the exact maxima the active sub-tiles write below: still upper bounds.)
                float s8 = fmaxf(sm, __shfl_xor(sm, 4));
                s8 = fmaxf(s8, __shfl_xor(s8, 32));
                const int Sy = (Qy0 >> 3) + wv, Sx = (Qx0 >> 3) + (lane >> 3), nsy = (hq + 7) >> 3, nsx = (wq + 7) >> 3;
                if (lane < 32 && (lane & 7) == 0 && Sy < nsy && Sx < nsx) supmax[((size_t)map * nsy + Sy) * nsx + Sx] = bf16_ceil(s8);
            }
            // sub-tile (row cyg, column cx >> 2): bit cyg * 8 + (cx >> 2), voted by the lane with cx & 3 == 0
            const bool act = ((cx & 3) == 0) && (Y0 + ST * cyg < H) && (X0 + 4 * cx < W) && !(sm <= thr);
            const u64 bal = __ballot(act);  // lanes 0, 4, .., 28 -> sub-tile row 2 wv, lanes 32, 36, .. -> row 2 wv + 1
            if (lane == 0) {
                u64 m8 = 0;
#pragma unroll
                for (int i = 0; i < 16; ++i) m8 |= ((bal >> (4 * i)) & 1ull) << i;
                s_part[wv] = m8 << (16 * wv);
            }
        }
        lds_barrier();
        PK_STAMP(3);
        const u64 smask = s_part[0] | s_part[1] | s_part[2] | s_part[3];
        const int mlo = __builtin_amdgcn_readfirstlane((int)(unsigned)smask), mhi = __builtin_amdgcn_readfirstlane((int)(unsigned)(smask >> 32));
        const u64 mask = ((u64)(unsigned)mhi << 32) | (u64)(unsigned)mlo;
        {   // cells of finished sub-tiles: the bound is all the refine scans get; the others are written exactly below
            const bool fin = !((mask >> (cyg * NS + (cx >> 2))) & 1ull);
            if (fin && Qx0 + cx < wq)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (Qy0 + 4 * cyg + j < hq) cellub[((size_t)map * hq + Qy0 + 4 * cyg + j) * wq + Qx0 + cx] = bf16_ceil(ub[j]);
        }
        const size_t obase = ((size_t)map * nreg + reg) * M;
        if (mask == 0ull) {  // (block-uniform) nothing above det_thr anywhere in the region
            if (tid < M) cand_key[obase + tid] = 0ull;
            if (tid == 0) s_next = ticket;
            lds_barrier();  // the next region's average overwrites the patch / ccount / s_part
            uidx = s_next;
            PK_CLEAR(4);
            PK_STAMP(6);
            continue;
        }

        // the best M of the n candidates in clist -> dst[0 .. M) (0 = empty), by every thread of the workgroup
        auto select_top = [&](int n, u64 *dst) {
            if (n <= 256) {
                if (tid < n) {
                    const u64 me = clist[tid];
                    int rank = 0;
                    for (int i = 0; i < n; ++i) rank += clist[i] > me;
                    if (rank < M) dst[rank] = me;
                }
                if (tid >= n && tid < M) dst[tid] = 0ull;
                return;
            }
            for (int r = 0; r < M; ++r) {  // rare: M rounds of workgroup-wide arg-max (the exchange buffer alternates: one barrier per round)
                u64 best = 0ull;
                int pos = -1;
                for (int i = tid; i < n; i += 256) {
                    const u64 kk = clist[i];
                    if (kk > best) { best = kk; pos = i; }
                }
                const u64 wb = wave_max_u64(best);
                if (best == wb && best != 0ull) { wbest[r & 1][wv] = wb; wpos[r & 1][wv] = pos; }
                else if (lane == 0 && wb == 0ull) { wbest[r & 1][wv] = 0ull; wpos[r & 1][wv] = -1; }
                lds_barrier();
                u64 g = 0ull;
                int gp = -1;
#pragma unroll
                for (int w = 0; w < 4; ++w)
                    if (wbest[r & 1][w] > g) { g = wbest[r & 1][w]; gp = wpos[r & 1][w]; }
                if (gp >= 0 && pos == gp) clist[gp] = 0ull;  // the owner retires its candidate; only the owner ever re-reads that slot
                if (tid == 0) dst[r] = g;
            }
        };

        // ---- 3. one wavefront per remaining sub-tile, in two batches (sub-tile rows 0-3, 4-7) so that the list never needs more than
        // a batch's 32 x M slots + the M kept from the first batch
        float (*hr)[VW] = hrs[wv];
        float (*rm)[VW] = rms[wv];
        const int l20 = (lane >= 20) + (lane >= 40) + (lane >= 60), xl = lane - 20 * l20;  // lane -> (row phase, column) of 3 x 20 values
#pragma unroll 1
        for (int batch = 0; batch < 2; ++batch) {
            u64 todo = batch ? (mask >> 32) << 32 : mask & 0xffffffffull;
            if (batch) {
                if (todo == 0ull) break;
                lds_barrier();  // every append of the first batch is in
                const int n0 = ccount;
                if (n0 + __popcll(todo) * M > CAP) {  // (block-uniform; rare) keep the first batch's best M only
                    select_top(n0, tmpk);
                    lds_barrier();
                    if (tid < M) clist[tid] = tmpk[tid];
                    if (tid == 0) ccount = M;  // (empty slots among them are zero keys: they rank last and are dropped again)
                    lds_barrier();
                }
            }
#pragma unroll 1
            for (int ord = 0; todo; ++ord) {
                const int s = __builtin_ctzll(todo);
                todo &= todo - 1;
                if ((ord & 3) != wv) continue;
                const int sy = s >> 3, sx = s & 7, Ys = Y0 + ST * sy, Xs = X0 + ST * sx;
                // does the 20x20 neighbourhood leave the image? (most sub-tiles: no, and the -inf padding selects are skipped)
                const bool edge = Ys == 0 || Xs == 0 || Ys + ST + 2 > H || Xs + ST + 2 > W;
                {   // (a) the 12 half-res rows under the neighbourhood (patch rows 8sy .. 8sy+11), interpolated along x at its 20 columns:
                    // lane = (row phase, column X = Xs - 2 + xl); X has the parity of xl, its static source pair starts at patch
                    // column (X >> 1) - 1 + (X & 1) - (C0 - 2), stored one further right
                    const int X = Xs - 2 + xl;
                    const int pc = (X >> 1) + (X & 1) - C0 + 2;
                    const float w0 = wlo(X, wh), w1 = whi(X, wh);
                    if (lane < 60) {
                        float s0[4], s1[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const float *row = &avgp[8 * sy + 3 * j + l20][pc];
                            s0[j] = row[0]; s1[j] = row[1];
                        }
                        __builtin_amdgcn_sched_barrier(0);  // (all reads in flight before the first use: one LDS round trip per pass)
#pragma unroll
                        for (int j = 0; j < 4; ++j) hr[3 * j + l20][xl] = __builtin_fmaf(s0[j], w0, s1[j] * w1);
                    }
                }
                wave_lds_sync();
                auto slide = [](const float (&in)[8], float (&out)[4]) {  // out[i] = max(in[i .. i+4])
                    float p2[7], p4[5];
#pragma unroll
                    for (int i = 0; i < 7; ++i) p2[i] = fmaxf(in[i], in[i + 1]);
#pragma unroll
                    for (int i = 0; i < 5; ++i) p4[i] = fmaxf(p2[i], p2[i + 2]);
#pragma unroll
                    for (int i = 0; i < 4; ++i) out[i] = fmaxf(p4[i], in[i + 4]);
                };
                // (b) full-res rows of the neighbourhood, 8 columns per lane (lane = (row, 4-column strip)), and their 5-wide maxima:
                // row Yl (Y = Ys - 2 + Yl, parity of Yl) reads interpolated rows (Yl >> 1) + (Yl & 1), + 1
#pragma unroll
                for (int pass = 0; pass < 2; ++pass) {  // rows 0 .. 15 (all lanes), then rows 16 .. 19 (16 lanes)
                    const int Yl = pass * 16 + (lane >> 2), x0 = (lane & 3) * 4, Y = Ys - 2 + Yl;
                    if (pass == 0 || lane < 16) {
                        const int ra = (Yl >> 1) + (Yl & 1);
                        const float w0 = wlo(Y, hh), w1 = whi(Y, hh);
                        const float4 a0 = *reinterpret_cast<const float4 *>(&hr[ra][x0]), a1 = *reinterpret_cast<const float4 *>(&hr[ra][x0 + 4]);
                        const float4 b0 = *reinterpret_cast<const float4 *>(&hr[ra + 1][x0]), b1 = *reinterpret_cast<const float4 *>(&hr[ra + 1][x0 + 4]);
                        __builtin_amdgcn_sched_barrier(0);
                        float in[8], out[4];
                        in[0] = __builtin_fmaf(a0.x, w0, b0.x * w1); in[1] = __builtin_fmaf(a0.y, w0, b0.y * w1);
                        in[2] = __builtin_fmaf(a0.z, w0, b0.z * w1); in[3] = __builtin_fmaf(a0.w, w0, b0.w * w1);
                        in[4] = __builtin_fmaf(a1.x, w0, b1.x * w1); in[5] = __builtin_fmaf(a1.y, w0, b1.y * w1);
                        in[6] = __builtin_fmaf(a1.z, w0, b1.z * w1); in[7] = __builtin_fmaf(a1.w, w0, b1.w * w1);
                        if (edge) {  // outside the image: -inf (the max-pool's padding)
                            const bool yin = Y >= 0 && Y < H;
#pragma unroll
                            for (int i = 0; i < 8; ++i) {
                                const int X = Xs - 2 + x0 + i;
                                in[i] = (yin && X >= 0 && X < W) ? in[i] : -INFINITY;
                            }
                        }
                        slide(in, out);
                        *reinterpret_cast<float4 *>(&rm[Yl][x0]) = make_float4(out[0], out[1], out[2], out[3]);
                    }
                }
                wave_lds_sync();
                // (c) 5-high maxima along y and the pixels themselves: lane = (column pcx, rows 4 pry .. 4 pry + 3) of the sub-tile
                const int pcx = lane & 15, pry = lane >> 4;
                float cen[4], mx[4];
                {
                    float in[8], hc[4];
#pragma unroll
                    for (int i = 0; i < 8; ++i) in[i] = rm[4 * pry + i][pcx];
#pragma unroll
                    for (int i = 0; i < 4; ++i) hc[i] = hr[2 * pry + 1 + i][pcx + 2];
                    __builtin_amdgcn_sched_barrier(0);
                    slide(in, mx);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {  // pixel row Yl = 4 pry + 2 + j: interpolated rows 2 pry + 1 + ((j + 1) >> 1), + 1
                        const int Y = Ys + 4 * pry + j;
                        cen[j] = __builtin_fmaf(hc[(j + 1) >> 1], wlo(Y, hh), hc[((j + 1) >> 1) + 1] * whi(Y, hh));
                    }
                }
                wave_lds_sync();  // (the next sub-tile of this wave overwrites hr / rm)
                // (d) peaks above det_thr -> the region's list
                const int Yp = Ys + 4 * pry, Xp = Xs + pcx;
                unsigned pm = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) pm |= (mx[j] == cen[j] && cen[j] > thr && Yp + j < H && Xp < W) ? (1u << j) : 0u;
                const int cnt = __popc(pm);
                auto key_of = [&](int j) {  // make_key() of a positive value
                    return ((u64)(__float_as_uint(cen[j]) | 0x80000000u) << 32) | (u64)(0xffffffffu - (unsigned)((Yp + j) * W + Xp));
                };
                if (__ballot(cnt != 0)) {  // wave-uniform
                    int incl = cnt;
#pragma unroll
                    for (int off = 1; off < 64; off <<= 1) {
                        const int o = __shfl_up(incl, off);
                        if (lane >= off) incl += o;
                    }
                    const int tot = __builtin_amdgcn_readlane(incl, 63);
                    const int take = tot < M ? tot : M;
                    int base = 0;
                    if (lane == 0) base = atomicAdd(&ccount, take);
                    base = __builtin_amdgcn_readfirstlane(base);
                    if (tot <= M) {
                        int pos = base + incl - cnt;
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (pm & (1u << j)) clist[pos++] = key_of(j);
                    } else {  // more peaks than a map's top-k can hold (plateaus, noise maps): the sub-tile's best M, by rounds of wave arg-max
                        u64 keys[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) keys[j] = (pm & (1u << j)) ? key_of(j) : 0ull;
                        for (int r = 0; r < M; ++r) {
                            u64 best = keys[0];
#pragma unroll
                            for (int j = 1; j < 4; ++j) best = keys[j] > best ? keys[j] : best;
                            const u64 wb = wave_max_u64(best);
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                if (keys[j] == wb) keys[j] = 0ull;  // keys are unique: one owner
                            if (lane == 0) clist[base + r] = wb;
                        }
                    }
                }
                // (e) exact maxima of the sub-tile's 4x4 cells (lanes pcx .. pcx+3 of one row group hold a cell's 16 pixels; pixels outside
                // the image only occur in cells outside the image, which are not written)
                float m4 = fmaxf(fmaxf(cen[0], cen[1]), fmaxf(cen[2], cen[3]));
                m4 = fmaxf(m4, __shfl_xor(m4, 1));
                m4 = fmaxf(m4, __shfl_xor(m4, 2));
                const int Qy = (Ys >> 2) + pry, Qx = (Xs >> 2) + (pcx >> 2);
                if ((pcx & 3) == 0 && Qy < hq && Qx < wq) cellub[((size_t)map * hq + Qy) * wq + Qx] = bf16_ceil(m4);
            }
        }
        lds_barrier();
        PK_STAMP(4);

        // ---- 4. the region's best M candidates -> its slots (empty slots = 0)
        select_top(ccount, cand_key + obase);
        if (tid == 0) s_next = ticket;
        lds_barrier();  // the next region's average overwrites the patch / the list / ccount / s_part
        uidx = s_next;
        PK_STAMP(6);
    }
}

int peaks_regions(int H, int W) { return ((H + RG - 1) / RG) * ((W + RG - 1) / RG); }

hipError_t launch_peaks(const DecodeSrc &src, int M, unsigned long long *cand_key, float *cellmax, unsigned short *supmax, float thr, int *ctr,
                        hipStream_t s)
{
    const int nrx = (src.W + RG - 1) / RG, nreg = peaks_regions(src.H, src.W), nunits = src.B * src.K * nreg;
    static int grid_max = 0;  // PEAKS_WPS 40 KB workgroups per CU
    if (!grid_max) {
        int dev = 0, cus = 0;
        (void)hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        grid_max = PEAKS_WPS * cus;
    }
    const int grid = nunits < grid_max ? nunits : grid_max;
    hipLaunchKernelGGL(peaks_region_kernel, dim3((unsigned)grid), dim3(256), 0, s, src, M, nrx, nreg, nunits, thr, cand_key,
                       reinterpret_cast<unsigned short *>(cellmax), supmax, ctr);
    return hipGetLastError();
}
