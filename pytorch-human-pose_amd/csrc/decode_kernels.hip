// Decode kernels: HBM/L2-bound integer + fp32 work, no MFMA.  Built with -ffp-contract=off;
// every fused multiply-add below is explicit because the results must be bit-identical to
// the reference's torch-CPU / numpy arithmetic (see oracle/decode_oracle.c for the
// experimentally pinned formulas).
#include "decode_dev.h"

#include <utility>

// ------------------------------------------------------------------ stage average
// results.py:225-226: match_heatmaps_size (1/4 -> 1/2) then torch.stack(...).mean(dim=0)
__global__ __launch_bounds__(256) void stage_average_kernel(const float *hm_q, int64_t hm_q_bs, const float *hm_h, int64_t hm_h_bs,
                                                            float *avg, int K, int hq, int wq, float sy, float sx)
{
    const int hh = 2 * hq, wh = 2 * wq;
    const int k = blockIdx.x, b = blockIdx.y, band = blockIdx.z, nband = gridDim.z;
    const float *q = hm_q + (size_t)b * hm_q_bs + (size_t)k * hq * wq;
    const float *hsrc = hm_h + (size_t)b * hm_h_bs + (size_t)k * hh * wh;
    float *dst = avg + ((size_t)b * K + k) * hh * wh;
    for (int x = threadIdx.x; x < wh; x += 256) {
        const Lin lx = src_index(wq, sx, x);
        for (int y = band; y < hh; y += nband) {
            const float up = bilerp(q, wq, src_index(hq, sy, y), lx);
            dst[(size_t)y * wh + x] = (up + hsrc[(size_t)y * wh + x]) / 2.0f;
        }
    }
}

hipError_t launch_stage_average(const float *hm_q, int64_t hm_q_bs, const float *hm_h, int64_t hm_h_bs, float *avg, int B,
                                int K, int hq, int wq, hipStream_t s)
{
    const float sy = (float)hq / (float)(2 * hq), sx = (float)wq / (float)(2 * wq);
    hipLaunchKernelGGL(stage_average_kernel, dim3(K, B, 8), dim3(256), 0, s, hm_q, hm_q_bs, hm_h, hm_h_bs, avg, K, hq, wq, sy, sx);
    return hipGetLastError();
}

// ------------------------------------------------------------------ NMS + per-tile top-M
// grouping.py:80-83 (5x5 max-pool NMS: hm * (pool(hm) == hm)) and the first half of
// top_k (grouping.py:147-153).  One workgroup = one 64x64 full-resolution tile of one (b,k)
// map, computed from L2-resident low-res data; separable 5x5 max through LDS; then M rounds
// of workgroup-wide arg-max (wave shuffles + one LDS exchange per round).
#ifdef HH_NMS_DEBUG  // phase stamps of a sample of workgroups, read by tools/probes/nms_probe.hip only
__device__ long long g_nms_dbg[4096 * 8];
#define NMS_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x >= 8192 && blockIdx.x < 8192 + 4096) g_nms_dbg[(blockIdx.x - 8192) * 8 + (i)] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define NMS_STAMP(i)
#endif

__global__ __launch_bounds__(256) void nms_tile_topk_kernel(const DecodeSrc src, int M, int tiles_x, int ntile, u64 *__restrict__ cand_key,
                                                            float *__restrict__ cand_val, float *__restrict__ cellmax, float skip_thr)
{
    // Geometry: a 60x60 tile has a 64x64 halo'ed neighbourhood, so in every pass a wave's 64 lanes are 64 columns (or 64
    // rows x 4 strips are the 256 threads) with nobody idle.
    constexpr int TS = HH_NMS_TILE, HS = TS + 4, PR = TS / 2 + 4, SL = TS / 4;  // SL = outputs of one sliding-window strip
    static_assert(HS == 64 && SL * 4 == TS, "the passes below map 64 lanes to the 64 halo'ed columns");
    __shared__ float v[HS][HS + 1];
    __shared__ float rm[HS][TS + 1];
    __shared__ u64 wbest[2][4];
    __shared__ u64 clist[256];
    __shared__ int ccount, nfilled, need_rows;
    __shared__ float smax[4];
    float (*hrow)[HS] = reinterpret_cast<float (*)[HS]>(&rm[0][0]);        // [PR][HS] horizontally interpolated half-res rows
    float (*patch)[PR + 1] = reinterpret_cast<float (*)[PR + 1]>(&rm[0][0]);  // [PR][PR+1] half-res patch (generic scales)
    static_assert(sizeof(float) * PR * HS <= sizeof(rm) && sizeof(float) * PR * (PR + 1) <= sizeof(rm), "aliases fit");
    // XCD-aware work order: workgroups go to the 8 XCDs round-robin by linear block id.  Each XCD takes a contiguous eighth of
    // the (image, joint, tile) list, i.e. whole maps: all tiles of a map, with the 128-byte lines their 2-pixel halos share,
    // go through one L2.
    const int total = gridDim.x;
    const int unit = (total % 8 == 0) ? (blockIdx.x % 8) * (total / 8) + blockIdx.x / 8 : blockIdx.x;
    const int tile = unit % ntile, k = (unit / ntile) % src.K, b = unit / (ntile * src.K);
    const int ty = tile / tiles_x, tx = tile % tiles_x;
    const int y0 = ty * TS, x0 = tx * TS;
    const int tid = threadIdx.x;
    NMS_STAMP(0);
    if (tid == 0) { ccount = 0; nfilled = 0; need_rows = 0; }  // (several barriers before their first use)

    if (src.mode == 0 && src.scale_h2 == 0.5f && src.scale_w2 == 0.5f) {
        // The stage average is exactly half resolution, so torch's source indices and weights are fixed patterns: even
        // X = 2c reads half-res (c-1, c) with weights (0.25, 0.75) -- (0, 1) with (1, 0) at X = 0 --, odd X = 2c+1 reads
        // (c, c+1) with (0.75, 0.25), upper index clamped; rows alike.  Separable, in bilerp()'s order (rows first
        // interpolated along x, then along y, the same fmaf expressions: bit-identical):
        //   pass A  lane = full-res column: hrow[r][X] for the 34 half-res rows of the neighbourhood, straight from global;
        //   pass B  wave = 16 full-res rows (row indices and weights are wave-uniform scalars), lane = column.
        const int hh = src.H >> 1, wh = src.W >> 1;
        const float *img = src.avg + ((size_t)b * src.K + k) * hh * wh;
        const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
        const int py0 = src_index(hh, 0.5f, max(y0 - 2, 0)).i0, pry = (y0 - 2) / 2 - py0;  // hrow row of half-res row (y0-2)/2
        const int X = x0 - 2 + lane, g = X >> 1;  // lane's column; g = floor(X / 2)
        const bool xb = X <= 0, xodd = X & 1;
        const int ca = min(xodd ? max(g, 0) : (xb ? max(g, 0) : g - 1), wh - 1), cb = min(ca + 1, wh - 1);
        const float wxa = xodd ? 0.75f : (xb ? 1.f : 0.25f), wxb = xodd ? 0.25f : (xb ? 0.f : 0.75f);
        constexpr int NR = (PR + 3) / 4;  // rows per wave; every load is issued before the first use (one round trip, not NR)
        float ga[NR], gb[NR];
#pragma unroll
        for (int t = 0; t < NR; ++t) {
            const float *row = img + (size_t)min(py0 + min(wv + 4 * t, PR - 1), hh - 1) * wh;
            ga[t] = row[ca]; gb[t] = row[cb];
        }
        float tmax = -INFINITY;  // the largest half-res value any pixel of the tile or its halo interpolates
#pragma unroll
        for (int t = 0; t < NR; ++t) tmax = fmaxf(tmax, fmaxf(ga[t], gb[t]));
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) tmax = fmaxf(tmax, __shfl_xor(tmax, off));
        if (lane == 0) smax[wv] = tmax;
        lds_barrier();
        NMS_STAMP(1);
        tmax = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
        if (tmax + 4e-7f * fabsf(tmax) <= skip_thr) {
            // Inactive tile (three of four at 10 people per image): every full-resolution value in it is a convex combination of
            // half-res values <= tmax (plus a few ulps of rounding, covered by the slack), so no pixel can pass det_thr and nothing
            // this tile could emit survives match_by_tag's score filter (grouping.py:98-102).  It emits no candidates, and the
            // refine kernel gets ONE upper bound for all its 4x4 cells, the tile's: such a cell is only ever looked at by a scan
            // whose best value so far is below det_thr, and this path -- two loads per row, a maximum, 225 stores -- is what most
            // workgroups of the launch run (per-cell bounds from the interpolated rows cost it a third more).
            if (tid < M) cand_key[((((size_t)b * src.K + k) * ntile) + tile) * M + tid] = 0ull;
            if (tid < (TS / 4) * (TS / 4)) {
                const int cy = tid / (TS / 4), cx = tid % (TS / 4);
                const int Y = y0 + 4 * cy, X = x0 + 4 * cx;
                if (Y < src.H && X < src.W)
                    reinterpret_cast<unsigned short *>(cellmax)[(((size_t)b * src.K + k) * (src.H >> 2) + (Y >> 2)) * (src.W >> 2) + (X >> 2)] =
                        bf16_ceil(tmax + 4e-7f * fabsf(tmax));
            }
            return;
        }
#pragma unroll
        for (int t = 0; t < NR; ++t)
            if (wv + 4 * t < PR) hrow[wv + 4 * t][lane] = __builtin_fmaf(ga[t], wxa, gb[t] * wxb);
        lds_barrier();
        const bool xin = X >= 0 && X < src.W;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int Yl = 16 * wv + j, Y = y0 - 2 + Yl, pr = pry + (Yl >> 1);
            const bool yb = Y <= 0;
            int ra;
            float wa, wb;
            if (j & 1) { ra = max(pr, 0); wa = 0.75f; wb = 0.25f; }
            else { ra = yb ? max(pr, 0) : pr - 1; wa = yb ? 1.f : 0.25f; wb = yb ? 0.f : 0.75f; }
            const float val = __builtin_fmaf(hrow[ra][lane], wa, hrow[ra + 1][lane] * wb);
            v[Yl][lane] = (xin && Y >= 0 && Y < src.H) ? val : -INFINITY;
        }
        lds_barrier();  // hrow (aliased on rm) is dead from here
        NMS_STAMP(2);
    } else if (src.mode == 0) {
        // other scales: stage the half-res neighbourhood once (coalesced), then every full-res value of the tile is the
        // same bilinear expression as heat_at(), evaluated from LDS
        const int hh = src.H >> 1, wh = src.W >> 1;
        const int py0 = src_index(hh, src.scale_h2, max(y0 - 2, 0)).i0, px0 = src_index(wh, src.scale_w2, max(x0 - 2, 0)).i0;
        const float *img = src.avg + ((size_t)b * src.K + k) * hh * wh;
        for (int i = tid; i < PR * PR; i += 256) {
            const int r = i / PR, c = i % PR;
            patch[r][c] = img[(size_t)min(py0 + r, hh - 1) * wh + min(px0 + c, wh - 1)];
        }
        lds_barrier();
        for (int i = tid; i < HS * HS; i += 256) {
            const int ly = i / HS, lx = i % HS;
            const int Y = y0 - 2 + ly, X = x0 - 2 + lx;
            float val = -INFINITY;
            if (Y >= 0 && Y < src.H && X >= 0 && X < src.W) {
                const Lin a = src_index(hh, src.scale_h2, Y), c = src_index(wh, src.scale_w2, X);
                const float *r0 = patch[a.i0 - py0], *r1 = patch[a.i1 - py0];
                const float t0 = __builtin_fmaf(r0[c.i0 - px0], c.w0, r0[c.i1 - px0] * c.w1);
                const float t1 = __builtin_fmaf(r1[c.i0 - px0], c.w0, r1[c.i1 - px0] * c.w1);
                val = __builtin_fmaf(t0, a.w0, t1 * a.w1);
            }
            v[ly][lx] = val;
        }
        lds_barrier();  // patch (aliased on rm) is dead from here
    } else {
        for (int i = tid; i < HS * HS; i += 256) {
            const int ly = i / HS, lx = i % HS;
            const int Y = y0 - 2 + ly, X = x0 - 2 + lx;
            v[ly][lx] = (Y >= 0 && Y < src.H && X >= 0 && X < src.W) ? heat_at(src, b, k, Y, X) : -INFINITY;
        }
        lds_barrier();
    }
    if (src.mode == 0 && tid < (TS / 4) * (TS / 4)) {  // exact maximum of every 4x4 full-res cell: the refine kernel prunes with it
        const int cy = tid / (TS / 4), cx = tid % (TS / 4);
        const int Y = y0 + 4 * cy, X = x0 + 4 * cx;
        if (Y < src.H && X < src.W) {
            float m = -INFINITY;
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int c = 0; c < 4; ++c) m = fmaxf(m, v[2 + 4 * cy + a][2 + 4 * cx + c]);
            // kept as bf16 rounded UP: the refine kernel only needs an upper bound of the cell, and reads it for every scan
            reinterpret_cast<unsigned short *>(cellmax)[(((size_t)b * src.K + k) * (src.H >> 2) + (Y >> 2)) * (src.W >> 2) + (X >> 2)] = bf16_ceil(m);
        }
    }
    // separable 5x5 maximum with sliding windows in registers: SL outputs from SL+4 inputs (pair maxima, then pairs of
    // pairs, then the fifth element) instead of 5 LDS reads + 4 max per output
    auto slide = [](const float (&in)[SL + 4], float (&out)[SL]) {
        float p2[SL + 3], p4[SL + 1];
#pragma unroll
        for (int i = 0; i < SL + 3; ++i) p2[i] = fmaxf(in[i], in[i + 1]);
#pragma unroll
        for (int i = 0; i < SL + 1; ++i) p4[i] = fmaxf(p2[i], p2[i + 2]);
#pragma unroll
        for (int i = 0; i < SL; ++i) out[i] = fmaxf(p4[i], in[i + 4]);
    };
    {  // rows: 64 x 4 strips of SL outputs, one per thread
        const int ly = tid >> 2, lx0 = (tid & 3) * SL;
        float in[SL + 4], out[SL];
#pragma unroll
        for (int i = 0; i < SL + 4; ++i) in[i] = v[ly][lx0 + i];
        slide(in, out);
#pragma unroll
        for (int i = 0; i < SL; ++i) rm[ly][lx0 + i] = out[i];
    }
    lds_barrier();
    NMS_STAMP(3);
    float vals[SL];
    const int mpx = tid & 63, mpy0 = (tid >> 6) * SL;  // this thread's pixels: column mpx (< TS), rows mpy0 .. mpy0+SL-1
    // its first `nin` pixels are inside the image
    const int nin = (mpx < TS && x0 + mpx < src.W) ? min(SL, src.H - (y0 + mpy0)) : 0;
    // the sort key of pixel j; only built for the few pixels that need one (positive peaks, the rare generic rounds)
    auto key_of = [&](int j) { return make_key(vals[j], (unsigned)((y0 + mpy0 + j) * src.W + x0 + mpx)); };
    if (mpx < TS) {
        float in[SL + 4], out[SL];
#pragma unroll
        for (int i = 0; i < SL + 4; ++i) in[i] = rm[mpy0 + i][mpx];
        slide(in, out);
#pragma unroll
        for (int j = 0; j < SL; ++j) {
            const float c = v[mpy0 + j + 2][mpx + 2];
            vals[j] = c * ((out[j] == c) ? 1.0f : 0.0f);
        }
    }
    const size_t obase = ((((size_t)b * src.K + k) * ntile) + tile) * M;
    // The tile's ordering is: positive peaks (value desc), then zero-valued pixels (index asc), then negative
    // peaks.  Fast path: positives are compacted and ranked in LDS, zeros are taken row by row with ballots;
    // the generic M-round arg-max below only runs for what is left (negative peaks) or on list overflow.
    float (*nv)[TS + 1] = rm;  // reuse (rare path below): NMS'ed values of the tile
    NMS_STAMP(4);
    {  // compact the positive peaks: per-thread counts -> wave prefix sum -> one LDS atomic per wave for the base slot
        // (tile skipping on: only peaks above det_thr can ever matter -- match_by_tag filters `score > det_thr`, grouping.py:98-102 --,
        // so the background's local maxima, a hundred per tile on noisy maps, are not compacted, ranked or written)
        const float pos_thr = skip_thr > 0.f ? skip_thr : 0.f;
        unsigned pm = 0;
#pragma unroll
        for (int j = 0; j < SL; ++j) pm |= (j < nin && vals[j] > pos_thr) ? (1u << j) : 0u;
        const int cnt = __popc(pm);
        int incl = cnt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(incl, off);
            if ((tid & 63) >= off) incl += o;
        }
        const int total = __builtin_amdgcn_readlane(incl, 63);
        int base = 0;
        if (total) {  // wave-uniform
            if ((tid & 63) == 0) base = atomicAdd(&ccount, total);
            base = __builtin_amdgcn_readfirstlane(base);
        }
        int pos = base + incl - cnt;
        const unsigned idx0 = (unsigned)((y0 + mpy0) * src.W + x0 + mpx);
#pragma unroll
        for (int j = 0; j < SL; ++j)
            if (pm & (1u << j)) {  // key of a positive value: make_key() without its special cases
                if (pos < 256)
                    clist[pos] = ((u64)(__float_as_uint(vals[j]) | 0x80000000u) << 32) | (u64)(0xffffffffu - (idx0 + (unsigned)(j * src.W)));
                ++pos;
            }
    }
    lds_barrier();
    const int npos = ccount;
    NMS_STAMP(5);
    int start = 0;          // first output slot the generic rounds must fill
    bool generic_all = npos > 256;
    if (!generic_all) {
        if (tid < npos) {
            const u64 me = clist[tid];
            int rank = 0;
            for (int i = 0; i < npos; ++i) rank += clist[i] > me;
            if (rank < M) {
                cand_key[obase + rank] = me;
                cand_val[obase + rank] = __uint_as_float((unsigned)(me >> 32) & 0x7fffffffu);
            }
        }
        start = npos < M ? npos : M;
        if (skip_thr >= 0.f) {  // tile skipping on: the rest of the list stays empty, as a skipped tile's whole list does
            if (tid >= start && tid < M) cand_key[obase + tid] = 0ull;
            return;
        }
        // zero-valued pixels in index order, one tile row per ballot.  The first SL rows are wave 0's own pixels (registers);
        // they almost always hold the M - npos zeros still wanted.
        if (start < M && tid < 64) {
            int filled = start;
#pragma unroll
            for (int j = 0; j < SL; ++j)
                if (filled < M) {  // wave-uniform
                    const bool is0 = j < nin && vals[j] == 0.f;
                    const u64 mask = __ballot(is0);
                    const int mypos = __popcll(mask & ((1ull << tid) - 1ull));
                    if (is0 && filled + mypos < M) {
                        cand_key[obase + filled + mypos] = make_key(vals[j], (unsigned)((y0 + j) * src.W + x0 + tid));
                        cand_val[obase + filled + mypos] = vals[j];
                    }
                    filled += __popcll(mask);
                }
            if (tid == 0) { nfilled = filled < M ? filled : M; need_rows = filled < M; }
        } else if (tid == 0) nfilled = start;
        lds_barrier();
        if (need_rows) {  // rare (a tile with fewer than M pixels in its first rows): the other waves' rows through LDS
            if (mpx < TS)
#pragma unroll
                for (int j = 0; j < SL; ++j) nv[mpy0 + j][mpx] = j < nin ? vals[j] : __builtin_nanf("");  // NaN = outside the image
            lds_barrier();
            if (tid < 64) {
                int filled = nfilled;
                for (int py = SL; py < TS && filled < M; ++py) {
                    const float z = tid < TS ? nv[py][tid] : 1.f;
                    const bool is0 = (z == 0.f);
                    const u64 mask = __ballot(is0);
                    const int mypos = __popcll(mask & ((1ull << tid) - 1ull));
                    if (is0 && filled + mypos < M) {
                        cand_key[obase + filled + mypos] = make_key(z, (unsigned)((y0 + py) * src.W + x0 + tid));
                        cand_val[obase + filled + mypos] = z;
                    }
                    filled += __popcll(mask);
                }
                if (tid == 0) nfilled = filled < M ? filled : M;
            }
            lds_barrier();
        }
        start = nfilled;
    }
    NMS_STAMP(6);
    if (start >= M) return;  // the usual case: positives and zeros filled the list
    u64 keys[SL];
#pragma unroll
    for (int j = 0; j < SL; ++j)  // generic rounds: everything on list overflow, else only the negative peaks that are left
        keys[j] = (j < nin && (generic_all || vals[j] < 0.f)) ? key_of(j) : 0ull;
    for (int r = start; r < M; ++r) {
        u64 best = keys[0];
#pragma unroll
        for (int j = 1; j < SL; ++j) best = keys[j] > best ? keys[j] : best;
        const u64 wb = wave_max_u64(best);
        if ((tid & 63) == 0) wbest[r & 1][tid >> 6] = wb;
        lds_barrier();  // one barrier per round: the exchange buffer alternates
        u64 g = wbest[r & 1][0];
#pragma unroll
        for (int w = 1; w < 4; ++w) g = wbest[r & 1][w] > g ? wbest[r & 1][w] : g;
        if (g != 0ull && best == g) {  // keys are unique: exactly one owner
#pragma unroll
            for (int j = 0; j < SL; ++j)
                if (keys[j] == g) { cand_val[obase + r] = vals[j]; keys[j] = 0ull; }
        }
        if (tid == 0) cand_key[obase + r] = g;
    }
}

hipError_t launch_nms_tile_topk(const DecodeSrc &src, int M, u64 *cand_key, float *cand_val, float *cellmax, float skip_thr, hipStream_t s)
{
    const int tiles_x = (src.W + HH_NMS_TILE - 1) / HH_NMS_TILE, tiles_y = (src.H + HH_NMS_TILE - 1) / HH_NMS_TILE;
    hipLaunchKernelGGL(nms_tile_topk_kernel, dim3(tiles_x * tiles_y * src.K * src.B), dim3(256), 0, s, src, M, tiles_x,
                       tiles_x * tiles_y, cand_key, cand_val, cellmax, skip_thr);
    return hipGetLastError();
}

// ------------------------------------------------------------------ the no-group fallback when tiles were skipped
// grouping.py:262-269 takes the best candidate of every joint (top_k's first entry) when no group formed.  With sub-threshold
// tiles skipped that entry may be missing, so for the (rare) flagged images it is recomputed here from the map itself: the
// first entry of torch.topk over hm * (maxpool5x5(hm) == hm) is the global maximum if that is positive (first index among
// equals: the tie rule of this implementation), else the first pixel in index order whose NMS'ed value is zero (any pixel that
// is not a negative peak).
__global__ __launch_bounds__(256) void fallback_top1_kernel(const DecodeSrc src, int M, const int32_t *__restrict__ flags, float *__restrict__ joints)
{
    __shared__ u64 wbest[4];
    const int k = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, D = 3 + src.E;
    if (!(flags[b] & HH_DECODE_FALLBACK)) return;
    u64 best = 0ull;
    for (int y = 0; y < src.H; ++y)
        for (int x = tid; x < src.W; x += 256) {
            const u64 key = make_key(heat_at(src, b, k, y, x), (unsigned)(y * src.W + x));
            best = key > best ? key : best;
        }
    best = wave_max_u64(best);
    if ((tid & 63) == 0) wbest[tid >> 6] = best;
    __syncthreads();
    if (tid != 0) return;
#pragma unroll
    for (int w = 0; w < 4; ++w) best = wbest[w] > best ? wbest[w] : best;
    unsigned idx = 0xffffffffu - (unsigned)(best & 0xffffffffull);
    int x = (int)(idx % (unsigned)src.W), y = (int)(idx / (unsigned)src.W);
    if (!(heat_at(src, b, k, y, x) > 0.f)) {  // no positive value: zeros of the NMS'ed map come first, in index order
        for (unsigned i = 0; i < (unsigned)(src.H * src.W); ++i) {
            const int py = (int)(i / (unsigned)src.W), px = (int)(i % (unsigned)src.W);
            const float c = heat_at(src, b, k, py, px);
            float m = c;
            for (int dy = -2; dy <= 2; ++dy)
                for (int dx = -2; dx <= 2; ++dx) {
                    const int yy = py + dy, xx = px + dx;
                    if (yy >= 0 && yy < src.H && xx >= 0 && xx < src.W) m = fmaxf(m, heat_at(src, b, k, yy, xx));
                }
            if (!(m == c) || c == 0.f) { x = px; y = py; break; }  // not a peak (NMS'ed to 0) or a zero-valued peak
        }
    }
    float *jr = joints + ((size_t)b * M * src.K + k) * D;
    jr[0] = (float)x; jr[1] = (float)y; jr[2] = 0.01f;
    for (int e = 0; e < src.E; ++e) { const float t = tag_at(src, b, k, y, x, e); jr[3 + e] = (t != t) ? 0.f : t; }
}
hipError_t launch_fallback_top1(const DecodeSrc &src, int M, const int32_t *flags, float *joints, hipStream_t s)
{
    hipLaunchKernelGGL(fallback_top1_kernel, dim3(src.K, src.B), dim3(256), 0, s, src, M, flags, joints);
    return hipGetLastError();
}

// ------------------------------------------------------------------ merge -> top_k outputs
// second half of top_k (grouping.py:152-170): global top-M, tag gather, x = idx % w, y = idx / w
__global__ __launch_bounds__(256) void topk_merge_kernel(const DecodeSrc src, int M, int ntiles, u64 *__restrict__ cand_key,
                                                         const float *__restrict__ cand_val, float *__restrict__ tags_k,
                                                         int32_t *__restrict__ coords_k, float *__restrict__ scores_k, int *__restrict__ peaks_ctr)
{
    // the work counters of peaks_region_kernel (the launch in front of this one) go back to zero for the next decode call
    if (peaks_ctr && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < HH_PEAKS_PARTS) peaks_ctr[threadIdx.x] = 0;
    constexpr int NST = 4096;  // candidate keys of a map staged in LDS when they fit (81 tiles x 30 at 512x512)
    __shared__ u64 skeys[NST];
    __shared__ u64 wbest[2][4];
    __shared__ int wpos[2][4];
    __shared__ u64 win_key[HH_MAX_PEOPLE];
    __shared__ int win_pos[HH_MAX_PEOPLE];
    const int k = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int N = ntiles * M;
    u64 *gkeys = cand_key + ((size_t)b * src.K + k) * N;
    const float *vals = cand_val + ((size_t)b * src.K + k) * N;
    const size_t obase = ((size_t)b * src.K + k) * M;
    const bool staged = N <= NST;  // else the rounds scan (and consume) the global list
    __shared__ u64 ckey[256];
    __shared__ int cpos[256];
    __shared__ int ccnt;
    if (tid == 0) ccnt = 0;
    if (staged) {
        for (int i = tid; i < N; i += 256) skeys[i] = gkeys[i];
        __syncthreads();
        // With sub-threshold tiles skipped and the tiles' lists cut at det_thr (nms_tile_topk_kernel) a map holds a handful of
        // non-empty keys among its ntiles * M slots: compact them, rank each against the others (keys are unique), done -- two
        // barriers instead of M rounds of workgroup arg-max.  More than 256 of them: the rounds below.
        for (int i = tid; i < N; i += 256) {
            const u64 kk = skeys[i];
            if (kk != 0ull) {
                const int slot = atomicAdd(&ccnt, 1);
                if (slot < 256) { ckey[slot] = kk; cpos[slot] = i; }
            }
        }
        __syncthreads();
    }
    const int nnz = staged ? ccnt : 257;
    if (nnz <= 256) {
        if (tid < M) { win_key[tid] = 0ull; win_pos[tid] = -1; }
        __syncthreads();
        if (tid < nnz) {
            const u64 me = ckey[tid];
            int rank = 0;
            for (int i = 0; i < nnz; ++i) rank += ckey[i] > me;
            if (rank < M) { win_key[rank] = me; win_pos[rank] = cpos[tid]; }
        }
    } else
    for (int r = 0; r < M; ++r) {  // M rounds of workgroup-wide arg-max; one barrier per round (the exchange buffer alternates)
        u64 best = 0ull;
        int pos = -1;
        if (staged) {
            for (int i = tid; i < N; i += 256) {
                const u64 kk = skeys[i];
                if (kk > best) { best = kk; pos = i; }
            }
        } else {
            for (int i = tid; i < N; i += 256) {
                const u64 kk = gkeys[i];
                if (kk > best) { best = kk; pos = i; }
            }
        }
        const u64 wb = wave_max_u64(best);
        if (best == wb && best != 0ull) { wbest[r & 1][tid >> 6] = wb; wpos[r & 1][tid >> 6] = pos; }  // keys are unique: one owner
        else if ((tid & 63) == 0 && wb == 0ull) { wbest[r & 1][tid >> 6] = 0ull; wpos[r & 1][tid >> 6] = -1; }
        __syncthreads();
        u64 g = 0ull;
        int gp = -1;
#pragma unroll
        for (int w = 0; w < 4; ++w)
            if (wbest[r & 1][w] > g) { g = wbest[r & 1][w]; gp = wpos[r & 1][w]; }
        if (gp >= 0 && pos == gp) {  // the owner retires its candidate before anyone scans again
            if (staged) skeys[gp] = 0ull; else gkeys[gp] = 0ull;
        }
        if (tid == 0) { win_key[r] = g; win_pos[r] = gp; }
        if (!staged) __syncthreads();  // global list: the retirement must be visible to the next scan
    }
    __syncthreads();
    // outputs of the M winners in parallel: score, x = idx % w, y = idx / w, tags gathered at (y, x) (grouping.py:152-170)
    if (tid < M) {
        const u64 g = win_key[tid];
        const int gp = win_pos[tid];
        float sc = 0.f;
        int x = 0, y = 0;
        // (a key always names a pixel of the map; the test keeps a corrupted list from turning into an out-of-bounds gather)
        if (gp >= 0 && 0xffffffffu - (unsigned)(g & 0xffffffffull) < (unsigned)(src.H * src.W)) {
            const unsigned idx = 0xffffffffu - (unsigned)(g & 0xffffffffull);
            // (the peaks pass writes keys only: a positive value's key holds its bits)
            sc = cand_val ? vals[gp] : __uint_as_float((unsigned)(g >> 32) & 0x7fffffffu);
            x = (int)(idx % (unsigned)src.W);
            y = (int)(idx / (unsigned)src.W);
        }
        scores_k[obase + tid] = sc;
        coords_k[(obase + tid) * 2 + 0] = x;
        coords_k[(obase + tid) * 2 + 1] = y;
        for (int e = 0; e < src.E; ++e) tags_k[(obase + tid) * src.E + e] = tag_at(src, b, k, y, x, e);
    }
}

hipError_t launch_topk_merge(const DecodeSrc &src, int M, int ntiles, u64 *cand_key, const float *cand_val, float *tags_k,
                             int32_t *coords_k, float *scores_k, int *peaks_ctr, hipStream_t s)
{
    hipLaunchKernelGGL(topk_merge_kernel, dim3(src.K, src.B), dim3(256), 0, s, src, M, ntiles, cand_key, cand_val, tags_k,
                       coords_k, scores_k, peaks_ctr);
    return hipGetLastError();
}

// ------------------------------------------------------------------ numpy float32 sums
// np.add.reduce over a contiguous float32 vector: pairwise with 8 partial sums for n >= 8
__device__ float np_sum_f32(const float *v, int n, int stride)
{
    if (n < 8) {
        float s = v[0];
        for (int i = 1; i < n; ++i) s = s + v[i * stride];
        return s;
    }
    float r[8];
    for (int q = 0; q < 8; ++q) r[q] = v[q * stride];
    int i = 8;
    for (; i + 8 <= n; i += 8)
        for (int q = 0; q < 8; ++q) r[q] = r[q] + v[(i + q) * stride];
    float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res = res + v[i * stride];
    return res;
}
// np.mean(list of [E] float32 rows, axis=0): E == 1 pairwise, E >= 2 row-sequential
__device__ void np_mean_rows(const float *rows, int n, int E, float *out)
{
    if (E == 1) { out[0] = __fdiv_rn(np_sum_f32(rows, n, 1), (float)n); return; }
    for (int e = 0; e < E; ++e) {
        float s = rows[e];
        for (int i = 1; i < n; ++i) s = s + rows[i * E + e];
        out[e] = __fdiv_rn(s, (float)n);
    }
}

// np.mean of up to 18 float32 values (E == 1: the tag list of one group, K + 1 <= 18 slots): the same additions in the same order
// as np_sum_f32 -- sequential below 8 values, 8 partial sums + pairwise tree + sequential tail from 8 on -- on values that are all
// loaded before the first use (the rolled loop waits for LDS once per value).
__device__ __forceinline__ float np_mean18(const float *rows, int n)
{
    float v[18];
#pragma unroll
    for (int i = 0; i < 18; ++i) v[i] = rows[i < n ? i : 0];
    float res;
    if (n < 8) {
        res = v[0];
#pragma unroll
        for (int i = 1; i < 7; ++i) res = i < n ? res + v[i] : res;
    } else {
        float r[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) r[q] = n >= 16 ? v[q] + v[8 + q] : v[q];
        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        if (n >= 16) {
            res = 16 < n ? res + v[16] : res;
            res = 17 < n ? res + v[17] : res;
        } else {
#pragma unroll
            for (int i = 8; i < 16; ++i) res = i < n ? res + v[i] : res;
        }
    }
    return __fdiv_rn(res, (float)n);
}

// ------------------------------------------------------------------ refine, step (1b): tag bounds
// per (b,k) quarter-res cell: [lo, hi] of the 3x3 tag taps (slack included) -- shared by every person whose missing joint k is
// searched on that map (refine_argmax_kernel).  It needs the input maps only, and the matching kernel below keeps ONE wave per
// image busy for 120-760 us while the rest of the chip idles: since round 3 the bounds are computed by extra 64-thread workgroups
// of the matching launch (virtual block vb of the bounds = blockIdx.x - B), one launch and ~25 us of the decode's critical path
// less.  Virtual block 0 also clears the 8 queue counters of the arg-max pass (filled by adjust_scores_kernel, behind this launch).
// thread = one quarter-res column x TBR consecutive rows: the 3-wide row minima / maxima are made once per source row and
// slide down the column (3.75 loads per cell instead of 9; clamped border rows / columns repeat a tap, which min / max ignore)
constexpr int TBR = 8;
// Round 4: also the tag hull of every 8x8-cell SUPER (suptag, for refine_bb_kernel): a thread's TBR = 8 rows are one super's rows, and
// eight neighbouring lanes hold its eight columns (the columns are padded to a multiple of 8 per row block, lanes past the map idle).
__host__ __device__ __forceinline__ long long tag_bounds_units(int B, int K, int hq, int wq)
{
    return (long long)B * K * ((hq + TBR - 1) / TBR) * ((wq + 7) & ~7);
}
__device__ __forceinline__ void tag_bounds_part(const DecodeSrc &src, float *__restrict__ tagb, unsigned *__restrict__ suptag, int32_t *__restrict__ ws_jobs,
                                                int vb, int lane)
{
    if (vb == 0 && lane < 8) ws_jobs[lane] = 0;
    const int hq = src.H >> 2, wq = src.W >> 2, E = src.E, wq8 = (wq + 7) & ~7;
    const int nrb = (hq + TBR - 1) / TBR, per_map = nrb * wq8;
    const long long L = (long long)vb * 64 + lane;
    if (L >= (long long)src.B * src.K * per_map) return;  // (whole groups of 8 lanes: per_map is a multiple of 8)
    const int map = (int)(L / per_map), it = (int)(L % per_map);
    const int k = map % src.K, b = map / src.K;
    {
        const int qx = it % wq8, qy0 = (it / wq8) * TBR;
        const bool live = qx < wq;
        const int qc = min(qx, wq - 1), xa = max(qc - 1, 0), xb = min(qc + 1, wq - 1);
        for (int e = 0; e < E; ++e) {
            const float *tq = src.tags_q[e] + (size_t)b * src.tags_bs[e] + (size_t)k * hq * wq;
            // Round 4: the hull of the cell's 16 PIXEL tags themselves (the x4 bilinear of the 3x3 taps, in tag_at()'s / the refine
            // evaluation's own expressions: three tap rows along x at the four sub-columns, then along y), not of the nine taps: a
            // pixel weighs its outer taps with at most 0.375, so the hull is much narrower where the tags change (blob rims), and
            // the refine scans open and evaluate correspondingly fewer cells.  The work rides beside the matcher on an idle chip.
            constexpr int TO4[4] = {0, 0, 1, 1};
            constexpr float TW4[4] = {0.625f, 0.875f, 0.125f, 0.375f};
            float hx[TBR + 2][4];
#pragma unroll
            for (int r = 0; r < TBR + 2; ++r) {
                const float *row = tq + (size_t)min(max(qy0 - 1 + r, 0), hq - 1) * wq;
                const float t[3] = {row[xa], row[qc], row[xb]};
#pragma unroll
                for (int jx = 0; jx < 4; ++jx) {
                    const bool first = qc == 0 && jx < 2;  // source position below 0: torch reads sample 0 with weight 1
                    const float w0 = first ? 0.f : 1.f - TW4[jx], w1 = first ? 1.f : TW4[jx];
                    hx[r][jx] = __builtin_fmaf(t[TO4[jx]], w0, t[TO4[jx] + 1] * w1);
                }
            }
            float slo = INFINITY, shi = -INFINITY;  // the hull of this thread's cells, as stored
#pragma unroll
            for (int r = 0; r < TBR; ++r) {
                if (qy0 + r >= hq) break;
                float lo = INFINITY, hi = -INFINITY;
#pragma unroll
                for (int jy = 0; jy < 4; ++jy) {
                    const bool first = qy0 + r == 0 && jy < 2;
                    const float w0 = first ? 0.f : 1.f - TW4[jy], w1 = first ? 1.f : TW4[jy];
#pragma unroll
                    for (int jx = 0; jx < 4; ++jx) {
                        const float tg = __builtin_fmaf(hx[r + TO4[jy]][jx], w0, hx[r + TO4[jy] + 1][jx] * w1);
                        lo = fminf(lo, tg); hi = fmaxf(hi, tg);
                    }
                }
                const float slack = 1e-6f * fmaxf(fabsf(lo), fabsf(hi)) + 1e-30f;  // (the values are the evaluation's own; the slack is belt and braces)
                // stored as a bf16 pair in one dword (lo rounded DOWN, hi rounded UP: the bound only gets looser), which
                // halves what the arg-max scans have to read per cell
                const unsigned lh = (unsigned)bf16_floor(lo - slack) | ((unsigned)bf16_ceil(hi + slack) << 16);
                if (live) reinterpret_cast<unsigned *>(tagb)[(((size_t)b * src.K + k) * hq * wq + (size_t)(qy0 + r) * wq + qx) * E + e] = lh;
                slo = fminf(slo, __uint_as_float(lh << 16)); shi = fmaxf(shi, __uint_as_float(lh & 0xffff0000u));
            }
            if (!live) { slo = INFINITY; shi = -INFINITY; }
#pragma unroll
            for (int off = 1; off < 8; off <<= 1) { slo = fminf(slo, __shfl_xor(slo, off)); shi = fmaxf(shi, __shfl_xor(shi, off)); }
            if (suptag && (qx & 7) == 0)  // (bf16 values: the halves are exact)
                suptag[(((size_t)b * src.K + k) * nrb + qy0 / TBR) * ((wq + 7) >> 3) * E + (size_t)(qx >> 3) * E + e] =
                    (__float_as_uint(slo) >> 16) | (__float_as_uint(shi) & 0xffff0000u);
        }
    }
}

// ------------------------------------------------------------------ match_by_tag
// grouping.py:85-145 with munkres 1.1.4 (munkres.py:114-340) restated wave-parallel: one wave per image.  The cost matrix
// (float64) is built in LDS, then column j lives in the registers of lane j; the zero pattern of row i is a 64-bit mask on
// lane i, covers are two wave-uniform masks and the star / prime of row i are two ints on lane i, so the search steps
// (munkres steps 2-5) are bit operations and lane reads - no memory, no barrier.  Every decision (which uncovered zero is
// taken first, the path of step 5) is the one munkres.py takes.
#define MLD 33
struct MatchShared {
    double Cm[HH_MAX_PEOPLE * MLD];
    double saved[HH_MAX_PEOPLE * MLD];
    double cj[HH_MAX_PEOPLE * 3];
    float ctag[HH_MAX_PEOPLE * HH_MAX_EMB];
    float gmean[HH_MAX_PEOPLE * HH_MAX_EMB];
    float gkey[HH_MAX_PEOPLE];
    int gnt[HH_MAX_PEOPLE];
    int assign[HH_MAX_PEOPLE];
    int G;
};

// Row / column sets of the assignment problem are 32-bit masks (n <= HH_MAX_PEOPLE = 32): one SGPR, one v_readlane / v_writelane.
static_assert(HH_MAX_PEOPLE <= 32, "the matcher's zero / cover masks are 32-bit");
__device__ __forceinline__ unsigned readlane_u32(unsigned v, int l) { return (unsigned)__builtin_amdgcn_readlane((int)v, l); }
// `old` with lane L replaced by the (wave-uniform) val.  L a compile-time constant: ONE v_writelane_b32 (inline asm: this compiler has
// no builtin for it); a run-time lane would need m0 as the lane select (the VOP3 form takes one SGPR), a register the compiler
// reserves, so those sites keep the compare + select.
template <int L>
__device__ __forceinline__ int writelane_const(int val, int old)
{
    // (s_nop 1: on gfx940+ a VALU that reads an SGPR a VALU has just written -- the ballot -- needs two wait states, and the compiler
    // does not look into an asm statement to count them: without it the lane got the PREVIOUS ballot now and then)
    asm("s_nop 1\n\tv_writelane_b32 %0, %1, %2" : "+v"(old) : "s"(val), "n"(L));
    return old;
}
__device__ __forceinline__ int setlane_i32(int val, int l, int old, int lane) { return lane == l ? val : old; }
template <typename F, int... I>
__device__ __forceinline__ void mk_for_impl(F &&f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void mk_for(F &&f) { mk_for_impl(f, std::make_integer_sequence<int, N>{}); }
// v of the lane ROT places further up its 16-lane row (row_ror): a vector-pipe move -- the xor shuffles they replace went through
// the LDS crossbar, twice per double
template <int ROT>
__device__ __forceinline__ double row_ror_f64(double v)
{
    const u64 b = __builtin_bit_cast(u64, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)b, 0x120 + ROT, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), 0x120 + ROT, 0xf, 0xf, false);
    return __builtin_bit_cast(double, ((u64)(unsigned)hi << 32) | (u64)(unsigned)lo);
}
template <int ROT>
__device__ __forceinline__ unsigned row_ror_u32(unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x120 + ROT, 0xf, 0xf, false); }
__device__ __forceinline__ double readlane_f64(double v, int l)
{
    const u64 b = __builtin_bit_cast(u64, v);
    const unsigned lo = readlane_u32((unsigned)b, l), hi = readlane_u32((unsigned)(b >> 32), l);
    return __builtin_bit_cast(double, ((u64)hi << 32) | lo);
}

// munkres on the n x n matrix in S.Cm.  -> 0 and `star` = the starred column of row `lane` (-1: none), or 1 if the iteration guard
// ran out.  NMAX >= n: the length of the unrolled loops (row `lane`, then column `lane`, live in NMAX registers).
// Round 3, second pass over the step machine (same decisions, fewer instructions on its one wave, whose every instruction's latency
// is exposed): 32-bit masks; a lane's value is set with v_writelane instead of compare + select; step 4 keeps the set of rows
// that hold an uncovered zero up to date from the per-COLUMN zero masks (`zt`: uncovering column c adds the rows with a zero in
// it) instead of recomputing it from every row's mask in every iteration; step 6's minimum runs on row rotations (DPP).
template <int NMAX>
__device__ int munkres_wave_n(MatchShared &S, int n, int lane, int &star)
{
    // step 1: subtract the row minimum (lane = row).  The row is read into registers in one go (a rolled loop waits for LDS once
    // per element), reduced there and written back.
    const unsigned nmask = n >= 32 ? 0xffffffffu : (1u << n) - 1u;  // lanes / rows / columns of the problem
    {
        double rowv[NMAX];
        unsigned rz = 0;  // the zeros step 1 leaves in row `lane`: the columns that hold the row minimum
        if (lane < n) {
#pragma unroll
            for (int j = 0; j < NMAX; ++j) rowv[j] = S.Cm[lane * MLD + (j < n ? j : 0)];
            double mn = rowv[0];
#pragma unroll
            for (int j = 1; j < NMAX; ++j) if (j < n && rowv[j] < mn) mn = rowv[j];
#pragma unroll
            for (int j = 0; j < NMAX; ++j) {
                rowv[j] = rowv[j] - mn;
                rz |= (j < n && rowv[j] == 0.0) ? (1u << j) : 0u;
            }
        }
        // Round 4: steps 2 and 3 straight from these masks.  If step 2 stars every row, step 3 finds every column covered and the
        // solver is done with exactly these stars -- the usual case on separable tags -- without the matrix write-back, the column
        // fetch and the per-row ballots that rebuild the same masks below (a third of a call's prologue).
        unsigned cc0 = 0;
        int sc0 = -1;
        for (int i = 0; i < n; ++i) {
            const unsigned z = readlane_u32(rz, i) & ~cc0;
            if (z) {
                const int j = __builtin_ctz(z);
                sc0 = setlane_i32(j, i, sc0, lane);
                cc0 |= 1u << j;
            }
        }
        if (__popc(cc0 & nmask) >= n) { star = sc0; return 0; }
        if (lane < n)
#pragma unroll
            for (int j = 0; j < NMAX; ++j) if (j < n) S.Cm[lane * MLD + j] = rowv[j];
    }
    __syncthreads();
    double col[NMAX];  // column `lane`
    int myz = 0;       // zeros of row `lane` (bit j = column j)
    unsigned zt = 0;   // zeros of column `lane` (bit i = row i)
    {
        const int cl = lane < n ? lane : 0;
#pragma unroll
        for (int i = 0; i < NMAX; ++i) {  // every read in flight before the first ballot needs one
            const double v = S.Cm[(i < n ? i : 0) * MLD + cl];
            col[i] = (i < n && lane < n) ? v : 1.0;
        }
        mk_for<NMAX>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            if (i < n) {
                // (lanes >= n hold 1.0 and are masked out of the ballot; their own `zt` is never read)
                const bool z = col[i] == 0.0;
                myz = writelane_const<i>((int)((unsigned)__ballot(z) & nmask), myz);
                zt |= z ? (1u << i) : 0u;
            }
        });
    }
    int starcol = -1, primecol = -1;  // of row `lane`
    unsigned ccm = 0, rcm = 0;        // covered columns / rows
    // step 2: star the first zero of each row whose column has no star yet
    for (int i = 0; i < n; ++i) {
        const unsigned z = readlane_u32((unsigned)myz, i) & ~ccm;
        if (z) {
            const int j = __builtin_ctz(z);
            starcol = setlane_i32(j, i, starcol, lane);
            ccm |= 1u << j;
        }
    }
    ccm = 0;
    int step = 3, z0r = 0, z0c = 0;
    for (int guard = 0; guard < 200000; ++guard) {
        if (step == 3) {  // cover every column that holds a star
            // (an OR over the lanes' star bits -- four row rotations and the two rows of lanes 0 .. 31 -- instead of one lane read per
            // starred row: this step runs once per augmentation, ~50 times per dense problem)
            unsigned x = starcol >= 0 ? (1u << starcol) : 0u;
            x |= row_ror_u32<1>(x); x |= row_ror_u32<2>(x); x |= row_ror_u32<4>(x); x |= row_ror_u32<8>(x);
            ccm |= readlane_u32(x, 0) | readlane_u32(x, 16);
            if (__popc(ccm) >= n) { star = starcol; return 0; }
            step = 4;
        } else if (step == 4) {
            int row = 0, colc = 0;
            // rows that hold an uncovered zero (their own cover aside): inside this step columns only get UNcovered, so the set
            // only grows -- by the rows with a zero in the column that was uncovered
            unsigned nz = (unsigned)__ballot(lane < n && ((unsigned)myz & ~ccm) != 0u);
            // (the column cover as a scalar inside the loop: the compiler keeps `ccm` in a vector register -- steps 4's ballot and 6
            // use it per lane -- and would run the bit scans below through the vector pipe)
            unsigned cs = (unsigned)__builtin_amdgcn_readfirstlane((int)ccm);
            for (;;) {
                // first uncovered row (cyclic from `row`) with an uncovered zero; in it the last uncovered zero in cyclic
                // column order starting at `colc`
                const unsigned rows = nz & ~rcm;
                if (!rows) { step = 6; break; }
                const unsigned hi = rows & ~((1u << row) - 1u);
                const int fr = __builtin_ctz(hi ? hi : rows);
                const unsigned unc = readlane_u32((unsigned)myz, fr) & ~cs;
                const unsigned low = unc & ((1u << colc) - 1u);
                const int fc = 31 - __builtin_clz(low ? low : unc);
                primecol = setlane_i32(fc, fr, primecol, lane);
                const int sc = __builtin_amdgcn_readlane(starcol, fr);
                if (sc >= 0) {
                    rcm |= 1u << fr;
                    cs &= ~(1u << sc);
                    nz |= readlane_u32(zt, sc);
                    row = fr; colc = sc;
                } else {
                    z0r = fr; z0c = fc; step = 5;
                    break;
                }
            }
            ccm = cs;
        } else if (step == 5) {  // alternate primes and stars from the uncovered prime: each row on the path takes its prime
            const int oldstar = starcol;
            int r = z0r, c = z0c;
            for (int hop = 0; hop <= HH_MAX_PEOPLE; ++hop) {
                const unsigned sm = (unsigned)__ballot(oldstar == c);  // the star of column c before the flips
                starcol = setlane_i32(c, r, starcol, lane);
                if (!sm) break;
                r = __builtin_ctz(sm);
                c = __builtin_amdgcn_readlane(primecol, r);
            }
            rcm = 0; ccm = 0; primecol = -1;
            step = 3;
        } else {  // step 6: smallest uncovered value; add it to covered rows, subtract it from uncovered columns
            double mn = 9223372036854775807.0;
            const bool colunc = !((ccm >> (lane & 31)) & 1u);
            const bool cu = lane < n && colunc;
#pragma unroll
            for (int i = 0; i < NMAX; ++i)
                if (i < n && cu && !((rcm >> i) & 1u) && mn > col[i]) mn = col[i];
            {   // minimum over lanes 0 .. 31 (the others hold the start value): four rotations inside the 16-lane rows, then the two rows
                double o;
                o = row_ror_f64<1>(mn); mn = o < mn ? o : mn;
                o = row_ror_f64<2>(mn); mn = o < mn ? o : mn;
                o = row_ror_f64<4>(mn); mn = o < mn ? o : mn;
                o = row_ror_f64<8>(mn); mn = o < mn ? o : mn;
                const double m0 = readlane_f64(mn, 0), m1 = readlane_f64(mn, 16);
                mn = m1 < m0 ? m1 : m0;
            }
            unsigned ztn = 0;
            // munkres.py:303-309 adds the minimum to covered rows and then subtracts it from uncovered columns: two roundings in that
            // order where both apply.  Written as c + a_i - sub with a_i = mn or 0 (wave-uniform: the row's cover) and sub = mn or 0
            // (the lane's column): adding or subtracting 0.0 returns c itself (up to the sign of a zero, which nothing here looks at),
            // and the selects act on one scalar and one register instead of on every element.
            const double sub = colunc ? mn : 0.0;
            mk_for<NMAX>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                if (i < n) {
                    const double a_i = ((rcm >> i) & 1u) ? mn : 0.0;
                    const double c = (col[i] + a_i) - sub;
                    col[i] = c;
                    const bool z = c == 0.0;
                    myz = writelane_const<i>((int)((unsigned)__ballot(z) & nmask), myz);
                    ztn |= z ? (1u << i) : 0u;
                }
            });
            zt = ztn;
            step = 4;
        }
    }
    return 1;
}

// most images hold far fewer than HH_MAX_PEOPLE candidates per joint: the short instantiations skip the masked-off iterations
__device__ __forceinline__ int munkres_wave(MatchShared &S, int n, int lane, int &star)
{
    if (n <= 8) return munkres_wave_n<8>(S, n, lane, star);
    if (n <= 16) return munkres_wave_n<16>(S, n, lane, star);
    if (n <= 24) return munkres_wave_n<24>(S, n, lane, star);
    return munkres_wave_n<HH_MAX_PEOPLE>(S, n, lane, star);
}

// the solver alone on one square float64 matrix (tests: the reference's pinned munkres goldens through the GPU step machine)
__global__ __launch_bounds__(64) void munkres_debug_kernel(const double *__restrict__ cost, int n, int32_t *__restrict__ out)
{
    __shared__ MatchShared S;
    const int lane = threadIdx.x;
    for (int i = lane; i < n * n; i += 64) S.Cm[(i / n) * MLD + i % n] = cost[i];
    __syncthreads();
    int star = -1;
    const int bad = munkres_wave(S, n, lane, star);
    if (lane < n) out[lane] = star;
    if (lane == 0) out[n] = bad;
}
hipError_t launch_munkres_debug(const double *cost, int n, int32_t *out, hipStream_t s)
{
    hipLaunchKernelGGL(munkres_debug_kernel, dim3(1), dim3(64), 0, s, cost, n, out);
    return hipGetLastError();
}

#ifdef HH_MATCH_STAMP  // diagnostic build (tools/probes/match_stamps.sh): cycle stamps of image 0's wave, 8 per joint
__device__ unsigned long long g_match_stamps[32 * 8];
#define MSTAMP(i) do { if (b == 0 && lane == 0) g_match_stamps[it * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
extern "C" int hh_debug_match_stamps(unsigned long long *out)
{
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_match_stamps), sizeof(g_match_stamps));
}
#else
#define MSTAMP(i)
#endif
__constant__ int c_joints_order[17] = {0, 1, 2, 3, 4, 5, 6, 11, 12, 7, 8, 9, 10, 13, 14, 15, 16};  // grouping.py:63-65

__global__ __launch_bounds__(64) void match_kernel(const float *tags_k, const int32_t *coords_k, const float *scores_k, int K, int M, int E,
                                                   double det_thr,
                                                   double tag_thr, float *__restrict__ joints, int32_t *__restrict__ num_people,
                                                   float *__restrict__ ws_tags, int32_t *__restrict__ flags, int stage, int nimg,
                                                   const DecodeSrc src, float *__restrict__ tagb, unsigned *__restrict__ suptag,
                                                   int32_t *__restrict__ ws_jobs)
{
    __shared__ MatchShared S;
    extern __shared__ float staged[];  // the image's candidates and group tag lists, when they fit (stage != 0)
    if ((int)blockIdx.x >= nimg) {  // the workgroups behind the images': tag bounds for the refine scans (tag_bounds_part)
        tag_bounds_part(src, tagb, suptag, ws_jobs, (int)blockIdx.x - nimg, (int)threadIdx.x);
        return;
    }
    const int b = blockIdx.x, lane = threadIdx.x, D = 3 + E;
    float *J = joints + (size_t)b * M * K * D;
    float *GT = ws_tags + (size_t)b * M * (K + 1) * E;  // per group: list of member tags
    tags_k += (size_t)b * K * M * E; coords_k += (size_t)b * K * M * 2; scores_k += (size_t)b * K * M;
    if (stage) {  // one coalesced read instead of a dependent global round trip per joint
        float *st = staged, *ss = st + K * M * E;
        int32_t *sc = reinterpret_cast<int32_t *>(ss + K * M);
        for (int i = lane; i < K * M * E; i += 64) st[i] = tags_k[i];
        for (int i = lane; i < K * M; i += 64) ss[i] = scores_k[i];
        for (int i = lane; i < K * M * 2; i += 64) sc[i] = coords_k[i];
        tags_k = st; scores_k = ss; coords_k = sc;
        GT = reinterpret_cast<float *>(sc + K * M * 2);
    }
    for (int i = lane; i < M * K * D; i += 64) J[i] = 0.f;
    if (lane == 0) S.G = 0;
    __syncthreads();
    int bad = 0;
    for (int it = 0; it < K; ++it) {
        const int idx = K == 17 ? c_joints_order[it] : it;
        MSTAMP(0);
        // candidates with score > det_thr, order kept (grouping.py:98-102)
        float s = 0.f;
        bool keep = false;
        if (lane < M) { s = scores_k[idx * M + lane]; keep = (double)s > det_thr; }
        const u64 kmask = __ballot(keep);
        const int na = __popcll(kmask);
        if (keep) {
            const int a = __popcll(kmask & ((1ull << lane) - 1ull));
            S.cj[a * 3 + 0] = (double)coords_k[(idx * M + lane) * 2 + 0];
            S.cj[a * 3 + 1] = (double)coords_k[(idx * M + lane) * 2 + 1];
            S.cj[a * 3 + 2] = (double)s;
            for (int e = 0; e < E; ++e) S.ctag[a * HH_MAX_EMB + e] = tags_k[(idx * M + lane) * E + e];
        }
        // the groups' mean tags (grouping.py:107-108) in the same stretch as the candidate gather: they only need what the previous joint
        // left (behind its closing barrier), and their LDS round trips overlap the gather's instead of following them behind a barrier
        const int G = S.G;
        const bool first = (it == 0) || (G == 0);
        int ng = 0;
        if (!first) {
            ng = G < M ? G : M;
            if (lane < ng) {
                const int nt = S.gnt[lane];
                if (E == 1 && nt <= 18) S.gmean[lane * HH_MAX_EMB] = np_mean18(GT + (size_t)lane * (K + 1), nt);
                else np_mean_rows(GT + (size_t)lane * (K + 1) * E, nt, E, &S.gmean[lane * HH_MAX_EMB]);
            }
        }
        __syncthreads();
        MSTAMP(1);
        if (na == 0) continue;
        if (!first) {
            MSTAMP(2);
            const int n = na > ng ? na : ng;
            const float inv_n = 1.0f / (float)n;
            for (int i = lane; i < n * n; i += 64) {
                // (a, g) = (i / n, i % n) without an integer division: (i + 0.5) / n is at least 1 / (2n) >= 1/64 away from an integer,
                // far beyond the float error of the product for i < 1024
                const int a = (int)(((float)i + 0.5f) * inv_n), g = i - a * n;
                double c = 0.0;  // munkres pad_matrix rows
                if (a < na) {
                    if (g < ng) {
                        double ss = 0.0;
                        for (int e = 0; e < E; ++e) {
                            const double d = (double)S.ctag[a * HH_MAX_EMB + e] - (double)S.gmean[g * HH_MAX_EMB + e];
                            ss = ss + d * d;
                        }
                        const double dist = __dsqrt_rn(ss);
                        S.saved[a * MLD + g] = dist;
                        c = rint(dist) * 100.0 - S.cj[a * 3 + 2];
                    } else {
                        c = 1e10;  // grouping.py:126-128
                    }
                }
                S.Cm[a * MLD + g] = c;
            }
            __syncthreads();
            MSTAMP(3);
            int star = -1;
            bad |= munkres_wave(S, n, lane, star);
            if (lane < na) S.assign[lane] = star;
            __syncthreads();
            MSTAMP(4);
        }
        // dict semantics of grouping.py:104-143.  A candidate matched to a group appends to that group only (the assignment is
        // one-to-one), so the matched ones are applied by their own lanes; what must stay in candidate order on lane 0 are the
        // unmatched ones (new keys, max_num_people cut-off).  The one interaction - an unmatched candidate whose key equals the
        // key of an EXISTING group resets that group's list, before or after a matched append depending on the order - sends
        // the whole joint down the serial path.
        bool matched = false;
        int mcol = -1;
        if (!first && lane < na) {
            mcol = S.assign[lane];
            matched = mcol >= 0 && mcol < ng && S.saved[lane * MLD + mcol] < tag_thr;
        }
        const u64 allmask = (1ull << na) - 1ull;  // na <= M <= 32
        const u64 mmask = __ballot(matched);
        u64 serial = allmask;
        if (mmask) {
            const u64 umask = allmask & ~mmask;
            bool collide = false;
            if (umask) {
                const float gk = lane < S.G ? S.gkey[lane] : __builtin_nanf("");
                const int myk = lane < na ? __float_as_int(S.ctag[lane * HH_MAX_EMB]) : 0;
                for (u64 um = umask; um; um &= um - 1) {
                    const float key = __int_as_float(__builtin_amdgcn_readlane(myk, __builtin_ctzll(um)));
                    collide |= __ballot(gk == key) != 0ull;
                }
            }
            if (!collide) {
                if (matched) {
                    const int t = mcol, pos = S.gnt[t];
                    float *jr = J + ((size_t)t * K + idx) * D;
                    jr[0] = (float)S.cj[lane * 3 + 0]; jr[1] = (float)S.cj[lane * 3 + 1]; jr[2] = (float)S.cj[lane * 3 + 2];
                    for (int e = 0; e < E; ++e) {
                        jr[3 + e] = S.ctag[lane * HH_MAX_EMB + e];
                        GT[((size_t)t * (K + 1) + pos) * E + e] = S.ctag[lane * HH_MAX_EMB + e];
                    }
                    S.gnt[t] = pos + 1;
                }
                serial = umask;
                __syncthreads();
            }
        }
        MSTAMP(5);
        {   // candidates in row order; the loop is wave-uniform: the search for an existing key is one ballot over the groups' keys
            // (lane q holds the key of group q) instead of a scan on one lane, lane 0 does the writes
            int Gc = S.G;
            float gkreg = lane < Gc ? S.gkey[lane] : __builtin_nanf("");
            for (u64 sm = serial; sm; sm &= sm - 1) {
                const int a = __builtin_ctzll(sm);
                int t;
                bool fresh = false;  // a new (or reset) tag list: its length is known to be 0, no LDS round trips to find out
                const int col = first ? -1 : S.assign[a];
                if (!first && col >= 0 && col < ng && S.saved[a * MLD + col] < tag_thr) t = col;
                else {
                    const float key = S.ctag[a * HH_MAX_EMB];
                    const u64 hit = __ballot(lane < Gc && gkreg == key);
                    if (hit) t = __builtin_ctzll(hit);  // (the first group with that key, as the scan in key order found it)
                    else {
                        if (Gc >= M) continue;  // groups past max_num_people are never matched nor returned
                        t = Gc++;
                        if (lane == t) gkreg = key;
                        if (lane == 0) S.gkey[t] = key;
                    }
                    fresh = true;
                }
                if (lane == 0) {
                    const int pos = fresh ? 0 : S.gnt[t];
                    float *jr = J + ((size_t)t * K + idx) * D;
                    jr[0] = (float)S.cj[a * 3 + 0]; jr[1] = (float)S.cj[a * 3 + 1]; jr[2] = (float)S.cj[a * 3 + 2];
                    for (int e = 0; e < E; ++e) {
                        jr[3 + e] = S.ctag[a * HH_MAX_EMB + e];
                        GT[((size_t)t * (K + 1) + pos) * E + e] = S.ctag[a * HH_MAX_EMB + e];
                    }
                    S.gnt[t] = pos + 1;
                }
            }
            if (lane == 0) S.G = Gc;
        }
        __syncthreads();
        MSTAMP(6);
    }
    if (lane == 0) {
        int P = S.G;
        if (P == 0) {  // grouping.py:262-269: no group -> best candidate per joint, score 0.01
            for (int k = 0; k < K; ++k) {
                float *jr = J + (size_t)k * D;
                jr[0] = (float)coords_k[(k * M) * 2 + 0];
                jr[1] = (float)coords_k[(k * M) * 2 + 1];
                jr[2] = 0.01f;
                for (int e = 0; e < E; ++e) { const float t = tags_k[(k * M) * E + e]; jr[3 + e] = (t != t) ? 0.f : t; }
            }
            P = 1;
            bad |= HH_DECODE_FALLBACK << 8;
        }
        num_people[b] = P;
        flags[b] = ((bad >> 8) & HH_DECODE_FALLBACK) | ((bad & 0xff) ? HH_DECODE_SOLVER_GUARD : 0);
    }
}

hipError_t launch_match(const float *tags_k, const int32_t *coords_k, const float *scores_k, int B, int K, int M, int E,
                        double det_thr, double tag_thr, float *joints, int32_t *num_people, float *ws_tags, int32_t *flags,
                        const DecodeSrc *bounds_src, float *tagb, unsigned *suptag, int32_t *ws_jobs, hipStream_t s)
{
    const size_t bytes = ((size_t)K * M * (E + 3) + (size_t)M * (K + 1) * E) * 4;  // candidates + group tag lists
    const int stage = bytes <= 40 * 1024;
    int extra = 0;
    DecodeSrc src{};
    if (bounds_src) {  // + the tag bounds of the refine scans (and the cleared queue counters) in the same launch
        src = *bounds_src;
        extra = (int)((tag_bounds_units(src.B, src.K, src.H >> 2, src.W >> 2) + 63) / 64);
    }
    hipLaunchKernelGGL(match_kernel, dim3(B + extra), dim3(64), stage ? bytes : 0, s, tags_k, coords_k, scores_k, K, M, E, det_thr, tag_thr,
                       joints, num_people, ws_tags, flags, stage, B, src, tagb, suptag, ws_jobs);
    return hipGetLastError();
}

// ------------------------------------------------------------------ adjust + person scores
// grouping.py:172-191 and :276 (scores = joints[..., 2].mean(1), taken BEFORE refine)
__global__ __launch_bounds__(256) void adjust_scores_kernel(const DecodeSrc src, int M, int adjust, int refine, float *__restrict__ joints,
                                                            const int32_t *__restrict__ num_people, float *__restrict__ scores,
                                                            float *__restrict__ ws_prev, int32_t *__restrict__ ws_jobs)
{
    // + the first step of refine (grouping.py:200-214: the mean tag of a person's detected joints, read at the adjusted
    // coordinates' pixel) and the work lists of the arg-max pass: the tag of every (person, joint) is sampled by the thread that
    // adjusts it -- P * K independent reads instead of one thread per person walking its joints one dependent read after the other
    __shared__ float tl[HH_MAX_PEOPLE * 64 * HH_MAX_EMB];  // [p][k][e]; K <= 64
    __shared__ float sl[HH_MAX_PEOPLE * 64];               // [p][k]: the joints' scores (the per-person loops below walk them here:
                                                           // from global memory each step was a dependent round trip, 2 K of them)
    const int b = blockIdx.x, tid = threadIdx.x, K = src.K, E = src.E, D = 3 + E;
    const int P = min(num_people[b], M);
    float *J = joints + (size_t)b * M * K * D;
    for (int i = tid; i < P * K; i += 256) {
        float *j = J + (size_t)i * D;
        const int k = i % K;
        sl[i] = j[2];
        if (j[2] == 0.f) continue;
        float x = j[0], y = j[1];
        const int xi = (int)x, yi = (int)y;
        // the four heat samples and the tag sample(s) are independent reads -- the tag is read at the pixel of the ADJUSTED coordinates
        // (grouping.py:206-207), which is (yi, xi) again: the candidate's integer coordinate +- 0.25 + 0.5 truncates back to it --,
        // so all of them are in flight before the first comparison (each heat sample is twenty loads behind index arithmetic)
        float hs[4] = {0.f, 0.f, 0.f, 0.f}, tg[HH_MAX_EMB] = {0.f, 0.f, 0.f, 0.f};
        const bool want_tag = refine && j[2] > 0.f;
        // (one embedding, the usual case: the four taps are only LOADED here -- bilerp()'s arithmetic follows behind the heat samples'
        // loads, so the two groups share one memory round trip; more embeddings take tag_at() one after the other)
        const bool tag1 = refine && E == 1 && src.mode == 0;
        float tt[4] = {0.f, 0.f, 0.f, 0.f};
        Lin tly{}, tlx{};
        if (tag1) {
            const int hq = src.H >> 2, wq = src.W >> 2;
            const float *tq = src.tags_q[0] + (size_t)b * src.tags_bs[0] + (size_t)k * hq * wq;
            tly = src_index(hq, src.scale_h4, yi); tlx = src_index(wq, src.scale_w4, xi);
            tt[0] = ldg(tq, tly.i0 * wq + tlx.i0); tt[1] = ldg(tq, tly.i0 * wq + tlx.i1);
            tt[2] = ldg(tq, tly.i1 * wq + tlx.i0); tt[3] = ldg(tq, tly.i1 * wq + tlx.i1);
        } else if (refine) {
#pragma unroll
            for (int e = 0; e < HH_MAX_EMB; ++e)
                if (e < E) tg[e] = tag_at(src, b, k, yi, xi, e);
        }
        if (adjust) {
            const int xr = min(xi + 1, src.W - 1), xl = max(xi - 1, 0), yd = min(yi + 1, src.H - 1), yu = max(yi - 1, 0);
            if (src.mode == 0 && !src.avg) {
                const int ys[4] = {yi, yi, yd, yu}, xs[4] = {xr, xl, xi, xi};
                heat_otf<4>(src, b, k, ys, xs, hs);
            } else {
                hs[0] = heat_at(src, b, k, yi, xr); hs[1] = heat_at(src, b, k, yi, xl);
                hs[2] = heat_at(src, b, k, yd, xi); hs[3] = heat_at(src, b, k, yu, xi);
            }
        }
        if (adjust) {
            if (hs[0] > hs[1]) x += 0.25f; else x -= 0.25f;
            if (hs[2] > hs[3]) y += 0.25f; else y -= 0.25f;
            x += 0.5f; y += 0.5f;
            j[0] = x; j[1] = y;
        }
        if (tag1) {  // bilerp(): rows along x, then along y
            const float a = __builtin_fmaf(tt[0], tlx.w0, tt[1] * tlx.w1), c = __builtin_fmaf(tt[2], tlx.w0, tt[3] * tlx.w1);
            tg[0] = __builtin_fmaf(a, tly.w0, c * tly.w1);
        }
        if (want_tag)
#pragma unroll
            for (int e = 0; e < HH_MAX_EMB; ++e)
                if (e < E) tl[i * E + e] = tg[e];
    }
    __syncthreads();
    for (int p = tid; p < M; p += 256)
        scores[(size_t)b * M + p] = p < P ? __fdiv_rn(np_sum_f32(sl + (size_t)p * K, K, 1), (float)K) : 0.f;
    if (!refine) return;
    __shared__ int qcnt[8], qbase[8];
    __shared__ unsigned char pok[HH_MAX_PEOPLE];  // the person has a mean tag (at least one detected joint)
    if (tid < 8) qcnt[tid] = 0;
    if (tid < P) {
        const int p = tid;
        const float *Sp = sl + (size_t)p * K;
        float *mine = tl + (size_t)p * K * E;  // compacted in place, in joint order (the write index never passes the read index)
        int nt = 0;
        for (int k = 0; k < K; ++k)
            if (Sp[k] > 0.f) {
                for (int e = 0; e < E; ++e) mine[nt * E + e] = mine[k * E + e];
                ++nt;
            }
        float *out = ws_prev + ((size_t)b * M + p) * (HH_MAX_EMB + 1);
        out[HH_MAX_EMB] = (float)nt;
        pok[p] = nt != 0;
        if (nt) np_mean_rows(mine, nt, E, out);
    }
    __syncthreads();
    // work lists for the arg-max kernel: every joint still missing of a person that has a mean tag, in 8 queues by map ((b*K + k) % 8).
    // A queue is served by the workgroups of ONE XCD, so the several people that miss the same joint of an image scan that map's
    // bounds out of the same L2 instead of fetching them once per XCD.  One global atomic per queue and image: the jobs take their
    // places inside the image's share by LDS atomics (round 3: one returning global atomic per job, issued joint by joint by the
    // person's thread -- fifteen dependent round trips, most of this kernel's time).
    const int cap = gridDim.x * M * K;  // queue capacity = every (b, p, k)
    __shared__ unsigned short jslot[HH_MAX_PEOPLE * 64];  // a job's place inside the image's share of its queue
    for (int i = tid; i < P * K; i += 256) {
        const int p = i / K;
        if (sl[i] == 0.f && pok[p]) jslot[i] = (unsigned short)atomicAdd(&qcnt[(b * K + i - p * K) & 7], 1);
    }
    __syncthreads();
    if (tid < 8) qbase[tid] = qcnt[tid] ? atomicAdd(ws_jobs + tid, qcnt[tid]) : 0;
    __syncthreads();
    for (int i = tid; i < P * K; i += 256) {
        const int p = i / K, kk = i - p * K, qx = (b * K + kk) & 7;
        if (sl[i] == 0.f && pok[p]) ws_jobs[8 + qx * cap + qbase[qx] + jslot[i]] = (b << 16) | (p << 8) | kk;
    }
}

hipError_t launch_adjust_scores(const DecodeSrc &src, int M, int adjust, int refine, float *joints, const int32_t *num_people, float *scores,
                                float *ws_prev, int32_t *ws_jobs, hipStream_t s)
{
    hipLaunchKernelGGL(adjust_scores_kernel, dim3(src.B), dim3(256), 0, s, src, M, adjust, refine, joints, num_people, scores, ws_prev, ws_jobs);
    return hipGetLastError();
}

// ------------------------------------------------------------------ refine
// grouping.py:193-250.  (1) per person: the mean tag of its detected joints and the lists of its missing joints: adjust_scores_kernel.
// (1b) per (b,k) quarter-res cell: [lo, hi] of the 3x3 tag taps: tag_bounds_part(), in front of match_kernel (it rides in that launch).
// (2) per (person, joint) with score == 0: argmax over the full map of hm - round(||tag - mean||)
// (first index among equal values, as np.argmax).  Exact branch-and-bound over 4x4-pixel cells:
//   ub(cell) = max_hm(cell) - rint(lower bound of the tag distance over the cell) >= every value in the cell,
//   where max_hm is the exact cell maximum written by the NMS kernel and the distance bound comes from the
//   min/max of the 3x3 quarter-res tag taps every pixel of the cell interpolates (+ rounding slack).
// Each thread evaluates its most promising cell exactly, the workgroup maximum of those is a valid lower
// bound, and only cells with ub >= that bound are evaluated (typically a handful out of H*W/16).
__global__ __launch_bounds__(256, 4) void refine_argmax_kernel(const DecodeSrc src, int M, const int32_t *__restrict__ ws_jobs,
                                                            const float *__restrict__ ws_prev, const float *__restrict__ cellmax,
                                                            const float *__restrict__ tagb, float *__restrict__ joints)
{
    __shared__ u64 wbest[4];
    const int tid = threadIdx.x, E = src.E;
    // persistent grid; workgroup x runs on XCD x % 8 (round-robin dispatch, grid a multiple of 8) and serves queue x % 8
    const int qx = blockIdx.x & 7, cap = src.B * M * src.K;
    const int njobs = ws_jobs[qx];
    for (int job = blockIdx.x >> 3; job < njobs; job += gridDim.x >> 3) {
        const int code = ws_jobs[8 + qx * cap + job];
        const int b = code >> 16, p = (code >> 8) & 0xff, k = code & 0xff;
        const float *prev = ws_prev + ((size_t)b * M + p) * (HH_MAX_EMB + 1);
        float mean[HH_MAX_EMB];
        for (int e = 0; e < E; ++e) mean[e] = prev[e];
        u64 best = 0ull;
        auto pixel = [&](int y, int x) {  // generic path: the same expressions as heat_at / tag_at
            float s = 0.f;
            for (int e = 0; e < E; ++e) {
                float d = tag_at(src, b, k, y, x, e) - mean[e];
                d = d * d;
                s = e ? s + d : d;
            }
            const float v = heat_at(src, b, k, y, x) - rintf(__fsqrt_rn(s));
            const u64 key = make_key(v, (unsigned)(y * src.W + x));
            best = key > best ? key : best;
        };
        if (src.mode == 1) {
            for (int x = tid; x < src.W; x += 256)
                for (int y = 0; y < src.H; ++y) pixel(y, x);
        } else {
            const int hq = src.H >> 2, wq = src.W >> 2, wh = src.W >> 1;
            const float *avg = src.avg ? src.avg + ((size_t)b * src.K + k) * (size_t)(src.H >> 1) * wh : nullptr;
            const unsigned short *cmaxu = reinterpret_cast<const unsigned short *>(cellmax) + ((size_t)b * src.K + k) * hq * wq;
            auto cmax_at = [&](int c) { return __uint_as_float((unsigned)cmaxu[c] << 16); };  // bf16 upper bound of the cell maximum
            constexpr int TO[4] = {0, 0, 1, 1};  // tag source row offset (from q-1) of sub-pixel j, x4 upsampling
            constexpr float TW1[4] = {0.625f, 0.875f, 0.125f, 0.375f};
            constexpr int HO[4] = {0, 1, 1, 2};  // heat source row offset (from 2q-1), x2 upsampling
            constexpr float HW1[4] = {0.75f, 0.25f, 0.75f, 0.25f};
            // exact evaluation of the 16 pixels of one cell.  Interior: the bilinear source rows/cols and
            // weights are fixed, exactly representable patterns, taps are loaded once and the horizontal
            // interpolations shared -- per pixel the arithmetic is bilerp()'s, bit for bit.
            auto eval_cell = [&](int qy, int qx) {
                if (qy == 0 || qx == 0 || qy == hq - 1 || qx == wq - 1) {  // clamped borders
                    for (int jy = 0; jy < 4; ++jy)
                        for (int jx = 0; jx < 4; ++jx) pixel(4 * qy + jy, 4 * qx + jx);
                    return;
                }
                float hrow[4][4];
                float a4[4][4];  // the stage average at half-res rows 2qy-1 .. 2qy+2, columns 2qx-1 .. 2qx+2
                if (src.avg) {
                    const float *a0 = avg + (size_t)(2 * qy - 1) * wh + 2 * qx - 1;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int cc = 0; cc < 4; ++cc) a4[r][cc] = a0[r * wh + cc];
                } else {
                    // not materialised (default path): formed from the 3x3 quarter-res and 4x4 half-res samples under the cell, all
                    // loaded before the first use; interior cells, so the x2 source patterns are the fixed ones of avg_at()
                    const float *q0 = src.hm_q + (size_t)b * src.hm_q_bs + ((size_t)k * hq + qy - 1) * wq + qx - 1;
                    const float *h0 = src.hm_h + (size_t)b * src.hm_h_bs + ((size_t)k * (src.H >> 1) + 2 * qy - 1) * wh + 2 * qx - 1;
                    float t9[3][3], h16[4][4];
#pragma unroll
                    for (int r = 0; r < 3; ++r)
#pragma unroll
                        for (int cc = 0; cc < 3; ++cc) t9[r][cc] = q0[r * wq + cc];
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int cc = 0; cc < 4; ++cc) h16[r][cc] = h0[r * wh + cc];
                    constexpr int QO[4] = {0, 0, 1, 1};  // lower quarter-res sample (from q-1) of half-res sample 2q-1+j
                    constexpr float QW0[4] = {0.75f, 0.25f, 0.75f, 0.25f}, QW1[4] = {0.25f, 0.75f, 0.25f, 0.75f};
                    float uph[3][4];
#pragma unroll
                    for (int r = 0; r < 3; ++r)
#pragma unroll
                        for (int cc = 0; cc < 4; ++cc) uph[r][cc] = __builtin_fmaf(t9[r][QO[cc]], QW0[cc], t9[r][QO[cc] + 1] * QW1[cc]);
#pragma unroll
                    for (int r = 0; r < 4; ++r)
#pragma unroll
                        for (int cc = 0; cc < 4; ++cc)
                            a4[r][cc] = (__builtin_fmaf(uph[QO[r]][cc], QW0[r], uph[QO[r] + 1][cc] * QW1[r]) + h16[r][cc]) / 2.0f;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int jx = 0; jx < 4; ++jx)
                        hrow[r][jx] = __builtin_fmaf(a4[r][HO[jx]], 1.f - HW1[jx], a4[r][HO[jx] + 1] * HW1[jx]);
                float dist2[16];
#pragma unroll
                for (int e = 0; e < HH_MAX_EMB; ++e) {
                    if (e >= E) break;
                    const float *t0 = src.tags_q[e] + (size_t)b * src.tags_bs[e] + (size_t)k * hq * wq + (size_t)(qy - 1) * wq + qx - 1;
                    float trow[3][4];
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        const float t[3] = {t0[r * wq], t0[r * wq + 1], t0[r * wq + 2]};
#pragma unroll
                        for (int jx = 0; jx < 4; ++jx)
                            trow[r][jx] = __builtin_fmaf(t[TO[jx]], 1.f - TW1[jx], t[TO[jx] + 1] * TW1[jx]);
                    }
#pragma unroll
                    for (int jy = 0; jy < 4; ++jy)
#pragma unroll
                        for (int jx = 0; jx < 4; ++jx) {
                            const float tg = __builtin_fmaf(trow[TO[jy]][jx], 1.f - TW1[jy], trow[TO[jy] + 1][jx] * TW1[jy]);
                            float d = tg - mean[e];
                            d = d * d;
                            dist2[jy * 4 + jx] = e ? dist2[jy * 4 + jx] + d : d;
                        }
                }
#pragma unroll
                for (int jy = 0; jy < 4; ++jy)
#pragma unroll
                    for (int jx = 0; jx < 4; ++jx) {
                        const float hv = __builtin_fmaf(hrow[HO[jy]][jx], 1.f - HW1[jy], hrow[HO[jy] + 1][jx] * HW1[jy]);
                        const float v = hv - rintf(__fsqrt_rn(dist2[jy * 4 + jx]));
                        const u64 key = make_key(v, (unsigned)((4 * qy + jy) * src.W + 4 * qx + jx));
                        best = key > best ? key : best;
                    }
            };
            // upper bound of a cell
            const unsigned *tb = reinterpret_cast<const unsigned *>(tagb) + ((size_t)b * src.K + k) * hq * wq * E;
            auto cell_ub = [&](int c) -> float {
                if (E == 1) {  // (the bound of the scan above, see there)
                    const unsigned lh = tb[c];
                    const float lo = __uint_as_float(lh << 16), hi = __uint_as_float(lh & 0xffff0000u);
                    return cmax_at(c) - rintf(fmaxf(fmaxf(mean[0] - hi, lo - mean[0]), 0.f) * (1.f - 2e-6f));
                }
                float lb2 = 0.f;
                for (int e = 0; e < E; ++e) {
                    const unsigned lh = tb[(size_t)c * E + e];
                    const float lo = __uint_as_float(lh << 16), hi = __uint_as_float(lh & 0xffff0000u);
                    const float d = fmaxf(fmaxf(mean[e] - hi, lo - mean[e]), 0.f);
                    lb2 += d * d;
                }
                const float lb = __fsqrt_rn(lb2) * (1.f - 2e-6f);  // below the reference's own rounded distance
                return cmax_at(c) - rintf(lb);
            };
            const int ncells = hq * wq;
            float my_ub = -INFINITY;
            int my_cell = -1;
            // Both scans are latency chains if taken one cell at a time (a memory round trip per iteration): U cells' loads are
            // issued before the first is used.
            constexpr int U = 8;
            int c = tid;
            if (E == 1)  // (the usual single embedding; more dimensions take the plain loop below)
                for (; c + (U - 1) * 256 < ncells; c += U * 256) {
                    unsigned lh[U];
                    unsigned short cm[U];
#pragma unroll
                    for (int u = 0; u < U; ++u) { cm[u] = cmaxu[c + u * 256]; lh[u] = tb[c + u * 256]; }
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const float lo = __uint_as_float(lh[u] << 16), hi = __uint_as_float(lh[u] & 0xffff0000u);
                        const float d = fmaxf(fmaxf(mean[0] - hi, lo - mean[0]), 0.f);
                        // One embedding: the reference's distance sqrt((tag - mean)^2) of any pixel of the cell is >= d (1 - 2^-24)^2.5, so
                        // d (1 - 2e-6) bounds it from below without the square and the IEEE square root of cell_ub() (two ldexp, a compare and
                        // two selects around v_sqrt_f32: 7 of the scan's 25 instructions per cell)
                        const float ub = __uint_as_float((unsigned)cm[u] << 16) - rintf(d * (1.f - 2e-6f));
                        if (ub > my_ub) { my_ub = ub; my_cell = c + u * 256; }
                    }
                }
            for (; c < ncells; c += 256) {
                const float ub = cell_ub(c);
                if (ub > my_ub) { my_ub = ub; my_cell = c; }
            }
            if (my_cell >= 0) eval_cell(my_cell / wq, my_cell % wq);
            // workgroup-wide lower bound = best exactly evaluated value so far
            u64 wb = wave_max_u64(best);
            if ((tid & 63) == 0) wbest[tid >> 6] = wb;
            __syncthreads();
            u64 g = wbest[0];
            for (int w = 1; w < 4; ++w) g = wbest[w] > g ? wbest[w] : g;
            __syncthreads();
            unsigned gb = (unsigned)(g >> 32);  // invert make_key's order-preserving map
            gb = (gb & 0x80000000u) ? (gb & 0x7fffffffu) : ~gb;
            const float bound = __uint_as_float(gb);
            // ub(c) <= cmax[c] (the distance term is >= 0): most cells are rejected on the cell maximum alone and the
            // tag bounds (2/3 of the bytes of a scan) are not read again
            constexpr int U2 = 16;
            c = tid;
            for (; c + (U2 - 1) * 256 < ncells; c += U2 * 256) {
                unsigned short cm[U2];
#pragma unroll
                for (int u = 0; u < U2; ++u) cm[u] = cmaxu[c + u * 256];
                unsigned surv = 0;
#pragma unroll
                for (int u = 0; u < U2; ++u)
                    surv |= (c + u * 256 != my_cell && !(__uint_as_float((unsigned)cm[u] << 16) < bound)) ? (1u << u) : 0u;
                while (surv) {  // rare
                    const int cc = c + (__builtin_ctz(surv)) * 256;
                    surv &= surv - 1;
                    if (cell_ub(cc) >= bound) eval_cell(cc / wq, cc % wq);
                }
            }
            for (; c < ncells; c += 256) {
                if (c == my_cell || cmax_at(c) < bound) continue;
                if (cell_ub(c) >= bound) eval_cell(c / wq, c % wq);
            }
        }
        const u64 wb2 = wave_max_u64(best);
        if ((tid & 63) == 0) wbest[tid >> 6] = wb2;
        __syncthreads();
        if (tid == 0) {
            u64 g = wbest[0];
            for (int w = 1; w < 4; ++w) g = wbest[w] > g ? wbest[w] : g;
            // (3) grouping.py:238-249: the joint is filled in if the value at the arg-max is positive (it had score 0: only such
            // joints are queued), with the quarter-pixel shift of `adjust` in float64 as numpy computes it.  Nothing else reads or
            // writes this joint's slot, so the job's own workgroup applies it (round 2: a launch of its own over a result table).
            if (g != 0ull) {  // (every scan evaluates at least one pixel, and no key is 0)
                float *j = joints + (((size_t)b * M + p) * src.K + k) * (3 + E);
                const unsigned idx = 0xffffffffu - (unsigned)(g & 0xffffffffull);
                const int y = (int)(idx / (unsigned)src.W), x = (int)(idx % (unsigned)src.W);
                const int xr = min(x + 1, src.W - 1), xl = max(x - 1, 0), yd = min(y + 1, src.H - 1), yu = max(y - 1, 0);
                // (all five samples are fetched before the first is looked at: one round trip, not three)
                const float val = heat_at(src, b, k, y, x);
                const float hr = heat_at(src, b, k, y, xr), hl = heat_at(src, b, k, y, xl), hd = heat_at(src, b, k, yd, x), hu = heat_at(src, b, k, yu, x);
                if (val > 0.f) {
                    double fx = (double)x + 0.5, fy = (double)y + 0.5;
                    if (hr > hl) fx += 0.25; else fx -= 0.25;
                    if (hd > hu) fy += 0.25; else fy -= 0.25;
                    j[0] = (float)fx; j[1] = (float)fy; j[2] = val;
                }
            }
        }
        __syncthreads();
    }
}

hipError_t launch_refine(const DecodeSrc &src, int M, float *joints, const float *ws_prev, const int32_t *ws_jobs, const float *cellmax,
                         const float *tagb, hipStream_t s)
{
    hipLaunchKernelGGL(refine_argmax_kernel, dim3(2048), dim3(256), 0, s, src, M, ws_jobs, ws_prev, cellmax, tagb, joints);
    return hipGetLastError();
}

// ------------------------------------------------------------------ multi-scale aggregation (extension)
// dst (+)= weight * bilinear(src -> HxW): F.interpolate(mode="bilinear", align_corners=False) arithmetic for any
// scale ratio.  The reference only ever calls its resize helper with scale 1 (keypoints/model.py:73); averaging the
// heatmaps of several input scales is the HigherHRNet-paper test-time augmentation named by BASELINE.json configs[3].
__global__ __launch_bounds__(256) void resize_accumulate_kernel(const float *__restrict__ src, int64_t src_bs, int K, int h, int w,
                                                                float *__restrict__ dst, int64_t dst_bs, int H, int W, float sy,
                                                                float sx, float weight, int init)
{
    const int k = blockIdx.y, b = blockIdx.z;
    const float *img = src + (size_t)b * src_bs + (size_t)k * h * w;
    float *o = dst + (size_t)b * dst_bs + (size_t)k * H * W;
    for (int x = threadIdx.x; x < W; x += 256) {
        const Lin lx = src_index(w, sx, x);
        for (int y = blockIdx.x; y < H; y += gridDim.x) {
            const float v = weight * bilerp(img, w, src_index(h, sy, y), lx);
            o[(size_t)y * W + x] = init ? v : o[(size_t)y * W + x] + v;
        }
    }
}
hipError_t launch_resize_accumulate(const float *src, int64_t src_bs, int B, int K, int h, int w, float *dst, int64_t dst_bs, int H,
                                    int W, float weight, int init, hipStream_t s)
{
    const float sy = (float)h / (float)H, sx = (float)w / (float)W;
    hipLaunchKernelGGL(resize_accumulate_kernel, dim3(H < 64 ? H : 64, K, B), dim3(256), 0, s, src, src_bs, K, h, w, dst, dst_bs, H, W,
                       sy, sx, weight, init);
    return hipGetLastError();
}
