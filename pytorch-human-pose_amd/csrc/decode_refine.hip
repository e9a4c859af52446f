// refine (grouping.py:193-250), round 4: the full-map arg-max of hm - round(||tags - mean||) for every missing joint as a TWO-LEVEL
// exact branch-and-bound, for the default path of hh_decode (mode 0, no materialised stage average).
//
// Round 3 scanned all H*W/16 cell bounds of the map twice per job (64 cells per thread, eight dependent rounds of loads) and evaluated
// survivors one per thread and iteration, so a job's time was the time of its unluckiest thread: ~150 us per job, one job per
// workgroup, the kernel as long as its slowest job.  Here:
//   * 8x8-cell SUPERS carry a bound of their own (supmax: the largest cell bound, written by peaks_region_kernel; suptag: the hull
//     of the cells' tag ranges, written by tag_bounds_part): one load round gives every thread the bound of "its" super;
//   * each wavefront evaluates all 64 cells of its most promising super exactly, one cell per lane: the workgroup maximum is the
//     lower bound B;
//   * only supers whose bound reaches B are opened, a wavefront per super and a lane per cell; the cells whose bound reaches B are
//     collected in an LDS list (ballot + prefix inside the wavefront, one LDS atomic per opened super) and evaluated 256 at a time.
// A cell is evaluated from the net's outputs directly (3x3 quarter-res + 4x4 half-res heat samples, 3x3 tag samples per embedding),
// border cells included: sources are fetched with clamped indices and the two cases where torch's source pair is not the static
// pair of the parity (destination index 0 of the x2 resize, the first two of the x4 resize) take the weights (0, 1), which return
// the one sample that counts as long as the other is finite.  Same expressions as bilerp() / stage_average_kernel per value, so the
// arg-max and the value written are bit-identical to round 3's (and to np.argmax: first index among equal values via the key).
#include "decode_dev.h"

namespace {

constexpr int RCAP = 4096;  // surviving cells kept per job before they are evaluated (more: evaluated where they are found)

struct CellEval {
    const DecodeSrc &src;
    int b, k, E;
    const float *mean;
    u64 best;

    // exact values of the 16 pixels of cell (qy, qx)
    __device__ __forceinline__ void operator()(int qy, int qx)
    {
        const int H = src.H, W = src.W, hq = H >> 2, wq = W >> 2, hh = H >> 1, wh = W >> 1;
        constexpr int QO[4] = {0, 0, 1, 1};  // first source (from q-1) of half-res sample 2q-1+j (x2) and of pixel 4q+j (x4)
        constexpr int HO[4] = {0, 1, 1, 2};  // first averaged sample (from 2q-1) of pixel 4q+j (x2)
        int yq[3], xq[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) { yq[r] = min(max(qy - 1 + r, 0), hq - 1) * wq; xq[r] = min(max(qx - 1 + r, 0), wq - 1); }
        // ---- the stage average at half-res rows 2qy-1 .. 2qy+2, columns 2qx-1 .. 2qx+2
        float a4[4][4];
        {
            const float *q = src.hm_q + (size_t)b * src.hm_q_bs + (size_t)k * hq * wq;
            const float *h = src.hm_h + (size_t)b * src.hm_h_bs + (size_t)k * hh * wh;
            float t9[3][3], h16[4][4];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) t9[r][c] = q[yq[r] + xq[c]];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int yo = min(max(2 * qy - 1 + i, 0), hh - 1) * wh;
#pragma unroll
                for (int j = 0; j < 4; ++j) h16[i][j] = h[yo + min(max(2 * qx - 1 + j, 0), wh - 1)];
            }
            float uph[3][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {  // half-res column 2qx-1+j: odd for even j
                const bool first = 2 * qx - 1 + j <= 0;
                const float w0 = (j & 1) ? (first ? 1.f : 0.25f) : 0.75f, w1 = (j & 1) ? (first ? 0.f : 0.75f) : 0.25f;
#pragma unroll
                for (int r = 0; r < 3; ++r) uph[r][j] = __builtin_fmaf(t9[r][QO[j]], w0, t9[r][QO[j] + 1] * w1);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool first = 2 * qy - 1 + i <= 0;
                const float w0 = (i & 1) ? (first ? 1.f : 0.25f) : 0.75f, w1 = (i & 1) ? (first ? 0.f : 0.75f) : 0.25f;
#pragma unroll
                for (int j = 0; j < 4; ++j) a4[i][j] = (__builtin_fmaf(uph[QO[i]][j], w0, uph[QO[i] + 1][j] * w1) + h16[i][j]) / 2.0f;
            }
            // last cell row / column: the pair of pixel 4q+3 is (2q+1, 2q+1) in torch (upper index clamped)
            if (qy == hq - 1)
#pragma unroll
                for (int j = 0; j < 4; ++j) a4[3][j] = a4[2][j];
            if (qx == wq - 1)
#pragma unroll
                for (int i = 0; i < 4; ++i) a4[i][3] = a4[i][2];
        }
        // ---- heat: x2 from the averaged samples; pixel 4q+j is even for even j
        float hrow[4][4];
#pragma unroll
        for (int jx = 0; jx < 4; ++jx) {
            const bool first = 4 * qx + jx == 0;
            const float w0 = (jx & 1) ? 0.75f : (first ? 0.f : 0.25f), w1 = (jx & 1) ? 0.25f : (first ? 1.f : 0.75f);
#pragma unroll
            for (int i = 0; i < 4; ++i) hrow[i][jx] = __builtin_fmaf(a4[i][HO[jx]], w0, a4[i][HO[jx] + 1] * w1);
        }
        // ---- tags: x4 from the quarter-res samples; squared distance to the person's mean, summed over the embeddings in order
        constexpr float TW1[4] = {0.625f, 0.875f, 0.125f, 0.375f};
        float dist2[16];
#pragma unroll
        for (int e = 0; e < HH_MAX_EMB; ++e) {
            if (e >= E) break;
            const float *t = src.tags_q[e] + (size_t)b * src.tags_bs[e] + (size_t)k * hq * wq;
            float t9[3][3], trow[3][4];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c) t9[r][c] = t[yq[r] + xq[c]];
#pragma unroll
            for (int jx = 0; jx < 4; ++jx) {
                const bool first = qx == 0 && jx < 2;  // source position below 0: torch reads sample 0 with weight 1
                const float w0 = first ? 0.f : 1.f - TW1[jx], w1 = first ? 1.f : TW1[jx];
#pragma unroll
                for (int r = 0; r < 3; ++r) trow[r][jx] = __builtin_fmaf(t9[r][QO[jx]], w0, t9[r][QO[jx] + 1] * w1);
            }
#pragma unroll
            for (int jy = 0; jy < 4; ++jy) {
                const bool first = qy == 0 && jy < 2;
                const float w0 = first ? 0.f : 1.f - TW1[jy], w1 = first ? 1.f : TW1[jy];
#pragma unroll
                for (int jx = 0; jx < 4; ++jx) {
                    float d = __builtin_fmaf(trow[QO[jy]][jx], w0, trow[QO[jy] + 1][jx] * w1) - mean[e];
                    d = d * d;
                    dist2[jy * 4 + jx] = e ? dist2[jy * 4 + jx] + d : d;
                }
            }
        }
#pragma unroll
        for (int jy = 0; jy < 4; ++jy) {
            const bool first = 4 * qy + jy == 0;
            const float w0 = (jy & 1) ? 0.75f : (first ? 0.f : 0.25f), w1 = (jy & 1) ? 0.25f : (first ? 1.f : 0.75f);
#pragma unroll
            for (int jx = 0; jx < 4; ++jx) {
                const float hv = __builtin_fmaf(hrow[HO[jy]][jx], w0, hrow[HO[jy] + 1][jx] * w1);
                const float v = hv - rintf(__fsqrt_rn(dist2[jy * 4 + jx]));
                const u64 key = make_key(v, (unsigned)((4 * qy + jy) * W + 4 * qx + jx));
                best = key > best ? key : best;
            }
        }
    }
};

// upper bound of hm - round(dist) over a set of pixels from the set's heat bound (bf16) and tag hull(s) (bf16 lo | hi << 16)
__device__ __forceinline__ float bound_of(unsigned short hb, const unsigned *__restrict__ tb, int E, const float *mean)
{
    float lb;
    if (E == 1) {
        // one embedding: the reference's distance sqrt((tag - mean)^2) of any pixel is >= d (1 - 2^-24)^2.5, so d (1 - 2e-6) bounds it
        // from below without the square and the IEEE square root
        const unsigned lh = tb[0];
        const float lo = __uint_as_float(lh << 16), hi = __uint_as_float(lh & 0xffff0000u);
        lb = fmaxf(fmaxf(mean[0] - hi, lo - mean[0]), 0.f) * (1.f - 2e-6f);
    } else {
        float lb2 = 0.f;
        for (int e = 0; e < E; ++e) {
            const unsigned lh = tb[e];
            const float lo = __uint_as_float(lh << 16), hi = __uint_as_float(lh & 0xffff0000u);
            const float d = fmaxf(fmaxf(mean[e] - hi, lo - mean[e]), 0.f);
            lb2 += d * d;
        }
        lb = __fsqrt_rn(lb2) * (1.f - 2e-6f);  // below the reference's own rounded distance
    }
    return __uint_as_float((unsigned)hb << 16) - rintf(lb);
}

}  // namespace

#ifndef REFINE_WPS
#define REFINE_WPS 3  // workgroups per CU (148 registers: 4 would spill)
#endif
__global__ __launch_bounds__(256, REFINE_WPS) void refine_bb_kernel(const DecodeSrc src, int M, const int32_t *__restrict__ ws_jobs,
                                                           const float *__restrict__ ws_prev, const unsigned short *__restrict__ cellub,
                                                           const unsigned *__restrict__ tagb, const unsigned short *__restrict__ supmax,
                                                           const unsigned *__restrict__ suptag, float *__restrict__ joints)
{
    __shared__ u64 wbest[4];
    __shared__ int sdone[4];
    __shared__ unsigned clist[RCAP];
    __shared__ int ncl;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, E = src.E;
    const int hq = src.H >> 2, wq = src.W >> 2, nsy = (hq + 7) >> 3, nsx = (wq + 7) >> 3, nsup = nsy * nsx;
    // persistent grid; workgroup x runs on XCD x % 8 (round-robin dispatch, grid a multiple of 8) and serves queue x % 8: the
    // several people that miss the same joint of an image read that map's bounds out of one L2
    const int qxi = blockIdx.x & 7, cap = src.B * M * src.K;
    const int njobs = ws_jobs[qxi];
    for (int job = blockIdx.x >> 3; job < njobs; job += gridDim.x >> 3) {
        const int code = ws_jobs[8 + qxi * cap + job];
        const int b = code >> 16, p = (code >> 8) & 0xff, k = code & 0xff;
        const float *prev = ws_prev + ((size_t)b * M + p) * (HH_MAX_EMB + 1);
        float mean[HH_MAX_EMB];
        for (int e = 0; e < E; ++e) mean[e] = prev[e];
        const size_t map = (size_t)b * src.K + k;
        const unsigned short *cm = cellub + map * hq * wq, *sm = supmax + map * nsup;
        const unsigned *tb = tagb + map * hq * wq * E, *st = suptag + map * nsup * E;
        CellEval ev{src, b, k, E, mean, 0ull};
        if (tid == 0) ncl = 0;

        // ---- A. the wavefront's most promising super, every cell of it evaluated (a lane per cell)
        {
            float bs = -INFINITY;
            int bi = -1;
            for (int s = tid; s < nsup; s += 256) {
                const float su = bound_of(sm[s], st + (size_t)s * E, E, mean);
                if (su > bs || bi < 0) { bs = su; bi = s; }
            }
            const u64 wk = wave_max_u64(bi >= 0 ? make_key(bs, (unsigned)bi) : 0ull);
            int s0 = -1;
            if (wk) {  // wave-uniform
                s0 = (int)(0xffffffffu - (unsigned)(wk & 0xffffffffull));
                const int qy = 8 * (s0 / nsx) + (lane >> 3), qx = 8 * (s0 % nsx) + (lane & 7);
                if (qy < hq && qx < wq) ev(qy, qx);
            }
            const u64 wb = wave_max_u64(ev.best);
            if (lane == 0) { wbest[wv] = wb; sdone[wv] = s0; }
        }
        __syncthreads();
        u64 g = wbest[0];
        for (int w = 1; w < 4; ++w) g = wbest[w] > g ? wbest[w] : g;
        unsigned gb = (unsigned)(g >> 32);  // invert make_key's order-preserving map
        gb = (gb & 0x80000000u) ? (gb & 0x7fffffffu) : ~gb;
        const float bound = __uint_as_float(gb);  // (every map has a super, every super a cell: g != 0)
        const int d0 = sdone[0], d1 = sdone[1], d2 = sdone[2], d3 = sdone[3];

        // ---- B. supers whose bound reaches it are opened: cells whose bound reaches it go to the list (or, list full, are evaluated here)
        for (int s0 = 0; s0 < nsup; s0 += 256) {
            const int s = s0 + tid;
            bool alive = false;
            if (s < nsup && s != d0 && s != d1 && s != d2 && s != d3) alive = bound_of(sm[s], st + (size_t)s * E, E, mean) >= bound;
            for (u64 m = __ballot(alive); m; m &= m - 1) {
                const int ss = s0 + wv * 64 + __builtin_ctzll(m);
                const int qy = 8 * (ss / nsx) + (lane >> 3), qx = 8 * (ss % nsx) + (lane & 7);
                const int c = qy * wq + qx;
                bool surv = false;
                // (the heat bound alone rejects most cells: the tag hull is only read behind it)
                if (qy < hq && qx < wq && !(__uint_as_float((unsigned)cm[c] << 16) < bound)) surv = bound_of(cm[c], tb + (size_t)c * E, E, mean) >= bound;
                const u64 smk = __ballot(surv);
                if (smk) {
                    int base = 0;
                    if (lane == 0) base = atomicAdd(&ncl, __popcll(smk));
                    base = __builtin_amdgcn_readfirstlane(base);
                    const int pos = base + __popcll(smk & ((1ull << lane) - 1ull));
                    if (surv) {
                        if (pos < RCAP) clist[pos] = (unsigned)c;
                        else ev(qy, qx);
                    }
                }
            }
        }
        __syncthreads();
        // ---- C. the listed cells, 256 at a time
        {
            const int n = min(ncl, RCAP);
            for (int i = tid; i < n; i += 256) {
                const int c = (int)clist[i];
                ev(c / wq, c % wq);
            }
        }
        const u64 wb2 = wave_max_u64(ev.best);
        __syncthreads();  // (wbest was read above)
        if (lane == 0) wbest[wv] = wb2;
        __syncthreads();
        if (tid == 0) {
            u64 gg = wbest[0];
            for (int w = 1; w < 4; ++w) gg = wbest[w] > gg ? wbest[w] : gg;
            // grouping.py:238-249: the joint is filled in if the heat value at the arg-max is positive (it had score 0: only such joints
            // are queued), with the quarter-pixel shift of `adjust` in float64 as numpy computes it.  Nothing else reads or writes this
            // joint's slot, so the job's own workgroup applies it.
            if (gg != 0ull) {
                float *j = joints + (((size_t)b * M + p) * src.K + k) * (3 + E);
                const unsigned idx = 0xffffffffu - (unsigned)(gg & 0xffffffffull);
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
        __syncthreads();  // (ncl / wbest / sdone are rewritten by the next job)
    }
}

hipError_t launch_refine_bb(const DecodeSrc &src, int M, float *joints, const float *ws_prev, const int32_t *ws_jobs, const float *cellmax,
                            const float *tagb, const unsigned short *supmax, const unsigned *suptag, hipStream_t s)
{
    hipLaunchKernelGGL(refine_bb_kernel, dim3(2048), dim3(256), 0, s, src, M, ws_jobs, ws_prev, reinterpret_cast<const unsigned short *>(cellmax),
                       reinterpret_cast<const unsigned *>(tagb), supmax, suptag, joints);
    return hipGetLastError();
}
