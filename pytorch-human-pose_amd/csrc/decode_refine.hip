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
                for (int c = 0; c < 3; ++c) t9[r][c] = ldg(q, yq[r] + xq[c]);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int yo = min(max(2 * qy - 1 + i, 0), hh - 1) * wh;
#pragma unroll
                for (int j = 0; j < 4; ++j) h16[i][j] = ldg(h, yo + min(max(2 * qx - 1 + j, 0), wh - 1));
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
                for (int c = 0; c < 3; ++c) t9[r][c] = ldg(t, yq[r] + xq[c]);
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

#ifndef REFINE_THREADS
#define REFINE_THREADS 128  // per job.  The evaluation needs ~160 registers, i.e. 3 wavefronts per SIMD: with 256-thread workgroups 768 jobs
#endif                      // run at once and the bench's ~900 take two rounds; with 128 threads 1536 do, each a little slower
constexpr int RT = REFINE_THREADS, RW = RT / 64;
__global__ __launch_bounds__(RT, 3) void refine_bb_kernel(const DecodeSrc src, int M, const int32_t *__restrict__ ws_jobs,
                                                           const float *__restrict__ ws_prev, const unsigned short *__restrict__ cellub,
                                                           const unsigned *__restrict__ tagb, const unsigned short *__restrict__ supmax,
                                                           const unsigned *__restrict__ suptag, float *__restrict__ joints)
{
    __shared__ u64 wbest[RW];
    __shared__ int sdone[RW];
    __shared__ unsigned clist[RCAP];
    __shared__ int ncl, ovf;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, E = src.E;
    const int hq = src.H >> 2, wq = src.W >> 2, nsy = (hq + 7) >> 3, nsx = (wq + 7) >> 3, nsup = nsy * nsx;
    // persistent grid; workgroup x runs on XCD x % 8 (round-robin dispatch, grid a multiple of 8) and serves queue x % 8: the
    // several people that miss the same joint of an image read that map's bounds out of one L2
    const int qxi = blockIdx.x & 7, cap = src.B * M * src.K;
    const int njobs = ws_jobs[qxi];
    for (int job = blockIdx.x >> 3; job < njobs; job += gridDim.x >> 3) {
        const int code = ws_jobs[8 + qxi * cap + job];
        const int b = __builtin_amdgcn_readfirstlane(code >> 16), p = __builtin_amdgcn_readfirstlane((code >> 8) & 0xff), k = __builtin_amdgcn_readfirstlane(code & 0xff);
        const float *prev = ws_prev + ((size_t)b * M + p) * (HH_MAX_EMB + 1);
        float mean[HH_MAX_EMB];
        for (int e = 0; e < E; ++e) mean[e] = prev[e];
        const size_t map = (size_t)b * src.K + k;
        const unsigned short *cm = cellub + map * hq * wq, *sm = supmax + map * nsup;
        const unsigned *tb = tagb + map * hq * wq * E, *st = suptag + map * nsup * E;
        CellEval ev{src, b, k, E, mean, 0ull};
        if (tid == 0) { ncl = 0; ovf = 0; }
        float bound = 0.f;
        int dn[RW];
#pragma unroll
        for (int w = 0; w < RW; ++w) dn[w] = -1;
        // Two rounds over ONE evaluation loop (the evaluation is ~400 instructions and 100+ registers: one copy of it):
        //   round 0: every wavefront lists the 64 cells of its most promising super; the best value found is the lower bound B;
        //   round 1: supers whose bound reaches B are opened, cells whose bound reaches B are listed.
#pragma unroll 1
        for (int round = 0; round < 2; ++round) {
            if (round == 0) {
                float bs = -INFINITY;
                int bi = -1;
                for (int s = tid; s < nsup; s += RT) {
                    const float su = bound_of(sm[s], st + (size_t)s * E, E, mean);
                    if (su > bs || bi < 0) { bs = su; bi = s; }
                }
                const u64 wk = wave_max_u64(bi >= 0 ? make_key(bs, (unsigned)bi) : 0ull);
                int s0 = -1;
                unsigned cell = ~0u;
                if (wk) {  // wave-uniform
                    s0 = (int)(0xffffffffu - (unsigned)(wk & 0xffffffffull));
                    const int qy = 8 * (s0 / nsx) + (lane >> 3), qx = 8 * (s0 % nsx) + (lane & 7);
                    if (qy < hq && qx < wq) cell = ((unsigned)qy << 16) | (unsigned)qx;
                }
                clist[tid] = cell;
                if (lane == 0) sdone[wv] = s0;
            } else {
                // Four supers per step: a step is two dependent load rounds (heat bounds, then tag hulls behind them), and a wavefront
                // that takes its supers one at a time waits for every one of them in turn.
                for (int s0 = 0; s0 < nsup; s0 += RT) {
                    const int s = s0 + tid;
                    bool alive = false;
                    if (s < nsup) {
                        alive = true;
#pragma unroll
                        for (int w = 0; w < RW; ++w) alive = alive && s != dn[w];
                        if (alive) alive = bound_of(sm[s], st + (size_t)s * E, E, mean) >= bound;
                    }
                    u64 m = __ballot(alive);
                    while (m) {
                        int cc[4];
                        unsigned pk[4];
                        unsigned short hb[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            cc[u] = -1; hb[u] = 0; pk[u] = ~0u;
                            if (m) {  // (wave-uniform)
                                const int ss = s0 + wv * 64 + __builtin_ctzll(m);
                                m &= m - 1;
                                const int qy = 8 * (ss / nsx) + (lane >> 3), qx = 8 * (ss % nsx) + (lane & 7);
                                if (qy < hq && qx < wq) { cc[u] = qy * wq + qx; pk[u] = ((unsigned)qy << 16) | (unsigned)qx; }
                            }
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            if (cc[u] >= 0) hb[u] = cm[cc[u]];
                        bool surv[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            // (the heat bound alone rejects most cells: the tag hull is only read behind it)
                            surv[u] = cc[u] >= 0 && !(__uint_as_float((unsigned)hb[u] << 16) < bound);
                            if (surv[u]) surv[u] = bound_of(hb[u], tb + (size_t)cc[u] * E, E, mean) >= bound;
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const u64 smk = __ballot(surv[u]);
                            if (smk) {
                                int base = 0;
                                if (lane == 0) base = atomicAdd(&ncl, __popcll(smk));
                                base = __builtin_amdgcn_readfirstlane(base);
                                const int pos = base + __popcll(smk & ((1ull << lane) - 1ull));
                                if (surv[u]) {
                                    if (pos < RCAP) clist[pos] = pk[u];
                                    else ovf = 1;  // (does not happen on real maps: thousands of cells tie with the bound)
                                }
                            }
                        }
                    }
                }
            }
            __syncthreads();
            const int n = round == 0 ? RT : min(ncl, RCAP);
            for (int i = tid; i < n; i += RT) {
                const unsigned c = clist[i];
                if (c != ~0u) ev((int)(c >> 16), (int)(c & 0xffffu));
            }
            const u64 wb = wave_max_u64(ev.best);
            if (round) __syncthreads();  // (round 0's wbest was read by everybody before round 1's list was complete)
            if (lane == 0) wbest[wv] = wb;
            __syncthreads();
            if (round == 0) {
                u64 g = wbest[0];
                for (int w = 1; w < RW; ++w) g = wbest[w] > g ? wbest[w] : g;
                unsigned gb = (unsigned)(g >> 32);  // invert make_key's order-preserving map
                gb = (gb & 0x80000000u) ? (gb & 0x7fffffffu) : ~gb;
                bound = __uint_as_float(gb);  // (every map has a super, every super a cell: g != 0)
                #pragma unroll
                for (int w = 0; w < RW; ++w) dn[w] = sdone[w];
            }
        }
        if (ovf) {  // (block-uniform: read behind the barriers above) the list was too short: every cell whose bound reaches B, where it stands
            for (int c = tid; c < hq * wq; c += RT)
                if (bound_of(cm[c], tb + (size_t)c * E, E, mean) >= bound) ev(c / wq, c % wq);
            const u64 wb = wave_max_u64(ev.best);
            __syncthreads();
            if (lane == 0) wbest[wv] = wb;
            __syncthreads();
        }
        if (wv == 0) {
            u64 gg = wbest[0];
            for (int w = 1; w < RW; ++w) gg = wbest[w] > gg ? wbest[w] : gg;
            // grouping.py:238-249: the joint is filled in if the heat value at the arg-max is positive (it had score 0: only such joints
            // are queued), with the quarter-pixel shift of `adjust` in float64 as numpy computes it.  Nothing else reads or writes this
            // joint's slot, so the job's own workgroup applies it.  The five heat samples (the pixel, right, left, below, above) are
            // formed by five lanes side by side: each is twenty loads behind a chain of index arithmetic.
            if (gg != 0ull) {  // (wave-uniform)
                const unsigned idx = 0xffffffffu - (unsigned)(gg & 0xffffffffull);
                const int y = (int)(idx / (unsigned)src.W), x = (int)(idx % (unsigned)src.W);
                const int sx = lane == 1 ? min(x + 1, src.W - 1) : lane == 2 ? max(x - 1, 0) : x;
                const int sy = lane == 3 ? min(y + 1, src.H - 1) : lane == 4 ? max(y - 1, 0) : y;
                float hs1[1] = {0.f};
                if (lane < 5) {
                    const int ys[1] = {sy}, xs[1] = {sx};
                    heat_otf<1>(src, b, k, ys, xs, hs1);
                }
                const float hs = hs1[0];
                const float val = __shfl(hs, 0), hr = __shfl(hs, 1), hl = __shfl(hs, 2), hd = __shfl(hs, 3), hu = __shfl(hs, 4);
                if (lane == 0 && val > 0.f) {
                    float *j = joints + (((size_t)b * M + p) * src.K + k) * (3 + E);
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
    hipLaunchKernelGGL(refine_bb_kernel, dim3(2048 * (256 / RT)), dim3(RT), 0, s, src, M, ws_jobs, ws_prev, reinterpret_cast<const unsigned short *>(cellmax),
                       reinterpret_cast<const unsigned *>(tagb), supmax, suptag, joints);
    return hipGetLastError();
}
