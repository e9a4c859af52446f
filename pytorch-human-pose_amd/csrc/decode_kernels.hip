// Decode kernels: HBM/L2-bound integer + fp32 work, no MFMA.  Built with -ffp-contract=off;
// every fused multiply-add below is explicit because the results must be bit-identical to
// the reference's torch-CPU / numpy arithmetic (see oracle/decode_oracle.c for the
// experimentally pinned formulas).
#include "decode_kernels.h"

#include <math.h>

typedef unsigned long long u64;

// ------------------------------------------------------------------ bilinear sampling
// F.interpolate(mode="bilinear", align_corners=False), torch CPU fp32 (results.py:48-67)
struct Lin { int i0, i1; float w0, w1; };

__device__ __forceinline__ Lin src_index(int in_size, int out_size, int dst)
{
    const float scale = __fdiv_rn((float)in_size, (float)out_size);
    float r = __builtin_fmaf(scale, (float)dst + 0.5f, -0.5f);
    if (r < 0.f) r = 0.f;
    const int a = (int)r;
    float l1 = r - (float)a;
    l1 = fminf(fmaxf(l1, 0.f), 1.f);
    Lin o;
    o.i0 = a; o.i1 = a + (a < in_size - 1 ? 1 : 0); o.w1 = l1; o.w0 = 1.f - l1;
    return o;
}

__device__ __forceinline__ float bilerp(const float *__restrict__ img, int w, const Lin &ly, const Lin &lx)
{
    const float *r0 = img + (size_t)ly.i0 * w, *r1 = img + (size_t)ly.i1 * w;
    const float a = __builtin_fmaf(r0[lx.i0], lx.w0, r0[lx.i1] * lx.w1);
    const float b = __builtin_fmaf(r1[lx.i0], lx.w0, r1[lx.i1] * lx.w1);
    return __builtin_fmaf(a, ly.w0, b * ly.w1);
}

// full-resolution heat value / tag value at (b,k,y,x)
__device__ __forceinline__ float heat_at(const DecodeSrc &s, int b, int k, int y, int x)
{
    if (s.mode == 1) return s.hm_full[(((size_t)b * s.K + k) * s.H + y) * s.W + x];
    const int hh = s.H >> 1, wh = s.W >> 1;
    return bilerp(s.avg + ((size_t)b * s.K + k) * hh * wh, wh, src_index(hh, s.H, y), src_index(wh, s.W, x));
}
__device__ __forceinline__ float tag_at(const DecodeSrc &s, int b, int k, int y, int x, int e)
{
    if (s.mode == 1) return s.tags_full[((((size_t)b * s.K + k) * s.H + y) * s.W + x) * s.E + e];
    const int hq = s.H >> 2, wq = s.W >> 2;
    return bilerp(s.tags_q[e] + (size_t)b * s.tags_bs[e] + (size_t)k * hq * wq, wq, src_index(hq, s.H, y), src_index(wq, s.W, x));
}

// ------------------------------------------------------------------ stage average
// results.py:225-226: match_heatmaps_size (1/4 -> 1/2) then torch.stack(...).mean(dim=0)
__global__ __launch_bounds__(256) void stage_average_kernel(const float *hm_q, int64_t hm_q_bs, const float *hm_h, int64_t hm_h_bs,
                                                            float *avg, int B, int K, int hq, int wq)
{
    const int hh = 2 * hq, wh = 2 * wq;
    const size_t plane = (size_t)hh * wh, total = (size_t)B * K * plane;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int x = (int)(i % wh), y = (int)((i / wh) % hh);
        const int k = (int)((i / plane) % K), b = (int)(i / (plane * K));
        const float up = bilerp(hm_q + (size_t)b * hm_q_bs + (size_t)k * hq * wq, wq, src_index(hq, hh, y), src_index(wq, wh, x));
        const float hv = hm_h[(size_t)b * hm_h_bs + (size_t)k * plane + (size_t)y * wh + x];
        avg[i] = (up + hv) / 2.0f;
    }
}

hipError_t launch_stage_average(const float *hm_q, int64_t hm_q_bs, const float *hm_h, int64_t hm_h_bs, float *avg, int B,
                                int K, int hq, int wq, hipStream_t s)
{
    const size_t total = (size_t)B * K * 4 * hq * wq;
    unsigned grid = (unsigned)((total + 255) / 256);
    if (grid > 16384) grid = 16384;
    hipLaunchKernelGGL(stage_average_kernel, dim3(grid), dim3(256), 0, s, hm_q, hm_q_bs, hm_h, hm_h_bs, avg, B, K, hq, wq);
    return hipGetLastError();
}

// ------------------------------------------------------------------ sortable keys
// Larger key = larger value; between equal values the smaller flat index wins (torch.topk
// leaves that order unspecified; the oracle uses the same rule). -0 == +0; NaN ranks lowest.
__device__ __forceinline__ u64 make_key(float v, unsigned idx)
{
    if (v != v) return 1ull + (u64)(0xffffffffu - idx);
    if (v == 0.f) v = 0.f;
    unsigned bits = __float_as_uint(v);
    bits = (bits & 0x80000000u) ? ~bits : (bits | 0x80000000u);
    return ((u64)bits << 32) | (u64)(0xffffffffu - idx);
}
__device__ __forceinline__ u64 wave_max_u64(u64 v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const u64 o = __shfl_xor(v, off);
        v = o > v ? o : v;
    }
    return v;
}

// ------------------------------------------------------------------ NMS + per-tile top-M
// grouping.py:80-83 (5x5 max-pool NMS: hm * (pool(hm) == hm)) and the first half of
// top_k (grouping.py:147-153).  One workgroup = one 64x64 full-resolution tile of one (b,k)
// map, computed from L2-resident low-res data; separable 5x5 max through LDS; then M rounds
// of workgroup-wide arg-max (wave shuffles + one LDS exchange per round).
__global__ __launch_bounds__(256) void nms_tile_topk_kernel(const DecodeSrc src, int M, int tiles_x, u64 *__restrict__ cand_key,
                                                            float *__restrict__ cand_val)
{
    constexpr int TS = HH_NMS_TILE, HS = TS + 4;
    __shared__ float v[HS][HS + 1];
    __shared__ float rm[HS][TS + 1];
    __shared__ u64 wbest[4];
    const int tile = blockIdx.x, k = blockIdx.y, b = blockIdx.z;
    const int ty = tile / tiles_x, tx = tile % tiles_x;
    const int y0 = ty * TS, x0 = tx * TS;
    const int tid = threadIdx.x;

    for (int i = tid; i < HS * HS; i += 256) {
        const int ly = i / HS, lx = i % HS;
        const int Y = y0 - 2 + ly, X = x0 - 2 + lx;
        v[ly][lx] = (Y >= 0 && Y < src.H && X >= 0 && X < src.W) ? heat_at(src, b, k, Y, X) : -INFINITY;
    }
    __syncthreads();
    for (int i = tid; i < HS * TS; i += 256) {
        const int ly = i / TS, lx = i % TS;
        float m = v[ly][lx];
#pragma unroll
        for (int d = 1; d < 5; ++d) m = fmaxf(m, v[ly][lx + d]);
        rm[ly][lx] = m;
    }
    __syncthreads();
    u64 keys[16];
    float vals[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int p = tid + 256 * j, py = p / TS, px = p % TS;
        float m = rm[py][px];
#pragma unroll
        for (int d = 1; d < 5; ++d) m = fmaxf(m, rm[py + d][px]);
        const float c = v[py + 2][px + 2];
        const int Y = y0 + py, X = x0 + px;
        vals[j] = c * ((m == c) ? 1.0f : 0.0f);
        keys[j] = (Y < src.H && X < src.W) ? make_key(vals[j], (unsigned)(Y * src.W + X)) : 0ull;
    }
    const size_t obase = ((((size_t)b * src.K + k) * gridDim.x) + tile) * M;
    for (int r = 0; r < M; ++r) {
        u64 best = keys[0];
#pragma unroll
        for (int j = 1; j < 16; ++j) best = keys[j] > best ? keys[j] : best;
        const u64 wb = wave_max_u64(best);
        if ((tid & 63) == 0) wbest[tid >> 6] = wb;
        __syncthreads();
        u64 g = wbest[0];
#pragma unroll
        for (int w = 1; w < 4; ++w) g = wbest[w] > g ? wbest[w] : g;
        if (g != 0ull && best == g) {  // keys are unique: exactly one owner
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (keys[j] == g) { cand_val[obase + r] = vals[j]; keys[j] = 0ull; }
        }
        if (tid == 0) cand_key[obase + r] = g;
        __syncthreads();
    }
}

hipError_t launch_nms_tile_topk(const DecodeSrc &src, int M, u64 *cand_key, float *cand_val, hipStream_t s)
{
    const int tiles_x = (src.W + HH_NMS_TILE - 1) / HH_NMS_TILE, tiles_y = (src.H + HH_NMS_TILE - 1) / HH_NMS_TILE;
    hipLaunchKernelGGL(nms_tile_topk_kernel, dim3(tiles_x * tiles_y, src.K, src.B), dim3(256), 0, s, src, M, tiles_x, cand_key,
                       cand_val);
    return hipGetLastError();
}

// ------------------------------------------------------------------ merge -> top_k outputs
// second half of top_k (grouping.py:152-170): global top-M, tag gather, x = idx % w, y = idx / w
__global__ __launch_bounds__(256) void topk_merge_kernel(const DecodeSrc src, int M, int ntiles, u64 *__restrict__ cand_key,
                                                         const float *__restrict__ cand_val, float *__restrict__ tags_k,
                                                         int32_t *__restrict__ coords_k, float *__restrict__ scores_k)
{
    __shared__ u64 wbest[4];
    __shared__ int wpos[4];
    const int k = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int N = ntiles * M;
    u64 *keys = cand_key + ((size_t)b * src.K + k) * N;
    const float *vals = cand_val + ((size_t)b * src.K + k) * N;
    const size_t obase = ((size_t)b * src.K + k) * M;
    for (int r = 0; r < M; ++r) {
        u64 best = 0ull;
        int pos = -1;
        for (int i = tid; i < N; i += 256) {
            const u64 kk = keys[i];
            if (kk > best) { best = kk; pos = i; }
        }
        const u64 wb = wave_max_u64(best);
        if (best == wb && best != 0ull) { wbest[tid >> 6] = wb; wpos[tid >> 6] = pos; }
        else if ((tid & 63) == 0 && wb == 0ull) { wbest[tid >> 6] = 0ull; wpos[tid >> 6] = -1; }
        __syncthreads();
        u64 g = 0ull;
        int gp = -1;
#pragma unroll
        for (int w = 0; w < 4; ++w)
            if (wbest[w] > g) { g = wbest[w]; gp = wpos[w]; }
        if (tid == 0) {
            float sc = 0.f;
            int x = 0, y = 0;
            if (gp >= 0) {
                const unsigned idx = 0xffffffffu - (unsigned)(g & 0xffffffffull);
                sc = vals[gp];
                x = (int)(idx % (unsigned)src.W);
                y = (int)(idx / (unsigned)src.W);
                keys[gp] = 0ull;
            }
            scores_k[obase + r] = sc;
            coords_k[(obase + r) * 2 + 0] = x;
            coords_k[(obase + r) * 2 + 1] = y;
            for (int e = 0; e < src.E; ++e) tags_k[(obase + r) * src.E + e] = tag_at(src, b, k, y, x, e);
        }
        __syncthreads();
    }
}

hipError_t launch_topk_merge(const DecodeSrc &src, int M, int ntiles, u64 *cand_key, const float *cand_val, float *tags_k,
                             int32_t *coords_k, float *scores_k, hipStream_t s)
{
    hipLaunchKernelGGL(topk_merge_kernel, dim3(src.K, src.B), dim3(256), 0, s, src, M, ntiles, cand_key, cand_val, tags_k,
                       coords_k, scores_k);
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

// ------------------------------------------------------------------ match_by_tag
// grouping.py:85-145 with munkres 1.1.4 (munkres.py:114-340) restated wave-parallel: one
// wave per image, cost matrix in LDS (float64), column j on lane j.  Only operations whose
// result is order-independent are spread over lanes; the rest runs on lane 0.
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
    int path[4 * HH_MAX_PEOPLE + 4];
    unsigned char marked[HH_MAX_PEOPLE * MLD];
    unsigned char rc[HH_MAX_PEOPLE], cc[HH_MAX_PEOPLE];
    int G;
};

__device__ int munkres_wave(MatchShared &S, int n, int lane)
{
    // step 1
    if (lane < n) {
        double mn = S.Cm[lane * MLD];
        for (int j = 1; j < n; ++j) { const double c = S.Cm[lane * MLD + j]; if (c < mn) mn = c; }
        for (int j = 0; j < n; ++j) S.Cm[lane * MLD + j] -= mn;
    }
    for (int i = lane; i < n * MLD; i += 64) S.marked[i] = 0;
    if (lane < n) { S.rc[lane] = 0; S.cc[lane] = 0; }
    __syncthreads();
    // step 2
    if (lane == 0) {
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j)
                if (S.Cm[i * MLD + j] == 0 && !S.cc[j] && !S.rc[i]) { S.marked[i * MLD + j] = 1; S.cc[j] = 1; S.rc[i] = 1; break; }
        for (int i = 0; i < n; ++i) { S.rc[i] = 0; S.cc[i] = 0; }
    }
    __syncthreads();
    int step = 3, z0r = 0, z0c = 0;
    for (int guard = 0; guard < 200000; ++guard) {
        if (step == 3) {
            bool star = false;
            if (lane < n)
                for (int i = 0; i < n; ++i) star |= (S.marked[i * MLD + lane] == 1);
            if (lane < n && star) S.cc[lane] = 1;
            const int count = __popcll(__ballot(star));
            __syncthreads();
            if (count >= n) return 0;
            step = 4;
        } else if (step == 4) {
            int row = 0, col = 0;
            for (;;) {
                int fr = -1, fc = -1;
                for (int t = 0; t < n; ++t) {
                    const int i = (row + t) % n;
                    if (S.rc[i]) continue;
                    const bool z = lane < n && S.Cm[i * MLD + lane] == 0 && !S.cc[lane];
                    const u64 mask = __ballot(z);
                    if (mask) {  // last uncovered zero in cyclic column order starting at `col`
                        const u64 low = mask & ((1ull << col) - 1ull);
                        fc = 63 - __builtin_clzll(low ? low : mask);
                        fr = i;
                        break;
                    }
                }
                if (fr < 0) { step = 6; break; }
                const u64 smask = __ballot(lane < n && S.marked[fr * MLD + lane] == 1);
                __syncthreads();
                if (lane == 0) S.marked[fr * MLD + fc] = 2;
                if (smask) {
                    const int sc = __builtin_ctzll(smask);
                    if (lane == 0) { S.rc[fr] = 1; S.cc[sc] = 0; }
                    row = fr; col = sc;
                    __syncthreads();
                } else {
                    z0r = fr; z0c = fc; step = 5;
                    __syncthreads();
                    break;
                }
            }
        } else if (step == 5) {
            if (lane == 0) {
                int count = 0;
                S.path[0] = z0r; S.path[1] = z0c;
                for (;;) {
                    int r = -1;
                    for (int i = 0; i < n; ++i) if (S.marked[i * MLD + S.path[count * 2 + 1]] == 1) { r = i; break; }
                    if (r < 0) break;
                    ++count; S.path[count * 2] = r; S.path[count * 2 + 1] = S.path[(count - 1) * 2 + 1];
                    int c = -1;
                    for (int j = 0; j < n; ++j) if (S.marked[S.path[count * 2] * MLD + j] == 2) { c = j; break; }
                    ++count; S.path[count * 2] = S.path[(count - 1) * 2]; S.path[count * 2 + 1] = c;
                }
                for (int i = 0; i <= count; ++i) {
                    unsigned char *m = &S.marked[S.path[i * 2] * MLD + S.path[i * 2 + 1]];
                    *m = (*m == 1) ? 0 : 1;
                }
                for (int i = 0; i < n; ++i) { S.rc[i] = 0; S.cc[i] = 0; }
            }
            __syncthreads();
            if (lane < n)
                for (int i = 0; i < n; ++i) if (S.marked[i * MLD + lane] == 2) S.marked[i * MLD + lane] = 0;
            __syncthreads();
            step = 3;
        } else {  // step 6
            double mn = 9223372036854775807.0;
            if (lane < n && !S.cc[lane])
                for (int i = 0; i < n; ++i)
                    if (!S.rc[i] && mn > S.Cm[i * MLD + lane]) mn = S.Cm[i * MLD + lane];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const double o = __shfl_xor(mn, off);
                mn = o < mn ? o : mn;
            }
            if (lane < n)
                for (int i = 0; i < n; ++i) {
                    double c = S.Cm[i * MLD + lane];
                    if (S.rc[i]) c += mn;
                    if (!S.cc[lane]) c -= mn;
                    S.Cm[i * MLD + lane] = c;
                }
            __syncthreads();
            step = 4;
        }
    }
    return 1;
}

__constant__ int c_joints_order[17] = {0, 1, 2, 3, 4, 5, 6, 11, 12, 7, 8, 9, 10, 13, 14, 15, 16};  // grouping.py:63-65

__global__ __launch_bounds__(64) void match_kernel(const float *__restrict__ tags_k, const int32_t *__restrict__ coords_k,
                                                   const float *__restrict__ scores_k, int K, int M, int E, double det_thr,
                                                   double tag_thr, float *__restrict__ joints, int32_t *__restrict__ num_people,
                                                   float *__restrict__ ws_tags, int32_t *__restrict__ status)
{
    __shared__ MatchShared S;
    const int b = blockIdx.x, lane = threadIdx.x, D = 3 + E;
    float *J = joints + (size_t)b * M * K * D;
    float *GT = ws_tags + (size_t)b * M * (K + 1) * E;  // per group: list of member tags
    tags_k += (size_t)b * K * M * E; coords_k += (size_t)b * K * M * 2; scores_k += (size_t)b * K * M;
    for (int i = lane; i < M * K * D; i += 64) J[i] = 0.f;
    if (lane == 0) S.G = 0;
    __syncthreads();
    int bad = 0;
    for (int it = 0; it < K; ++it) {
        const int idx = K == 17 ? c_joints_order[it] : it;
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
        __syncthreads();
        if (na == 0) continue;
        const int G = S.G;
        const bool first = (it == 0) || (G == 0);
        int ng = 0;
        if (!first) {
            ng = G < M ? G : M;
            if (lane < ng) np_mean_rows(GT + (size_t)lane * (K + 1) * E, S.gnt[lane], E, &S.gmean[lane * HH_MAX_EMB]);
            __syncthreads();
            const int n = na > ng ? na : ng;
            for (int i = lane; i < n * n; i += 64) {
                const int a = i / n, g = i % n;
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
            bad |= munkres_wave(S, n, lane);
            if (lane < na) {
                int col = -1;
                for (int j = 0; j < n; ++j) if (S.marked[lane * MLD + j] == 1) { col = j; break; }
                S.assign[lane] = col;
            }
            __syncthreads();
        }
        if (lane == 0) {  // dict semantics of grouping.py:104-143, candidates in row order
            int Gc = S.G;
            for (int a = 0; a < na; ++a) {
                int t;
                bool append = false;
                const int col = first ? -1 : S.assign[a];
                if (!first && col >= 0 && col < ng && S.saved[a * MLD + col] < tag_thr) { t = col; append = true; }
                else {
                    const float key = S.ctag[a * HH_MAX_EMB];
                    t = -1;
                    for (int q = 0; q < Gc; ++q) if (S.gkey[q] == key) { t = q; break; }
                    if (t < 0) {
                        if (Gc >= M) continue;  // groups past max_num_people are never matched nor returned
                        t = Gc++;
                        S.gkey[t] = key;
                    }
                    S.gnt[t] = 0;
                }
                float *jr = J + ((size_t)t * K + idx) * D;
                jr[0] = (float)S.cj[a * 3 + 0]; jr[1] = (float)S.cj[a * 3 + 1]; jr[2] = (float)S.cj[a * 3 + 2];
                for (int e = 0; e < E; ++e) {
                    jr[3 + e] = S.ctag[a * HH_MAX_EMB + e];
                    GT[((size_t)t * (K + 1) + S.gnt[t]) * E + e] = S.ctag[a * HH_MAX_EMB + e];
                }
                S.gnt[t] += 1;
                (void)append;
            }
            S.G = Gc;
        }
        __syncthreads();
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
        }
        num_people[b] = P;
        if (bad) atomicOr(status, 1);
    }
}

hipError_t launch_match(const float *tags_k, const int32_t *coords_k, const float *scores_k, int B, int K, int M, int E,
                        double det_thr, double tag_thr, float *joints, int32_t *num_people, float *ws_tags, int32_t *status,
                        hipStream_t s)
{
    hipLaunchKernelGGL(match_kernel, dim3(B), dim3(64), 0, s, tags_k, coords_k, scores_k, K, M, E, det_thr, tag_thr, joints,
                       num_people, ws_tags, status);
    return hipGetLastError();
}

// ------------------------------------------------------------------ adjust + person scores
// grouping.py:172-191 and :276 (scores = joints[..., 2].mean(1), taken BEFORE refine)
__global__ __launch_bounds__(256) void adjust_scores_kernel(const DecodeSrc src, int M, int adjust, float *__restrict__ joints,
                                                            const int32_t *__restrict__ num_people, float *__restrict__ scores)
{
    const int b = blockIdx.x, tid = threadIdx.x, K = src.K, D = 3 + src.E;
    const int P = num_people[b];
    float *J = joints + (size_t)b * M * K * D;
    if (adjust)
        for (int i = tid; i < P * K; i += 256) {
            float *j = J + (size_t)i * D;
            const int k = i % K;
            if (j[2] == 0.f) continue;
            float x = j[0], y = j[1];
            const int xi = (int)x, yi = (int)y;
            const int xr = min(xi + 1, src.W - 1), xl = max(xi - 1, 0), yd = min(yi + 1, src.H - 1), yu = max(yi - 1, 0);
            if (heat_at(src, b, k, yi, xr) > heat_at(src, b, k, yi, xl)) x += 0.25f; else x -= 0.25f;
            if (heat_at(src, b, k, yd, xi) > heat_at(src, b, k, yu, xi)) y += 0.25f; else y -= 0.25f;
            j[0] = x + 0.5f; j[1] = y + 0.5f;
        }
    __syncthreads();
    for (int p = tid; p < M; p += 256)
        scores[(size_t)b * M + p] = p < P ? __fdiv_rn(np_sum_f32(J + (size_t)p * K * D + 2, K, D), (float)K) : 0.f;
}

hipError_t launch_adjust_scores(const DecodeSrc &src, int M, int adjust, float *joints, const int32_t *num_people, float *scores,
                                hipStream_t s)
{
    hipLaunchKernelGGL(adjust_scores_kernel, dim3(src.B), dim3(256), 0, s, src, M, adjust, joints, num_people, scores);
    return hipGetLastError();
}

// ------------------------------------------------------------------ refine
// grouping.py:193-250.  (1) per person: mean tag of its detected joints.
__global__ __launch_bounds__(64) void refine_mean_kernel(const DecodeSrc src, int M, const float *__restrict__ joints,
                                                         const int32_t *__restrict__ num_people, float *__restrict__ ws_prev)
{
    const int b = blockIdx.x, p = threadIdx.x, K = src.K, E = src.E, D = 3 + E;
    if (p >= num_people[b] || p >= M) return;
    const float *J = joints + ((size_t)b * M + p) * K * D;
    float tl[64 * HH_MAX_EMB];
    int nt = 0;
    for (int k = 0; k < K; ++k)
        if (J[k * D + 2] > 0.f) {
            const int x = (int)J[k * D + 0], y = (int)J[k * D + 1];
            for (int e = 0; e < E; ++e) tl[nt * E + e] = tag_at(src, b, k, y, x, e);
            ++nt;
        }
    float *out = ws_prev + ((size_t)b * M + p) * (HH_MAX_EMB + 1);
    out[HH_MAX_EMB] = (float)nt;
    if (nt) np_mean_rows(tl, nt, E, out);
}

// (2) per (person, joint) with score == 0: argmax over the full map of hm - round(||tag - mean||)
__global__ __launch_bounds__(256) void refine_argmax_kernel(const DecodeSrc src, int M, float *__restrict__ joints,
                                                            const int32_t *__restrict__ num_people,
                                                            const float *__restrict__ ws_prev)
{
    __shared__ u64 wbest[4];
    const int k = blockIdx.x, p = blockIdx.y, b = blockIdx.z, tid = threadIdx.x, E = src.E, D = 3 + E;
    if (p >= num_people[b]) return;
    float *j = joints + (((size_t)b * M + p) * src.K + k) * D;
    if (!(j[2] == 0.f)) return;
    const float *prev = ws_prev + ((size_t)b * M + p) * (HH_MAX_EMB + 1);
    if (prev[HH_MAX_EMB] == 0.f) return;
    float mean[HH_MAX_EMB];
    for (int e = 0; e < E; ++e) mean[e] = prev[e];
    const int HW = src.H * src.W;
    u64 best = 0ull;
    for (int i = tid; i < HW; i += 256) {
        const int y = i / src.W, x = i % src.W;
        float s = 0.f;
        for (int e = 0; e < E; ++e) {
            float d = tag_at(src, b, k, y, x, e) - mean[e];
            d = d * d;
            s = e ? s + d : d;
        }
        const float v = heat_at(src, b, k, y, x) - rintf(__fsqrt_rn(s));
        const u64 key = make_key(v, (unsigned)i);
        best = key > best ? key : best;
    }
    const u64 wb = wave_max_u64(best);
    if ((tid & 63) == 0) wbest[tid >> 6] = wb;
    __syncthreads();
    if (tid == 0) {
        u64 g = wbest[0];
        for (int w = 1; w < 4; ++w) g = wbest[w] > g ? wbest[w] : g;
        const unsigned idx = 0xffffffffu - (unsigned)(g & 0xffffffffull);
        const int y = (int)(idx / (unsigned)src.W), x = (int)(idx % (unsigned)src.W);
        const float val = heat_at(src, b, k, y, x);
        if (val > 0.f) {
            double fx = (double)x + 0.5, fy = (double)y + 0.5;
            const int xr = min(x + 1, src.W - 1), xl = max(x - 1, 0), yd = min(y + 1, src.H - 1), yu = max(y - 1, 0);
            if (heat_at(src, b, k, y, xr) > heat_at(src, b, k, y, xl)) fx += 0.25; else fx -= 0.25;
            if (heat_at(src, b, k, yd, x) > heat_at(src, b, k, yu, x)) fy += 0.25; else fy -= 0.25;
            j[0] = (float)fx; j[1] = (float)fy; j[2] = val;
        }
    }
}

hipError_t launch_refine(const DecodeSrc &src, int M, float *joints, const int32_t *num_people, float *ws_prev, hipStream_t s)
{
    hipLaunchKernelGGL(refine_mean_kernel, dim3(src.B), dim3(64), 0, s, src, M, joints, num_people, ws_prev);
    hipLaunchKernelGGL(refine_argmax_kernel, dim3(src.K, M, src.B), dim3(256), 0, s, src, M, joints, num_people, ws_prev);
    return hipGetLastError();
}
