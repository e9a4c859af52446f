/*
 * oracle/decode_oracle.c -- CPU restatement (plain C) of the reference's heatmap
 * aggregation + associative-embedding decode.  TEST INFRASTRUCTURE ONLY: imported by
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as the checker; the
 * product path (pytorch-human-pose_amd/) never links or calls it.
 *
 * Parity status: PINNED by tests/golden/{decode,munkres}.npz, which tools/make_golden.py
 * produced by importing the reference (src/keypoints/grouping.py) and munkres 1.1.4 in
 * the build container.  The reference has no tests of its own.
 *
 * Every function cites the reference lines it follows (paths relative to
 * /root/reference).  Third-party arithmetic restated here:
 *   - torch.nn.functional.interpolate(mode="bilinear", align_corners=False), CPU fp32
 *     path of torch 2.10 (ATen UpSampleKernel.cpp generic N-d kernel).  Determined
 *     experimentally in the build container to be, per output pixel,
 *         src = fmaf(scale, dst + 0.5f, -0.5f) clamped at 0;  w1 = src - floor(src); w0 = 1 - w1
 *         row(r) = fmaf(in[r][x0], wx0, in[r][x1] * wx1)
 *         out    = fmaf(row(y0), wy0, row(y1) * wy1)
 *     bit-for-bit (0 mismatches over 1.1 M outputs per case).
 *   - torch.topk: tie order between equal values is unspecified by torch (the CPU path is a
 *     heap-based partial sort, the CUDA path a radix select); this restatement orders ties
 *     by ascending flat index.
 *   - munkres 1.1.4 (pure Python, pinned in the reference's poetry.lock:1359-1367).
 *   - numpy float32 reductions: np.mean over a [n,1] stack is numpy's pairwise sum
 *     (8 partial sums, n >= 8), over [n,E>=2] it is a sequential row-by-row sum
 *     (verified against numpy 2.2.6 in the build container).
 *
 * Build: see oracle/Makefile  (gcc -O2 -ffp-contract=off -shared -fPIC).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_MAXK 64

/* ---------------------------------------------------------------- bilinear resize */
/* results.py:48-67 -> F.interpolate(..., mode="bilinear", align_corners=False) */
static void src_index(int in_size, int out_size, int dst, int *i0, int *i1, float *w0, float *w1)
{
    float scale = (float)in_size / (float)out_size;
    float r = fmaf(scale, (float)dst + 0.5f, -0.5f);
    if (r < 0.f) r = 0.f;
    int a = (int)r;
    int off = (a < in_size - 1) ? 1 : 0;
    float l1 = r - (float)a;
    if (l1 < 0.f) l1 = 0.f;
    if (l1 > 1.f) l1 = 1.f;
    *i0 = a; *i1 = a + off; *w1 = l1; *w0 = 1.f - l1;
}

void orc_bilinear(const float *in, int C, int h, int w, float *out, int H, int W, int out_cstride, int out_pstride)
{
    /* out element (c,y,x) is written at out[c*out_cstride + (y*W+x)*out_pstride] so the same
       routine can fill the [K,H,W,E] tag tensor of results.py:233 (pstride=E). */
    int *x0 = malloc(sizeof(int) * W * 2), *x1 = x0 + W;
    float *wx0 = malloc(sizeof(float) * W * 2), *wx1 = wx0 + W;
    for (int x = 0; x < W; ++x) src_index(w, W, x, &x0[x], &x1[x], &wx0[x], &wx1[x]);
    for (int c = 0; c < C; ++c) {
        const float *ic = in + (size_t)c * h * w;
        for (int y = 0; y < H; ++y) {
            int y0, y1; float wy0, wy1;
            src_index(h, H, y, &y0, &y1, &wy0, &wy1);
            const float *r0 = ic + (size_t)y0 * w, *r1 = ic + (size_t)y1 * w;
            float *o = out + (size_t)c * out_cstride + (size_t)y * W * out_pstride;
            for (int x = 0; x < W; ++x) {
                float a = fmaf(r0[x0[x]], wx0[x], r0[x1[x]] * wx1[x]);
                float b = fmaf(r1[x0[x]], wx0[x], r1[x1[x]] * wx1[x]);
                o[(size_t)x * out_pstride] = fmaf(a, wy0, b * wy1);
            }
        }
    }
    free(x0); free(wx0);
}

/* results.py:225-234: match_heatmaps_size (1/4 -> 1/2), stack+mean over the two stages,
   resize to the model-input size; tags of every TTA pass resized and stacked on a last dim. */
void orc_aggregate(const float *hm_q, const float *hm_h, const float *const *tags_q, int E, int K, int hq, int wq,
                   float *hm_full, float *tags_full)
{
    int hh = 2 * hq, wh = 2 * wq, H = 4 * hq, W = 4 * wq;
    float *avg = malloc(sizeof(float) * (size_t)K * hh * wh);
    orc_bilinear(hm_q, K, hq, wq, avg, hh, wh, hh * wh, 1);
    for (size_t i = 0; i < (size_t)K * hh * wh; ++i) avg[i] = (avg[i] + hm_h[i]) / 2.0f; /* stack().mean(0) */
    orc_bilinear(avg, K, hh, wh, hm_full, H, W, H * W, 1);
    for (int e = 0; e < E; ++e) orc_bilinear(tags_q[e], K, hq, wq, tags_full + e, H, W, H * W * E, E);
    free(avg);
}

/* ---------------------------------------------------------------- NMS + top-k */
/* grouping.py:74,80-83 (nms) and :147-170 (top_k) */
void orc_nms(const float *hm, int K, int H, int W, float *out)
{
    for (int k = 0; k < K; ++k) {
        const float *m = hm + (size_t)k * H * W;
        float *o = out + (size_t)k * H * W;
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                float best = -INFINITY;
                for (int dy = -2; dy <= 2; ++dy) {
                    int yy = y + dy; if (yy < 0 || yy >= H) continue;
                    for (int dx = -2; dx <= 2; ++dx) {
                        int xx = x + dx; if (xx < 0 || xx >= W) continue;
                        float v = m[(size_t)yy * W + xx]; if (v > best) best = v;
                    }
                }
                float v = m[(size_t)y * W + x];
                o[(size_t)y * W + x] = v * ((best == v) ? 1.0f : 0.0f);
            }
    }
}

void orc_topk(const float *nms, const float *tags_full, int K, int H, int W, int E, int maxp,
              float *tags_k, int32_t *coords_k, float *scores_k)
{
    size_t n = (size_t)H * W;
    for (int k = 0; k < K; ++k) {
        const float *m = nms + (size_t)k * n;
        float *bv = scores_k + (size_t)k * maxp;
        int64_t bi[1024];
        int cnt = 0;
        for (size_t i = 0; i < n; ++i) { /* insertion into a sorted list: value desc, index asc */
            float v = m[i];
            if (cnt == maxp && !(v > bv[cnt - 1])) continue;
            int p = cnt < maxp ? cnt : maxp - 1;
            while (p > 0 && v > bv[p - 1]) { bv[p] = bv[p - 1]; bi[p] = bi[p - 1]; --p; }
            bv[p] = v; bi[p] = (int64_t)i;
            if (cnt < maxp) ++cnt;
        }
        for (int j = 0; j < maxp; ++j) {
            int64_t idx = bi[j];
            coords_k[((size_t)k * maxp + j) * 2 + 0] = (int32_t)(idx % W);
            coords_k[((size_t)k * maxp + j) * 2 + 1] = (int32_t)(long)((float)idx / (float)W);
            for (int e = 0; e < E; ++e)
                tags_k[((size_t)k * maxp + j) * E + e] = tags_full[((size_t)k * n + (size_t)idx) * E + e];
        }
    }
}

/* ---------------------------------------------------------------- munkres 1.1.4 */
/* munkres.py:114-171 (compute) and the six steps :184-340, literal control flow. */
int orc_munkres(const double *cost, int rows, int cols, int32_t *pairs)
{
    int n = rows > cols ? rows : cols;
    double *C = calloc((size_t)n * n, sizeof(double)); /* pad_matrix: zeros */
    char *marked = calloc((size_t)n * n, 1), *rc = calloc(n, 1), *cc = calloc(n, 1);
    int *path = calloc((size_t)4 * n + 4, sizeof(int));
    for (int i = 0; i < rows; ++i) for (int j = 0; j < cols; ++j) C[i * n + j] = cost[i * cols + j];
    int z0r = 0, z0c = 0, step = 1, guard = 0;
    while (step != 7 && ++guard < 1000000) {
        if (step == 1) {
            for (int i = 0; i < n; ++i) {
                double mn = C[i * n];
                for (int j = 1; j < n; ++j) if (C[i * n + j] < mn) mn = C[i * n + j];
                for (int j = 0; j < n; ++j) C[i * n + j] -= mn;
            }
            step = 2;
        } else if (step == 2) {
            for (int i = 0; i < n; ++i)
                for (int j = 0; j < n; ++j)
                    if (C[i * n + j] == 0 && !cc[j] && !rc[i]) { marked[i * n + j] = 1; cc[j] = 1; rc[i] = 1; break; }
            memset(rc, 0, n); memset(cc, 0, n);
            step = 3;
        } else if (step == 3) {
            int count = 0;
            for (int i = 0; i < n; ++i)
                for (int j = 0; j < n; ++j)
                    if (marked[i * n + j] == 1 && !cc[j]) { cc[j] = 1; ++count; }
            step = count >= n ? 7 : 4;
        } else if (step == 4) {
            int row = 0, col = 0, done = 0;
            while (!done) {
                /* __find_a_zero(row, col): cyclic scan; within the first row holding an uncovered
                   zero the LAST one in cyclic column order wins (no break in the inner loop). */
                int fr = -1, fc = -1, i = row, fin = 0;
                while (!fin) {
                    int j = col;
                    for (;;) {
                        if (C[i * n + j] == 0 && !rc[i] && !cc[j]) { fr = i; fc = j; fin = 1; }
                        j = (j + 1) % n;
                        if (j == col) break;
                    }
                    i = (i + 1) % n;
                    if (i == row) fin = 1;
                }
                row = fr; col = fc;
                if (row < 0) { done = 1; step = 6; }
                else {
                    marked[row * n + col] = 2;
                    int sc = -1;
                    for (int j = 0; j < n; ++j) if (marked[row * n + j] == 1) { sc = j; break; }
                    if (sc >= 0) { col = sc; rc[row] = 1; cc[col] = 0; }
                    else { done = 1; z0r = row; z0c = col; step = 5; }
                }
            }
        } else if (step == 5) {
            int count = 0, done = 0;
            path[0] = z0r; path[1] = z0c;
            while (!done) {
                int r = -1;
                for (int i = 0; i < n; ++i) if (marked[i * n + path[count * 2 + 1]] == 1) { r = i; break; }
                if (r >= 0) { ++count; path[count * 2] = r; path[count * 2 + 1] = path[(count - 1) * 2 + 1]; }
                else done = 1;
                if (!done) {
                    int c = -1;
                    for (int j = 0; j < n; ++j) if (marked[path[count * 2] * n + j] == 2) { c = j; break; }
                    ++count; path[count * 2] = path[(count - 1) * 2]; path[count * 2 + 1] = c;
                }
            }
            for (int i = 0; i <= count; ++i) {
                char *m = &marked[path[i * 2] * n + path[i * 2 + 1]];
                *m = (*m == 1) ? 0 : 1;
            }
            memset(rc, 0, n); memset(cc, 0, n);
            for (int i = 0; i < n * n; ++i) if (marked[i] == 2) marked[i] = 0;
            step = 3;
        } else { /* step 6 */
            double mn = 9223372036854775807.0; /* sys.maxsize */
            for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j)
                if (!rc[i] && !cc[j] && mn > C[i * n + j]) mn = C[i * n + j];
            for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) {
                if (rc[i]) C[i * n + j] += mn;
                if (!cc[j]) C[i * n + j] -= mn;
            }
            step = 4;
        }
    }
    int np = 0;
    for (int i = 0; i < rows; ++i) for (int j = 0; j < cols; ++j)
        if (marked[i * n + j] == 1) { pairs[np * 2] = i; pairs[np * 2 + 1] = j; ++np; }
    free(C); free(marked); free(rc); free(cc); free(path);
    return step == 7 ? np : -1;
}

/* ---------------------------------------------------------------- numpy float32 mean */
static float np_sum_f32(const float *v, int n, int stride)
{
    if (n < 8) { float s = v[0]; for (int i = 1; i < n; ++i) s += v[i * stride]; return s; }
    float r[8];
    for (int k = 0; k < 8; ++k) r[k] = v[k * stride];
    int i = 8;
    for (; i + 8 <= n; i += 8) for (int k = 0; k < 8; ++k) r[k] += v[(i + k) * stride];
    float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += v[i * stride];
    return res;
}

/* np.mean(list_of_[E]_float32, axis=0): E == 1 pairwise, E >= 2 sequential over rows */
static void np_mean_rows(const float *rows, int n, int E, float *out)
{
    if (E == 1) { out[0] = np_sum_f32(rows, n, 1) / (float)n; return; }
    for (int e = 0; e < E; ++e) {
        float s = rows[e];
        for (int i = 1; i < n; ++i) s += rows[i * E + e];
        out[e] = s / (float)n;
    }
}

/* ---------------------------------------------------------------- match_by_tag */
static const int JOINTS_ORDER[17] = {0, 1, 2, 3, 4, 5, 6, 11, 12, 7, 8, 9, 10, 13, 14, 15, 16}; /* grouping.py:63-65 */

typedef struct {
    float key;
    double *joints;   /* [K][3+E] */
    float *tags;      /* [ntags][E] */
    int ntags;
} group_t;

/* grouping.py:85-145. Output grouped [min(G,maxp)][K][3+E] float32; returns number of rows. */
int orc_match_by_tag(const float *tags_k, const int32_t *coords_k, const float *scores_k, int K, int maxp, int E,
                     double det_thr, double tag_thr, float *grouped)
{
    int D = 3 + E, cap = K * maxp + 1, G = 0;
    group_t *g = calloc(cap, sizeof(group_t));
    double *joints = malloc(sizeof(double) * maxp * D);
    float *tags = malloc(sizeof(float) * maxp * E);
    double *cost = malloc(sizeof(double) * maxp * maxp * 2), *saved = malloc(sizeof(double) * maxp * maxp);
    int32_t *pairs = malloc(sizeof(int32_t) * 2 * maxp * 2);
    for (int it = 0; it < K; ++it) {
        int idx = K == 17 ? JOINTS_ORDER[it] : it;
        int na = 0;
        for (int c = 0; c < maxp; ++c) { /* mask = score > det_thr (joints is float64) */
            float s = scores_k[idx * maxp + c];
            if (!((double)s > det_thr)) continue;
            joints[na * D + 0] = coords_k[(idx * maxp + c) * 2 + 0];
            joints[na * D + 1] = coords_k[(idx * maxp + c) * 2 + 1];
            joints[na * D + 2] = s;
            for (int e = 0; e < E; ++e) { tags[na * E + e] = tags_k[(idx * maxp + c) * E + e]; joints[na * D + 3 + e] = tags[na * E + e]; }
            ++na;
        }
        if (na == 0) continue;
        int first = (it == 0 || G == 0);
        int ng = 0;
        char matched_row[1024]; int match_col[1024];
        memset(matched_row, 0, sizeof matched_row);
        if (!first) {
            ng = G < maxp ? G : maxp;
            /* diff[a][b] = ||tag_a - mean(tags of group b)||_2 in float64 (np.linalg.norm) */
            for (int b = 0; b < ng; ++b) {
                float mean[ORC_MAXK];
                np_mean_rows(g[b].tags, g[b].ntags, E, mean);
                for (int a = 0; a < na; ++a) {
                    double s = 0;
                    for (int e = 0; e < E; ++e) { double d = joints[a * D + 3 + e] - (double)mean[e]; s += d * d; }
                    saved[a * ng + b] = sqrt(s);
                }
            }
            int cols = na > ng ? na : ng;
            for (int a = 0; a < na; ++a)
                for (int b = 0; b < cols; ++b)
                    cost[a * cols + b] = b < ng ? nearbyint(saved[a * ng + b]) * 100 - joints[a * D + 2] : 1e10;
            int np = orc_munkres(cost, na, cols, pairs);
            for (int p = 0; p < np; ++p) { matched_row[pairs[2 * p]] = 1; match_col[pairs[2 * p]] = pairs[2 * p + 1]; }
        }
        /* pairs come out row-major, so visiting rows in order reproduces the reference loop;
           in the seeding pass every candidate starts (or overwrites) its own group. */
        for (int a = 0; a < na; ++a) {
            if (!first && !matched_row[a]) continue; /* cannot happen: every row is assigned */
            int col = first ? -1 : match_col[a];
            if (!first && col < ng && saved[a * ng + col] < tag_thr) {
                memcpy(g[col].joints + idx * D, joints + a * D, sizeof(double) * D);
                memcpy(g[col].tags + g[col].ntags * E, tags + a * E, sizeof(float) * E);
                g[col].ntags++;
            } else {
                float key = tags[a * E];
                int t = -1;
                for (int q = 0; q < G; ++q) if (g[q].key == key) { t = q; break; } /* dict lookup, float == */
                if (t < 0) {
                    t = G++;
                    g[t].key = key;
                    g[t].joints = calloc(K * D, sizeof(double));
                    g[t].tags = malloc(sizeof(float) * (K + 1) * E * 2);
                }
                memcpy(g[t].joints + idx * D, joints + a * D, sizeof(double) * D);
                memcpy(g[t].tags, tags + a * E, sizeof(float) * E);
                g[t].ntags = 1;
            }
        }
    }
    int P = G < maxp ? G : maxp;
    for (int p = 0; p < P; ++p) for (int i = 0; i < K * D; ++i) grouped[p * K * D + i] = (float)g[p].joints[i];
    for (int q = 0; q < G; ++q) { free(g[q].joints); free(g[q].tags); }
    free(g); free(joints); free(tags); free(cost); free(saved); free(pairs);
    return P;
}

/* ---------------------------------------------------------------- adjust / refine */
static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

/* grouping.py:172-191 (the reference's x/y names are swapped; here x is the column) */
void orc_adjust(float *grouped, int P, int K, int E, const float *hm_full, int H, int W)
{
    int D = 3 + E;
    for (int p = 0; p < P; ++p)
        for (int k = 0; k < K; ++k) {
            float *j = grouped + ((size_t)p * K + k) * D;
            if (j[2] == 0) continue;
            float x = j[0], y = j[1];
            int xi = (int)x, yi = (int)y;
            const float *m = hm_full + (size_t)k * H * W;
            if (m[(size_t)yi * W + imin(xi + 1, W - 1)] > m[(size_t)yi * W + imax(xi - 1, 0)]) x += 0.25f; else x -= 0.25f;
            if (m[(size_t)imin(yi + 1, H - 1) * W + xi] > m[(size_t)imax(0, yi - 1) * W + xi]) y += 0.25f; else y -= 0.25f;
            j[0] = x + 0.5f; j[1] = y + 0.5f;
        }
}

/* grouping.py:193-250 for one person */
void orc_refine_person(float *pj, int K, int E, const float *hm_full, const float *tags_full, int H, int W)
{
    int D = 3 + E;
    float tl[ORC_MAXK * 8], mean[8];
    int nt = 0;
    for (int k = 0; k < K; ++k)
        if (pj[k * D + 2] > 0) {
            int x = (int)pj[k * D + 0], y = (int)pj[k * D + 1];
            for (int e = 0; e < E; ++e) tl[nt * E + e] = tags_full[(((size_t)k * H + y) * W + x) * E + e];
            ++nt;
        }
    if (nt == 0) return; /* np.mean([]) is nan in the reference; unreachable from parse() */
    np_mean_rows(tl, nt, E, mean);
    for (int k = 0; k < K; ++k) {
        const float *m = hm_full + (size_t)k * H * W;
        const float *t = tags_full + (size_t)k * H * W * E;
        float best = -INFINITY; size_t bi = 0;
        for (size_t i = 0; i < (size_t)H * W; ++i) {
            float s = 0;
            for (int e = 0; e < E; ++e) { float d = t[i * E + e] - mean[e]; d = d * d; s = e ? s + d : d; }
            float v = m[i] - nearbyintf(sqrtf(s));
            if (v > best) { best = v; bi = i; } /* np.argmax: first maximum */
        }
        if (!(pj[k * D + 2] == 0)) continue;
        int y = (int)(bi / W), x = (int)(bi % W);
        float val = m[bi];
        if (!(val > 0)) continue;
        double fx = x + 0.5, fy = y + 0.5;
        if (m[(size_t)y * W + imin(x + 1, W - 1)] > m[(size_t)y * W + imax(x - 1, 0)]) fx += 0.25; else fx -= 0.25;
        if (m[(size_t)imin(y + 1, H - 1) * W + x] > m[(size_t)imax(0, y - 1) * W + x]) fy += 0.25; else fy -= 0.25;
        pj[k * D + 0] = (float)fx; pj[k * D + 1] = (float)fy; pj[k * D + 2] = val;
    }
}

/* grouping.py:252-283. joints [maxp][K][3+E], scores [maxp]; returns P (>= 1), or -1 when no group was formed and the
   result is the one pseudo-person of grouping.py:262-269 (the reference's arrays are float64 there: its concatenate mixes
   int32 coordinates with float32 scores, and the score becomes the double 0.01 -- oracle/decode.py widens accordingly). */
int orc_parse(const float *hm_full, const float *tags_full, int K, int H, int W, int E, int maxp, double det_thr,
              double tag_thr, int adjust, int refine, float *joints, float *scores,
              float *tags_k_out, int32_t *coords_k_out, float *scores_k_out)
{
    int D = 3 + E;
    float *nms = malloc(sizeof(float) * (size_t)K * H * W);
    float *tags_k = tags_k_out ? tags_k_out : malloc(sizeof(float) * K * maxp * E);
    int32_t *coords_k = coords_k_out ? coords_k_out : malloc(sizeof(int32_t) * K * maxp * 2);
    float *scores_k = scores_k_out ? scores_k_out : malloc(sizeof(float) * K * maxp);
    orc_nms(hm_full, K, H, W, nms);
    orc_topk(nms, tags_full, K, H, W, E, maxp, tags_k, coords_k, scores_k);
    free(nms);
    int P = orc_match_by_tag(tags_k, coords_k, scores_k, K, maxp, E, det_thr, tag_thr, joints);
    const int fallback = P == 0;
    if (P == 0) { /* grouping.py:262-269: best candidate per joint, score forced to 0.01 */
        for (int k = 0; k < K; ++k) {
            float *j = joints + k * D;
            j[0] = (float)coords_k[(k * maxp) * 2 + 0];
            j[1] = (float)coords_k[(k * maxp) * 2 + 1];
            j[2] = 0.01f;
            for (int e = 0; e < E; ++e) { float t = tags_k[(k * maxp) * E + e]; j[3 + e] = isnan(t) ? 0.f : t; }
        }
        P = 1;
    }
    if (adjust) orc_adjust(joints, P, K, E, hm_full, H, W);
    for (int p = 0; p < P; ++p) { /* person_scores = joints[..., 2].mean(1) BEFORE refine (grouping.py:276) */
        scores[p] = np_sum_f32(joints + (size_t)p * K * D + 2, K, D) / (float)K;
    }
    if (refine) for (int p = 0; p < P; ++p) orc_refine_person(joints + (size_t)p * K * D, K, E, hm_full, tags_full, H, W);
    if (!tags_k_out) free(tags_k);
    if (!coords_k_out) free(coords_k);
    if (!scores_k_out) free(scores_k);
    return fallback ? -P : P;
}

