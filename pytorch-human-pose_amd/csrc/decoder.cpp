// Host side of the decode path: workspace + launch sequence behind hh_decode / hh_parse.
#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/hhrnet.h"
#include "decode_kernels.h"
#include "engine.h"

struct hh_decoder {
    int K, M;
    double det_thr, tag_thr;
    // reserved capacity
    int rB = 0, rH = 0, rW = 0, rE = 0;
    float *avg = nullptr, *cellmax = nullptr, *tagb = nullptr, *cand_val = nullptr, *tags_k = nullptr, *scores_k = nullptr, *ws_tags = nullptr, *ws_prev = nullptr;
    unsigned long long *cand_key = nullptr;
    int32_t *coords_k = nullptr, *flags = nullptr, *ws_jobs = nullptr;  // flags [rB]: HH_DECODE_* bits of the last call
    unsigned short *supmax = nullptr;  // per 8x8-cell super: heat bound / tag hulls (refine_bb_kernel)
    unsigned *suptag = nullptr;
    int rsup = 0;  // supers per map the two hold
    int *pk_ctr = nullptr;  // work counters of the peaks pass: zero between decode calls
    bool pk_fresh = false;  // just allocated: zeroed on the decode's own stream in front of its first use
    std::vector<void *> allocs;
    int lastB = 0, lastE = 0;
    int exact_topk = 0;   // 1: every tile is processed, so hh_decoder_read_topk returns the reference's full top_k
    int last_exact = 0;
    const int32_t *flags_last = nullptr;
    void release()
    {
        for (void *p : allocs) hipFree(p);
        allocs.clear();
        rB = rH = rW = rE = rsup = 0;
    }
    int reserve(int B, int H, int W, int E);
    int run(DecodeSrc &src, int adjust, int refine, float *joints, float *scores, int32_t *num_people, int32_t *flags_out, hipStream_t s);
};

static int nsup_of(int H, int W) { return ((H / 4 + 7) / 8) * ((W / 4 + 7) / 8); }
static int ntiles_of(int H, int W)
{
    return ((H + HH_NMS_TILE - 1) / HH_NMS_TILE) * ((W + HH_NMS_TILE - 1) / HH_NMS_TILE);
}

int hh_decoder::reserve(int B, int H, int W, int E)
{
    if (B <= rB && H * W <= rH * rW && ntiles_of(H, W) <= ntiles_of(rH, rW) && nsup_of(H, W) <= rsup && E <= rE) return 0;
    const int nB = std::max(B, rB), nH = std::max(H, rH), nW = std::max(W, rW), nE = std::max(E, rE);
    release();
    auto alloc = [&](size_t bytes, void **out) -> int {
        HH_CHECK_HIP(hipMalloc(out, bytes));
        allocs.push_back(*out);
        return 0;
    };
    const size_t nt = (size_t)ntiles_of(nH, nW);
    if (alloc((size_t)nB * K * (nH / 2) * (nW / 2) * 4, (void **)&avg)) return 1;
    if (alloc((size_t)nB * K * (nH / 4 + 1) * (nW / 4 + 1) * 4, (void **)&cellmax)) return 1;
    if (alloc((size_t)nB * K * (nH / 4 + 1) * (nW / 4 + 1) * nE * 2 * 4, (void **)&tagb)) return 1;
    if (alloc((size_t)nB * K * nt * M * 8, (void **)&cand_key)) return 1;
    if (alloc((size_t)nB * K * nt * M * 4, (void **)&cand_val)) return 1;
    if (alloc((size_t)nB * K * M * nE * 4, (void **)&tags_k)) return 1;
    if (alloc((size_t)nB * K * M * 2 * 4, (void **)&coords_k)) return 1;
    if (alloc((size_t)nB * K * M * 4, (void **)&scores_k)) return 1;
    if (alloc((size_t)nB * M * (K + 1) * nE * 4, (void **)&ws_tags)) return 1;
    if (alloc((size_t)nB * M * (HH_MAX_EMB + 1) * 4, (void **)&ws_prev)) return 1;
    if (alloc(((size_t)nB * M * K * 8 + 8) * 4, (void **)&ws_jobs)) return 1;  // 8 counters + 8 job queues (one per XCD)
    if (alloc((size_t)nB * 4, (void **)&flags)) return 1;
    HH_CHECK_HIP(hipMemset(flags, 0, (size_t)nB * 4));
    HH_CHECK_HIP(hipDeviceSynchronize());  // (a null-stream memset is not ordered in front of launches on a non-blocking stream)
    const int nsup = std::max(std::max(nsup_of(nH, nW), nsup_of(H, W)), rsup);
    if (alloc((size_t)nB * K * nsup * 2, (void **)&supmax)) return 1;
    if (alloc((size_t)nB * K * nsup * nE * 4, (void **)&suptag)) return 1;
    if (alloc(HH_PEAKS_PARTS * 4, (void **)&pk_ctr)) return 1;
    pk_fresh = true;  // (a hipMemset here runs on the null stream and may still be pending when a non-blocking stream starts the
                      // first launch: counters that are not zero make the persistent grid skip regions)
    rB = nB; rH = nH; rW = nW; rE = nE; rsup = nsup;
    return 0;
}

int hh_decoder::run(DecodeSrc &src, int adjust, int refine, float *joints, float *scores, int32_t *num_people, int32_t *flags_out, hipStream_t s)
{
    src.K = K;
    src.scale_h2 = (float)(src.H / 2) / (float)src.H; src.scale_w2 = (float)(src.W / 2) / (float)src.W;
    src.scale_h4 = (float)(src.H / 4) / (float)src.H; src.scale_w4 = (float)(src.W / 4) / (float)src.W;
    // Work that cannot produce a pixel above det_thr is skipped (only for det_thr >= 0: the empty candidate slots read as score 0,
    // which must fail `score > det_thr`); mode 1 and the exhaustive top-k (hh_decoder_set_exact_topk) process every 60x60 tile.
    const bool skip = !exact_topk && det_thr >= 0.0 && src.mode == 0;
    // the largest float <= det_thr: `bound <= thr_f` then implies `(double)score <= det_thr` for every pixel the bound covers
    float thr_f = (float)det_thr;
    if ((double)thr_f > det_thr) thr_f = nextafterf(thr_f, -INFINITY);
    if (skip) {
        // default path (round 4): one pass over the net's outputs forms the stage average in LDS, finds the peaks above det_thr and
        // leaves the cell bounds of the refine scans; the averaged map is not written (src.avg stays null: readers form its values)
        if (pk_fresh) {
            HH_CHECK_HIP(hipMemsetAsync(pk_ctr, 0, HH_PEAKS_PARTS * 4, s));
            pk_fresh = false;
        }
        HH_CHECK_HIP(launch_peaks(src, M, cand_key, cellmax, supmax, thr_f, pk_ctr, s));
        HH_CHECK_HIP(launch_topk_merge(src, M, peaks_regions(src.H, src.W), cand_key, nullptr, tags_k, coords_k, scores_k, pk_ctr, s));
    } else {
        if (src.mode == 0) {
            HH_CHECK_HIP(launch_stage_average(src.hm_q, src.hm_q_bs, src.hm_h, src.hm_h_bs, avg, src.B, K, src.H / 4, src.W / 4, s));
            src.avg = avg;
        }
        HH_CHECK_HIP(launch_nms_tile_topk(src, M, cand_key, cand_val, cellmax, -INFINITY, s));
        HH_CHECK_HIP(launch_topk_merge(src, M, ntiles_of(src.H, src.W), cand_key, cand_val, tags_k, coords_k, scores_k, nullptr, s));
    }
    last_exact = !skip;
    // (mode 0: the tag bounds of the refine scans and the cleared queue counters ride in the matching launch)
    const bool bounds = refine && src.mode == 0;
    if (refine && !bounds) HH_CHECK_HIP(hipMemsetAsync(ws_jobs, 0, 32, s));  // the 8 queue counters
    HH_CHECK_HIP(launch_match(tags_k, coords_k, scores_k, src.B, K, M, src.E, det_thr, tag_thr, joints, num_people, ws_tags, flags_out ? flags_out : flags,
                              bounds ? &src : nullptr, tagb, skip ? suptag : nullptr, ws_jobs, s));
    flags_last = flags_out ? flags_out : flags;
    if (skip) HH_CHECK_HIP(launch_fallback_top1(src, M, flags_last, joints, s));
    HH_CHECK_HIP(launch_adjust_scores(src, M, adjust, refine, joints, num_people, scores, ws_prev, ws_jobs, s));
    if (refine && skip) HH_CHECK_HIP(launch_refine_bb(src, M, joints, ws_prev, ws_jobs, cellmax, tagb, supmax, suptag, s));
    else if (refine) HH_CHECK_HIP(launch_refine(src, M, joints, ws_prev, ws_jobs, cellmax, tagb, s));
    lastB = src.B; lastE = src.E;
    return 0;
}

extern "C" {

hh_decoder *hh_decoder_create(int num_kpts, int max_people, double det_thr, double tag_thr)
{
    if (num_kpts <= 0 || num_kpts > 64 || max_people <= 0 || max_people > HH_MAX_PEOPLE) {
        hh_set_error("hh_decoder_create: need 0 < num_kpts <= 64 and 0 < max_people <= 32");
        return nullptr;
    }
    hh_decoder *d = new hh_decoder();
    d->K = num_kpts; d->M = max_people; d->det_thr = det_thr; d->tag_thr = tag_thr;
    return d;
}
void hh_decoder_destroy(hh_decoder *dec)
{
    if (!dec) return;
    dec->release();
    delete dec;
}
int hh_decoder_reserve(hh_decoder *dec, int B, int H, int W, int E) { return dec->reserve(B, H, W, E); }
int hh_decoder_set_exact_topk(hh_decoder *dec, int enable) { dec->exact_topk = enable != 0; return 0; }

int hh_decode(hh_decoder *dec, const float *hm_q, int64_t hm_q_bstride, const float *hm_h, int64_t hm_h_bstride,
              const float *const *tags_q, const int64_t *tags_bstride, int E, int B, int hq, int wq, int adjust, int refine,
              float *joints, float *scores, int32_t *num_people, int32_t *flags, void *stream)
{
    if (E < 1 || E > HH_MAX_EMB) { hh_set_error("hh_decode: 1 <= E <= 4"); return 1; }
    if (B <= 0 || hq <= 0 || wq <= 0 || (size_t)hq * wq * 16 >= (1u << 24)) { hh_set_error("hh_decode: bad shape (need H*W < 2^24)"); return 1; }
    const int H = 4 * hq, W = 4 * wq;
    if (dec->reserve(B, H, W, E)) return 1;
    hipStream_t s = (hipStream_t)stream;
    DecodeSrc src{};
    src.mode = 0; src.avg = nullptr; src.hm_q = hm_q; src.hm_q_bs = hm_q_bstride; src.hm_h = hm_h; src.hm_h_bs = hm_h_bstride;
    src.B = B; src.H = H; src.W = W; src.E = E;
    for (int e = 0; e < E; ++e) { src.tags_q[e] = tags_q[e]; src.tags_bs[e] = tags_bstride[e]; }
    return dec->run(src, adjust, refine, joints, scores, num_people, flags, s);
}

int hh_parse(hh_decoder *dec, const float *hm_full, const float *tags_full, int E, int B, int H, int W, int adjust, int refine,
             float *joints, float *scores, int32_t *num_people, int32_t *flags, void *stream)
{
    if (E < 1 || E > HH_MAX_EMB) { hh_set_error("hh_parse: 1 <= E <= 4"); return 1; }
    if (B <= 0 || H <= 0 || W <= 0 || (size_t)H * W >= (1u << 24)) { hh_set_error("hh_parse: bad shape (need H*W < 2^24)"); return 1; }
    if (dec->reserve(B, H, W, E)) return 1;
    DecodeSrc src{};
    src.mode = 1; src.hm_full = hm_full; src.tags_full = tags_full; src.B = B; src.H = H; src.W = W; src.E = E;
    return dec->run(src, adjust, refine, joints, scores, num_people, flags, (hipStream_t)stream);
}

int hh_resize_accumulate(const float *src, int64_t src_bstride, int B, int K, int h, int w, float *dst, int64_t dst_bstride, int H,
                         int W, float weight, int init, void *stream)
{
    if (!src || !dst || B <= 0 || K <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0) { hh_set_error("hh_resize_accumulate: bad argument"); return 1; }
    HH_CHECK_HIP(launch_resize_accumulate(src, src_bstride, B, K, h, w, dst, dst_bstride, H, W, weight, init, (hipStream_t)stream));
    return 0;
}

int hh_debug_munkres(const double *cost, int n, int32_t *star)
{
    if (!cost || !star || n <= 0 || n > HH_MAX_PEOPLE) { hh_set_error("hh_debug_munkres: need 0 < n <= 32"); return 1; }
    double *d_cost = nullptr;
    int32_t *d_out = nullptr;
    HH_CHECK_HIP(hipMalloc((void **)&d_cost, (size_t)n * n * 8));
    if (hipMalloc((void **)&d_out, (size_t)(n + 1) * 4) != hipSuccess) { hipFree(d_cost); hh_set_error("hh_debug_munkres: hipMalloc"); return 1; }
    std::vector<int32_t> out(n + 1, -2);
    hipError_t e = hipMemcpy(d_cost, cost, (size_t)n * n * 8, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = launch_munkres_debug(d_cost, n, d_out, nullptr);
    if (e == hipSuccess) e = hipMemcpy(out.data(), d_out, (size_t)(n + 1) * 4, hipMemcpyDeviceToHost);
    hipFree(d_cost); hipFree(d_out);
    HH_CHECK_HIP(e);
    if (out[n]) { hh_set_error("hh_debug_munkres: the solver hit its iteration guard"); return 1; }
    for (int i = 0; i < n; ++i) star[i] = out[i];
    return 0;
}

int hh_decoder_read_topk(hh_decoder *dec, float *tags_k, int32_t *coords_k, float *scores_k)
{
    if (!dec->lastB) { hh_set_error("hh_decoder_read_topk: nothing decoded yet"); return 1; }
    if (!dec->last_exact) { hh_set_error("hh_decoder_read_topk: the last call skipped sub-threshold tiles; hh_decoder_set_exact_topk(dec, 1) first"); return 1; }
    HH_CHECK_HIP(hipDeviceSynchronize());
    const size_t n = (size_t)dec->lastB * dec->K * dec->M;
    HH_CHECK_HIP(hipMemcpy(tags_k, dec->tags_k, n * dec->lastE * 4, hipMemcpyDeviceToHost));
    HH_CHECK_HIP(hipMemcpy(coords_k, dec->coords_k, n * 2 * 4, hipMemcpyDeviceToHost));
    HH_CHECK_HIP(hipMemcpy(scores_k, dec->scores_k, n * 4, hipMemcpyDeviceToHost));
    std::vector<int32_t> fl(dec->lastB);
    HH_CHECK_HIP(hipMemcpy(fl.data(), dec->flags_last, fl.size() * 4, hipMemcpyDeviceToHost));
    for (int32_t f : fl)
        if (f & HH_DECODE_SOLVER_GUARD) { hh_set_error("decode: assignment solver hit its iteration guard"); return 1; }
    return 0;
}

}  // extern "C"
