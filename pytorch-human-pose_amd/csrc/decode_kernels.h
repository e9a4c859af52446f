// Launch interface of the decode kernels (heatmap aggregation, NMS/top-k, tag grouping,
// adjust, refine).  All index/float results are bit-exact restatements of
// /root/reference/src/keypoints/{results.py:225-234, grouping.py:80-283}.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define HH_MAX_EMB 4
#define HH_MAX_PEOPLE 32  // one wave column per candidate / group in the matching kernel
#define HH_DECODE_FALLBACK 1      // flags[b]: no group was formed, the one person is the pseudo-person of grouping.py:262-269
#define HH_DECODE_SOLVER_GUARD 2  // flags[b]: the assignment solver hit its iteration guard (result invalid)
#define HH_NMS_TILE 60  // + the 2-pixel halo of the 5x5 maximum = 64 = one wave of columns

// Where the full-resolution maps come from.
//  mode 0: computed on the fly -- heat = bilinear x2 of the stage average of the two heatmap outputs, tags = bilinear x4 of the
//          1/4-res tag maps; nothing full-res is ever stored in HBM.  The stage average itself (1/2-res) is materialised in
//          `avg` only on the exhaustive path (hh_decoder_set_exact_topk); on the default path `avg` is null and every reader
//          forms the average values it needs from hm_q / hm_h (round 4: the map was written once and read back three times).
//  mode 1: explicit full-res arrays (the MPPEHeatmapParser.parse boundary).
struct DecodeSrc {
    int mode;
    const float *avg;                  // [B,K,H/2,W/2] or null
    const float *hm_q, *hm_h;          // mode 0: the net's heatmap outputs [B,K,H/4,W/4] / [B,K,H/2,W/2], batch strides hm_*_bs
    int64_t hm_q_bs, hm_h_bs;
    const float *tags_q[HH_MAX_EMB];   // each [B,K,H/4,W/4], batch stride tags_bs[e]
    int64_t tags_bs[HH_MAX_EMB];
    const float *hm_full;              // [B,K,H,W]
    const float *tags_full;            // [B,K,H,W,E]
    int B, K, H, W, E;
    // torch's area_pixel_compute_scale: (float)in / (float)out, evaluated once on the host (IEEE division)
    float scale_h2, scale_w2;  // half-res -> full-res
    float scale_h4, scale_w4;  // quarter-res -> full-res
};

hipError_t launch_stage_average(const float *hm_q, int64_t hm_q_bs, const float *hm_h, int64_t hm_h_bs, float *avg, int B,
                                int K, int hq, int wq, hipStream_t s);
// per (b,k,tile): top-M candidates of the NMS'ed map as sortable keys + exact values
// skip_thr: tiles whose values cannot exceed it emit no candidates (-INFINITY: every tile is processed, the exact top-k)
hipError_t launch_nms_tile_topk(const DecodeSrc &src, int M, unsigned long long *cand_key, float *cand_val, float *cellmax,
                                float skip_thr, hipStream_t s);
// the default front end of hh_decode (mode 0, src.avg == null, det_thr >= 0): stage average + x2 resize + 5x5 NMS + candidates above
// `thr` in one pass over src.hm_q / src.hm_h (decode_peaks.hip).  cand_key [B*K][peaks_regions(H, W)][M] (0 = empty slot), cellmax =
// the bf16 cell bounds of the refine scans
int peaks_regions(int H, int W);
// ctr: HH_PEAKS_PARTS work counters of the persistent grid, zero at launch (launch_topk_merge, the next launch, clears them again)
#define HH_PEAKS_PARTS 64
hipError_t launch_peaks(const DecodeSrc &src, int M, unsigned long long *cand_key, float *cellmax, unsigned short *supmax, float thr, int *ctr,
                        hipStream_t s);
// images flagged HH_DECODE_FALLBACK: joints[b, 0, k] = the top-1 candidate of joint k recomputed from the map
hipError_t launch_fallback_top1(const DecodeSrc &src, int M, const int32_t *flags, float *joints, hipStream_t s);
// per (b,k): merge the tiles' candidates -> scores_k, coords_k (x,y), tags_k (cand_val null: keys of positive values only, the score is
// read out of the key)
hipError_t launch_topk_merge(const DecodeSrc &src, int M, int ntiles, unsigned long long *cand_key, const float *cand_val,
                             float *tags_k, int32_t *coords_k, float *scores_k, int *peaks_ctr, hipStream_t s);
// per image: match_by_tag (+ the "no group" fallback); joints [B,M,K,3+E], num_people [B].  bounds_src != nullptr (mode 0, refine):
// extra workgroups of the launch also write the tag bounds of the refine scans into tagb (and, suptag != nullptr, the supers' hulls) and
// clear the 8 queue counters of ws_jobs
hipError_t launch_match(const float *tags_k, const int32_t *coords_k, const float *scores_k, int B, int K, int M, int E,
                        double det_thr, double tag_thr, float *joints, int32_t *num_people, float *ws_tags, int32_t *flags,
                        const DecodeSrc *bounds_src, float *tagb, unsigned *suptag, int32_t *ws_jobs, hipStream_t s);
// debug: the assignment solver alone, one wave on one n x n float64 matrix (n <= HH_MAX_PEOPLE); out[0..n) = starred column of each
// row, out[n] = 1 if the iteration guard ran out
hipError_t launch_munkres_debug(const double *cost, int n, int32_t *out, hipStream_t s);
// per image: quarter-pixel adjust (optional) and person scores
// adjust + person scores + (refine != 0) the mean tag of every person and the lists of its missing joints
hipError_t launch_adjust_scores(const DecodeSrc &src, int M, int adjust, int refine, float *joints, const int32_t *num_people, float *scores,
                                float *ws_prev, int32_t *ws_jobs, hipStream_t s);
// refine: the full-map argmax for every missing joint (work lists from launch_adjust_scores, tag bounds from launch_match), applied
// to `joints` by the scanning workgroup
hipError_t launch_refine(const DecodeSrc &src, int M, float *joints, const float *ws_prev, const int32_t *ws_jobs, const float *cellmax,
                         const float *tagb, hipStream_t s);
// the same for the default path of hh_decode (mode 0, src.avg == null): a two-level exact branch-and-bound over 8x8-cell supers
// (supmax [B*K][ceil(hq/8)*ceil(wq/8)] bf16 from launch_peaks, suptag [..][E] bf16 pairs from launch_match) and cells (decode_refine.hip)
hipError_t launch_refine_bb(const DecodeSrc &src, int M, float *joints, const float *ws_prev, const int32_t *ws_jobs, const float *cellmax,
                            const float *tagb, const unsigned short *supmax, const unsigned *suptag, hipStream_t s);
// dst (+)= weight * bilinear(src -> HxW), torch CPU arithmetic, any ratio (multi-scale heatmap aggregation)
hipError_t launch_resize_accumulate(const float *src, int64_t src_bs, int B, int K, int h, int w, float *dst, int64_t dst_bs, int H,
                                    int W, float weight, int init, hipStream_t s);
