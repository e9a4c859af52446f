"""COCO keypoint AP / AR (OKS) evaluation -- the `eval_coco` step of `src/keypoints/bin/eval.py:52-65`.

The reference delegates this to pycocotools 2.0.7 (`COCO.loadRes` + `COCOeval(iouType="keypoints")`), a third-party
package that is NOT in this image and not under /root/reference.  This module restates its published algorithm
(cocoeval.py: computeOks, evaluateImg, accumulate, summarize; coco.py: loadRes for keypoint results) so that the
result lists of `keypoints.evaluation` can be scored without it.  PARITY UNPINNED: no golden vector from pycocotools
could be generated here; the tests check hand-derived cases only.  Host-side numpy, not on the hot path.
"""
from __future__ import annotations

import json
from collections import defaultdict

import numpy as np

KPT_OKS_SIGMAS = np.array([.26, .25, .25, .35, .35, .79, .79, .72, .72, .62, .62, 1.07, 1.07, .87, .87, .89, .89]) / 10.0
IOU_THRS = np.linspace(.5, 0.95, int(np.round((0.95 - .5) / .05)) + 1, endpoint=True)
REC_THRS = np.linspace(.0, 1.00, int(np.round((1.00 - .0) / .01)) + 1, endpoint=True)
MAX_DETS = [20]
AREA_RNG = [[0 ** 2, 1e5 ** 2], [32 ** 2, 96 ** 2], [96 ** 2, 1e5 ** 2]]
AREA_LBL = ["all", "medium", "large"]


def load_results(results: list[dict]) -> list[dict]:
    """COCO.loadRes for keypoint results: area / bbox from the keypoint extent, ids 1..N."""
    out = []
    for i, ann in enumerate(results):
        a = dict(ann)
        s = a["keypoints"]
        x, y = s[0::3], s[1::3]
        x0, x1, y0, y1 = np.min(x), np.max(x), np.min(y), np.max(y)
        a["area"] = float((x1 - x0) * (y1 - y0))
        a["id"] = i + 1
        a["bbox"] = [x0, y0, x1 - x0, y1 - y0]
        out.append(a)
    return out


class COCOKeypointsEval:
    """evaluate() + accumulate() + summarize() of COCOeval(iouType="keypoints"), person category only (useCats with one
    category).  `gt_annotations` = the "annotations" list of a person_keypoints json, `results` = list of result dicts."""

    def __init__(self, gt_annotations: list[dict], results: list[dict], img_ids=None, sigmas=KPT_OKS_SIGMAS):
        self.sigmas = np.asarray(sigmas, dtype=np.float64)
        dts = load_results(results)
        self.img_ids = sorted(set(img_ids) if img_ids is not None else {d["image_id"] for d in dts})
        keep = set(self.img_ids)
        self._gts, self._dts = defaultdict(list), defaultdict(list)
        for g in gt_annotations:
            if g["image_id"] not in keep or g.get("category_id", 1) != 1:
                continue
            g = dict(g)
            g["ignore"] = bool(g.get("iscrowd", 0))               # cocoeval.py _prepare
            g["ignore"] = (g.get("num_keypoints", 0) == 0) or g["ignore"]
            self._gts[g["image_id"]].append(g)
        for d in dts:
            if d["image_id"] in keep and d.get("category_id", 1) == 1:
                self._dts[d["image_id"]].append(d)
        self.stats = None

    # ------------------------------------------------------------------ computeOks
    def _oks(self, img_id):
        gts, dts = self._gts[img_id], self._dts[img_id]
        inds = np.argsort([-d["score"] for d in dts], kind="mergesort")
        dts = [dts[i] for i in inds][: MAX_DETS[-1]]
        if len(gts) == 0 or len(dts) == 0:
            return []
        ious = np.zeros((len(dts), len(gts)))
        vars_ = (self.sigmas * 2) ** 2
        k = len(self.sigmas)
        for j, gt in enumerate(gts):
            g = np.array(gt["keypoints"], dtype=np.float64)
            xg, yg, vg = g[0::3], g[1::3], g[2::3]
            k1 = np.count_nonzero(vg > 0)
            bb = gt["bbox"]
            x0, x1 = bb[0] - bb[2], bb[0] + bb[2] * 2
            y0, y1 = bb[1] - bb[3], bb[1] + bb[3] * 2
            for i, dt in enumerate(dts):
                d = np.array(dt["keypoints"], dtype=np.float64)
                xd, yd = d[0::3], d[1::3]
                if k1 > 0:
                    dx, dy = xd - xg, yd - yg
                else:  # no labelled joint: distance to the doubled box
                    z = np.zeros(k)
                    dx = np.max((z, x0 - xd), axis=0) + np.max((z, xd - x1), axis=0)
                    dy = np.max((z, y0 - yd), axis=0) + np.max((z, yd - y1), axis=0)
                e = (dx ** 2 + dy ** 2) / vars_ / (gt["area"] + np.spacing(1)) / 2
                if k1 > 0:
                    e = e[vg > 0]
                ious[i, j] = np.sum(np.exp(-e)) / e.shape[0]
        return ious

    # ------------------------------------------------------------------ evaluateImg
    def _evaluate_img(self, img_id, ious, a_rng, max_det):
        gt, dt = self._gts[img_id], self._dts[img_id]
        if len(gt) == 0 and len(dt) == 0:
            return None
        g_ign = [1 if (g["ignore"] or g["area"] < a_rng[0] or g["area"] > a_rng[1]) else 0 for g in gt]
        gtind = np.argsort(g_ign, kind="mergesort")
        gt = [gt[i] for i in gtind]
        dtind = np.argsort([-d["score"] for d in dt], kind="mergesort")
        dt = [dt[i] for i in dtind[:max_det]]
        iscrowd = [int(g.get("iscrowd", 0)) for g in gt]
        ious = ious[:, gtind] if len(ious) > 0 else ious
        T, G, D = len(IOU_THRS), len(gt), len(dt)
        gtm, dtm = np.zeros((T, G)), np.zeros((T, D))
        gt_ig = np.array([g_ign[i] for i in gtind])
        dt_ig = np.zeros((T, D))
        if len(ious) != 0:
            for tind, t in enumerate(IOU_THRS):
                for dind, d in enumerate(dt):
                    iou = min([t, 1 - 1e-10])
                    m = -1
                    for gind in range(G):
                        if gtm[tind, gind] > 0 and not iscrowd[gind]:
                            continue
                        if m > -1 and gt_ig[m] == 0 and gt_ig[gind] == 1:
                            break
                        if ious[dind, gind] < iou:
                            continue
                        iou = ious[dind, gind]
                        m = gind
                    if m == -1:
                        continue
                    dt_ig[tind, dind] = gt_ig[m]
                    dtm[tind, dind] = gt[m]["id"]
                    gtm[tind, m] = d["id"]
        a = np.array([d["area"] < a_rng[0] or d["area"] > a_rng[1] for d in dt]).reshape((1, len(dt)))
        dt_ig = np.logical_or(dt_ig, np.logical_and(dtm == 0, np.repeat(a, T, 0)))
        return {"dtMatches": dtm, "dtScores": [d["score"] for d in dt], "gtIgnore": gt_ig, "dtIgnore": dt_ig}

    # ------------------------------------------------------------------ evaluate + accumulate + summarize
    def evaluate(self):
        ious = {i: self._oks(i) for i in self.img_ids}
        max_det = MAX_DETS[-1]
        T, R, A, M = len(IOU_THRS), len(REC_THRS), len(AREA_RNG), len(MAX_DETS)
        precision, recall = -np.ones((T, R, A, M)), -np.ones((T, A, M))
        for a, a_rng in enumerate(AREA_RNG):
            E = [e for e in (self._evaluate_img(i, ious[i], a_rng, max_det) for i in self.img_ids) if e is not None]
            for m, md in enumerate(MAX_DETS):
                if len(E) == 0:
                    continue
                dt_scores = np.concatenate([e["dtScores"][0:md] for e in E])
                inds = np.argsort(-dt_scores, kind="mergesort")
                dtm = np.concatenate([e["dtMatches"][:, 0:md] for e in E], axis=1)[:, inds]
                dt_ig = np.concatenate([e["dtIgnore"][:, 0:md] for e in E], axis=1)[:, inds]
                gt_ig = np.concatenate([e["gtIgnore"] for e in E])
                npig = np.count_nonzero(gt_ig == 0)
                if npig == 0:
                    continue
                tps = np.logical_and(dtm, np.logical_not(dt_ig))
                fps = np.logical_and(np.logical_not(dtm), np.logical_not(dt_ig))
                tp_sum = np.cumsum(tps, axis=1).astype(dtype=float)
                fp_sum = np.cumsum(fps, axis=1).astype(dtype=float)
                for t, (tp, fp) in enumerate(zip(tp_sum, fp_sum)):
                    nd = len(tp)
                    rc = tp / npig
                    pr = tp / (fp + tp + np.spacing(1))
                    q = np.zeros((R,))
                    recall[t, a, m] = rc[-1] if nd else 0
                    pr = pr.tolist()
                    q = q.tolist()
                    for i in range(nd - 1, 0, -1):  # precision envelope
                        if pr[i] > pr[i - 1]:
                            pr[i - 1] = pr[i]
                    inds_r = np.searchsorted(rc, REC_THRS, side="left")
                    for ri, pi in enumerate(inds_r):
                        if pi >= nd:
                            break
                        q[ri] = pr[pi]
                    precision[t, :, a, m] = np.array(q)
        self.precision, self.recall = precision, recall

        def summ(ap, iou_thr=None, area="all"):
            aind = AREA_LBL.index(area)
            s = precision[:, :, aind, 0] if ap else recall[:, aind, 0]
            if iou_thr is not None:
                s = s[np.where(np.isclose(iou_thr, IOU_THRS))[0]]
            return -1.0 if len(s[s > -1]) == 0 else float(np.mean(s[s > -1]))

        self.stats = np.array([summ(1), summ(1, .5), summ(1, .75), summ(1, area="medium"), summ(1, area="large"),
                               summ(0), summ(0, .5), summ(0, .75), summ(0, area="medium"), summ(0, area="large")])
        return self.stats

    def summary_text(self) -> str:
        names = ["AP", "AP .5", "AP .75", "AP (M)", "AP (L)", "AR", "AR .5", "AR .75", "AR (M)", "AR (L)"]
        return "\n".join(f" {n:7s} = {v:0.3f}" for n, v in zip(names, self.stats))


def eval_coco(annots_path: str, results_path: str) -> np.ndarray:
    """bin/eval.py:52-65 with files: person_keypoints_*.json + the results json written by the evaluation loop."""
    with open(annots_path) as f:
        gt = json.load(f)["annotations"]
    with open(results_path) as f:
        res = json.load(f)
    ev = COCOKeypointsEval(gt, res)
    ev.evaluate()
    print(ev.summary_text())
    return ev.stats
