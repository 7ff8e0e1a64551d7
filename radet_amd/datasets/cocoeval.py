"""COCO-protocol bounding-box evaluation on the host (NumPy), as BOPDataset.evaluate drives it
(radet/datasets/bop.py:120-302 -> pycocotools COCO / COCOeval, iouType 'bbox').

pycocotools is not part of the reference tree nor of this image, so this is a restatement of its published algorithm
(cocoeval.py of pycocotools 2.0.x: per image / category / area range greedy matching in descending score order,
101-point interpolated precision, the 12 summary statistics).  PARITY UNPINNED against pycocotools itself; checked
on hand-worked cases in tests/test_bop.py.  Pure host code: no kernels, no GPU.

Knife-edges decided here from the published algorithm, with nothing to check them against:
  * matching: a detection takes the unmatched gt with the HIGHEST IoU among those with IoU >= min(t, 1 - 1e-10) (pycocotools
    starts `iou = min([t, 1 - 1e-10])` and skips candidates with `ious < iou`, i.e. equality at the threshold MATCHES);
    equal IoUs keep the first gt in the evaluator's order (non-ignored gts first, then ignored / crowd);
  * detections are ranked by score with a STABLE merge sort (`np.argsort(-score, kind="mergesort")`), so equal scores keep
    file order -- per image and again across images in accumulate();
  * area ranges are closed on both ends (`area < lo or area > hi` is out of range); bbox areas come from the annotation's
    `area` field for gts and w * h for detections;
  * precision = tp / (tp + fp + eps) with eps = np.spacing(1), recall thresholds searched with `np.searchsorted(..., "left")`,
    empty categories / ranges report -1;
  * `segm` is not implemented (boxes only); `proposal_fast` = eval_recalls() at the end of this file, which IS pinned (bit-equal
    to the reference's eval_recalls on a golden written by it, tests/golden/recall.npz).
tests/test_bop.py also runs the evaluator against a second, deliberately naive loop-by-loop restatement of the same published
algorithm on random data (crowds, score ties, all three area ranges, images without ground truth / detections)."""
import collections
import copy
import json

import numpy as np


class COCO:
    """Minimal index over a COCO-style annotation dict / file (pycocotools.coco.COCO's getters under mmdet's names)."""

    def __init__(self, annotation=None):
        self.dataset = dict(images=[], annotations=[], categories=[])
        if isinstance(annotation, str):
            with open(annotation) as f:
                self.dataset = json.load(f)
        elif annotation is not None:
            self.dataset = annotation
        self.create_index()

    def create_index(self):
        self.anns, self.imgs, self.cats = {}, {}, {}
        self.img_ann_map = collections.defaultdict(list)
        self.cat_img_map = collections.defaultdict(list)
        for ann in self.dataset.get("annotations", []):
            self.img_ann_map[ann["image_id"]].append(ann)
            self.anns[ann["id"]] = ann
            self.cat_img_map[ann["category_id"]].append(ann["image_id"])
        for img in self.dataset.get("images", []):
            self.imgs[img["id"]] = img
        for cat in self.dataset.get("categories", []):
            self.cats[cat["id"]] = cat

    def get_cat_ids(self, cat_names=()):
        cats = self.dataset.get("categories", [])
        if cat_names:
            cats = [c for c in cats if c["name"] in cat_names]
        return [c["id"] for c in cats]

    def get_img_ids(self):
        return list(self.imgs.keys())

    def get_ann_ids(self, img_ids=()):
        img_ids = img_ids if isinstance(img_ids, (list, tuple)) else [img_ids]
        return [a["id"] for i in img_ids for a in self.img_ann_map.get(i, [])]

    def load_anns(self, ids):
        return [self.anns[i] for i in ids]

    def load_imgs(self, ids):
        return [self.imgs[i] for i in ids]

    def load_cats(self, ids):
        ids = ids if isinstance(ids, (list, tuple)) else [ids]
        return [self.cats[i] for i in ids]

    loadCats, loadAnns, loadImgs = load_cats, load_anns, load_imgs

    def loadRes(self, results):
        """detections (list of dict(image_id, category_id, bbox xywh, score) or a json file of them) -> COCO index"""
        if isinstance(results, str):
            with open(results) as f:
                results = json.load(f)
        if len(results) == 0:
            raise IndexError("empty results")
        res = COCO()
        res.dataset["images"] = list(self.dataset.get("images", []))
        res.dataset["categories"] = copy.deepcopy(self.dataset.get("categories", []))
        anns = copy.deepcopy(results)
        assert set(a["image_id"] for a in anns) <= set(self.get_img_ids()), "Results do not correspond to current coco set"
        for k, a in enumerate(anns):
            x, y, w, h = a["bbox"]
            a["area"] = w * h
            a["id"] = k + 1
            a["iscrowd"] = 0
        res.dataset["annotations"] = anns
        res.create_index()
        return res


class Params:
    def __init__(self):
        self.imgIds, self.catIds = [], []
        self.iouThrs = np.linspace(.5, 0.95, int(np.round((0.95 - .5) / .05)) + 1, endpoint=True)
        self.recThrs = np.linspace(.0, 1.00, int(np.round((1.00 - .0) / .01)) + 1, endpoint=True)
        self.maxDets = [1, 10, 100]
        self.areaRng = [[0 ** 2, 1e5 ** 2], [0 ** 2, 32 ** 2], [32 ** 2, 96 ** 2], [96 ** 2, 1e5 ** 2]]
        self.areaRngLbl = ["all", "small", "medium", "large"]
        self.useCats = 1


def bbox_iou(dt, gt, iscrowd):
    """IoU matrix of xywh boxes (maskApi bbIou): crowd gts use the detection's area as the union"""
    dt, gt = np.asarray(dt, np.float64).reshape(-1, 4), np.asarray(gt, np.float64).reshape(-1, 4)
    out = np.zeros((dt.shape[0], gt.shape[0]))
    for j in range(gt.shape[0]):
        ga = gt[j, 2] * gt[j, 3]
        w = np.minimum(dt[:, 0] + dt[:, 2], gt[j, 0] + gt[j, 2]) - np.maximum(dt[:, 0], gt[j, 0])
        h = np.minimum(dt[:, 1] + dt[:, 3], gt[j, 1] + gt[j, 3]) - np.maximum(dt[:, 1], gt[j, 1])
        inter = np.where((w <= 0) | (h <= 0), 0.0, w * h)
        da = dt[:, 2] * dt[:, 3]
        union = da if iscrowd[j] else da + ga - inter
        out[:, j] = inter / union
    return out


class COCOeval:
    def __init__(self, cocoGt, cocoDt, iouType="bbox"):
        if iouType != "bbox":
            raise NotImplementedError("only bounding-box evaluation (the RADet detector's metric) is implemented")
        self.cocoGt, self.cocoDt = cocoGt, cocoDt
        self.params = Params()
        self.params.imgIds = sorted(cocoGt.get_img_ids())
        self.params.catIds = sorted(cocoGt.get_cat_ids())
        self.evalImgs, self.eval, self.stats = [], {}, []

    def _prepare(self):
        p = self.params
        img_set, cat_set = set(p.imgIds), set(p.catIds)
        self._gts, self._dts = collections.defaultdict(list), collections.defaultdict(list)
        for g in self.cocoGt.dataset.get("annotations", []):
            if g["image_id"] not in img_set or (p.useCats and g["category_id"] not in cat_set):
                continue
            g = dict(g)
            g["ignore"] = bool(g.get("iscrowd", 0))              # cocoeval._prepare: 'ignore' is overridden by iscrowd
            self._gts[g["image_id"], g["category_id"] if p.useCats else -1].append(g)
        for d in self.cocoDt.dataset.get("annotations", []):
            if d["image_id"] not in img_set or (p.useCats and d["category_id"] not in cat_set):
                continue
            self._dts[d["image_id"], d["category_id"] if p.useCats else -1].append(d)

    def evaluate(self):
        p = self.params
        p.imgIds = list(np.unique(p.imgIds))
        p.catIds = list(np.unique(p.catIds)) if p.useCats else [-1]
        p.maxDets = sorted(p.maxDets)
        self._prepare()
        max_det = p.maxDets[-1]
        self.ious = {(i, c): self._compute_iou(i, c, max_det) for i in p.imgIds for c in p.catIds}
        self.evalImgs = [self._evaluate_img(i, c, a, max_det) for c in p.catIds for a in p.areaRng for i in p.imgIds]

    def _compute_iou(self, img, cat, max_det):
        gt, dt = self._gts[img, cat], self._dts[img, cat]
        if not gt or not dt:
            return []
        dt = [dt[i] for i in np.argsort([-d["score"] for d in dt], kind="mergesort")][:max_det]
        return bbox_iou([d["bbox"] for d in dt], [g["bbox"] for g in gt], [int(g.get("iscrowd", 0)) for g in gt])

    def _evaluate_img(self, img, cat, a_rng, max_det):
        p = self.params
        gt, dt = self._gts[img, cat], self._dts[img, cat]
        if not gt and not dt:
            return None
        g_ig = np.array([1 if (g["ignore"] or g["area"] < a_rng[0] or g["area"] > a_rng[1]) else 0 for g in gt], int)
        gtind = np.argsort(g_ig, kind="mergesort")               # ignored gts last
        gt = [gt[i] for i in gtind]
        g_ig = g_ig[gtind]
        dt = [dt[i] for i in np.argsort([-d["score"] for d in dt], kind="mergesort")][:max_det]
        iscrowd = [int(g.get("iscrowd", 0)) for g in gt]
        ious = self.ious[img, cat]
        ious = ious[:, gtind] if len(ious) > 0 else ious
        T, G, D = len(p.iouThrs), len(gt), len(dt)
        gtm, dtm, dt_ig = np.zeros((T, G)), np.zeros((T, D)), np.zeros((T, D))
        if len(ious) != 0:
            for ti, t in enumerate(p.iouThrs):
                for di in range(D):
                    iou = min(t, 1 - 1e-10)
                    m = -1
                    for gi in range(G):
                        if gtm[ti, gi] > 0 and not iscrowd[gi]:
                            continue                             # gt already matched (crowds may match repeatedly)
                        if m > -1 and g_ig[m] == 0 and g_ig[gi] == 1:
                            break                                # matched a regular gt and only ignored ones remain
                        if ious[di, gi] < iou:
                            continue
                        iou = ious[di, gi]
                        m = gi
                    if m == -1:
                        continue
                    dt_ig[ti, di] = g_ig[m]
                    dtm[ti, di] = gt[m]["id"]
                    gtm[ti, m] = dt[di]["id"]
        a = np.array([d["area"] < a_rng[0] or d["area"] > a_rng[1] for d in dt]).reshape(1, D)
        dt_ig = np.logical_or(dt_ig, np.logical_and(dtm == 0, np.repeat(a, T, 0)))
        return dict(image_id=img, category_id=cat, aRng=a_rng, maxDet=max_det, dtIds=[d["id"] for d in dt],
                    gtIds=[g["id"] for g in gt], dtMatches=dtm, gtMatches=gtm, dtScores=[d["score"] for d in dt],
                    gtIgnore=g_ig, dtIgnore=dt_ig)

    def accumulate(self):
        p = self.params
        T, R, K, A, M = len(p.iouThrs), len(p.recThrs), len(p.catIds), len(p.areaRng), len(p.maxDets)
        precision, recall, scores = -np.ones((T, R, K, A, M)), -np.ones((T, K, A, M)), -np.ones((T, R, K, A, M))
        I = len(p.imgIds)
        for k in range(K):
            for a in range(A):
                for m, max_det in enumerate(p.maxDets):
                    E = [self.evalImgs[k * A * I + a * I + i] for i in range(I)]
                    E = [e for e in E if e is not None]
                    if not E:
                        continue
                    dt_scores = np.concatenate([e["dtScores"][:max_det] for e in E])
                    inds = np.argsort(-dt_scores, kind="mergesort")
                    dt_scores = dt_scores[inds]
                    dtm = np.concatenate([e["dtMatches"][:, :max_det] for e in E], axis=1)[:, inds]
                    dt_ig = np.concatenate([e["dtIgnore"][:, :max_det] for e in E], axis=1)[:, inds]
                    g_ig = np.concatenate([e["gtIgnore"] for e in E])
                    npig = np.count_nonzero(g_ig == 0)
                    if npig == 0:
                        continue
                    tps = np.logical_and(dtm, np.logical_not(dt_ig))
                    fps = np.logical_and(np.logical_not(dtm), np.logical_not(dt_ig))
                    tp_sum = np.cumsum(tps, axis=1).astype(float)
                    fp_sum = np.cumsum(fps, axis=1).astype(float)
                    for t, (tp, fp) in enumerate(zip(tp_sum, fp_sum)):
                        nd = len(tp)
                        rc = tp / npig
                        pr = tp / (fp + tp + np.spacing(1))
                        q, ss = np.zeros(R), np.zeros(R)
                        recall[t, k, a, m] = rc[-1] if nd else 0
                        pr = pr.tolist()
                        for i in range(nd - 1, 0, -1):             # precision envelope (monotone non-increasing in recall)
                            if pr[i] > pr[i - 1]:
                                pr[i - 1] = pr[i]
                        ri = np.searchsorted(rc, p.recThrs, side="left")
                        for r, pi in enumerate(ri):
                            if pi >= nd:
                                break
                            q[r] = pr[pi]
                            ss[r] = dt_scores[pi]
                        precision[t, :, k, a, m] = q
                        scores[t, :, k, a, m] = ss
        self.eval = dict(params=p, counts=[T, R, K, A, M], precision=precision, recall=recall, scores=scores)

    def _summarize(self, ap=1, iou_thr=None, area="all", max_dets=100):
        p = self.params
        aind = [i for i, a in enumerate(p.areaRngLbl) if a == area]
        mind = [i for i, m in enumerate(p.maxDets) if m == max_dets]
        s = self.eval["precision"] if ap == 1 else self.eval["recall"]
        if iou_thr is not None:
            s = s[np.where(np.isclose(iou_thr, p.iouThrs))[0]]
        s = s[:, :, :, aind, mind] if ap == 1 else s[:, :, aind, mind]
        return -1.0 if len(s[s > -1]) == 0 else float(np.mean(s[s > -1]))

    def summarize(self):
        """the 12 COCO detection statistics (AP, AP50, AP75, AP small / medium / large, AR@maxDets, AR s / m / l)"""
        md = self.params.maxDets
        self.stats = np.array([
            self._summarize(1, max_dets=md[2]), self._summarize(1, iou_thr=.5, max_dets=md[2]),
            self._summarize(1, iou_thr=.75, max_dets=md[2]), self._summarize(1, area="small", max_dets=md[2]),
            self._summarize(1, area="medium", max_dets=md[2]), self._summarize(1, area="large", max_dets=md[2]),
            self._summarize(0, max_dets=md[0]), self._summarize(0, max_dets=md[1]), self._summarize(0, max_dets=md[2]),
            self._summarize(0, area="small", max_dets=md[2]), self._summarize(0, area="medium", max_dets=md[2]),
            self._summarize(0, area="large", max_dets=md[2])])
        return self.stats


# ----------------------------------------------------------------------------- class-agnostic recall ('proposal_fast')
def box_iou_matrix(a, b, eps=1e-6):
    """IoU of every box of a (n, 4) with every box of b (k, 4), fp32, the arithmetic of the reference's NumPy
    `bbox_overlaps` (radet/core/evaluation/bbox_overlaps.py:4-50: plain x2 - x1 extents, union clamped at eps)."""
    a, b = np.asarray(a, np.float32).reshape(-1, 4), np.asarray(b, np.float32).reshape(-1, 4)
    if a.shape[0] == 0 or b.shape[0] == 0:
        return np.zeros((a.shape[0], b.shape[0]), np.float32)
    area_a = (a[:, 2] - a[:, 0]) * (a[:, 3] - a[:, 1])
    area_b = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    w = np.maximum(np.minimum(a[:, None, 2], b[None, :, 2]) - np.maximum(a[:, None, 0], b[None, :, 0]), 0)
    h = np.maximum(np.minimum(a[:, None, 3], b[None, :, 3]) - np.maximum(a[:, None, 1], b[None, :, 1]), 0)
    inter = w * h
    union = np.maximum(area_a[:, None] + area_b[None, :] - inter, eps)
    return (inter / union).astype(np.float32)


def eval_recalls(gts, proposals, proposal_nums=(100, 300, 1000), iou_thrs=0.5):
    """Recall of ground-truth boxes by the top-N proposals per image -> array [len(proposal_nums), len(iou_thrs)]
    (radet/core/evaluation/recall.py:10-106).  Per image the proposals are ranked by score (column 4, when present), cut
    at the largest N, and ground truths and proposals are matched one to one greedily by IoU: the best remaining pair is
    taken (first in row-major order among equals), its row and column leave the pool, until every ground truth has had a
    turn -- a ground truth left without a proposal scores -1 (0 when the image has no proposal at all).  A ground truth
    counts as recalled at threshold t when its matched IoU >= t."""
    nums = np.atleast_1d(np.asarray(proposal_nums)).astype(np.int64)
    thrs = np.atleast_1d(np.asarray(iou_thrs, dtype=np.float64))
    assert len(gts) == len(proposals)
    mats = []
    for gt, pr in zip(gts, proposals):
        pr = np.asarray(pr)
        if pr.ndim == 2 and pr.shape[1] == 5:
            pr = pr[np.argsort(pr[:, 4])[::-1]]
        pr = pr[:min(pr.shape[0], int(nums[-1]))]
        n_gt = 0 if gt is None else np.asarray(gt).shape[0]
        mats.append(box_iou_matrix(gt, pr[:, :4]) if n_gt else np.zeros((0, pr.shape[0]), np.float32))
    total = sum(m.shape[0] for m in mats)
    matched = np.zeros((nums.size, total), np.float32)
    for k, n in enumerate(nums):
        col = 0
        for m in mats:
            pool = m[:, :n].copy()
            g = pool.shape[0]
            if pool.size:
                for j in range(g):
                    r, c = np.unravel_index(np.argmax(pool), pool.shape)
                    matched[k, col + j] = pool[r, c]
                    pool[r, :] = -1
                    pool[:, c] = -1
            col += g
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.stack([(matched >= t).sum(axis=1) / float(total) for t in thrs], axis=1)

