"""BOP I/O and evaluation (radet/datasets/bop.py, tools/bop_to_coco.py, tools/coco_to_bop.py): host-only.
The COCO-protocol evaluator is a restatement of pycocotools (absent here), so it is checked on cases whose AP / AR
follow by hand from the protocol's definition."""
import json
import os

import numpy as np
import pytest


def make_bop_tree(root, scenes=(1, 48), frames=(0, 7, 12)):
    """two BOP scenes x three frames, 2-3 objects per frame, YCB-V ids; returns the image list path"""
    rng = np.random.RandomState(0)
    lines = []
    for s in scenes:
        d = os.path.join(root, "train_pbr", f"{s:06d}")
        os.makedirs(os.path.join(d, "rgb"))
        gt, info = {}, {}
        for fr in frames:
            open(os.path.join(d, "rgb", f"{fr:06d}.jpg"), "wb").close()
            n = 2 + (fr % 2)
            gt[str(fr)] = [dict(obj_id=int(rng.randint(1, 22)), cam_R_m2c=[1, 0, 0, 0, 1, 0, 0, 0, 1], cam_t_m2c=[0, 0, 500])
                           for _ in range(n)]
            info[str(fr)] = []
            for k in range(n):
                x, y, w, h = int(rng.randint(0, 400)), int(rng.randint(0, 300)), int(rng.randint(20, 200)), int(rng.randint(20, 150))
                info[str(fr)].append(dict(bbox_obj=[x, y, w, h], bbox_visib=[x + 2, y + 2, w - 4, h - 4],
                                          visib_fract=float(rng.choice([0.05, 0.5, 0.95])), px_count_all=w * h))
            lines.append(f"{s:06d}/rgb/{fr:06d}.jpg")
        json.dump(gt, open(os.path.join(d, "scene_gt.json"), "w"))
        json.dump(info, open(os.path.join(d, "scene_gt_info.json"), "w"))
    lst = os.path.join(root, "train_pbr.txt")
    open(lst, "w").write("\n".join(lines[:-1]) + "\n")            # the last frame is not in the image list
    return lst


@pytest.fixture()
def bop(tmp_path):
    from radet_amd.datasets.bop_convert import bop_to_coco
    lst = make_bop_tree(str(tmp_path))
    coco = bop_to_coco(os.path.join(str(tmp_path), "train_pbr"), lst, "ycbv")
    ann = os.path.join(str(tmp_path), "train_pbr.json")
    json.dump(coco, open(ann, "w"))
    return str(tmp_path), ann, coco


def test_bop_to_coco_layout(bop):
    root, ann, coco = bop
    assert len(coco["images"]) == 5 and len(coco["categories"]) == 21 and coco["categories"][0] == dict(id=1, name="master_chef_can")
    assert [im["id"] for im in coco["images"]] == [1, 2, 3, 4, 5]          # running ids over scenes, 1-based
    assert coco["images"][3]["file_name"] == "000048/rgb/000000.jpg" and coco["images"][0]["width"] == 640
    info = json.load(open(os.path.join(root, "train_pbr", "000001", "scene_gt_info.json")))
    a0 = coco["annotations"][0]
    assert a0["bbox"] == info["0"][0]["bbox_obj"] and a0["area"] == a0["bbox"][2] * a0["bbox"][3] and a0["iscrowd"] == 0
    assert a0["visib_fract"] == info["0"][0]["visib_fract"] and a0["id"] == 1 and a0["image_id"] == 1
    assert len(coco["annotations"]) == 2 + 3 + 2 + 2 + 3                    # frame 12 of scene 48 was not listed
    from radet_amd.datasets.bop_convert import bop_to_coco
    amodal = bop_to_coco(os.path.join(root, "train_pbr"), os.path.join(root, "train_pbr.txt"), "ycbv", amodal=True)
    assert amodal["annotations"][0]["bbox"] == info["0"][0]["bbox_visib"]
    test = bop_to_coco(os.path.join(root, "train_pbr"), os.path.join(root, "train_pbr.txt"), "ycbv", without_gt=True)
    assert "annotations" not in test and [im["id"] for im in test["images"]] == [0, 1, 2, 3, 4]


def test_dataset_parse_and_pipeline(bop):
    from radet_amd.datasets import build_dataset
    root, ann, coco = bop
    seen = []
    ds = build_dataset(dict(type="BOPDataset", ann_file="train_pbr.json", data_root=root, img_prefix="train_pbr/",
                            pipeline=[lambda r: seen.append(r) or r], min_visib_frac=0.1))
    assert len(ds) == 5 and ds.img_prefix == os.path.join(root, "train_pbr/") and len(ds.flag) == 5 and ds.flag.all()
    info = ds.get_ann_info(1)                                              # frame 7 of scene 1: three objects
    raw = [a for a in coco["annotations"] if a["image_id"] == 2]
    keep = [a for a in raw if a["visib_fract"] >= 0.1]
    assert info["bboxes"].dtype == np.float32 and info["bboxes"].shape == (len(keep), 4)
    assert info["bboxes_ignore"].shape == (len(raw) - len(keep), 4)
    x, y, w, h = keep[0]["bbox"]
    assert np.allclose(info["bboxes"][0], [x, y, x + w, y + h])
    assert info["labels"].tolist() == [a["category_id"] - 1 for a in keep]
    idx_of = {a["id"]: i for i, a in enumerate(raw)}
    assert info["masks"] == [f"000001/mask_visib/000007_{idx_of[a['id']]:06d}.png" for a in keep]
    assert info["seg_map"] == "000001/rgb/000007.png"
    out = ds[0]
    assert out["img_info"]["filename"] == "000001/rgb/000000.jpg" and out["bbox_fields"] == [] and seen
    # filter_empty_gt: an image without annotations of the wanted classes is dropped in training mode only
    coco2 = json.loads(json.dumps(coco))
    coco2["images"].append(dict(file_name="000048/rgb/000099.jpg", id=99, width=640, height=480))
    p2 = os.path.join(root, "with_empty.json")
    json.dump(coco2, open(p2, "w"))
    assert len(build_dataset(dict(type="BOPDataset", ann_file=p2, pipeline=[]))) == 5
    assert len(build_dataset(dict(type="BOPDataset", ann_file=p2, pipeline=[], test_mode=True))) == 6


def gt_as_results(ds, coco, score=lambda a: 0.9, jitter=0.0):
    """detections = the ground truth boxes (xyxy + score) in the detector's per-class list format"""
    res = []
    for i in range(len(ds)):
        per = [np.zeros((0, 5), np.float32) for _ in ds.CLASSES]
        for a in coco["annotations"]:
            if a["image_id"] == ds.img_ids[i]:
                x, y, w, h = a["bbox"]
                per[ds.cat2label[a["category_id"]]] = np.concatenate(
                    [per[ds.cat2label[a["category_id"]]], np.array([[x + jitter, y, x + w + jitter, y + h, score(a)]], np.float32)])
        res.append(per)
    return res


def test_evaluate_perfect_and_degraded(bop):
    from radet_amd.datasets import BOPDataset
    root, ann, coco = bop
    ds = BOPDataset(ann, pipeline=[], test_mode=True)
    ev = ds.evaluate(gt_as_results(ds, coco), logger="silent")
    assert ev["bbox_mAP"] == 1.0 and ev["bbox_mAP_50"] == 1.0 and ev["bbox_mAP_75"] == 1.0 and ev["bbox_AR@100"] == 1.0
    assert ev["bbox_mAP_copypaste"].startswith("1.000 1.000 1.000")
    # every box shifted by 20 % of ... a fixed 6 px: IoU stays > 0.5 for all (w >= 20) but drops below 0.95 for small ones
    ev2 = ds.evaluate(gt_as_results(ds, coco, jitter=6.0), logger="silent", classwise=True)
    assert ev2["bbox_mAP_50"] == 1.0 and ev2["bbox_mAP"] < 1.0 and len(ev2["classwise"]) == 21
    # no detections at all
    empty = [[np.zeros((0, 5), np.float32) for _ in ds.CLASSES] for _ in range(len(ds))]
    assert ds.evaluate(empty, logger="silent") == {}
    with pytest.raises(KeyError):
        ds.evaluate(empty, metric="keypoints")
    # BOP submission records and the per-scene files of tools/coco_to_bop.py
    sub = BOPDataset(ann, pipeline=[], test_mode=True, bop_submission=True)
    files, tmp = sub.format_results(gt_as_results(sub, coco), jsonfile_prefix=os.path.join(root, "sub"))
    recs = json.load(open(files["bbox"]))
    assert recs[0].keys() == {"scene_id", "image_id", "category_id", "bbox", "score", "time"} and recs[0]["time"] == -1.0
    assert sorted({r["scene_id"] for r in recs}) == [1, 48] and {r["image_id"] for r in recs} == {0, 7, 12}
    from radet_amd.datasets.bop_convert import coco_to_bop
    conv = coco_to_bop(recs, save_dir=os.path.join(root, "bop_out"))
    saved = json.load(open(os.path.join(root, "bop_out", "000048", "scene_gt_info.json")))
    assert set(saved) == {"0", "7"} and saved["0"][0].keys() == {"bbox_obj", "obj_id", "score"} and 1 in conv


def _naive_coco_stats(images, anns, dets, cat_ids, iou_thrs, max_dets=(1, 10, 100)):
    """pycocotools' published bbox evaluation, loop by loop (no vectorisation, no shared code with cocoeval.py): returns the 12
    summary numbers.  Written to be obviously the algorithm, not to be fast."""
    area_rngs = [(0, 1e10), (0, 32 ** 2), (32 ** 2, 96 ** 2), (96 ** 2, 1e10)]
    rec_thrs = np.linspace(0.0, 1.0, 101)

    def iou(d, g, crowd):
        ix = max(0.0, min(d[0] + d[2], g[0] + g[2]) - max(d[0], g[0]))
        iy = max(0.0, min(d[1] + d[3], g[1] + g[3]) - max(d[1], g[1]))
        inter = ix * iy
        union = d[2] * d[3] if crowd else d[2] * d[3] + g[2] * g[3] - inter
        return inter / union if union > 0 else 0.0

    T, R, K, A, M = len(iou_thrs), 101, len(cat_ids), 4, len(max_dets)
    precision = -np.ones((T, R, K, A, M))
    recall = -np.ones((T, K, A, M))
    for k, cat in enumerate(cat_ids):
        for ai, (lo, hi) in enumerate(area_rngs):
            for mi, md in enumerate(max_dets):
                scores, matched, ignored, npos = [], [[] for _ in range(T)], [[] for _ in range(T)], 0
                for img in images:
                    gts = [a for a in anns if a["image_id"] == img and a["category_id"] == cat]
                    dts = [d for d in dets if d["image_id"] == img and d["category_id"] == cat]
                    if not gts and not dts:
                        continue
                    g_ign = [bool(g["iscrowd"]) or g["area"] < lo or g["area"] > hi for g in gts]
                    order = sorted(range(len(gts)), key=lambda i: g_ign[i])            # stable: non-ignored first
                    gts, g_ign = [gts[i] for i in order], [g_ign[i] for i in order]
                    d_order = np.argsort([-d["score"] for d in dts], kind="mergesort")[:md]
                    dts = [dts[i] for i in d_order]
                    npos += sum(1 for x in g_ign if not x)
                    for ti, t in enumerate(iou_thrs):
                        g_taken = [False] * len(gts)
                        for d in dts:
                            best, m = min(t, 1 - 1e-10), -1
                            for gi, g in enumerate(gts):
                                if g_taken[gi] and not g["iscrowd"]:
                                    continue
                                if m > -1 and not g_ign[m] and g_ign[gi]:
                                    break
                                v = iou(d["bbox"], g["bbox"], g["iscrowd"])
                                if v < best:
                                    continue
                                best, m = v, gi
                            if m > -1:
                                g_taken[m] = True
                                matched[ti].append(True)
                                ignored[ti].append(g_ign[m])
                            else:
                                a_d = d["bbox"][2] * d["bbox"][3]
                                matched[ti].append(False)
                                ignored[ti].append(a_d < lo or a_d > hi)
                    scores += [d["score"] for d in dts]
                if npos == 0:
                    continue
                rank = np.argsort([-x for x in scores], kind="mergesort")
                for ti in range(T):
                    tp = fp = 0
                    rc, pr = [], []
                    for i in rank:
                        if ignored[ti][i]:
                            continue
                        if matched[ti][i]:
                            tp += 1
                        else:
                            fp += 1
                        rc.append(tp / npos)
                        pr.append(tp / (tp + fp + np.spacing(1)))
                    recall[ti, k, ai, mi] = rc[-1] if rc else 0
                    for i in range(len(pr) - 1, 0, -1):
                        if pr[i] > pr[i - 1]:
                            pr[i - 1] = pr[i]
                    q = np.zeros(R)
                    for ri, pi in enumerate(np.searchsorted(rc, rec_thrs, side="left")):
                        if pi < len(pr):
                            q[ri] = pr[pi]
                    precision[ti, :, k, ai, mi] = q

    def mean(x):
        x = x[x > -1]
        return float(x.mean()) if x.size else -1.0
    t50, t75 = int(np.argmin(abs(np.asarray(iou_thrs) - 0.5))), int(np.argmin(abs(np.asarray(iou_thrs) - 0.75)))
    return [mean(precision[:, :, :, 0, 2]), mean(precision[t50, :, :, 0, 2]), mean(precision[t75, :, :, 0, 2]),
            mean(precision[:, :, :, 1, 2]), mean(precision[:, :, :, 2, 2]), mean(precision[:, :, :, 3, 2]),
            mean(recall[:, :, 0, 0]), mean(recall[:, :, 0, 1]), mean(recall[:, :, 0, 2]),
            mean(recall[:, :, 1, 2]), mean(recall[:, :, 2, 2]), mean(recall[:, :, 3, 2])]


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_cocoeval_random_vs_naive_restatement(seed):
    """The NumPy evaluator against the loop-by-loop restatement above on random data: 3 categories, 9 images (one without
    ground truth, one without detections), crowd boxes, tied scores, boxes of all three area ranges, more than 10
    detections of a category in an image (so maxDets = 1 / 10 / 100 differ)."""
    from radet_amd.datasets.cocoeval import COCO, COCOeval
    rng = np.random.RandomState(seed)
    images, anns, dets = list(range(1, 10)), [], []
    for img in images:
        for c in (1, 2, 3):
            n_gt = 0 if img == 4 else rng.randint(0, 5)
            for _ in range(n_gt):
                w, h = rng.choice([12, 40, 150]) * rng.uniform(0.8, 1.2), rng.choice([12, 40, 150]) * rng.uniform(0.8, 1.2)
                x, y = rng.uniform(0, 400), rng.uniform(0, 300)
                anns.append(dict(id=len(anns) + 1, image_id=img, category_id=c, bbox=[float(x), float(y), float(w), float(h)],
                                 area=float(w * h), iscrowd=int(rng.rand() < 0.15)))
            n_dt = 0 if img == 7 else rng.randint(0, 14)
            mine = [a for a in anns if a["image_id"] == img and a["category_id"] == c]
            for j in range(n_dt):
                if mine and rng.rand() < 0.6:                   # a jittered copy of a ground truth
                    g = mine[rng.randint(len(mine))]["bbox"]
                    b = [g[0] + rng.uniform(-8, 8), g[1] + rng.uniform(-8, 8), g[2] * rng.uniform(0.8, 1.2), g[3] * rng.uniform(0.8, 1.2)]
                else:
                    b = [rng.uniform(0, 400), rng.uniform(0, 300), rng.uniform(8, 160), rng.uniform(8, 160)]
                dets.append(dict(image_id=img, category_id=c, bbox=[float(v) for v in b],
                                 score=float(rng.choice([0.3, 0.5, 0.9]) if rng.rand() < 0.3 else rng.rand())))
    gt = COCO(dict(images=[dict(id=i, width=640, height=480, file_name=str(i)) for i in images],
                   categories=[dict(id=c, name=f"c{c}") for c in (1, 2, 3)], annotations=anns))
    ev = COCOeval(gt, gt.loadRes(dets), "bbox")
    ev.evaluate(); ev.accumulate(); ev.summarize()
    ref = _naive_coco_stats(images, anns, dets, [1, 2, 3], list(ev.params.iouThrs))
    assert np.allclose(ev.stats, ref, rtol=0, atol=1e-12), (list(ev.stats), ref)


def test_eval_recalls_vs_reference_golden():
    """radet_amd.datasets.cocoeval.eval_recalls against the reference's eval_recalls (tests/golden/recall.npz, written by
    tests/golden/gen_golden.py recall): ragged images, tied scores, exact-threshold IoUs; bit-equal recalls."""
    from radet_amd.datasets.cocoeval import eval_recalls, box_iou_matrix
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "recall.npz"))
    gts = [g[f"gt{i}"] for i in range(int(g["n"]))]
    prs = [g[f"pr{i}"] for i in range(int(g["n"]))]
    rec = eval_recalls(gts, prs, g["nums"], g["thrs"])
    assert rec.shape == g["recalls"].shape and np.array_equal(rec, g["recalls"])
    assert np.array_equal(eval_recalls(gts, prs, 10, 0.5), g["recalls_single"])
    assert box_iou_matrix(np.zeros((0, 4)), prs[1][:, :4]).shape == (0, prs[1].shape[0])
    iou = box_iou_matrix(gts[0], prs[0][:, :4])
    assert iou.dtype == np.float32 and iou[0, 0] == np.float32(0.5) and iou[0, 1] == np.float32(0.75)


def test_proposal_fast_metric(bop):
    from radet_amd.datasets import BOPDataset
    root, ann, coco = bop
    ds = BOPDataset(ann, pipeline=[], test_mode=True)
    ev = ds.evaluate(gt_as_results(ds, coco), metric="proposal_fast", logger="silent", proposal_nums=(1, 10, 100))
    assert set(ev) == {"AR@1", "AR@10", "AR@100"} and ev["AR@100"] == 1.0 and 0.0 < ev["AR@1"] < 1.0
    # pooled (k, 5) arrays per image are accepted as well (what the reference's RPN-style results look like)
    pooled = [np.concatenate(r, axis=0) for r in gt_as_results(ds, coco, jitter=6.0)]
    ev2 = ds.evaluate(pooled, metric=["proposal_fast"], logger="silent", proposal_nums=(100,), iou_thrs=[0.5, 0.95])
    assert 0.5 <= ev2["AR@100"] < 1.0
    with pytest.raises(NotImplementedError):
        ds.evaluate(pooled, metric="segm")


def test_cocoeval_hand_worked_case():
    """One image, one class, 2 gts, 3 detections in score order [TP (IoU 1.0), FP, TP (IoU 0.6)]:
    at IoU thresholds <= 0.6 the precision / recall points are (1, .5), (.5, .5), (2/3, 1): the envelope gives precision 1
    for recall thresholds 0..0.5 (51 of 101) and 2/3 above -> AP = (51 + 50 * 2/3) / 101; at thresholds > 0.6 only the
    first detection matches -> AP = 51 / 101, recall 0.5."""
    from radet_amd.datasets.cocoeval import COCO, COCOeval
    gt = COCO(dict(images=[dict(id=1, width=640, height=480, file_name="a")], categories=[dict(id=1, name="c")],
                   annotations=[dict(id=1, image_id=1, category_id=1, bbox=[10, 10, 100, 100], area=10000, iscrowd=0),
                                dict(id=2, image_id=1, category_id=1, bbox=[300, 200, 100, 50], area=5000, iscrowd=0)]))
    dt = gt.loadRes([dict(image_id=1, category_id=1, bbox=[10, 10, 100, 100], score=0.9),
                     dict(image_id=1, category_id=1, bbox=[500, 400, 50, 50], score=0.8),
                     dict(image_id=1, category_id=1, bbox=[300, 200, 60, 50], score=0.7)])      # IoU 3000 / 5000 = 0.6
    ev = COCOeval(gt, dt, "bbox")
    ev.evaluate(); ev.accumulate(); ev.summarize()
    ap_lo, ap_hi = (51 + 50 * 2 / 3) / 101, 51 / 101
    n_lo = int(np.sum(ev.params.iouThrs <= 0.6 + 1e-9))
    assert abs(ev.stats[1] - ap_lo) < 1e-12                               # AP50
    assert abs(ev.stats[2] - ap_hi) < 1e-12                               # AP75
    assert abs(ev.stats[0] - (n_lo * ap_lo + (10 - n_lo) * ap_hi) / 10) < 1e-12
    assert abs(ev.stats[8] - (n_lo * 1.0 + (10 - n_lo) * 0.5) / 10) < 1e-12     # AR@100
    assert abs(ev.stats[6] - 0.5) < 1e-12                                 # AR@1: only the best detection counts
    # area ranges: gt 1 (10000 px) is 'large', gt 2 (5000 px) 'medium'; the unmatched 2500-px FP is 'medium' too
    assert abs(ev.stats[5] - 1.0) < 1e-9 and ev.stats[3] == -1.0          # precision = tp / (tp + fp + eps), as pycocotools
    # a crowd gt absorbs detections without counting as FP or as a positive
    gt2 = COCO(dict(images=[dict(id=1, width=640, height=480, file_name="a")], categories=[dict(id=1, name="c")],
                    annotations=[dict(id=1, image_id=1, category_id=1, bbox=[10, 10, 100, 100], area=10000, iscrowd=0),
                                 dict(id=2, image_id=1, category_id=1, bbox=[300, 200, 200, 200], area=40000, iscrowd=1)]))
    dt2 = gt2.loadRes([dict(image_id=1, category_id=1, bbox=[10, 10, 100, 100], score=0.9),
                       dict(image_id=1, category_id=1, bbox=[320, 220, 50, 50], score=0.95),
                       dict(image_id=1, category_id=1, bbox=[350, 250, 60, 60], score=0.5)])
    ev2 = COCOeval(gt2, dt2, "bbox")
    ev2.evaluate(); ev2.accumulate(); ev2.summarize()
    assert abs(ev2.stats[0] - 1.0) < 1e-9 and ev2.stats[8] == 1.0
