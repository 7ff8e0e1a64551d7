"""BOP detection dataset behind the reference's dataset interface (`radet/datasets/bop.py:13-302` and the parts of
`CocoDataset` / `CustomDataset` it inherits, `datasets/coco.py:47-362`, `datasets/custom.py:55-262`): same constructor
arguments, attributes (`data_infos`, `img_ids`, `cat_ids`, `cat2label`, `flag`, `CLASSES`), method names, annotation
dictionaries, result-file formats and metric names -- a config or a tool written for the reference keeps working.
The implementation is this package's own: one pass over the annotation file builds per-image NumPy tables, and filtering,
grouping, annotation parsing and result serialisation work on those tables.

Host-only Python; image decoding / augmentation stay outside the hot-path scope, so `pipeline` is a list of callables or of
configs of the registered stages (LabelAssignment, GenerateDistanceMap).  Evaluation runs on radet_amd.datasets.cocoeval
(a NumPy restatement of pycocotools' COCOeval, which is absent here; parity with pycocotools itself is unpinned)."""
import json
import logging
import os
import tempfile

import numpy as np

from ..utils import Registry, build_from_cfg
from .cocoeval import COCO, COCOeval, eval_recalls
from .pipelines import PIPELINES

osp = os.path
DATASETS = Registry("dataset")


def build_dataset(cfg, default_args=None):
    return build_from_cfg(cfg, DATASETS, default_args)


class Compose:
    """Stage list of a data pipeline (the reference's pipelines/compose.py contract): stages run in order, a stage that
    returns None drops the sample."""

    def __init__(self, transforms):
        self.transforms = [self._stage(t) for t in (transforms or [])]

    @staticmethod
    def _stage(t):
        if isinstance(t, dict):
            return build_from_cfg(t, PIPELINES)
        if callable(t):
            return t
        raise TypeError("transform must be callable or a dict")

    def __call__(self, data):
        for stage in self.transforms:
            if data is None:
                break
            data = stage(data)
        return data


def _log(msg, logger=None, level=logging.INFO):
    if logger == "silent":
        return
    if isinstance(logger, logging.Logger):
        logger.log(level, msg)
    else:
        print(msg)


YCBV_CLASSES = ("master_chef_can", "cracker_box", "sugar_box", "tomato_soup_can", "mustard_bottle", "tuna_fish_can",
                "pudding_box", "gelatin_box", "potted_meat_can", "banana", "pitcher_base", "bleach_cleanser", "bowl", "mug",
                "power_drill", "wood_block", "scissors", "large_marker", "large_clamp", "extra_large_clamp", "foam_brick")

# COCO summary slots behind the reference's metric names
_STAT_SLOT = dict(zip(("mAP", "mAP_50", "mAP_75", "mAP_s", "mAP_m", "mAP_l", "AR@1", "AR@10", "AR@100", "AR_s@100",
                              "AR_m@100", "AR_l@100"), range(12)))
_PROPOSAL_ITEMS = ("AR@1", "AR@10", "AR@100", "AR_s@100", "AR_m@100", "AR_l@100")


def _under(root, path):
    """`path` relative to `root` unless it is absolute / empty-by-design (None)"""
    return path if (root is None or path is None or osp.isabs(path)) else osp.join(root, path)


def _bop_frame(filename):
    """('000048', 12) from '<...>/000048/rgb/000012.jpg': scene directory and frame number of a BOP image path"""
    scene, _, leaf = filename.rsplit("/", 3)[-3:]
    return scene, int(osp.splitext(leaf)[0])


@DATASETS.register_module()
class BOPDataset:
    CLASSES = YCBV_CLASSES
    mask_path_template = "{:06d}/mask_visib/{:06}_{:06}.png"

    def __init__(self, ann_file, pipeline, classes=None, data_root=None, img_prefix="", bop_submission=False,
                 seg_prefix=None, proposal_file=None, test_mode=False, min_visib_frac=0., filter_empty_gt=True):
        self.data_root, self.test_mode, self.filter_empty_gt = data_root, test_mode, filter_empty_gt
        self.ann_file = _under(data_root, ann_file)
        self.img_prefix = _under(data_root, img_prefix)
        self.seg_prefix = _under(data_root, seg_prefix)
        self.proposal_file, self.proposals = proposal_file, None
        self.min_visib_fract = min_visib_frac
        self.bop_submission = bop_submission
        self.CLASSES = self.get_classes(classes)
        self.data_infos = self.load_annotations(self.ann_file)
        if not test_mode:                                    # training: usable images only, grouped by aspect ratio
            keep = self._filter_imgs()
            self.data_infos = [self.data_infos[i] for i in keep]
            self._set_group_flag()
        self.pipeline = Compose(pipeline)
        if bop_submission:
            self._det2json = self._bop_det2json

    # ------------------------------------------------------------------ class list / annotation index
    @classmethod
    def get_classes(cls, classes=None):
        if classes is None:
            return cls.CLASSES
        if isinstance(classes, str):                         # a text file, one class name per line
            with open(classes) as f:
                return [ln.strip() for ln in f if ln.strip()]
        if isinstance(classes, (tuple, list)):
            return classes
        raise ValueError(f"Unsupported type {type(classes)} of classes.")

    def __len__(self):
        return len(self.data_infos)

    def load_annotations(self, ann_file):
        coco = self.coco = COCO(ann_file)
        self.cat_ids = coco.get_cat_ids(cat_names=self.CLASSES)
        self.cat2label = dict(zip(self.cat_ids, range(len(self.cat_ids))))
        self.img_ids = coco.get_img_ids()
        infos = coco.load_imgs(self.img_ids)
        for info in infos:
            info["filename"] = info["file_name"]
        return infos

    def _anns_of(self, idx):
        return self.coco.load_anns(self.coco.get_ann_ids(img_ids=[self.data_infos[idx]["id"]]))

    def get_ann_info(self, idx):
        return self._parse_ann_info(self.data_infos[idx], self._anns_of(idx))

    def get_cat_ids(self, idx):
        return [a["category_id"] for a in self._anns_of(idx)]

    def _filter_imgs(self, min_size=32):
        """Indices of the images a training run can use: both sides >= min_size and, with filter_empty_gt, at least one
        annotation of a wanted class.  Narrows `img_ids` to the survivors (same order)."""
        sizes = np.array([[i["width"], i["height"]] for i in self.data_infos], dtype=np.int64).reshape(-1, 2)
        ok = sizes.min(axis=1) >= min_size
        if self.filter_empty_gt:
            wanted = set(self.cat_ids)
            annotated = {a["image_id"] for a in self.coco.anns.values() if a["category_id"] in wanted}
            ok &= np.fromiter((i in annotated for i in self.img_ids), dtype=bool, count=len(self.img_ids))
        keep = np.flatnonzero(ok).tolist()
        self.img_ids = [self.img_ids[i] for i in keep]
        return keep

    def _set_group_flag(self):
        """flag[i] = 1 for landscape images (width > height): the group sampler batches within an aspect-ratio group"""
        wh = np.array([[i["width"], i["height"]] for i in self.data_infos], dtype=np.float64).reshape(-1, 2)
        self.flag = (wh[:, 0] > wh[:, 1]).astype(np.uint8)

    def _rand_another(self, idx):
        return np.random.choice(np.flatnonzero(self.flag == self.flag[idx]))

    # ------------------------------------------------------------------ samples
    def pre_pipeline(self, results):
        results.update(img_prefix=self.img_prefix, seg_prefix=self.seg_prefix, proposal_file=self.proposal_file,
                       bbox_fields=[], mask_fields=[], seg_fields=[])

    def _sample(self, idx, with_ann):
        results = dict(img_info=self.data_infos[idx])
        if with_ann:
            results["ann_info"] = self.get_ann_info(idx)
        self.pre_pipeline(results)
        return self.pipeline(results)

    def prepare_train_img(self, idx):
        return self._sample(idx, True)

    def prepare_test_img(self, idx):
        return self._sample(idx, False)

    def __getitem__(self, idx):
        sample = self._sample(idx, not self.test_mode)
        while sample is None and not self.test_mode:          # the pipeline dropped it: another image of the same group
            sample = self._sample(self._rand_another(idx), True)
        return sample

    # ------------------------------------------------------------------ annotations -> training targets
    def _parse_ann_info(self, img_info, ann_info):
        """COCO records of one image -> dict(bboxes f32[n,4] xyxy, labels i64[n], bboxes_ignore f32[m,4], masks [n paths],
        seg_map).  Dropped: `ignore` records, boxes outside the image, degenerate boxes (area <= 0 or a side < 1 px),
        unwanted classes.  Objects whose visible fraction is below `min_visib_frac` become ignore boxes.  The mask of the
        i-th record of the frame is '<scene>/mask_visib/<frame>_<i>.png' (i counts ALL records: the BOP file layout)."""
        scene, frame = _bop_frame(img_info["filename"])
        n = len(ann_info)
        box = np.array([a["bbox"] for a in ann_info], dtype=np.float64).reshape(n, 4)
        x1, y1, w, h = box.T
        x2, y2 = x1 + w, y1 + h
        inside = (np.minimum(x2, img_info["width"]) - np.maximum(x1, 0)).clip(min=0) * \
                 (np.minimum(y2, img_info["height"]) - np.maximum(y1, 0)).clip(min=0)
        usable = np.array([not a.get("ignore", False) and a["area"] > 0 and a["category_id"] in self.cat2label
                           for a in ann_info], dtype=bool).reshape(n)
        usable &= (inside != 0) & (w >= 1) & (h >= 1)
        visible = np.array([a["visib_fract"] >= self.min_visib_fract for a in ann_info], dtype=bool).reshape(n)
        xyxy = np.stack([x1, y1, x2, y2], axis=1).astype(np.float32)
        pos = np.flatnonzero(usable & visible)
        return dict(bboxes=xyxy[pos].reshape(-1, 4),
                    labels=np.array([self.cat2label[ann_info[i]["category_id"]] for i in pos], dtype=np.int64),
                    bboxes_ignore=xyxy[usable & ~visible].reshape(-1, 4),
                    masks=[self.mask_path_template.format(int(scene), frame, int(i)) for i in pos],
                    seg_map=img_info["filename"].replace("jpg", "png"))

    # ------------------------------------------------------------------ detections -> result files
    def xyxy2xywh(self, bbox):
        x1, y1, x2, y2 = (float(v) for v in np.asarray(bbox)[:4])
        return [x1, y1, x2 - x1, y2 - y1]

    def _records(self, results, head):
        """one dict per detection: `head(idx)` (the image's identifying fields) + category_id, bbox (xywh), score"""
        out = []
        for idx, per_class in enumerate(results):
            ident = head(idx)
            for label, dets in enumerate(per_class):
                for det in np.asarray(dets).reshape(-1, 5):
                    out.append(dict(ident, category_id=self.cat_ids[label], bbox=self.xyxy2xywh(det), score=float(det[4])))
        return out

    def _det2json(self, results):
        """per-class [k, 5] arrays per image -> COCO result records"""
        return self._records(results[:len(self)], lambda idx: dict(image_id=self.img_ids[idx]))

    def _bop_det2json(self, results):
        """BOP-COCO submission records: scene_id / image_id from the BOP file layout, time = -1"""
        def head(idx):
            scene, frame = _bop_frame(self.data_infos[idx]["filename"])
            return dict(scene_id=int(scene), image_id=frame, time=-1.0)
        return self._records(results[:len(self)], head)

    def results2json(self, results, outfile_prefix):
        if not isinstance(results[0], list):
            raise TypeError("invalid type of results (the RADet detector produces per-class box lists)")
        path = f"{outfile_prefix}.bbox.json"
        with open(path, "w") as f:
            json.dump(self._det2json(results), f)
        return dict(bbox=path, proposal=path)

    def format_results(self, results, jsonfile_prefix=None, **kwargs):
        if not isinstance(results, list):
            raise AssertionError("results must be a list")
        if len(results) != len(self):
            raise AssertionError(f"The length of results is not equal to the dataset len: {len(results)} != {len(self)}")
        scratch = tempfile.TemporaryDirectory() if jsonfile_prefix is None else None
        prefix = jsonfile_prefix if scratch is None else osp.join(scratch.name, "results")
        return self.results2json(results, prefix), scratch

    # ------------------------------------------------------------------ evaluation
    @staticmethod
    def _check_metrics(metric, metric_items):
        metrics = list(metric) if isinstance(metric, (list, tuple)) else [metric]
        for m in metrics:
            if m == "segm":
                raise NotImplementedError("metric segm is outside the detector hot-path scope (the detector outputs boxes only)")
            if m not in ("bbox", "proposal", "proposal_fast"):
                raise KeyError(f"metric {m} is not supported")
        items = None if metric_items is None else (list(metric_items) if isinstance(metric_items, (list, tuple))
                                                   else [metric_items])
        for item in items or ():
            if item not in _STAT_SLOT:
                raise KeyError(f"metric item {item} is not supported")
        return metrics, items

    def _classwise(self, precision, logger):
        """AP per category (all IoU thresholds, all areas, the largest detection budget) + a 3-pairs-per-row table"""
        rows = []
        for k, cat_id in enumerate(self.cat_ids):
            p = precision[:, :, k, 0, -1]
            p = p[p > -1]
            rows.append((self.coco.loadCats(cat_id)[0]["name"], f"{p.mean() if p.size else float('nan'):0.3f}"))
        per_row = min(3, len(rows))
        lines = [" | ".join(f"{c:>18s}" for c in ("category", "AP") * per_row)]
        for i in range(0, len(rows), per_row):
            lines.append(" | ".join(f"{c:>18s}" for pair in rows[i:i + per_row] for c in pair))
        _log("\n" + "\n".join(lines), logger)
        return rows

    def fast_eval_recall(self, results, proposal_nums, iou_thrs, logger=None):
        """Average recall (over `iou_thrs`) of the non-crowd, non-ignored ground-truth boxes by the top-N boxes of every image,
        straight from the result arrays (coco.py:311-333).  An image's result is an (k, 5) / (k, 4) array, or the detector's
        per-class list, whose classes are pooled."""
        gts = []
        for img_id in self.img_ids:
            anns = self.coco.load_anns(self.coco.get_ann_ids(img_ids=[img_id]))
            keep = [a["bbox"] for a in anns if not a.get("ignore", False) and not a["iscrowd"]]
            xywh = np.asarray(keep, np.float64).reshape(-1, 4)          # (corners formed in double, then stored as fp32)
            gts.append(np.concatenate([xywh[:, :2], xywh[:, :2] + xywh[:, 2:]], axis=1).astype(np.float32) if len(keep)
                       else np.zeros((0, 4)))
        pooled = [np.concatenate([np.asarray(c).reshape(-1, 5) for c in r], axis=0) if isinstance(r, (list, tuple)) else np.asarray(r)
                  for r in results]
        return eval_recalls(gts, pooled, proposal_nums, iou_thrs).mean(axis=1)

    def evaluate(self, results, metric="bbox", logger=None, jsonfile_prefix=None, classwise=False,
                 proposal_nums=(1, 10, 100), iou_thrs=None, metric_items=None):
        """COCO-protocol evaluation of per-class box lists; returns e.g. {'bbox_mAP': .., 'bbox_mAP_50': .., ...,
        'bbox_mAP_copypaste': '...'} ('proposal': class-agnostic recall, keys 'AR@1' ...).  An empty result set gives {}."""
        metrics, items = self._check_metrics(metric, metric_items)
        if self.bop_submission:
            raise RuntimeError("bop_submission=True formats BOP submission files (scene_id / image_id records); "
                               "evaluate with bop_submission=False")
        if iou_thrs is None:
            iou_thrs = np.linspace(0.5, 0.95, 10)
        files, scratch = (None, None) if all(m == "proposal_fast" for m in metrics) else self.format_results(results, jsonfile_prefix)
        out = {}
        try:
            for m in metrics:
                _log(("\n" if logger is None else "") + f"Evaluating {m}...", logger)
                if m == "proposal_fast":
                    ar = self.fast_eval_recall(results, proposal_nums, iou_thrs)
                    out.update((f"AR@{n}", ar[i]) for i, n in enumerate(proposal_nums))
                    _log("".join(f"\nAR@{n}\t{ar[i]:.4f}" for i, n in enumerate(proposal_nums)), logger)
                    continue
                try:
                    detections = self.coco.loadRes(files[m])
                except IndexError:
                    _log("The testing results of the whole dataset is empty.", logger, logging.ERROR)
                    break
                ev = COCOeval(self.coco, detections, "bbox")
                ev.params.catIds, ev.params.imgIds = self.cat_ids, self.img_ids
                ev.params.maxDets, ev.params.iouThrs = list(proposal_nums), iou_thrs
                ev.params.useCats = 0 if m == "proposal" else ev.params.useCats
                ev.evaluate()
                ev.accumulate()
                ev.summarize()
                stat = lambda name: float(f"{ev.stats[_STAT_SLOT[name]]:.3f}")  # noqa: E731
                if m == "proposal":
                    out.update((name, stat(name)) for name in (items or _PROPOSAL_ITEMS))
                    continue
                if classwise:
                    assert ev.eval["precision"].shape[2] == len(self.cat_ids)
                    out["classwise"] = self._classwise(ev.eval["precision"], logger)
                out.update((f"{m}_{name}", stat(name)) for name in (items or _STAT_SLOT))
                out[f"{m}_mAP_copypaste"] = " ".join(f"{v:.3f}" for v in ev.stats[:6])
        finally:
            if scratch:
                scratch.cleanup()
        return out


@DATASETS.register_module()
class YcbvDataset(BOPDataset):
    """The reference's `YcbvDataset` (radet/datasets/ycbv.py:5-12: a COCO dataset with the 21 YCB-V classes): plain COCO
    files work here too -- missing BOP fields default to "fully visible", and an image path without the scene / frame
    structure is parsed as frame <digits of its name> of scene 0."""
    CLASSES = YCBV_CLASSES

    def _parse_ann_info(self, img_info, ann_info):
        for ann in ann_info:
            ann.setdefault("visib_fract", 1.0)
        name = img_info["filename"]
        if name.count("/") < 2:
            digits = "".join(ch for ch in osp.basename(name) if ch.isdigit() or ch == ".")
            out = super()._parse_ann_info(dict(img_info, filename="0/rgb/" + digits), ann_info)
            out["seg_map"] = name.replace("jpg", "png")
            return out
        return super()._parse_ann_info(img_info, ann_info)
