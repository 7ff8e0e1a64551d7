"""BOP detection dataset with the reference's interface (radet/datasets/bop.py:13-302 on top of the parts of
CocoDataset / CustomDataset it inherits: datasets/coco.py:47-362, datasets/custom.py:55-262): annotation parsing
(visibility fraction -> ignore boxes, mask paths of the BOP layout), detections -> COCO / BOP submission json, and the
COCO-protocol evaluation.  Host-only Python; image decoding / augmentation stay outside the hot-path scope, so `pipeline`
is a list of callables or of configs of the registered stages (LabelAssignment, GenerateDistanceMap).

Evaluation runs on radet_amd.datasets.cocoeval (a NumPy restatement of pycocotools' COCOeval, which is absent here)."""
import itertools
import json
import logging
import os.path as osp
import tempfile
from collections import OrderedDict

import numpy as np

from ..utils import Registry, build_from_cfg
from .cocoeval import COCO, COCOeval
from .pipelines import PIPELINES

DATASETS = Registry("dataset")


def build_dataset(cfg, default_args=None):
    return build_from_cfg(cfg, DATASETS, default_args)


class Compose:
    """pipelines/compose.py: run the stages in order; a stage returning None drops the sample."""

    def __init__(self, transforms):
        self.transforms = []
        for t in transforms or []:
            if isinstance(t, dict):
                t = build_from_cfg(t, PIPELINES)
            elif not callable(t):
                raise TypeError("transform must be callable or a dict")
            self.transforms.append(t)

    def __call__(self, data):
        for t in self.transforms:
            data = t(data)
            if data is None:
                return None
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


@DATASETS.register_module()
class BOPDataset:
    CLASSES = YCBV_CLASSES

    def __init__(self, ann_file, pipeline, classes=None, data_root=None, img_prefix="", bop_submission=False,
                 seg_prefix=None, proposal_file=None, test_mode=False, min_visib_frac=0., filter_empty_gt=True):
        self.ann_file, self.data_root, self.img_prefix, self.seg_prefix = ann_file, data_root, img_prefix, seg_prefix
        self.proposal_file, self.test_mode, self.filter_empty_gt = proposal_file, test_mode, filter_empty_gt
        self.CLASSES = self.get_classes(classes)
        if self.data_root is not None:                               # custom.py:75-86
            if not osp.isabs(self.ann_file):
                self.ann_file = osp.join(self.data_root, self.ann_file)
            if not (self.img_prefix is None or osp.isabs(self.img_prefix)):
                self.img_prefix = osp.join(self.data_root, self.img_prefix)
            if not (self.seg_prefix is None or osp.isabs(self.seg_prefix)):
                self.seg_prefix = osp.join(self.data_root, self.seg_prefix)
        self.data_infos = self.load_annotations(self.ann_file)
        self.proposals = None
        if not test_mode:
            valid_inds = self._filter_imgs()
            self.data_infos = [self.data_infos[i] for i in valid_inds]
            self._set_group_flag()
        self.pipeline = Compose(pipeline)
        self.min_visib_fract = min_visib_frac
        self.mask_path_template = "{:06d}/mask_visib/{:06}_{:06}.png"
        self.bop_submission = bop_submission
        if self.bop_submission:
            self._det2json = self._bop_det2json

    # ------------------------------------------------------------------ CustomDataset / CocoDataset parts
    @classmethod
    def get_classes(cls, classes=None):
        if classes is None:
            return cls.CLASSES
        if isinstance(classes, str):
            with open(classes) as f:
                return [line.strip() for line in f if line.strip()]
        if isinstance(classes, (tuple, list)):
            return classes
        raise ValueError(f"Unsupported type {type(classes)} of classes.")

    def __len__(self):
        return len(self.data_infos)

    def load_annotations(self, ann_file):
        self.coco = COCO(ann_file)
        self.cat_ids = self.coco.get_cat_ids(cat_names=self.CLASSES)
        self.cat2label = {cat_id: i for i, cat_id in enumerate(self.cat_ids)}
        self.img_ids = self.coco.get_img_ids()
        data_infos = []
        for i in self.img_ids:
            info = self.coco.load_imgs([i])[0]
            info["filename"] = info["file_name"]
            data_infos.append(info)
        return data_infos

    def get_ann_info(self, idx):
        img_id = self.data_infos[idx]["id"]
        ann_info = self.coco.load_anns(self.coco.get_ann_ids(img_ids=[img_id]))
        return self._parse_ann_info(self.data_infos[idx], ann_info)

    def get_cat_ids(self, idx):
        img_id = self.data_infos[idx]["id"]
        return [ann["category_id"] for ann in self.coco.load_anns(self.coco.get_ann_ids(img_ids=[img_id]))]

    def _filter_imgs(self, min_size=32):
        """coco.py:98-120: drop images that are too small or (filter_empty_gt) carry no annotation of a wanted class"""
        valid_inds = []
        ids_with_ann = set(a["image_id"] for a in self.coco.anns.values())
        ids_in_cat = set()
        for class_id in self.cat_ids:
            ids_in_cat |= set(self.coco.cat_img_map[class_id])
        ids_in_cat &= ids_with_ann
        valid_img_ids = []
        for i, img_info in enumerate(self.data_infos):
            img_id = self.img_ids[i]
            if self.filter_empty_gt and img_id not in ids_in_cat:
                continue
            if min(img_info["width"], img_info["height"]) >= min_size:
                valid_inds.append(i)
                valid_img_ids.append(img_id)
        self.img_ids = valid_img_ids
        return valid_inds

    def _set_group_flag(self):
        self.flag = np.zeros(len(self), dtype=np.uint8)
        for i in range(len(self)):
            if self.data_infos[i]["width"] / self.data_infos[i]["height"] > 1:
                self.flag[i] = 1

    def _rand_another(self, idx):
        pool = np.where(self.flag == self.flag[idx])[0]
        return np.random.choice(pool)

    def pre_pipeline(self, results):
        results["img_prefix"] = self.img_prefix
        results["seg_prefix"] = self.seg_prefix
        results["proposal_file"] = self.proposal_file
        results["bbox_fields"], results["mask_fields"], results["seg_fields"] = [], [], []

    def prepare_train_img(self, idx):
        results = dict(img_info=self.data_infos[idx], ann_info=self.get_ann_info(idx))
        self.pre_pipeline(results)
        return self.pipeline(results)

    def prepare_test_img(self, idx):
        results = dict(img_info=self.data_infos[idx])
        self.pre_pipeline(results)
        return self.pipeline(results)

    def __getitem__(self, idx):
        if self.test_mode:
            return self.prepare_test_img(idx)
        while True:
            data = self.prepare_train_img(idx)
            if data is None:
                idx = self._rand_another(idx)
                continue
            return data

    # ------------------------------------------------------------------ bop.py:43-118
    def _parse_ann_info(self, img_info, ann_info):
        gt_bboxes, gt_labels, gt_bboxes_ignore, gt_masks_ann = [], [], [], []
        filename = img_info["filename"]
        seq_name, _, img_name = filename.rsplit("/", 3)
        img_id = int(osp.splitext(img_name)[0])
        for i, ann in enumerate(ann_info):
            if ann.get("ignore", False):
                continue
            x1, y1, w, h = ann["bbox"]
            inter_w = max(0, min(x1 + w, img_info["width"]) - max(x1, 0))
            inter_h = max(0, min(y1 + h, img_info["height"]) - max(y1, 0))
            mask_path = self.mask_path_template.format(int(seq_name), img_id, i)
            if inter_w * inter_h == 0:
                continue
            if ann["area"] <= 0 or w < 1 or h < 1:
                continue
            if ann["category_id"] not in self.cat_ids:
                continue
            bbox = [x1, y1, x1 + w, y1 + h]
            if ann["visib_fract"] < self.min_visib_fract:
                gt_bboxes_ignore.append(bbox)
            else:
                gt_bboxes.append(bbox)
                gt_labels.append(self.cat2label[ann["category_id"]])
                gt_masks_ann.append(mask_path)
        if gt_bboxes:
            gt_bboxes = np.array(gt_bboxes, dtype=np.float32)
            gt_labels = np.array(gt_labels, dtype=np.int64)
        else:
            gt_bboxes = np.zeros((0, 4), dtype=np.float32)
            gt_labels = np.array([], dtype=np.int64)
        gt_bboxes_ignore = (np.array(gt_bboxes_ignore, dtype=np.float32) if gt_bboxes_ignore
                            else np.zeros((0, 4), dtype=np.float32))
        seg_map = img_info["filename"].replace("jpg", "png")
        return dict(bboxes=gt_bboxes, labels=gt_labels, bboxes_ignore=gt_bboxes_ignore, masks=gt_masks_ann, seg_map=seg_map)

    def xyxy2xywh(self, bbox):
        b = bbox.tolist()
        return [b[0], b[1], b[2] - b[0], b[3] - b[1]]

    def _det2json(self, results):
        """coco.py:216-231: per-class [k, 5] arrays -> COCO result dicts"""
        json_results = []
        for idx in range(len(self)):
            img_id = self.img_ids[idx]
            result = results[idx]
            for label in range(len(result)):
                bboxes = result[label]
                for i in range(bboxes.shape[0]):
                    json_results.append(dict(image_id=img_id, bbox=self.xyxy2xywh(bboxes[i]), score=float(bboxes[i][4]),
                                             category_id=self.cat_ids[label]))
        return json_results

    def _bop_det2json(self, results):
        """bop.py:98-118: BOP-COCO submission records (scene_id / image_id from the BOP file layout, time = -1)"""
        json_results = []
        for idx in range(len(self)):
            filename = self.data_infos[idx]["filename"]
            scene_id, _, img_name = filename.rsplit("/", 3)
            result = results[idx]
            for label in range(len(result)):
                bboxes = result[label]
                for i in range(bboxes.shape[0]):
                    json_results.append(dict(scene_id=int(scene_id), image_id=int(img_name.split(".")[0]),
                                             category_id=self.cat_ids[label], bbox=self.xyxy2xywh(bboxes[i]),
                                             score=float(bboxes[i][4]), time=-1.0))
        return json_results

    def results2json(self, results, outfile_prefix):
        result_files = dict()
        if isinstance(results[0], list):
            json_results = self._det2json(results)
            result_files["bbox"] = f"{outfile_prefix}.bbox.json"
            result_files["proposal"] = f"{outfile_prefix}.bbox.json"
            with open(result_files["bbox"], "w") as f:
                json.dump(json_results, f)
        else:
            raise TypeError("invalid type of results (the RADet detector produces per-class box lists)")
        return result_files

    def format_results(self, results, jsonfile_prefix=None, **kwargs):
        assert isinstance(results, list), "results must be a list"
        assert len(results) == len(self), (
            "The length of results is not equal to the dataset len: {} != {}".format(len(results), len(self)))
        if jsonfile_prefix is None:
            tmp_dir = tempfile.TemporaryDirectory()
            jsonfile_prefix = osp.join(tmp_dir.name, "results")
        else:
            tmp_dir = None
        return self.results2json(results, jsonfile_prefix), tmp_dir

    # ------------------------------------------------------------------ bop.py:120-302
    def evaluate(self, results, metric="bbox", logger=None, jsonfile_prefix=None, classwise=False,
                 proposal_nums=(1, 10, 100), iou_thrs=None, metric_items=None):
        """COCO-protocol evaluation; returns e.g. {'bbox_mAP': .., 'bbox_mAP_50': .., ..., 'bbox_mAP_copypaste': '...'}"""
        metrics = metric if isinstance(metric, list) else [metric]
        for m in metrics:
            if m not in ("bbox", "proposal"):
                if m in ("segm", "proposal_fast"):
                    raise NotImplementedError(f"metric {m} is outside the detector hot-path scope (boxes only)")
                raise KeyError(f"metric {m} is not supported")
        if self.bop_submission:
            raise RuntimeError("bop_submission=True formats BOP submission files (scene_id / image_id records); "
                               "evaluate with bop_submission=False")
        if iou_thrs is None:
            iou_thrs = np.linspace(.5, 0.95, int(np.round((0.95 - .5) / .05)) + 1, endpoint=True)
        if metric_items is not None and not isinstance(metric_items, list):
            metric_items = [metric_items]
        result_files, tmp_dir = self.format_results(results, jsonfile_prefix)
        eval_results = OrderedDict()
        cocoGt = self.coco
        names = {"mAP": 0, "mAP_50": 1, "mAP_75": 2, "mAP_s": 3, "mAP_m": 4, "mAP_l": 5, "AR@1": 6, "AR@10": 7,
                 "AR@100": 8, "AR_s@100": 9, "AR_m@100": 10, "AR_l@100": 11}
        for metric in metrics:
            _log(("\n" if logger is None else "") + f"Evaluating {metric}...", logger)
            try:
                cocoDt = cocoGt.loadRes(result_files[metric])
            except IndexError:
                _log("The testing results of the whole dataset is empty.", logger, logging.ERROR)
                break
            ev = COCOeval(cocoGt, cocoDt, "bbox")
            ev.params.catIds = self.cat_ids
            ev.params.imgIds = self.img_ids
            ev.params.maxDets = list(proposal_nums)
            ev.params.iouThrs = iou_thrs
            if metric_items is not None:
                for item in metric_items:
                    if item not in names:
                        raise KeyError(f"metric item {item} is not supported")
            if metric == "proposal":
                ev.params.useCats = 0
            ev.evaluate()
            ev.accumulate()
            ev.summarize()
            if metric == "proposal":
                items = metric_items or ["AR@1", "AR@10", "AR@100", "AR_s@100", "AR_m@100", "AR_l@100"]
                for item in items:
                    eval_results[item] = float(f"{ev.stats[names[item]]:.3f}")
                continue
            if classwise:
                precisions = ev.eval["precision"]                    # (iou, recall, cls, area range, max dets)
                assert len(self.cat_ids) == precisions.shape[2]
                rows = []
                for idx, cat_id in enumerate(self.cat_ids):
                    nm = self.coco.loadCats(cat_id)[0]
                    pr = precisions[:, :, idx, 0, -1]
                    pr = pr[pr > -1]
                    rows.append((f'{nm["name"]}', f"{float(np.mean(pr)) if pr.size else float('nan'):0.3f}"))
                eval_results["classwise"] = rows
                ncol = min(6, len(rows) * 2)
                flat = list(itertools.chain(*rows))
                table = [["category", "AP"] * (ncol // 2)] + [list(r) for r in itertools.zip_longest(
                    *[flat[i::ncol] for i in range(ncol)], fillvalue="")]
                _log("\n" + "\n".join(" | ".join(f"{c:>18s}" for c in row) for row in table), logger)
            items = metric_items or ["mAP", "mAP_50", "mAP_75", "mAP_s", "mAP_m", "mAP_l", "AR@1", "AR@10", "AR@100",
                                     "AR_s@100", "AR_m@100", "AR_l@100"]
            for item in items:
                eval_results[f"{metric}_{item}"] = float(f"{ev.stats[names[item]]:.3f}")
            ap = ev.stats[:6]
            eval_results[f"{metric}_mAP_copypaste"] = (f"{ap[0]:.3f} {ap[1]:.3f} {ap[2]:.3f} {ap[3]:.3f} "
                                                      f"{ap[4]:.3f} {ap[5]:.3f}")
        if tmp_dir is not None:
            tmp_dir.cleanup()
        return eval_results


@DATASETS.register_module()
class YcbvDataset(BOPDataset):
    """radet/datasets/ycbv.py:5-12 (a CocoDataset with the 21 YCB-V classes); the BOP annotation fields are optional here"""
    CLASSES = YCBV_CLASSES

    def _parse_ann_info(self, img_info, ann_info):
        for ann in ann_info:
            ann.setdefault("visib_fract", 1.0)
        if img_info["filename"].count("/") < 2:                  # plain COCO layout: no scene / frame structure
            img_info = dict(img_info, filename="0/rgb/" + "".join(c for c in osp.basename(img_info["filename"]) if c.isdigit() or c == ".") )
        return super()._parse_ann_info(img_info, ann_info)
