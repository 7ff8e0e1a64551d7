"""BOP scene annotations -> COCO-style detector annotations (tools/bop_to_coco.py of the reference: scene_gt.json /
scene_gt_info.json per sequence, image list file, `bbox_obj` (modal, default) or `bbox_visib` (--amodal, the
reference's flag naming) boxes, `visib_fract` carried per annotation), and detections -> per-scene BOP files
(tools/coco_to_bop.py).  Host-only; the polygon extraction of `--segmentation` (skimage / shapely / cv2) is not
restated -- masks are read by path at training time (`BOPDataset.mask_path_template`)."""
import json
import os
from os import path as osp

CLASS_NAMES = dict(
    icbin=("coffee_cup", "juice_carton"),
    tudl=("dragon", "frog", "can"),
    lmo=("ape", "benchvise", "bowl", "cam", "can", "cat", "cup", "driller", "duck", "eggbox", "glue", "holepuncher", "iron",
         "lamp", "phone"),
    ycbv=("master_chef_can", "cracker_box", "sugar_box", "tomato_soup_can", "mustard_bottle", "tuna_fish_can", "pudding_box",
          "gelatin_box", "potted_meat_can", "banana", "pitcher_base", "bleach_cleanser", "bowl", "mug", "power_drill",
          "wood_block", "scissors", "large_marker", "large_clamp", "extra_large_clamp", "foam_brick"),
    hb=tuple(i + 1 for i in range(33)),
    itodd=tuple(i + 1 for i in range(28)),
    tless=tuple(i + 1 for i in range(30)),
)
IMAGE_RESOLUTION = dict(icbin=(640, 480), tudl=(640, 480), ycbv=(640, 480), lmo=(640, 480), hb=(640, 480),
                        itodd=(1280, 960), tless=(720, 540))


def scan_ids(sequence_dirs):
    """running (start, end] image / annotation id ranges per sequence (bop_to_coco.py:178-196)"""
    img_ranges, ann_ranges = [], []
    img0 = ann0 = 0
    for d in sequence_dirs:
        with open(osp.join(d, "scene_gt_info.json")) as f:
            info = json.load(f)
        img1, ann1 = img0 + len(info), ann0 + sum(len(v) for v in info.values())
        img_ranges.append((img0, img1))
        ann_ranges.append((ann0, ann1))
        img0, ann0 = img1, ann1
    return img_ranges, ann_ranges


def sequence_annotations(data_root, sequence_dir, ann_range, img_range, bbox_key="bbox_obj"):
    """bop_to_coco.py:99-175 without the polygon branch: {relative image path: dict(id, gts_info=[annotation, ...])}"""
    with open(osp.join(sequence_dir, "scene_gt_info.json")) as f:
        gt_info = json.load(f)
    with open(osp.join(sequence_dir, "scene_gt.json")) as f:
        gt = json.load(f)
    image_id, anno_id = img_range[0], ann_range[0]
    out = {}
    for key in gt_info.keys():
        image_id += 1
        rel = None
        for ext in ("jpg", "png"):
            p = osp.join(sequence_dir, "rgb", key.zfill(6) + "." + ext)
            if osp.exists(p):
                rel = osp.join(sequence_dir.split(data_root)[-1], "rgb", key.zfill(6) + "." + ext)[1:]
                break
        assert rel is not None, f"no rgb image for frame {key} of {sequence_dir}"
        per_img = []
        for info, obj in zip(gt_info[key], gt[key]):
            anno_id += 1
            box = info[bbox_key]
            per_img.append(dict(id=anno_id, image_id=image_id, category_id=obj["obj_id"], visib_fract=info["visib_fract"],
                                bbox=box, area=box[2] * box[3], iscrowd=0))
        out[rel] = dict(id=image_id, gts_info=per_img)
    assert anno_id == ann_range[1] and image_id == img_range[1]
    return out


def bop_to_coco(images_dir, images_list, dataset, amodal=False, without_gt=False):
    """Returns the COCO-style annotation dict for the images named in `images_list` (one relative path per line)."""
    names = CLASS_NAMES[dataset]
    w, h = IMAGE_RESOLUTION[dataset]
    categories = [dict(id=i + 1, name=n) for i, n in enumerate(names)]
    with open(images_list) as f:
        paths = f.read().split()
    if without_gt:                                                   # bop_to_coco.py:214-230 (test split, no gts)
        return dict(images=[dict(file_name=p, id=i, width=w, height=h) for i, p in enumerate(paths)], categories=categories)
    seqs = [osp.join(images_dir, s) for s in sorted(os.listdir(images_dir))]
    seqs = [s for s in seqs if osp.isdir(s)]
    img_ranges, ann_ranges = scan_ids(seqs)
    collected = {}
    for s, ir, ar in zip(seqs, img_ranges, ann_ranges):
        collected.update(sequence_annotations(images_dir, s, ar, ir, "bbox_visib" if amodal else "bbox_obj"))
    coco = dict(images=[], annotations=[], categories=categories)
    for p in paths:
        if p in collected:
            coco["images"].append(dict(file_name=p, id=collected[p]["id"], width=w, height=h))
            coco["annotations"].extend(collected[p]["gts_info"])
    return coco


def coco_to_bop(json_results, save_dir=None):
    """tools/coco_to_bop.py: BOP-COCO submission records (`BOPDataset(bop_submission=True)._det2json`: scene_id, image_id,
    category_id, bbox, score) -> {scene_id: {str(image_id): [dict(bbox_obj, obj_id, score), ...]}}; with `save_dir` each
    scene is also written to <save_dir>/<scene_id:06d>/scene_gt_info.json like the reference tool does."""
    converted = {}
    for r in json_results:
        scene = converted.setdefault(r["scene_id"], {})
        scene.setdefault(str(r["image_id"]), []).append(dict(bbox_obj=r["bbox"], obj_id=r["category_id"], score=r["score"]))
    if save_dir is not None:
        for scene_id, frames in converted.items():
            path = osp.join(save_dir, f"{scene_id:06d}", "scene_gt_info.json")
            os.makedirs(osp.dirname(path), exist_ok=True)
            with open(path, "w") as f:
                json.dump(frames, f)
    return converted
