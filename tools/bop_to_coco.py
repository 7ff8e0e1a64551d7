#!/usr/bin/env python
"""CLI of radet_amd.datasets.bop_convert.bop_to_coco (the reference's tools/bop_to_coco.py arguments)."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from radet_amd.datasets.bop_convert import CLASS_NAMES, bop_to_coco


def main():
    ap = argparse.ArgumentParser(description="Extract ground annotations from BOP format to COCO format")
    ap.add_argument("--images-dir", default="data/hb/train_pbr")
    ap.add_argument("--images-list", default="data/hb/image_lists/train_pbr.txt")
    ap.add_argument("--save-path", default="data/hb/detector_annotations/train_pbr.json")
    ap.add_argument("--segmentation", action="store_true")
    ap.add_argument("--without-gt", action="store_true")
    ap.add_argument("--amodal", action="store_true")
    ap.add_argument("--dataset", choices=sorted(CLASS_NAMES), required=True)
    a = ap.parse_args()
    if a.segmentation:
        raise SystemExit("--segmentation (polygon extraction with skimage / shapely) is not restated; masks are read by path")
    coco = bop_to_coco(a.images_dir, a.images_list, a.dataset, amodal=a.amodal, without_gt=a.without_gt)
    os.makedirs(os.path.dirname(a.save_path) or ".", exist_ok=True)
    with open(a.save_path, "w") as f:
        json.dump(coco, f)
    print(f"{len(coco['images'])} images, {len(coco.get('annotations', []))} annotations -> {a.save_path}")


if __name__ == "__main__":
    main()
