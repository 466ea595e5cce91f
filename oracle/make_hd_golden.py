#!/usr/bin/env python3
"""Mint ``tests/golden/hd_grid.json``: the crop-grid choices of the REAL reference helper
``Image_Patch.calculate`` (reference ``llava/patch_divide.py:71-104``) on a sweep of image sizes.

Run in the build container only (``/root/reference`` does not exist on the GPU box):

    python oracle/make_hd_golden.py

The reference file is imported by path; its one missing dependency (``torchvision.ops.boxes.box_area``,
absent here) is stubbed with the published definition (x2-x1)*(y2-y1).  Nothing is copied from it.
"""
import importlib.util
import json
import os
import random
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_FILE = "/root/reference/llava/patch_divide.py"


def load_reference():
    tv = types.ModuleType("torchvision"); ops = types.ModuleType("torchvision.ops"); boxes = types.ModuleType("torchvision.ops.boxes")
    boxes.box_area = lambda b: (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    sys.modules.setdefault("torchvision", tv); sys.modules.setdefault("torchvision.ops", ops)
    sys.modules.setdefault("torchvision.ops.boxes", boxes)
    spec = importlib.util.spec_from_file_location("ref_patch_divide", REF_FILE)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    ref = load_reference()
    rng = random.Random(20260926)
    sizes = [(336, 336), (672, 672), (1088, 1088), (1008, 1008), (336, 672), (672, 336), (337, 335), (100, 3000),
             (3000, 100), (480, 640), (640, 480), (1080, 1920), (1920, 1080), (768, 1024), (2000, 2000), (50, 50)]
    sizes += [(rng.randint(40, 2600), rng.randint(40, 2600)) for _ in range(240)]
    sizes += [(s, s) for s in range(200, 2400, 97)]                       # squares: ties between (a,b) and (b,a)
    out = {"sizes": sizes, "choices": {}, "candidates": {}}
    for patch_num in (9, 16, 25):
        ip = ref.Image_Patch(image_size=336, patch_num=patch_num)
        out["choices"][str(patch_num)] = [list(ip.calculate(h, w)) for h, w in sizes]
        out["candidates"][str(patch_num)] = len(ip.patch_list)
    path = os.path.join(ROOT, "tests", "golden", "hd_grid.json")
    with open(path, "w") as f:
        json.dump(out, f)
    print("wrote", path, {k: len(v) for k, v in out["choices"].items()})


if __name__ == "__main__":
    main()
