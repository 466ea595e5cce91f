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


# ------------------------------------------------------------------------------------------------------------------
# HD image slicing: golden crops from the reference's OWN data-loader code.  The code sits inside
# LazySupervisedDataset.__getitem__ (llava/train/train.py:695-731), which cannot be imported here (it needs
# deepspeed / the llava package), so the 'slice' branch is lifted out of the source TEXT at mint time, wrapped in a
# function and executed — nothing is copied into this repository.
def load_reference_slicer():
    import textwrap
    import torch.nn.functional as F
    src = open("/root/reference/llava/train/train.py").read().splitlines()
    start = next(i for i, l in enumerate(src) if "image_aspect_ratio == 'slice'" in l and l.lstrip().startswith("elif"))
    end = next(i for i in range(start, len(src)) if "image_tensor = torch.cat(split_images, dim=0)" in src[i])
    body = textwrap.dedent("\n".join(src[start + 1:end + 1]))
    body = body.replace("image = self.preprocess(image)\n", "")          # the test image is already normalised
    code = "def ref_slice(image, self):\n" + textwrap.indent(body, "    ") + "\n    return image_tensor, h_block, w_block\n"
    ns = {"torch": torch, "F": F}
    exec(compile(code, "<reference train.py:695-731>", "exec"), ns)
    return ns["ref_slice"]


def mint_slices():
    import numpy as np
    ref = load_reference()
    slicer = load_reference_slicer()
    holder = types.SimpleNamespace(image_patch=ref.Image_Patch(patch_num=9))
    cases = [(336, 336), (500, 700), (1088, 1088), (300, 1400), (901, 413), (224, 224)]
    out = {"sizes": np.array(cases), "stride": np.array(11)}
    for k, (h, w) in enumerate(cases):
        g = torch.Generator().manual_seed(1000 + k)
        img = torch.randn(3, h, w, generator=g)
        crops, hb, wb = slicer(img.clone(), holder)
        out[f"grid_{k}"] = np.array([hb, wb])
        out[f"sub_{k}"] = crops[:, :, ::11, ::11].numpy()
        out[f"sum_{k}"] = np.array([float(crops.double().sum()), float(crops.double().abs().sum())])
    path = os.path.join(ROOT, "tests", "golden", "hd_slice.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: tuple(v.shape) for k, v in out.items() if k.startswith("sub_")})


# ------------------------------------------------------------------------------------------------------------------
# HD token assembly + splice: what the reference's UNMODIFIED prepare_inputs_labels_for_multimodal
# (llava/model/llava_arch.py:100-233, mode == 'slice') returns for a seeded batch.  llava_arch is imported as it lies
# (oracle/reference_loader.py); only its collaborators are stand-ins: encode_images returns the seeded "projector
# output", the tokenizer maps ',' / '\\n' to two ids, the LM is an nn.Embedding.
SPLICE = dict(D=32, M=4, vocab=40, sep_id=7, ret_id=11, seed=4242,
              # (ids with -200 = image token, h_block, w_block) per sample; sample 2 has no image (consumes one crop)
              samples=[([1, 2, -200, 3, 4, 5], 2, 3), ([6, -200, 8, -200, 17, 18], 1, 1), ([9, 10, 12, 13, 14, 15], 1, 1),
                       ([-200, 16, 19, 20, 21, 22], 3, 1)])


def splice_inputs():
    import torch
    g = torch.Generator().manual_seed(SPLICE["seed"])
    D, M = SPLICE["D"], SPLICE["M"]
    L = max(len(s[0]) for s in SPLICE["samples"])
    assert all(len(s[0]) == L for s in SPLICE["samples"]), "the reference takes a rectangular input_ids"
    ids = torch.tensor([s[0] for s in SPLICE["samples"]])
    hb, wb = [s[1] for s in SPLICE["samples"]], [s[2] for s in SPLICE["samples"]]
    n_crops = 0
    for row, h, w in SPLICE["samples"]:
        k = row.count(-200)
        n_crops += max(k, 1) * (h * w + (1 if h * w > 1 else 0)) if k else 1
    # values representable in bf16 / fp16, so that pure copies are exact in every dtype the kernel handles
    feats = torch.randn(n_crops, M, D, generator=g).to(torch.bfloat16).half().float()
    table = torch.randn(SPLICE["vocab"], D, generator=g).to(torch.bfloat16).half().float()
    return ids, hb, wb, feats, table


def mint_splice():
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    for name in ("torchvision", "torchvision.ops", "torchvision.ops.boxes"):      # load_reference()'s stubs confuse transformers
        if name in sys.modules and getattr(sys.modules[name], "__spec__", None) is None:
            del sys.modules[name]
    from oracle import reference_loader as rl
    arch = rl.import_llava_arch()
    ids, hb, wb, feats, table = splice_inputs()

    class LM(arch.LlavaMetaForCausalLM, torch.nn.Module):
        def __init__(self):
            torch.nn.Module.__init__(self)
            self.model = torch.nn.Module()
            self.model.embed_tokens = torch.nn.Embedding.from_pretrained(table)
            self.config = types.SimpleNamespace()
            self.tokenizer = types.SimpleNamespace(
                convert_tokens_to_ids=lambda toks: [{",": SPLICE["sep_id"], "\n": SPLICE["ret_id"]}[t] for t in toks])
            self.device = torch.device("cpu")

        def get_model(self):
            return self.model

        def get_vision_tower(self):
            return object()                              # "there is a tower": llava_arch.py:103-104

        def encode_images(self, images):                 # the projector output for all crops of the batch
            return feats

    with torch.no_grad():
        _, _, _, embeds, _ = LM().prepare_inputs_labels_for_multimodal(ids, None, None, None, torch.zeros(1), "slice", hb, wb)
    out = {"new_input_embeds": embeds.numpy(), "inputs_sha256": np.array(_digest(ids, feats, table))}
    path = os.path.join(ROOT, "tests", "golden", "hd_splice.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, tuple(embeds.shape))


def _digest(*ts):
    import hashlib
    h = hashlib.sha256()
    for t in ts:
        h.update(t.contiguous().numpy().tobytes())
    return h.hexdigest()


if __name__ == "__main__":
    mint_slices()
    mint_splice()
