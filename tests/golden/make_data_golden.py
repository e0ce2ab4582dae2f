"""Generates tests/golden/data_pipeline.npz from the reference's own input-pipeline code (build container only).

    python tests/golden/make_data_golden.py

The reference module datasets/dataset_synapse.py is loaded by path (the name `datasets` collides with an installed package) with
stub modules for h5py / imgaug, which it imports but which the functions exercised here never touch:
  random_rot_flip (:39-46), random_rotate (:48-52) and RandomGenerator.__call__ (:60-73, rot/flip or rotate, then
  scipy.ndimage.zoom order 3 / order 0).
The random draws are replayed beside each call so the fixture records the parameters (k, axis, angle) with the outputs.
The 512 -> 224 case (the size the trainer runs, dataset_synapse.py:108-112) goes through RandomGenerator with a seed that takes
neither augmentation branch.  The fixture holds inputs and outputs only.
"""
import importlib.util
import os
import random
import sys
import types
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def load_reference():
    warnings.simplefilter("ignore")
    for n in ("h5py", "imgaug", "imgaug.augmenters"):
        sys.modules.setdefault(n, types.ModuleType(n))
    sys.modules["imgaug"].augmenters = sys.modules["imgaug.augmenters"]
    spec = importlib.util.spec_from_file_location("ref_dataset_synapse", "/root/reference/datasets/dataset_synapse.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def slice_pair(seed: int, n: int):
    g = np.random.default_rng(seed)
    return g.random((n, n)).astype(np.float32), g.integers(0, 9, (n, n)).astype(np.float32)


def main():
    ref = load_reference()
    out = {}
    n = 48
    for s in range(4):
        img, lab = slice_pair(100 + s, n)
        np.random.seed(s)
        k, axis = np.random.randint(0, 4), np.random.randint(0, 2)
        np.random.seed(s)
        oi, ol = ref.random_rot_flip(img, lab)
        out[f"rot_flip/{s}/params"] = np.array([k, axis], np.int64)
        out[f"rot_flip/{s}/image"], out[f"rot_flip/{s}/label"] = oi, ol.astype(np.uint8)
        np.random.seed(50 + s)
        angle = np.random.randint(-20, 20)
        np.random.seed(50 + s)
        oi, ol = ref.random_rotate(img, lab)
        out[f"rotate/{s}/params"] = np.array([angle], np.int64)
        out[f"rotate/{s}/image"], out[f"rotate/{s}/label"] = oi, ol.astype(np.uint8)
        out[f"input/{s}/seed_n"] = np.array([100 + s, n], np.int64)
    # RandomGenerator: 64 -> 28 (one whose last row/column hits scipy's outside-coordinate rule) and 96 -> 40, 512 -> 224
    gen_cases = []
    for size_in, size_out, want in ((64, 28, "rot_flip"), (64, 28, "rotate"), (96, 40, "none"), (96, 40, "rot_flip"), (512, 224, "none")):
        for s in range(1000):
            random.seed(s)
            np.random.seed(s)
            kind, params = "none", [0, 0, 0]
            if random.random() > 0.5:
                kind, params = "rot_flip", [np.random.randint(0, 4), np.random.randint(0, 2), 0]
            elif random.random() > 0.5:
                kind, params = "rotate", [0, 0, np.random.randint(-20, 20)]
            if kind == want and (s, size_in) not in [(c[0], c[1]) for c in gen_cases]:
                break
        gen_cases.append((s, size_in, size_out, kind, params))
    for i, (s, size_in, size_out, kind, params) in enumerate(gen_cases):
        img, lab = slice_pair(200 + i, size_in)
        random.seed(s)
        np.random.seed(s)
        sample = ref.RandomGenerator((size_out, size_out))({"image": img, "label": lab})
        out[f"generator/{i}/meta"] = np.array([200 + i, size_in, size_out, {"none": 0, "rot_flip": 1, "rotate": 2}[kind]] + list(params), np.int64)
        out[f"generator/{i}/image"] = sample["image"].numpy()[0]
        out[f"generator/{i}/label"] = sample["label"].numpy().astype(np.uint8)
    out["generator/count"] = np.array([len(gen_cases)], np.int64)
    path = os.path.join(HERE, "data_pipeline.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes;", [(c[1], c[2], c[3], c[4]) for c in gen_cases])


if __name__ == "__main__":
    main()
