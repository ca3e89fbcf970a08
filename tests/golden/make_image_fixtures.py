"""Generates tests/golden/vg_sample_images.json: for each sample Visual Genome JPEG the reference repository ships
(dataset/visual genome/*.jpg -- data files, not source), the decoded shape and a SHA-1 of the decoded RGB bytes as PIL
decodes them here, plus the shape / window / scale resize_image gives for the reference's IMAGE_MIN_DIM = 800,
IMAGE_MAX_DIM = 1024 and a SHA-1 of the resized + padded pixels.  Run in the build container (needs /root/reference):
    python tests/golden/make_image_fixtures.py
The test (tests/test_host_logic.py) re-derives the same numbers from the same files when /root/reference is present."""
import glob
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
REF = "/root/reference/dataset/visual genome"


def describe(path):
    from image_captioning_amd import utils
    img = utils.imread(path)
    out, window, scale, padding = utils.resize_image(img, min_dim=800, max_dim=1024, padding=True)
    return {"shape": list(img.shape), "sha1": hashlib.sha1(np.ascontiguousarray(img).tobytes()).hexdigest(),
            "resized_shape": list(out.shape), "window": [int(v) for v in window], "scale": float(scale),
            "resized_sha1": hashlib.sha1(np.ascontiguousarray(out).tobytes()).hexdigest(),
            "mean": float(img.mean()), "resized_mean": float(out[window[0]:window[2], window[1]:window[3]].mean())}


if __name__ == "__main__":
    rows = {os.path.basename(p): describe(p) for p in sorted(glob.glob(os.path.join(REF, "*.jpg")))}
    with open(os.path.join(HERE, "vg_sample_images.json"), "w") as f:
        json.dump(rows, f, indent=1, sort_keys=True)
    print(json.dumps(rows, indent=1)[:1500])
