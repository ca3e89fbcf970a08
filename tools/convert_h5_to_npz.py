"""Convert a Keras weight file of the reference (mask_rcnn_coco.h5, rcnn_coco.h5, img_cap_dense.h5, model-47-1.74.h5, ...) to the
.npz layout this package loads natively: one array per '<layer name>/<weight name>' (kernel, bias, gamma, beta, moving_mean,
moving_variance, recurrent_kernel, embeddings).  Uses the package's own HDF5 reader (no h5py needed; load_weights() also
takes the .h5 file directly -- the conversion only saves the parse on later loads):

    python tools/convert_h5_to_npz.py mask_rcnn_coco.h5 mask_rcnn_coco.npz
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    if len(sys.argv) != 3:
        raise SystemExit(__doc__)
    from image_captioning_amd.modified_dense_model import load_weight_file
    weights = load_weight_file(sys.argv[1])
    np.savez(sys.argv[2], **weights)
    print("wrote %s: %d arrays, %.1f M parameters" % (sys.argv[2], len(weights), sum(v.size for v in weights.values()) / 1e6))


if __name__ == "__main__":
    main()
