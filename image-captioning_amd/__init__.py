"""MI355X-native dense-captioning hot path (RoI feature extractor + inject/par-inject caption decoders).

Layout: csrc/ = hand-written HIP kernels behind the C-ABI declared in include/dcap.h;
the Python modules mirror the reference's own module names (config, utils, modified_dense_model,
generate_one_roi_features, text_generation_model, text_generation_model_v2, parallel_model)."""
__version__ = "0.1.0"
