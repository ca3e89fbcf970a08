"""Layer tables of the encoder and decoders, keyed by the reference's Keras layer names.

The conv table is what the encoder plan (modified_dense_model.py) walks; names follow
feature_generation/dense_model.py:82-83,117-118 (res<stage><block>_branch{2a,2b,2c,1}, bn...),
:146-148 (conv1 / bn_conv1) and :1406-1421 (fpn_*).
"""
from collections import namedtuple

ConvSpec = namedtuple("ConvSpec", "name bn k cin cout stride padding")


def resnet_fpn_convs(stage4_blocks=22):
    """All convolutions of ResNet-101 (stage4_blocks=22; 5 = the reference's 'resnet50' option)
    + the FPN, in execution order."""
    L = [ConvSpec("conv1", "bn_conv1", 7, 3, 64, 2, "pad3")]

    def stage(s, blocks, cin, mid, cout, first_stride):
        c = cin
        for i, blk in enumerate(blocks):
            cn, bn = "res%d%s_branch" % (s, blk), "bn%d%s_branch" % (s, blk)
            st = first_stride if i == 0 else 1
            L.append(ConvSpec(cn + "2a", bn + "2a", 1, c, mid, st, "valid"))
            L.append(ConvSpec(cn + "2b", bn + "2b", 3, mid, mid, 1, "same"))
            L.append(ConvSpec(cn + "2c", bn + "2c", 1, mid, cout, 1, "valid"))
            if i == 0:
                L.append(ConvSpec(cn + "1", bn + "1", 1, c, cout, st, "valid"))
            c = cout

    stage(2, "abc", 64, 64, 256, 1)
    stage(3, "abcd", 256, 128, 512, 2)
    stage(4, ["a"] + [chr(98 + i) for i in range(stage4_blocks)], 512, 256, 1024, 2)
    stage(5, "abc", 1024, 512, 2048, 2)
    for name, cin in (("fpn_c5p5", 2048), ("fpn_c4p4", 1024), ("fpn_c3p3", 512), ("fpn_c2p2", 256)):
        L.append(ConvSpec(name, None, 1, cin, 256, 1, "valid"))
    for name in ("fpn_p2", "fpn_p3", "fpn_p4", "fpn_p5"):
        L.append(ConvSpec(name, None, 3, 256, 256, 1, "same"))
    return L


def vgg16_convs():
    """The 13 convolutions of keras.applications VGG16 (block<b>_conv<i>, 3x3 'same' + bias + ReLU), the model
    `image captioning/vgg16.py:9-18` loads.  Alternative-backbone benchmark config only: no dense-captioning path of the
    reference runs it."""
    L, cin = [], 3
    for b, (n, cout) in enumerate(((2, 64), (2, 128), (3, 256), (3, 512), (3, 512)), 1):
        for i in range(1, n + 1):
            L.append(ConvSpec("block%d_conv%d" % (b, i), None, 3, cin, cout, 1, "same"))
            cin = cout
    return L


HEAD_LAYERS = ("mrcnn_class_conv1", "mrcnn_class_bn1", "mrcnn_class_conv2", "mrcnn_class_bn2")
V2_WORD_LSTM = "lstm_1"   # the unnamed KL.LSTM(1024) of text_generation_model_v2.py:157 (Keras auto name)
