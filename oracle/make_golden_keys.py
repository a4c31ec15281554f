"""TEST INFRASTRUCTURE.  Which gradient slices tests/golden/kd_step_cfg3_B16.npz holds (shared by the generator
oracle/make_goldens.py, which needs /root/reference, and by tests/test_kd_step_b16_gpu.py, which must not)."""
import numpy as np

B16_KEYS = {   # tensor -> slice kept in the fixture (a few thousand values each, spread over the whole tensor)
    "encoder.resnet.6.0.conv1.weight": np.s_[::4, ::8, 0, 0], "encoder.resnet.6.0.conv2.weight": np.s_[::8, ::8],
    "encoder.resnet.6.0.bn1.weight": np.s_[:], "encoder.resnet.6.0.downsample.0.weight": np.s_[::16, ::8, 0, 0],
    "encoder.resnet.6.3.conv2.weight": np.s_[::8, ::8], "encoder.resnet.6.5.conv3.weight": np.s_[::16, ::4, 0, 0],
    "encoder.resnet.6.5.bn3.bias": np.s_[:], "encoder.resnet.7.0.conv1.weight": np.s_[::8, ::16, 0, 0],
    "encoder.resnet.7.0.conv2.weight": np.s_[::16, ::16], "encoder.resnet.7.0.downsample.0.weight": np.s_[::32, ::16, 0, 0],
    "encoder.resnet.7.1.conv2.weight": np.s_[::16, ::16], "encoder.resnet.7.2.conv3.weight": np.s_[::32, ::8, 0, 0],
    "encoder.resnet.7.2.bn3.weight": np.s_[:], "encoder.projection.0.weight": np.s_[::4, ::32],
    "attention_refinement.attention.in_proj_weight": np.s_[::8, ::8], "attention_refinement.ffn.0.weight": np.s_[::8, ::8],
    "decoder.lstm.weight_hh_l0": np.s_[::16, ::8], "decoder.lstm.weight_ih_l1": np.s_[::16, ::8],
    "decoder.attention.weight": np.s_[::4, ::8], "decoder.output_projection.3.weight": np.s_[::40, ::4],
    "decoder.embedding.weight": np.s_[::40, ::4],
}

# cfg5's student (embed 384 / hidden 768 / 3 LSTM layers): the trunk slices of B16_KEYS plus its own wider / deeper heads
CFG5_KEYS = {k: v for k, v in B16_KEYS.items() if k.startswith("encoder.resnet.")}
CFG5_KEYS.update({
    "encoder.projection.0.weight": np.s_[::4, ::32], "attention_refinement.attention.in_proj_weight": np.s_[::12, ::8],
    "attention_refinement.ffn.0.weight": np.s_[::8, ::8], "decoder.lstm.weight_hh_l0": np.s_[::24, ::8],
    "decoder.lstm.weight_ih_l1": np.s_[::24, ::8], "decoder.lstm.weight_hh_l2": np.s_[::24, ::8],
    "decoder.attention.weight": np.s_[::4, ::8], "decoder.output_projection.3.weight": np.s_[::40, ::4],
    "decoder.embedding.weight": np.s_[::40, ::4],
})

# CompactCaptioningStudent at B = 16 (tests/golden/compact_student_B16.npz): MobileNetV2 trunk tensors + head / decoder tensors
COMPACT_B16_KEYS = {
    "encoder.backbone.18.0.weight": np.s_[::8, ::4, 0, 0], "encoder.backbone.17.conv.1.0.weight": np.s_[::4, 0],
    "encoder.backbone.17.conv.2.weight": np.s_[::4, ::8, 0, 0], "encoder.backbone.14.conv.0.0.weight": np.s_[::8, ::4, 0, 0],
    "encoder.backbone.12.conv.2.weight": np.s_[::2, ::8, 0, 0], "encoder.backbone.10.conv.2.weight": np.s_[::2, ::8, 0, 0],
    "encoder.backbone.10.conv.1.1.weight": np.s_[:], "encoder.backbone.12.conv.3.weight": np.s_[:],
    "encoder.projection.0.weight": np.s_[::4, ::16], "decoder.attention.weight": np.s_[::4, ::4],
    "decoder.lstm.weight_hh_l0": np.s_[::16, ::4], "decoder.lstm.weight_ih_l0": np.s_[::16, ::4],
    "decoder.embedding.weight": np.s_[::40, ::4], "decoder.output_projection.weight": np.s_[::40, ::4],
}

# the optimized recipe's five steps (tests/golden/optimized_recipe.npz): parameter CHANGES after the last step
OPT_RECIPE_KEYS = {"encoder.projection.0.weight": np.s_[::4, ::16], "encoder.backbone.18.0.weight": np.s_[::8, ::4, 0, 0],
                   "encoder.backbone.14.conv.2.weight": np.s_[::4, ::8, 0, 0], "decoder.attention.weight": np.s_[::4, ::4],
                   "decoder.lstm.weight_hh_l0": np.s_[::16, ::4], "decoder.lstm.bias_ih_l0": np.s_[:], "decoder.embedding.weight": np.s_[::40, ::4],
                   "decoder.output_projection.weight": np.s_[::40, ::4], "projector.feature_projection.0.weight": np.s_[::4, ::8]}
