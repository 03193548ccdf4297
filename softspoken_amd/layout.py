"""State-dict layout and front-end tables of SpecUNet_2D (product module: the drop-in's model class builds its parameter
holders from it, checkpoint packing and the synthetic test checkpoints use the same tables).

Key layout follows SpecUNet_2D.state_dict() of the reference (root/code/backend/pytorch_neural_nets.py:83-140; SURVEY.md
section 8(a) row A5): 222 conv/BN entries plus the two torchaudio buffers `mel_spectrogram.spectrogram.window` and
`mel_spectrogram.mel_scale.fb`.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np

SR = 22050
WINDOW = 3 * SR

# (name, cin, cout) of every 2-D residual block, in module-definition order
# (reference pytorch_neural_nets.py:101-128).
RESBLOCKS_2D = [
    ("conv1_1", 1, 32),
    ("conv2_1", 32, 64),
    ("conv3_1", 64, 96),
    ("conv4_1", 96, 128),
    ("conv_bottleneck", 128, 128),
    ("encoder_out", 128, 128),
    ("conv6", 256, 96),
    ("conv7", 192, 64),
    ("conv8", 128, 32),
    ("conv9_1", 64, 32),
    ("spec_output_conv.0", 32, 32),
]


def state_dict_layout():
    """Ordered {key: (shape, kind)} of the reference model's state_dict (224 keys).
    kind: window | fb | conv_w | conv_b | bn_gamma | bn_beta | bn_mean | bn_var | bn_count."""
    lay = OrderedDict()
    lay["mel_spectrogram.spectrogram.window"] = ((512,), "window")
    lay["mel_spectrogram.mel_scale.fb"] = ((1025, 128), "fb")

    def bn(prefix, c):
        lay[prefix + ".weight"] = ((c,), "bn_gamma")
        lay[prefix + ".bias"] = ((c,), "bn_beta")
        lay[prefix + ".running_mean"] = ((c,), "bn_mean")
        lay[prefix + ".running_var"] = ((c,), "bn_var")
        lay[prefix + ".num_batches_tracked"] = ((), "bn_count")

    def resblock(name, cin, cout, one_d=False):
        k1 = (1,) if one_d else (1, 1)
        k3 = (3,) if one_d else (3, 3)
        lay[f"{name}.residual.0.weight"] = ((cout, cin) + k1, "conv_w")
        bn(f"{name}.residual.1", cout)
        lay[f"{name}.conv1.0.weight"] = ((cout, cin) + k3, "conv_w")
        bn(f"{name}.conv1.1", cout)
        lay[f"{name}.conv2.0.weight"] = ((cout, cout) + k3, "conv_w")
        bn(f"{name}.conv2.1", cout)

    for name, cin, cout in RESBLOCKS_2D[:-1]:
        resblock(name, cin, cout)
    resblock("spec_output_conv.0", 32, 32)
    lay["spec_output_conv.1.weight"] = ((2, 32, 1, 1), "conv_w")
    lay["spec_output_conv.1.bias"] = ((2,), "conv_b")
    lay["conv_flatten.weight"] = ((4, 32, 128, 1), "conv_w")
    lay["conv_flatten.bias"] = ((4,), "conv_b")
    resblock("mask_output_conv.0", 4, 4, one_d=True)
    lay["mask_output_conv.1.weight"] = ((1, 4, 1), "conv_w")
    lay["mask_output_conv.1.bias"] = ((1,), "conv_b")
    return lay


def hann_window_512():
    """torch.hann_window(512): the buffer torchaudio's Spectrogram registers and a real checkpoint
    carries.  torch evaluates 0.5 - 0.5*cos(2*pi*n/512) in float32, which is up to 1.8e-7 away from
    the exactly rounded Hann near the ends, so the tensor itself is used, not a re-derivation."""
    import torch
    return torch.hann_window(512).numpy().copy()


def mel_filterbank():
    """HTK mel filterbank (1025, 128) float32, torchaudio melscale_fbanks recipe in float32.

    Restated from SURVEY.md section 8(a) row A3: f_min 0, f_max 8000, sr 22050, norm None, 'htk'.
    float32 arithmetic throughout, as torchaudio does it on float32 tensors.
    """
    f32 = np.float32
    n_freqs, n_mels = 1025, 128
    all_freqs = np.linspace(0.0, float(SR // 2), n_freqs, dtype=np.float64).astype(f32)
    m_min = 2595.0 * math.log10(1.0 + 0.0 / 700.0)
    m_max = 2595.0 * math.log10(1.0 + 8000.0 / 700.0)
    # torch.linspace(float32): start + step*i for the first half, end - step*(n-1-i) for the second
    steps = n_mels + 2
    step = f32((f32(m_max) - f32(m_min)) / f32(steps - 1))
    idx = np.arange(steps)
    half = steps // 2
    m_pts = np.where(idx < half, f32(m_min) + step * idx.astype(f32),
                     f32(m_max) - step * (steps - 1 - idx).astype(f32)).astype(f32)
    f_pts = (f32(700.0) * (np.power(f32(10.0), m_pts / f32(2595.0), dtype=f32) - f32(1.0))).astype(f32)
    f_diff = (f_pts[1:] - f_pts[:-1]).astype(f32)
    slopes = (f_pts[None, :] - all_freqs[:, None]).astype(f32)
    down = ((f32(-1.0) * slopes[:, :-2]) / f_diff[:-1]).astype(f32)
    up = (slopes[:, 2:] / f_diff[1:]).astype(f32)
    fb = np.maximum(f32(0.0), np.minimum(down, up)).astype(f32)
    return fb
