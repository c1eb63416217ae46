"""CPU: the oracle's two CNN restatements (oracle/caelo_oracle.c) against an INDEPENDENT implementation of the Keras
layer semantics -- torch.nn.functional in float64, driven by the layer stack parsed from each .h5's ``model_config``
(caelo.keras_config), not by anything hard-coded here or in the oracle.  Keras/TensorFlow cannot be installed in this
environment (CNN parity is otherwise pinned only to the restatement); this closes the gap as far as a
second, differently written implementation of the documented semantics can: channels-last Conv2D/Conv3D = zero 'same'
padding + cross-correlation, MaxPooling3D(2, 2, 'same'), Flatten in (x, y, z, c) order, Dense = x @ W + b.
Also: the model_config validation refuses every semantic change."""
import copy
import os

import numpy as np
import pytest

from conftest import GOLDEN, WEIGHTS


def _weights(path):
    from caelo.h5lite import H5File
    h = H5File(path)
    out = {}
    for ln in [n.decode() for n in h.attrs("/model_weights")["layer_names"]]:
        ws = [np.asarray(h.dataset("/model_weights/%s/%s" % (ln, wn.decode())), np.float64)
              for wn in h.attrs("/model_weights/" + ln).get("weight_names", [])]
        if ws:
            out[ln] = ws
    return out


def _torch_forward(path, x):
    """Run the layer stack of a Keras .h5 on a channels-last batch ``x`` in float64 with torch.nn.functional."""
    import torch
    import torch.nn.functional as F
    from caelo import keras_config
    acts = {"relu": torch.relu, "tanh": torch.tanh, "linear": lambda t: t}
    w = _weights(path)
    t = torch.from_numpy(np.asarray(x, np.float64))
    for cls, cfg in keras_config.layers(path):
        if cls == "InputLayer":
            assert list(t.shape[1:]) == cfg["batch_input_shape"][1:]
        elif cls in ("Conv2D", "Conv3D"):
            nd = 2 if cls == "Conv2D" else 3
            assert cfg["data_format"] == "channels_last" and cfg["padding"] == "same" and set(cfg["strides"]) == {1}
            k, b = w[cfg["name"]]
            kt = torch.from_numpy(k).permute(nd + 1, nd, *range(nd))            # [..., cin, cout] -> [cout, cin, ...]
            tin = t.permute(0, nd + 1, *range(1, nd + 1))                        # channels-last -> channels-first
            pad = [s // 2 for s in cfg["kernel_size"]]                           # odd kernels, stride 1: symmetric zero pad
            y = (F.conv2d if nd == 2 else F.conv3d)(tin, kt, torch.from_numpy(b), padding=pad)   # cross-correlation, like Keras
            t = acts[cfg["activation"]](y.permute(0, *range(2, nd + 2), 1))
        elif cls == "MaxPooling3D":
            assert cfg["padding"] == "same"
            tin = t.permute(0, 4, 1, 2, 3)
            y = F.max_pool3d(tin, tuple(cfg["pool_size"]), tuple(cfg["strides"]), ceil_mode=True)   # 'same' on even sizes: no pad
            t = y.permute(0, 2, 3, 4, 1)
        elif cls == "Flatten":
            t = t.reshape(t.shape[0], -1)                                        # channels_last: (x, y, z, c), c fastest
        elif cls == "Dense":
            k, b = w[cfg["name"]]
            t = acts[cfg["activation"]](t @ torch.from_numpy(k) + torch.from_numpy(b))
        else:
            raise AssertionError("layer %s not handled" % cls)
    return t.numpy()


def test_response_layer_restatement_vs_torch_f64(orc, models, scans):
    ring, _ = orc.ProjectPC2SphericalRing(scans(0))
    x = np.ascontiguousarray(ring[None, 0:64, 0:1792, 0:3])
    got = models[0].predict(x)
    want = _torch_forward(os.path.join(WEIGHTS, "SphericalRingPCRespondLayer.h5"), x)
    assert got.shape == want.shape == (1, 64, 1792, 8)
    scale = np.abs(want).max()
    assert scale > 1.0 and (want > 0).mean() > 0.05                   # a live image, not all-relu-zero
    # f32 accumulation over 27 + 32 products of O(50 m) coordinates: ~1e-6 relative to the image scale
    assert np.abs(got - want).max() <= 2e-6 * scale, np.abs(got - want).max() / scale


def test_encoder_restatement_vs_torch_f64(orc, models):
    g = np.load(os.path.join(GOLDEN, "frame_0.npz"))
    rs = np.random.RandomState(2)
    bits = np.concatenate([g["patch_bits"][::16, s] for s in range(3)])       # 192 real patches, all three scales
    dense_rand = rs.uniform(size=(8, 4096)) < np.array([0.001, 0.01, 0.05, 0.1, 0.3, 0.5, 0.9, 1.0])[:, None]
    bits = np.concatenate([bits, np.packbits(dense_rand, axis=1, bitorder="little").view(np.uint64), np.zeros((1, 64), np.uint64)])
    x = orc.unpack_patches(bits)                                               # [n,16,16,16,1] like GetPatchesList
    got = models[1].predict_bits(bits)
    want = _torch_forward(os.path.join(WEIGHTS, "EncoderModel4VoxelPatch.h5"), x)
    assert got.shape == want.shape == (len(bits), 20)
    assert np.abs(want).std() > 0.05 and np.abs(want).max() < 1.0
    assert np.abs(got - want).max() <= 2e-6, np.abs(got - want).max()         # f32 vs f64 through 5 layers, outputs in (-1, 1)
    # ... and the golden descriptors the GPU tests use were produced by that restatement
    feats = np.concatenate([models[1].predict_bits(np.ascontiguousarray(g["patch_bits"][:, s])) for s in range(3)], axis=1)
    assert np.abs(feats - g["features"]).max() <= 1e-6


def test_model_config_is_parsed_and_anything_else_is_refused():
    from caelo import keras_config
    for name, kind in (("SphericalRingPCRespondLayer.h5", "respond"), ("EncoderModel4VoxelPatch.h5", "encoder")):
        lys = keras_config.layers(os.path.join(WEIGHTS, name))
        assert keras_config.check(lys) == kind
        for i, (cls, cfg) in enumerate(lys):
            for key, bad in (("activation", "relu" if kind == "encoder" else "tanh"), ("padding", "valid"),
                             ("strides", [3] * len(cfg.get("strides", []))), ("data_format", "channels_first"),
                             ("dilation_rate", [2] * len(cfg.get("dilation_rate", []))), ("use_bias", False),
                             ("kernel_size", [5] * len(cfg.get("kernel_size", []))), ("units", 7), ("filters", 7),
                             ("pool_size", [3, 3, 3])):
                if key not in cfg:
                    continue
                mod = copy.deepcopy(lys)
                mod[i][1][key] = bad
                with pytest.raises(ValueError):
                    keras_config.check(mod)
        with pytest.raises(ValueError):
            keras_config.check(lys[:-1])
    # the stale training script's relu/linear encoder (AE4VoxelPatch.py:177-197) is exactly such a refused stack
    lys = keras_config.layers(os.path.join(WEIGHTS, "EncoderModel4VoxelPatch.h5"))
    mod = copy.deepcopy(lys)
    mod[-1][1]["activation"] = "linear"
    with pytest.raises(ValueError):
        keras_config.check(mod)
