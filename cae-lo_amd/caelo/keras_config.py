"""The layer stack recorded in a Keras ``.h5`` (root attribute ``model_config``) -- parsed, never assumed.

The reference obtains its two networks with ``keras.models.load_model(path)`` (Dirs.py:29-30, Match.py:313,324), which
rebuilds the layers from this JSON.  The HIP kernels implement exactly one architecture per network; ``check`` compares
every semantic field of every layer (kernel size, strides, padding, dilation, data format, activation, bias, pooling,
units, input shape) with what the kernels do and refuses anything else, so a different ``.h5`` can never be run
silently through the wrong arithmetic.
"""
import json

from .h5lite import H5File

# what ring.hip::k_respond and encoder.hip / config5.hip implement
_CONV = dict(padding="same", data_format="channels_last", use_bias=True)
RESPOND_SPEC = [
    ("InputLayer", dict(batch_input_shape=[None, 64, 1792, 3])),
    ("Conv2D", dict(_CONV, filters=32, kernel_size=[3, 3], strides=[1, 1], dilation_rate=[1, 1], activation="relu")),
    ("Conv2D", dict(_CONV, filters=8, kernel_size=[1, 1], strides=[1, 1], dilation_rate=[1, 1], activation="relu")),
]
_C3 = dict(_CONV, kernel_size=[3, 3, 3], strides=[1, 1, 1], dilation_rate=[1, 1, 1], activation="tanh")
_POOL = dict(pool_size=[2, 2, 2], strides=[2, 2, 2], padding="same", data_format="channels_last")
ENCODER_SPEC = [
    ("InputLayer", dict(batch_input_shape=[None, 16, 16, 16, 1])),
    ("Conv3D", dict(_C3, filters=8)), ("MaxPooling3D", _POOL),
    ("Conv3D", dict(_C3, filters=16)), ("MaxPooling3D", _POOL),
    ("Conv3D", dict(_C3, filters=32)),
    ("Flatten", dict(data_format="channels_last")),
    ("Dense", dict(units=200, activation="tanh", use_bias=True)),
    ("Dense", dict(units=20, activation="tanh", use_bias=True)),
]


def _norm(v):
    return list(v) if isinstance(v, (list, tuple)) else v


def layers(path_or_h5):
    """-> [(class_name, config dict)] of a Keras Model / Sequential ``.h5`` in execution order (linear stacks only)."""
    h = path_or_h5 if isinstance(path_or_h5, H5File) else H5File(path_or_h5)
    cfg = json.loads(h.attrs("/")["model_config"].decode("utf8"))
    lys = cfg["config"]["layers"] if isinstance(cfg["config"], dict) else cfg["config"]
    out = []
    for i, l in enumerate(lys):
        inbound = l.get("inbound_nodes") or []
        if i > 0 and inbound:   # functional Model: every layer must consume exactly its predecessor
            srcs = [n[0] for n in inbound[0]]
            if srcs != [lys[i - 1]["config"]["name"]]:
                raise ValueError("model_config is not a linear stack at layer %s" % l["config"].get("name"))
        out.append((l["class_name"], l["config"]))
    return out


def kind_of(lys):
    classes = [c for c, _ in lys]
    if "Conv2D" in classes:
        return "respond"
    if "Conv3D" in classes:
        return "encoder"
    raise ValueError("not one of the two CAE-LO inference models (layers: %s)" % classes)


def check(lys, kind=None):
    """Raise ValueError unless the stack is exactly the one the kernels implement; returns 'respond' | 'encoder'."""
    kind = kind or kind_of(lys)
    spec = RESPOND_SPEC if kind == "respond" else ENCODER_SPEC
    if [c for c, _ in lys] != [c for c, _ in spec]:
        raise ValueError("unsupported %s layer stack %s (kernels implement %s)" % (kind, [c for c, _ in lys], [c for c, _ in spec]))
    for (cls, cfg), (_, want) in zip(lys, spec):
        for key, val in want.items():
            # Keras 2.2 writes Flatten.data_format / InputLayer shapes; an absent optional field means the Keras default
            got = cfg.get(key, "channels_last" if key == "data_format" else None)
            if _norm(got) != _norm(val):
                raise ValueError("unsupported %s: %s(%s).%s = %r, the kernels implement %r" % (kind, cls, cfg.get("name"), key, got, val))
    return kind
