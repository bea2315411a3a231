"""Reader for TensorFlow 2 checkpoints (tensor-bundle format) without TensorFlow, and the mapping of a
DMCF checkpoint onto the model classes of this package.

The reference saves ``tf.train.Checkpoint(step, optimizer, model)`` (pipelines/base_pipeline.py:155-169) and
restores with ``expect_partial`` (:171-187).  Format: ``ckpt.index`` is a leveldb-style
SSTable (prefix-compressed blocks + restart array, 48-byte footer with magic 0xdb4775248b80fb57) whose
values are ``BundleEntryProto`` messages {1: dtype, 2: shape, 3: shard, 4: offset, 5: size}; tensor bytes are
raw little-endian row-major in ``ckpt.data-00000-of-00001``.

Object-graph keys are attribute paths: ``model/_all_convs/<i>/1/{kernel,bias}`` (i = creation order of
PBFNet.get_cconv, models/pbf_model.py:223), ``model/denses/<l>/<s>/<k>/<i>/{kernel,bias}``,
``model/{fluid,obs}_dense/{kernel,bias}``, each followed by ``/.ATTRIBUTES/VARIABLE_VALUE``.
"""
import os
import re

import numpy as np

_MAGIC = 0xDB4775248B80FB57
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 9: np.int64, 10: np.bool_}


def _varint(buf, pos):
    result, shift = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7


def _parse_block(buf, offset, size):
    """-> list of (key bytes, value bytes) of one SSTable block."""
    block = buf[offset:offset + size]
    num_restarts = int.from_bytes(block[-4:], "little")
    end = len(block) - 4 - 4 * num_restarts
    pos, key, out = 0, b"", []
    while pos < end:
        shared, pos = _varint(block, pos)
        non_shared, pos = _varint(block, pos)
        vlen, pos = _varint(block, pos)
        key = key[:shared] + bytes(block[pos:pos + non_shared])
        pos += non_shared
        out.append((key, bytes(block[pos:pos + vlen])))
        pos += vlen
    return out


def _parse_proto(buf):
    """Minimal protobuf wire parser -> {field: [values]} (varint, 64-bit, length-delimited, 32-bit)."""
    pos, out = 0, {}
    while pos < len(buf):
        tag, pos = _varint(buf, pos)
        field, wire = tag >> 3, tag & 7
        if wire == 0:
            val, pos = _varint(buf, pos)
        elif wire == 1:
            val = buf[pos:pos + 8]
            pos += 8
        elif wire == 2:
            n, pos = _varint(buf, pos)
            val = buf[pos:pos + n]
            pos += n
        elif wire == 5:
            val = buf[pos:pos + 4]
            pos += 4
        else:
            raise ValueError(f"unsupported wire type {wire}")
        out.setdefault(field, []).append(val)
    return out


def read_index(index_path):
    """-> {key: dict(dtype, shape, shard, offset, size)} for every tensor entry of ``ckpt.index``."""
    buf = open(index_path, "rb").read()
    footer = buf[-48:]
    if int.from_bytes(footer[-8:], "little") != _MAGIC:
        raise ValueError(f"{index_path}: not a tensor-bundle index (bad magic)")
    pos = 0
    _, pos = _varint(footer, pos)  # metaindex handle
    _, pos = _varint(footer, pos)
    idx_off, pos = _varint(footer, pos)
    idx_size, pos = _varint(footer, pos)
    entries = {}
    for _, handle in _parse_block(buf, idx_off, idx_size):
        off, p = _varint(handle, 0)
        size, p = _varint(handle, p)
        if buf[off + size] != 0:
            raise NotImplementedError("compressed SSTable blocks")
        for key, value in _parse_block(buf, off, size):
            if key == b"":
                continue  # bundle header
            msg = _parse_proto(value)
            shape = []
            if 2 in msg:
                for dim in _parse_proto(msg[2][0]).get(2, []):
                    shape.append(_parse_proto(dim).get(1, [0])[0])
            entries[key.decode()] = dict(dtype=msg.get(1, [0])[0], shape=tuple(shape), shard=msg.get(3, [0])[0],
                                         offset=msg.get(4, [0])[0], size=msg.get(5, [0])[0])
    return entries


def load_checkpoint(prefix, include_optimizer=False):
    """``prefix`` = path without ``.index`` / ``.data-...`` -> {variable path: numpy array}.
    Keys have the ``/.ATTRIBUTES/VARIABLE_VALUE`` suffix stripped; Adam slots are skipped by default."""
    entries = read_index(prefix + ".index")
    shards = {}
    out = {}
    for key, e in entries.items():
        if not key.endswith("/.ATTRIBUTES/VARIABLE_VALUE"):
            continue
        if not include_optimizer and ".OPTIMIZER_SLOT" in key:
            continue
        if e["dtype"] not in _DTYPES:
            continue
        if e["shard"] not in shards:
            cands = [f for f in os.listdir(os.path.dirname(prefix) or ".")
                     if f.startswith(os.path.basename(prefix) + ".data-%05d-of-" % e["shard"])]
            if not cands:
                raise FileNotFoundError(f"data shard {e['shard']} of {prefix} is missing")
            shards[e["shard"]] = np.memmap(os.path.join(os.path.dirname(prefix) or ".", cands[0]), dtype=np.uint8,
                                           mode="r")
        raw = shards[e["shard"]][e["offset"]:e["offset"] + e["size"]]
        arr = np.frombuffer(bytes(raw), dtype=_DTYPES[e["dtype"]]).reshape(e["shape"])
        out[key[:-len("/.ATTRIBUTES/VARIABLE_VALUE")]] = arr
    return out


def _assign(module, attr, value, device):
    import torch
    t = torch.from_numpy(np.array(value, copy=True)).to(device)
    setattr(module, attr, torch.nn.Parameter(t, requires_grad=False))
    if hasattr(module, "invalidate_packed"):
        module.invalidate_packed()


def model_weight_items(model):
    """-> list of ([candidate checkpoint key prefixes], module) for every weight-bearing layer of a
    PBFNet-family model.  TensorFlow names a variable by the shortest attribute path to it, so a conv that
    is also a direct attribute is stored under that name (``model/fluid_convs``, ``model/obs_convs``,
    ``model/sym_convs/<i>``, ``model/adv_convs/<i>``) and all others under ``model/_all_convs/<i>/1``
    (observed in checkpoints/*/ckpt.index)."""
    alias = {id(model.fluid_convs): "model/fluid_convs", id(model.obs_convs): "model/obs_convs"}
    for i, conv in enumerate(getattr(model, "sym_convs", [])):
        alias[id(conv)] = f"model/sym_convs/{i}"
    for i, conv in enumerate(getattr(model, "adv_convs", []) or []):
        alias[id(conv)] = f"model/adv_convs/{i}"
    items = []
    for i, (_, conv) in enumerate(model._all_convs):
        cands = [f"model/_all_convs/{i}/1"]
        if id(conv) in alias:
            cands.insert(0, alias[id(conv)])
        items.append((cands, conv))
    items.append((["model/fluid_dense"], model.fluid_dense))
    items.append((["model/obs_dense"], model.obs_dense))
    if getattr(model, "equivar", False):  # models/pbf_model.py:183-189
        items.append((["model/scale_dens"], model.scale_dens))
        items.append((["model/rot_dens"], model.rot_dens))
    for i, dense in enumerate(getattr(model, "adv_dense", []) or []):
        items.append(([f"model/adv_dense/{i}"], dense))
    denses = getattr(model, "denses", [])
    if denses and isinstance(denses[0], list):  # HRNet / SymNet: denses[layer][scale][k][inp]
        for a, la in enumerate(denses):
            for b, lb in enumerate(la):
                for c, lc in enumerate(lb):
                    for d, dense in enumerate(lc):
                        items.append(([f"model/denses/{a}/{b}/{c}/{d}"], dense))
    else:  # CConv: flat list
        for a, dense in enumerate(denses):
            items.append(([f"model/denses/{a}"], dense))
    return items


def load_into_model(model, weights, device="cuda", strict=True):
    """Assign ``weights`` ({key: array} from :func:`load_checkpoint` or an ``.npz``) to ``model``.
    Layers the checkpoint has no entry for stay lazily initialised (``expect_partial`` semantics,
    pipelines/base_pipeline.py:172-173; e.g. the never-called cross-scale Dense layers); with ``strict``
    every conv kernel must be present.  Returns the number of layers loaded."""
    loaded = 0
    for cands, module in model_weight_items(model):
        prefix = next((c for c in cands if c + "/kernel" in weights), None)
        if prefix is None:
            if strict and hasattr(module, "fixed_radius_search"):
                raise KeyError(f"{cands}: kernel missing from the checkpoint")
            continue
        k = weights[prefix + "/kernel"]
        if hasattr(module, "in_channels"):  # ContinuousConv
            module.in_channels = int(k.shape[-2])
        _assign(module, "kernel", k, device)
        loaded += 1
        b = weights.get(prefix + "/bias")
        if b is not None and getattr(module, "use_bias", True):
            _assign(module, "bias", b, device)
    return loaded


def checkpoint_epoch(path, save_ckpt_freq=1):
    """Epoch recovered from the name of a checkpoint manager's newest checkpoint, like the reference
    (pipelines/base_pipeline.py:182-185): ``(n - 1) * save_ckpt_freq + 1`` for ``ckpt-<n>``."""
    nums = re.findall(r"\d+", os.path.basename(path))
    return (int(nums[-1]) - 1) * int(save_ckpt_freq) + 1 if nums else 0
