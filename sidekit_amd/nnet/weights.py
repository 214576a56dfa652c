"""Checkpoint layout of the two extractors and a seeded random initialiser.

``state_dict_spec`` lists, in the reference's own ``state_dict()`` order, every key a checkpoint
of ``Xtractor(model_archi='halfresnet34' | 'xvector')`` holds (``sidekit/nnet/xvector.py:453-513,
569-599``; loaded ``strict=True`` at ``sidekit/bin/extract_xvectors.py:86``).  The pooling shape
is the only one consistent with the trunk and ``lin_be`` (SURVEY F1': num_freqs=10).

``seeded_state_dict`` draws weights from ``numpy.random.RandomState(seed)`` (stable across
numpy/torch versions and machines) with non-trivial BatchNorm statistics, so BatchNorm folding is
exercised by every parity test and the same weights can be rebuilt on the GPU box.
"""
from collections import OrderedDict

import numpy
import torch

from .preprocessor import MelSpecFrontEnd, MfccFrontEnd

HALF_PLANES = (32, 64, 128, 256)
HALF_BLOCKS = (3, 4, 6, 3)


def _bn(spec, prefix, c):
    spec[prefix + ".weight"] = ((c,), "bn_w")
    spec[prefix + ".bias"] = ((c,), "bn_b")
    spec[prefix + ".running_mean"] = ((c,), "bn_m")
    spec[prefix + ".running_var"] = ((c,), "bn_v")
    spec[prefix + ".num_batches_tracked"] = ((), "count")


def state_dict_spec(model_archi, speaker_number, embedding_size=256, loss="aam"):
    """OrderedDict key -> (shape, kind)."""
    spec = OrderedDict()
    E, S = int(embedding_size), int(speaker_number)
    if model_archi == "halfresnet34":
        for k, v in MelSpecFrontEnd().buffers().items():
            spec["preprocessor." + k] = (tuple(v.shape), "buffer")
        sn = "sequence_network"
        spec[sn + ".conv1.weight"] = ((32, 1, 3, 3), "conv")
        _bn(spec, sn + ".bn1", 32)
        in_planes = 32
        for li, (planes, nb) in enumerate(zip(HALF_PLANES, HALF_BLOCKS), start=1):
            for bi in range(nb):
                p = f"{sn}.layer{li}.{bi}"
                spec[p + ".conv1.weight"] = ((planes, in_planes, 3, 3), "conv")
                _bn(spec, p + ".bn1", planes)
                spec[p + ".conv2.weight"] = ((planes, planes, 3, 3), "conv")
                _bn(spec, p + ".bn2", planes)
                spec[p + ".se.fc.0.weight"] = ((planes // 16, planes), "linear")
                spec[p + ".se.fc.2.weight"] = ((planes, planes // 16), "linear")
                if bi == 0:  # tuple stride != 1 -> every first block has a conv shortcut
                    spec[p + ".shortcut.0.weight"] = ((planes, in_planes, 1, 1), "conv")
                    _bn(spec, p + ".shortcut.1", planes)
                in_planes = planes
        spec["before_speaker_embedding.lin_be.weight"] = ((E, 5120), "linear")
        _bn(spec, "before_speaker_embedding.bn_be", E)
        spec["stat_pooling.attention.0.weight"] = ((128, 7680, 1), "conv")
        spec["stat_pooling.attention.0.bias"] = ((128,), "bias")
        _bn(spec, "stat_pooling.attention.2", 128)
        spec["stat_pooling.attention.4.weight"] = ((2560, 128, 1), "conv")
        spec["stat_pooling.attention.4.bias"] = ((2560,), "bias")
        spec["after_speaker_embedding.weight"] = ((S, E), "linear")
    elif model_archi == "xvector":
        for k, v in MfccFrontEnd().buffers().items():
            spec["preprocessor." + k] = (tuple(v.shape), "buffer")
        cin, cout, ks = (80, 512, 512, 512, 512), (512, 512, 512, 512, 1536), (5, 3, 3, 1, 1)
        for i in range(5):
            spec[f"sequence_network.conv{i + 1}.weight"] = ((cout[i], cin[i], ks[i]), "conv")
            spec[f"sequence_network.conv{i + 1}.bias"] = ((cout[i],), "bias")
            _bn(spec, f"sequence_network.batch_norm{i + 1}", cout[i])
        spec["before_speaker_embedding.linear6.weight"] = ((E, 3072), "linear")
        spec["before_speaker_embedding.linear6.bias"] = ((E,), "bias")
        if loss == "aam":
            spec["after_speaker_embedding.weight"] = ((S, E), "linear")
        else:  # 'cce' head of xvector.py:499-507: training only, held but never run at eval
            _bn(spec, "after_speaker_embedding.batch_norm6", 512)
            spec["after_speaker_embedding.linear7.weight"] = ((512, 512), "linear")
            spec["after_speaker_embedding.linear7.bias"] = ((512,), "bias")
            _bn(spec, "after_speaker_embedding.batch_norm7", 512)
            spec["after_speaker_embedding.linear8.weight"] = ((S, 512), "linear")
            spec["after_speaker_embedding.linear8.bias"] = ((S,), "bias")
    else:
        raise NotImplementedError(f"model_archi={model_archi!r}: only 'halfresnet34' and 'xvector' are built (SURVEY 8a)")
    return spec


def seeded_state_dict(model_archi, speaker_number, embedding_size=256, loss="aam", seed=1234):
    """Deterministic random checkpoint with the reference's key names, shapes and dtypes."""
    rs = numpy.random.RandomState(seed)
    fe = (MelSpecFrontEnd() if model_archi == "halfresnet34" else MfccFrontEnd()).buffers()
    sd = OrderedDict()
    for key, (shape, kind) in state_dict_spec(model_archi, speaker_number, embedding_size, loss).items():
        if kind == "buffer":
            sd[key] = fe[key[len("preprocessor."):]].clone()
            continue
        if kind == "count":
            sd[key] = torch.tensor(0, dtype=torch.int64)
            continue
        if kind in ("conv", "linear"):
            fan_in = int(numpy.prod(shape[1:]))
            gain = 2.0 if kind == "conv" else 1.0
            a = rs.standard_normal(shape) * numpy.sqrt(gain / fan_in)
        elif kind == "bias":
            a = rs.standard_normal(shape) * 0.1
        elif kind == "bn_w":
            a = rs.uniform(0.8, 1.2, shape)
        elif kind == "bn_b":
            a = rs.standard_normal(shape) * 0.1
        elif kind == "bn_m":
            a = rs.standard_normal(shape) * 0.1
        elif kind == "bn_v":
            a = rs.uniform(0.8, 1.2, shape)
        else:
            raise AssertionError(kind)
        sd[key] = torch.from_numpy(numpy.ascontiguousarray(a, dtype=numpy.float32))
    return sd
