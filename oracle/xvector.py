"""Oracle networks: features (or wav) -> x-vector, from a reference-format ``state_dict``.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).

Functional torch-CPU restatement of

* ``sidekit/nnet/res_net.py:258-320`` (SELayer, BasicBlock) and ``:500-554`` (PreHalfResNet34)
* ``sidekit/nnet/pooling.py:44-70`` (MeanStdPooling) and ``:123-171`` (AttentivePooling, num_freqs=10: SURVEY F1')
* ``sidekit/nnet/loss.py:91-100`` (l2_norm), ``:299-310`` (ArcMarginProduct, target=None)
* ``sidekit/nnet/xvector.py:453-513`` (TDNN), ``:569-599`` (halfresnet34), ``:876-907`` (forward)

Everything is driven by the checkpoint key names the reference uses, so a
``state_dict`` that loads ``strict=True`` into the reference model drives this
oracle unchanged.  Pinned by ``tests/golden/*.npz`` (made with the imported
reference, see ``tests/golden/make_golden.py``).
"""
import torch
import torch.nn.functional as F

from . import frontend

HALF_LAYERS = ((32, 3, 1), (64, 4, 2), (128, 6, 2), (256, 3, 2))  # res_net.py:518-521
TDNN_LAYERS = (("conv1", 5, 1), ("conv2", 3, 2), ("conv3", 3, 3), ("conv4", 1, 1), ("conv5", 1, 1))  # xvector.py:467-483


def _bn(x, sd, p, eps=1e-5):
    """BatchNorm in eval mode (running stats), SURVEY N1."""
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"],
                        False, 0.0, eps)


def basic_block(x, sd, p, stride, taps=None):
    """res_net.py:309-320 (+ SELayer :272-281)."""
    out = F.relu(_bn(F.conv2d(x, sd[p + ".conv1.weight"], stride=stride, padding=1), sd, p + ".bn1"))
    out = _bn(F.conv2d(out, sd[p + ".conv2.weight"], padding=1), sd, p + ".bn2")
    y = out.mean(dim=(2, 3))
    y = F.relu(F.linear(y, sd[p + ".se.fc.0.weight"]))
    y = torch.sigmoid(F.linear(y, sd[p + ".se.fc.2.weight"]))
    out = out * y[:, :, None, None]
    if (p + ".shortcut.0.weight") in sd:
        sc = _bn(F.conv2d(x, sd[p + ".shortcut.0.weight"], stride=stride), sd, p + ".shortcut.1")
    else:
        sc = x
    return F.relu(out + sc)


def halfresnet34_trunk(feats, sd, p="sequence_network", taps=None):
    """PreHalfResNet34.forward, res_net.py:539-554.  feats (B, 80, T) -> (B, 256, T', 10)."""
    x = feats.unsqueeze(1).permute(0, 1, 3, 2)  # (B,1,T,80)
    x = F.relu(_bn(F.conv2d(x, sd[p + ".conv1.weight"], padding=1), sd, p + ".bn1"))
    if taps is not None:
        taps["stem"] = x
    for li, (planes, nblocks, stride) in enumerate(HALF_LAYERS, start=1):
        for bi in range(nblocks):
            x = basic_block(x, sd, f"{p}.layer{li}.{bi}", stride if bi == 0 else 1)
        if taps is not None:
            taps[f"layer{li}"] = x
    return x


def mean_std_pooling(x):
    """pooling.py:55-70: mean and *unbiased* std over the last axis."""
    if x.dim() == 4:
        x = x.permute(0, 1, 3, 2).flatten(1, 2)
    return torch.cat([x.mean(dim=2), x.std(dim=2)], dim=1)


def attentive_pooling(x, sd, p="stat_pooling"):
    """pooling.py:151-171 with global_context=True.  x (B, C, T', F) -> (B, 2*C*F)."""
    if x.dim() == 4:
        x = x.permute(0, 1, 3, 2).flatten(1, 2)  # (B, C*F, T)
    ctx = mean_std_pooling(x).unsqueeze(2).repeat(1, 1, x.shape[-1])
    h = F.conv1d(torch.cat([x, ctx], dim=1), sd[p + ".attention.0.weight"], sd[p + ".attention.0.bias"])
    h = torch.tanh(_bn(F.relu(h), sd, p + ".attention.2"))
    w = torch.softmax(F.conv1d(h, sd[p + ".attention.4.weight"], sd[p + ".attention.4.bias"]), dim=2)
    mu = torch.sum(x * w, dim=2)
    rh = torch.sqrt((torch.sum((x ** 2) * w, dim=2) - mu ** 2).clamp(min=1e-9))
    return torch.cat((mu, rh), 1)


def l2_norm(x):
    """loss.py:91-100 (no eps)."""
    return x / torch.norm(x, 2, 1, True)


def aam_logits(x, weight, s):
    """ArcMarginProduct.forward(target=None), loss.py:307-310."""
    return F.linear(F.normalize(x), F.normalize(weight)) * s


def halfresnet34_from_feats(feats, sd, aam_s=30.0, taps=None):
    """Everything after xvector.py:885 for model_archi='halfresnet34', loss='aam'.
    Returns (logits (B, n_spk), emb (B, E))."""
    x = halfresnet34_trunk(feats, sd, taps=taps)
    x = attentive_pooling(x, sd)
    if taps is not None:
        taps["pooled"] = x
    x = F.linear(x, sd["before_speaker_embedding.lin_be.weight"])
    x = _bn(x, sd, "before_speaker_embedding.bn_be")
    if taps is not None:
        taps["pre_norm"] = x
    x = l2_norm(x)
    return aam_logits(x, sd["after_speaker_embedding.weight"], aam_s), F.normalize(x, dim=1)


def halfresnet34_forward(wav, sd, aam_s=30.0):
    """Xtractor.forward(x, is_eval=True), xvector.py:876-907, halfresnet34 + aam."""
    feats = frontend.melspec_frontend(wav, fb=sd.get("preprocessor.MelSpec.mel_scale.fb"),
                                      window=sd.get("preprocessor.MelSpec.spectrogram.window"))
    return halfresnet34_from_feats(feats, sd, aam_s)


def tdnn_trunk(feats, sd, p="sequence_network", taps=None):
    """xvector.py:467-483: conv1d -> LeakyReLU(0.2) -> BatchNorm1d, five times, no padding."""
    x = feats
    for i, (name, k, dil) in enumerate(TDNN_LAYERS, start=1):
        x = F.conv1d(x, sd[f"{p}.{name}.weight"], sd[f"{p}.{name}.bias"], dilation=dil)
        x = _bn(F.leaky_relu(x, 0.2), sd, f"{p}.batch_norm{i}")
        if taps is not None:
            taps[name] = x
    return x


def tdnn_from_feats(feats, sd, loss="aam", aam_s=64.0, taps=None):
    """Sub-modules of Xtractor('xvector') called in forward order.
    Returns (logits, emb) for 'aam', emb for 'cce' (is_eval, xvector.py:896-898)."""
    x = tdnn_trunk(feats, sd, taps=taps)
    x = mean_std_pooling(x)
    if taps is not None:
        taps["pooled"] = x
    x = F.linear(x, sd["before_speaker_embedding.linear6.weight"], sd["before_speaker_embedding.linear6.bias"])
    if taps is not None:
        taps["pre_norm"] = x
    x = l2_norm(x)
    if loss == "cce":
        return x
    return aam_logits(x, sd["after_speaker_embedding.weight"], aam_s), F.normalize(x, dim=1)


def tdnn_forward(wav, sd, loss="aam", aam_s=64.0):
    feats = frontend.mfcc_frontend(wav, fb=sd.get("preprocessor.MFCC.MelSpectrogram.mel_scale.fb"),
                                   dct=sd.get("preprocessor.MFCC.dct_mat"),
                                   window=sd.get("preprocessor.MFCC.MelSpectrogram.spectrogram.window"))
    return tdnn_from_feats(feats, sd, loss, aam_s)


def forward_ragged(wavs, sd, arch="halfresnet34", **kw):
    """parity for variable-length batches is defined per utterance, each run alone."""
    fn = halfresnet34_forward if arch == "halfresnet34" else tdnn_forward
    outs = [fn(w.reshape(1, -1), sd, **kw) for w in wavs]
    if isinstance(outs[0], tuple):
        return torch.cat([o[0] for o in outs]), torch.cat([o[1] for o in outs])
    return torch.cat(outs)
