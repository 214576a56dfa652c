"""Measured per-stage relative error of the bf16 trunk against the fp32 oracle (GPU box): the numbers the budgets of
tests/test_gpu_config1.py::test_bf16_stage_taps_against_the_fp32_oracle are set from."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy, torch
from oracle import xvector as oxv
from sidekit_amd.nnet import Xtractor
from sidekit_amd.nnet.weights import seeded_state_dict

sd = seeded_state_dict("halfresnet34", 16, seed=1234)
model = Xtractor(16, model_archi="halfresnet34", loss="aam", seed=0).to("cuda").eval()
model.load_state_dict(sd, strict=True)
model.compute_dtype = "bf16"
names = ["stem", "layer1", "layer2", "layer3", "layer4"]
for seed, frames in ((55, [401, 401]), (56, [401, 137, 260, 52]), (57, [801])):
    g = torch.Generator().manual_seed(seed)
    T = max(frames)
    feats = torch.randn(len(frames), 80, T, generator=g)
    model.set_debug(True)
    _, emb = model.forward_features(feats.cuda(), frames=frames)
    raw = model.debug_taps(names)
    model.set_debug(False)
    for b, t in enumerate(frames):
        taps = {}
        with torch.no_grad():
            _, o_emb = oxv.halfresnet34_from_feats(feats[b:b + 1, :, :t], sd, taps=taps)
        errs = []
        for li, name in enumerate(names):
            ref = taps[name][0]
            C, H, W = ref.shape
            Hmax = T
            for _ in range(max(li - 1, 0)):
                Hmax = (Hmax + 1) // 2
            got = torch.from_numpy((raw[name].view(numpy.uint16).astype(numpy.uint32) << 16).view(numpy.float32).copy()).reshape(len(frames), Hmax, W, C)[b, :H].permute(2, 0, 1)
            errs.append(float((got.double() - ref.double()).norm() / ref.double().norm()))
        cos = float(torch.nn.functional.cosine_similarity(emb[b:b + 1].cpu(), o_emb))
        print(frames, b, " ".join(f"{n}={e:.2e}" for n, e in zip(names, errs)), f"emb 1-cos={1 - cos:.2e}", flush=True)
