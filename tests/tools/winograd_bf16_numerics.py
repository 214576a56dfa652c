"""What Winograd F(2x2, 3x3) would cost the bf16 trunk in accuracy (round-5 verdict item 2, measurement (d)): a CPU emulation of the bf16 path's
rounding points -- bf16 activations and weights, f32 accumulation, f32 BatchNorm / SE gate, bf16 stored outputs -- run twice over the same
features: every stride-1 3x3 convolution of layer 3 (and optionally layer 4) either direct, or as Winograd with the transformed weights
U = G g G' rounded to bf16 ONCE (from the f32 weights) and the transformed inputs V = B' d B formed in f32 from the bf16 activations and rounded
to bf16 (what an MFMA operand must be), products accumulated in f32, output transform A' M A in f32.  Reports the relative error of the
layer-3 / layer-4 taps and of the x-vector against the fp32 oracle (it calls the oracle, so it lives with the tests).

    python tests/tools/winograd_bf16_numerics.py          # CPU only, ~1 min

The emulated DIRECT path lands near the GPU's measured taps (layer 3: 6.2e-3 measured, 8.8e-3 emulated; tests/test_gpu_config1.py budgets 9.3e-3), which is what
makes the Winograd column comparable (measured here: direct 8.8e-3, Winograd 8.3e-3 at layer 3)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F

from oracle import xvector as oxv
from sidekit_amd.nnet.weights import seeded_state_dict

BT = torch.tensor([[1., 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]])
G = torch.tensor([[1., 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]])
AT = torch.tensor([[1., 1, 1, 0], [0, 1, -1, -1]])


def r(x):
    return x.bfloat16().float()


def conv_direct(x, w):
    return F.conv2d(x, r(w), padding=1)


def conv_winograd(x, w):
    """x (B, C, H, W) bf16-valued f32, w (Co, C, 3, 3) f32 -> (B, Co, H, W) f32; tiles of 2 x 2 outputs from 4 x 4 inputs."""
    B, C, H, W = x.shape
    Hp, Wp = (H + 1) // 2 * 2, (W + 1) // 2 * 2
    xp = F.pad(x, (1, 1 + Wp - W, 1, 1 + Hp - H))
    d = xp.unfold(2, 4, 2).unfold(3, 4, 2)                        # (B, C, th, tw, 4, 4)
    V = r(torch.einsum("xi,bchwij,yj->bchwxy", BT, d, BT))        # B' d B, rounded to bf16 (the MFMA operand)
    U = r(torch.einsum("xi,ocij,yj->ocxy", G, w, G))              # G g G' from the f32 weights, rounded once
    M = torch.einsum("ocxy,bchwxy->bohwxy", U, V)                 # 16 independent contractions over c, f32
    Y = torch.einsum("px,bohwxy,qy->bohwpq", AT, M, AT)           # A' M A, f32
    th, tw = Y.shape[2], Y.shape[3]
    return Y.permute(0, 1, 2, 4, 3, 5).reshape(B, -1, th * 2, tw * 2)[:, :, :H, :W]


def fold(sd, p):
    s = sd[p + ".weight"] / torch.sqrt(sd[p + ".running_var"] + 1e-5)
    return s[None, :, None, None], (sd[p + ".bias"] - sd[p + ".running_mean"] * s)[None, :, None, None]


def block(x, sd, p, stride, conv3):
    s1, h1 = fold(sd, p + ".bn1")
    c1 = conv3(x, sd[p + ".conv1.weight"]) if stride == 1 else F.conv2d(x, r(sd[p + ".conv1.weight"]), stride=stride, padding=1)
    o1 = r(F.relu(c1 * s1 + h1))
    s2, h2 = fold(sd, p + ".bn2")
    z = conv3(o1, sd[p + ".conv2.weight"]) * s2 + h2
    y = torch.sigmoid(F.linear(F.relu(F.linear(z.mean(dim=(2, 3)), sd[p + ".se.fc.0.weight"])), sd[p + ".se.fc.2.weight"]))
    if (p + ".shortcut.0.weight") in sd:
        ss, hs = fold(sd, p + ".shortcut.1")
        sc = F.conv2d(x, r(sd[p + ".shortcut.0.weight"] * ss.reshape(-1, 1, 1, 1)), stride=stride) + hs
    else:
        sc = x
    return r(F.relu(z * y[:, :, None, None] + sc))


def trunk(feats, sd, wino_layers=()):
    p = "sequence_network"
    x = feats.unsqueeze(1).permute(0, 1, 3, 2)
    s, h = fold(sd, p + ".bn1")
    x = r(F.relu(F.conv2d(x, sd[p + ".conv1.weight"] * s.reshape(-1, 1, 1, 1), padding=1) + h))      # the stem runs in f32
    taps = {"stem": x}
    for li, (planes, nblocks, stride) in enumerate(oxv.HALF_LAYERS, start=1):
        conv3 = conv_winograd if li in wino_layers else conv_direct
        for bi in range(nblocks):
            x = block(x, sd, f"{p}.layer{li}.{bi}", stride if bi == 0 else 1, conv3)
        taps[f"layer{li}"] = x
    return x, taps


def main():
    torch.set_num_threads(8)
    sd = seeded_state_dict("halfresnet34", 16, seed=1234)
    rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
    print("feats      path                  layer2     layer3     layer4     x-vector 1-cos")
    for seed, T in ((55, 401), (57, 801)):
        feats = torch.randn(1, 80, T, generator=torch.Generator().manual_seed(seed))
        with torch.no_grad():
            ref_taps = {}
            _, ref_emb = oxv.halfresnet34_from_feats(feats, sd, taps=ref_taps)
            for name, layers in (("direct (emulation)", ()), ("Winograd layer 3", (3,)), ("Winograd layers 3+4", (3, 4))):
                x, taps = trunk(feats, sd, layers)
                e = oxv._bn(F.linear(oxv.attentive_pooling(x, sd), sd["before_speaker_embedding.lin_be.weight"]), sd, "before_speaker_embedding.bn_be")
                cos = f"{1 - float(F.cosine_similarity(oxv.l2_norm(e), ref_emb)):.2e}"
                print(f"T={T:4d}    {name:20s}  " + "  ".join(f"{rel(taps[k], ref_taps[k]):.2e}" for k in ("layer2", "layer3", "layer4")) + f"   {cos}", flush=True)


if __name__ == "__main__":
    main()
