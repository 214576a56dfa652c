"""GPU box: consecutive forwards issued on ONE caller stream vs alternately on TWO caller streams with TWO model instances (own handles, own
workspaces: no sharing between the forwards in flight).  B = 256 x 4 s, bf16."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sidekit_amd.nnet import Xtractor
dev = torch.device("cuda", 0)
ms = [Xtractor(7205, model_archi="halfresnet34", loss="aam", seed=1234).to(dev).eval() for _ in range(3)]
for m in ms: m.compute_dtype = "bf16"
g = torch.Generator(device=dev).manual_seed(0)
wavs = [0.1 * torch.randn(256, 64000, device=dev, generator=g) for _ in range(5)]
for m in ms:
    for _ in range(3): m(wavs[0], is_eval=True)
torch.cuda.synchronize()
streams = [torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()]
ref = ms[0](wavs[1], is_eval=True)[1].clone()
for lanes in (1,):
    for m in ms: m.set_lanes(lanes)
    for mode, depth in (("one stream, one model", 1), ("two streams x two models alternating", 2), ("three streams x three models", 3)):
        best = 1e9
        for rep in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            outs = []
            for i in range(20):
                if mode.startswith("one"):
                    outs.append(ms[0](wavs[i % 5], is_eval=True)[1])
                else:
                    with torch.cuda.stream(streams[i % depth]):
                        outs.append(ms[i % depth](wavs[i % 5], is_eval=True)[1])
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 20 * 1e3)
        ok = torch.equal(outs[1], ref) and torch.equal(outs[16], ref)
        print(f"lanes={lanes} {mode}: {best:.3f} ms per step, outputs {'identical' if ok else 'DIFFERENT'}", flush=True)
