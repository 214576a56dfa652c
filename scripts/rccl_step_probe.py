"""GPU box, under torch.distributed.run with ONE rank: what the per-step RCCL all-gather of bench.py costs, variant by variant.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29555 scripts/rccl_step_probe.py

Variants (20 steps each, best of 3, B = 256 x 4 s, bf16, product lanes):
  none        forward only
  wait        all_gather_into_tensor(async_op=True) + work.wait() on the compute stream (bench.py up to round 3)
  nowait      all_gather_into_tensor(async_op=True), the work handle is only waited for two steps later
  sync        all_gather_into_tensor (blocking form)
  lanes1_wait as `wait` with the forward on one stream
  every4      gather of the last four steps' x-vectors every fourth step (one collective per 1024 utterances)
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from sidekit_amd.nnet import Xtractor

local = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
dist.init_process_group("nccl", device_id=dev)
world = dist.get_world_size()
m = Xtractor(7205, model_archi="halfresnet34", loss="aam", seed=1234).to(dev).eval()
m.compute_dtype = "bf16"
g = torch.Generator(device=dev).manual_seed(0)
B = 256
wavs = [0.1 * torch.randn(B, 64000, device=dev, generator=g) for _ in range(5)]
gath = [torch.empty(world * B, 256, device=dev) for _ in range(2)]
gath4 = torch.empty(world * 4 * B, 256, device=dev)
for _ in range(3):
    m(wavs[0], is_eval=True)
torch.cuda.synchronize()


def run(variant, steps=20):
    pend, blocks = [], []
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        _, emb = m(wavs[k % 5], is_eval=True)
        if variant in ("wait", "lanes1_wait"):
            w = dist.all_gather_into_tensor(gath[k % 2], emb, async_op=True)
            w.wait()
            pend.append((w, emb))
        elif variant == "nowait":
            while len(pend) >= 2:
                pend.pop(0)[0].wait()
            w = dist.all_gather_into_tensor(gath[k % 2], emb, async_op=True)
            pend.append((w, emb))
        elif variant == "sync":
            dist.all_gather_into_tensor(gath[k % 2], emb)
        elif variant == "every4":
            blocks.append(emb)
            if len(blocks) == 4:
                w = dist.all_gather_into_tensor(gath4, torch.cat(blocks), async_op=True)
                pend.append((w, blocks))
                blocks = []
    t_host = time.perf_counter() - t0
    for w, _ in pend:
        w.wait()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3, t_host / steps * 1e3


for variant in (sys.argv[1].split(",") if len(sys.argv) > 1 else ("none", "wait", "nowait", "sync", "every4", "lanes1_none", "lanes1_wait", "none")):
    if variant.startswith("lanes1"):
        m.set_lanes(1)
        m(wavs[0], is_eval=True)
    best = min(run(variant.replace("lanes1_", "") if variant == "lanes1_none" else variant) for _ in range(3))
    print(f"{variant:12s} {best[0]:.3f} ms per step (host enqueue {best[1]:.3f} ms)", flush=True)
    if variant.startswith("lanes1"):
        m.set_lanes(2)
dist.destroy_process_group()
