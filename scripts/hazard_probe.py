"""GPU box: the probes of the round-3 hunt for the packed-f32 / MFMA hazard that still run against the shipped library, as ONE script
(the verbatim output of the hunt's ten scripts and LDS canaries is profiles/r03_two_lane_frontend_hazard.txt).

    python scripts/hazard_probe.py [probe ...]          probes: corunner  beside-forward  trunk   (default: all three)

  corunner        the log-mel front-end on a side stream beside synthetic co-runners on the default stream (scripts/diag/aggressor_lib.hip:
                  LDS-fed MFMAs, register-operand MFMAs, ds_read_b128 spam) -- the experiment that isolated the hazard: register-operand bf16
                  MFMAs of ANOTHER kernel were enough to corrupt SLP-formed packed-f32 arithmetic;
  beside-forward  both front-ends on a side stream beside a second model's bf16 trunk / whole forward, 30 trials;
  trunk           stem + trunk (fp32 and bf16 taps) on a side stream beside the register-operand MFMA co-runner.
Every probe prints how many results differed from the same computation run alone; the shipped library gives 0 everywhere."""
import ctypes, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SIDEKIT_AMD_LANES"] = "1"
import torch
from sidekit_amd.nnet import Xtractor

dev = torch.device("cuda", 0)


def aggressor():
    """scripts/diag/aggressor_lib.hip built on demand (no binary in the tree)."""
    src = os.path.join(ROOT, "scripts", "diag", "aggressor_lib.hip")
    so = os.path.join(ROOT, "gpurun_out", "libaggressor.so")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        os.makedirs(os.path.dirname(so), exist_ok=True)
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, src], check=True)
    ag = ctypes.CDLL(so)
    ag.aggressor_launch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    return ag


def corunner():
    ag = aggressor()
    m = Xtractor(7205, model_archi="halfresnet34", loss="aam", seed=1234).to(dev).eval()
    b = 0.1 * torch.randn(128, 64000, device=dev, generator=torch.Generator(device=dev).manual_seed(0))
    ref = m.features(b)
    sink = torch.zeros(4, device=dev)
    torch.cuda.synchronize()
    side, main = torch.cuda.Stream(), torch.cuda.current_stream()
    for mode, name in ((3, "random-operand MFMA fed from LDS"), (2, "MFMA, register operands"), (1, "ds_read_b128 spam")):
        nbad, worst = 0, 0.0
        for _ in range(12):
            ag.aggressor_launch(ctypes.c_void_p(main.cuda_stream), mode, 1024, 4000, ctypes.c_void_p(sink.data_ptr()))
            with torch.cuda.stream(side):
                outs = [m.features(b) for _ in range(3)]
            torch.cuda.synchronize()
            for o in outs:
                d = (o - ref).abs()
                nbad += int((d.amax(dim=(1, 2)) > 0).sum()); worst = max(worst, float(d.max()))
        print(f"co-runner {name:34s}: utterances whose features differed {nbad} of {12 * 3 * 128}, max abs diff {worst:.3e}", flush=True)


def beside_forward():
    m1 = Xtractor(7205, model_archi="halfresnet34", loss="aam", seed=1234).to(dev).eval()
    m2 = Xtractor(7205, model_archi="halfresnet34", loss="aam", seed=1234).to(dev).eval()
    m3 = Xtractor(64, model_archi="xvector", loss="aam", seed=4321).to(dev).eval()
    wav = 0.1 * torch.randn(256, 64000, device=dev, generator=torch.Generator(device=dev).manual_seed(0))
    a, b = wav[:128].contiguous(), wav[128:].contiguous()
    ref2, ref3 = m2.features(b), m3.features(b)
    feats_a = m1.features(a)
    m1.compute_dtype = "bf16"
    for _ in range(2): m1.forward_features(feats_a); m1(a, is_eval=True)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    for name, aggr in (("bf16 trunk", lambda: m1.forward_features(feats_a)), ("bf16 forward", lambda: m1(a, is_eval=True))):
        bad2 = bad3 = 0
        for _ in range(30):
            aggr()
            with torch.cuda.stream(side):
                f2 = m2.features(b); f3 = m3.features(b)
            torch.cuda.synchronize()
            bad2 += int(((f2 - ref2).abs().amax(dim=(1, 2)) > 0).sum()); bad3 += int(((f3 - ref3).abs().amax(dim=(1, 2)) > 0).sum())
        print(f"beside the {name}: utterances whose features differed over 30 trials: log-mel front-end {bad2}, MFCC front-end {bad3} (of 3840 each)", flush=True)
    for model, x in ((m2, b), (m3, b)):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): model.features(x)
        torch.cuda.synchronize()
        print(f"{model.model_archi} front-end alone: {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms per 128 x 4 s", flush=True)


def trunk():
    ag = aggressor()
    m = Xtractor(64, model_archi="halfresnet34", loss="aam", seed=1234).to(dev).eval()
    b = 0.1 * torch.randn(128, 64000, device=dev, generator=torch.Generator(device=dev).manual_seed(0))
    feats = m.features(b)
    sink = torch.zeros(4, device=dev)
    main, side = torch.cuda.current_stream(), torch.cuda.Stream()
    for dtype in ("bf16", "fp32"):
        m.compute_dtype = dtype
        m.set_debug(True)
        m.forward_features(feats); torch.cuda.synchronize()
        ref = {k: v.copy() for k, v in m.debug_taps(["stem", "layer1", "layer4", "pooled"]).items()}
        bad = {k: 0 for k in ref}
        for _ in range(20):
            ag.aggressor_launch(ctypes.c_void_p(main.cuda_stream), 2, 1024, 6000 if dtype == "bf16" else 30000, ctypes.c_void_p(sink.data_ptr()))
            with torch.cuda.stream(side):
                m.forward_features(feats)
            torch.cuda.synchronize()
            got = m.debug_taps(list(ref))
            for k in ref:
                bad[k] += int((got[k] != ref[k]).sum())
        m.set_debug(False)
        print(f"{dtype} trunk beside the register-operand MFMA co-runner, 20 trials: differing bytes per tap {bad}", flush=True)


if __name__ == "__main__":
    probes = {"corunner": corunner, "beside-forward": beside_forward, "trunk": trunk}
    for name in (sys.argv[1:] or list(probes)):
        print(f"== {name}", flush=True)
        probes[name]()
