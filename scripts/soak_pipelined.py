"""GPU box: soak of the pipelined forwards.  Random batch shapes (1..300 utterances, 0.3..6 s, ragged or uniform, float32 or int16 PCM) go through
Xtractor.submit / collect two in flight, interleaved at random with plain (two-lane) forwards on the same handle; every result is compared,
bit for bit, with a second model instance that runs one forward at a time on one stream.  usage: python scripts/soak_pipelined.py [iterations]"""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sidekit_amd.nnet import Xtractor


def main(n_iter=200, verbose=True, arch="halfresnet34", dtype="bf16"):
    """-> number of mismatching results (0 = every pipelined / plain forward equals the one-at-a-time reference, bit for bit)."""
    dev = torch.device("cuda", 0)
    a = Xtractor(64, model_archi=arch, loss="aam", seed=99).to(dev).eval()
    b = Xtractor(64, model_archi=arch, loss="aam", seed=99).to(dev).eval()
    a.compute_dtype = b.compute_dtype = dtype
    b.set_lanes(1)
    rnd = random.Random(5)
    g = torch.Generator(device=dev).manual_seed(5)
    pend, t0 = [], time.time()
    stat = {"bad": 0, "n": 0}

    def check(out, want, what):
        stat["n"] += 1
        if not (torch.equal(out[0], want[0]) and torch.equal(out[1], want[1])):
            stat["bad"] += 1
            print("MISMATCH", what, int((out[1] != want[1]).any(dim=1).sum()), "rows", flush=True)

    for it in range(n_iter):
        B = rnd.choice([1, 3, 17, 64, 127, 128, 129, 200, 256, 300])
        L = rnd.choice([16000, 33333, 64000, 96000] if arch == "xvector" else [4800, 16000, 33333, 64000, 96000])
        if B * L > 256 * 64000:
            B = max(1, 256 * 64000 // L)
        wav = 0.1 * torch.randn(B, L, device=dev, generator=g)
        if rnd.random() < 0.3:
            wav = (wav * 32768.0).round().clamp(-32768, 32767).to(torch.int16)
        lens = [rnd.randint(max(8000 if arch == "xvector" else 600, L // 3), L) for _ in range(B)] if rnd.random() < 0.5 else None
        want = tuple(t.clone() for t in b(wav, is_eval=True, lengths=lens))
        if rnd.random() < 0.7:
            pend.append((a.submit(wav, lengths=lens), want, (B, L)))
            if len(pend) == a.pipeline_depth:
                tk, w, what = pend.pop(0)
                check(a.collect(tk), w, ("pipelined", what))
        else:
            check(a(wav, is_eval=True, lengths=lens), want, ("plain", (B, L)))
        if verbose and it % 50 == 49:
            torch.cuda.synchronize()
            print(f"iteration {it + 1}: {stat['n']} results compared, {stat['bad']} mismatches, {time.time() - t0:.0f} s", flush=True)
    while pend:
        tk, w, what = pend.pop(0)
        check(a.collect(tk), w, ("pipelined", what))
    torch.cuda.synchronize()
    print(f"done: {stat['n']} results compared, {stat['bad']} mismatches")
    return stat["bad"]


if __name__ == "__main__":
    sys.exit(1 if main(int(sys.argv[1]) if len(sys.argv) > 1 else 200, True, *(sys.argv[2:4])) else 0)
