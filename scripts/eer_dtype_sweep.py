"""GPU box: the corpus of bin/shard_extract_score through the fp32 and the bf16 trunk at several noise levels -> the three EERs of each
run side by side (calibration of tests/test_gpu_eer_dtype.py: which noise level puts the cosine EER where).
usage: python scripts/eer_dtype_sweep.py [utterances] [noise ...]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sidekit_amd.bin import shard_extract_score
from sidekit_amd.nnet import Xtractor

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
noises = [float(x) for x in sys.argv[2:]] or [0.0005, 0.001, 0.002, 0.004]
plda = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "config5.npz")
model = Xtractor(7205, model_archi="halfresnet34", loss="aam", seed=1234).to("cuda:0").eval()
for noise in noises:
    for plda_arg in ([], ["--plda", plda]):
        row = {}
        for dtype in ("fp32", "bf16"):
            model.compute_dtype = dtype
            out = shard_extract_score.main(["--utterances", str(N), "--batch", "256", "--seconds", "4", "--trials", "1000", "--noise", str(noise),
                                            "--dtype", dtype, "--all-pairs"] + plda_arg, model=model)
            row[dtype] = {k: out[k] for k in ("cosine_eer", "plda_eer", "all_pairs_eer")}
        d = {k: row["bf16"][k] - row["fp32"][k] for k in row["fp32"]}
        print("SWEEP " + json.dumps({"noise": noise, "utterances": N, "plda": "config5" if plda_arg else "moments", **row, "delta": d}), flush=True)
