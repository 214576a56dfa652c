"""GPU box: the extraction CLI started through torch.distributed.run (1 rank: RCCL init + gather path with world 1) against the
plain invocation -- same ark bytes.  Writes 40 synthetic wavs + a seeded checkpoint under /tmp."""
import os, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy, scipy.io.wavfile, torch
from sidekit_amd.nnet.weights import seeded_state_dict
d = tempfile.mkdtemp(prefix="skcli_", dir="/tmp")
sd = seeded_state_dict("halfresnet34", 16, seed=5)
torch.save({"speaker_number": 16, "model_archi": {"model_type": "halfresnet34", "loss": {"type": "aam"}}, "model_state_dict": sd}, f"{d}/model.pt")
rs = numpy.random.RandomState(0)
with open(f"{d}/wav.scp", "w") as f:
    for i in range(40):
        x = (rs.randn(rs.randint(16000, 70000)) * 3000).astype(numpy.int16)
        scipy.io.wavfile.write(f"{d}/u{i}.wav", 16000, x)
        f.write(f"utt{i} {d}/u{i}.wav\n")
common = ["--model", f"{d}/model.pt", "--wav-scp", f"{d}/wav.scp", "--device", "cuda", "--batch-size", "16", "--dtype", "bf16"]
env = dict(os.environ, PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
subprocess.run([sys.executable, "-m", "sidekit_amd.bin.extract_xvectors", *common, "--out-scp", f"{d}/a.scp"], check=True, env=env)
subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", "29541",
                "-m", "sidekit_amd.bin.extract_xvectors", *common, "--out-scp", f"{d}/b.scp"], check=True, env=env)
a, b = open(f"{d}/a.ark", "rb").read(), open(f"{d}/b.ark", "rb").read()
print("ark bytes", len(a), "identical" if a == b else "DIFFERENT")
assert a == b and len(a) > 40 * 1024
