"""profiles/<round>_pmc_hbm_bytes*.json (scripts/pmc_summary.py) -> profiles/traffic.json: mean HBM bytes per launch of
each bench.py profile slot (conv_L1 ...), what bench.py reports as roofline.traffic for the dominant kernel.

usage: python scripts/traffic_from_pmc.py profiles/r01_pmc_hbm_bytes_bf16_b256.json > profiles/traffic.json
"""
import json, re, sys, collections
src = sys.argv[1]
pmc = json.load(open(src))
SLOT = {(32, 32, 1, 9): "conv_L1", (32, 32, 1, 1): "conv_L1S", (32, 64, 2, 9): "conv_L2A", (32, 64, 2, 1): "conv_L2S",
        (64, 64, 1, 9): "conv_L2", (64, 128, 2, 9): "conv_L3A", (64, 128, 2, 1): "conv_L3S", (128, 128, 1, 9): "conv_L3",
        (128, 256, 2, 9): "conv_L4A", (128, 256, 2, 1): "conv_L4S", (256, 256, 1, 9): "conv_L4"}
tot = collections.defaultdict(lambda: [0.0, 0])
for name, v in pmc.items():
    m = re.search(r"ConvCfg<sk::bf16_t, (\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+)", name)
    if not m:
        continue
    a = [int(x) for x in m.groups()]
    slot = SLOT.get((a[0], a[1], a[2], a[10]))
    if slot:
        tot[slot][0] += v["hbm_bytes_per_launch"] * v["launches"]
        tot[slot][1] += v["launches"]
out = {"_note": "mean HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes; FETCH x2 = the gfx950 "
                "correction of MI355X_MICROARCH.md), bench.py B=256 4 s bf16; source " + src,
       "bf16": {k: tot[k][0] / tot[k][1] for k in sorted(tot)}}
json.dump(out, sys.stdout, indent=1)
