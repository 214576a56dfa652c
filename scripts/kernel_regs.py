"""Register / scratch / LDS use of every kernel in a HIP source (CPU side: hipcc -S, no GPU needed).

usage: python scripts/kernel_regs.py sidekit_amd/csrc/conv3x3.hip [filter]
"""
import re, subprocess, sys, tempfile, os
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, "k.s")
    subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-slp-vectorize", "-mllvm", "-pragma-unroll-threshold=65536", "-I" + os.path.join(root, "include"),
                    "--offload-device-only", "-S", src, "-o", out], check=True, stderr=subprocess.DEVNULL)
    s = open(out).read()
for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", s, re.S):
    name, body = m.group(1), m.group(2)
    d = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    if flt and flt not in d:
        continue
    def g(k):
        r = re.search(k + r"\s+(\d+)", body)
        return int(r.group(1)) if r else -1
    print("%-150s regs %3d  scratch %5d  lds %6d" % (d[:150], g(r"\.amdhsa_next_free_vgpr"), g(r"\.amdhsa_private_segment_fixed_size"),
                                                   g(r"\.amdhsa_group_segment_fixed_size")))
