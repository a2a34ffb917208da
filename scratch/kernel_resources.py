"""Per-kernel resource table from `hipcc -Rpass-analysis=kernel-resource-usage` output (stderr of a compile).
    python scratch/kernel_resources.py res.txt [filter]"""
import re, subprocess, sys

text = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
blocks = re.split(r"remark: [^\n]*Function Name: ", text)[1:]
rows = []
for b in blocks:
    name = b.split()[0]
    try:
        name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip()
    except Exception:
        pass
    name = re.sub(r"\(.*", "", name).replace("void drt::", "")
    g = lambda k: (re.search(k + r": (\d+)", b) or [0, "?"])[1]
    rows.append((name, g("VGPRs"), g("AGPRs"), g("VGPRs Spill"), g("SGPRs Spill"), g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")))
print(f"{'kernel':70s} VGPR AGPR vspill sspill scratch occ lds")
for r in rows:
    if flt in r[0]:
        print(f"{r[0][:70]:70s} {r[1]:>4} {r[2]:>4} {r[3]:>6} {r[4]:>6} {r[5]:>7} {r[6]:>3} {r[7]:>5}")
