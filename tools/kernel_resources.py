"""Per-kernel register / LDS / spill summary of a .hip file compiled for gfx950 (no GPU needed).
usage: kernel_resources.py file.hip [extra hipcc flags...]"""
import re
import subprocess
import sys

src, extra = sys.argv[1], sys.argv[2:]
cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-I", "include", "-c", src, "-o", "/dev/null",
       "-Rpass-analysis=kernel-resource-usage"] + extra
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark:\s+(Function Name|Name): (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(2)], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r"\(anonymous namespace\)::", "", cur)
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\S+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = m.group(2)
for name, r in rows.items():
    print(f"{name[:90]:90s} VGPR {r.get('VGPRs', '?'):>4} AGPR {r.get('AGPRs', '?'):>3} SGPR {r.get('TotalSGPRs', r.get('SGPRs', '?')):>3} "
          f"spill {r.get('VGPR Spill', r.get('VGPRs Spill', '?'))} scratch {r.get('ScratchSize [bytes/lane]', '?')} "
          f"occ {r.get('Occupancy [waves/SIMD]', '?')} LDS {r.get('LDS Size [bytes/block]', '?')}")
