"""Instruction mix per basic block of the gfx950 ISA hipcc emits for a .hip file (no GPU needed): where the scratch (spill)
traffic sits and what the hot loop bodies issue.   usage: isa_blocks.py file.hip <kernel-name substring> [min block size]"""
import re
import subprocess
import sys

src, pat = sys.argv[1], sys.argv[2]
min_len = int(sys.argv[3]) if len(sys.argv) > 3 else 120
asm = "/tmp/isa_blocks.s"
subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-I", "include", "-S",
                "--cuda-device-only", src, "-o", asm], check=True, capture_output=True)
text = open(asm).read()
starts = [(m.start(), m.group(1)) for m in re.finditer(r"^(_Z\w+):", text, re.M)]
for k, (pos, name) in enumerate(starts):
    if pat not in name:
        continue
    body = text[pos: starts[k + 1][0] if k + 1 < len(starts) else len(text)]
    body = body.split(".Lfunc_end")[0]
    blocks, lab, n_split = {}, "entry", [0]
    for line in body.splitlines()[1:]:
        m = re.match(r"^(\.LBB\d+_\d+):", line)
        if m:
            lab = m.group(1)
            continue
        t = line.strip().split()
        if not t or t[0].startswith((".", ";", "//")):
            continue
        blocks.setdefault(lab, []).append(t[0])
        if t[0].startswith(("s_cbranch", "s_branch")):   # a block also ends at a branch: what follows is only reached by fall-through
            n_split[0] += 1
            lab = f"{lab.split('+')[0]}+{n_split[0]}"
    print(subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip())
    for b, ins in blocks.items():
        sc = sum(x.startswith("scratch_") for x in ins)
        if len(ins) < min_len and not sc:
            continue
        c = lambda f: sum(1 for x in ins if f(x))
        print(f"  {b:10s} n={len(ins):4d} mfma={c(lambda x: 'mfma' in x):3d} exp={c(lambda x: x.startswith('v_exp')):3d} "
              f"valu={c(lambda x: x.startswith('v_') and 'mfma' not in x):4d} ds={c(lambda x: x.startswith('ds_')):3d} "
              f"vmem={c(lambda x: x.startswith(('buffer_', 'global_'))):2d} salu={c(lambda x: x.startswith('s_')):3d} "
              f"waitcnt={c(lambda x: x == 's_waitcnt'):2d} nop={c(lambda x: x == 's_nop'):2d} scratch={sc}")
