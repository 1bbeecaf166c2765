"""Apply tools/r5_patches/*.patch to a scratch copy of the package sources and build the patched library NEXT TO the product's
(tools/r5_patches/build/libelastic_hip_patched.so; the product tree is not touched).  Fails if a patch no longer applies.
    python tools/r5_patches/build_patched.py [--only 0001,0003] [--out NAME.so]      # a subset (in order) / another file name under build/"""
import glob
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from elasticdiffusion_official_amd import _hip   # noqa: E402

only = None
name = "libelastic_hip_patched.so"
argv = sys.argv[1:]
while argv:
    flag = argv.pop(0)
    if flag == "--only":
        only = argv.pop(0).split(",")
    elif flag == "--out":
        name = argv.pop(0)
    else:
        raise SystemExit(f"unknown argument {flag}")
out = os.path.join(HERE, "build", name)
os.makedirs(os.path.dirname(out), exist_ok=True)
with tempfile.TemporaryDirectory() as tmp:
    shutil.copytree(os.path.join(ROOT, "elasticdiffusion_official_amd", "csrc"), os.path.join(tmp, "elasticdiffusion_official_amd", "csrc"))
    for p in sorted(glob.glob(os.path.join(HERE, "*.patch"))):
        if "-tests-" in os.path.basename(p):
            continue      # (the scratch copy holds csrc/ only)
        if only is not None and os.path.basename(p)[:4] not in only:
            continue
        if only is None and os.path.basename(p).startswith("0006"):
            continue      # optional: probe it first (--only 0001,0003,0005,0006 --out ...)
        subprocess.run(["patch", "-p1", "--no-backup-if-mismatch", "-i", p], cwd=tmp, check=True, stdout=subprocess.DEVNULL)
        print("applied", os.path.basename(p))
    srcs = [os.path.join(tmp, "elasticdiffusion_official_amd", "csrc", os.path.basename(s)) for s in _hip.SOURCES]
    subprocess.run(["hipcc", *_hip.HIPCC_FLAGS, "-I", _hip.INCLUDE, *srcs, "-o", out], check=True)
print("built", out)
