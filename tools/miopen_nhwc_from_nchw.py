"""Derive channels-last (NHWC) entries of the in-tree MIOpen user db from the tuned NCHW ones -- no GPU needed.

Why this is sound: for an NCHW problem MIOpen's best solver on gfx950 is almost always the CK grouped-conv xdl kernel
(`ConvHipImplicitGemmGroupFwdXdlops`), which is an NHWC kernel wrapped in layout transposes (the `batched_transpose_*`
launches in the profiles).  tools/miopen_nhwc_diag.py measured, on the GPU, what MIOpen itself records for the NHWC
form of a problem after a search: the SAME CK instance string under the perf-db key with `xNHWCx` in place of `xNCHWx`,
and a find-db entry `...-NHWC-NHWC-NHWC-BF16-F` with the CK solver first, workspace 0 (no transposes), 0.80 ms vs
0.99 ms.  This script writes exactly those two kinds of records for every tuned NCHW bf16 problem.

    python tools/miopen_nhwc_from_nchw.py            (rewrites miopen_cache/*.ufdb.txt / *.udb.txt in place)
"""
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CACHE = os.path.join(ROOT, "miopen_cache")
CK = "ConvHipImplicitGemmGroupFwdXdlops"
ASM = "ConvAsmImplicitGemmGTCDynamicFwdXdlopsNHWC"


def read(path):
    out = {}
    for line in open(path):
        line = line.rstrip("\n")
        if "=" in line:
            k, v = line.split("=", 1)
            out[k] = v
    return out


def write(path, d):
    with open(path, "w") as f:
        for k, v in d.items():
            f.write(f"{k}={v}\n")


def main():
    ufdb_path = glob.glob(os.path.join(CACHE, "*.ufdb.txt"))[0]
    udb_path = glob.glob(os.path.join(CACHE, "*.udb.txt"))[0]
    ufdb, udb = read(ufdb_path), read(udb_path)
    n_f = n_p = 0
    for k, v in list(ufdb.items()):
        if not k.endswith("-NCHW-BF16-F"):
            continue
        sols = {}
        for rec in v.split(";"):
            name, rest = rec.split(":", 1)
            t, ws, algo = rest.split(",")
            sols[name] = (float(t), int(ws), algo)
        keep = {}
        if CK in sols:       # same CK instance without the two layout transposes (measured 0.80 / 0.99 = 0.81)
            keep[CK] = (sols[CK][0] * 0.81, 0, sols[CK][2])
        if ASM in sols:      # an NHWC kernel already; in the NCHW problem it too was wrapped in transposes
            keep[ASM] = (sols[ASM][0] * 0.85, sols[ASM][1] // 2, sols[ASM][2])
        if not keep:
            continue
        nk = k[: -len("-NCHW-BF16-F")] + "-NHWC-NHWC-NHWC-BF16-F"
        ufdb[nk] = ";".join(f"{n}:{t:.6g},{ws},{algo}" for n, (t, ws, algo) in sorted(keep.items(), key=lambda kv: kv[1][0]))
        n_f += 1
    for k, v in list(udb.items()):
        if k.endswith("xNCHWxBF16xF"):
            udb[k[: -len("xNCHWxBF16xF")] + "xNHWCxBF16xF"] = v
            n_p += 1
    write(ufdb_path, ufdb)
    write(udb_path, udb)
    print(f"{n_f} NHWC find-db records, {n_p} NHWC perf-db records written "
          f"({len(ufdb)} / {len(udb)} records in total)")


if __name__ == "__main__":
    main()
