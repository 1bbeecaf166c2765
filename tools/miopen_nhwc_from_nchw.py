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
        if nk in ufdb:
            continue  # a record that is already there (possibly from a real channels-last find) is never overwritten
        ufdb[nk] = ";".join(f"{n}:{t:.6g},{ws},{algo}" for n, (t, ws, algo) in sorted(keep.items(), key=lambda kv: kv[1][0]))
        n_f += 1
    for k, v in list(udb.items()):
        if k.endswith("xNCHWxBF16xF") and (k[: -len("xNCHWxBF16xF")] + "xNHWCxBF16xF") not in udb:
            udb[k[: -len("xNCHWxBF16xF")] + "xNHWCxBF16xF"] = v
            n_p += 1
    write(ufdb_path, ufdb)
    write(udb_path, udb)
    print(f"{n_f} NHWC find-db records, {n_p} NHWC perf-db records written "
          f"({len(ufdb)} / {len(udb)} records in total)")


if __name__ == "__main__" and not (len(sys.argv) > 1 and sys.argv[1] in ("borrow", "derive-find")):
    main()  # `miopen_nhwc_from_nchw.py borrow FP16` runs only the second step, for that dtype


# ---------------------------------------------------------------------------------------------------------------------
# Second step: perf-db records for conv problems that were never tuned.
#
# Measured (gpurun session 7, profiles/r2_s7_*): for a channels-last bf16 problem MIOpen's immediate mode runs the CK
# grouped-conv solver whether or not the user find-db ranks it first; WITH a perf-db record it runs the tuned instance
# (0.8 ms on 320->320 3x3 at batch 20), WITHOUT one a default instance that is ~40x slower (33.9 ms per call at batch 32:
# 1.26 s of a 1.49 s forward).  The tuned instances exist only for the batch sizes a find pass was run at (20, 6, 10, 3).
# A CK instance is a GEMM tile configuration; the implicit-GEMM dimensions it tiles are M = N*Ho*Wo, N = Cout,
# K = Cin*kh*kw, so the instance tuned for one batch size of a convolution is a sound (if not provably optimal) choice for
# the same convolution at another batch size, and -- one step further -- for the same (Cin, Cout, kernel, stride) at another
# spatial size.  This writes such borrowed records for every bf16 problem in the find-db that has none, and for a grid of
# batch sizes (the per-rank batches of row sharding and of several images in flight) of every UNet convolution shape.
# ---------------------------------------------------------------------------------------------------------------------
EXTRA_BATCHES = (1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 13, 14, 16, 18, 20, 24, 26, 32, 36, 40, 52)


def _parse_udb_key(k):
    f = k.split("x")
    # [2, Cin, H, W, 1, kh, kw, 1, Cout, N, ph, pw, pd, sh, sw, sd, dh, dw, dd, bias, group, layout, dtype, dir]
    return dict(cin=int(f[1]), H=int(f[2]), W=int(f[3]), kh=int(f[5]), kw=int(f[6]), cout=int(f[8]), n=int(f[9]),
                pad=(f[10], f[11]), stride=(f[13], f[14]), layout=f[21], dtype=f[22], fields=f)


def borrow_ck_instances(dtype="BF16"):
    """``dtype`` = "BF16" or "FP16" (round 3: the fp16 UNet's records come from a find pass at batch 20 / 6; the per-rank
    batches of row sharding -- 10, 3, ... -- borrow from them exactly as the bf16 ones did)."""
    udb_path = glob.glob(os.path.join(CACHE, "*.udb.txt"))[0]
    ufdb_path = glob.glob(os.path.join(CACHE, "*.ufdb.txt"))[0]
    udb, ufdb = read(udb_path), read(ufdb_path)
    donors = {}  # (cin, cout, kh, kw, stride, pad) -> list of (H, W, n, instance)
    for k, v in udb.items():
        if f"x{dtype}xF" not in k:
            continue
        p = _parse_udb_key(k)
        inst = [r for r in v.split(";") if r.startswith(CK + ":")]
        if inst:
            donors.setdefault((p["cin"], p["cout"], p["kh"], p["kw"], p["stride"], p["pad"]), []).append(
                (p["H"], p["W"], p["n"], inst[0]))
    # every distinct bf16 convolution shape the find-db has seen (either layout, any batch)
    shapes = set()
    for k in ufdb:
        # every 16-bit convolution shape the find-db has seen in EITHER 16-bit dtype is wanted for this dtype (the fp16
        # find pass of round 3 covered only the headline workload; SD1.5's and cfg4's shapes were seen in bf16)
        if not (k.endswith(f"-{dtype}-F") or k.endswith("-BF16-F")):
            continue
        f = k.split("-")
        cin, H, W, kk, cout, Ho, Wo, n, pad, stride = f[0], f[1], f[2], f[3], f[4], f[5], f[6], f[7], f[8], f[9]
        kh, kw = kk.split("x")
        shapes.add((int(cin), int(H), int(W), int(kh), int(kw), int(cout), tuple(pad.split("x")), tuple(stride.split("x")),
                    int(n)))
    base = {s[:-1] for s in shapes}
    want = {s for s in shapes} | {b + (n,) for b in base for n in EXTRA_BATCHES}
    added = 0
    for (cin, H, W, kh, kw, cout, pad, stride, n) in sorted(want):
        cand = donors.get((cin, cout, kh, kw, stride, pad))
        if not cand:
            continue
        # same spatial size first, then nearest batch
        inst = min(cand, key=lambda c: ((c[0], c[1]) != (H, W), abs(c[0] * c[1] - H * W), abs(c[2] - n)))[3]
        for layout in ("NHWC", "NCHW"):
            key = "x".join(str(v) for v in (2, cin, H, W, 1, kh, kw, 1, cout, n, pad[0], pad[1], 0, stride[0], stride[1], 0,
                                            1, 1, 0, 0, 1, layout, dtype, "F"))
            cur = udb.get(key, "")
            if (CK + ":") in cur:
                continue
            udb[key] = (cur + ";" if cur else "") + inst
            added += 1
    write(udb_path, udb)
    print(f"{added} borrowed {dtype} CK perf-db records written ({len(udb)} records in total)")


# ---------------------------------------------------------------------------------------------------------------------
# Third step (round 3): find-db records for the batch sizes that were never searched.
#
# Measured in round-3 session 4: when the user find-db has NO record for a problem, MIOpen's immediate mode does not just
# take a default -- it runs a full find for it on first use (the "30-130 s one-off" of round 2; 113 s for the SDXL UNet at
# batch 10 in fp16), and eight ranks doing that at once on a shared GPU blew a 15-minute limit.  A find-db record is a
# ranking of solvers with their times; the ranking of a convolution at one batch size is a sound stand-in for the same
# convolution at another (the CK solver wins almost everywhere, and its instance comes from the borrowed perf-db record),
# so every 16-bit channels-last record is copied to the batch sizes of EXTRA_BATCHES that have none, times scaled by the
# batch ratio.  Real finds (20, 6, 10, 3) are never overwritten.
# ---------------------------------------------------------------------------------------------------------------------
def derive_find_records(dtype="FP16"):
    ufdb_path = glob.glob(os.path.join(CACHE, "*.ufdb.txt"))[0]
    ufdb = read(ufdb_path)
    suffix = f"-NHWC-NHWC-NHWC-{dtype}-F"
    by_shape = {}
    for k, v in ufdb.items():
        if not k.endswith(suffix):
            continue
        f = k[: -len(suffix)].split("-")
        if int(f[0]) < 320 and int(f[4]) < 320:
            continue  # only the full-size UNet / ControlNet convolutions
        by_shape.setdefault(tuple(f[:7] + f[8:]), []).append((int(f[7]), v))
    added = 0
    for shape, recs in by_shape.items():
        have = {n for n, _ in recs}
        for n in EXTRA_BATCHES:
            if n in have:
                continue
            n0, v = min(recs, key=lambda r: abs(r[0] - n))
            out = []
            for rec in v.split(";"):
                name, rest = rec.split(":", 1)
                t, ws, algo = rest.split(",")
                # workspace: scaled with the batch, rounded UP to 256 B (a truncated estimate could be smaller than what the
                # solver needs for the derived batch -- ADVICE r3); zero stays zero
                ws_n = -(-int(ws) * n // n0)
                ws_n = -(-ws_n // 256) * 256 if ws_n else 0
                out.append(f"{name}:{float(t) * n / n0:.6g},{ws_n},{algo}")
            key = "-".join(list(shape[:7]) + [str(n)] + list(shape[7:])) + suffix
            ufdb[key] = ";".join(out)
            added += 1
    write(ufdb_path, ufdb)
    print(f"{added} derived {dtype} find-db records written ({len(ufdb)} records in total)")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "derive-find":
        derive_find_records(sys.argv[2] if len(sys.argv) > 2 else "FP16")
    else:
        borrow_ck_instances(sys.argv[2] if len(sys.argv) > 2 and sys.argv[1] == "borrow" else "BF16")
