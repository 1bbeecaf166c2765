"""Summarise rocprofv3 --pmc passes (one *_counter_collection.csv per pass) into profiles/<name>.json:
mean counter value per launch for the hand-written kernels (second half of the launches: the first repetition warms).

usage: pmc_summarise.py out.json pass_dir [pass_dir ...]
HBM bytes per launch follow MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE are KiB on rocprofv3; on gfx950
FETCH_SIZE tallies the 128-B requests of wide (16 B / lane) coalesced reads at 64 B, so hbm_bytes = (2*FETCH + WRITE)*1024
for these streaming kernels (all of them read with 16-byte vectors); the uncorrected sum is kept next to it."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

ENTRY = {"k_flash_attn_fwd": "ed_flash_attention", "k_flash_attn_pipe": "ed_flash_attention",
         "k_flash_attn_smallkv": "ed_flash_attention", "k_bias_residual_add_cl": "ed_bias_residual_add",
         "k_softmax_rows_f32": "ed_softmax_rows", "k_geglu": "ed_geglu", "k_groupnorm": "ed_groupnorm",
         "k_gn_split_stats": "ed_groupnorm[split stats]", "k_gn_split_apply": "ed_groupnorm[split apply]",
         "k_gn_nhwc_partial": "ed_groupnorm_nhwc[partial]", "k_gn_nhwc_apply": "ed_groupnorm_nhwc[apply]",
         "k_add_layernorm": "ed_add_layernorm", "k_layernorm": "ed_layernorm", "k_tokens_add_nchw": "ed_tokens_add_nchw",
         "k_bias_residual_add": "ed_bias_residual_add", "k_assemble_rows": "ed_assemble_rows",
         "k_phase_epilogue": "ed_phase_epilogue", "k_pick_assemble": "ed_pick_assemble",
         "k_gather_windows": "ed_gather_views", "k_undo_v4": "ed_undo_step", "k_unpad_direction": "ed_unpad_direction",
         "k_fill_directions": "ed_fill_directions", "k_scatter_centres": "ed_scatter_centres", "k_cfg_ddim": "ed_cfg_ddim_step",
         "k_rrg_update": "ed_rrg_update"}


def entry_of(kernel):
    m = re.search(r"k_gemm_8phase<([^>]*)>", kernel)   # one main loop: <T, EPI, CONV, ADD, OUT32, TWO> (round 5; defaults may be elided)
    if m:
        args = [a.strip() for a in m.group(1).split(",")]
        epi, conv = args[1], args[2] in ("true", "1", "2", "3")     # CONV: bool until round 5; round 6: 0 GEMM, 1 conv, 2 conv of the 2x upsampling, 3 stride 2
        out32 = len(args) > 4 and args[4] == "true"
        if epi == "0":
            return "ed_geglu_gemm"
        if args[2] == "2":
            return "ed_conv3x3_nhwc_up2x"
        if args[2] == "3":
            return "ed_conv3x3_nhwc_s2"
        return "ed_conv3x3_nhwc_f32out" if (conv and out32) else ("ed_conv3x3_nhwc" if conv else "ed_linear")
    if "k_geglu_persist" in kernel:
        return "ed_geglu_gemm"
    if "k_gn32_nhwc_partial" in kernel:
        return "ed_groupnorm_nhwc_f32[partial]"
    if "k_gn32_nhwc_apply" in kernel:
        return "ed_groupnorm_nhwc_f32[apply]"
    for k, v in ENTRY.items():  # dict order: longer, more specific names come before their prefixes
        if re.search(k + r"(?![a-z])", kernel):
            return v
    return None


workload = "sdxl_1024x2048"
if "--workload" in sys.argv:
    i = sys.argv.index("--workload")
    workload = sys.argv[i + 1]
    del sys.argv[i:i + 2]
rows = [20, 6]          # rows of the phase-A / phase-B forwards the passes ran (tools/pmc_unet.py ROWS); bench.py matches them with its own
if "--rows" in sys.argv:
    i = sys.argv.index("--rows")
    rows = [int(v) for v in sys.argv[i + 1].split(",")]
    del sys.argv[i:i + 2]
out_path, dirs = sys.argv[1], sys.argv[2:]
vals = defaultdict(lambda: defaultdict(list))
for d in dirs:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            e = entry_of(row.get("Kernel_Name", ""))
            if e:
                vals[e][row["Counter_Name"]].append(float(row["Counter_Value"]))
res = {}
for e, counters in vals.items():
    r = {}
    for c, v in counters.items():
        half = v[len(v) // 2:]
        r[c + "_mean"] = sum(half) / len(half)
        r["launches_counted"] = len(half)
    if "FETCH_SIZE_mean" in r and "WRITE_SIZE_mean" in r:
        r["hbm_bytes_per_launch_mean"] = round((2 * r["FETCH_SIZE_mean"] + r["WRITE_SIZE_mean"]) * 1024)
        r["hbm_bytes_per_launch_uncorrected"] = round((r["FETCH_SIZE_mean"] + r["WRITE_SIZE_mean"]) * 1024)
    if "SQ_VALU_MFMA_BUSY_CYCLES_mean" in r and "SQ_BUSY_CU_CYCLES_mean" in r and r["SQ_BUSY_CU_CYCLES_mean"]:
        r["mfma_busy_over_cu_busy"] = round(r["SQ_VALU_MFMA_BUSY_CYCLES_mean"] / r["SQ_BUSY_CU_CYCLES_mean"], 4)
    res[e] = r
json.dump({"workload": workload, "rows": rows, "note": __doc__.strip().split("usage")[0].strip() + (f" Launch mix: one phase-A ({rows[0]}-row) and one phase-B ({rows[1]}-row) "
                   "SDXL forward of the 1024x2048 workload (tools/pmc_unet.py)." if workload == "sdxl_1024x2048" else
                   f" Launch mix: the second of two eager denoising timesteps of bench.py's workload {workload} (tools/pmc_workload.py)."), "kernels": res}, open(out_path, "w"), indent=1)
print(json.dumps({k: {c: round(v, 1) if isinstance(v, float) else v for c, v in r.items()} for k, r in res.items()}, indent=1))
