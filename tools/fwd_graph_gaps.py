"""How much of a hipGraph-replayed UNet forward is idle time BETWEEN kernels?

    python tools/fwd_graph_gaps.py run [batch] [family] [fp16|bf16]  # builds the UNet with the product's switches, captures one
                                                                    # forward (text k / v outside the graph), replays it 4 times
                                                                    # between marker kernels; prints the wall time per replay
    rocprofv3 --kernel-trace -d DIR -o NAME --output-format csv -- python tools/fwd_graph_gaps.py run 20
    python tools/fwd_graph_gaps.py analyse DIR/.../NAME_kernel_trace.csv   # -> one JSON line: per replay wall / busy / gaps

The marker is torch.cumsum on 7 integers (a scan kernel nothing in the forward launches).  Compare the un-profiled wall per replay with
the traced one before trusting the gaps: the tracer serialises completion signals."""
import collections
import csv
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(batch, fam, dtype="fp16"):
    import torch

    import elasticdiffusion_official_amd  # noqa: F401
    from elasticdiffusion_official_amd import models as M
    cfg = M.UNET_CONFIGS[fam]
    dt = torch.float16 if dtype == "fp16" else torch.bfloat16
    torch.manual_seed(0)
    unet = M.UNet2DConditionModel(**cfg).to("cuda", dt).eval().requires_grad_(False)
    if M.CHANNELS_LAST:
        unet = unet.to(memory_format=torch.channels_last)
    S = cfg["sample_size"]
    x = torch.randn(batch, 4, S, S, device="cuda", dtype=dt)
    e = torch.randn(batch, 77, cfg["cross_attention_dim"], device="cuda", dtype=dt)
    kw = None
    if cfg["pooled_projection_dim"]:
        kw = {"text_embeds": torch.randn(batch, cfg["pooled_projection_dim"], device="cuda", dtype=dt), "time_ids": torch.zeros(batch, 6, device="cuda")}
    t = torch.tensor(500, device="cuda")
    marker = torch.arange(7, device="cuda")
    with torch.no_grad():
        kv = unet.cross_attention_kv(e, None)
        fwd = lambda: unet(x, t, encoder_hidden_states=e, added_cond_kwargs=kw, cross_kv=kv).sample   # noqa: E731
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                fwd()
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = fwd()
    torch.cuda.synchronize()
    walls = []
    for _ in range(4):
        marker.cumsum(0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        g.replay()
        torch.cuda.synchronize()
        walls.append(round(1e3 * (time.perf_counter() - t0), 2))
    marker.cumsum(0)
    torch.cuda.synchronize()
    print(json.dumps({"family": fam, "batch": batch, "dtype": dtype, "replay_wall_ms_host_clock": walls, "finite": bool(torch.isfinite(out).all())}), flush=True)


def short(name):
    if "k_geglu_persist" in name:
        return "k_geglu_persist"
    for key in ("k_gemm_8phase", "k_flash_attn", "k_add_layernorm", "k_gn", "Cijk", "ck::", "k_bias_residual", "k_tokens_add", "elementwise",
                "CatArray", "upsample", "k_geglu", "miopen", "naive_conv", "igemm"):
        if key in name:
            if key == "k_gemm_8phase":
                args = [v.strip() for v in name.split("k_gemm_8phase<", 1)[1].split(">", 1)[0].split(",")]   # <T, EPI, CONV, ADD, OUT32, TWO>
                conv = len(args) > 2 and args[2] in ("true", "1", "2", "3")
                return "k_gemm_8phase" + ("<geglu>" if args[1] == "0" else ("<conv long-K>" if args[-1] == "true" and len(args) > 5 else "<conv>") if conv else "<linear>")
            return key
    return name.split("(")[0][-40:]


def analyse(path):
    ev = []
    with open(path) as f:
        for r in csv.DictReader(f):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    ev.sort()
    marks = [i for i, e in enumerate(ev) if "scan" in e[2].lower() or "cumsum" in e[2].lower()]
    reps = []
    for a, b in zip(marks[:-1], marks[1:]):
        seg = ev[a + 1:b]
        if len(seg) < 100:
            continue
        wall = seg[-1][1] - seg[0][0]
        busy = sum(e - s for s, e, _ in seg)
        gaps = [(s2 - e1, n1, n2) for (s1, e1, n1), (s2, e2, n2) in zip(seg[:-1], seg[1:])]
        pos = [g for g in gaps if g[0] > 0]
        hist = collections.Counter()
        for g in pos:
            hist["<1us" if g[0] < 1000 else "1-2us" if g[0] < 2000 else "2-4us" if g[0] < 4000 else "4-8us" if g[0] < 8000 else "8-20us" if g[0] < 20000 else ">20us"] += 1
        after = collections.defaultdict(lambda: [0, 0])
        before = collections.defaultdict(lambda: [0, 0])
        for d, n1, n2 in pos:
            after[short(n1)][0] += 1
            after[short(n1)][1] += d
            before[short(n2)][0] += 1
            before[short(n2)][1] += d
        top = lambda dd: {k: {"n": v[0], "total_us": round(v[1] / 1e3, 1), "mean_us": round(v[1] / v[0] / 1e3, 2)}   # noqa: E731
                          for k, v in sorted(dd.items(), key=lambda kv: -kv[1][1])[:8]}
        reps.append({"kernels": len(seg), "wall_ms": round(wall / 1e6, 3), "busy_ms": round(busy / 1e6, 3),
                     "gap_total_ms": round(sum(g[0] for g in pos) / 1e6, 3), "overlap_ms": round(-sum(g[0] for g in gaps if g[0] < 0) / 1e6, 3),
                     "gap_hist": dict(hist), "gap_after_kernel": top(after), "gap_before_kernel": top(before),
                     "short_kernels_lt10us": sum(1 for s, e, _ in seg if e - s < 10000),
                     "short_kernels_busy_ms": round(sum(e - s for s, e, _ in seg if e - s < 10000) / 1e6, 3)})
    # per-kernel totals of the LAST replay (full names cut to 110 characters: enough to tell the library's instances apart)
    fam = collections.defaultdict(lambda: [0, 0])
    if marks and len(marks) >= 2:
        for s_, e_, n_ in ev[marks[-2] + 1:marks[-1]]:
            fam[n_[:110]][0] += 1
            fam[n_[:110]][1] += e_ - s_
    kernels = [{"name": k, "calls": v[0], "total_ms": round(v[1] / 1e6, 3)} for k, v in sorted(fam.items(), key=lambda kv: -kv[1][1])]
    print(json.dumps({"replays": reps, "kernels_last_replay": kernels}))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 20, sys.argv[3] if len(sys.argv) > 3 else "sdxl", sys.argv[4] if len(sys.argv) > 4 else "fp16")
    else:
        analyse(sys.argv[2])
