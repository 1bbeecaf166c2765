"""Summarise a rocprofv3 kernel-trace CSV of bench.py: per-phase GPU busy vs wall, idle gaps, per-kernel totals of the
hand-written kernels.  Run on the GPU box right after rocprofv3 (the raw trace is too big to bring back)."""
import csv, sys, collections
path = sys.argv[1]
ev = []
with open(path) as f:
    for r in csv.DictReader(f):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
ev.sort()
t0 = ev[0][0]
picks = [i for i, e in enumerate(ev) if "k_pick_assemble" in e[2]]
print("kernels", len(ev), "pick_assemble launches", len(picks))
# phases = intervals between consecutive pick_assemble launches (skip the graph-replay block at the end: bursts)
phases = []
for a, b in zip(picks[:-1], picks[1:]):
    if b - a < 500:
        continue
    seg = ev[a:b]
    wall = (seg[-1][1] - seg[0][0]) / 1e6
    busy = sum(e - s for s, e, _ in seg) / 1e6
    lead = (ev[b][0] - seg[-1][1]) / 1e6
    phases.append((len(seg), wall, busy, lead))
A = [p for p in phases if p[0] > 1900]
B = [p for p in phases if p[0] <= 1900]
for name, ps in (("phase A (K=R+1)", A), ("phase B (K=1)", B)):
    if ps:
        n = len(ps)
        print(f"{name}: n={n} kernels/phase={ps[0][0]} mean wall {sum(p[1] for p in ps)/n:.2f} ms, mean busy {sum(p[2] for p in ps)/n:.2f} ms, "
              f"mean idle-before-next-phase {sum(p[3] for p in ps)/n:.3f} ms, max wall {max(p[1] for p in ps):.1f}")
# biggest gaps overall inside the image loops
if picks:
    lo, hi = picks[0], picks[-1]
    gaps = []
    for (s1, e1, n1), (s2, e2, n2) in zip(ev[lo:hi], ev[lo + 1:hi + 1]):
        if s2 - e1 > 100000:
            gaps.append(((s2 - e1) / 1e6, n1[:60], n2[:60]))
    gaps.sort(reverse=True)
    print("gaps > 0.1 ms:", len(gaps), "total", round(sum(g[0] for g in gaps), 1), "ms")
    for g in gaps[:12]:
        print(f"  {g[0]:8.2f} ms  {g[1]}  ->  {g[2]}")
tot = collections.defaultdict(lambda: [0, 0])
for s, e, n in ev:
    if "(anonymous namespace)::k_" in n:
        key = n.split("(anonymous namespace)::")[-1].split("(")[0].split("<")[0]
        tot[key][0] += 1
        tot[key][1] += e - s
for k, (c, d) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:28s} calls {c:6d} total {d/1e6:9.3f} ms mean {d/c/1e3:8.2f} us")
