"""Diagnostic: what does MIOpen do for ONE channels-last bf16 convolution of the SDXL UNet (320->320 3x3 at 128x128,
batch 20)?  Prints the solver it picks in immediate mode and under `benchmark=True` (find), with MIOpen's own log lines
about find-db / perf-db keys, so the NHWC entries can be understood (round 1 measured the CK solver at 15.6 ms in NHWC
vs 0.99 ms in NCHW: untuned instance)."""
import os
import sys
import time

os.environ["MIOPEN_USER_DB_PATH"] = "/tmp/miopen_diag"
os.environ["MIOPEN_CUSTOM_CACHE_DIR"] = "/tmp/miopen_diag"
os.makedirs("/tmp/miopen_diag", exist_ok=True)
os.environ["MIOPEN_ENABLE_LOGGING"] = "1"
os.environ["MIOPEN_LOG_LEVEL"] = os.environ.get("DIAG_LOG_LEVEL", "5")
import torch

mode = sys.argv[1] if len(sys.argv) > 1 else "nhwc"
torch.backends.cudnn.benchmark = True
conv = torch.nn.Conv2d(320, 320, 3, padding=1).to("cuda", torch.bfloat16)
x = torch.randn(20, 320, 128, 128, device="cuda", dtype=torch.bfloat16)
if mode == "nhwc":
    conv = conv.to(memory_format=torch.channels_last)
    x = x.contiguous(memory_format=torch.channels_last)
with torch.no_grad():
    t0 = time.time()
    y = conv(x)
    torch.cuda.synchronize()
    print(f"DIAG first call {time.time() - t0:.1f}s", flush=True)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        y = conv(x)
    b.record()
    torch.cuda.synchronize()
    print(f"DIAG {mode} conv {a.elapsed_time(b) / 10:.3f} ms  out contiguous-cl={y.is_contiguous(memory_format=torch.channels_last)}", flush=True)
for f in os.listdir("/tmp/miopen_diag"):
    p = os.path.join("/tmp/miopen_diag", f)
    if f.endswith(".txt"):
        print("DIAG DB", f)
        print(open(p).read()[:3000])
