"""GPU probe: UNet forward time vs batch (plumbing measurement, not a test).
usage: probe_unet.py <sdxl|sd15> <batches comma list> [find]   ('find' => torch.backends.cudnn.benchmark=True, MIOpen find mode)"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import elasticdiffusion_official_amd  # sets the MIOpen cache location
from elasticdiffusion_official_amd import models as M

def bench(fn, n=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n

fam = sys.argv[1] if len(sys.argv) > 1 else "sdxl"
batches = [int(b) for b in (sys.argv[2] if len(sys.argv) > 2 else "20,6").split(",")]
if len(sys.argv) > 3 and sys.argv[3] == "find":
    torch.backends.cudnn.benchmark = True
if os.environ.get("ED_FUSED", "1") == "0":
    M.FUSED_KERNELS = False
if os.environ.get("ED_LN", "1") == "0":
    M.FUSED_LAYERNORM = False
if os.environ.get("ED_SCGEMM", "1") == "0":
    M.SHORTCUT_AS_GEMM = False
if os.environ.get("ED_CL", "0") == "1":
    M.CHANNELS_LAST = True
cfg = M.UNET_CONFIGS[fam]
dt = {"bf16": torch.bfloat16, "fp16": torch.float16}[os.environ.get("ED_DTYPE", "bf16")]
torch.manual_seed(0)
unet = M.UNet2DConditionModel(**cfg).to("cuda", dt).eval().requires_grad_(False)
if M.CHANNELS_LAST:
    unet = unet.to(memory_format=torch.channels_last)
S = cfg["sample_size"]
flops = {"sdxl": 6.761e12, "sd15": 0.803e12}[fam]
for B in batches:
    x = torch.randn(B, 4, S, S, device="cuda", dtype=dt)
    e = torch.randn(B, 77, cfg["cross_attention_dim"], device="cuda", dtype=dt)
    kw = None
    if cfg["pooled_projection_dim"]:
        kw = {"text_embeds": torch.randn(B, cfg["pooled_projection_dim"], device="cuda", dtype=dt), "time_ids": torch.zeros(B, 6, device="cuda")}
    t = torch.tensor(500, device="cuda")
    t0 = time.perf_counter()
    with torch.no_grad():
        dtm = bench(lambda: unet(x, t, encoder_hidden_states=e, added_cond_kwargs=kw))
    print(f"{fam} {os.environ.get('ED_DTYPE', 'bf16')} cl={M.CHANNELS_LAST} ln={M.FUSED_LAYERNORM} fused={M.FUSED_KERNELS} scgemm={M.SHORTCUT_AS_GEMM} benchmark={torch.backends.cudnn.benchmark} B={B:2d}: {dtm*1e3:8.1f} ms  {dtm*1e3/B:7.1f} ms/sample  {B*flops/dtm/1e12:7.1f} TFLOP/s  (wall incl. first call {time.perf_counter()-t0:.1f}s)", flush=True)
