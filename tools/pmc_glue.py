"""Launch every glue kernel a few times at the headline workload's shapes, for rocprofv3 --pmc passes
(HBM traffic per launch: FETCH_SIZE / WRITE_SIZE in separate passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
calls = bench.glue_launchers(torch.device("cuda", 0), bench.WORKLOADS["sdxl_1024x2048"], 50, torch.bfloat16)
for name, fn in calls.items():
    for _ in range(12):
        fn()
torch.cuda.synchronize()
print("done", list(calls))
