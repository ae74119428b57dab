"""Dev probe (round 4): the masked forward (bench Job.step, ViT-base B = 48 K = 32) with the persistent GEMM's grid limited to n CUs
(ag_set_stream_cus on the current stream; no CU mask: the other CUs idle), alternating, two passes."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from autognothi_amd import _lib as L, engine  # noqa: E402
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
engine.set_precision("bf16")
job = bench.Job(os.environ.get("CP_WORKLOAD", "vit_base"), dev, 0, 1, int(os.environ.get("CP_B", "48")), 0, "bf16")
for _ in range(3): job.step()
torch.cuda.synchronize()
cur = torch.cuda.current_stream().cuda_stream
res = {}
for rnd in range(3):
    for n in (256, 248, 240, 232, 224, 208, 192):
        L.check(L.lib().ag_set_stream_cus(cur, n if n < 256 else 0))
        job.step(); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(6): job.step()
        torch.cuda.synchronize()
        res.setdefault(n, []).append(round((time.perf_counter() - t) / 6 * 1e3, 3))
L.check(L.lib().ag_set_stream_cus(cur, 0))
print(json.dumps(res))
