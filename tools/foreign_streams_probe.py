"""probe: the training epoch in a process whose host program created streams and captured a graph BEFORE this package's first launch
(the state of tests/test_gpu_scripts.py::test_two_stream_epoch_is_safe_after_foreign_streams_and_graphs): one stream / default / forced two streams"""
import json, os, sys, torch
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
streams = [torch.cuda.Stream() for _ in range(5)]
x = torch.zeros(1 << 20, device=dev)
for st in streams:
    with torch.cuda.stream(st):
        x.add_(1)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
y = torch.zeros(1 << 16, device=dev)
with torch.cuda.graph(g):
    y.mul_(2).add_(1)
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
job = bench.Job("vit_base", dev, 0, 1, 8, 0, "bf16")
out = {}
for part in ("0", None, "24", "0", None, "24"):
    rate, _, _ = bench.train_step_rate(job, None, 24, 8, "bf16", partition=part)
    out.setdefault({"0": "one", None: "default", "24": "forced_two"}[part], []).append(round(rate, 1))
print(json.dumps({"epoch_stream": os.environ.get("AG_EPOCH_STREAM", "1"), **out}))
