import json, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from autognothi_amd import engine, _lib as L
dev = torch.device("cuda:0"); torch.cuda.set_device(dev); engine.set_precision("bf16")
job = bench.Job("vit_base", dev, 0, 1, 48, 0, "bf16")
def both(tag):
    a = bench.train_step_rate(job, None, 36, 8, "bf16", partition="24")[0]
    b = bench.train_step_rate(job, None, 36, 8, "bf16", partition="0")[0]
    print(tag, "two_streams", round(a, 1), "one_stream", round(b, 1), "reserved GB", round(torch.cuda.memory_reserved() / 2**30, 1), flush=True)
for _ in range(3): job.step()
torch.cuda.synchronize()
both("fresh")
which = os.environ.get("DBG", "graph48")
if which == "graph48":
    g = engine.GraphedStep(job.step); bench.timed(g, 5, 2, None, dev); del g
elif which == "eager_sweep":
    for b_s in (1, 4, 16, 48):
        job.set_batch(b_s); bench.timed(job.step, 10, 3, None, dev)
    job.set_batch(48)
elif which == "graph_sweep":
    for b_s in (1, 4, 16, 48):
        job.set_batch(b_s); g = engine.GraphedStep(job.step); bench.timed(g, 10, 2, None, dev); del g
    job.set_batch(48)
both("after " + which)
torch.cuda.empty_cache()
both("after empty_cache")
