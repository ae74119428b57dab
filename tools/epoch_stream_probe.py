"""probe: the explainer training epoch on the default (null) stream vs a non-blocking stream of its own, with and without RCCL initialised first"""
import json, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import bench
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
if os.environ.get("WITH_RCCL") == "1":
    import torch.distributed as tdist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
    tdist.init_process_group("nccl", rank=0, world_size=1, device_id=dev); tdist.barrier()
bench.engine.set_precision("bf16")
job = bench.Job("vit_base", dev, 0, 1, 8, 0, "bf16")
out = {}
for name, use in (("null", False), ("own", True), ("null", False), ("own", True)):
    st = torch.cuda.Stream(dev) if use else torch.cuda.current_stream()
    with torch.cuda.stream(st):
        for part in (None, "0"):
            r, _, _ = bench.train_step_rate(job, None, 36, 8, "bf16", partition=part)
            out.setdefault(f"{name}_{'two' if part is None else 'one'}", []).append(round(r, 1))
    torch.cuda.synchronize()
print(json.dumps({"rccl": os.environ.get("WITH_RCCL", "0"), **out}))
