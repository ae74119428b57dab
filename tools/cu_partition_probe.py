"""Dev probe (round 4): the K-mask target forward and the explainer training step on two CU-partitioned streams at the same time
(ops.cu_partition_streams: c CUs of every XCD for the forward, 32 - c for the step) against both on the default stream back to back.
One JSON line per split.  AG_TRAIN_SIDE=0 recommended (the step's side streams are not partitioned here)."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from autognothi_amd import engine, ops, training as _tr  # noqa: E402
from autognothi_amd.utils import synth  # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
engine.set_precision("bf16")
_tr.MIXED_BF16 = True
wl = os.environ.get("CP_WORKLOAD", "vit_base")
tb = int(os.environ.get("CP_TB", "8"))
job = bench.Job(wl, dev, 0, 1, 48, 0, "bf16")
recipe, cfg = job.recipe, job.cfg
m_exp = recipe.t_explainer(cfg)
synth.load_synth_weights(m_exp, seed=1)
m_exp = m_exp.to(dev).train()
trainer = _tr.make_explainer_trainer(recipe, m_exp)
xs = torch.from_numpy(job.inputs(tb, 300)).to(dev)
c_ = cfg.num_labels
bits = ops.mask_shapley_new(ops.DeviceMT19937(dev, 5), tb * job.K, job.P, want_i64=False, want_bits=True)[1]
v_s, v_1 = torch.softmax(torch.randn(tb * job.K, c_, device=dev), -1), torch.softmax(torch.randn(tb, c_, device=dev), -1)
v_0 = torch.full((1, c_), 1.0 / c_, device=dev)
labels = torch.zeros(tb, dtype=torch.long, device=dev)
params = [q for q in m_exp.parameters() if q.requires_grad]


def estep():
    for q in params:
        q.grad = None
    trainer.loss_and_grads(xs, bits, v_0, v_s, v_1, job.K, labels=labels, train=True, seed=1)


def wall(fn, n):
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


for _ in range(3):
    job.step(); estep()
n_e = max(1, 1536 // (tb * job.K))          # explainer steps per 1 536-row target forward
base_f, base_e = wall(job.step, 5), wall(estep, 12)
seq = wall(lambda: (job.step(), [estep() for _ in range(n_e)]), 3)
print(json.dumps({"default_stream": {"forward_ms": round(base_f, 3), "explainer_step_ms": round(base_e, 3), "steps_per_forward": n_e,
                                     "back_to_back_ms_per_group": round(seq, 3), "ms_per_training_step": round(seq / n_e, 3)}}), flush=True)
for c in [int(x) for x in os.environ.get("CP_SPLITS", "28,26,24,22").split(",")]:
    if os.environ.get("CP_GRID_ONLY", "0") == "1":
        # no CU masks: two ordinary (non-blocking) streams; the forward's persistent GEMM is only told to launch 8 c workgroups
        # (one per CU: the other CUs stay free for whatever else is queued)
        from autognothi_amd import _lib as L
        sa, sb, na, nb = torch.cuda.Stream(dev), torch.cuda.Stream(dev), 8 * c, 256 - 8 * c
        L.check(L.lib().ag_set_stream_cus(sa.cuda_stream, na))
    else:
        sa, sb, na, nb = ops.cu_partition_streams(dev, c)

    def fwd_a():
        with torch.cuda.stream(sa):
            job.step()

    sb_free = os.environ.get("CP_STEP_UNMASKED", "0") == "1"      # the explainer step on an ordinary (unmasked) stream: any free CU

    def exp_b():
        if sb_free:
            estep()
        else:
            with torch.cuda.stream(sb):
                estep()

    def both():
        fwd_a()
        for _ in range(n_e):
            exp_b()

    for _ in range(2):
        both()
    torch.cuda.synchronize()
    f_a, e_b = wall(fwd_a, 5), wall(exp_b, 12)
    tog = wall(both, 4)
    print(json.dumps({"cus_forward": na, "cus_step": nb, "forward_alone_ms": round(f_a, 3), "explainer_step_alone_ms": round(e_b, 3),
                      "together_ms_per_group": round(tog, 3), "ms_per_training_step": round(tog / n_e, 3),
                      "vs_back_to_back": round(seq / tog, 3)}), flush=True)
