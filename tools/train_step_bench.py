"""The config-5 training-step rates of bench.py's secondary block on their own (development loop): images/s, fraction of peak,
library launches per step for vanilla ViT-base, duo BERT-base, froyo ViT-base at --train-batch images x K = 32."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

if __name__ == "__main__":
    tb = int(os.environ.get("TB", "8"))
    steps = int(os.environ.get("STEPS", "12"))
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    for wl in (sys.argv[1:] or ["vit_base", "duo_bert_base", "froyo_vit_base"]):
        job = bench.Job(wl, dev, 0, 1, tb, 0, "bf16")
        rate, fl, frozen = bench.train_step_rate(job, None, steps, tb, "bf16")
        tf = rate / tb * fl / 1e12
        print(json.dumps({"workload": wl, "images_per_s": round(rate, 1), "ms_per_step": round(1e3 * tb / rate, 3), "gflop_per_step": round(fl / 1e9, 1),
                          "tflops": round(tf, 1), "frac": round(tf / bench.PEAK_BF16_TFLOPS, 4), "launches_per_step": bench.LAST_TRAIN_LAUNCHES[0],
                          "backbone_frozen": frozen, "side_stream": os.environ.get("AG_TRAIN_SIDE", "1")}), flush=True)
        del job
