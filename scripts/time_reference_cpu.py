"""Oracle-vs-reference CPU timing, build container only (the reference's Python cannot travel to the GPU box).

    python scripts/time_reference_cpu.py [--batch 4] [--threads 8]  ->  profiles/reference_cpu_ratio.json

Times forward + loss + backward + SGD step of the imported reference (networks/MSTr.py::MSTransception with utils.DiceLoss, via
tests/golden/ref_shim.py) and of oracle/transception_oracle.py on the same seeded weights and batch, same thread count, median of 3
repetitions after a warm-up.  bench.py attaches the result to cpu_baseline as `reference_cpu_ratio`."""
import argparse
import json
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--threads", type=int, default=8)
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    from ref_shim import import_reference
    from oracle.transception_oracle import TransCeptionOracle, ce_dice_loss, load_params
    from transception_amd.seeded_init import seeded_input, seeded_labels, seeded_state_dict
    MSTransception, DiceLoss = import_reference()
    sd = seeded_state_dict()
    x = torch.from_numpy(seeded_input(a.batch))
    y = torch.from_numpy(seeded_labels(a.batch))

    ref = MSTransception(num_classes=9)
    ref.load_state_dict(sd, strict=True)
    ref.train()
    ce, dice = torch.nn.CrossEntropyLoss(), DiceLoss(9)
    ropt = torch.optim.SGD(ref.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-4)

    def ref_step():
        t0 = time.perf_counter()
        out = ref(x)
        loss = 0.4 * ce(out, y.long()) + 0.6 * dice(out, y, softmax=True)
        ropt.zero_grad()
        loss.backward()
        ropt.step()
        return time.perf_counter() - t0

    P = load_params(sd, requires_grad=True)
    leaves = list({id(v): v for v in P.values() if v.requires_grad}.values())
    oopt = torch.optim.SGD(leaves, lr=0.05, momentum=0.9, weight_decay=1e-4)
    orc = TransCeptionOracle(P, 9, training=True)

    def orc_step():
        t0 = time.perf_counter()
        loss, _, _ = ce_dice_loss(orc(x), y, 9)
        oopt.zero_grad()
        loss.backward()
        oopt.step()
        return time.perf_counter() - t0

    ref_step(); orc_step()
    tr = statistics.median(ref_step() for _ in range(3))
    to = statistics.median(orc_step() for _ in range(3))
    out = {"what": "fwd + 0.4 CE + 0.6 Dice + bwd + SGD step on CPU, fp32, same seeded weights and batch, median of 3 after warm-up, build container",
           "batch": a.batch, "threads": a.threads, "reference_s_per_step": tr, "oracle_s_per_step": to,
           "reference_images_per_sec": a.batch / tr, "oracle_images_per_sec": a.batch / to, "oracle_over_reference_speed": tr / to,
           "cpu": next((l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")), "?"),
           "script": "scripts/time_reference_cpu.py"}
    json.dump(out, open(os.path.join(ROOT, "profiles", "reference_cpu_ratio.json"), "w"), indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
