"""Which stage of the bf16 forward first differs between two runs on the same weights and input?  (taps of model.capture_taps)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from transception_amd import MSTransception
from transception_amd.seeded_init import seeded_input, seeded_state_dict
dev = torch.device("cuda:0")
for mode in ("eval", "train"):
    m = MSTransception(num_classes=9); m.load_state_dict(seeded_state_dict(), strict=True); m = m.to(dev)
    m.train(mode == "train"); m.compute_dtype = torch.bfloat16; m.capture_taps = True
    x = torch.from_numpy(seeded_input(4)).to(dev)
    ref, ndiff = None, {}
    for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
        with torch.no_grad():
            lg = m(x)
        taps = dict(m.taps); taps["logits"] = lg.clone()
        if ref is None: ref = {k: v.clone() for k, v in taps.items()}; continue
        for k, v in taps.items():
            if not torch.equal(v, ref[k]):
                d = ndiff.setdefault(k, [0, 0.0]); d[0] += 1
                d[1] = max(d[1], ((v.float() - ref[k].float()).abs().max() / ref[k].float().abs().max()).item())
    print(mode, {k: (ndiff.get(k, [0, 0.0])[0], float("%.2e" % ndiff.get(k, [0, 0.0])[1])) for k in ref})
