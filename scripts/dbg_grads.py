import sys, math, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from golden_util import load, sample_idx
from transception_amd import MSTransception
from transception_amd.seeded_init import seeded_input, seeded_labels, seeded_state_dict
from transception_amd.train import SegLoss
from oracle.transception_oracle import TransCeptionOracle, ce_dice_loss, load_params
g = load("model_b2.npz")
sd = seeded_state_dict()
m = MSTransception(9); m.load_state_dict(sd); m.to("cuda:0").train()
m.use_fused_attention = ("--nofuse" not in sys.argv)
x = torch.from_numpy(seeded_input(2)); lab = torch.from_numpy(seeded_labels(2))
logits = m(x.cuda()); loss, ce, dice = SegLoss(9)(logits, lab.cuda()); loss.backward()
print("loss", loss.item(), g["loss"])
orc = TransCeptionOracle(load_params(sd, requires_grad=True), 9, training=True)
lo = orc(x); ol,_,_ = ce_dice_loss(lo, lab, 9); ol.backward()
print("logit err", (logits.detach().cpu()-lo.detach()).abs().max().item())
named = dict(m.named_parameters())
rows = []
seen=set()
for k, p in named.items():
    ref = orc.P[k].grad
    if ref is None or id(p) in seen: continue
    seen.add(id(p))
    got = p.grad
    if got is None: rows.append((float('inf'), k, 0, 0)); continue
    e = (got.cpu()-ref).abs().max().item(); s = ref.abs().max().item()
    rows.append((e/(s+1e-12), k, e, s))
rows.sort(reverse=True)
for r in rows[:40]: print("%.3e  %-80s err %.3e ref %.3e" % r)
print("n bad (rel>1e-2):", sum(1 for r in rows if r[0] > 1e-2), "of", len(rows))
