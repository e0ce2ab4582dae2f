import sys; sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
import numpy as np, torch
from golden_util import load
from transception_amd import MSTransception
from transception_amd.seeded_init import seeded_state_dict, seeded_input, seeded_labels
from transception_amd.train import FusedSGD, SegLoss, cosine_lr, train_step
g = load("train_trace.npz")
m = MSTransception(num_classes=9); m.load_state_dict(seeded_state_dict(), strict=True); m = m.cuda().train()
opt = FusedSGD(m, lr=0.05, momentum=0.9, weight_decay=1e-4); loss_fn = SegLoss(9)
for step in range(len(g["trace"])):
    x = torch.from_numpy(seeded_input(2, seed=7 + step)).cuda(); lab = torch.from_numpy(seeded_labels(2, seed=7 + step)).cuda()
    loss, ce, dice = train_step(m, loss_fn, opt, x, lab)
    opt.lr = cosine_lr(0.05, step + 1, 100)
    got = np.array([loss.item(), ce.item(), dice.item()]); want = g["trace"][step][:3]
    named = dict(m.named_parameters()); worst = (0, "")
    for key in [k.split("/", 1)[1] for k in g.files if k.startswith(f"step{step}/")]:
        a = named[key].detach().double().cpu(); w = g[f"step{step}/{key}"]
        r = abs(a.abs().sum().item() - w[1]) / w[1]
        r0 = abs(a.sum().item() - w[0])
        if r > worst[0]: worst = (r, key)
    print(step, "loss rel", np.abs(got - want) / want, "worst probe |w|-sum rel %.2e %s" % worst)
