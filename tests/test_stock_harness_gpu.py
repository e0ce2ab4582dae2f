"""The boundary claim "drops into trainer.py unchanged" (SURVEY.md 8(b)), drilled with stock PyTorch pieces only.

The module is driven exactly the way /root/reference/trainer.py:110-153 drives the reference: `torch.nn.CrossEntropyLoss`, a Dice
loss written against torch ops (restated here from the definition in utils.py:11-47: per class, 1 - (2 sum(p t) + 1e-5) /
(sum(p p) + sum(t t) + 1e-5) over the whole batch tensor, mean over classes), `torch.optim.SGD(model.parameters(), ...)`,
`CosineAnnealingLR`, `optimizer.zero_grad()`, `loss.backward()`, optional `nn.utils.clip_grad_norm_`, and an `nn.DataParallel`
wrap.  None of the repo's own SegLoss / FusedSGD / GraphedStep is used.  Two steps must reproduce the reference's recorded trace
(tests/golden/train_trace.npz, produced by the reference itself with the same recipe).
"""
import math

import numpy as np
import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu

from golden_util import load  # noqa: E402
from transception_amd.seeded_init import seeded_input, seeded_labels, seeded_state_dict  # noqa: E402

DEV = "cuda:0"


class TorchDice(nn.Module):
    def __init__(self, n_classes: int):
        super().__init__()
        self.n = n_classes

    def forward(self, logits, target):
        p = torch.softmax(logits, dim=1)
        total = 0.0
        for c in range(self.n):
            t = (target == c).float()
            pc = p[:, c]
            total = total + (1.0 - (2.0 * (pc * t).sum() + 1e-5) / ((pc * pc).sum() + (t * t).sum() + 1e-5))
        return total / self.n


def _model():
    from transception_amd import MSTransception
    m = MSTransception(num_classes=9)
    m.load_state_dict(seeded_state_dict(), strict=True)
    return m.cuda()                                          # test.py:178 style


def test_reference_training_recipe_with_stock_torch_pieces():
    g = load("train_trace.npz")
    model = _model()
    model.train()
    ce_loss, dice_loss = nn.CrossEntropyLoss(), TorchDice(9)
    optimizer = torch.optim.SGD(model.parameters(), lr=0.05, momentum=0.9, weight_decay=0.0001)
    scheduler = torch.optim.lr_scheduler.CosineAnnealingLR(optimizer, T_max=100)
    for step in range(len(g["trace"])):
        image = torch.from_numpy(seeded_input(2, seed=7 + step)).cuda()
        label = torch.from_numpy(seeded_labels(2, seed=7 + step)).cuda()
        outputs = model(image)
        loss_ce = ce_loss(outputs, label[:].long())
        loss_dice = dice_loss(outputs, label)
        loss = 0.4 * loss_ce + 0.6 * loss_dice
        optimizer.zero_grad()
        loss.backward()
        gn = math.sqrt(sum(float((p.grad.double() ** 2).sum()) for p in model.parameters() if p.grad is not None))
        optimizer.step()
        scheduler.step()
        lr = scheduler.get_last_lr()[-1]
        np.testing.assert_allclose([loss.item(), loss_ce.item(), loss_dice.item(), lr], g["trace"][step][:4], rtol=max(5e-5, 1e-5 * 2 ** step), atol=2e-5)
        assert abs(gn - g["trace"][step][4]) <= 5e-4 * g["trace"][step][4]
        named = dict(model.named_parameters())
        for key in [k.split("/", 1)[1] for k in g.files if k.startswith(f"step{step}/")]:
            a = named[key].detach().double().cpu()
            np.testing.assert_allclose([a.sum().item(), a.abs().sum().item()], g[f"step{step}/{key}"], rtol=max(2e-5, 4e-6 * 3 ** step), atol=2e-4)
    assert sum(1 for p in model.parameters() if p.grad is None) == 332       # stock SGD skipped the reference's grad-less set


def test_clip_grad_norm_and_polynomial_lr_branch():
    """trainer.py:147-148,154-157: clip_grad_norm_(model.parameters(), 5) and the manual param_group lr assignment."""
    model = _model().train()
    optimizer = torch.optim.SGD(model.parameters(), lr=0.05, momentum=0.9, weight_decay=0.0001)
    image = torch.from_numpy(seeded_input(2)).cuda()
    label = torch.from_numpy(seeded_labels(2)).cuda()
    outputs = model(image)
    loss = 0.4 * nn.CrossEntropyLoss()(outputs, label) + 0.6 * TorchDice(9)(outputs, label)
    optimizer.zero_grad()
    loss.backward()
    before = [p.grad.clone() for p in model.parameters() if p.grad is not None]
    total = nn.utils.clip_grad_norm_(model.parameters(), max_norm=0.05, norm_type=2)
    ref = math.sqrt(sum(float((b.double() ** 2).sum()) for b in before))
    assert abs(float(total) - ref) <= 1e-4 * ref and ref > 0.05
    after = [p.grad for p in model.parameters() if p.grad is not None]
    scale = 0.05 / (ref + 1e-6)
    for a, b in zip(after[:20], before[:20]):
        assert torch.allclose(a, b * scale, rtol=1e-4, atol=1e-9)
    for group in optimizer.param_groups:
        group["lr"] = 0.05 * (1.0 - 1 / 100) ** 0.9
    optimizer.step()
    assert torch.isfinite(model.flat_parameters()).all()


def test_data_parallel_wrap_on_one_device():
    """trainer.py:110-111 wraps the model in nn.DataParallel; on one device that must forward identically, train, and save
    `module.`-prefixed keys (trainer.py:184 saves model.state_dict() of the wrapper)."""
    model = _model()
    wrapped = nn.DataParallel(model, device_ids=[0])
    wrapped.eval()
    x = torch.from_numpy(seeded_input(2)).cuda()
    with torch.no_grad():
        a, b = model(x), wrapped(x)
    assert torch.equal(a, b)
    keys = list(wrapped.state_dict().keys())
    assert len(keys) == 2200 and all(k.startswith("module.") for k in keys)
    assert [k[len("module."):] for k in keys] == list(seeded_state_dict().keys())
    wrapped.train()
    label = torch.from_numpy(seeded_labels(2)).cuda()
    opt = torch.optim.SGD(wrapped.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-4)
    loss = nn.CrossEntropyLoss()(wrapped(x), label)
    opt.zero_grad()
    loss.backward()
    opt.step()
    assert torch.isfinite(loss) and torch.isfinite(model.flat_parameters()).all()
