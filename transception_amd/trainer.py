"""`trainer_synapse` of the reference (trainer.py:49-237) on the MI355X path: the same schedule, loss, optimiser, checkpoint
cadence and evaluation calls, with the device input pipeline in place of `Synapse_dataset` + `DataLoader` and the captured
training step in place of the per-iteration Python loop body.

What is kept from the reference loop:
  * global batch = batch_size * n_gpu (trainer.py:86) -- here batch_size per rank, one process per GPU;
  * loss 0.4 CE + 0.6 Dice (trainer.py:141-143), SGD(momentum 0.9, weight_decay 1e-4) (:125);
  * CosineAnnealingLR(T_max = max_epochs * len(loader)) stepped every iteration (:126-127,151-153), or the polynomial decay
    base_lr (1 - iter/max_iter)^0.9 when use_scheduler is off (:154-157); optional clip_grad_norm_(5) (:147-148);
  * the learning-rate scaling quirk of the launcher (train_MSTransception.py:123-124) as `scaled_base_lr`;
  * checkpoints `<model_name>_epoch_<n>.pth` = `state_dict()` at the reference's epochs (:180-214), each followed by `inference`
    when evaluation volumes are given, and always after the last epoch.
Not kept: TensorBoard writers and the matplotlib/CSV result plots (:216-237).
"""
from __future__ import annotations

import logging
import os
from dataclasses import dataclass
from typing import Callable, List, Optional

import torch
import torch.distributed as dist

from .data import DeviceLoader, SynapseSlices
from .evaluate import inference
from .train import FusedSGD, GraphedStep, SegLoss, cosine_lr, train_step


@dataclass
class TrainConfig:
    root_path: str
    list_dir: str
    num_classes: int = 9
    max_epochs: int = 400
    batch_size: int = 24               # per GPU (train_MSTransception.py:35-36)
    base_lr: float = 0.05
    img_size: int = 224
    seed: int = 1234
    eval_interval: int = 20
    model_name: str = "transCeption"
    grad_clipping: bool = False
    use_scheduler: bool = True
    augment: bool = True
    graphed: bool = True               # replay the captured step (hipGraph); False launches every kernel from Python
    log_every: int = 1                 # iterations between log lines (each one reads three scalars back from the GPU)


def scaled_base_lr(base_lr: float, batch_size: int) -> float:
    """train_MSTransception.py:123-124: the learning rate is rescaled only when batch_size != 24 and batch_size % 5 == 0."""
    return base_lr * batch_size / 24 if batch_size != 24 and batch_size % 5 == 0 else base_lr


def checkpoint_epochs(max_epoch: int, eval_interval: int) -> List[int]:
    """Epochs after which the reference saves + evaluates (trainer.py:180-214)."""
    out = []
    for e in range(max_epoch):
        early = e >= int(max_epoch / 2) and e < int(max_epoch - 100) and (e + 1) % 20 == 0
        late = e >= int(max_epoch - 100) and (e + 1) % eval_interval == 0
        if early or late or e >= max_epoch - 1:
            out.append(e)
    return out


def trainer_synapse(cfg: TrainConfig, model, snapshot_path: str, volumes: Optional[Callable[[], list]] = None, group=None,
                    log: Optional[Callable[[str], None]] = None) -> dict:
    """Trains `model` (a transception_amd.MSTransception on the GPU) and returns the history.  `volumes` is a callable that
    yields the evaluation volumes as (image, label, case) triples (the `.npy.h5` reader needs h5py, see data.SynapseSlices)."""
    log = log or logging.info
    distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1
    rank = dist.get_rank(group) if distributed else 0
    world = dist.get_world_size(group) if distributed else 1
    dev = next(model.parameters()).device
    os.makedirs(snapshot_path, exist_ok=True)
    ds = SynapseSlices(cfg.root_path, cfg.list_dir, split="train")
    loader = DeviceLoader(ds, cfg.batch_size, img_size=cfg.img_size, device=dev, seed=cfg.seed, rank=rank, world=world,
                          augment=cfg.augment, epochs=cfg.max_epochs)
    per_epoch = len(loader) // cfg.max_epochs            # >= 1 for a non-empty set: data.rank_batches completes the last global batch
    max_iterations = cfg.max_epochs * per_epoch
    log("The length of train set is: {}".format(len(ds)))
    log("{} iterations per epoch. {} max iterations ".format(per_epoch, max_iterations))
    model.train()
    loss_fn = SegLoss(cfg.num_classes, group=group)
    opt = FusedSGD(model, lr=cfg.base_lr, momentum=0.9, weight_decay=1e-4, clip_norm=5.0 if cfg.grad_clipping else None)

    def lr_at(it: int) -> float:
        """Learning rate of step `it` (0-based).  Cosine: scheduler.step() follows every optimizer.step() (trainer.py:151-153).
        Polynomial: the reference computes lr_ from iter_num BEFORE incrementing it and assigns it for the NEXT step (:154-159),
        so steps 0 and 1 both run at base_lr and step k at base_lr (1 - (k-1)/max)^0.9."""
        if cfg.use_scheduler:
            return cosine_lr(cfg.base_lr, it, max_iterations)
        return cfg.base_lr * (1.0 - max(it - 1, 0) / max_iterations) ** 0.9

    if distributed:
        # identical replicas before the first step (the reference's DataParallel re-broadcasts rank 0's weights every forward,
        # trainer.py:110-111): parameters and BatchNorm statistics come from rank 0
        model._ensure_flat(dev)
        dist.broadcast(model.flat_parameters(), src=0, group=group)
        model.invalidate_working_copy()
        for buf in model.buffers():
            dist.broadcast(buf, src=0, group=group)

    saves = set(checkpoint_epochs(cfg.max_epochs, cfg.eval_interval))
    hist = {"loss": [], "lr": [], "dice": [], "hd95": [], "checkpoints": []}
    step = None
    iter_num = 0
    for x, y in loader:
        opt.set_lr(lr_at(iter_num))
        if iter_num == 0 or not cfg.graphed:
            # the first iteration runs eagerly: it creates the optimiser state and workspaces the captured step then refers to
            # (capturing at step 0 would bake "first step" into the SGD kernel's arguments), and it counts as iteration 1
            loss, ce, dice = train_step(model, loss_fn, opt, x, y, group)
        else:
            if step is None:
                step = GraphedStep(model, loss_fn, opt, x, y, group, warmup=0)
                loader.out = (step.x, step.y)                     # later batches land in the step's static inputs: no copy per step
            loss, ce, dice = step(x, y)
        iter_num += 1
        if iter_num % cfg.log_every == 0:
            lv, cv, dv = float(loss.detach()), float(ce.detach()), float(dice.detach())
            hist["loss"].append(lv)
            hist["lr"].append(lr_at(iter_num))
            log('iteration %d : lr: %f, loss : %f, loss_ce: %f, loss_dice: %f' % (iter_num, lr_at(iter_num), lv, cv, dv))
            if opt.last_step_skipped():                           # float16: a non-finite gradient norm skips the update (no weights were touched)
                log('iteration %d : update SKIPPED, gradient norm not finite (%d skipped so far; lower the loss scale if this persists)'
                    % (iter_num, opt.skipped_steps))
        if iter_num % per_epoch == 0:
            epoch_num = iter_num // per_epoch - 1
            if epoch_num in saves:
                if rank == 0:
                    path = os.path.join(snapshot_path, f'{cfg.model_name}_epoch_{epoch_num}.pth')
                    torch.save(model.state_dict(), path)
                    hist["checkpoints"].append(path)
                    log("save model to {}".format(path))
                if volumes is not None:
                    # Every rank evaluates its share of the volumes with the replica the checkpoint holds (rank 0's BatchNorm statistics
                    # are broadcast first; the weights are identical already) and the per-case sums are all-reduced: no rank sits in a
                    # collective while another spends minutes in host-side HD95 (a process-group timeout would abort the job).
                    vols = list(volumes())
                    if distributed:
                        for buf in model.buffers():
                            dist.broadcast(buf, src=0, group=group)
                        vols = vols[rank::world]
                    res = torch.zeros(3, dtype=torch.float64, device=dev)
                    if vols:
                        if rank == 0:
                            log(f"Running Inference after epoch {epoch_num}")
                        mean_dice, mean_hd95 = inference(model, vols, cfg.num_classes, cfg.img_size, log=log)
                        res[0], res[1], res[2] = mean_dice * len(vols), mean_hd95 * len(vols), len(vols)
                        model.train()
                    if distributed:
                        dist.all_reduce(res, group=group)
                    hist["dice"].append(float(res[0] / res[2].clamp(min=1)))
                    hist["hd95"].append(float(res[1] / res[2].clamp(min=1)))
    hist["iterations"] = iter_num
    return hist
