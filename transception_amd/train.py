"""Training-step harness: the reference's trainer.py semantics on the MI355X engine.

  loss = 0.4 * CrossEntropy + 0.6 * Dice(softmax=True)           trainer.py:141-143, utils.py:11-47
  SGD(momentum 0.9, weight_decay 1e-4), per-iteration cosine LR   trainer.py:125-127, 151-153
  batch-dim sharding: one process per GPU, gradients all-reduced over RCCL (the reference uses
  nn.DataParallel, trainer.py:110-111); the Dice sums are all-reduced so the loss equals the
  single-process global-batch loss.

Loss and optimiser are HIP kernels too (tc_seg_loss_*, tc_sgd_step); torch provides memory, streams and
torch.distributed only.
"""
from __future__ import annotations

import math
import os
from typing import Optional

import torch
import torch.distributed as dist

from ._lib import TC_BF16, TC_F16, TC_F32, lib


def _dt(t: torch.Tensor) -> int:
    return {torch.float32: TC_F32, torch.bfloat16: TC_BF16, torch.float16: TC_F16}[t.dtype]


COMM_AT_WORLD_1 = os.environ.get("TC_COMM_WORLD1", "0") == "1"


def comm_on(group=None) -> bool:
    """True when the step's collectives are to be issued: a process group of more than one rank -- or of exactly ONE rank when
    COMM_AT_WORLD_1 is set (TC_COMM_WORLD1=1; bench.py --gpus 1 --force-split and tests/test_rccl_gpu.py): a 1-rank RCCL all-reduce
    still goes through communicator creation, the collective stream, the async work handles and the no-collective-in-capture rule
    of the split step, so the structure the driver's 8-GPU run uses is executed on one GPU first."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or COMM_AT_WORLD_1


def seg_sums_allreduce(sums: torch.Tensor, n_pix: float, group=None):
    """C2 of SURVEY.md section 8(e): the [1 + 3*classes] vector (sum CE, then intersect/y_sum/z_sum per class) is summed over
    ranks so that every rank forms the loss of the GLOBAL batch (Dice is not linear in the batch, utils.py:24-32)."""
    world = 1
    if comm_on(group):
        world = dist.get_world_size(group)
        dist.all_reduce(sums, group=group)
    return sums, n_pix * world, world


def loss_from_sums(sums: torch.Tensor, n_pix: float, w_ce: float, w_dice: float):
    """0.4*CE + 0.6*Dice from the (global) sums: trainer.py:141-143, utils.py:34-47 (smooth 1e-5, mean over classes).
    Device sums: one tc_seg_loss_value launch.  Host sums (the gloo tests): the same expression in torch, double."""
    if sums.is_cuda:
        out3 = torch.empty(3, dtype=torch.float32, device=sums.device)
        lib().tc_seg_loss_value(sums.data_ptr(), (sums.numel() - 1) // 3, float(n_pix), float(w_ce), float(w_dice), out3.data_ptr(),
                                torch.cuda.current_stream(sums.device).cuda_stream)
        return out3[0], out3[1], out3[2]
    s = sums.double()
    ce = s[0] / n_pix
    inter, ysum, zsum = s[1::3], s[2::3], s[3::3]
    dice = (1.0 - (2.0 * inter + 1e-5) / (zsum + ysum + 1e-5)).mean()
    return w_ce * ce + w_dice * dice, ce, dice


class _SegLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, labels, ncls, w_ce, w_dice, group, loss_scale=1.0):
        L = lib()
        logits = logits.contiguous()
        B, C, H, W = logits.shape
        assert C == ncls and labels.shape == (B, H, W) and labels.dtype == torch.int64
        stream = torch.cuda.current_stream(logits.device).cuda_stream
        prob = torch.empty((B, C, H, W), dtype=torch.float32, device=logits.device)
        sums = torch.zeros(1 + 3 * ncls, dtype=torch.float32, device=logits.device)
        labels = labels.contiguous()
        L.tc_seg_loss_fwd(logits.data_ptr(), labels.data_ptr(), prob.data_ptr(), sums.data_ptr(), B, ncls, H * W, _dt(logits), stream)
        sums, n_pix, world = seg_sums_allreduce(sums, float(B * H * W), group)
        loss, ce, dice = loss_from_sums(sums, n_pix, w_ce, w_dice)
        ctx.save_for_backward(prob, labels, sums)
        ctx.meta = (ncls, w_ce, w_dice, n_pix, world, logits.dtype, float(loss_scale))
        return loss.float(), ce.float(), dice.float()

    @staticmethod
    def backward(ctx, gloss, gce, gdice):
        prob, labels, sums = ctx.saved_tensors
        ncls, w_ce, w_dice, n_pix, world, dtype, loss_scale = ctx.meta
        B, C, H, W = prob.shape
        L = lib()
        stream = torch.cuda.current_stream(prob.device).cuda_stream
        d = torch.empty((B, C, H, W), dtype=dtype, device=prob.device)
        # gradients of the *global* loss w.r.t. local logits; the later gradient all-reduce SUMS ranks, so the usual
        # DDP mean is folded in by the caller (gscale).  gloss is 1 for a plain loss.backward().
        gl = gloss.detach().float().contiguous()             # upstream d(loss) stays on the device: no host sync
        L.tc_seg_loss_bwd(prob.data_ptr(), labels.data_ptr(), sums.data_ptr(), d.data_ptr(), B, ncls, H * W, float(w_ce), float(w_dice),
                          float(n_pix), loss_scale, gl.data_ptr(), _dt(d), stream)
        return d, None, None, None, None, None, None


class SegLoss(torch.nn.Module):
    """0.4*CE + 0.6*Dice over the (global) batch; returns (loss, ce, dice)."""

    def __init__(self, n_classes: int = 9, w_ce: float = 0.4, w_dice: float = 0.6, group=None, loss_scale: float = 1.0):
        """loss_scale: the gradient that leaves this loss is multiplied by it (the reported loss is not) -- static loss scaling for
        float16 storage; FusedSGD.step divides it out again (train_step wires the two together)."""
        super().__init__()
        self.n_classes, self.w_ce, self.w_dice, self.group, self.loss_scale = n_classes, w_ce, w_dice, group, float(loss_scale)

    def forward(self, logits: torch.Tensor, labels: torch.Tensor):
        return _SegLossFn.apply(logits, labels.long(), self.n_classes, self.w_ce, self.w_dice, self.group, self.loss_scale)


def cosine_lr(base_lr: float, step: int, t_max: int) -> float:
    """CosineAnnealingLR(eta_min=0) after `step` scheduler.step() calls (trainer.py:126-127,151-153)."""
    return 0.5 * base_lr * (1.0 + math.cos(math.pi * step / t_max))


class FusedSGD:
    """torch.optim.SGD(momentum, weight_decay) semantics as one kernel over the model's flat arenas.

    Parameters that never receive a gradient (332 tensors in the reference) are left untouched, exactly as
    torch.optim.SGD skips `p.grad is None`: the update runs over the used segments only."""

    def __init__(self, model, lr: float = 0.05, momentum: float = 0.9, weight_decay: float = 1e-4, clip_norm: Optional[float] = None):
        self.model, self.lr, self.momentum, self.wd = model, lr, momentum, weight_decay
        self.clip_norm = clip_norm                       # nn.utils.clip_grad_norm_(max_norm, 2) before the update (trainer.py:147-148)
        self.buf: Optional[torch.Tensor] = None
        self.steps = 0
        self._segments = None
        self._segs_dev: Optional[torch.Tensor] = None
        self.lr_dev: Optional[torch.Tensor] = None       # device scalar read by the kernel (hipGraph-friendly schedule)
        self._sumsq: Optional[torch.Tensor] = None       # squared gradient norm (clip_norm)
        self.grad_scale = 1.0                            # gradients in the arena are this many times too small (1 / loss scale)
        self.guard_overflow: Optional[bool] = None       # skip the update when the gradient norm is not finite; None: on for float16 storage

        self.skipped_steps = 0                           # updates found skipped by last_step_skipped() (read back at the caller's log cadence)

    def last_step_skipped(self) -> bool:
        """True when the last update was skipped because the gradient norm was not finite (float16 overflow under a static loss scale:
        sgd_multi_kernel returns without touching the weights).  Reads one float back (a host sync): call it where the loop already
        reads its loss scalars; a run whose every step overflows would otherwise look like a plateau."""
        if self._sumsq is None:
            return False
        skipped = not math.isfinite(float(self._sumsq.item()))
        self.skipped_steps += int(skipped)
        return skipped

    def set_lr(self, lr: float):
        self.lr = lr
        if self.lr_dev is not None:
            self.lr_dev.fill_(lr)

    def _segs(self):
        if self._segments is None:
            views = sorted((off, math.prod(shape)) for off, shape in self.model._used_views.values())
            segs = []
            for off, n in views:
                n8 = (n + 7) // 8 * 8
                if segs and segs[-1][0] + segs[-1][1] == off:
                    segs[-1][1] += n8
                else:
                    segs.append([off, n8])
            self._segments = [(o, n) for o, n in segs]
        return self._segments

    def zero_grad(self):
        if self.model._gflat is not None:
            self.model._gflat.zero_()

    def step(self, grad_scale: Optional[float] = None):
        grad_scale = self.grad_scale if grad_scale is None else grad_scale
        M = self.model
        flat, g = M._flat, M._gflat
        if self.buf is None:
            self.buf = torch.zeros_like(flat)
        if self.lr_dev is None:
            self.lr_dev = torch.full((1,), float(self.lr), dtype=torch.float32, device=flat.device)
        L = lib()
        stream = torch.cuda.current_stream(flat.device).cuda_stream
        if self._segs_dev is None:
            segs = [(off, min(n, flat.numel() - off)) for off, n in self._segs()]
            self._segs_dev = torch.tensor([v for s in segs for v in s], dtype=torch.int64, device=flat.device)
            self._nseg, self._maxlen = len(segs), max(n for _, n in segs)
        sumsq = None
        guard = self.guard_overflow if self.guard_overflow is not None else M.compute_dtype == torch.float16
        if self.clip_norm or guard:
            # clip_grad_norm_(max_norm, 2): one reduction kernel for the squared norm (grad-less parameters hold zeros, so the arena's
            # norm is the model's), the coefficient min(1, max_norm / (norm + 1e-6)) is applied inside the update kernel
            if self._sumsq is None:
                self._sumsq = torch.zeros(1, dtype=torch.float32, device=flat.device)
            L.tc_fill_f32(self._sumsq.data_ptr(), 1, 0.0, stream)
            L.tc_grad_sumsq(g.data_ptr(), g.numel(), self._sumsq.data_ptr(), stream)
            sumsq = self._sumsq.data_ptr()
        # the 16-bit working copy of the weights is refreshed by the same kernel (the forward then skips its cast pass over the arena)
        lp = M._flat_lp if (M.compute_dtype != torch.float32 and M._flat_lp is not None and M._flat_lp.dtype == M.compute_dtype) else None
        L.tc_sgd_step_multi(flat.data_ptr(), g.data_ptr(), self.buf.data_ptr(), self._segs_dev.data_ptr(), self._nseg, self._maxlen,
                            float(self.lr), self.lr_dev.data_ptr(), float(self.momentum), float(self.wd), float(grad_scale),
                            int(self.steps == 0), sumsq, float((self.clip_norm or math.inf) / grad_scale),
                            lp.data_ptr() if lp is not None else None, M._tc_dtype() if lp is not None else 0, stream)
        M._lp_fresh = lp is not None
        self.steps += 1


def gradient_buckets(model, max_gap: int = 1 << 18):
    """Contiguous ranges of the flat gradient arena that hold live gradients, merged across gaps of < max_gap elements: the
    332 grad-less tensors (9.2 M of 47.3 M elements, SURVEY 8e: `conv1_1_s*`, decoder_3's unused blocks, ...) are not worth
    sending, but neither are dozens of tiny collectives.  Falls back to the whole arena before the first forward."""
    views = getattr(model, "_used_views", None)
    n = model._gflat.numel()
    if not views:
        return [(0, n)]
    rng = sorted((off, off + math.prod(shape)) for off, shape in views.values())
    out = [list(rng[0])]
    for a, b in rng[1:]:
        if a - out[-1][1] < max_gap:
            out[-1][1] = max(out[-1][1], b)
        else:
            out.append([a, b])
    return [(a, min(b, n)) for a, b in out]


def split_buckets(buckets, cut: int):
    """(early, late): the buckets below / at-or-above arena offset `cut`, a bucket that straddles it cut in two."""
    early, late = [], []
    for a, b in buckets:
        if b <= cut:
            early.append((a, b))
        elif a >= cut:
            late.append((a, b))
        else:
            early.append((a, cut))
            late.append((cut, b))
    return early, late


def clip_buckets(buckets, ranges):
    """The parts of `buckets` inside the arena `ranges` (both lists of half-open element ranges)."""
    out = []
    for lo, hi in ranges:
        for a, b in buckets:
            a2, b2 = max(a, lo), min(b, hi)
            if a2 < b2:
                out.append((a2, b2))
    return sorted(out)


SPAN_GAP_MAX = int(os.environ.get("TC_DDP_SPAN_GAP_MAX", str(2 << 20)))     # elements: a gap of dead (grad-less, zero) words up to this size is sent along rather than starting a new collective.  8 MB of zeros cost a one-link ring about what a collective's own latency does (2 (N-1) hops x 6 us at N = 8 = 84 us x 153 GB/s / 1.75); the 8 M elements of round 5 put 32 MB of zeros on the wire to save one


def comm_schedule(model):
    """ONE collective per stop of the split backward sweep (the nine bucket collectives of round 4 cost ~75 us each
    even on a 1-rank communicator and 2 (N-1) ring hops each on N ranks).  For every piece of model.gradient_pieces(), in order:
      * its live buckets are merged across gaps that hold only grad-less words (zeros in every rank's arena: sending them is harmless)
        up to SPAN_GAP_MAX elements -- never across words that belong to a LATER piece (they are not final yet and would be reduced twice);
      * the largest merged run is the piece's collective ("span"); smaller stragglers of at most COALESCE_MAX elements in all are
        deferred to the last piece;
      * the last piece (nothing left to hide it under) travels packed through a staging buffer ("pack"), stragglers included.
    Returns [(stop, [("span", a, b) | ("pack", [(a, b), ...]), ...]), ...]; every live word appears in exactly one entry."""
    live = gradient_buckets(model)
    pieces = model.gradient_pieces()

    def dead_only(x, y):
        return not any(a < y and x < b for a, b in live)

    sched, deferred = [], []
    for pi, (stop, ranges) in enumerate(pieces):
        bs = clip_buckets(live, ranges)
        runs = []
        for a, b in bs:
            if runs and (a == runs[-1][1] or (a - runs[-1][1] <= SPAN_GAP_MAX and dead_only(runs[-1][1], a))):
                runs[-1][1] = b
            else:
                runs.append([a, b])
        runs = [(a, b) for a, b in runs]
        if pi == len(pieces) - 1:
            allr = sorted(runs + deferred)
            if not allr:                                   #  nothing live in the last piece and nothing deferred: no collective, not an empty pack
                sched.append((stop, []))
            elif len(allr) == 1:
                sched.append((stop, [("span",) + allr[0]]))
            elif sum(b - a for a, b in allr) <= COALESCE_MAX:
                sched.append((stop, [("pack", allr)]))
            else:
                sched.append((stop, [("span", a, b) for a, b in allr]))
            continue
        if not runs:
            sched.append((stop, []))
            continue
        big = max(runs, key=lambda r: r[1] - r[0])
        small = [r for r in runs if r != big]
        if sum(b - a for a, b in small) <= COALESCE_MAX // 2:
            deferred += small
            sched.append((stop, [("span",) + big]))
        else:
            sched.append((stop, [("span", a, b) for a, b in runs]))
    return sched


def allreduce_scheduled(model, entries, group=None, async_op: bool = False):
    """Issues the collectives of one stop of comm_schedule(); returns the work handles (async_op) -- _PackedWork for a packed entry."""
    works = []
    if not comm_on(group):
        return works
    for e in entries:
        if e[0] == "span":
            w = dist.all_reduce(model._gflat[e[1]:e[2]], group=group, async_op=async_op)
            if async_op:
                works.append(w)
        else:
            w = _PackedWork(model._gflat, e[1], group, async_op)
            if async_op:
                works.append(w)
    return works


XGMI_LINK_GBS, XGMI_LINKS, RING_HOP_US = 153.0, 7, 6.0     # MI355X: 7 point-to-point links per GPU; per-hop latency of a ring step (assumed)


def comm_plan(model, n_ranks: int):
    """What a data-parallel step of `n_ranks` GPUs sends, without sending it: the pieces of the split backward sweep
    (model.gradient_pieces), the all-reduce buckets inside each (fp32 elements of the flat gradient arena, the grad-less tensors left
    out), and the time a ring all-reduce of each would take over xGMI -- a ring moves 2 (N-1)/N of the bytes through every GPU and is
    bound by ONE link per direction (153 GB/s) unless RCCL runs one ring per link (7), so both bounds are given, plus 2 (N-1) hops of
    latency per collective.  Lets the first multi-GPU run be read against an expectation."""
    out, tot, live = [], 0, 0
    f = 2.0 * (n_ranks - 1) / max(n_ranks, 1)
    lb = gradient_buckets(model)
    for stop, entries in comm_schedule(model):
        spans = [(e[1], e[2]) for e in entries if e[0] == "span"] + [r for e in entries if e[0] == "pack" for r in e[1]]
        nbytes = 4 * sum(b - a for a, b in spans)
        nlive = 4 * sum(max(0, min(b, y) - max(a, x)) for a, b in spans for x, y in lb)
        tot += nbytes
        live += nlive
        ncoll = len(entries)
        lat = 2 * (n_ranks - 1) * RING_HOP_US * 1e-3 * ncoll
        out.append({"sent_when_backward_reaches": stop or "end of backward (exposed)", "collectives": ncoll,
                    "how": [e[0] for e in entries],
                    "buckets_elements": [[int(a), int(b)] for a, b in spans], "megabytes": nbytes / 1e6, "live_megabytes": nlive / 1e6,
                    "ring_ms_one_link": f * nbytes / (XGMI_LINK_GBS * 1e9) * 1e3 + lat,
                    "ring_ms_seven_links": f * nbytes / (XGMI_LINK_GBS * XGMI_LINKS * 1e9) * 1e3 + lat})
    return {"ranks": n_ranks, "pieces": out, "gradient_megabytes_per_step": tot / 1e6, "live_gradient_megabytes_per_step": live / 1e6,
            "collectives_per_step": sum(p["collectives"] for p in out) + 1, "loss_sums_bytes": 28 * 4,
            "total_ring_ms_one_link": sum(p["ring_ms_one_link"] for p in out), "total_ring_ms_seven_links": sum(p["ring_ms_seven_links"] for p in out),
            "exposed_ring_ms_one_link": out[-1]["ring_ms_one_link"], "exposed_ring_ms_seven_links": out[-1]["ring_ms_seven_links"],
            "model": "ring all-reduce: 2 (N-1)/N x bytes per GPU over 153 GB/s per xGMI link (one ring) or 7 links (one ring per link), + 2 (N-1) "
                     "hops x 6 us per collective; every piece but the last travels under the rest of the backward sweep"}


class _PackedWork:
    """Several small, non-adjacent buckets sent as ONE collective: packed into a staging buffer, all-reduced, unpacked on wait()."""

    def __init__(self, grads, buckets, group, async_op):
        self.grads, self.buckets = grads, buckets
        self.flat = torch.cat([grads[a:b] for a, b in buckets])
        self.work = dist.all_reduce(self.flat, group=group, async_op=async_op)
        if not async_op:
            self._unpack()

    def _unpack(self):
        at = 0
        for a, b in self.buckets:
            self.grads[a:b].copy_(self.flat[at:at + b - a])
            at += b - a

    def wait(self):
        if self.work is not None:
            self.work.wait()
        self._unpack()


COALESCE_MAX = int(os.environ.get("TC_DDP_COALESCE_MAX", str(4 << 20)))     # elements: pieces up to this size travel as one collective


def allreduce_gradients(model, group=None, part: Optional[str] = None, async_op: bool = False, ranges=None, coalesce: bool = False):
    """C1: all-reduce(sum) of the live parts of the flat gradient arena over RCCL/xGMI (gloo in CPU tests): a handful of
    large buckets, every rank the same ones (the used set is a property of the architecture).  part="late" / "early" sends only
    the buckets at or above / below model.late_gradient_offset() (bridge + decoders / encoder); async_op returns the work
    handles instead of waiting, so the late part can travel under the encoder's backward.
    coalesce: a piece of several small non-adjacent buckets (the LAST piece of the split sweep: three ranges, 2.7 MB, nothing left to
    hide it under) goes as one collective through a packed staging buffer -- every collective costs 2 (N-1) ring hops of latency."""
    works = []
    if comm_on(group):
        buckets = gradient_buckets(model)
        if part is not None:
            early, late = split_buckets(buckets, model.late_gradient_offset())
            buckets = late if part == "late" else early
        if ranges is not None:                           # one piece of a split backward sweep (model.gradient_pieces)
            buckets = clip_buckets(buckets, ranges)
        if coalesce and len(buckets) > 1 and sum(b - a for a, b in buckets) <= COALESCE_MAX:
            w = _PackedWork(model._gflat, buckets, group, async_op)
            return [w] if async_op else []
        for a, b in buckets:
            w = dist.all_reduce(model._gflat[a:b], group=group, async_op=async_op)
            if async_op:
                works.append(w)
    return works


def train_step(model, loss_fn: SegLoss, opt: FusedSGD, images: torch.Tensor, labels: torch.Tensor, group=None):
    """One step with trainer.py semantics; returns (loss, ce, dice) tensors (no host sync)."""
    opt.zero_grad()
    opt.grad_scale = 1.0 / getattr(loss_fn, "loss_scale", 1.0)
    logits = model(images)
    loss, ce, dice = loss_fn(logits, labels)
    loss.backward()
    allreduce_gradients(model, group)
    opt.step()
    return loss, ce, dice


# Only the capturing thread is held to the capture rules: the RCCL watchdog thread polls events and the input pipeline's reader
# thread page-locks staging buffers while a step is being captured.
_CAPTURE_MODE = "thread_local"


def graph_kernel_nodes(g) -> Optional[int]:
    """Kernel nodes of a captured hipGraph (hipGraphGetNodes / hipGraphNodeGetType on the raw handle; the graph must have been
    created with keep_graph=True).  None when the runtime does not expose it."""
    import ctypes as C
    try:
        hip = C.CDLL("libamdhip64.so")
        raw = C.c_void_p(g.raw_cuda_graph())
        n = C.c_size_t(0)
        if hip.hipGraphGetNodes(raw, None, C.byref(n)) != 0:
            return None
        arr = (C.c_void_p * max(n.value, 1))()
        if hip.hipGraphGetNodes(raw, arr, C.byref(n)) != 0:
            return None
        k, t = 0, C.c_int(-1)
        for i in range(n.value):
            if hip.hipGraphNodeGetType(C.c_void_p(arr[i]), C.byref(t)) == 0 and t.value == 0:       # hipGraphNodeTypeKernel
                k += 1
        return k
    except Exception:
        return None


def _new_graph():
    try:
        return torch.cuda.CUDAGraph(keep_graph=True)             # keeps the hipGraph_t so that its nodes can be counted
    except TypeError:
        return torch.cuda.CUDAGraph()


def _instantiate(g):
    if hasattr(g, "instantiate"):
        try:
            g.instantiate()
        except Exception:
            pass


class GraphedStep:
    """A whole training step captured into hipGraphs and replayed: ~1800 kernel launches per step become graph launches,
    removing the host from the loop.  Inputs are copied into static buffers; the learning rate is a device scalar
    (FusedSGD.set_lr).

    world == 1: ONE graph = zero_grad + forward + loss + backward + SGD (through torch.autograd).
    world  > 1: no collective is ever captured.  Three graphs with the two RCCL all-reduces between them, driven through
    the engine directly (no autograd):  A = zero_grad + forward + per-pixel softmax / CE / Dice partial sums;
    all-reduce(28 floats);  B1 = loss gradient + backward of decoders and bridge;  async all-reduce of their gradient buckets
    (72 % of the live bytes) on the collective stream while B2 = the encoder's backward runs;  all-reduce of the encoder
    buckets;  C = fused SGD."""

    def __init__(self, model, loss_fn: SegLoss, opt: FusedSGD, images: torch.Tensor, labels: torch.Tensor, group=None, warmup: int = 3,
                 force_split: bool = False, pre=None):
        """pre(x, y): launches (on the current stream) that FILL the step's static inputs, captured at the head of the step's first graph
        -- the device input pipeline's augmentation / resize kernels reading one raw batch slot (data.DeviceLoader.iter_raw): a
        loader-fed loop is then graph launch after graph launch, with no eager launch in between."""
        self.model, self.loss_fn, self.opt, self.group = model, loss_fn, opt, group
        self._pre = pre
        self.x, self.y = images.clone(), labels.clone().long().contiguous()
        self._dtype = model.compute_dtype                # the captured launches hold pointers into this storage type's working copy
        opt.grad_scale = 1.0 / getattr(loss_fn, "loss_scale", 1.0)
        self.distributed = comm_on(group)
        self.split = self.distributed or force_split
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                if self.split:
                    self._fwd(); self._reduce_sums(); self._bwd(); self._bwd_rest(); allreduce_gradients(model, group); opt.step()
                else:
                    self._whole_step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.g_main = _new_graph()
        if not self.split:
            with torch.cuda.graph(self.g_main, capture_error_mode=_CAPTURE_MODE):
                self._whole_step()
            self.g_bwd = self.g_bwd_rest = self.g_opt = None
        else:
            with torch.cuda.graph(self.g_main, capture_error_mode=_CAPTURE_MODE):
                self._fwd()
            self._reduce_sums()
            self.g_bwd = _new_graph()
            with torch.cuda.graph(self.g_bwd, capture_error_mode=_CAPTURE_MODE):
                self._bwd()
            # the encoder's sweep in legs (stage 4, stage 3, the rest): each leg's gradients travel while the next leg runs
            self._pieces = model.gradient_pieces()
            self.g_bwd_legs = []
            for until, _ in self._pieces[1:]:
                g = _new_graph()
                with torch.cuda.graph(g, capture_error_mode=_CAPTURE_MODE):
                    self._bwd_leg(until)
                self.g_bwd_legs.append(g)
            self.g_bwd_rest = self.g_bwd_legs[-1]
            self._sched = comm_schedule(model)
            assert [st for st, _ in self._sched] == [st for st, _ in self._pieces]
            allreduce_gradients(model, group)
            self.g_opt = _new_graph()
            with torch.cuda.graph(self.g_opt, capture_error_mode=_CAPTURE_MODE):
                opt.step()
        for g in [self.g_main, self.g_bwd, self.g_opt] + (self.g_bwd_legs if self.split else []):
            if g is not None:
                _instantiate(g)

    def kernel_nodes(self) -> Optional[int]:
        """Kernel launches of one replayed step = kernel nodes of its captured graph(s)."""
        ks = [graph_kernel_nodes(g) for g in [self.g_main, self.g_bwd, self.g_opt] + (self.g_bwd_legs if self.split else []) if g is not None]
        return None if any(k is None for k in ks) else sum(ks)

    def _whole_step(self):
        """One process, no exchange: the same engine-driven pieces back to back in one graph (the loss kernels read the token-major
        logits of the classifier directly; train_step's nn.Module boundary would add the NCHW transpose and fp32 copies both ways)."""
        self._fwd()
        self._npix = self._npix_local
        self._bwd()
        self._bwd_rest()
        self.opt.step()

    # ---- the three pieces of the split step (engine driven directly; no torch.autograd in between)
    def _fwd(self):
        M, L = self.model, lib()
        M._ensure_flat(self.x.device)
        if self._pre is not None:
            self._pre(self.x, self.y)
        self.opt.zero_grad()
        logits, G, out_var = M._run(self.x, record=True, token_logits=True)     # [B*H*W, classes], storage type: no transpose, no fp32 copy
        self._G, self._out_var = G, out_var
        B, H, W = self.x.shape[0], self.x.shape[2], self.x.shape[3]
        C = logits.shape[1]
        assert logits.shape[0] == B * H * W and logits.stride(1) == 1
        stream = torch.cuda.current_stream(logits.device).cuda_stream
        self._logits, self._shape = logits, (B, C, H, W)           # (no probability map: the backward recomputes the softmax)
        self._sums = torch.zeros(1 + 3 * C, dtype=torch.float32, device=logits.device)
        L.tc_seg_loss_fwd_tok(logits.data_ptr(), logits.stride(0), self.y.data_ptr(), None, self._sums.data_ptr(), B, C, H * W,
                              _dt(logits), stream)
        self._npix_local = float(B * H * W)

    def _reduce_sums(self):
        self._sums, self._npix, _ = seg_sums_allreduce(self._sums, self._npix_local, self.group)

    def _bwd(self):
        M, L = self.model, lib()
        B, C, H, W = self._shape
        lf, lg = self.loss_fn, self._logits
        self.out = tuple(t.float() for t in loss_from_sums(self._sums, self._npix, lf.w_ce, lf.w_dice))
        stream = torch.cuda.current_stream(lg.device).cuda_stream
        # the gradient of the token-major logits, in the logits' own layout (their rows are padded to 16 bytes when the fused LayerNorm +
        # classifier produced them: Graph.ln_cls)
        d = torch.empty((B * H * W, lg.stride(0)), dtype=self.model.compute_dtype, device=lg.device)[:, :C]
        L.tc_seg_loss_bwd_tok(None, lg.data_ptr(), lg.stride(0), self.y.data_ptr(), self._sums.data_ptr(), d.data_ptr(), d.stride(0), B, C, H * W,
                              float(lf.w_ce), float(lf.w_dice), float(self._npix), float(lf.loss_scale), None, _dt(d), stream)
        M._backward(self._G, self._out_var, d, until="encoder_done")     # loss gradient, decoders, bridge

    def _bwd_leg(self, until: Optional[str]):
        self.model._backward_continue(self._G, until)                   # one leg of the encoder's sweep (None: to the end)
        if until is None:
            self._G = self._out_var = None

    def _bwd_rest(self):
        self._bwd_leg(None)                                             # the whole encoder in one go

    def __call__(self, images: Optional[torch.Tensor] = None, labels: Optional[torch.Tensor] = None, comm: bool = True):
        """comm=False replays the same graphs without the collectives between them (bench.py: the time the all-reduces add to a
        step is the difference; the ranks' parameters diverge, so it is a timing aid only)."""
        if self.model.compute_dtype != self._dtype:
            raise RuntimeError("the model's compute dtype changed after this step was captured: capture a new GraphedStep")
        if images is not None and images.data_ptr() != self.x.data_ptr():     # a loader may write straight into self.x / self.y
            self.x.copy_(images, non_blocking=True)
            self.y.copy_(labels, non_blocking=True)
        self.g_main.replay()
        if self.split:
            works = []
            if comm:
                self._reduce_sums()                  # C2: in place on the static 28-float buffer
            self.g_bwd.replay()
            if comm:                                 # C1, bridge + decoders: ONE collective (comm_schedule); the collective stream waits for g_bwd only
                works = allreduce_scheduled(self.model, self._sched[0][1], self.group, async_op=True)
            for g, (until, entries) in zip(self.g_bwd_legs, self._sched[1:]):
                g.replay()                           # stages 4 + 3, rest of the encoder: each leg's collective leaves under the next leg
                if comm:                             # (the last piece has nothing to hide under: everything left goes packed as ONE collective)
                    works += allreduce_scheduled(self.model, entries, self.group, async_op=True)
            for w in works:
                w.wait()                                     # the compute stream waits for the collectives, the host does not
            self.g_opt.replay()
        return self.out
