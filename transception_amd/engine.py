"""Host-side execution engine: a reverse-mode tape over raw HIP kernel launches.

Instead of ~16k ATen autograd nodes (what the reference's eager forward builds), one
training step here is a flat list of C-ABI kernel launches recorded on a tape; backward replays the tape in
reverse, each entry launching the hand-written gradient kernels.  Gradients of activations live in buffers
with the same layout as the activations (so column / row slices of a buffer receive their gradients in place,
residual fan-in is fused as `accumulate` flags or by aliasing), and parameter gradients are accumulated by
the kernels straight into the model's flat fp32 gradient arena.  Everything is asynchronous on the current
HIP stream, with static shapes and no host synchronisation, so a whole step can be captured in a hipGraph.

torch is used for device memory (torch.empty) and streams only.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Callable, Dict, List, Optional, Tuple

import torch

from ._lib import (ATTN_DKV_SPLITS, EW_COPY, EW_DEINTERLEAVE, EW_PATCHIFY, TcEffAtt, TcEwSeg, TcFfnBwd, TcFfnFused, ACT_COORD, ACT_GELU, ACT_HSWISH, ACT_NONE, ACT_SCALE, ACT_SIGMOID, FFN_EP, FFN_LN_A, FFN_LN_B, TC_BF16, TC_F16, TC_F32, TcDwSeg, TcFfnSeg,
                   TcGemm, lib)

_DT = {torch.float32: TC_F32, torch.bfloat16: TC_BF16, torch.float16: TC_F16}

# bench.py sets this to a dict {kernel name: [(start_event, stop_event, algorithmic_flops), ...]} to time individual
# launches with HIP events on the launching stream (the events are recorded on torch's current stream, which is the
# stream every kernel here is launched on).
PROFILE: Optional[dict] = None
BN_STATS_IN_GEMM = os.environ.get("TC_BN_STATS_IN_GEMM", "1") != "0"     # A/B switch: BatchNorm statistics in the producing GEMM's epilogue


SEG_MARKS = os.environ.get("TC_SEG_MARKS", "")      # json path: section markers on (Graph.segment)
SEG_LABELS: List[str] = []


def _timed(name: str, flops: float, fn):
    if PROFILE is None:
        fn()
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    PROFILE.setdefault(name, []).append((e0, e1, flops))


def _gemm_work(descs):
    """Algorithmic (flops, bytes) of a list of TcGemm problems: 2MNK per product; every operand read once and the result written
    once (read + written when it accumulates), in the storage type (fp32 for weight-gradient outputs)."""
    fl = by = 0.0
    for g in descs:
        nb = g.nb1 * g.nb2
        es = 4 if g.dtype == TC_F32 else 2
        ec = 4 if g.c_f32 else es
        fl += 2.0 * g.M * g.N * g.K * nb
        by += nb * ((g.M * g.K + g.K * g.N) * es + g.M * g.N * ec * (2 if g.accumulate else 1) + (g.M * g.N * es if g.R else 0))
    return fl, by


def _timed_gemm(kind: str, descs, fn):
    """bench.py: HIP events around a GEMM-family launch (tc_gemm / tc_gemm_pair / tc_gemm_multi) with its algorithmic flops and bytes."""
    if PROFILE is None:
        fn()
        return
    _timed("gemm:" + kind, _gemm_work(descs), fn)


class P:
    """Engine-side handle of one parameter: compute-dtype data + fp32 gradient view (or None when frozen).
    gs = distance (elements) to the same parameter of the next group when an op runs several weight sets at once
    (the three MB paths: same shapes, different weights, laid out at a constant stride in the flat arenas)."""
    __slots__ = ("data", "grad", "gs")

    def __init__(self, data: torch.Tensor, grad: Optional[torch.Tensor], gs: int = 0):
        self.data = data
        self.grad = grad
        self.gs = gs


class Var:
    """A 2-D activation view [rows, cols] (last dim contiguous, row stride ld) plus its lazily created gradient.

    Children made by colslice()/rowslice()/reshape() share the root's storage *and* the root's gradient buffer;
    every view knows the rectangle of the root it covers so that gradient writes from overlapping views accumulate."""
    __slots__ = ("data", "root", "path", "kids", "grad_t", "whole_written", "written", "requires_grad", "region", "reshaped", "covered", "bn_part")

    def __init__(self, data: torch.Tensor, root: "Var" = None, path=None, requires_grad: bool = True, region=None,
                 reshaped: bool = False):
        assert data.dim() == 2 and (data.stride(1) == 1 or data.shape[1] == 1), (data.shape, data.stride())
        self.data = data
        self.root = root if root is not None else self
        self.path = path or ()            # sequence of ('c', a, b) / ('r', a, b) / ('v', rows, cols) from the root
        self.kids: Dict[tuple, Var] = {}
        self.grad_t: Optional[torch.Tensor] = None     # root only
        self.whole_written = False                     # root only
        self.written: List[tuple] = []                 # root only: rectangles (r0, r1, c0, c1) that received a gradient
        self.requires_grad = requires_grad
        self.region = region if region is not None else (0, data.shape[0], 0, data.shape[1])   # in root coordinates
        self.covered = False                           # root only: slice writers are known to cover the whole gradient before any read
        self.reshaped = reshaped

    rows = property(lambda s: s.data.shape[0])
    cols = property(lambda s: s.data.shape[1])
    ld = property(lambda s: s.data.stride(0))

    def _kid(self, key, data, region, reshaped=False):
        k = self.kids.get(key)
        if k is None:
            k = Var(data, self.root, self.path + (key,), self.root.requires_grad, region, reshaped)
            self.kids[key] = k
        return k

    def colslice(self, a: int, b: int) -> "Var":
        assert not self.reshaped, "slice before reshaping"
        r0, r1, c0, _ = self.region
        return self._kid(("c", a, b), self.data[:, a:b], (r0, r1, c0 + a, c0 + b))

    def rowslice(self, a: int, b: int) -> "Var":
        assert not self.reshaped, "slice before reshaping"
        r0, _, c0, c1 = self.region
        return self._kid(("r", a, b), self.data[a:b], (r0 + a, r0 + b, c0, c1))

    def reshape(self, rows: int, cols: int) -> "Var":
        assert self.data.is_contiguous() and self.region[2] == 0 and self.region[3] == self.root.data.shape[1]
        return self._kid(("v", rows, cols), self.data.view(rows, cols), self.region, True)

    @property
    def is_whole(self) -> bool:
        rt = self.root.data
        return self.region == (0, rt.shape[0], 0, rt.shape[1])

    def apply_path(self, t: torch.Tensor) -> torch.Tensor:
        for k in self.path:
            if k[0] == "c":
                t = t[:, k[1]:k[2]]
            elif k[0] == "r":
                t = t[k[1]:k[2]]
            else:
                t = t.view(k[1], k[2])
        return t


def _overlap(a, b) -> bool:
    return a[0] < b[1] and b[0] < a[1] and a[2] < b[3] and b[2] < a[3]


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


_SIDE_STREAMS: Dict[str, list] = {}


def _side_streams(device, n: int):
    key = str(device)
    pool = _SIDE_STREAMS.setdefault(key, [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream(device))
    return pool[:n]


_WORKSPACES: Dict[Tuple[str, int, str], torch.Tensor] = {}
WORKSPACE_BYTES = 16384 + 1024 * 16384             # arrival counters + 1024 fp32 64x64 partial tiles


def _workspace(device, stream_ptr: int, tag: str = "") -> torch.Tensor:
    """Split-K fix-up workspace of the kernels launched on `stream_ptr` (include/transception_hip.h, TcGemm.ws): launches on one
    stream never overlap, so one buffer per stream is enough; zeroed once, every launch leaves its counters zero.  `tag`: a
    separate buffer for launches that cut it into per-problem slices (their counter areas must never have held partial tiles)."""
    # Single-stream mode (the default): warm-up stream, capture stream and replay stream take turns and never run launches
    # concurrently, so they share ONE workspace -- a workspace per stream pointer was created (and zero-filled: 134 MB for the
    # "many" one) INSIDE the capture of a step, whose stream is new, and the fill was replayed with every step.
    key = (str(device), int(stream_ptr or 0) if Graph.use_streams else 0, tag)
    # CONSTRAINT: this holds only while the engine launches of a device are serialized -- one training / evaluation loop at a
    # time.  Two loops driving the engine concurrently on one device (an evaluation thread beside the training thread, two models on two
    # streams) would share split-K counters and partial tiles; run them with TC_STREAMS=1 (a workspace per stream) or one after the
    # other.  (Not asserted here: autograd runs the backward of a step on its own worker thread, so thread identity says nothing.)
    ws = _WORKSPACES.get(key)
    if ws is None:
        ws = _WORKSPACES[key] = torch.zeros(WORKSPACE_BYTES * (8 if tag == "many" else 1), dtype=torch.uint8, device=device)
    return ws


_SPLITK_CAP = int(os.environ.get("TC_SPLITK_CAP", "128"))
_SPLITK_BLOCKS = int(os.environ.get("TC_SPLITK_BLOCKS", "512"))
_SPLITK_KMIN = int(os.environ.get("TC_SPLITK_KMIN", "384"))              # shortest K range of a split of a weight-gradient product: 256 / 320 / 384 / 448 / 512 -> 12.00 / 11.94 / 11.95 / 12.11 / 12.20 ms (fewer fp32 atomics against longer serial K loops)
_THR128 = int(os.environ.get("TC_GEMM_THR128", "100000"))     # keep in step with gemm.hip (gemm_plan)
_GEMM_PAIR = os.environ.get("TC_GEMM_PAIR", "1") != "0"
_N_WSTREAMS = int(os.environ.get("TC_WGRAD_STREAMS", "4"))
_NO_PENDING = bool(os.environ.get("TC_DEBUG_NO_PENDING"))   # timing what-if only (racy)
_SKIP_WGRAD = bool(os.environ.get("TC_DEBUG_SKIP_WGRAD"))
_FFN_STORE_ACT = os.environ.get("TC_FFN_STORE_ACT", "1") != "0"   # MixFFN: keep GELU(LN(d)) from the forward pass (0: recompute it in dW2's loader)
_FFN_LN_GEMM_MAXC = int(os.environ.get("TC_FFN_LN_GEMM_MAXC", "64"))  # widest fc2 output for which LayerNorm + GELU run in its A loader
_FFN_TILED = os.environ.get("TC_FFN_TILED", "1") != "0"        # MixFFN forward as one spatially tiled kernel where the library supports the width (csrc/mixffn.hip)
_DW_DEFER = os.environ.get("TC_DW_DEFER", "1") != "0"                  # depthwise weight-gradient sums folded once per backward leg (tc_dw_fold)
_LN_DEFER = os.environ.get("TC_LN_DEFER", "1") != "0"                  # LayerNorm dgamma / dbeta partials folded once per backward leg (tc_layernorm_fold) instead of at every launch's tail
_EFFATT_FUSED = os.environ.get("TC_EFFATT_FUSED", "1") != "0"        # EfficientAttention blocks through csrc/effatt.hip where the library supports the width
# The LayerNorm ahead of a MixFFN site (the block's norm2) inside the tiled kernels (both directions tiled: C = 64).  OFF by default:
# measured a wash -- 13.12 vs 13.08 ms per step with it on: the tiled kernels are VALU-bound, so the +17 us (backward) / +2 us (forward)
# the in-kernel LayerNorm costs per site cancel the two memory-bound launches it removes (negative results).
_FFN_PRE_LN = os.environ.get("TC_FFN_PRE_LN", "0") != "0"
_LIN_LN_FUSED = os.environ.get("TC_LIN_LN_FUSED", "1") != "0"       # square Linear + residual + LayerNorm (proj / reprojection + skip + norm2) as one forward launch (csrc/linln.hip)
_RIPM_FUSED = os.environ.get("TC_RIPM_FUSED", "1") != "0"            # a DWConv2d_BN step of the RIPM stages per launch, BatchNorm applied by the consumer (csrc/ripm.hip)
_LN_CLS_FUSED = os.environ.get("TC_LN_CLS_FUSED", "1") != "0"        # FinalPatchExpand_X4's rearrange + LayerNorm and the classifier as one forward / one backward launch (csrc/lncls.hip, round 6)
_DW_LN_FUSED = os.environ.get("TC_DW_LN_FUSED", "1") != "0"          # cpe (dw3x3 + skip) + norm1 of an MHCABlock as one forward launch
_MHCA_ATT_BWD_FUSED = os.environ.get("TC_MHCA_ATT_BWD_FUSED", "1") != "0"  # ... and the backward of crpe + attention core as one launch
_MHCA_ATT_FUSED = os.environ.get("TC_MHCA_ATT_FUSED", "1") != "0"  # qkv + crpe + factorised attention of an MHCABlock as one forward launch (csrc/factoratt.hip)
_DW_BWD_ONE = os.environ.get("TC_DW_BWD_ONE", "1") != "0"          # input + weight gradient of a stride-1 depthwise conv in one launch
_FFN_TILED_BWD = os.environ.get("TC_FFN_TILED_BWD", "1") != "0"   # MixFFN backward on the chip (csrc/mixffn_bwd.hip) where the library supports the width
_FFN_TILE_BWD = (0, 0)                                         # forced pixel tile of the tiled backward's second launch (tests)
_FFN_TILE = (0, 0)                                             # forced pixel tile of the tiled MixFFN kernels (tests); (0, 0): the library's choice
_POISON = bool(os.environ.get("TC_DEBUG_POISON"))         # fill every fresh buffer with NaN: finds reads of memory no kernel wrote


def _empty(shape, dtype, device) -> torch.Tensor:
    if _POISON:
        return torch.full(tuple(shape), float("nan"), dtype=dtype, device=device)
    return torch.empty(tuple(shape), dtype=dtype, device=device)


# Gradient buffers that need a zero start (partially written roots, scatter targets) come out of ONE arena per backward sweep that
# is zeroed by a single launch at the first request: ~30 separate 5 us fill kernels per step otherwise.  The arena's size is what the
# previous sweep on this (device, dtype) asked for -- shapes are static from step to step -- and anything beyond it falls back to
# torch.zeros.
_ZERO_NEED: Dict[Tuple[str, torch.dtype], int] = {}


class _ZeroArena:
    def __init__(self, device, dtype, shared: bool = True):
        """shared=False (the TC_STREAMS=1 mode): every request gets its own torch.zeros storage -- Graph._pending keys the events of
        side-stream weight-gradient kernels by gradient STORAGE, and one arena storage under several gradient buffers would let one
        buffer's event overwrite another's; the single zero fill would also run on whichever branch stream asked first."""
        self.key = (str(device), dtype)
        self.shared = shared
        self.device, self.dtype = device, dtype
        self.buf: Optional[torch.Tensor] = None
        self.used = 0
        self.asked = 0

    def zeros_like(self, t: torch.Tensor) -> torch.Tensor:
        n = (t.numel() + 127) // 128 * 128                       # keep 256-byte alignment of every sub-buffer
        if not self.shared:
            return torch.zeros_like(t)
        self.asked += n
        if self.buf is None:
            need = _ZERO_NEED.get(self.key, 0)
            if need >= n:
                self.buf = torch.zeros(need, dtype=self.dtype, device=self.device)
        if self.buf is not None and self.used + n <= self.buf.numel() and t.is_contiguous():
            out = self.buf[self.used:self.used + t.numel()].view(t.shape)
            self.used += n
            return out
        return torch.zeros_like(t)

    def close(self):
        _ZERO_NEED[self.key] = max(self.asked, 0)


class _Branch:
    def __init__(self, G, stream):
        self.G, self.stream, self.ctx, self.prev = G, stream, None, None

    def __enter__(self):
        G = self.G
        self.prev = (G.cur, G.stream)
        G.cur, G.stream = self.stream, self.stream.cuda_stream
        self.ctx = torch.cuda.stream(self.stream)
        self.ctx.__enter__()
        return self

    def __exit__(self, *a):
        self.ctx.__exit__(*a)
        self.G.cur, self.G.stream = self.prev


class _Parallel:
    """Fork/join of independent sub-graphs over HIP streams (branch 0 stays on the main stream).  In a captured hipGraph the
    branches become parallel node chains; the tape replays them in reverse with the fork and join swapped."""

    def __init__(self, G, n: int, shared=()):
        self.G, self.n = G, n
        self.shared = [v.root for v in shared]      # buffers several branches write gradient slices of
        self.sides = _side_streams(G.dev, n - 1) if (n > 1 and G.use_streams and torch.device(G.dev).type == "cuda") else []

    def __enter__(self):
        G = self.G
        for s in self.sides:
            s.wait_stream(G.cur)
        if G.record and self.sides:
            G.tape.append(("join", G.cur, self.sides))          # backward: main waits for the branches
        return self

    def branch(self, i: int):
        G = self.G
        return _Branch(G, self.sides[i - 1] if (i > 0 and self.sides) else G.cur)

    def __exit__(self, *a):
        G = self.G
        for s in self.sides:
            G.cur.wait_stream(s)
        if G.record and self.sides:
            G.tape.append(("fork", G.cur, self.sides, self.shared))   # backward: the branches wait for main


class _Grouped:
    def __init__(self, G, n, pgs):
        self.G, self.n, self.pgs, self.prev = G, n, pgs, None

    def __enter__(self):
        self.prev = (self.G.ngroups, self.G.pgs)
        self.G.ngroups, self.G.pgs = self.n, self.pgs
        return self

    def __exit__(self, *a):
        self.G.ngroups, self.G.pgs = self.prev


class Graph:
    # Side streams are OFF by default.  Measured on MI355X (ROCm 7.2): a captured step executes at the SUM of its kernel times
    # whether its branches are captured on 1 stream or 8 (kernels from different HW queues overlap only at their tails, and a
    # graph wider than the 4 HW queues shares queues), while the single-stream graph runs every kernel un-contended and
    # back-to-back -- 671 vs 634-648 img/s -- and avoids the hipStreamEndCapture crashes some multi-stream shapes trigger.
    use_streams = os.environ.get("TC_STREAMS", "0") == "1"
    overlap_wgrad = os.environ.get("TC_NO_WGRAD_OVERLAP", "0") != "1"
    ngroups = 1          # ops with parameters run `ngroups` stacked row blocks, block g with the weights at +g*P.gs
    pgs = 0

    def grouped(self, n: int, pgs: int) -> _Grouped:
        return _Grouped(self, n, pgs)

    def parallel(self, n: int, shared=()) -> _Parallel:
        return _Parallel(self, n, shared)

    def __init__(self, dtype: torch.dtype, device, training: bool, record: bool):
        self.L = lib()
        self.dtype = dtype
        self.dt = _DT[dtype]
        self.dev = device
        self.training = training
        self.record = record
        self.tape: List[Callable[[], None]] = []
        self._wnext = 0
        self._wstream = None          # side stream for weight-gradient kernels (nothing downstream in backward needs them)
        self._pending: Dict[int, torch.cuda.Event] = {}     # gradient storage -> last weight-gradient kernel still reading it
        self._keep: list = []
        self._ln_pending: list = []                          # LayerNorm backward launches whose dgamma / dbeta partials wait for the leg's fold
        self._dw_pending: list = []                          # depthwise backward launches whose weight-gradient sums wait for the leg's fold
        self.cur = torch.cuda.current_stream(device) if torch.device(device).type == "cuda" else None
        self.stream = self.cur.cuda_stream if self.cur is not None else 0
        self.n_launch = 0
        self._seg_prev = None
        if SEG_MARKS:
            del SEG_LABELS[:]
        self._zeros = _ZeroArena(device, dtype, shared=not self.use_streams)

    # ------------------------------------------------------------------ memory
    def new(self, rows: int, cols: int, requires_grad: bool = True, covered: bool = False) -> Var:
        """covered=True: every part of this buffer's gradient is written by some slice writer before anything reads it (the
        q | k | v column split, the per-scale row split of the bridge), so its gradient buffer needs no zero fill."""
        v = Var(_empty((rows, cols), self.dtype, self.dev), requires_grad=requires_grad)
        v.covered = covered
        return v

    def f32(self, *shape) -> torch.Tensor:
        return _empty(shape, torch.float32, self.dev)

    # ------------------------------------------------------------------ gradient bookkeeping
    def grad_of(self, v: Var) -> Optional[torch.Tensor]:
        """Gradient view of v for READING (None if nothing was ever written to any part of it)."""
        r = v.root
        if r.grad_t is None:
            return None
        if not (r.whole_written or any(_overlap(v.region, w) for w in r.written)):
            return None
        return v.apply_path(r.grad_t)

    def wgrad(self, v: Var, ext_rows: int = 0) -> Tuple[torch.Tensor, int]:
        """Gradient view of v for WRITING and whether the kernel must accumulate into it.  ext_rows: the kernel also
        writes that many root rows beyond the view (batch-strided launches whose view is batch 0)."""
        r = v.root
        whole = v.is_whole
        reg = (v.region[0], v.region[1] + ext_rows, v.region[2], v.region[3])
        if r.grad_t is None:
            if whole:
                r.grad_t = _empty(r.data.shape, r.data.dtype, r.data.device)
                assert r.grad_t.stride() == r.data.stride(), "gradient layout must mirror the data layout"
            else:
                assert r.data.is_contiguous()
                r.grad_t = _empty(r.data.shape, r.data.dtype, r.data.device) if r.covered else self._zeros.zeros_like(r.data)
        acc = r.whole_written or any(_overlap(reg, w) for w in r.written)
        if whole:
            r.whole_written = True
        else:
            r.written.append(reg)
        if self._pending and not _NO_PENDING:
            ev = self._pending.pop(r.grad_t.untyped_storage().data_ptr(), None)
            if ev is not None:
                self.cur.wait_event(ev)                  # a weight-gradient kernel on the side stream still reads this buffer
        return v.apply_path(r.grad_t), int(acc)

    def _region_grad(self, root: Var, off: int, nelem: int) -> torch.Tensor:
        """Mark `nelem` elements starting `off` elements into the (contiguous) root as gradient-carrying (the writer
        accumulates); returns the root gradient buffer, zero-initialised on first use."""
        assert root.data.is_contiguous()
        cols = root.data.shape[1]
        if root.grad_t is None:
            root.grad_t = self._zeros.zeros_like(root.data)
        root.written.append((off // cols, (off + nelem + cols - 1) // cols, 0, cols))
        return root.grad_t

    def pass_grad(self, v: Var, src: torch.Tensor, defer: bool = False):
        """v.grad (+)= src without a copy when v has no gradient yet (identity / residual branches).  defer: a copy that is needed
        comes back as a tc_ew_multi segment (None otherwise) for the caller to merge with its neighbours' into one launch."""
        if not v.requires_grad:
            return None
        r = v.root
        if v.is_whole and r.grad_t is None and src.stride() == r.data.stride() and src.shape == r.data.shape:
            r.grad_t = src                                   # alias: later contributions accumulate in place
            r.whole_written = True
            return None
        g, acc = self.wgrad(v)
        if defer:
            return TcEwSeg(EW_COPY, acc, src.data_ptr(), g.data_ptr(), 0, 0, src.stride(0), g.stride(0), 1, g.shape[0], g.shape[1], 0, 0, 0)
        if acc:
            self.L.tc_add(_ptr(g), g.stride(0), _ptr(src), src.stride(0), _ptr(g), g.stride(0), g.shape[0], g.shape[1],
                          self.dt, self.stream)
        else:
            self.L.tc_copy3d(_ptr(src), 0, src.stride(0), _ptr(g), 0, g.stride(0), 1, g.shape[0], g.shape[1], 0, self.dt,
                             self.stream)

    def _write_or_add(self, v: Var, fn: Callable[[torch.Tensor], None]):
        """For kernels without an accumulate mode: write directly, or via a temporary + add."""
        g, acc = self.wgrad(v)
        if not acc:
            fn(g)
        else:
            tmp = torch.empty((g.shape[0], g.shape[1]), dtype=g.dtype, device=g.device)
            fn(tmp)
            self.L.tc_add(_ptr(g), g.stride(0), _ptr(tmp), tmp.stride(0), _ptr(g), g.stride(0), g.shape[0], g.shape[1],
                          self.dt, self.stream)

    def _weight_grad(self, fn: Callable[[], None], reads: Optional[torch.Tensor] = None):
        """Run a weight-gradient kernel on the side stream: dW / dw / db are only consumed by the optimizer, so they overlap
        with the activation-gradient chain.  `reads` = the upstream gradient buffer the kernel reads: a later in-place
        accumulation into that buffer (aliased residual gradients) must wait for this kernel (see wgrad)."""
        if _SKIP_WGRAD:                                  # timing what-if only (results are wrong): critical-path analysis
            return
        if not (self.overlap_wgrad and self.use_streams and self.cur is not None):
            fn()
            return
        if self._wstream is None:
            self._wstream = _side_streams(self.dev, 3 + _N_WSTREAMS)[3:]
        ws = self._wstream[self._wnext % len(self._wstream)]   # independent kernels: round-robin so they also overlap each other
        self._wnext += 1
        ws.wait_stream(self.cur)
        with _Branch(self, ws):
            fn()
        if reads is not None:
            ev = torch.cuda.Event()
            ev.record(ws)
            self._pending[reads.untyped_storage().data_ptr()] = ev
            self._keep.append(reads)

    def mark(self, name: str):
        """A named cut in the tape: backward(until=name) stops there, so a caller can start exchanging the gradients that are
        complete (everything recorded after the mark) while the rest of the backward sweep runs."""
        if self.record:
            self.tape.append(("mark", name))

    def segment(self, name: str):
        """Profiling aid (TC_SEG_MARKS=<json path>, off by default): an empty marker launch at the start of section `name` of the forward
        sweep and -- through the tape -- at the start of the backward of the section that ends here, so that a kernel trace of a replayed
        step can be cut into sections (scripts/seg_timeline.py).  The marker's id is its grid size; the id -> label table is written to
        the path when the backward sweep ends."""
        if not SEG_MARKS:
            return
        prev = self._seg_prev
        self._seg_prev = name
        SEG_LABELS.append("F:" + name)
        self.L.tc_seg_marker(len(SEG_LABELS) - 1, self.stream)

        def bwd():
            SEG_LABELS.append("B:" + (prev or "head"))
            self.L.tc_seg_marker(len(SEG_LABELS) - 1, self.stream)
        self._rec(bwd)

    def backward(self, until: Optional[str] = None) -> bool:
        """Runs the tape in reverse.  With `until`, stops at that mark and returns False (call again to finish)."""
        main = (self.cur, self.stream)
        while self.tape:
            entry = self.tape.pop()
            if entry[0] == "mark":
                if entry[1] == until:
                    self.cur, self.stream = main
                    self._flush_param_folds()
                    if self._wstream is not None:
                        for ws in self._wstream:
                            self.cur.wait_stream(ws)
                    return False
            elif entry[0] == "join":                       # forward fork point: everything the branches produced flows back to main
                for s in entry[2]:
                    entry[1].wait_stream(s)
            elif entry[0] == "fork":                     # forward join point: branch backward starts after main's upstream work
                for r in entry[3]:                       # gradient buffers shared by the branches are created (zeroed) on main
                    if r.grad_t is None:                 # first, so no branch's lazy zero-fill can race another branch's write
                        r.grad_t = self._zeros.zeros_like(r.data)
                for s in entry[2]:
                    s.wait_stream(entry[1])
            else:
                fn, st = entry
                if st is None or st is self.cur:
                    fn()
                else:
                    with _Branch(self, st):
                        fn()
        self.cur, self.stream = main
        self._flush_param_folds()
        if self._wstream is not None:
            for ws in self._wstream:
                self.cur.wait_stream(ws)                 # all weight gradients are in the arena before anything follows
        self._pending.clear()
        self._keep.clear()
        self.tape = []
        self._zeros.close()
        if SEG_MARKS:
            import json
            with open(SEG_MARKS, "w") as f:
                json.dump(SEG_LABELS, f)
        return True

    def _flush_dw_folds(self):
        """One launch adds the parked walker sums of every depthwise weight gradient since the last flush (tc_dw_fold)."""
        from ._lib import TcDwFold
        for c0 in range(0, len(self._dw_pending), 48):                # (tc_dw_fold takes up to 48 sites: its argument block stays below 4 KiB)
            chunk = self._dw_pending[c0:c0 + 48]
            arr = (TcDwFold * len(chunk))(*[site for site, _ in chunk])
            self.n_launch += 1
            self.L.tc_dw_fold(arr, len(chunk), self.stream)
        self._dw_pending = []

    def _flush_param_folds(self):
        """End of a backward leg: the parked parameter-gradient partials of the leg's launches are added up -- depthwise / mid-backward walkers
        (tc_dw_fold), then the LayerNorms' dgamma / dbeta (tc_layernorm_fold: one launch for every LayerNorm backward since the last flush)."""
        self._flush_dw_folds()
        if not self._ln_pending:
            return
        from ._lib import TcLnFold
        arr = (TcLnFold * len(self._ln_pending))()
        for i, (part, dg, db, gs, nblk, Cc, Gn) in enumerate(self._ln_pending):
            arr[i].part, arr[i].dgamma, arr[i].dbeta, arr[i].pstride = _ptr(part), _ptr(dg), _ptr(db), gs
            arr[i].nblk, arr[i].C, arr[i].groups = nblk, Cc, Gn
        self.n_launch += 1
        self.L.tc_layernorm_fold(arr, len(self._ln_pending), self.stream)
        self._ln_pending = []                    # (the buffers may go back to the allocator now: whatever reuses them runs after this launch)

    def _rec(self, fn):
        if self.record:
            self.tape.append((fn, self.cur))

    # ------------------------------------------------------------------ GEMM plumbing
    def _gemm(self, A, lda, B, ldb, Cm, ldc, M, N, K, tA, tB, bias=None, R=None, ldr=0, alpha=1.0, acc=0, act=ACT_NONE,
              splitk=1, nb1=1, nb2=1, sA=(0, 0), sB=(0, 0), sC=(0, 0), sR=(0, 0), c_f32=0, atomic=0, rowsum=None, sbias=0, srow=0,
              bn_part=None, bn_shift=None):
        g = self._gemm_desc(A, lda, B, ldb, Cm, ldc, M, N, K, tA, tB, bias, R, ldr, alpha, acc, act, splitk, nb1, nb2, sA, sB, sC, sR,
                            c_f32, atomic, rowsum, sbias, srow)
        if bn_part is not None:
            g.bn_part, g.bn_shift = bn_part, bn_shift
        self.n_launch += 1
        _timed_gemm("single", [g], lambda: self.L.tc_gemm(C.byref(g), self.stream))

    def _gemm_desc(self, A, lda, B, ldb, Cm, ldc, M, N, K, tA, tB, bias=None, R=None, ldr=0, alpha=1.0, acc=0, act=ACT_NONE,
                   splitk=1, nb1=1, nb2=1, sA=(0, 0), sB=(0, 0), sC=(0, 0), sR=(0, 0), c_f32=0, atomic=0, rowsum=None, sbias=0,
                   srow=0, use_ws=True, bgap=(0, 0)) -> TcGemm:
        ws = _workspace(self.dev, self.stream) if use_ws else None
        return TcGemm(A, B, Cm, bias, R, M, N, K, lda, ldb, ldc, ldr, tA, tB, nb1, nb2, sA[0], sA[1], sB[0], sB[1], sC[0], sC[1],
                      sR[0], sR[1], alpha, acc, act, splitk, self.dt, c_f32, atomic, rowsum, sbias, srow, bgap[0], bgap[1],
                      ws.data_ptr() if ws is not None else None, ws.numel() if ws is not None else 0)

    @staticmethod
    def _small_tiles(m_out: int, n_out: int, nb: int) -> bool:
        """tc_gemm's tile choice (gemm.hip, gemm_plan): 64x64 tiles unless >= 100000 tiles of 128x128 exist."""
        return ((m_out + 127) // 128) * ((n_out + 127) // 128) * nb < _THR128 or m_out < 96 or n_out < 96

    @staticmethod
    def _splitk(m_out: int, n_out: int, k_red: int) -> int:
        tiles = ((m_out + 63) // 64) * ((n_out + 63) // 64)
        cap = _SPLITK_CAP if k_red < 262144 else 1024      # > 128: tc_gemm folds groups of 16 splits through the workspace
        return max(1, min(max(_SPLITK_BLOCKS, cap) // max(tiles, 1), k_red // _SPLITK_KMIN, cap))

    # ------------------------------------------------------------------ ops
    def linear(self, x: Var, W: P, b: Optional[P] = None, out: Optional[Var] = None, residual: Optional[Var] = None,
               act: int = ACT_NONE, wcols: Optional[Tuple[int, int]] = None, accumulate: bool = False,
               batch: Optional[Tuple[int, int, int, int]] = None, post_scale: Optional[float] = None,
               bn_shift: Optional[torch.Tensor] = None, launch: bool = True) -> Var:
        """out = act(x @ W[:, wcols]^T + b + residual)  (or out += ... when accumulate).  W is [N, K] (nn.Linear).

        bn_shift (fp32 [N], the running mean of the BatchNorm this output feeds in training mode): on 16-bit storage the GEMM's
        epilogue leaves the statistics pass of that BatchNorm behind (TcGemm.bn_part) and batchnorm(out, ...) skips its own.

        post_scale c: out = c * (x W^T + b + residual), rounded once from the fp32 accumulator (TC_ACT_SCALE) -- the bridge's q
        projection hands the attention kernels q * scale * log2(e).  The consumer must hand back d loss / d(x W^T + b) as
        grad_of(out), i.e. the scale belongs to the consumer's backward, not to this layer's (attention_seg(q_prescaled=True) does).

        batch=(nb, sx, so, sr): x / out / residual are the views of batch 0 ([M, K] / [M, N]); batch i lives
        sx / so / sr ELEMENTS further on in the same buffers (used to re-lay-out rows between buffers)."""
        Wt = W.data if wcols is None else W.data[:, wcols[0]:wcols[1]]
        N, K = Wt.shape
        assert x.cols == K, (x.cols, K)
        Gn = self.ngroups
        grouped = Gn > 1
        M = x.rows // Gn
        if out is None:
            assert batch is None
            out = self.new(x.rows, N)
        if grouped:
            # stacked groups: x / out / residual hold Gn row blocks of M rows; weights of group g at +g*W.gs.
            # `out` may also be a [M, Gn*N] column range of a wider buffer (group g -> columns g*N..): the IFF concat.
            assert batch is None and W.gs > 0 and (b is None or b.gs == W.gs)
            side_by_side = (out.rows == M and out.cols == Gn * N)
            assert side_by_side or (out.rows == Gn * M and out.cols == N)
            nb, sx, so, sr = Gn, M * x.ld, (N if side_by_side else M * out.ld), (M * residual.ld if residual is not None else 0)
            sw = W.gs
        else:
            side_by_side = False
            assert out.rows == M and out.cols == N
            nb, sx, so, sr = batch if batch is not None else (1, 0, 0, 0)
            sw = 0
        if post_scale is not None:
            assert act == ACT_NONE and not accumulate
        bn_part = None
        if (bn_shift is not None and BN_STATS_IN_GEMM and self.training and self.dtype != torch.float32 and nb == 1 and not accumulate
                and act == ACT_NONE and post_scale is None and N % 8 == 0 and out.ld % 8 == 0 and out.data.data_ptr() % 16 == 0):
            tiles = (M + 63) // 64
            bn_part = self.f32(N * (1 + 2 * max(tiles, 128)))        # (also the scratch of the BatchNorm's backward sums)
            out.bn_part = (bn_part, tiles)
        if launch:          # (False: a fused kernel of the caller computes `out`; only the backward closure is recorded)
            self._gemm(_ptr(x.data), x.ld, _ptr(Wt), Wt.stride(0), _ptr(out.data), out.ld, M, N, K, 0, 1,
                       bias=_ptr(b.data) if b is not None else None, R=_ptr(residual.data) if residual is not None else None,
                       ldr=residual.ld if residual is not None else 0, acc=int(accumulate), act=act if post_scale is None else ACT_SCALE,
                       alpha=1.0 if post_scale is None else post_scale, nb1=nb, sA=(sx, 0), sB=(sw, 0), sC=(so, 0), sR=(sr, 0), sbias=sw,
                       bn_part=_ptr(bn_part) if bn_part is not None else None, bn_shift=_ptr(bn_shift) if bn_part is not None else None)

        def bwd():
            dy = self.grad_of(out)
            if dy is None:
                return
            dz = dy
            if act == ACT_SIGMOID:
                assert dy.is_contiguous() and out.data.is_contiguous() and (nb == 1 or (grouped and not side_by_side))
                dz = torch.empty_like(dy)
                self.L.tc_sigmoid_bwd(_ptr(dy), _ptr(out.data), _ptr(dz), dy.numel(), self.dt, self.stream)
            want_db = b is not None and b.grad is not None
            paired = (_GEMM_PAIR and x.requires_grad and W.grad is not None and self.dtype != torch.float32
                      and self._small_tiles(M, K, nb) and self._small_tiles(N, K, nb))
            if paired:
                # dX and dW (+db) of this layer share one launch: two short grids fill the CUs together (tc_gemm_pair)
                gx, acc = self.wgrad(x, (nb - 1) * sx // x.root.cols if (nb > 1 and not grouped) else 0)
                gW = W.grad if wcols is None else W.grad[:, wcols[0]:wcols[1]]
                ga = self._gemm_desc(_ptr(dz), dz.stride(0), _ptr(Wt), Wt.stride(0), _ptr(gx), gx.stride(0), M, K, N, 0, 0, acc=acc,
                                     nb1=nb, sA=(so, 0), sB=(sw, 0), sC=(sx, 0))
                gb = self._gemm_desc(_ptr(dz), dz.stride(0), _ptr(x.data), x.ld, _ptr(gW), gW.stride(0), N, K, M, 1, 0, acc=1,
                                     splitk=self._splitk(N, K, M), c_f32=1, nb1=nb, sA=(so, 0), sB=(sx, 0), sC=(sw, 0),
                                     atomic=int(nb > 1 and not grouped), rowsum=_ptr(b.grad) if want_db else None, srow=sw,
                                     use_ws=False)
                if gb.splitk > 128:                      # the one workspace of this stream goes to the problem that folds partials
                    gb.ws, gb.ws_bytes, ga.ws, ga.ws_bytes = ga.ws, ga.ws_bytes, None, 0
                self.n_launch += 1
                _timed_gemm("pair", [ga, gb], lambda: self.L.tc_gemm_pair(C.byref(ga), C.byref(gb), self.stream))
            elif x.requires_grad:
                gx, acc = self.wgrad(x, (nb - 1) * sx // x.root.cols if (nb > 1 and not grouped) else 0)
                self._gemm(_ptr(dz), dz.stride(0), _ptr(Wt), Wt.stride(0), _ptr(gx), gx.stride(0), M, K, N, 0, 0, acc=acc,
                           nb1=nb, sA=(so, 0), sB=(sw, 0), sC=(sx, 0))
            if paired:
                pass
            elif W.grad is not None:
                gW = W.grad if wcols is None else W.grad[:, wcols[0]:wcols[1]]
                self._weight_grad(lambda: self._gemm(
                    _ptr(dz), dz.stride(0), _ptr(x.data), x.ld, _ptr(gW), gW.stride(0), N, K, M, 1, 0, acc=1,
                    splitk=self._splitk(N, K, M), c_f32=1, nb1=nb, sA=(so, 0), sB=(sx, 0), sC=(sw, 0),
                    atomic=int(nb > 1 and not grouped), rowsum=_ptr(b.grad) if want_db else None, srow=sw), reads=dz)
            elif want_db:
                assert not grouped
                self.L.tc_colsum(_ptr(dz), M, N, dz.stride(0), nb, so, _ptr(b.grad), 1, self.dt, self.stream)
            if residual is not None:
                if nb == 1 or (grouped and not side_by_side):
                    self.pass_grad(residual, dy)
                else:
                    assert grouped or sr == so
                    self._pass_grad_batched(residual, dy, nb, M, N, so, sr if grouped else so)
        self._rec(bwd)
        return out

    def linear_many(self, items: List[tuple]) -> List[Var]:
        """Independent Linear layers (x, W, b, out, residual[, batch[, post_scale]]) of DIFFERENT shapes in one launch (tc_gemm_multi,
        at most 12 problems per launch); their gradient GEMMs likewise.  out_i = x_i W_i^T + b_i + residual_i (times post_scale, see
        linear()).  batch = (nb, sx, so, sr) as in linear(): batch j of x / out / residual lives sx / so / sr elements after batch 0
        (row re-layout between buffers)."""
        n = len(items)
        assert self.ngroups == 1 and n >= 1
        items = [tuple(it) + (None,) * (7 - len(it)) for it in items]
        wsb = _workspace(self.dev, self.stream, "many")
        sl = (wsb.numel() // 8) & ~16383                    # a private, full-size workspace slice per problem (counters + partials)

        def desc(i, *a, **k):
            g = self._gemm_desc(*a, use_ws=False, **k)
            if i < 8:
                g.ws, g.ws_bytes = wsb.data_ptr() + i * sl, sl
            return g

        def launch(probs):
            for c0 in range(0, len(probs), 12):
                chunk = probs[c0:c0 + 12]
                arr = (TcGemm * len(chunk))(*chunk)
                self.n_launch += 1
                _timed_gemm("multi", chunk, lambda: self.L.tc_gemm_multi(arr, len(chunk), self.stream))
        fw = []
        for i, (x, W, b, out, res, batch, ps) in enumerate(items):
            N, K = W.data.shape
            nb, sx, so, sr = batch if batch is not None else (1, 0, 0, 0)
            assert x.cols == K and out.rows == x.rows and out.cols == N
            fw.append(desc(i % 12, _ptr(x.data), x.ld, _ptr(W.data), W.data.stride(0), _ptr(out.data), out.ld, x.rows, N, K, 0, 1,
                           bias=_ptr(b.data) if b is not None else None, R=_ptr(res.data) if res is not None else None,
                           ldr=res.ld if res is not None else 0, nb1=nb, sA=(sx, 0), sC=(so, 0), sR=(sr, 0),
                           alpha=1.0 if ps is None else ps, act=ACT_NONE if ps is None else ACT_SCALE))
        launch(fw)

        def bwd():
            dxs, dws = [], []
            waves: List[list] = []                       # dX problems whose outputs overlap must not share a launch
            for i, (x, W, b, out, res, batch, _ps) in enumerate(items):
                dy = self.grad_of(out)
                if dy is None:
                    continue
                N, K = W.data.shape
                M = x.rows
                nb, sx, so, sr = batch if batch is not None else (1, 0, 0, 0)
                if x.requires_grad:
                    ext = (nb - 1) * sx // x.root.cols if nb > 1 else 0
                    gx, acc = self.wgrad(x, ext)
                    reg = (id(x.root), (x.region[0], x.region[1] + ext, x.region[2], x.region[3]))
                    wi = 0
                    while wi < len(waves) and any(r[0] == reg[0] and _overlap(r[1], reg[1]) for r, _ in waves[wi]):
                        wi += 1
                    if wi == len(waves):
                        waves.append([])
                    waves[wi].append((reg, desc(len(waves[wi]) % 8, _ptr(dy), dy.stride(0), _ptr(W.data), W.data.stride(0), _ptr(gx),
                                                gx.stride(0), M, K, N, 0, 0, acc=acc, nb1=nb, sA=(so, 0), sC=(sx, 0))))
                if W.grad is not None:
                    dws.append(self._gemm_desc(_ptr(dy), dy.stride(0), _ptr(x.data), x.ld, _ptr(W.grad), W.grad.stride(0), N, K, M, 1, 0,
                                               acc=1, splitk=self._splitk(N, K, M), c_f32=1, nb1=nb, sA=(so, 0), sB=(sx, 0),
                                               atomic=1,      # several problems / batches may share one weight
                                               rowsum=_ptr(b.grad) if (b is not None and b.grad is not None) else None,
                                               use_ws=False))
                if res is not None:
                    if nb == 1:
                        self.pass_grad(res, dy)
                    else:
                        self._pass_grad_batched(res, dy, nb, M, N, so, sr)
            for wi, wv in enumerate(waves):              # (a later wave accumulates onto what an earlier one stored)
                launch([d for _, d in wv] + (dws if wi == 0 else []))
            if not waves:
                launch(dws)
        self._rec(bwd)
        return [it[3] for it in items]

    def linear_multi(self, x: Var, Ws: List[P], bs: List[P], out: Var) -> Var:
        """n Linear layers with the same [N, K] shape on the SAME input, written side by side into out [M, n*N], as ONE batched
        GEMM (weight i lives a constant stride after weight 0 in the parameter arena) -- EfficientAttention's keys / queries /
        values (MSTr.py:109-111).  Backward: dX is one K = n*N product over the gapped weight stack (TcGemm.bgap), the n weight
        gradients one batched GEMM; both share a launch (tc_gemm_pair)."""
        n = len(Ws)
        N, K = Ws[0].data.shape
        M = x.rows
        es = Ws[0].data.element_size()
        sw = (Ws[1].data.data_ptr() - Ws[0].data.data_ptr()) // es
        assert self.ngroups == 1 and out.rows == M and out.cols == n * N and x.cols == K
        assert all(W.data.shape == (N, K) and W.data.is_contiguous() for W in Ws)
        assert all((Ws[i].data.data_ptr() - Ws[0].data.data_ptr()) // es == i * sw and
                   (bs[i].data.data_ptr() - bs[0].data.data_ptr()) // es == i * sw for i in range(n))
        assert N % 64 == 0 and (sw - N * K) % 8 == 0
        self._gemm(_ptr(x.data), x.ld, _ptr(Ws[0].data), K, _ptr(out.data), out.ld, M, N, K, 0, 1, bias=_ptr(bs[0].data), nb1=n,
                   sA=(0, 0), sB=(sw, 0), sC=(N, 0), sbias=sw)

        def bwd():
            dy = self.grad_of(out)
            if dy is None:
                return
            have_w = Ws[0].grad is not None
            gsw = (Ws[1].grad.data_ptr() - Ws[0].grad.data_ptr()) // 4 if have_w else 0
            ga = gb = None
            if x.requires_grad:
                gx, acc = self.wgrad(x)
                ga = self._gemm_desc(_ptr(dy), dy.stride(0), _ptr(Ws[0].data), K, _ptr(gx), gx.stride(0), M, K, n * N, 0, 0, acc=acc,
                                     bgap=(N, sw - N * K))
            if have_w:
                gb = self._gemm_desc(_ptr(dy), dy.stride(0), _ptr(x.data), x.ld, _ptr(Ws[0].grad), K, N, K, M, 1, 0, acc=1,
                                     splitk=self._splitk(N, K, M), c_f32=1, nb1=n, sA=(N, 0), sB=(0, 0), sC=(gsw, 0),
                                     rowsum=_ptr(bs[0].grad) if bs[0].grad is not None else None, srow=gsw, use_ws=False)
            self.n_launch += 1
            if ga is not None and gb is not None:
                _timed_gemm("pair", [ga, gb], lambda: self.L.tc_gemm_pair(C.byref(ga), C.byref(gb), self.stream))
            elif ga is not None:
                _timed_gemm("single", [ga], lambda: self.L.tc_gemm(C.byref(ga), self.stream))
            elif gb is not None:
                _timed_gemm("single", [gb], lambda: self.L.tc_gemm(C.byref(gb), self.stream))
        self._rec(bwd)
        return out

    # ------------------------------------------------------------------ MixFFN_skip, fused
    def _launch_gemms(self, descs: List[TcGemm]):
        """One launch for a list of GEMM problems: tc_gemm, the dX / dW pair kernel, or the merged grid (<= 12 problems)."""
        if not descs:
            return
        self.n_launch += 1
        if len(descs) == 1:
            _timed_gemm("single", descs, lambda: self.L.tc_gemm(C.byref(descs[0]), self.stream))
        elif len(descs) == 2 and not descs[0].transA and not descs[0].transB and descs[1].transA and not descs[1].transB:
            _timed_gemm("pair", descs, lambda: self.L.tc_gemm_pair(C.byref(descs[0]), C.byref(descs[1]), self.stream))
        else:
            for c0 in range(0, len(descs), 12):
                chunk = descs[c0:c0 + 12]
                _timed_gemm("multi", chunk, lambda: self.L.tc_gemm_multi((TcGemm * len(chunk))(*chunk), len(chunk), self.stream))

    def mixffn(self, sites: List[dict]) -> List[Var]:
        """MixFFN_skip (MSTr.py:889-902, fc1 evaluated once): out = fc2(GELU(LN(dw3x3(h) + h))) + residual, h = fc1(x), for one site
        (optionally `ngroups` stacked weight groups) or several independent sites (the four scales of a bridge layer) level by level.

        Forward = 3 launches: fc1 GEMM; depthwise conv + skip that also leaves LayerNorm chunk partials (tc_ffn_dw_fwd); fc2 GEMM whose
        A-operand loader applies LayerNorm + GELU (TC_FFN_LN_A) -- the normalised / activated hidden map never exists.
        Backward = 3 launches: {fc2 input gradient with GELU' and the LayerNorm row sums in its epilogue (TC_FFN_EP), fc2 weight gradient
        recomputing GELU(LN(d)) in its B loader (TC_FFN_LN_B)}; tc_ffn_mid_bwd (LayerNorm backward + depthwise input and weight gradients +
        LayerNorm parameter gradients); {fc1 input gradient, fc1 weight gradient}.  The unfused form takes 4 + 5 launches per site and
        6 + 10 passes over hidden-width maps; this one 4 + 8.
        site = dict(x, fc1=(W, b), dw=(w, b), ln=(g, b), fc2=(W, b), geo=(B, H, W), residual=None|Var, out=None|Var)."""
        n, Gn, L = len(sites), self.ngroups, self.L
        assert n >= 1 and (Gn == 1 or n == 1)
        gs = self.pgs if Gn > 1 else 0
        many = n > 1
        wsb = _workspace(self.dev, self.stream, "many") if many else None
        sl = (wsb.numel() // 8) & ~16383 if many else 0

        def desc(i, *a, **k):
            if not many:
                return self._gemm_desc(*a, **k)
            g = self._gemm_desc(*a, use_ws=False, **{kk: vv for kk, vv in k.items() if kk != "use_ws"})
            if i < 8 and k.get("use_ws", True):
                g.ws, g.ws_bytes = wsb.data_ptr() + i * sl, sl
            return g

        st = []
        for s_ in sites:
            x = s_["x"]
            (W1, b1), (wd, bd), (lg, lb), (W2, b2) = s_["fc1"], s_["dw"], s_["ln"], s_["fc2"]
            B, H, W = s_["geo"]
            C4, Cin = W1.data.shape
            M = x.rows // Gn
            assert x.cols == Cin and M == B * H * W and W2.data.shape == (Cin, C4)
            cn = int(L.tc_ffn_chunk(C4, self.dt))
            assert cn > 0 and C4 % cn == 0 and C4 % 8 == 0
            out = s_.get("out")
            if out is None:
                out = self.new(x.rows, Cin)
            # stacked weight groups write row blocks of `out`, or -- the last MB layer -- column blocks of the IFF concat buffer
            side = Gn > 1 and out.rows == M and out.cols == Gn * Cin
            assert side or (out.rows == x.rows and out.cols == Cin)
            # LayerNorm + GELU inside the fc2 GEMM's A loader pays when the product has ONE 64-column tile (Cin <= 64: every element is
            # transformed once); with N tiles across Cin the exact-erf GELU would be recomputed N / 64 times, so wider sites run the
            # LayerNorm kernel on d (it also writes the activated map the weight gradient reads) and a plain fc2 GEMM
            lng = Cin <= _FFN_LN_GEMM_MAXC
            # one spatially tiled kernel for the whole forward site (hidden maps in LDS) where the library has it: 16-bit storage, C = 64 / 128
            fz = (_FFN_TILED and self.dtype != torch.float32 and bool(L.tc_ffn_fused_supported(Cin, self.dt)) and x.ld % 8 == 0 and out.ld % 8 == 0
                  and (s_.get("residual") is None or s_["residual"].ld % 8 == 0) and gs % 8 == 0)
            # ... and its backward from d + the row statistics alone (h, a, gp, dh never exist in HBM)
            fzb = (fz and _FFN_TILED_BWD and self.record and bool(L.tc_ffn_fused_bwd_supported(Cin, self.dt)) and x.requires_grad
                   and all(q.grad is not None for q in (W1, b1, wd, bd, lg, lb, W2, b2)))
            # the block's LayerNorm ahead of the site (norm2): inside the tiled kernels where both directions are tiled, a launch of its own otherwise
            pre = s_.get("pre_ln")
            if pre is not None and not (_FFN_PRE_LN and fz and (fzb or not self.record) and Cin == 64 and (not self.record or pre[0].grad is not None)):
                x = self.layernorm(x, pre[0], pre[1], pre[2])
                pre = None
            st.append(dict(pre=pre, fz=fz, fzb=fzb, lng=lng, so=Cin if side else M * out.ld, side=side, x=x, W1=W1, b1=b1, wd=wd, bd=bd, lg=lg, lb=lb, W2=W2, b2=b2, B=B, H=H, W=W, C4=C4, Cin=Cin, M=M, cn=cn,
                           nch=C4 // cn, nch2=(C4 + 63) // 64, res=s_.get("residual"), out=out,
                           h=_empty((x.rows, C4), self.dtype, self.dev) if not fzb else None, d=_empty((x.rows, C4), self.dtype, self.dev),
                           a=_empty((x.rows, C4), self.dtype, self.dev) if ((_FFN_STORE_ACT or not lng or fz) and not fzb) else None,
                           part=self.f32(x.rows * (C4 // cn) * 2) if not fz else None, stat=self.f32(x.rows * 2)))
        for t in st:
            if t["fz"]:
                res, out = t["res"], t["out"]
                f = TcFfnFused(_ptr(t["x"].data), _ptr(t["W1"].data), _ptr(t["b1"].data), _ptr(t["wd"].data), _ptr(t["bd"].data), _ptr(t["lg"].data),
                               _ptr(t["lb"].data), _ptr(t["W2"].data), _ptr(t["b2"].data), _ptr(res.data) if res is not None else None, _ptr(out.data),
                               _ptr(t["h"]) if (self.record and t["h"] is not None) else None, _ptr(t["d"]) if self.record else None,
                               _ptr(t["a"]) if (self.record and t["a"] is not None) else None,
                               _ptr(t["stat"]) if self.record else None,
                               t["M"] * res.ld if res is not None else 0, t["so"], gs, t["x"].ld, res.ld if res is not None else 0, out.ld,
                               t["Cin"], t["B"], t["H"], t["W"], Gn, 1e-5, *_FFN_TILE,
                               *((_ptr(t["pre"][0].data), _ptr(t["pre"][1].data), t["pre"][2]) if t["pre"] is not None else (None, None, 0.0)))
                self.n_launch += 1
                _timed("hbm:ffn_fused_fwd (MixFFN forward, one tiled kernel)", (2.0 * t["Cin"] + (t["Cin"] if res is not None else 0) + t["C4"]) * t["x"].rows * t["d"].element_size(),
                       lambda f=f: L.tc_ffn_fused_fwd(C.byref(f), self.dt, self.stream))
        st_all, st = st, [t for t in st if not t["fz"]]
        nold = len(st)

        def hook(g: TcGemm, t, mode):
            g.ffn_mode, g.ffn_nchunk, g.ffn_chunk_n, g.ffn_ldd, g.ffn_eps = mode, t["nch"], t["cn"], t["C4"], 1e-5
            g.ffn_part, g.ffn_stat, g.ffn_gamma, g.ffn_beta = _ptr(t["part"]), _ptr(t["stat"]), _ptr(t["lg"].data), _ptr(t["lb"].data)
            g.ffn_d = _ptr(t["d"])
            g.ffn_sRow1, g.ffn_sPar1 = t["M"], gs
            if mode == FFN_LN_A and t["a"] is not None:
                g.ffn_aout = _ptr(t["a"])
            return g

        # fc1
        self._launch_gemms([desc(i, _ptr(t["x"].data), t["x"].ld, _ptr(t["W1"].data), t["Cin"], _ptr(t["h"]), t["C4"], t["M"], t["C4"], t["Cin"], 0, 1,
                                 bias=_ptr(t["b1"].data), nb1=Gn, sA=(t["M"] * t["x"].ld, 0), sB=(gs, 0), sC=(t["M"] * t["C4"], 0), sbias=gs)
                            for i, t in enumerate(st)])
        # depthwise 3x3 + bias + skip, LayerNorm chunk partials on the side
        if nold == 0:
            pass
        elif nold == 1:
            t = st[0]
            _timed("hbm:ffn_dw_fwd (MixFFN dw3x3 + skip + LayerNorm partials)", 2.0 * t["x"].rows * t["C4"] * t["h"].element_size(),
                   lambda: L.tc_ffn_dw_fwd(_ptr(t["h"]), t["C4"], _ptr(t["wd"].data), _ptr(t["bd"].data), _ptr(t["d"]), t["C4"],
                                           _ptr(t["part"]) if t["lng"] else None, t["B"], t["H"], t["W"], t["C4"], Gn, gs, self.dt, self.stream))
        else:
            assert nold <= 4
            arr = (TcDwSeg * nold)()
            for i, t in enumerate(st):
                arr[i] = TcDwSeg(_ptr(t["h"]), _ptr(t["wd"].data), _ptr(t["bd"].data), _ptr(t["d"]), None, None, None, t["C4"], 3, t["C4"], t["C4"], 0,
                                 t["B"], t["H"], t["W"], _ptr(t["part"]) if t["lng"] else None)
            L.tc_dwconv_multi(arr, nold, 0, 1, 0, 1, 0, None, 0, self.dt, self.stream)
        # fc2 on GELU(LN(d)) (+ bias + residual)
        fw = []
        for i, t in enumerate(st):
            out, res = t["out"], t["res"]
            if not t["lng"]:                                 # statistics land interleaved ([rows][2]) where the backward hooks read them
                L.tc_layernorm_fwd(_ptr(t["d"]), t["C4"], _ptr(t["lg"].data), _ptr(t["lb"].data), _ptr(t["a"]), t["C4"], _ptr(t["stat"]),
                                   t["stat"].data_ptr() + 4, t["M"], t["C4"], 1e-5, ACT_GELU, Gn, gs, self.dt, self.stream)
            g = desc(i, _ptr(t["d"] if t["lng"] else t["a"]), t["C4"], _ptr(t["W2"].data), t["C4"], _ptr(out.data), out.ld, t["M"], t["Cin"], t["C4"],
                     0, 1, bias=_ptr(t["b2"].data), R=_ptr(res.data) if res is not None else None, ldr=res.ld if res is not None else 0,
                     nb1=Gn, sA=(t["M"] * t["C4"], 0), sB=(gs, 0), sC=(t["so"], 0), sR=(t["M"] * res.ld if res is not None else 0, 0), sbias=gs)
            fw.append(hook(g, t, FFN_LN_A) if t["lng"] else g)
        self._launch_gemms(fw)

        def bwd_tiled(t, dy):
            """One site through csrc/mixffn_bwd.hip: three launches, gd the only hidden-width map in HBM."""
            x = t["x"]
            gx, acc = self.wgrad(x)
            gd = _empty((x.rows, t["C4"]), self.dtype, self.dev)
            nf = int(L.tc_ffn_fused_bwd_scratch_floats(t["Cin"], Gn))
            part = self.f32(nf)
            f = TcFfnBwd(_ptr(x.data), _ptr(dy), _ptr(t["d"]), _ptr(t["stat"]), _ptr(t["W1"].data), _ptr(t["b1"].data), _ptr(t["wd"].data),
                         _ptr(t["lg"].data), _ptr(t["lb"].data), _ptr(t["W2"].data), _ptr(gx), _ptr(gd), _ptr(part), nf,
                         _ptr(t["W1"].grad), _ptr(t["b1"].grad), _ptr(t["wd"].grad), _ptr(t["bd"].grad), _ptr(t["lg"].grad), _ptr(t["lb"].grad),
                         _ptr(t["W2"].grad), _ptr(t["b2"].grad), t["so"], gs, x.ld, dy.stride(0), gx.stride(0), t["Cin"], t["B"], t["H"], t["W"], Gn,
                         acc, 1e-5, *_FFN_TILE_BWD,
                         *((_ptr(t["pre"][0].data), _ptr(t["pre"][1].data), _ptr(t["pre"][0].grad), _ptr(t["pre"][1].grad), t["pre"][2]) if t["pre"] is not None
                           else (None, None, None, None, 0.0)))
            self.n_launch += 3
            es = t["d"].element_size()
            _timed("hbm:ffn_fused_bwd (MixFFN backward on the chip: LayerNorm/GELU backward + fc2 gradients, dw3x3/fc1 gradients, partial fold)",
                   (3.0 * t["Cin"] + 3.0 * t["C4"]) * x.rows * es, lambda: L.tc_ffn_fused_bwd(C.byref(f), self.dt, self.stream))

        def bwd():
            dys_all = [self.grad_of(t["out"]) for t in st_all]
            if all(d is None for d in dys_all):
                return
            assert all(d is not None for d in dys_all)
            for t, dy in zip(st_all, dys_all):
                if t["fzb"]:
                    bwd_tiled(t, dy)
            pairs = [(t, dy) for t, dy in zip(st_all, dys_all) if not t["fzb"]]
            st, dys = [t for t, _ in pairs], [dy for _, dy in pairs]
            n = len(st)
            if n:
                bwd_opbyop(st, dys, n)
            copies = []                                     # the sites' residual gradients leave in one launch (<= 4 per tc_ew_multi)
            for t, dy in zip(st_all, dys_all):
                if t["res"] is not None:
                    if t["side"]:
                        copies.append(self._pass_grad_batched(t["res"], dy, Gn, t["M"], t["Cin"], t["so"], t["M"] * t["res"].ld, defer=True))
                    else:
                        copies.append(self.pass_grad(t["res"], dy, defer=True))
            copies = [c for c in copies if c is not None]
            for c0 in range(0, len(copies), 4):
                self._ew_multi(copies[c0:c0 + 4])

        def bwd_opbyop(st, dys, n):
            # fc2: gp = (dY W2) (.) GELU'(u) with the LayerNorm row sums, and dW2 = dY^T GELU(LN(d)) (+ db2)
            g1 = []
            for i, (t, dy) in enumerate(zip(st, dys)):
                t["gp"] = _empty((t["x"].rows, t["C4"]), self.dtype, self.dev)
                t["part2"] = self.f32(t["x"].rows * t["nch2"] * 2)
                ga = hook(desc(i, _ptr(dy), dy.stride(0), _ptr(t["W2"].data), t["C4"], _ptr(t["gp"]), t["C4"], t["M"], t["C4"], t["Cin"], 0, 0,
                               nb1=Gn, sA=(t["so"], 0), sB=(gs, 0), sC=(t["M"] * t["C4"], 0), use_ws=False), t, FFN_EP)
                ga.ffn_part2 = _ptr(t["part2"])
                g1.append(ga)
            for i, (t, dy) in enumerate(zip(st, dys)):
                if t["W2"].grad is None:
                    continue
                # dW2 = dY^T a: `a` as stored by the forward fc2 kernel, or recomputed from d in the B loader (TC_FFN_LN_B)
                gb = desc(n + i, _ptr(dy), dy.stride(0), _ptr(t["a"] if t["a"] is not None else t["d"]), t["C4"], _ptr(t["W2"].grad), t["C4"], t["Cin"],
                          t["C4"], t["M"], 1, 0, acc=1, splitk=self._splitk(t["Cin"], t["C4"], t["M"]), c_f32=1, nb1=Gn, sA=(t["so"], 0),
                          sB=(t["M"] * t["C4"], 0), sC=(gs, 0), rowsum=_ptr(t["b2"].grad) if t["b2"].grad is not None else None, srow=gs,
                          use_ws=False)
                g1.append(gb if t["a"] is not None else hook(gb, t, FFN_LN_B))
            self._launch_gemms(g1)
            # LayerNorm backward + depthwise input / weight gradients + LayerNorm parameter gradients
            segs = (TcFfnSeg * n)()
            for i, t in enumerate(st):
                t["dh"] = _empty((t["x"].rows, t["C4"]), self.dtype, self.dev)
                hg = t["wd"].grad is not None
                segs[i] = TcFfnSeg(_ptr(t["gp"]), _ptr(t["d"]), _ptr(t["h"]), _ptr(t["dh"]), _ptr(t["stat"]), _ptr(t["part2"]), _ptr(t["wd"].data),
                                   _ptr(t["lg"].data), _ptr(t["wd"].grad) if hg else None, _ptr(t["bd"].grad) if hg else None,
                                   _ptr(t["lg"].grad) if hg else None, _ptr(t["lb"].grad) if hg else None,
                                   t["C4"], t["C4"], t["C4"], t["C4"], t["C4"], t["B"], t["H"], t["W"], t["nch2"])
            nf = 0
            if _DW_DEFER and not self.use_streams and all(t["wd"].grad is not None for t in st):     # the walkers' sums parked; one tc_dw_fold per backward leg
                from ._lib import TcDwFold
                sites, offs = (TcDwFold * n)(), (C.c_longlong * n)()
                nf = int(L.tc_ffn_mid_plan(segs, n, Gn, self.dt, sites, offs))
            if nf > 0:
                part = self.f32(nf)
                _timed("hbm:ffn_mid_bwd (MixFFN LayerNorm backward + dw3x3 input/weight gradients)",
                       sum(4.0 * t["x"].rows * t["C4"] * t["h"].element_size() for t in st),
                       lambda: L.tc_ffn_mid_bwd(segs, n, Gn, gs, _ptr(part), -4 * nf, self.dt, self.stream))
                for i, t in enumerate(st):
                    sf = TcDwFold()
                    C.memmove(C.byref(sf), C.byref(sites[i]), C.sizeof(TcDwFold))
                    sf.part, sf.wstride = _ptr(part) + 4 * offs[i], gs
                    sf.dw, sf.db, sf.dgamma, sf.dbeta = _ptr(t["wd"].grad), _ptr(t["bd"].grad), _ptr(t["lg"].grad), _ptr(t["lb"].grad)
                    self._dw_pending.append((sf, part))
            else:
                ws = _workspace(self.dev, self.stream)
                _timed("hbm:ffn_mid_bwd (MixFFN LayerNorm backward + dw3x3 input/weight gradients)",
                       sum(4.0 * t["x"].rows * t["C4"] * t["h"].element_size() for t in st),
                       lambda: L.tc_ffn_mid_bwd(segs, n, Gn, gs, ws.data_ptr(), ws.numel(), self.dt, self.stream))
            # fc1: dX = dh W1, dW1 = dh^T x (+ db1)
            g2 = []
            for i, t in enumerate(st):
                x = t["x"]
                if x.requires_grad:
                    gx, acc = self.wgrad(x)
                    g2.append(desc(i, _ptr(t["dh"]), t["C4"], _ptr(t["W1"].data), t["Cin"], _ptr(gx), gx.stride(0), t["M"], t["Cin"], t["C4"], 0, 0,
                                   acc=acc, nb1=Gn, sA=(t["M"] * t["C4"], 0), sB=(gs, 0), sC=(t["M"] * gx.stride(0), 0)))
            for i, t in enumerate(st):
                if t["W1"].grad is not None:
                    x = t["x"]
                    g2.append(desc(n + i, _ptr(t["dh"]), t["C4"], _ptr(x.data), x.ld, _ptr(t["W1"].grad), t["Cin"], t["C4"], t["Cin"], t["M"], 1, 0,
                                   acc=1, splitk=self._splitk(t["C4"], t["Cin"], t["M"]), c_f32=1, nb1=Gn, sA=(t["M"] * t["C4"], 0),
                                   sB=(t["M"] * x.ld, 0), sC=(gs, 0), rowsum=_ptr(t["b1"].grad) if t["b1"].grad is not None else None, srow=gs,
                                   use_ws=False))
            self._launch_gemms(g2)
            for t in st:
                for k in ("gp", "part2", "dh"):
                    t.pop(k, None)
        self._rec(bwd)
        return [t["out"] for t in st_all]

    def _pass_grad_batched(self, v: Var, src: torch.Tensor, nb: int, M: int, N: int, sb_src: int, sb_dst: Optional[int] = None,
                           defer: bool = False):
        """v.grad (+)= src for nb batch-strided [M, N] blocks (tc_copy3d); defer: return the copy as a tc_ew_multi segment instead."""
        sb_dst = sb_src if sb_dst is None else sb_dst
        g, acc = self.wgrad(v, 0 if v.rows >= nb * M else (nb - 1) * sb_dst // v.root.cols)
        if defer:
            return TcEwSeg(EW_COPY, acc, src.data_ptr(), g.data_ptr(), sb_src, sb_dst, src.stride(0), g.stride(0), nb, M, N, 0, 0, 0)
        self.L.tc_copy3d(_ptr(src), sb_src, src.stride(0), _ptr(g), sb_dst, g.stride(0), nb, M, N, acc, self.dt, self.stream)

    def effatt_supported(self, t: Var, params: Optional[Tuple[P, ...]] = None) -> bool:
        """params: the ten parameters of the block.  The fused backward writes the input gradient and all ten parameter gradients in its
        launches, so when recording a backward every one of them (and t) must take a gradient -- a frozen norm / projection falls back to
        the op-by-op path, which handles any subset."""
        if self.record and (not t.requires_grad or (params is not None and any(p.grad is None for p in params))):
            return False
        return (_EFFATT_FUSED and self.ngroups == 1 and not self.use_streams and self.dt != TC_F32
                and bool(self.L.tc_effatt_supported(t.cols, self.dt)) and t.data.is_contiguous())

    def eff_attention_block(self, t: Var, ln: Tuple[P, P], keys: Tuple[P, P], queries: Tuple[P, P], values: Tuple[P, P],
                            reproj: Tuple[P, P], B: int, N: int, eps: float = 1e-5) -> Var:
        """t + EfficientAttention(LayerNorm(t)) (MSTr.py:166-167 around :106-143, one head) through csrc/effatt.hip: 3 launches forward,
        4 backward, nothing N-sized in HBM but t, the result and one C-wide gradient scratch (op by op: 10 + 12 launches and the
        K | Q | V maps, both softmaxes and the attended map, each twice)."""
        L, Cc = self.L, t.cols
        assert t.rows == B * N and self.effatt_supported(t)
        out = self.new(t.rows, Cc)
        nf = int(L.tc_effatt_scratch_floats(Cc, B, N))
        ctx, kstat = self.f32(B, Cc, Cc), self.f32(B, 2, Cc)
        es = t.data.element_size()

        def desc(part, dout=None, dt=None, g1=None, acc=0, grads=False):
            f = TcEffAtt()
            f.t, f.gamma, f.beta = _ptr(t.data), _ptr(ln[0].data), _ptr(ln[1].data)
            for nm, (W, b) in (("k", keys), ("q", queries), ("v", values), ("r", reproj)):
                setattr(f, "w" + nm, _ptr(W.data)); setattr(f, "b" + nm, _ptr(b.data))
                if grads:
                    setattr(f, "dw" + nm, _ptr(W.grad)); setattr(f, "db" + nm, _ptr(b.grad))
            if grads:
                f.dgamma, f.dbeta = _ptr(ln[0].grad), _ptr(ln[1].grad)
            f.out, f.ctx, f.kstat, f.part, f.part_floats = _ptr(out.data), _ptr(ctx), _ptr(kstat), _ptr(part), nf
            f.ldt, f.ldo, f.C, f.B, f.N, f.eps = t.ld, out.ld, Cc, B, N, eps
            if dout is not None:
                f.dout, f.lddo, f.dt, f.lddt, f.g1, f.acc_dt = _ptr(dout), dout.stride(0), _ptr(dt), dt.stride(0), _ptr(g1), acc
            return f

        part = self.f32(nf)
        f = desc(part)
        self.n_launch += 3
        _timed("hbm:effatt_fwd (LayerNorm + EfficientAttention + residual: token statistics, context fold, output)",
               3.0 * t.rows * Cc * es, lambda: L.tc_effatt_fwd(C.byref(f), self.dt, self.stream))
        del part

        def bwd():
            dy = self.grad_of(out)
            if dy is None or not t.requires_grad:
                return
            gx, acc = self.wgrad(t)
            part = self.f32(nf)
            g1 = _empty((t.rows, Cc), self.dtype, self.dev)
            fb = desc(part, dy, gx, g1, acc, grads=True)
            self.n_launch += 4
            _timed("hbm:effatt_bwd (query side + d_ctx partials, d_ctx fold, key/value side + LayerNorm backward, parameter fold)",
                   (6.0 + acc) * t.rows * Cc * es, lambda: L.tc_effatt_bwd(C.byref(fb), self.dt, self.stream))
        self._rec(bwd)
        return out

    def layernorm(self, x: Var, g: P, b: P, eps: float = 1e-5, act: int = ACT_NONE, out: Optional[Var] = None,
                  stats: Optional[Tuple[torch.Tensor, torch.Tensor]] = None, launch: bool = True) -> Var:
        """launch=False: a fused kernel of the caller writes `out` and stats = (mean, rstd); only the backward closure is recorded."""
        Gn = self.ngroups
        rows, Cc = x.rows // Gn, x.cols
        assert Gn == 1 or (g.gs > 0 and b.gs == g.gs)
        if out is None:
            out = self.new(x.rows, Cc)
        mean, rstd = stats if stats is not None else (self.f32(x.rows), self.f32(x.rows))
        es = x.data.element_size()
        if launch:
            _timed("hbm:layernorm_fwd", 2.0 * x.rows * Cc * es, lambda: self.L.tc_layernorm_fwd(
                _ptr(x.data), x.ld, _ptr(g.data), _ptr(b.data), _ptr(out.data), out.ld, _ptr(mean), _ptr(rstd), rows, Cc, eps, act, Gn, g.gs,
                self.dt, self.stream))

        def bwd():
            dy = self.grad_of(out)
            if dy is None or not x.requires_grad:
                return
            gx, acc = self.wgrad(x)
            fused = g.grad is not None and not (self.overlap_wgrad and self.use_streams)
            nblk = int(self.L.tc_layernorm_bwd_nblk(rows, Cc)) if (fused and _LN_DEFER and not self.use_streams) else 0
            if nblk > 0:                                 # dx + per-workgroup dgamma / dbeta partials; the partials of ALL LayerNorms of a backward
                nf = Gn * nblk * 2 * Cc                  # leg are added up by one launch when the leg ends (_flush_param_folds)
                part = self.f32(nf)
                _timed("hbm:layernorm_bwd", (3.0 + acc) * x.rows * Cc * es, lambda: self.L.tc_layernorm_bwd_defer(
                    _ptr(dy), dy.stride(0), _ptr(x.data), x.ld, _ptr(g.data), _ptr(b.data), _ptr(mean), _ptr(rstd), _ptr(gx), gx.stride(0),
                    _ptr(gx) if acc else None, gx.stride(0), _ptr(g.grad), _ptr(b.grad), rows, Cc, act, Gn, g.gs, _ptr(part), nf, self.dt,
                    self.stream))
                self._ln_pending.append((part, g.grad, b.grad, g.gs, nblk, Cc, Gn))
                if len(self._ln_pending) == 64:
                    self._flush_param_folds()
                return
            if fused:                                    # one pass: dx + per-workgroup dgamma/dbeta partials + a tiny folding launch
                ws = _workspace(self.dev, self.stream)
                scratch, n = ws, ws.numel() // 4
                _timed("hbm:layernorm_bwd", (3.0 + acc) * x.rows * Cc * es, lambda: self.L.tc_layernorm_bwd(
                    _ptr(dy), dy.stride(0), _ptr(x.data), x.ld, _ptr(g.data), _ptr(b.data), _ptr(mean), _ptr(rstd), _ptr(gx), gx.stride(0),
                    _ptr(gx) if acc else None, gx.stride(0), _ptr(g.grad), _ptr(b.grad), rows, Cc, act, Gn, g.gs, _ptr(scratch), n, self.dt,
                    self.stream))
                return
            self.L.tc_layernorm_bwd(_ptr(dy), dy.stride(0), _ptr(x.data), x.ld, _ptr(g.data), _ptr(b.data), _ptr(mean),
                                    _ptr(rstd), _ptr(gx), gx.stride(0), _ptr(gx) if acc else None, gx.stride(0),
                                    None, None, rows, Cc, act, Gn, g.gs, None, 0, self.dt, self.stream)
            if g.grad is not None:
                self._weight_grad(lambda: self.L.tc_layernorm_bwd_params(
                    _ptr(dy), dy.stride(0), _ptr(x.data), x.ld, _ptr(g.data), _ptr(b.data), _ptr(mean), _ptr(rstd), _ptr(g.grad),
                    _ptr(b.grad), rows, Cc, act, Gn, g.gs, self.dt, self.stream), reads=dy)
        self._rec(bwd)
        return out

    def layernorm_shuffled(self, x: Var, g: P, b: P, B: int, H: int, W: int, p: int, eps: float = 1e-5) -> Var:
        """LayerNorm(c) of the pixel-shuffled map 'b h w (p1 p2 c) -> b (h p1) (w p2) c' of x [B*H*W, p*p*c] (PatchExpand,
        MSTr.py:196-199,222-225) with the shuffle done by the kernel's addressing: no shuffled copy in either direction."""
        assert self.ngroups == 1 and x.rows == B * H * W and x.cols % (p * p) == 0
        c = x.cols // (p * p)
        rows = B * H * p * W * p
        out = self.new(rows, c)
        mean, rstd = self.f32(rows), self.f32(rows)
        es = x.data.element_size()
        _timed("hbm:layernorm_fwd", 2.0 * rows * c * es, lambda: self.L.tc_layernorm_ps_fwd(
            _ptr(x.data), x.ld, _ptr(g.data), _ptr(b.data), _ptr(out.data), out.ld, _ptr(mean), _ptr(rstd), B, H, W, p, c, eps, self.dt,
            self.stream))

        def bwd():
            dy = self.grad_of(out)
            if dy is None or not x.requires_grad:
                return
            gx, acc = self.wgrad(x)
            assert not acc, "the expanded map has one consumer"
            ws = _workspace(self.dev, self.stream)
            _timed("hbm:layernorm_bwd", 3.0 * rows * c * es, lambda: self.L.tc_layernorm_ps_bwd(
                _ptr(dy), dy.stride(0), _ptr(x.data), x.ld, _ptr(g.data), _ptr(b.data), _ptr(mean), _ptr(rstd), _ptr(gx), gx.stride(0),
                _ptr(g.grad), _ptr(b.grad), B, H, W, p, c, _ptr(ws) if g.grad is not None else None, ws.numel() // 4, self.dt, self.stream))
        self._rec(bwd)
        return out

    def ln_cls_supported(self, x: Var, p: int, g: P, b: P, Wc: P, bc: Optional[P], B: int = 0, H: int = 0, W: int = 0) -> bool:
        c = x.cols // max(p * p, 1)
        if p >= 2 and B * H * p * W * p * max(H, W) * p >= 1 << 32:        # the kernels' pixel-shuffle addressing divides by multiplication (csrc/lncls.hip, ps_map)
            return False
        return (_LN_CLS_FUSED and self.dt != TC_F32 and self.ngroups == 1 and not self.use_streams and bc is not None and x.ld % 8 == 0
                and all(q.data.data_ptr() % 16 == 0 for q in (x, g, b, Wc)) and Wc.data.is_contiguous() and Wc.data.shape[1] == c
                and bool(self.L.tc_ln_cls_supported(c, Wc.data.shape[0], self.dt)))

    def ln_cls(self, x: Var, g: P, b: P, Wc: P, bc: P, B: int, H: int, W: int, p: int, eps: float = 1e-5, pad_rows: bool = False) -> Var:
        """logits = classifier(LayerNorm(pixel-shuffled x)) -- FinalPatchExpand_X4's rearrange + norm and last_layer (MSTr.py:222-225, 281) -- as
        ONE forward launch and one backward call (tc_ln_cls_fwd / _bwd): the normalised [B (H p) (W p), c] map and its gradient never reach
        memory.  pad_rows: the logits (and so their gradient) get 16-byte-aligned rows (row stride ncls rounded up to 8); the returned Var is
        the [rows, ncls] column slice of that buffer."""
        assert self.ngroups == 1 and x.rows == B * H * W and x.cols % max(p * p, 1) == 0
        c = x.cols // max(p * p, 1)
        ncls = Wc.data.shape[0]
        rows = B * H * p * W * p if p else x.rows
        out = self.new(rows, (ncls + 7) // 8 * 8).colslice(0, ncls) if pad_rows else self.new(rows, ncls)
        mean, rstd = self.f32(rows), self.f32(rows)
        es = x.data.element_size()
        self.n_launch += 1
        _timed("hbm:ln_cls_fwd (pixel shuffle + LayerNorm + classifier, one launch)", 1.0 * rows * (c + ncls) * es, lambda: self.L.tc_ln_cls_fwd(
            _ptr(x.data), x.ld, _ptr(g.data), _ptr(b.data), _ptr(Wc.data), _ptr(bc.data), _ptr(out.data), out.ld, _ptr(mean), _ptr(rstd),
            B, H, W, p, c, ncls, eps, self.dt, self.stream))

        def bwd():
            dl = self.grad_of(out)
            if dl is None or not x.requires_grad:
                return
            gx, acc = self.wgrad(x)
            assert not acc, "the expanded map has one consumer"
            n = int(self.L.tc_ln_cls_scratch_floats(rows, ncls))
            scratch = self.f32(n)
            _timed("hbm:ln_cls_bwd", 1.0 * rows * (2 * c + ncls) * es, lambda: self.L.tc_ln_cls_bwd(
                _ptr(dl), dl.stride(0), _ptr(x.data), x.ld, _ptr(g.data), _ptr(b.data), _ptr(Wc.data), _ptr(mean), _ptr(rstd), _ptr(gx), gx.stride(0),
                _ptr(g.grad), _ptr(b.grad), _ptr(Wc.grad), _ptr(bc.grad), _ptr(scratch), n, B, H, W, p, c, ncls, self.dt, self.stream))
        self._rec(bwd)
        return out

    def linear_ln_supported(self, x: Var, W: P, residual: Optional[Var], *params: P) -> bool:
        """params: the bias / gamma / beta the call will pass (their 16-byte alignment is the kernel's: it loads them as whole 8-element pieces)"""
        Cc = x.cols
        return (all(q.data.data_ptr() % 16 == 0 for q in params) and _LIN_LN_FUSED and self.dt != TC_F32 and not self.use_streams and tuple(W.data.shape) == (Cc, Cc) and W.data.is_contiguous()
                and x.ld % 8 == 0 and x.data.data_ptr() % 16 == 0 and (residual is None or (residual.ld % 8 == 0 and residual.data.data_ptr() % 16 == 0))
                and (self.ngroups == 1 or self.pgs % 8 == 0) and bool(self.L.tc_linear_ln_supported(Cc, self.dt)))

    def linear_ln(self, x: Var, W: P, b: P, residual: Optional[Var], g: P, beta: P, eps: float, out: Optional[Var] = None,
                  ln_out: Optional[Var] = None) -> Tuple[Var, Var]:
        """(t, LayerNorm(t)) with t = x W^T + b + residual for a square Linear -- `proj` + skip + norm2 of an MHCABlock / bridge layer,
        `reprojection` + skip + norm2 of an EfficientTransformerBlock -- as ONE forward launch (tc_linear_ln_fwd); the backward is the two
        ops' own (their closures are recorded without their forward launches)."""
        Gn = self.ngroups
        t = self.linear(x, W, b, out=out, residual=residual, launch=False)
        mean, rstd = self.f32(x.rows), self.f32(x.rows)
        xn = self.layernorm(t, g, beta, eps, out=ln_out, stats=(mean, rstd), launch=False)
        assert t.rows == x.rows and t.cols == x.cols, "row-stacked groups only"
        self.n_launch += 1
        es = x.data.element_size()
        _timed("hbm:linear_ln_fwd (Linear + residual + LayerNorm, one launch)", (4.0 if residual is not None else 3.0) * x.rows * x.cols * es,
               lambda: self.L.tc_linear_ln_fwd(_ptr(x.data), x.ld, _ptr(W.data), _ptr(b.data), W.gs if Gn > 1 else 0,
                                               _ptr(residual.data) if residual is not None else None, residual.ld if residual is not None else 0,
                                               _ptr(g.data), _ptr(beta.data), g.gs if Gn > 1 else 0, _ptr(t.data), t.ld, _ptr(xn.data), xn.ld,
                                               _ptr(mean), _ptr(rstd), Gn, x.rows // Gn, x.cols, eps, self.dt, self.stream))
        return t, xn

    def dw_ln_supported(self, x: Var) -> bool:
        return (_DW_LN_FUSED and self.dt != TC_F32 and not self.use_streams and x.ld % 8 == 0 and x.data.data_ptr() % 16 == 0
                and (self.pgs % 8 == 0 or self.ngroups == 1) and bool(self.L.tc_dw_ln_supported(x.cols, self.dt)))

    def dw_ln(self, x: Var, w: P, b: P, g: P, beta: P, B: int, H: int, W: int, eps: float) -> Tuple[Var, Var]:
        """(t1, LayerNorm(t1)) with t1 = x + dw3x3(x) + bias -- the head of an MHCABlock (cpe + norm1, MSTr.py:744-752, 935-940) -- as ONE forward
        launch (tc_dw_ln_fwd); the backward is the two ops' own (their closures are recorded without their forward launches)."""
        Gn = self.ngroups
        t1 = self.dwconv(x, w, b, B, H, W, 3, 1, True, launch=False)
        mean, rstd = self.f32(x.rows), self.f32(x.rows)
        xn = self.layernorm(t1, g, beta, eps, stats=(mean, rstd), launch=False)
        self.n_launch += 1
        es = x.data.element_size()
        _timed("hbm:dw_ln_fwd (cpe dw3x3 + skip + LayerNorm, one launch)", 3.0 * x.rows * x.cols * es, lambda: self.L.tc_dw_ln_fwd(
            _ptr(x.data), x.ld, _ptr(w.data), _ptr(b.data), w.gs if Gn > 1 else 0, _ptr(g.data), _ptr(beta.data), g.gs if Gn > 1 else 0,
            _ptr(t1.data), t1.ld, _ptr(xn.data), xn.ld, _ptr(mean), _ptr(rstd), Gn, B, H, W, x.cols, eps, self.dt, self.stream))
        return t1, xn

    def dwconv(self, x: Var, w: P, b: Optional[P], B: int, H: int, W: int, k: int, stride: int = 1, add_input: bool = False,
               out: Optional[Var] = None, launch: bool = True) -> Var:
        Cc = x.cols
        Gn = self.ngroups                                    # B = images per group
        assert x.rows == Gn * B * H * W and (Gn == 1 or (w.gs > 0 and stride == 1))
        Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
        if out is None:
            out = self.new(Gn * B * Ho * Wo, Cc)
        es = x.data.element_size()
        if launch:
            _timed("hbm:dwconv_fwd", (x.rows + out.rows) * Cc * es, lambda: self.L.tc_dwconv_fwd(
                _ptr(x.data), x.ld, _ptr(w.data), _ptr(b.data) if b is not None else None, _ptr(out.data), out.ld, B, H, W, Cc, k, stride,
                int(add_input), Gn, w.gs, self.dt, self.stream))

        def bwd():
            dy = self.grad_of(out)
            if dy is None:
                return
            if x.requires_grad and w.grad is not None and stride == 1 and _DW_BWD_ONE and not (self.overlap_wgrad and self.use_streams):
                gx, acc = self.wgrad(x)                       # both gradients in one launch (they share dy and nothing else)
                vec = 16 // es
                from ._lib import TcDwFold
                site = TcDwFold()
                nf = (int(self.L.tc_dwconv_bwd_plan(B, H, W, Cc, k, Gn, self.dt, C.byref(site)))
                      if (_DW_DEFER and not self.use_streams and not (x.ld % vec or dy.stride(0) % vec or gx.stride(0) % vec or _ptr(dy) % 16 or _ptr(x.data) % 16 or _ptr(gx) % 16))
                      else 0)
                if nf > 0:                                    # the walkers' sums parked in a buffer of this launch's own; one tc_dw_fold per backward leg
                    part = self.f32(nf)
                    _timed("hbm:dwconv_bwd (input + weight gradient, one launch)", (2.0 * x.rows * (1 + acc) + 2.0 * out.rows) * Cc * es,
                           lambda: self.L.tc_dwconv_bwd(_ptr(dy), dy.stride(0), _ptr(x.data), x.ld, _ptr(w.data), _ptr(gx), gx.stride(0), _ptr(w.grad),
                                                        _ptr(b.grad) if b is not None else None, B, H, W, Cc, k, int(add_input), acc, Gn, w.gs,
                                                        _ptr(part), -4 * nf, self.dt, self.stream))
                    site.part, site.dw, site.db, site.wstride = _ptr(part), _ptr(w.grad), _ptr(b.grad) if b is not None else None, w.gs
                    self._dw_pending.append((site, part))
                    return
                ws = _workspace(self.dev, self.stream)
                _timed("hbm:dwconv_bwd (input + weight gradient, one launch)", (2.0 * x.rows * (1 + acc) + 2.0 * out.rows) * Cc * es,
                       lambda: self.L.tc_dwconv_bwd(_ptr(dy), dy.stride(0), _ptr(x.data), x.ld, _ptr(w.data), _ptr(gx), gx.stride(0), _ptr(w.grad),
                                                    _ptr(b.grad) if b is not None else None, B, H, W, Cc, k, int(add_input), acc, Gn, w.gs,
                                                    ws.data_ptr(), ws.numel(), self.dt, self.stream))
                return
            if x.requires_grad:
                gx, acc = self.wgrad(x)
                _timed("hbm:dwconv_bwd_input", (x.rows * (1 + acc) + out.rows) * Cc * es, lambda: self.L.tc_dwconv_bwd_input(
                    _ptr(dy), dy.stride(0), _ptr(w.data), _ptr(gx), gx.stride(0), B, H, W, Cc, k, stride, int(add_input), acc, Gn, w.gs,
                    self.dt, self.stream))
            if w.grad is not None:
                def dwgrad():
                    ws = _workspace(self.dev, self.stream)           # of the stream this actually runs on
                    _timed("hbm:dwconv_bwd_weight", (x.rows + out.rows) * Cc * es, lambda: self.L.tc_dwconv_bwd_weight(
                        _ptr(dy), dy.stride(0), _ptr(x.data), x.ld, _ptr(w.grad), _ptr(b.grad) if b is not None else None, B, H, W, Cc, k,
                        stride, Gn, w.gs, ws.data_ptr(), ws.numel(), self.dt, self.stream))
                self._weight_grad(dwgrad, reads=dy)
        self._rec(bwd)
        return out

    def dwconv_multi(self, xs: List[Var], ws: List[P], bs: List[Optional[P]], geo, ks: List[int], outs: List[Optional[Var]],
                     add_input: bool = False, launch: bool = True) -> List[Var]:
        """Up to four independent stride-1 depthwise convolutions in one launch each for forward, input gradient and weight
        gradient (tc_dwconv_multi): column slices of one map with different kernel sizes (crpe), or different maps (the per-scale
        MixFFNs of a bridge layer).  geo = (B, H, W) for all, or a list of them per segment; outs[i] None -> a new buffer."""
        n, Gn = len(xs), self.ngroups
        assert 1 <= n <= 4
        geos = [geo] * n if isinstance(geo[0], int) else list(geo)
        gs = ws[0].gs
        outs = [o if o is not None else self.new(x.rows, x.cols) for o, x in zip(outs, xs)]

        def segs(xp, wp, bp, yp, dyp, dwp, dbp, ldx, ldy, lddy):
            arr = (TcDwSeg * n)()
            for i in range(n):
                arr[i] = TcDwSeg(xp[i], wp[i], bp[i], yp[i], dyp[i], dwp[i], dbp[i], xs[i].cols, ks[i], ldx[i], ldy[i], lddy[i], *geos[i])
            return arr
        none, zero = [None] * n, [0] * n
        wd, bd = [_ptr(w.data) for w in ws], [_ptr(b.data) if b is not None else None for b in bs]
        if launch:
            self.L.tc_dwconv_multi(segs([_ptr(x.data) for x in xs], wd, bd, [_ptr(o.data) for o in outs], none, none, none,
                                        [x.ld for x in xs], [o.ld for o in outs], zero), n, 0, int(add_input), 0, Gn, gs, None, 0,
                                   self.dt, self.stream)

        def bwd():
            dys = [self.grad_of(o) for o in outs]
            if any(d is None for d in dys):
                assert all(d is None for d in dys)
                return
            ldd = [d.stride(0) for d in dys]
            if xs[0].requires_grad and ws[0].grad is not None and _DW_BWD_ONE and not (self.overlap_wgrad and self.use_streams):
                g = [self.wgrad(x) for x in xs]                   # input and weight gradients of all segments in one launch
                acc = g[0][1]
                assert all(a == acc for _, a in g)
                sg = segs([_ptr(x.data) for x in xs], wd, none, [_ptr(t) for t, _ in g], [_ptr(d) for d in dys],
                          [_ptr(w.grad) for w in ws], [_ptr(b.grad) if b is not None else None for b in bs],
                          [x.ld for x in xs], [t.stride(0) for t, _ in g], ldd)
                if _DW_DEFER and not self.use_streams:         # the walkers' sums parked; one tc_dw_fold per backward leg (single-stream sweeps only:
                    #                                            a fold issued on one branch stream must not read a sibling branch's unfinished sums)
                    from ._lib import TcDwFold
                    sites, offs = (TcDwFold * n)(), (C.c_longlong * n)()
                    nf = int(self.L.tc_dwconv_multi_plan(sg, n, Gn, self.dt, sites, offs))
                    if nf > 0:
                        part = self.f32(nf)
                        self.L.tc_dwconv_multi(sg, n, 3, int(add_input), acc, Gn, gs, _ptr(part), -4 * nf, self.dt, self.stream)
                        for i in range(n):
                            st = TcDwFold()
                            C.memmove(C.byref(st), C.byref(sites[i]), C.sizeof(TcDwFold))
                            st.part, st.dw, st.wstride = _ptr(part) + 4 * offs[i], _ptr(ws[i].grad), gs
                            st.db = _ptr(bs[i].grad) if bs[i] is not None else None
                            self._dw_pending.append((st, part))
                        return
                wk = _workspace(self.dev, self.stream)
                self.L.tc_dwconv_multi(sg, n, 3, int(add_input), acc, Gn, gs, wk.data_ptr(), wk.numel(), self.dt, self.stream)
                return
            if xs[0].requires_grad:
                g = [self.wgrad(x) for x in xs]
                acc = g[0][1]
                assert all(a == acc for _, a in g)
                self.L.tc_dwconv_multi(segs([_ptr(d) for d in dys], wd, none, [_ptr(t) for t, _ in g], none, none, none, ldd,
                                            [t.stride(0) for t, _ in g], zero), n, 1, int(add_input), acc, Gn, gs, None, 0, self.dt,
                                       self.stream)
            if ws[0].grad is not None:
                def dwgrad():
                    wk = _workspace(self.dev, self.stream)
                    self.L.tc_dwconv_multi(segs([_ptr(x.data) for x in xs], none, none, none, [_ptr(d) for d in dys],
                                                [_ptr(w.grad) for w in ws], [_ptr(b.grad) if b is not None else None for b in bs],
                                                [x.ld for x in xs], zero, ldd), n, 2, 0, 0, Gn, gs, wk.data_ptr(), wk.numel(), self.dt,
                                           self.stream)
                self._weight_grad(dwgrad, reads=dys[0])
        self._rec(bwd)
        return outs

    def batchnorm(self, x: Var, gamma: P, beta: P, running_mean: torch.Tensor, running_var: torch.Tensor, act: int = ACT_NONE,
                  residual: Optional[Var] = None, out: Optional[Var] = None, saved: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
                  launch: bool = True) -> Var:
        """saved = (save_mean, save_rstd) buffers to use; launch=False: a fused kernel of the caller applies this BatchNorm (writes `out`,
        the saved statistics and the running statistics); only the backward closure is recorded."""
        rows, Cc = x.rows, x.cols
        if out is None:
            out = self.new(rows, Cc)
        if self.record and not self.training:
            raise NotImplementedError("backward through BatchNorm in eval mode is not part of the reference path")
        smean = srstd = part = None
        chunks = 0
        if self.training:
            smean, srstd = saved if saved is not None else (self.f32(Cc), self.f32(Cc))
            pre = getattr(x, "bn_part", None)                     # the producing GEMM's epilogue already made the statistics pass
            if pre is not None:
                part, chunks = pre
            else:
                part = self.f32(int(self.L.tc_bn_scratch_floats(rows, Cc)))
        es = x.data.element_size()
        if launch:
            _timed("hbm:batchnorm_fwd", ((3.0 if self.training and not chunks else 2.0) + (residual is not None)) * rows * Cc * es, lambda: self.L.tc_bn_fwd(
                _ptr(x.data), x.ld, _ptr(gamma.data), _ptr(beta.data), _ptr(running_mean), _ptr(running_var),
                _ptr(residual.data) if residual is not None else None, residual.ld if residual is not None else 0, _ptr(out.data), out.ld,
                _ptr(smean), _ptr(srstd), _ptr(part), rows, Cc, 1e-5, 0.1, int(self.training), chunks, act, self.dt, self.stream))

        def bwd():
            dy = self.grad_of(out)
            if dy is None:
                return
            gx, acc = self.wgrad(x)
            _timed("hbm:batchnorm_bwd", (5.0 + acc) * rows * Cc * es, lambda: self.L.tc_bn_bwd(
                _ptr(dy), dy.stride(0), _ptr(x.data), x.ld, _ptr(gamma.data), _ptr(beta.data), _ptr(smean), _ptr(srstd), _ptr(gx),
                gx.stride(0), _ptr(gamma.grad), _ptr(beta.grad), _ptr(part), rows, Cc, act, acc, self.dt, self.stream))
            if residual is not None:
                self.pass_grad(residual, dy)
        self._rec(bwd)
        return out

    def ripm_supported(self, m: Var) -> bool:
        return (_RIPM_FUSED and self.dt != TC_F32 and not self.use_streams and self.ngroups == 1 and m.ld % 8 == 0
                and m.data.data_ptr() % 16 == 0 and bool(self.L.tc_ripm_supported(m.cols, self.dt)))

    def ripm_stage(self, m: Var, steps: List[dict], B: int, side: int, stack: Var) -> int:
        """Patch_Embed_stage (MSTr.py:725-732): three DWConv2d_BN steps -- dw3x3 (stride 2, 1, 1), pw1x1, BatchNorm, Hardswish -- as
        3 + 1 forward launches instead of 9 (tc_ripm_fwd: a step per launch; the BatchNorm + Hardswish of a step is applied by the NEXT step
        on the way in, which also writes the normalised map into `stack`; the last step's by tc_bn_fwd from the sums the kernel left).
        steps[i] = dict(dw=P, pw=P, gamma=P, beta=P, rmean=tensor, rvar=tensor).  The backward is the nine ops' own.  Returns the output side."""
        Cc, L = m.cols, self.L
        xin, prev = m, None
        for i, st in enumerate(steps):
            stride = 2 if i == 0 else 1
            so = (side - 1) // stride + 1
            rows = B * so * so
            y = self.dwconv(xin, st["dw"], None, B, side, side, 3, stride, launch=False)
            z = self.new(rows, Cc)
            T = int(L.tc_ripm_tiles(B, so, so))
            part = self.f32(Cc * (1 + 2 * max(T, 128)))              # (also the scratch of the BatchNorm's backward sums)
            if self.training:
                z.bn_part = (part, T)
            self.linear(y, st["pw"], None, out=z, launch=False)
            saved = (self.f32(Cc), self.f32(Cc))
            self.n_launch += 1
            pp = prev
            _timed("hbm:ripm_fwd (BatchNorm + Hardswish of the input, dw3x3, pw1x1, statistics: one launch)", (2.0 * xin.rows + 2.0 * rows) * Cc * m.data.element_size(),
                   lambda: L.tc_ripm_fwd(_ptr(pp["z"].data if pp else m.data), (pp["z"].ld if pp else m.ld), int(pp is not None),
                                         _ptr(pp["part"]) if pp else None, pp["T"] if pp else 0,
                                         _ptr(pp["st"]["gamma"].data) if pp else None, _ptr(pp["st"]["beta"].data) if pp else None,
                                         _ptr(pp["st"]["rmean"]) if pp else None, _ptr(pp["st"]["rvar"]) if pp else None,
                                         _ptr(pp["saved"][0]) if pp else None, _ptr(pp["saved"][1]) if pp else None, 1e-5, 0.1, int(self.training),
                                         _ptr(xin.data) if pp else None, xin.ld if pp else 0, _ptr(st["dw"].data), _ptr(st["pw"].data), _ptr(y.data), y.ld,
                                         _ptr(z.data), z.ld, _ptr(part), _ptr(st["rmean"]) if self.training else None, B, side, side, Cc, stride, self.dt,
                                         self.stream))
            last = i == len(steps) - 1
            xout = self.batchnorm(z, st["gamma"], st["beta"], st["rmean"], st["rvar"], ACT_HSWISH, out=stack.rowslice(i * rows, (i + 1) * rows),
                                  saved=saved, launch=last)
            prev = dict(z=z, part=part, T=T, st=st, saved=saved)
            xin, side = xout, so
        return side

    def softmax(self, x: Var, nb: int, axis: int, out: Optional[Var] = None) -> Var:
        """Softmax over axis (0 = rows, 1 = cols) of each of the nb stacked [rows/nb, cols] matrices."""
        R, Cc = x.rows // nb, x.cols
        if out is None:
            out = self.new(x.rows, Cc)
        scr = self.f32(int(self.L.tc_softmax_scratch_floats(nb, R, Cc))) if axis == 0 else None
        self.L.tc_softmax_fwd(_ptr(x.data), _ptr(out.data), _ptr(scr), nb, R * x.ld, R * out.ld, R, Cc, x.ld, out.ld, axis, self.dt,
                              self.stream)

        def bwd():
            dy = self.grad_of(out)
            if dy is None:
                return
            gx, acc = self.wgrad(x)
            self.L.tc_softmax_bwd(_ptr(dy), _ptr(out.data), _ptr(gx), _ptr(scr), nb, R * dy.stride(0), R * out.ld, R * gx.stride(0), R,
                                  Cc, dy.stride(0), out.ld, gx.stride(0), axis, acc, self.dt, self.stream)
        self._rec(bwd)
        return out

    def bmm(self, A: Var, B: Var, out: Var, M: int, N: int, K: int, tA: int, tB: int, nb1: int = 1, nb2: int = 1,
            sA=(0, 0), sB=(0, 0), sC=(0, 0), alpha: float = 1.0) -> Var:
        """out[b] = alpha * op(A[b]) op(B[b]); A/B/out are 2-D views whose data pointer is batch (0,0)."""
        assert not (tA and tB)
        # few output tiles with a long reduction (token-reduced context matrices): tc_gemm splits K itself and folds the
        # partials in-kernel through the per-stream workspace
        self._gemm(_ptr(A.data), A.ld, _ptr(B.data), B.ld, _ptr(out.data), out.ld, M, N, K, tA, tB, alpha=alpha, nb1=nb1,
                   nb2=nb2, sA=sA, sB=sB, sC=sC)

        def bwd():
            dC = self.grad_of(out)
            if dC is None:
                return
            ldc = dC.stride(0)
            kw = dict(alpha=alpha, nb1=nb1, nb2=nb2)
            if A.requires_grad:
                gA, acc = self.wgrad(A)
                if not tA and not tB:      # dA[M,K] = dC[M,N] . B[K,N]^T
                    self._gemm(_ptr(dC), ldc, _ptr(B.data), B.ld, _ptr(gA), gA.stride(0), M, K, N, 0, 1, acc=acc, sA=sC, sB=sB,
                               sC=sA, **kw)
                elif not tA and tB:        # B stored [N,K]: dA = dC . B
                    self._gemm(_ptr(dC), ldc, _ptr(B.data), B.ld, _ptr(gA), gA.stride(0), M, K, N, 0, 0, acc=acc, sA=sC, sB=sB,
                               sC=sA, **kw)
                else:                      # A stored [K,M], B stored [K,N]: dA[K,M] = B[K,N] . dC[M,N]^T
                    self._gemm(_ptr(B.data), B.ld, _ptr(dC), ldc, _ptr(gA), gA.stride(0), K, M, N, 0, 1, acc=acc, sA=sB, sB=sC,
                               sC=sA, **kw)
            if B.requires_grad:
                gB, acc = self.wgrad(B)
                if not tA and not tB:      # dB[K,N] = A[M,K]^T . dC[M,N]
                    self._gemm(_ptr(A.data), A.ld, _ptr(dC), ldc, _ptr(gB), gB.stride(0), K, N, M, 1, 0, acc=acc, sA=sA, sB=sC,
                               sC=sB, **kw)
                elif not tA and tB:        # dB[N,K] = dC[M,N]^T . A[M,K]
                    self._gemm(_ptr(dC), ldc, _ptr(A.data), A.ld, _ptr(gB), gB.stride(0), N, K, M, 1, 0, acc=acc, sA=sC, sB=sA,
                               sC=sB, **kw)
                else:                      # dB[K,N] = A_stored[K,M] . dC[M,N]
                    self._gemm(_ptr(A.data), A.ld, _ptr(dC), ldc, _ptr(gB), gB.stride(0), K, N, M, 0, 0, acc=acc, sA=sA, sB=sC,
                               sC=sB, **kw)
        self._rec(bwd)
        return out

    def mhca_att_supported(self, n: Var, N: int, heads: int = 8, windows: Optional[List[Tuple[int, int]]] = None) -> bool:
        """tc_mhca_att_fwd / _bwd hard-code the reference's head layout (8 heads; 3x3 / 5x5 / 7x7 windows over 2 / 3 / 3 of them,
        MSTr.py:1404 crpe_window): anything else takes the op-by-op form."""
        if heads != 8 or (windows is not None and [tuple(w_) for w_ in windows] != [(3, 2), (5, 3), (7, 3)]):
            return False
        return (_MHCA_ATT_FUSED and self.dt != TC_F32 and not self.use_streams and n.ld % 8 == 0 and n.data.data_ptr() % 16 == 0
                and (self.pgs % 8 == 0 or self.ngroups == 1) and bool(self.L.tc_mhca_att_supported(n.cols, N, self.dt)))

    def mhca_attention(self, n: Var, Wqkv: P, bqkv: P, cws: List[P], cbs: List[P], B: int, side: int, heads: int, scale: float,
                       windows: List[Tuple[int, int]]) -> Var:
        """The attention half of an MHCABlock after norm1 -- qkv projection, ConvRelPosEnc over v, factorised attention
        (MSTr.py:852-886, 801-823) -- as ONE forward launch per (image, head) (tc_mhca_att_fwd) instead of three.  q | k | v, crpe(v)
        and the key-softmax statistics are stored as before, and the backward is the three ops' own (their closures are recorded
        without their forward launches)."""
        C_, N = n.cols, side * side
        Ch = C_ // heads
        Bt = B * self.ngroups
        qkv = self.linear(n, Wqkv, bqkv, out=self.new(n.rows, 3 * C_, covered=True), launch=False)
        q, k, v = qkv.colslice(0, C_), qkv.colslice(C_, 2 * C_), qkv.colslice(2 * C_, 3 * C_)
        convv = self.new(n.rows, C_)
        c0, xs, outs, kss = 0, [], [], []
        for ksz, nh in windows:
            w = nh * Ch
            xs.append(v.colslice(c0, c0 + w)); outs.append(convv.colslice(c0, c0 + w)); kss.append(ksz)
            c0 += w
        stats = self.f32(int(self.L.tc_factor_att_stats_floats(Bt, heads, Ch)))
        gs = Wqkv.gs if self.ngroups > 1 else 0
        assert heads == 8 and [tuple(w_) for w_ in windows] == [(3, 2), (5, 3), (7, 3)], "tc_mhca_att_* hard-code this head layout: ask mhca_att_supported first"
        fused_bwd = (_MHCA_ATT_BWD_FUSED and self.record and heads == 8 and [k_ for k_, _ in windows] == [3, 5, 7]
                     and all(w_.grad is not None for w_ in cws) and all(b_ is not None and b_.grad is not None for b_ in cbs)
                     and bool(self.L.tc_mhca_att_bwd_supported(C_, N, self.dt)))
        if fused_bwd:
            # backward of crpe + attention core as ONE launch per (image, head) (tc_mhca_att_bwd): dconvv never leaves the chip
            o = self.new(n.rows, C_)
            Gn = self.ngroups

            def bwd():
                go = self.grad_of(o)
                if go is None:
                    return
                gq, aq = self.wgrad(q)
                gk, ak = self.wgrad(k)
                gv, av = self.wgrad(v)
                assert gq.stride(0) == gk.stride(0) == gv.stride(0) and gk.data_ptr() - gq.data_ptr() == C_ * gq.element_size()
                self.n_launch += 1
                _timed("hbm:mhca_att_bwd (factorised attention + crpe backward, one launch)", (3.0 + 1.0 + 1.0 + 3.0) * n.rows * C_ * n.data.element_size(),
                       lambda: self.L.tc_mhca_att_bwd(_ptr(qkv.data), qkv.ld, _ptr(convv.data), convv.ld, _ptr(go), go.stride(0), _ptr(stats), _ptr(gq),
                                                      gq.stride(0), aq, ak, av, _ptr(cws[0].data), _ptr(cws[1].data), _ptr(cws[2].data),
                                                      _ptr(cws[0].grad), _ptr(cbs[0].grad), _ptr(cws[1].grad), _ptr(cbs[1].grad), _ptr(cws[2].grad),
                                                      _ptr(cbs[2].grad), gs, Gn, B, side, side, C_, scale, self.dt, self.stream))
            self._rec(bwd)
        else:
            self.dwconv_multi(xs, cws, cbs, (B, side, side), kss, outs, launch=False)
            o = self.factor_att_core(q, k, v, convv, Bt, N, heads, scale, stats=stats, launch=False)
        self.n_launch += 1
        _timed("hbm:mhca_att_fwd (qkv projection + crpe + factorised attention, one launch)", (1.0 + 3.0 + 1.0 + 1.0) * n.rows * C_ * n.data.element_size(),
               lambda: self.L.tc_mhca_att_fwd(_ptr(n.data), n.ld, _ptr(Wqkv.data), _ptr(bqkv.data), _ptr(cws[0].data), _ptr(cbs[0].data),
                                              _ptr(cws[1].data), _ptr(cbs[1].data), _ptr(cws[2].data), _ptr(cbs[2].data), gs, _ptr(qkv.data), qkv.ld,
                                              _ptr(convv.data), convv.ld, _ptr(o.data), o.ld, _ptr(stats), self.ngroups, B, side, side, C_, scale,
                                              self.dt, self.stream))
        return o

    def factor_att_core(self, q: Var, k: Var, v: Var, convv: Var, Bt: int, N: int, heads: int, scale: float,
                        stats: Optional[torch.Tensor] = None, launch: bool = True) -> Var:
        """o = scale * q (softmax_N(k)^T v) + q (.) convv per (image, head) in one launch (tc_factor_att_fwd); q/k/v are
        column slices of one qkv buffer."""
        C_ = q.cols
        Ch = C_ // heads
        assert q.ld == k.ld == v.ld and q.rows == Bt * N
        out = self.new(q.rows, C_)
        if stats is None:
            stats = self.f32(int(self.L.tc_factor_att_stats_floats(Bt, heads, Ch)))
        if launch:
            self.L.tc_factor_att_fwd(_ptr(q.data), _ptr(k.data), _ptr(v.data), q.ld, _ptr(convv.data), convv.ld, _ptr(out.data), out.ld,
                                     _ptr(stats), Bt, N, heads, Ch, scale, self.dt, self.stream)

        def bwd():
            go = self.grad_of(out)
            if go is None:
                return
            gq, aq = self.wgrad(q)
            gk, ak = self.wgrad(k)
            gv, av = self.wgrad(v)
            gc, ac = self.wgrad(convv)
            assert not ac and gq.stride(0) == gk.stride(0) == gv.stride(0)
            self.L.tc_factor_att_bwd(_ptr(q.data), _ptr(k.data), _ptr(v.data), q.ld, _ptr(convv.data), convv.ld, _ptr(go), go.stride(0),
                                     _ptr(stats), _ptr(gq), _ptr(gk), _ptr(gv), gq.stride(0), aq, ak, av, _ptr(gc), gc.stride(0), Bt, N,
                                     heads, Ch, scale, self.dt, self.stream)
        self._rec(bwd)
        return out

    def fma3(self, a: Var, b: Var, c: Var, alpha: float, out: Optional[Var] = None) -> Var:
        """out = alpha*a + b*c"""
        rows, Cc = a.rows, a.cols
        if out is None:
            out = self.new(rows, Cc)
        self.L.tc_fma3_fwd(_ptr(a.data), a.ld, _ptr(b.data), b.ld, _ptr(c.data), c.ld, _ptr(out.data), out.ld, rows, Cc, alpha,
                           self.dt, self.stream)

        def bwd():
            d = self.grad_of(out)
            if d is None:
                return
            ga, acca = self.wgrad(a)
            gb, accb = self.wgrad(b)
            gc, accc = self.wgrad(c)
            assert not acca and not accc, "fma3: a and c must be single-use"
            self.L.tc_fma3_bwd(_ptr(d), d.stride(0), _ptr(b.data), b.ld, _ptr(c.data), c.ld, _ptr(ga), ga.stride(0), _ptr(gb),
                               gb.stride(0), accb, _ptr(gc), gc.stride(0), rows, Cc, alpha, self.dt, self.stream)
        self._rec(bwd)
        return out

    def coord_pool(self, x: Var, B: int, H: int, W: int) -> Var:
        Cc = x.cols
        assert x.data.is_contiguous()
        out = self.new(B * (H + W), Cc)
        _timed("hbm:coord_pool_fwd", (x.rows + out.rows) * Cc * x.data.element_size(),
               lambda: self.L.tc_coord_pool_fwd(_ptr(x.data), _ptr(out.data), B, H, W, Cc, self.dt, self.stream))

        def bwd():
            d = self.grad_of(out)
            if d is None:
                return
            gx, acc = self.wgrad(x)
            self.L.tc_coord_pool_bwd(_ptr(d), _ptr(gx), B, H, W, Cc, acc, self.dt, self.stream)
        self._rec(bwd)
        return out

    def coord_gate(self, x: Var, att: Var, B: int, H: int, W: int) -> Var:
        Cc = x.cols
        assert x.data.is_contiguous() and att.data.is_contiguous()
        out = self.new(x.rows, Cc)
        _timed("hbm:coord_gate_fwd", (2 * x.rows + att.rows) * Cc * x.data.element_size(),
               lambda: self.L.tc_coord_gate_fwd(_ptr(x.data), _ptr(att.data), _ptr(out.data), B, H, W, Cc, self.dt, self.stream))

        def bwd():
            d = self.grad_of(out)
            if d is None:
                return
            gx, acc = self.wgrad(x)
            ga, acca = self.wgrad(att)
            assert not acca
            self.L.tc_coord_gate_bwd(_ptr(d), _ptr(x.data), _ptr(att.data), _ptr(gx), acc, _ptr(ga), B, H, W, Cc, self.dt,
                                     self.stream)
        self._rec(bwd)
        return out

    def permuted_weight(self, W: P, nb: int, R: int, Cc: int) -> P:
        """A weight stored [nb, R, Cc] handed out as [nb, Cc * R] with the two inner axes swapped (Conv3d(kernel (4, 1, 1)) over the stacked
        branch maps = a Linear over their concatenation, MSTr.py:441-462: the weight is stored [O][C][path], the concatenation runs
        [path][C]).  The copy is made per pass (it is a few KB); its gradient is swapped back into W.grad."""
        assert W.data.is_contiguous() and W.data.numel() == nb * R * Cc
        data = torch.empty((nb, Cc * R), dtype=W.data.dtype, device=self.dev)
        self.L.tc_transpose(_ptr(W.data), _ptr(data), nb, R, Cc, self.dt, self.stream)
        if W.grad is None or not self.record:
            return P(data, None)
        grad = torch.zeros((nb, Cc * R), dtype=torch.float32, device=self.dev)

        def bwd():
            tmp = torch.empty((nb, R * Cc), dtype=torch.float32, device=self.dev)
            self.L.tc_transpose(_ptr(grad), _ptr(tmp), nb, Cc, R, TC_F32, self.stream)
            g = W.grad.view(nb, R * Cc)
            self.L.tc_add(_ptr(g), g.stride(0), _ptr(tmp), tmp.stride(0), _ptr(g), g.stride(0), nb, R * Cc, TC_F32, self.stream)
        self._rec(bwd)
        return P(data, grad)

    def chan_pool(self, x: Var, B: int, N: int) -> Var:
        """SE_Block squeeze (MSTr.py:586): [B, C] means over each image's N token rows."""
        out = self.new(B, x.cols)
        self.L.tc_chan_pool_fwd(_ptr(x.data), x.ld, _ptr(out.data), B, N, x.cols, self.dt, self.stream)

        def bwd():
            d = self.grad_of(out)
            if d is None:
                return
            assert d.is_contiguous()
            gx, acc = self.wgrad(x)
            self.L.tc_chan_pool_bwd(_ptr(d), _ptr(gx), gx.stride(0), B, N, x.cols, acc, self.dt, self.stream)
        self._rec(bwd)
        return out

    def chan_gate(self, x: Var, gate: Var, B: int, N: int) -> Var:
        """SE_Block gating (MSTr.py:588): out[b, n, :] = x[b, n, :] * gate[b, :]."""
        assert gate.data.is_contiguous() and gate.rows == B and gate.cols == x.cols
        out = self.new(x.rows, x.cols)
        self.L.tc_chan_gate_fwd(_ptr(x.data), x.ld, _ptr(gate.data), _ptr(out.data), out.ld, B, N, x.cols, self.dt, self.stream)

        def bwd():
            d = self.grad_of(out)
            if d is None:
                return
            gx, acc = self.wgrad(x)
            gg, accg = self.wgrad(gate)
            assert not accg and gg.is_contiguous()
            self.L.tc_chan_gate_bwd(_ptr(d), d.stride(0), _ptr(x.data), x.ld, _ptr(gate.data), _ptr(gx), gx.stride(0), acc, _ptr(gg), B, N,
                                    x.cols, self.dt, self.stream)
        self._rec(bwd)
        return out

    def add(self, a: Var, b: Var) -> Var:
        """a + b (same shape)."""
        assert a.rows == b.rows and a.cols == b.cols and a.cols % 4 == 0
        out = self.new(a.rows, a.cols)
        self.L.tc_add(_ptr(a.data), a.ld, _ptr(b.data), b.ld, _ptr(out.data), out.ld, a.rows, a.cols, self.dt, self.stream)

        def bwd():
            d = self.grad_of(out)
            if d is None:
                return
            self.pass_grad(a, d)
            self.pass_grad(b, d)
        self._rec(bwd)
        return out

    def chan_pool2(self, x: Var, B: int, N: int) -> Var:
        """CBAM ChannelAttention pooling (MSTr.py:1141-1142): [2B, C] -- rows 0..B-1 the per-image channel maxima, rows B..2B-1 the means."""
        out = self.new(2 * B, x.cols)
        idx = torch.empty((B, x.cols), dtype=torch.int32, device=self.dev)
        self.L.tc_chan_pool2_fwd(_ptr(x.data), x.ld, _ptr(out.data), _ptr(idx), B, N, x.cols, self.dt, self.stream)

        def bwd():
            d = self.grad_of(out)
            if d is None:
                return
            assert d.is_contiguous()
            gx, acc = self.wgrad(x)
            self.L.tc_chan_pool2_bwd(_ptr(d), _ptr(idx), _ptr(gx), gx.stride(0), B, N, x.cols, acc, self.dt, self.stream)
        self._rec(bwd)
        return out

    def pix_stats(self, x: Var) -> Var:
        """CBAM SpatialAttention statistics (MSTr.py:1156-1158): [rows, 2] = (max over the channels, mean over them) per token."""
        out = self.new(x.rows, 2)
        idx = torch.empty((x.rows,), dtype=torch.int32, device=self.dev)
        self.L.tc_pix_stats_fwd(_ptr(x.data), x.ld, _ptr(out.data), _ptr(idx), x.rows, x.cols, self.dt, self.stream)

        def bwd():
            d = self.grad_of(out)
            if d is None:
                return
            assert d.is_contiguous()
            gx, acc = self.wgrad(x)
            self.L.tc_pix_stats_bwd(_ptr(d), _ptr(idx), _ptr(gx), gx.stride(0), x.rows, x.cols, acc, self.dt, self.stream)
        self._rec(bwd)
        return out

    def sa_conv(self, st: Var, W: P, b: P, B: int, H: int, Wd: int, k: int) -> Var:
        """sigmoid(Conv2d(2 -> 1, k x k, padding k / 2)(st)) over the token grid (MSTr.py:1151, 1161-1163): [rows, 1]."""
        assert st.cols == 2 and st.data.is_contiguous() and st.rows == B * H * Wd
        out = self.new(st.rows, 1)
        self.L.tc_sa_conv_fwd(_ptr(st.data), _ptr(W.data), _ptr(b.data), _ptr(out.data), B, H, Wd, k, self.dt, self.stream)

        def bwd():
            d = self.grad_of(out)
            if d is None:
                return
            assert d.is_contiguous() and W.grad is not None and b.grad is not None
            tmp = torch.empty((st.rows, 2), dtype=st.data.dtype, device=self.dev)
            self.L.tc_sa_conv_bwd(_ptr(d), _ptr(out.data), _ptr(st.data), _ptr(W.data), _ptr(tmp), _ptr(W.grad), _ptr(b.grad), B, H, Wd, k, self.dt, self.stream)
            self.pass_grad(st, tmp)
        self._rec(bwd)
        return out

    def pix_gate(self, x: Var, g: Var) -> Var:
        """out[row, :] = x[row, :] * g[row] (MSTr.py:1206)."""
        assert g.cols == 1 and g.rows == x.rows and g.data.is_contiguous()
        out = self.new(x.rows, x.cols)
        self.L.tc_pix_gate_fwd(_ptr(x.data), x.ld, _ptr(g.data), _ptr(out.data), out.ld, x.rows, x.cols, self.dt, self.stream)

        def bwd():
            d = self.grad_of(out)
            if d is None:
                return
            gx, acc = self.wgrad(x)
            gg, accg = self.wgrad(g)
            assert not accg and gg.is_contiguous()
            self.L.tc_pix_gate_bwd(_ptr(d), d.stride(0), _ptr(x.data), x.ld, _ptr(g.data), _ptr(gx), gx.stride(0), acc, _ptr(gg), x.rows, x.cols,
                                   self.dt, self.stream)
        self._rec(bwd)
        return out

    def cam(self, x: Var, gamma: P, B: int, N: int) -> Var:
        """CAM_Module (MSTr.py:478-509) over the four branch maps side by side in x [B*N, 4C]: gamma * (path attention applied) + x."""
        Cc = x.cols // 4
        out = self.new(x.rows, x.cols)
        att = self.f32(B * Cc * 16)
        g32 = gamma.data.float() if gamma.data.dtype != torch.float32 else gamma.data       # the kernels read gamma as one fp32 scalar
        self.L.tc_cam_att_fwd(_ptr(x.data), x.ld, _ptr(att), B, N, Cc, self.dt, self.stream)
        self.L.tc_cam_apply_fwd(_ptr(x.data), x.ld, _ptr(att), _ptr(g32), _ptr(out.data), out.ld, B, N, Cc, self.dt, self.stream)

        def bwd():
            d = self.grad_of(out)
            if d is None:
                return
            gx, acc = self.wgrad(x)
            att2 = self.f32(B * Cc * 16)
            dg = gamma.grad if gamma.grad is not None else self.f32(1)
            self.L.tc_cam_bwd(_ptr(x.data), x.ld, _ptr(d), d.stride(0), _ptr(att), _ptr(g32), _ptr(att2), _ptr(dg), _ptr(gx), gx.stride(0), acc, B, N, Cc,
                              self.dt, self.stream)
        self._rec(bwd)
        return out

    def gamma_residual(self, a: Var, x: Var, gamma: P) -> Var:
        """gamma * a + x, gamma a one-element parameter (MSTr.py:508, 566)."""
        assert a.rows == x.rows and a.cols == x.cols
        out = self.new(x.rows, x.cols)
        g32 = gamma.data.float() if gamma.data.dtype != torch.float32 else gamma.data
        self.L.tc_gamma_res_fwd(_ptr(a.data), a.ld, _ptr(x.data), x.ld, _ptr(g32), _ptr(out.data), out.ld, x.rows, x.cols, self.dt, self.stream)

        def bwd():
            d = self.grad_of(out)
            if d is None:
                return
            ga, acca = self.wgrad(a)
            assert not acca
            gx, acc = self.wgrad(x)
            dg = gamma.grad if gamma.grad is not None else self.f32(1)
            self.L.tc_gamma_res_bwd(_ptr(d), d.stride(0), _ptr(a.data), a.ld, _ptr(g32), _ptr(ga), ga.stride(0), _ptr(gx), gx.stride(0), acc, _ptr(dg),
                                    x.rows, x.cols, self.dt, self.stream)
        self._rec(bwd)
        return out

    def gelu(self, x: Var) -> Var:
        assert x.data.is_contiguous()
        out = self.new(x.rows, x.cols)
        self.L.tc_gelu_fwd(_ptr(x.data), _ptr(out.data), x.data.numel(), self.dt, self.stream)

        def bwd():
            d = self.grad_of(out)
            if d is None:
                return
            assert d.is_contiguous()
            self._write_or_add(x, lambda g: self.L.tc_gelu_bwd(_ptr(d), _ptr(x.data), _ptr(g), d.numel(), self.dt, self.stream))
        self._rec(bwd)
        return out

    def relu(self, x: Var) -> Var:
        assert x.data.is_contiguous()
        out = self.new(x.rows, x.cols)
        self.L.tc_relu_fwd(_ptr(x.data), _ptr(out.data), x.data.numel(), self.dt, self.stream)

        def bwd():
            d = self.grad_of(out)
            if d is None:
                return
            assert d.is_contiguous()
            self._write_or_add(x, lambda g: self.L.tc_relu_bwd(_ptr(d), _ptr(out.data), _ptr(g), d.numel(), self.dt, self.stream))
        self._rec(bwd)
        return out

    def pixel_shuffle(self, x: Var, B: int, H: int, W: int, p: int) -> Var:
        c = x.cols // (p * p)
        assert x.data.is_contiguous()
        out = self.new(B * H * p * W * p, c)
        _timed("hbm:pixel_shuffle", 2.0 * x.rows * x.cols * x.data.element_size(),
               lambda: self.L.tc_pixel_shuffle(_ptr(x.data), _ptr(out.data), B, H, W, p, c, 0, self.dt, self.stream))

        def bwd():
            d = self.grad_of(out)
            if d is None:
                return
            assert d.is_contiguous()
            self._write_or_add(x, lambda g: self.L.tc_pixel_shuffle(_ptr(d), _ptr(g), B, H, W, p, c, 1, self.dt, self.stream))
        self._rec(bwd)
        return out

    def patchify(self, buf: Var, off: int, sb: int, B: int, H: int, W: int, Cc: int, k: int) -> Var:
        """Gather the k x k patches of the [B,H,W,Cc] map that starts `off` elements into each batch of `buf`."""
        assert buf.data.is_contiguous() and buf.is_whole
        out = self.new(B * (H // k) * (W // k), k * k * Cc)
        base = buf.data.data_ptr() + off * buf.data.element_size()
        self.L.tc_patchify(base, sb, Cc, _ptr(out.data), B, H, W, Cc, k, 0, self.dt, self.stream)

        def bwd():
            d = self.grad_of(out)
            if d is None:
                return
            gt = self._region_grad(buf.root, off, B * sb)
            gbase = gt.data_ptr() + off * gt.element_size()
            self.L.tc_patchify(gbase, sb, Cc, _ptr(d), B, H, W, Cc, k, 2, self.dt, self.stream)
        self._rec(bwd)
        return out

    def _ew_multi(self, segs):
        arr = (TcEwSeg * len(segs))(*segs)
        self.L.tc_ew_multi(arr, len(segs), self.dt, self.stream)

    def patchify_many(self, buf: Var, items) -> List[Var]:
        """patchify of several maps of `buf` in ONE launch (and one launch for their gradients).  items: (off, sb, B, H, W, Cc, k)."""
        assert buf.data.is_contiguous() and buf.is_whole and len(items) <= 4
        es = buf.data.element_size()
        outs = [self.new(B * (H // k) * (W // k), k * k * Cc) for (_, _, B, H, W, Cc, k) in items]
        self._ew_multi([TcEwSeg(EW_PATCHIFY, 0, buf.data.data_ptr() + off * es, _ptr(o.data), sb, 0, Cc, 0, B, H, W, Cc, k, 0)
                        for (off, sb, B, H, W, Cc, k), o in zip(items, outs)])

        def bwd():
            segs = []
            for (off, sb, B, H, W, Cc, k), o in zip(items, outs):
                d = self.grad_of(o)
                if d is None:
                    continue
                gt = self._region_grad(buf.root, off, B * sb)
                segs.append(TcEwSeg(EW_PATCHIFY, 2, gt.data_ptr() + off * gt.element_size(), _ptr(d), sb, 0, Cc, 0, B, H, W, Cc, k, 0))
            if segs:
                self._ew_multi(segs)
        self._rec(bwd)
        return outs

    def sr_gather(self, parts, copy, dst: Var):
        """The de-interleaves of Scale_reduce (parts: (x, dst_off, dst_sb, B, Pn, Cc, mult)) and the stage-4 row copy
        (copy: (src, src_off, src_sb, dst_off, dst_sb, nb, rows, cols)) into `dst` in ONE launch, their gradients in one more."""
        assert dst.data.is_contiguous() and dst.is_whole and len(parts) <= 3
        es = dst.data.element_size()
        src, s_off, s_sb, d_off, d_sb, nb, rows, cols = copy
        assert src.data.is_contiguous() and src.is_whole
        segs = [TcEwSeg(EW_DEINTERLEAVE, 0, _ptr(x.data), dst.data.data_ptr() + off * es, 0, sb, 0, dst.ld, B, Pn, Cc, mult, 0, 0)
                for (x, off, sb, B, Pn, Cc, mult) in parts]
        segs.append(TcEwSeg(EW_COPY, 0, src.data.data_ptr() + s_off * es, dst.data.data_ptr() + d_off * es, s_sb, d_sb, src.ld, dst.ld,
                            nb, rows, cols, 0, 0, 0))
        self._ew_multi(segs)

        def bwd():
            dg = self.grad_of(dst)
            if dg is None:
                return
            segs, late = [], []
            for (x, off, sb, B, Pn, Cc, mult) in parts:
                g, acc = self.wgrad(x)
                if acc:                                          # (not on the model's path: x is the fresh output of a convolution)
                    late.append((x, off, sb, B, Pn, Cc, mult))
                    continue
                segs.append(TcEwSeg(EW_DEINTERLEAVE, 1, _ptr(g), dg.data_ptr() + off * es, 0, sb, 0, dst.ld, B, Pn, Cc, mult, 0, 0))
            gt = self._region_grad(src.root, s_off, nb * s_sb)
            segs.append(TcEwSeg(EW_COPY, 1, dg.data_ptr() + d_off * es, gt.data_ptr() + s_off * es, d_sb, s_sb, dst.ld, src.ld,
                                nb, rows, cols, 0, 0, 0))
            self._ew_multi(segs)
            for (x, off, sb, B, Pn, Cc, mult) in late:
                self._write_or_add(x, lambda g_: self.L.tc_sr_deinterleave(_ptr(g_), dg.data_ptr() + off * es, sb, dst.ld, B, Pn, Cc, mult,
                                                                            1, self.dt, self.stream))
        self._rec(bwd)

    def copy_rows(self, src: Var, src_off: int, src_sb: int, dst: Var, dst_off: int, dst_sb: int, nb: int, rows: int, cols: int):
        """dst[b, r, :] = src[b, r, :] for batch-strided row blocks (offsets/strides in elements of the contiguous roots)."""
        assert src.data.is_contiguous() and dst.data.is_contiguous() and src.is_whole and dst.is_whole
        es = src.data.element_size()
        self.L.tc_copy3d(src.data.data_ptr() + src_off * es, src_sb, src.ld, dst.data.data_ptr() + dst_off * es, dst_sb, dst.ld,
                         nb, rows, cols, 0, self.dt, self.stream)

        def bwd():
            dg = self.grad_of(dst)
            if dg is None:
                return
            gt = self._region_grad(src.root, src_off, nb * src_sb)
            self.L.tc_copy3d(dg.data_ptr() + dst_off * es, dst_sb, dst.ld, gt.data_ptr() + src_off * es, src_sb, src.ld,
                             nb, rows, cols, 1, self.dt, self.stream)
        self._rec(bwd)

    def sr_deinterleave(self, x: Var, dst: Var, dst_off: int, dst_sb: int, B: int, Pn: int, Cc: int, mult: int):
        """dst[b, g*Pn+pos, c] = x[b*Pn+pos, c*mult+g] written at element offset dst_off of each dst batch."""
        assert x.data.is_contiguous() and dst.data.is_contiguous()
        es = x.data.element_size()
        self.L.tc_sr_deinterleave(_ptr(x.data), dst.data.data_ptr() + dst_off * es, dst_sb, dst.ld, B, Pn, Cc, mult, 0, self.dt,
                                  self.stream)

        def bwd():
            dg = self.grad_of(dst)
            if dg is None:
                return
            self._write_or_add(x, lambda g: self.L.tc_sr_deinterleave(_ptr(g), dg.data_ptr() + dst_off * es, dst_sb, dst.ld, B,
                                                                      Pn, Cc, mult, 1, self.dt, self.stream))
        self._rec(bwd)

    def transpose(self, x: Var, nb: int) -> Var:
        """[nb, R, C] -> [nb, C, R]"""
        R, Cc = x.rows // nb, x.cols
        assert x.data.is_contiguous()
        out = self.new(nb * Cc, R)
        self.L.tc_transpose(_ptr(x.data), _ptr(out.data), nb, R, Cc, self.dt, self.stream)

        def bwd():
            d = self.grad_of(out)
            if d is None:
                return
            assert d.is_contiguous()
            self._write_or_add(x, lambda g: self.L.tc_transpose(_ptr(d), _ptr(g), nb, Cc, R, self.dt, self.stream))
        self._rec(bwd)
        return out

    def stem_im2col(self, img: torch.Tensor, B: int, in_ch: int, H: int, W: int) -> Var:
        """7x7 s4 p3 patches of the NCHW image -> [B*(H/4)*(W/4), 148] (147 real columns); no gradient (network input)."""
        Ho, Wo = (H - 1) // 4 + 1, (W - 1) // 4 + 1
        out = self.new(B * Ho * Wo, 148, requires_grad=False)
        self.L.tc_stem_im2col(_ptr(img), _ptr(out.data), 148, B, in_ch, H, W, self.dt, self.stream)
        return out

    def window_rows(self, src: Var, dst: Var, B: int, H: int, W: int, ws: int, ntw: int, off: int, to_map: bool):
        """SpatialAwareTrans' window partition (to_map False: dst = window-token matrix [B*(H/ws)*(W/ws)*ntw, C] <- src = token map [B*H*W, C]) or
        its reverse (to_map True: dst = map <- src = windows): a row permutation (tc_window_rows); the gradient takes the same road back."""
        Cc = src.cols
        assert dst.cols == Cc and Cc % 8 == 0
        self.n_launch += 1
        self.L.tc_window_rows(_ptr(src.data), src.ld, _ptr(dst.data), dst.ld, B, H, W, ws, ntw, off, Cc, int(to_map), 0, self.dt, self.stream)

        def bwd():
            d = self.grad_of(dst)
            if d is None or not src.requires_grad:
                return
            if to_map:
                # the reverse step reads ONE scale's tokens of every window: its gradient covers those rows of the window matrix only (the four
                # scales' launches together cover it) -- a zero-filled buffer, always added to
                r = src.root
                assert src.is_whole
                if r.grad_t is None:
                    r.grad_t = self._zeros.zeros_like(r.data)
                r.whole_written = True
                g, acc = src.apply_path(r.grad_t), 1
            else:
                g, acc = self.wgrad(src)
            self.n_launch += 1
            self.L.tc_window_rows(_ptr(d), d.stride(0), _ptr(g), g.stride(0), B, H, W, ws, ntw, off, Cc, int(not to_map), acc, self.dt, self.stream)
        self._rec(bwd)

    def dropout(self, x: Var, p: float, seed: torch.Tensor, salt: int) -> Var:
        """nn.Dropout(p) in training mode (identity otherwise): the keep mask comes from a counter-based generator keyed by (*seed + salt, element);
        the backward applies the same mask to the gradient (tc_dropout)."""
        if not self.training or p <= 0.0:
            return x
        assert x.data.is_contiguous()
        out = self.new(x.rows, x.cols)
        n = x.rows * x.cols
        self.n_launch += 1
        self.L.tc_dropout(_ptr(x.data), _ptr(out.data), n, float(p), _ptr(seed), salt, self.dt, self.stream)

        def bwd():
            d = self.grad_of(out)
            if d is None or not x.requires_grad:
                return
            assert d.is_contiguous()
            tmp = torch.empty_like(d)
            self.L.tc_dropout(_ptr(d), _ptr(tmp), n, float(p), _ptr(seed), salt, self.dt, self.stream)
            self.pass_grad(x, tmp)
        self._rec(bwd)
        return out

    def im2col3s2(self, x, B: int, Cin: int, H: int, W: int, src_ch: int = 0) -> Var:
        """Patches of a 3x3 stride-2 pad-1 convolution (the Conv2d_BN stem of MSViT_4Stages): [B*Ho*Wo, 9*Cin] (row pitch rounded up to 8), column
        c*9 + ky*3 + kx -- the rows of the [Cout, Cin, 3, 3] weight.  x: a token-major Var [B*H*W, Cin] (its gradient comes back through
        tc_col2im3s2), or -- src_ch 1 | Cin -- the NCHW network input (a tensor: no gradient)."""
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        image = not isinstance(x, Var)
        ldc = (9 * Cin + 7) // 8 * 8
        buf = self.new(B * Ho * Wo, ldc, requires_grad=not image)
        self.n_launch += 1
        self.L.tc_im2col3s2(_ptr(x) if image else _ptr(x.data), 0 if image else x.ld, int(image), src_ch if image else 0, _ptr(buf.data), ldc, B, Cin, H, W,
                            self.dt, self.stream)
        cols = buf.colslice(0, 9 * Cin)
        if image:
            return cols

        def bwd():
            d = self.grad_of(cols)
            if d is None or not x.requires_grad:
                return
            gx, acc = self.wgrad(x)
            self.n_launch += 1
            self.L.tc_col2im3s2(_ptr(d), d.stride(0), _ptr(gx), gx.stride(0), B, Cin, H, W, acc, self.dt, self.stream)
        self._rec(bwd)
        return cols

    def attention(self, q: Var, k: Var, v: Var, B: int, Nq: int, Nk: int, scale: float, out: Optional[Var] = None) -> Var:
        """softmax(q k^T * scale) v per batch (single head, d = q.cols)."""
        d = q.cols
        if self.use_fused_attention:
            return self._attention_fused(q, k, v, B, Nq, Nk, scale, out)
        S = self.new(B * Nq, Nk)
        self.bmm(q, k, S, Nq, Nk, d, 0, 1, nb1=B, sA=(Nq * q.ld, 0), sB=(Nk * k.ld, 0), sC=(Nq * Nk, 0), alpha=scale)
        Pm = self.softmax(S, B, 1)
        if out is None:
            out = self.new(B * Nq, d)
        self.bmm(Pm, v, out, Nq, d, Nk, 0, 0, nb1=B, sA=(Nq * Nk, 0), sB=(Nk * v.ld, 0), sC=(Nq * out.ld, 0))
        return out

    use_fused_attention = False

    def attention_seg(self, q: Var, k: Var, v: Var, B: int, nq: List[int], Nk: int, scale: float, out: Optional[Var] = None,
                      q_prescaled: bool = False) -> Var:
        """softmax(q k^T * scale) v where q / out are stage-major: segment i = B images x nq[i] query rows, K/V image-major.
        One launch for all segments on the bf16 path.  q_prescaled: q holds (projection output) * scale * log2(e) (linear(...,
        post_scale=...)); the gradient written for q is then the one of the unscaled projection output."""
        assert not (q_prescaled and self.dtype == torch.float32)
        rows = B * sum(nq)
        assert q.rows == rows and q.cols == 64 and q.data.is_contiguous()
        if out is None:
            out = self.new(rows, 64)
        assert out.data.is_contiguous()
        lse = self.f32(rows)
        nq_c = (C.c_int * len(nq))(*nq)
        flops = 4.0 * rows * Nk * 64
        _timed("attn_fwd", flops, lambda: self.L.tc_attn_fwd_seg(
            _ptr(q.data), q.ld, _ptr(k.data), k.ld, _ptr(v.data), v.ld, Nk * k.ld, _ptr(out.data), out.ld, _ptr(lse), B, len(nq),
            nq_c, Nk, scale, int(q_prescaled), self.dt, self.stream))

        def bwd():
            dO = self.grad_of(out)
            if dO is None:
                return
            assert dO.is_contiguous()
            gq, aq = self.wgrad(q)
            gk, ak = self.wgrad(k)
            gv, av = self.wgrad(v)
            assert not (aq or ak or av), "attention_seg expects single-use q / k / v"
            delta = self.f32(rows)
            dkv32 = self.f32(ATTN_DKV_SPLITS * B * Nk * 128) if self.dtype != torch.float32 else None
            _timed("attn_bwd", 10.0 * rows * Nk * 64, lambda: self.L.tc_attn_bwd_seg(
                _ptr(q.data), q.ld, _ptr(k.data), k.ld, _ptr(v.data), v.ld, Nk * k.ld, _ptr(out.data), out.ld, _ptr(dO),
                dO.stride(0), _ptr(lse), _ptr(delta), _ptr(dkv32), _ptr(gq), gq.stride(0), _ptr(gk), gk.stride(0), _ptr(gv),
                gv.stride(0), Nk * gk.stride(0), B, len(nq), nq_c, Nk, scale, int(q_prescaled), self.dt, self.stream))
        self._rec(bwd)
        return out

    def _attention_fused(self, q: Var, k: Var, v: Var, B: int, Nq: int, Nk: int, scale: float, out: Optional[Var] = None) -> Var:
        d = q.cols
        assert d == 64
        if out is None:
            out = self.new(B * Nq, d)
        lse = self.f32(B * Nq)
        _timed("attn_fwd", 4.0 * B * Nq * Nk * d, lambda: self.L.tc_attn_fwd(
            _ptr(q.data), q.ld, Nq * q.ld, _ptr(k.data), k.ld, _ptr(v.data), v.ld, Nk * k.ld, _ptr(out.data), out.ld, Nq * out.ld,
            _ptr(lse), B, Nq, Nk, scale, self.dt, self.stream))

        def bwd():
            dO = self.grad_of(out)
            if dO is None:
                return
            gq, aq = self.wgrad(q)
            gk, ak = self.wgrad(k)
            gv, av = self.wgrad(v)
            assert not aq and ak == av, "fused attention: q single-use, k/v written together"
            delta = self.f32(B * Nq)
            _timed("attn_bwd", 10.0 * B * Nq * Nk * d, lambda: self.L.tc_attn_bwd(
                _ptr(q.data), q.ld, Nq * q.ld, _ptr(k.data), k.ld, _ptr(v.data), v.ld, Nk * k.ld, _ptr(out.data), out.ld,
                Nq * out.ld, _ptr(dO), dO.stride(0), Nq * dO.stride(0), _ptr(lse), _ptr(delta), _ptr(gq), gq.stride(0),
                Nq * gq.stride(0), _ptr(gk), gk.stride(0), _ptr(gv), gv.stride(0), Nk * gk.stride(0), ak, B, Nq, Nk, scale,
                self.dt, self.stream))
        self._rec(bwd)
        return out
