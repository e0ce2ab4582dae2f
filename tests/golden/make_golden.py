"""Generates the committed golden fixtures from the imported reference (build container only).

    python tests/golden/make_golden.py

The reference (/root/reference, PyTorch, networks/MSTr.py::MSTransception) has no golden vectors
of its own, so the pins are outputs of the reference itself, run here on CPU
fp32 with the name-seeded weights of transception_amd/seeded_init.py loaded strict=True:

  model_b2.npz     whole model, B=2, train mode: sampled stage activations, sampled logits, the packed
                   argmax mask, the top-2 margin, loss / ce / dice, sampled parameter gradients,
                   updated BatchNorm running statistics; eval-mode logits sample as well
  train_trace.npz  six SGD steps (lr 0.05, m 0.9, wd 1e-4, cosine T_max=100): loss, ce, dice, lr,
                   grad norm, post-step parameter checksums
  modules.npz      per-module forward outputs and input gradients for the sub-modules listed in
                   SURVEY.md section 8(c), driven by seeded inputs / upstream gradients
  variants.npz     the constructor's ablation switches that the build implements: concat="normal",
                   have_bridge="None", a non-default br_ch_att_list -- B=1 train-mode logits, loss, gradient probes, eval
                   logits and the digest of each variant's state_dict schema

Fixtures hold data only (inputs are re-derived from seeds; outputs are sampled at seeded positions
plus float64 checksums), never reference source.
"""
import math
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from ref_shim import import_reference  # noqa: E402
from transception_amd.seeded_init import (seeded_input, seeded_labels, seeded_state_dict,  # noqa: E402
                                           seeded_tensor, _stream, schema_entries, schema_digest)

NSAMP = 2048


def sample_idx(tag: str, numel: int, n: int = NSAMP) -> np.ndarray:
    g = _stream(f"sample:{tag}", 3)
    return g.integers(0, numel, size=min(n, numel), dtype=np.int64)


def pack(store: dict, tag: str, t: torch.Tensor):
    a = t.detach().contiguous().float().reshape(-1).numpy()
    idx = sample_idx(tag, a.size)
    store[tag + "/shape"] = np.array(t.shape, dtype=np.int64)
    store[tag + "/samples"] = a[idx].astype(np.float32)
    store[tag + "/sum"] = np.array([a.astype(np.float64).sum(), np.abs(a.astype(np.float64)).sum()])


GRAD_PROBES = [
    "backbone.patch_embed1.proj.weight",
    "backbone.block1.0.attn.keys.weight",
    "backbone.block1.1.mlp.dwconv.dwconv.weight",
    "backbone.patch_embed_stage2.patch_embeds.0.patch_conv.bn.weight",
    "backbone.mhca_stage2.mhca_blks.0.cpe.proj.weight",
    "backbone.mhca_stage2.mhca_blks.1.crpe.conv_list.2.weight",
    "backbone.mhca_stage3.mhca_blks.2.MHCA_layers.5.factoratt_crpe.qkv.weight",
    "backbone.mhca_stage3.InvRes.conv2.bn.bias",
    "backbone.mhca_stage4.aggregate.conv_h.weight",
    "backbone.mhca_stage4.aggregate.bn1.weight",
    "bridge.bridge_layer1.attn.q.weight",
    "bridge.bridge_layer2.attn.scale_reduce.sr1.weight",
    "bridge.bridge_layer3.attn.kv.weight",
    "bridge.bridge_layer4.mixffn3.fc2.weight",
    "decoder_3.layer_up.expand.weight",
    "decoder_2.concat_linear.weight",
    "decoder_1.layer_former_2.mlp.norm1.weight",
    "decoder_0.layer_up.norm.bias",
    "decoder_0.last_layer.weight",
    "decoder_0.last_layer.bias",
]


def whole_model(MST, Dice):
    out = {}
    sd = seeded_state_dict()
    ref = MST(num_classes=9)
    ref.load_state_dict(sd, strict=True)
    ref.train()
    x = torch.from_numpy(seeded_input(2))
    y_lab = torch.from_numpy(seeded_labels(2))
    taps = {}
    ref.backbone.patch_embed1.register_forward_hook(lambda m, i, o: taps.update(patch_embed1=o[0]))
    def enc_hook(m, i, o):
        for k, t in enumerate(o):
            taps[f"enc{k}"] = t.permute(0, 2, 3, 1)

    ref.backbone.register_forward_hook(enc_hook)
    for k in range(4):
        getattr(ref.bridge, f"bridge_layer{k + 1}").register_forward_hook(
            lambda m, i, o, k=k: taps.update({f"bridge{k + 1}": o}))
    ref.decoder_1.register_forward_hook(lambda m, i, o: taps.update(dec1=o))
    logits = ref(x)
    for k, v in taps.items():
        pack(out, "tap/" + k, v)
    pack(out, "logits", logits)
    lg = logits.detach()
    out["argmax"] = lg.argmax(1).numpy().astype(np.uint8)
    top2 = lg.topk(2, dim=1).values
    out["margin_f16"] = (top2[:, 0] - top2[:, 1]).numpy().astype(np.float16)
    ce = torch.nn.functional.cross_entropy(logits, y_lab)
    dice = Dice(9)(logits, y_lab, softmax=True)
    loss = 0.4 * ce + 0.6 * dice
    loss.backward()
    out["loss"] = np.array([loss.item(), ce.item(), dice.item()], dtype=np.float64)
    named = dict(ref.named_parameters())
    sq = 0.0
    for n, p in named.items():
        if p.grad is not None:
            sq += float((p.grad.double() ** 2).sum())
    out["grad_norm"] = np.array([math.sqrt(sq)])
    for n in GRAD_PROBES:
        pack(out, "grad/" + n, named[n].grad)
    rsd = ref.state_dict()
    for k in ("backbone.patch_embed_stage2.patch_embeds.0.patch_conv.bn",
              "backbone.mhca_stage3.InvRes.norm", "backbone.mhca_stage4.aggregate.bn1"):
        out["bn/" + k + ".running_mean"] = rsd[k + ".running_mean"].numpy()
        out["bn/" + k + ".running_var"] = rsd[k + ".running_var"].numpy()
        out["bn/" + k + ".num_batches_tracked"] = rsd[k + ".num_batches_tracked"].numpy()
    # eval mode, fresh buffers
    ref2 = MST(num_classes=9)
    ref2.load_state_dict(sd, strict=True)
    ref2.eval()
    with torch.no_grad():
        le = ref2(x)
    pack(out, "logits_eval", le)
    out["argmax_eval"] = le.argmax(1).numpy().astype(np.uint8)
    t2 = le.topk(2, dim=1).values
    out["margin_eval_f16"] = (t2[:, 0] - t2[:, 1]).numpy().astype(np.float16)
    # 3-channel input path
    x3 = torch.from_numpy(seeded_input(1, in_ch=3))
    with torch.no_grad():
        pack(out, "logits_eval_rgb", ref2(x3))
    np.savez_compressed(os.path.join(HERE, "model_b2.npz"), **out)
    print("model_b2: loss", out["loss"], "grad_norm", out["grad_norm"])


TRACE_PROBES = ["backbone.patch_embed1.proj.weight", "backbone.mhca_stage2.mhca_blks.0.MHCA_layers.0.mlp.fc1.weight",
                "backbone.mhca_stage3.aggregate.conv1.weight", "bridge.bridge_layer2.attn.q.weight",
                "bridge.bridge_layer4.mixffn4.fc2.bias", "decoder_2.layer_former_1.attn.values.weight",
                "decoder_0.last_layer.weight", "backbone.mhca_stage4.InvRes.conv1.bn.weight"]


TRACE_STEPS = 6          # rounds 1-3 pinned two steps; six show that the trajectories stay together once momentum and the running statistics feed back


def train_trace(MST, Dice):
    out = {}
    ref = MST(num_classes=9)
    ref.load_state_dict(seeded_state_dict(), strict=True)
    ref.train()
    opt = torch.optim.SGD(ref.parameters(), lr=0.05, momentum=0.9, weight_decay=1e-4)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=100)
    dice_fn = Dice(9)
    rows = []
    for step in range(TRACE_STEPS):
        x = torch.from_numpy(seeded_input(2, seed=7 + step))
        lab = torch.from_numpy(seeded_labels(2, seed=7 + step))
        logits = ref(x)
        ce = torch.nn.functional.cross_entropy(logits, lab)
        dice = dice_fn(logits, lab, softmax=True)
        loss = 0.4 * ce + 0.6 * dice
        opt.zero_grad()
        loss.backward()
        gn = math.sqrt(sum(float((p.grad.double() ** 2).sum()) for p in ref.parameters() if p.grad is not None))
        opt.step()
        sched.step()
        rows.append([loss.item(), ce.item(), dice.item(), opt.param_groups[0]["lr"], gn])
        named = dict(ref.named_parameters())
        for n in TRACE_PROBES:
            a = named[n].detach().double()
            out[f"step{step}/{n}"] = np.array([a.sum().item(), a.abs().sum().item()])
    out["trace"] = np.array(rows, dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "train_trace.npz"), **out)
    print("train_trace:\n", out["trace"])


def modules(MST):
    """Per-module goldens: y = m(x); gx = d(sum(y*g))/dx and gw/<parameter> = d(sum(y*g))/dw for every parameter of the module,
    on seeded x, g (B=2)."""
    out = {}
    ref = MST(num_classes=9)
    ref.load_state_dict(seeded_state_dict(), strict=True)
    ref.train()
    bb, br = ref.backbone, ref.bridge

    def run(tag, fn, shapes, scale=1.0):
        xs = [torch.from_numpy(seeded_tensor(f"{tag}/x{i}", s, scale)).requires_grad_(True) for i, s in enumerate(shapes)]
        ref.zero_grad(set_to_none=True)
        y = fn(*xs)
        g = torch.from_numpy(seeded_tensor(f"{tag}/g", tuple(y.shape)))
        (y * g).sum().backward()
        pack(out, f"{tag}/y", y)
        for i, xi in enumerate(xs):
            pack(out, f"{tag}/gx{i}", xi.grad)
        nw = 0
        for name, p in ref.named_parameters():       # a sampled grad_weight per kernel golden -- every parameter the module touched
            if p.grad is not None:
                pack(out, f"{tag}/gw/{name}", p.grad)
                nw += 1
        print(tag, tuple(y.shape), f"{nw} weight gradients")

    B = 2
    run("patch_embed1", lambda x: bb.patch_embed1(x)[0], [(B, 3, 224, 224)])
    run("eff_attn_s1", lambda x: bb.block1[0].attn(x), [(B, 64, 56, 56)])
    run("eff_attn_d2", lambda x: ref.decoder_2.layer_former_1.attn(x), [(B, 320, 14, 14)])
    run("eff_block_s1", lambda x: bb.block1[1](x, 56, 56), [(B, 3136, 64)])
    run("mixffn_s2", lambda x: bb.mhca_stage2.mhca_blks[0].MHCA_layers[0].mlp(x, 28, 28), [(B, 784, 64)])
    run("mixffn_b4", lambda x: br.bridge_layer2.mixffn4(x, 7, 7), [(B, 49, 512)])
    run("ripm_s2", lambda x: torch.cat(bb.patch_embed_stage2(x), 1), [(B, 64, 56, 56)])
    run("resblock_s3", lambda x: bb.mhca_stage3.InvRes(x), [(B, 128, 14, 14)])
    for s, (C, hw) in zip((2, 3, 4), ((64, 28), (128, 14), (320, 7))):
        st = getattr(bb, f"mhca_stage{s}")
        run(f"mhca_block_s{s}", lambda x, st=st, hw=hw: st.mhca_blks[1].MHCA_layers[1](x, (hw, hw)), [(B, hw * hw, C)])
        run(f"factoratt_s{s}", lambda x, st=st, hw=hw: st.mhca_blks[2].MHCA_layers[0].factoratt_crpe(x, (hw, hw)),
            [(B, hw * hw, C)])
        run(f"coordatt_s{s}", lambda x, st=st: st.aggregate(x), [(B, 4 * C, hw, hw)])
    run("chan_att", lambda x: br.bridge_layer1.attn(x), [(B, 6076, 64)])
    run("scale_reduce", lambda x: br.bridge_layer2.attn.scale_reduce(x), [(B, 6076, 64)])
    run("self_att", lambda x: br.bridge_layer3.attn(x), [(B, 6076, 64)])
    run("bridge_layer1", lambda x: br.bridge_layer1(x), [(B, 6076, 64)])
    run("bridge_layer4", lambda x: br.bridge_layer4(x), [(B, 6076, 64)])
    run("dec3", lambda x: ref.decoder_3(x), [(B, 49, 512)])
    run("dec2", lambda a, b: ref.decoder_2(a, b), [(B, 196, 256), (B, 14, 14, 320)])
    run("dec0", lambda a, b: ref.decoder_0(a, b), [(B, 3136, 64), (B, 56, 56, 64)])
    np.savez_compressed(os.path.join(HERE, "modules.npz"), **out)


VARIANTS = {   # name -> (constructor kwargs, gradient probes)
    "concat_normal": (dict(concat="normal"),
                      ["backbone.patch_embed1.proj.weight", "backbone.mhca_stage2.aggregate.conv.weight",
                       "backbone.mhca_stage4.aggregate.bn.weight", "bridge.bridge_layer3.attn.kv.weight", "decoder_0.last_layer.weight"]),
    "no_bridge": (dict(have_bridge="None"),
                  ["backbone.patch_embed1.proj.weight", "backbone.mhca_stage3.aggregate.conv_w.weight",
                   "decoder_2.concat_linear.weight", "decoder_0.last_layer.bias"]),
    "ch_att_1101": (dict(br_ch_att_list=[True, True, False, True]),
                    ["backbone.mhca_stage4.aggregate.conv_h.weight", "bridge.bridge_layer2.attn.k.weight",
                     "bridge.bridge_layer3.attn.scale_reduce.sr0.weight", "bridge.bridge_layer4.attn.proj.weight",
                     "bridge.bridge_layer4.mixffn2.fc1.weight", "decoder_0.last_layer.weight"]),
    "concat_se": (dict(concat="se"),
                  ["backbone.mhca_stage2.aggregate.excitation.0.weight", "backbone.mhca_stage3.aggregate.excitation.2.weight",
                   "backbone.mhca_stage4.aggregate.conv.weight", "backbone.mhca_stage4.aggregate.conv.bias", "backbone.mhca_stage3.aggregate.bn.weight",
                   "backbone.mhca_stage2.mhca_blks.1.MHCA_layers.0.mlp.fc1.weight", "decoder_0.last_layer.weight"]),
    "concat_3d": (dict(concat="3d"),
                  ["backbone.mhca_stage2.aggregate.interact_concat.0.weight", "backbone.mhca_stage3.aggregate.interact_concat.0.bias",
                   "backbone.mhca_stage4.aggregate.interact_concat.0.weight", "backbone.mhca_stage4.aggregate.bn.weight",
                   "backbone.mhca_stage3.mhca_blks.2.MHCA_layers.1.factoratt_crpe.qkv.weight", "decoder_0.last_layer.weight"]),
    "concat_skn": (dict(concat="skn"),
                   ["backbone.mhca_stage2.aggregate.fc.weight", "backbone.mhca_stage2.aggregate.fc.bias", "backbone.mhca_stage3.aggregate.fcs.0.weight",
                    "backbone.mhca_stage3.aggregate.fcs.3.bias", "backbone.mhca_stage4.aggregate.conv_bn_ac.0.weight",
                    "backbone.mhca_stage4.aggregate.conv_bn_ac.0.bias", "backbone.mhca_stage2.aggregate.conv_bn_ac.2.weight", "decoder_0.last_layer.weight"]),
    "concat_cbam": (dict(concat="cbam"),
                    ["backbone.mhca_stage2.aggregate.ca.se.0.weight", "backbone.mhca_stage3.aggregate.ca.se.2.weight", "backbone.mhca_stage2.aggregate.sa.conv.weight",
                     "backbone.mhca_stage3.aggregate.sa.conv.bias", "backbone.mhca_stage4.aggregate.conv2d_bn_act.0.weight",
                     "backbone.mhca_stage2.aggregate.conv2d_bn_act.1.weight", "backbone.mhca_stage2.mhca_blks.0.MHCA_layers.2.mlp.fc2.weight",
                     "decoder_0.last_layer.weight"]),
    "concat_cbam_sa4_k3": (dict(concat="cbam", use_sa_config=4, sa_ker=3),
                           ["backbone.mhca_stage4.aggregate.sa.conv.weight", "backbone.mhca_stage4.aggregate.ca.se.0.weight", "decoder_0.last_layer.weight"]),
    "concat_cam": (dict(concat="cam"),
                   ["backbone.mhca_stage2.aggregate.channelAttention.gamma", "backbone.mhca_stage3.aggregate.channelAttention.gamma",
                    "backbone.mhca_stage2.aggregate.bn3d.weight", "backbone.mhca_stage4.aggregate.bn3d.bias", "backbone.mhca_stage3.aggregate.interact_concat.0.weight",
                    "backbone.mhca_stage2.aggregate.bn.weight", "backbone.mhca_stage2.mhca_blks.1.MHCA_layers.2.mlp.fc1.weight", "decoder_0.last_layer.weight"]),
    "concat_cam_fact": (dict(concat="cam_fact"),
                        ["backbone.mhca_stage2.aggregate.channelAttention.gamma", "backbone.mhca_stage3.aggregate.channelAttention.qkv.weight",
                         "backbone.mhca_stage4.aggregate.channelAttention.qkv.bias", "backbone.mhca_stage2.aggregate.channelAttention.proj.weight",
                         "backbone.mhca_stage3.aggregate.bn3d.weight", "backbone.mhca_stage4.aggregate.interact_concat.0.weight", "decoder_0.last_layer.weight"]),
    # Stage_3or4 = 5 builds MSViT_casa: "coord" falls through to the factorized path attention; "cbam" is CBAMBlock_casa (inter)
    "stage5_coord": (dict(Stage_3or4=5),
                     ["backbone.mhca_stage2.aggregate.channelAttention.gamma", "backbone.mhca_stage4.aggregate.channelAttention.qkv.weight",
                      "backbone.mhca_stage3.aggregate.bn3d.weight", "backbone.mhca_stage2.aggregate.interact_concat.0.weight",
                      "backbone.mhca_stage3.InvRes.conv2.conv.weight", "decoder_0.last_layer.weight"]),
    "stage5_cbam_res": (dict(Stage_3or4=5, concat="cbam", inter="res"),
                        ["backbone.mhca_stage2.aggregate.sa.conv.weight", "backbone.mhca_stage3.aggregate.sa.conv.bias", "backbone.mhca_stage3.aggregate.ca.se.0.weight",
                         "backbone.mhca_stage2.InvRes.conv2.conv.weight", "backbone.mhca_stage2.aggregate.conv2d_bn_act.1.weight", "decoder_0.last_layer.weight"]),
    "token_mlp_mix": (dict(token_mlp_mode="mix"),
                      ["backbone.block1.0.mlp.fc1.weight", "backbone.block1.1.mlp.dwconv.dwconv.weight", "backbone.block1.1.mlp.dwconv.dwconv.bias",
                       "decoder_2.layer_former_1.mlp.fc2.weight", "decoder_1.layer_former_2.mlp.dwconv.dwconv.weight", "decoder_0.layer_former_1.mlp.fc1.bias",
                       "backbone.mhca_stage3.mhca_blks.0.MHCA_layers.0.mlp.norm1.weight", "decoder_0.last_layer.weight"]),
    # have_bridge = "sp": BridgeBlock_sp.  Its MLP_FFN holds a live nn.Dropout(0.1): the vectors are made with p = 0 on those modules (set on the
    # constructed reference model, _variant below) -- the arithmetic of every other operation is pinned, the random mask cannot be
    "bridge_sp": (dict(have_bridge="sp"),
                  ["bridge.bridge_layer1.scale_fuse_att.fc1.weight", "bridge.bridge_layer1.scale_fuse_att.fc3.bias", "bridge.bridge_layer1.scale_fuse_att.fc_back.2.weight",
                   "bridge.bridge_layer1.scale_fuse_att.group_attention.0.Attention.qkv_linear.weight", "bridge.bridge_layer1.scale_fuse_att.group_attention.0.Attention.proj.bias",
                   "bridge.bridge_layer1.scale_fuse_att.group_attention.0.mlp.fc1.weight", "bridge.bridge_layer1.scale_fuse_att.group_attention.0.SlayerNorm_2.weight",
                   "bridge.bridge_layer1.attn.kv.weight", "bridge.bridge_layer4.mixffn3.fc2.weight", "backbone.mhca_stage4.aggregate.conv1.weight",
                   "decoder_0.last_layer.weight"]),
    # Stage_3or4 = 4 builds MSViT_4Stages: Conv2d_BN stem + a first MHCA stage of two paths
    "stage4_coord": (dict(Stage_3or4=4),
                     ["backbone.stem.0.conv.weight", "backbone.stem.0.bn.weight", "backbone.stem.1.conv.weight", "backbone.stem.1.bn.bias",
                      "backbone.patch_embed_stage1.patch_embeds.1.patch_conv.dwconv.weight", "backbone.mhca_stage1.mhca_blks.1.MHCA_layers.0.factoratt_crpe.qkv.weight",
                      "backbone.mhca_stage1.mhca_blks.0.crpe.conv_list.2.weight", "backbone.mhca_stage1.InvRes.conv1.conv.weight",
                      "backbone.mhca_stage1.aggregate.conv1.weight", "backbone.mhca_stage1.aggregate.conv_in_out.weight",
                      "backbone.mhca_stage3.mhca_blks.2.MHCA_layers.7.mlp.fc2.weight", "decoder_0.last_layer.weight"]),
    "stage4_normal": (dict(Stage_3or4=4, concat="normal"),
                      ["backbone.mhca_stage1.aggregate.conv.weight", "backbone.mhca_stage1.aggregate.bn.weight", "backbone.stem.1.conv.weight",
                       "backbone.mhca_stage1.mhca_blks.1.MHCA_layers.0.mlp.fc1.weight", "decoder_0.last_layer.weight"]),
    "stage4_se": (dict(Stage_3or4=4, concat="se"),
                  ["backbone.mhca_stage1.aggregate.excitation.0.weight", "backbone.mhca_stage1.aggregate.excitation.2.weight", "backbone.mhca_stage1.aggregate.conv.weight",
                   "backbone.mhca_stage1.aggregate.conv.bias", "backbone.stem.0.conv.weight", "decoder_0.last_layer.weight"]),
    # (with four stages the spatial attention runs in stages 1-3 whatever use_sa_config says, MSTr.py:2778-2779)
    "stage4_cbam_k3": (dict(Stage_3or4=4, concat="cbam", use_sa_config=3, sa_ker=3),
                       ["backbone.mhca_stage1.aggregate.ca.se.0.weight", "backbone.mhca_stage1.aggregate.sa.conv.weight", "backbone.mhca_stage3.aggregate.sa.conv.weight",
                        "backbone.mhca_stage1.aggregate.conv2d_bn_act.0.weight", "backbone.stem.1.bn.weight", "decoder_0.last_layer.weight"]),
    "stage4_skn": (dict(Stage_3or4=4, concat="skn"),
                   ["backbone.mhca_stage1.aggregate.fc.weight", "backbone.mhca_stage1.aggregate.fcs.2.weight", "backbone.mhca_stage1.aggregate.conv_bn_ac.0.weight",
                    "backbone.mhca_stage2.aggregate.fcs.3.bias", "backbone.stem.0.bn.weight", "decoder_0.last_layer.weight"]),
    "bridge_para": (dict(have_bridge="para"),
                    ["backbone.mhca_stage3.aggregate.conv1.weight", "bridge.bridge_layer1.attn.q.weight", "bridge.bridge_layer2.attn.kv.weight",
                     "bridge.proj_act.0.weight", "bridge.proj_act.0.bias", "bridge.proj_act.1.weight", "bridge.bridge_layer3.mixffn4.fc2.weight",
                     "bridge.bridge_layer4.attn.proj.bias", "decoder_0.last_layer.weight"]),
}


class dense_batchnorm3d_input:
    """Works around a host-framework defect while the cam_fact vectors are made, without touching the reference's arithmetic.

    CAM_Factorized_Module (MSTr.py:559-567) hands BatchNorm3d a tensor whose strides are channels-last-3d.  torch 2.10's CPU batch-norm backward
    returns a wrong input gradient for that layout when the incoming gradient is dense (it differs from the same call on a dense copy of the input by
    more than the gradient's own norm; forward is unaffected; scripts/exp/bn3d_cl_backward.py reproduces it in ten lines).  The vectors must pin the
    reference's algorithm, not this build's CPU kernel, so BatchNorm3d sees a dense copy of its input while this variant runs."""

    def __enter__(self):
        self.orig = orig = torch.nn.BatchNorm3d.forward
        torch.nn.BatchNorm3d.forward = lambda m, x: orig(m, x.contiguous())

    def __exit__(self, *a):
        torch.nn.BatchNorm3d.forward = self.orig


class _nothing:
    def __enter__(self): pass
    def __exit__(self, *a): pass


def variants(MST, Dice, only=()):
    """only: names to (re)generate -- the other entries of the committed file are kept as they are."""
    out = {}
    if only:
        with np.load(os.path.join(HERE, "variants.npz")) as old:
            out = {k: old[k] for k in old.files if k.split("/")[0] not in only}
    for name, (kw, probes) in VARIANTS.items():
        if only and name not in only:
            continue
        factorized = kw.get("concat") == "cam_fact" or (kw.get("Stage_3or4", 3) == 5 and kw.get("concat", "coord") not in ("normal", "3d", "se", "skn", "cbam", "cam"))
        with (dense_batchnorm3d_input() if factorized else _nothing()):
            _variant(out, name, kw, probes, MST, Dice)
    np.savez_compressed(os.path.join(HERE, "variants.npz"), **out)


def _variant(out, name, kw, probes, MST, Dice):
    x = torch.from_numpy(seeded_input(1))
    y_lab = torch.from_numpy(seeded_labels(1))
    ref = MST(num_classes=9, **kw)
    entries = schema_entries(ref)
    out[name + "/schema_sha256"] = np.frombuffer(schema_digest(entries).encode(), dtype=np.uint8)
    out[name + "/n_keys"] = np.array([len(entries), len({c for _, _, c in entries})], dtype=np.int64)
    sd = seeded_state_dict(entries)
    ref.load_state_dict(sd, strict=True)
    if kw.get("have_bridge") == "sp":
        for mod in ref.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
    ref.train()
    logits = ref(x)
    pack(out, name + "/logits", logits)
    ce = torch.nn.functional.cross_entropy(logits, y_lab)
    dice = Dice(9)(logits, y_lab, softmax=True)
    loss = 0.4 * ce + 0.6 * dice
    loss.backward()
    out[name + "/loss"] = np.array([loss.item(), ce.item(), dice.item()], dtype=np.float64)
    named = dict(ref.named_parameters())
    live = sorted(n for n, p in named.items() if p.grad is not None)
    out[name + "/n_live"] = np.array([len(live)], dtype=np.int64)
    for n in probes:
        pack(out, name + "/grad/" + n, named[n].grad)
    ref2 = MST(num_classes=9, **kw)
    ref2.load_state_dict(sd, strict=True)
    ref2.eval()
    with torch.no_grad():
        pack(out, name + "/logits_eval", ref2(x))
    print(name, "keys", len(entries), "live grads", len(live), "loss", loss.item())


if __name__ == "__main__":
    torch.set_num_threads(8)
    MST, Dice = import_reference()
    which = [a for a in sys.argv[1:] if not a.startswith("variant=")] or ["model", "trace", "modules", "variants"]
    only_variants = tuple(a[len("variant="):] for a in sys.argv[1:] if a.startswith("variant="))      # e.g.  variants variant=token_mlp_mix
    if "model" in which:
        whole_model(MST, Dice)
    if "trace" in which:
        train_trace(MST, Dice)
    if "modules" in which:
        modules(MST)
    if "variants" in which:
        variants(MST, Dice, only_variants)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))
