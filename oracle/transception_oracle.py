"""CPU oracle for the TransCeption (MSTransception) forward/backward path.

TEST INFRASTRUCTURE ONLY.  This file is the checker, never the product: only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it.  ``transception_amd`` (the product) never imports anything from
``oracle/`` and has no CPU fallback.

What it is: an independent restatement, in plain fp32 PyTorch-CPU functional
ops, of the arithmetic of the reference's default model
``networks/MSTr.py::MSTransception(num_classes=9)`` (reference file:line cited
per function below), written token-major / NHWC from the index-level
description in SURVEY.md section 8(a) and Appendix C.  It consumes a flat
``{state_dict key: tensor}`` mapping in the reference's own key schema and
weight layouts.

Parity pin: the reference itself has no golden vectors or tests for this path
(SURVEY.md section 4).  The oracle is therefore pinned against *outputs of the
reference run in the build container*: ``tests/golden/make_golden.py`` imports
``/root/reference``, loads the name-seeded weights with ``strict=True`` and
stores sampled activations / gradients / a 2-step train trace under
``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks this file against
them (<= 1e-5 abs on logits).  For image sizes other than 224 the reference
cannot run (hard-coded 224 literals, MSTr.py:2228-2231,2394-2397,2786); there
this oracle, pinned at 224, is the only reference.
"""
from __future__ import annotations

import math
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


class TransCeptionOracle:
    """Functional model over a reference-schema parameter dict (fp32, CPU)."""

    DIMS = (64, 128, 320, 512)
    LAYERS = (3, 8, 3)          # MHCA blocks per path, stages 2..4   (MSTr.py:1580-1582)
    HEADS = 8
    CRPE_WINDOW = ((3, 2), (5, 3), (7, 3))   # (kernel, heads)       (MSTr.py:958)

    def __init__(self, params: Dict[str, Tensor], num_classes: int = 9, training: bool = True, concat: str = "coord",
                 have_bridge: str = "original", br_ch_att_list=(True, False, False, False), use_sa_config: int = 1, sa_ker: int = 7,
                 Stage_3or4: int = 3, inter: str = "res", token_mlp_mode: str = "mix_skip", num_sp: int = 1, sp_dropout: float = 0.1):
        self.P = params
        self.num_classes = num_classes
        self.training = training
        # ablation switches of the reference constructor (MSTr.py:2760-2823) that this restatement follows: the aggregate of a
        # stage (:1384-1403), whether the bridge runs (:2840) and which bridge layers use channel attention (:2413-2420)
        assert concat in ("coord", "normal", "se", "3d", "skn", "cbam", "cam", "cam_fact") and len(br_ch_att_list) == 4
        # have_bridge = "sp": BridgeBlock_sp (MSTr.py:2728-2757) -- SpatialAwareTrans ahead of the first of four all-spatial bridge layers.  Its
        # MLP_FFN carries a live Dropout(0.1) (:63-77): sp_dropout = 0 restates the arithmetic without the random mask (what the fixtures pin)
        self.num_sp, self.sp_dropout = int(num_sp), float(sp_dropout)
        if have_bridge == "sp":
            br_ch_att_list = (False, False, False, False)
        # Stage_3or4 = 4: MSViT_4Stages (MSTr.py:1746-1988) -- a Conv2d_BN stem and a first MHCA stage with two paths in place of the
        # OverlapPatchEmbeddings + EfficientTransformerBlocks of MSViT; restated for the default aggregate ("coord") only
        assert Stage_3or4 != 4 or concat in ("coord", "normal", "se", "cbam", "skn")     # ("3d" / "cam" / "cam_fact" hard-code four maps: the reference fails on the three of a two-path stage)
        self.four_stages = Stage_3or4 == 4
        # token_mlp of the EfficientTransformerBlocks (stage 1 and the decoder, MSTr.py:157-162): "mix_skip" (default) | "mix" (MixFFN, :35-46).  Any
        # other value builds MLP_FFN (:63-77), whose forward(x) the block calls with (x, H, W): the reference raises there, nothing to follow.
        assert token_mlp_mode in ("mix_skip", "mix")
        self.token_mlp_mode = token_mlp_mode
        self.inter = "out"                      # CBAMBlock: the spatial attention reads the gated concatenation
        if Stage_3or4 not in (3, 4):
            # MSViT_casa (MSTr.py:2788-2791, 1990-2207): MSViT with MHCA_stage_casa (:1443-1534), which has no CoordAtt branch -- "coord" (any name it
            # does not know) falls through to Conv3d_BN_channel_attention_concat with CAM_Factorized_Module (:1497-1502, 631-635) -- and whose "cbam"
            # is CBAMBlock_casa (:1213-1257): spatial attention from the ResBlock branch (inter "res"), from the gated concatenation ("out"), or none
            if concat not in ("normal", "3d", "se", "skn", "cbam", "cam"):
                concat = "cam_fact"
            self.inter = inter
        self.concat, self.have_bridge, self.br_ch_att_list = concat, have_bridge, tuple(bool(b) for b in br_ch_att_list)
        # which stages' CBAM blocks apply the spatial attention (MSTr.py:2766-2775)
        self.use_sa_list = {1: (True, True, False), 2: (True, False, False), 3: (False, False, False), 4: (True, True, True)}.get(use_sa_config, (True, True, True))
        # running statistics are buffers: updated in place in training mode
        self.buffers = {k: v.clone() for k, v in params.items()
                        if k.endswith(("running_mean", "running_var", "num_batches_tracked"))}
        self.taps: Dict[str, Tensor] = {}

    # ------------------------------------------------------------------ primitives
    def linear(self, x: Tensor, name: str, bias: bool = True) -> Tensor:
        w = self.P[name + ".weight"]
        w = w.reshape(w.shape[0], -1)                      # 1x1 convs are Linear on tokens
        b = self.P.get(name + ".bias") if bias else None
        return F.linear(x, w, b)

    def layernorm(self, x: Tensor, name: str, eps: float = 1e-5) -> Tensor:
        return F.layer_norm(x, (x.shape[-1],), self.P[name + ".weight"], self.P[name + ".bias"], eps)

    def batchnorm_rows(self, x: Tensor, name: str) -> Tensor:
        """BatchNorm2d over the last (channel) dim of a [..., C] tensor: statistics over all leading dims.
        eps 1e-5, momentum 0.1, biased var for normalisation, unbiased for the running estimate."""
        C = x.shape[-1]
        x2 = x.reshape(-1, C)
        y = F.batch_norm(x2, self.buffers[name + ".running_mean"], self.buffers[name + ".running_var"],
                         self.P[name + ".weight"], self.P[name + ".bias"], self.training, 0.1, 1e-5)
        if self.training:
            self.buffers[name + ".num_batches_tracked"] += 1
        return y.reshape(x.shape)

    def dwconv_map(self, x: Tensor, name: str, k: int, stride: int = 1) -> Tensor:
        """Depthwise k x k conv on an NHWC map, same-padding (k-1)//2."""
        w = self.P[name + ".weight"]
        b = self.P.get(name + ".bias")
        y = F.conv2d(x.permute(0, 3, 1, 2), w, b, stride=stride, padding=(k - 1) // 2, groups=w.shape[0])
        return y.permute(0, 2, 3, 1)

    @staticmethod
    def hardswish(x: Tensor) -> Tensor:
        return x * torch.clamp(x + 3.0, 0.0, 6.0) / 6.0

    # ------------------------------------------------------------------ stage 1
    def patch_embed1(self, x: Tensor) -> Tuple[Tensor, int, int]:
        """OverlapPatchEmbeddings, MSTr.py:292-304 (k7 s4 p3, 3->64, then LN eps 1e-5)."""
        y = F.conv2d(x, self.P["backbone.patch_embed1.proj.weight"], self.P["backbone.patch_embed1.proj.bias"],
                     stride=4, padding=3)
        B, C, H, W = y.shape
        t = y.permute(0, 2, 3, 1).reshape(B, H * W, C)
        return self.layernorm(t, "backbone.patch_embed1.norm"), H, W

    def efficient_attention(self, t: Tensor, name: str) -> Tensor:
        """EfficientAttention with head_count forced to 1, MSTr.py:106-143,154-155 (Appendix C.6)."""
        k = self.linear(t, name + ".keys")
        q = self.linear(t, name + ".queries")
        v = self.linear(t, name + ".values")
        ksm = torch.softmax(k, dim=1)                       # over tokens, per channel
        qsm = torch.softmax(q, dim=2)                       # over channels, per token
        ctx = ksm.transpose(1, 2) @ v                       # [B, C, C']
        self.taps[name + ".ctx"] = ctx                      # (the fused kernels keep this context matrix: tests compare it)
        att = qsm @ ctx                                     # [B, N, C']
        return self.linear(att, name + ".reprojection")

    def mixffn_skip(self, t: Tensor, name: str, H: int, W: int) -> Tensor:
        """MixFFN_skip, MSTr.py:889-902 with DWConv :21-31; fc1 is evaluated once (same value)."""
        B, N, _ = t.shape
        h = self.linear(t, name + ".fc1")
        C4 = h.shape[-1]
        d = self.dwconv_map(h.reshape(B, H, W, C4), name + ".dwconv.dwconv", 3).reshape(B, N, C4) + h
        a = F.gelu(self.layernorm(d, name + ".norm1"))
        return self.linear(a, name + ".fc2")

    def mixffn(self, t: Tensor, name: str, H: int, W: int) -> Tensor:
        """MixFFN, MSTr.py:35-46: fc2(GELU(dw3x3(fc1(x)))) -- no skip around the convolution, no LayerNorm."""
        B, N, _ = t.shape
        h = self.linear(t, name + ".fc1")
        C4 = h.shape[-1]
        d = self.dwconv_map(h.reshape(B, H, W, C4), name + ".dwconv.dwconv", 3).reshape(B, N, C4)
        return self.linear(F.gelu(d), name + ".fc2")

    def efficient_block(self, t: Tensor, name: str, H: int, W: int) -> Tensor:
        """EfficientTransformerBlock, MSTr.py:164-173."""
        tx = t + self.efficient_attention(self.layernorm(t, name + ".norm1"), name + ".attn")
        mlp = self.mixffn_skip if self.token_mlp_mode == "mix_skip" else self.mixffn
        return tx + mlp(self.layernorm(tx, name + ".norm2"), name + ".mlp", H, W)

    # ------------------------------------------------------------------ RIPM
    def dwconv2d_bn(self, x: Tensor, name: str, stride: int) -> Tensor:
        """DWConv2d_BN, MSTr.py:355-362: dw3x3 (no bias) -> pw1x1 (no bias) -> BN -> Hardswish."""
        y = self.dwconv_map(x, name + ".dwconv", 3, stride)
        y = self.linear(y, name + ".pwconv", bias=False)
        return self.hardswish(self.batchnorm_rows(y, name + ".bn"))

    def patch_embed_stage(self, x: Tensor, name: str, num_path: int = 3, pool: bool = True) -> List[Tensor]:
        """Patch_Embed_stage, MSTr.py:704-732: chained, first one stride 2 (isPool)."""
        outs = []
        for i in range(num_path):
            x = self.dwconv2d_bn(x, f"{name}.patch_embeds.{i}.patch_conv", 2 if (i == 0 and pool) else 1)
            outs.append(x)
        return outs

    def conv2d_bn_hswish(self, x: Tensor, name: str, stride: int, pad: int) -> Tensor:
        """Conv2d_BN(act_layer = Hardswish), MSTr.py:364-404: conv (no bias) -> BatchNorm -> Hardswish, NCHW in, NHWC out."""
        y = F.conv2d(x, self.P[name + ".conv.weight"], None, stride=stride, padding=pad).permute(0, 2, 3, 1)
        return self.hardswish(self.batchnorm_rows(y, name + ".bn"))

    def resblock(self, x: Tensor, name: str) -> Tensor:
        """ResBlock ('InvRes'), MSTr.py:1042-1050."""
        f = self.hardswish(self.batchnorm_rows(self.linear(x, name + ".conv1.conv", bias=False), name + ".conv1.bn"))
        f = self.dwconv_map(f, name + ".dwconv", 3)
        f = self.hardswish(self.batchnorm_rows(f, name + ".norm"))
        f = self.batchnorm_rows(self.linear(f, name + ".conv2.conv", bias=False), name + ".conv2.bn")
        return x + f

    # ------------------------------------------------------------------ MB transformer
    def factor_att(self, t: Tensor, blk: str, enc: str, H: int, W: int) -> Tensor:
        """FactorAtt_ConvRelPosEnc + ConvRelPosEnc, MSTr.py:852-886, 801-823 (Appendix C.1, C.2)."""
        B, N, C = t.shape
        h = self.HEADS
        Ch = C // h
        qkv = self.linear(t, blk + ".factoratt_crpe.qkv")
        q, k, v = qkv[..., :C], qkv[..., C:2 * C], qkv[..., 2 * C:]
        ksm = torch.softmax(k, dim=1).reshape(B, N, h, Ch)
        vh = v.reshape(B, N, h, Ch)
        qh = q.reshape(B, N, h, Ch)
        ctx = torch.einsum("bnhi,bnhj->bhij", ksm, vh)
        fa = torch.einsum("bnhi,bhij->bnhj", qh, ctx).reshape(B, N, C)
        # conv relative position encoding: head groups get 3x3 / 5x5 / 7x7 depthwise convs of v
        vmap = v.reshape(B, H, W, C)
        pieces, c0 = [], 0
        for i, (ksz, nh) in enumerate(self.CRPE_WINDOW):
            width = nh * Ch
            w = self.P[f"{enc}.crpe.conv_list.{i}.weight"]
            b = self.P[f"{enc}.crpe.conv_list.{i}.bias"]
            seg = vmap[..., c0:c0 + width].permute(0, 3, 1, 2)
            pieces.append(F.conv2d(seg, w, b, padding=ksz // 2, groups=width).permute(0, 2, 3, 1))
            c0 += width
        conv_v = torch.cat(pieces, dim=-1).reshape(B, N, C)
        out = (Ch ** -0.5) * fa + q * conv_v
        return self.linear(out, blk + ".factoratt_crpe.proj")

    def mhca_block(self, t: Tensor, blk: str, enc: str, H: int, W: int) -> Tensor:
        """MHCABlock, MSTr.py:935-946: shared cpe applied in every block; LN eps 1e-6."""
        B, N, C = t.shape
        t = self.dwconv_map(t.reshape(B, H, W, C), enc + ".cpe.proj", 3).reshape(B, N, C) + t
        t = t + self.factor_att(self.layernorm(t, blk + ".norm1", 1e-6), blk, enc, H, W)
        return t + self.mixffn_skip(self.layernorm(t, blk + ".norm2", 1e-6), blk + ".mlp", H, W)

    def coord_att(self, x: Tensor, name: str) -> Tensor:
        """CoordAtt (IFF), MSTr.py:1322-1348 with silu_swish :1270-1286 (Appendix C.8)."""
        B, H, W, C = x.shape
        y = torch.cat([x.mean(dim=2), x.mean(dim=1)], dim=1)          # [B, H+W, C]
        y = self.batchnorm_rows(self.linear(y, name + ".conv1"), name + ".bn1")
        y = y * torch.clamp(F.silu(y + 3.0) / 6.0, max=1.0)
        a_h = torch.sigmoid(self.linear(y[:, :H], name + ".conv_h"))  # [B, H, C]
        a_w = torch.sigmoid(self.linear(y[:, H:], name + ".conv_w"))  # [B, W, C]
        gated = x * a_w[:, None, :, :] * a_h[:, :, None, :]
        return self.linear(gated, name + ".conv_in_out")

    def mhca_stage(self, maps: List[Tensor], name: str, layers: int) -> Tensor:
        """MHCA_stage, MSTr.py:1412-1441."""
        outs = [self.resblock(maps[0], name + ".InvRes")]
        for p, m in enumerate(maps):
            B, H, W, C = m.shape
            t = m.reshape(B, H * W, C)
            enc = f"{name}.mhca_blks.{p}"
            for l in range(layers):
                t = self.mhca_block(t, f"{enc}.MHCA_layers.{l}", enc, H, W)
            outs.append(t.reshape(B, H, W, C))
        cat = torch.cat(outs, dim=-1)
        if self.concat == "coord":
            return self.coord_att(cat, name + ".aggregate")
        if self.concat == "3d":
            # Conv3d_BN_concat, MSTr.py:447-462: the branch maps stacked on a depth axis, Conv3d(C, out, kernel (4, 1, 1)) = a sum over the
            # four paths and the channels, ReLU, then BatchNorm
            agg = name + ".aggregate"
            w = self.P[agg + ".interact_concat.0.weight"][:, :, :, 0, 0]          # [O, C, 4]
            z = sum(torch.einsum("bhwc,oc->bhwo", outs[p], w[:, :, p]) for p in range(4)) + self.P[agg + ".interact_concat.0.bias"]
            B, H, W, O = z.shape
            return self.batchnorm_rows(torch.relu(z).reshape(B, H * W, O), agg + ".bn").reshape(B, H, W, O)
        if self.concat in ("cam", "cam_fact"):
            # Conv3d_BN_channel_attention_concat with CAM_Module, MSTr.py:642-668, 478-509: BatchNorm3d of the stacked maps, the 4 x 4 path attention
            # per image and channel, BatchNorm3d again, Conv3d(kernel (4, 1, 1)) + GELU, BatchNorm2d.  (bn3d's passes over the partial stacks, whose
            # results the reference discards, only touch its running statistics and are not restated.)
            agg = name + ".aggregate"
            B, H, W, C = outs[0].shape
            x = torch.stack(outs, dim=3).reshape(B, H * W * 4, C)                 # rows (token, path)
            x = self.batchnorm_rows(x, agg + ".bn3d").reshape(B, H * W, 4, C)
            if self.concat == "cam":
                e = torch.einsum("bnpc,bnqc->bcpq", x, x)
                att = torch.softmax(e.max(dim=-1, keepdim=True)[0] - e, dim=-1)
                x = self.P[agg + ".channelAttention.gamma"] * torch.einsum("bcpq,bnqc->bnpc", att, x) + x
            else:
                # CAM_Factorized_Module, MSTr.py:528-568: all 4 N tokens of an image through one factorized attention (no position term)
                ca, h = agg + ".channelAttention", self.HEADS
                t = x.reshape(B, H * W * 4, C)
                qkv = self.linear(t, ca + ".qkv")
                q, k, v = (qkv[..., i * C:(i + 1) * C].reshape(B, -1, h, C // h) for i in range(3))
                ctx = torch.einsum("bnhi,bnhj->bhij", torch.softmax(k, dim=1), v)
                fa = (C // h) ** -0.5 * torch.einsum("bnhi,bhij->bnhj", q, ctx).reshape(B, -1, C)
                x = (self.P[ca + ".gamma"] * self.linear(fa, ca + ".proj") + t).reshape(B, H * W, 4, C)
            x = self.batchnorm_rows(x.reshape(B, H * W * 4, C), agg + ".bn3d").reshape(B, H * W, 4, C)
            w = self.P[agg + ".interact_concat.0.weight"][:, :, :, 0, 0]          # [O, C, 4]
            z = torch.einsum("bnpc,ocp->bno", x, w) + self.P[agg + ".interact_concat.0.bias"]
            return self.batchnorm_rows(F.gelu(z), agg + ".bn").reshape(B, H, W, -1)
        if self.concat == "cbam":
            # CBAMBlock, MSTr.py:1198-1211 with ChannelAttention :1141-1146 and SpatialAttention :1155-1165: out = x * sigmoid(se(max) + se(avg));
            # (stages with use_sa) out = out * sigmoid(conv_kxk([max_c out, mean_c out])); then Conv1x1(out + x) -> BatchNorm -> ReLU
            agg = name + ".aggregate"
            B, H, W, C4 = cat.shape
            flat = cat.reshape(B, H * W, C4)
            se = lambda t: self.linear(torch.relu(self.linear(t, agg + ".ca.se.0", bias=False)), agg + ".ca.se.2", bias=False)
            ca = torch.sigmoid(se(flat.max(dim=1).values) + se(flat.mean(dim=1)))
            o = cat * ca[:, None, None, :]
            stage = int(name[-1])
            # MSViT_4Stages (MSTr.py:2778-2779): use_sa_list = [True, True, True, False] for its stages 1..4 whatever use_sa_config says
            use_sa = (True, True, True, False)[stage - 1] if self.four_stages else self.use_sa_list[stage - 2]
            if use_sa and self.inter in ("res", "out"):
                src = o if self.inter == "out" else outs[0]                                             # CBAMBlock_casa :1243-1251: x[0], the ResBlock branch
                st = torch.stack([src.max(dim=-1).values, src.mean(dim=-1)], dim=1)                     # [B, 2, H, W]
                k = self.P[agg + ".sa.conv.weight"].shape[-1]
                sa = torch.sigmoid(F.conv2d(st, self.P[agg + ".sa.conv.weight"], self.P[agg + ".sa.conv.bias"], padding=k // 2))
                o = o * sa.permute(0, 2, 3, 1)
            z = self.linear((o + cat).reshape(B, H * W, C4), agg + ".conv2d_bn_act.0", bias=False)
            return torch.relu(self.batchnorm_rows(z, agg + ".conv2d_bn_act.1")).reshape(B, H, W, -1)
        if self.concat == "skn":
            # SK_Block, MSTr.py:1076-1107: U = sum of the branch maps, S = its spatial mean, Z = fc(S), one Linear(d, C) per branch, softmax over
            # the branches, V = sum_k a_k x_k, then Conv1x1 (bias) -> ReLU -> BatchNorm
            agg = name + ".aggregate"
            S = sum(outs).mean(dim=(1, 2))
            Z = self.linear(S, agg + ".fc")
            a = torch.softmax(torch.stack([self.linear(Z, f"{agg}.fcs.{k}") for k in range(len(outs))], 0), dim=0)      # [branches, B, C]
            V = sum(a[k][:, None, None, :] * outs[k] for k in range(len(outs)))
            B, H, W, C = V.shape
            z = torch.relu(self.linear(V.reshape(B, H * W, C), agg + ".conv_bn_ac.0"))
            return self.batchnorm_rows(z, agg + ".conv_bn_ac.2").reshape(B, H, W, -1)
        if self.concat == "se":
            # SE_Block, MSTr.py:571-594: squeeze (mean over the map) -> Linear(4C, 4C/16, no bias) -> ReLU -> Linear(4C/16, 4C, no bias)
            # -> sigmoid gates the channels; then conv1x1 (with bias) -> BatchNorm -> ReLU
            B, H, W, C4 = cat.shape
            agg = name + ".aggregate"
            y = cat.mean(dim=(1, 2))
            y = torch.relu(self.linear(y, agg + ".excitation.0", bias=False))
            y = torch.sigmoid(self.linear(y, agg + ".excitation.2", bias=False))
            z = self.linear((cat * y[:, None, None, :]).reshape(B, H * W, C4), agg + ".conv")
            return torch.relu(self.batchnorm_rows(z, agg + ".bn")).reshape(B, H, W, -1)
        # "normal": Conv2d_BN(4C -> C_out, 1x1, no bias) + BatchNorm + Hardswish, MSTr.py:1384-1390 with Conv2d_BN :364-404
        B, H, W, C4 = cat.shape
        y = self.linear(cat.reshape(B, H * W, C4), name + ".aggregate.conv", bias=False)
        return self.hardswish(self.batchnorm_rows(y, name + ".aggregate.bn")).reshape(B, H, W, -1)

    def backbone4(self, x: Tensor) -> List[Tensor]:
        """MSViT_4Stages.forward, MSTr.py:1956-1988: stem (two 3x3 stride-2 Conv2d_BN + Hardswish, :1793-1810), then four RIPM + MHCA stages --
        the first with two paths, one layer and no pooling (:1747-1790: num_path [2,3,3,3], num_layers [1,3,8,3])."""
        m = self.conv2d_bn_hswish(x, "backbone.stem.0", 2, 1)
        m = self.conv2d_bn_hswish(m.permute(0, 3, 1, 2), "backbone.stem.1", 2, 1)
        outs = []
        for s, layers, npath in zip((1, 2, 3, 4), (1,) + tuple(self.LAYERS), (2, 3, 3, 3)):
            maps = self.patch_embed_stage(m, f"backbone.patch_embed_stage{s}", npath, pool=s > 1)
            m = self.mhca_stage(maps, f"backbone.mhca_stage{s}", layers)
            outs.append(m)
        return outs

    def backbone(self, x: Tensor) -> List[Tensor]:
        """MSViT.forward, MSTr.py:1709-1744.  Returns four NHWC maps."""
        if self.four_stages:
            return self.backbone4(x)
        t, H, W = self.patch_embed1(x)
        self.taps["patch_embed1"] = t
        for i in range(2):
            t = self.efficient_block(t, f"backbone.block1.{i}", H, W)
        t = self.layernorm(t, "backbone.norm1")
        m = t.reshape(t.shape[0], H, W, -1)
        outs = [m]
        for s, layers in zip((2, 3, 4), self.LAYERS):
            maps = self.patch_embed_stage(m, f"backbone.patch_embed_stage{s}")
            m = self.mhca_stage(maps, f"backbone.mhca_stage{s}", layers)
            outs.append(m)
        return outs

    # ------------------------------------------------------------------ bridge
    @staticmethod
    def stage_tokens(h4: int) -> List[Tuple[int, int, int]]:
        """(side, width multiple of 64, number of 64-wide tokens) for the four stages."""
        return [(8 * h4, 1, (8 * h4) ** 2), (4 * h4, 2, (4 * h4) ** 2 * 2),
                (2 * h4, 5, (2 * h4) ** 2 * 5), (h4, 8, h4 * h4 * 8)]

    def channel_atten(self, t: Tensor, name: str) -> Tensor:
        """M_EfficientChannelAtten, MSTr.py:2309-2353: .reshape(B,C,N) is a flat re-view (Appendix C.3)."""
        B, N, C = t.shape
        k = self.linear(t, name + ".k").reshape(B, C, N)
        q = self.linear(t, name + ".q").reshape(B, C, N)
        v = self.linear(t, name + ".v").reshape(B, C, N)
        ctx = torch.softmax(k, dim=2) @ v.transpose(1, 2)              # [B, C, C']
        o = ctx.transpose(1, 2) @ torch.softmax(q, dim=1)             # [B, C', N]
        return self.linear(o.permute(0, 2, 1), name + ".proj")

    def scale_reduce(self, t: Tensor, name: str, h4: int) -> Tensor:
        """Scale_reduce, MSTr.py:2225-2249 (Appendix C.4): patchify convs k=s=8/4/2 + channel de-interleave."""
        B, _, C = t.shape
        P = h4 * h4
        parts, off = [], 0
        for i, ((side, mult, ntok), ks) in enumerate(zip(self.stage_tokens(h4)[:3], (8, 4, 2))):
            m = t[:, off:off + ntok].reshape(B, side, side, C * mult).permute(0, 3, 1, 2)
            o = F.conv2d(m, self.P[f"{name}.sr{i}.weight"], self.P[f"{name}.sr{i}.bias"], stride=ks)  # [B, C*mult, h4, h4]
            # reduced token (g*P + pos), feature c  <-  conv channel c*mult + g at position pos
            parts.append(o.reshape(B, C, mult, P).permute(0, 2, 3, 1).reshape(B, mult * P, C))
            off += ntok
        parts.append(t[:, off:])
        return self.layernorm(torch.cat(parts, dim=1), name + ".norm")

    def self_atten(self, t: Tensor, name: str, h4: int) -> Tensor:
        """M_EfficientSelfAtten, MSTr.py:2267-2292: single head, d=64, spatial-reduction K/V."""
        C = t.shape[-1]
        q = self.linear(t, name + ".q")
        kv = self.linear(self.scale_reduce(t, name + ".scale_reduce", h4), name + ".kv")
        k, v = kv[..., :C], kv[..., C:]
        att = torch.softmax((q @ k.transpose(1, 2)) * (C ** -0.5), dim=-1)
        return self.linear(att @ v, name + ".proj")

    def bridge_layer(self, t: Tensor, name: str, ch_att: bool, h4: int) -> Tensor:
        """BridgLayer_4, MSTr.py:2373-2409."""
        B, _, C = t.shape
        n = self.layernorm(t, name + ".norm1")
        tx1 = t + (self.channel_atten(n, name + ".attn") if ch_att else self.self_atten(n, name + ".attn", h4))
        tx = self.layernorm(tx1, name + ".norm2")
        outs, off = [], 0
        for i, (side, mult, ntok) in enumerate(self.stage_tokens(h4)):
            g = tx[:, off:off + ntok].reshape(B, side * side, C * mult)
            outs.append(self.mixffn_skip(g, f"{name}.mixffn{i + 1}", side, side).reshape(B, ntok, C))
            off += ntok
        return tx1 + torch.cat(outs, dim=1)

    def multi_scale_atten(self, x: Tensor, name: str) -> Tensor:
        """MultiScaleAtten, MSTr.py:2542-2559: eight heads over the N tokens of a window, UNSCALED scores (self.scale is never used)."""
        B, nb, _, N, C = x.shape
        h = 8
        qkv = self.linear(x, name + ".qkv_linear").reshape(B, nb, nb, N, 3, h, C // h).permute(4, 0, 1, 2, 5, 3, 6)
        q, k, v = qkv[0], qkv[1], qkv[2]
        att = torch.softmax(q @ k.transpose(-1, -2), dim=-1)
        return self.linear((att @ v).transpose(-2, -3).reshape(B, nb, nb, N, C), name + ".proj")

    def inter_trans_block(self, x: Tensor, name: str) -> Tensor:
        """InterTransBlock, MSTr.py:2562-2583 with MLP_FFN :63-77 (fc1, GELU, Dropout, fc2, Dropout)."""
        x = x + self.multi_scale_atten(self.layernorm(x, name + ".SlayerNorm_1", 1e-6), name + ".Attention")
        y = F.gelu(self.linear(self.layernorm(x, name + ".SlayerNorm_2", 1e-6), name + ".mlp.fc1"))
        y = F.dropout(y, self.sp_dropout, self.training)
        y = F.dropout(self.linear(y, name + ".mlp.fc2"), self.sp_dropout, self.training)
        return x + y

    def spatial_aware_trans(self, maps: List[Tensor], name: str) -> List[Tensor]:
        """SpatialAwareTrans, MSTr.py:2586-2664: every scale projected to 64 channels, cut into windows of 8 / 4 / 2 / 1 pixels a side (the same
        H/8 x W/8 grid of windows at every scale: 64 + 16 + 4 + 1 = 85 tokens each), num_sp InterTransBlocks over the window tokens, windows
        put back and projected to their scale's width (fc_back; fc1_back .. fc4_back exist and are never used)."""
        wins, xs = (8, 4, 2, 1), []
        for j, m in enumerate(maps):
            t = self.linear(m, f"{name}.fc{j + 1}")
            B, H, W, C = t.shape
            ws = wins[j]
            xs.append(t.reshape(B, H // ws, ws, W // ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B, H // ws, W // ws, ws * ws, C))
        x = torch.cat(xs, dim=-2)
        for i in range(self.num_sp):
            x = self.inter_trans_block(x, f"{name}.group_attention.{i}")
        outs = []
        for j, item in enumerate(torch.split(x, [w * w for w in wins], dim=-2)):
            B, nb, _, N, C = item.shape
            ws = wins[j]
            item = item.reshape(B, nb, nb, ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B, nb * ws, nb * ws, C)
            outs.append(self.linear(item, f"{name}.fc_back.{j}"))
        return outs

    def bridge(self, maps: List[Tensor]) -> List[Tensor]:
        """BridgeBlock_4, MSTr.py:2422-2442 (NHWC maps are already the packed token layout, Appendix C.5)."""
        B = maps[0].shape[0]
        h4 = maps[3].shape[1]
        if self.have_bridge == "sp" and self.num_sp > 0:    # BridgeLayer_new, MSTr.py:2686-2704: only the first layer is handed the list of maps
            maps = self.spatial_aware_trans(maps, "bridge.bridge_layer1.scale_fuse_att")
        t = torch.cat([m.reshape(B, -1, 64) for m in maps], dim=1)
        if self.have_bridge == "para":                      # BridgeBlock_para, MSTr.py:2500-2524 (BridgLayer_para = BridgLayer_4's arithmetic)
            b1 = self.bridge_layer(t, "bridge.bridge_layer1", True, h4)
            b2 = self.bridge_layer(t, "bridge.bridge_layer2", False, h4)
            d = F.gelu(self.layernorm(self.linear(torch.cat([b1, b2], dim=2), "bridge.proj_act.0"), "bridge.proj_act.1"))
            t = self.bridge_layer(self.bridge_layer(d, "bridge.bridge_layer3", False, h4), "bridge.bridge_layer4", False, h4)
            self.taps["bridge4"] = t
        else:
            for i, ch_att in enumerate(self.br_ch_att_list):
                t = self.bridge_layer(t, f"bridge.bridge_layer{i + 1}", ch_att, h4)
                self.taps[f"bridge{i + 1}"] = t
        outs, off = [], 0
        for side, mult, ntok in self.stage_tokens(h4):
            outs.append(t[:, off:off + ntok].reshape(B, side, side, 64 * mult))
            off += ntok
        return outs

    # ------------------------------------------------------------------ decoder
    def patch_expand(self, t: Tensor, name: str, H: int, W: int, p: int) -> Tensor:
        """PatchExpand (p=2) / FinalPatchExpand_X4 (p=4), MSTr.py:184-201, 213-227 (Appendix C.7)."""
        B, L, _ = t.shape
        assert L == H * W, "input feature has wrong size"
        y = self.linear(t, name + ".expand", bias=False)
        c = y.shape[-1] // (p * p)
        y = y.reshape(B, H, W, p, p, c).permute(0, 1, 3, 2, 4, 5).reshape(B, H * p * W * p, c)
        return self.layernorm(y, name + ".norm")

    def decoder_layer(self, x1: Tensor, skip, name: str, last: bool) -> Tensor:
        """MyDecoderLayer, MSTr.py:271-290."""
        if skip is None:
            side = int(math.isqrt(x1.shape[1]))
            return self.patch_expand(x1, name + ".layer_up", side, side, 2)
        B, H, W, C = skip.shape
        t = self.linear(torch.cat([x1, skip.reshape(B, H * W, C)], dim=-1), name + ".concat_linear")
        t = self.efficient_block(t, name + ".layer_former_1", H, W)
        t = self.efficient_block(t, name + ".layer_former_2", H, W)
        if not last:
            return self.patch_expand(t, name + ".layer_up", H, W, 2)
        y = self.patch_expand(t, name + ".layer_up", H, W, 4)           # [B, 16HW, 64]
        logits = self.linear(y, name + ".last_layer")                   # 1x1 conv 64 -> classes
        return logits.reshape(B, 4 * H, 4 * W, -1).permute(0, 3, 1, 2)

    # ------------------------------------------------------------------ whole model
    def forward(self, x: Tensor) -> Tensor:
        """MSTransception.forward, MSTr.py:2826-2852."""
        if x.shape[1] == 1:
            x = x.repeat(1, 3, 1, 1)
        enc = self.backbone(x)
        for i, m in enumerate(enc):
            self.taps[f"enc{i}"] = m
        br = self.bridge(enc) if self.have_bridge != "None" else enc          # MSTr.py:2840
        B = x.shape[0]
        t3 = self.decoder_layer(br[3].reshape(B, -1, br[3].shape[-1]), None, "decoder_3", False)
        t2 = self.decoder_layer(t3, br[2], "decoder_2", False)
        t1 = self.decoder_layer(t2, br[1], "decoder_1", False)
        self.taps["dec1"] = t1
        return self.decoder_layer(t1, br[0], "decoder_0", True)

    __call__ = forward


# ---------------------------------------------------------------------- loss / step (trainer.py, utils.py)
def dice_loss_sums(prob: Tensor, target: Tensor, n_classes: int):
    """Per-class (intersect, y_sum, z_sum) over the whole batch tensor, utils.py:24-32."""
    onehot = F.one_hot(target.long(), n_classes).permute(0, 3, 1, 2).to(prob.dtype)
    inter = (prob * onehot).sum(dim=(0, 2, 3))
    y_sum = (onehot * onehot).sum(dim=(0, 2, 3))
    z_sum = (prob * prob).sum(dim=(0, 2, 3))
    return inter, y_sum, z_sum


def ce_dice_loss(logits: Tensor, target: Tensor, n_classes: int = 9):
    """loss = 0.4*CE + 0.6*Dice(softmax=True), trainer.py:141-143, utils.py:34-47 (smooth 1e-5, mean over classes)."""
    ce = F.cross_entropy(logits, target.long())
    inter, y_sum, z_sum = dice_loss_sums(torch.softmax(logits, dim=1), target, n_classes)
    dice = (1.0 - (2.0 * inter + 1e-5) / (z_sum + y_sum + 1e-5)).mean()
    return 0.4 * ce + 0.6 * dice, ce, dice


def cosine_lr(base_lr: float, step: int, t_max: int) -> float:
    """CosineAnnealingLR(eta_min=0) closed form after `step` scheduler steps, trainer.py:126-127,151-153."""
    return 0.5 * base_lr * (1.0 + math.cos(math.pi * step / t_max))


def load_params(state: Dict[str, Tensor], requires_grad: bool = False) -> Dict[str, Tensor]:
    """Float parameters become leaf tensors (aliases share one leaf); buffers are passed through."""
    leaves: Dict[int, Tensor] = {}
    out = {}
    for k, v in state.items():
        if v.dtype.is_floating_point and not k.endswith(("running_mean", "running_var")):
            key = v.data_ptr()
            if key not in leaves:
                leaves[key] = v.detach().clone().float().requires_grad_(requires_grad)
            out[k] = leaves[key]
        else:
            out[k] = v.detach().clone()
    return out


def eval_argmax_counts(logits, labels, ncls: int):
    """Evaluation oracle (utils.py:72-76,96-97 + calculate_metric_percase :50-60): (pred, counts[ncls,3]) with
    pred = argmax(softmax(logits, 1), 1) and counts[k] = (|pred==k & gt==k|, |pred==k|, |gt==k|)."""
    import torch
    pred = torch.argmax(torch.softmax(logits.float(), dim=1), dim=1)
    counts = torch.zeros(ncls, 3, dtype=torch.float64)
    for k in range(ncls):
        p, g = pred == k, labels == k
        counts[k, 0], counts[k, 1], counts[k, 2] = (p & g).sum(), p.sum(), g.sum()
    return pred, counts


def eval_dice(counts):
    """medpy.metric.binary.dc (2|A&B| / (|A|+|B|)) under calculate_metric_percase's empty-set conventions, classes 1.."""
    out = []
    for inter, p, g in counts[1:].tolist():
        out.append(2.0 * inter / (p + g) if (p > 0 and g > 0) else (1.0 if p > 0 else 0.0))
    return out


def eval_hd95(pred, gt, spacing=None):
    """95th-percentile symmetric surface distance by brute force, the definition medpy.metric.binary.hd95 implements (called at
    utils.py:55): a surface voxel is a mask voxel with at least one face-neighbour (connectivity 1; outside the array counts as
    background... as in scipy's binary_erosion with border_value 0) outside the mask; distances are Euclidean between voxel
    centres scaled by `spacing`; both directions pooled, numpy percentile 95.  Small masks only (O(n^2))."""
    import numpy as np
    pred, gt = np.asarray(pred) > 0, np.asarray(gt) > 0
    sp = np.ones(pred.ndim) if spacing is None else np.asarray(spacing, np.float64)

    def surface(m):
        pad = np.pad(m, 1, constant_values=False)
        inner = np.ones_like(m)
        for ax in range(m.ndim):
            for sh in (-1, 1):
                sl = [slice(1, -1)] * m.ndim
                sl[ax] = slice(1 + sh, pad.shape[ax] - 1 + sh)
                inner &= pad[tuple(sl)]
        return np.argwhere(m & ~inner) * sp

    a, b = surface(pred), surface(gt)
    dab, dba = np.empty(len(a)), np.full(len(b), np.inf)
    for i0 in range(0, len(a), 256):                                   # (row blocks: the full distance matrix of two 5 k-voxel surfaces is 0.6 GB)
        d = np.sqrt(((a[i0:i0 + 256, None, :] - b[None, :, :]) ** 2).sum(-1))
        dab[i0:i0 + 256] = d.min(1)
        dba = np.minimum(dba, d.min(0))
    return float(np.percentile(np.hstack((dab, dba)), 95))


def eval_metric_percase(pred, gt):
    """(dice, hd95) of one class of one volume under utils.py:50-60's conventions, from the definitions (no code shared with the product)."""
    import numpy as np
    pred, gt = np.asarray(pred) > 0, np.asarray(gt) > 0
    ps, gs = int(pred.sum()), int(gt.sum())
    if ps > 0 and gs > 0:
        return 2.0 * float((pred & gt).sum()) / (ps + gs), eval_hd95(pred, gt)
    return (1.0, 0.0) if ps > 0 else (0.0, 0.0)
