"""MSTransception / TransCeption: the reference's nn.Module surface over the MI355X engine.

Drop-in boundary: same constructor signature as the reference's
``networks/MSTr.py::MSTransception`` (MSTr.py:2760-2761), ``forward(x[B,1|3,H,W]) -> float32 logits
[B,num_classes,H,W]``, and an identical ``state_dict`` schema (2200 keys, PyTorch-native weight layouts,
aliases of the shared cpe/crpe modules included), so ``load_state_dict`` of a reference checkpoint is
strict-clean and ``trainer.py`` / ``test.py`` style loops (``.train()/.eval()``, ``.parameters()``,
``loss.backward()``, ``torch.no_grad()``) work unchanged.

The sub-modules below are *parameter holders only* (never called); they are created in the reference's
constructor order with the reference's initialisers so that a model built under the same torch seed has
the same weights.  All arithmetic runs in ``_run`` as HIP kernel launches through ``engine.Graph``:
token-major / NHWC activations, the four encoder scales written straight into one stage-major bridge
buffer (no pack/unpack copies), weights and gradients in flat fp32 arenas.
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from ._lib import ACT_COORD, ACT_GELU, ACT_HSWISH, ACT_NONE, ACT_RELU, ACT_SIGMOID, TC_BF16, TC_F16, TC_F32, lib
from .engine import Graph, P, Var

DIMS = (64, 128, 320, 512)
LAYERS = (3, 8, 3)
HEADS = 8
CRPE_WINDOW = ((3, 2), (5, 3), (7, 3))
SR_K = (8, 4, 2)
MULT = (1, 2, 5, 8)


# ----------------------------------------------------------------------------------------------------------
# parameter holders, built in the reference's construction order (RNG consumption == reference)
# ----------------------------------------------------------------------------------------------------------
def _xavier_convs(mod: nn.Module):
    for m in mod.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.xavier_uniform_(m.weight)
            if m.bias is not None:
                nn.init.zeros_(m.bias)
        elif isinstance(m, nn.BatchNorm2d):
            m.weight.data.fill_(1)
            m.bias.data.zero_()


def _mk_dwconv2d_bn(ch: int, stride: int) -> nn.Module:          # MSTr.py:309-353
    m = nn.Module()
    m.dwconv = nn.Conv2d(ch, ch, 3, stride, 1, groups=ch, bias=False)
    m.pwconv = nn.Conv2d(ch, ch, 1, 1, 0, bias=False)
    m.bn = nn.BatchNorm2d(ch)
    _xavier_convs(m)
    return m


def _mk_patch_embed_stage(ch: int, npath: int = 3, pool: bool = True) -> nn.Module:                 # MSTr.py:704-722
    m = nn.Module()
    pes = []
    for idx in range(npath):
        pe = nn.Module()
        pe.patch_conv = _mk_dwconv2d_bn(ch, 2 if (idx == 0 and pool) else 1)
        pes.append(pe)
    m.patch_embeds = nn.ModuleList(pes)
    return m


def _mk_conv2d_bn(cin: int, cout: int) -> nn.Module:             # MSTr.py:364-394
    m = nn.Module()
    m.conv = nn.Conv2d(cin, cout, 1, 1, 0, bias=False)
    m.bn = nn.BatchNorm2d(cout)
    nn.init.constant_(m.bn.weight, 1)
    nn.init.constant_(m.bn.bias, 0)
    nn.init.xavier_uniform_(m.conv.weight)
    return m


def _mk_mixffn_skip(c1: int, c2: int) -> nn.Module:              # MSTr.py:889-898
    m = nn.Module()
    m.fc1 = nn.Linear(c1, c2)
    m.dwconv = nn.Module()
    m.dwconv.dwconv = nn.Conv2d(c2, c2, 3, 1, 1, groups=c2)
    m.fc2 = nn.Linear(c2, c1)
    m.norm1 = nn.LayerNorm(c2)
    m.norm2 = nn.LayerNorm(c2)      # present in the reference, never used (MSTr.py:897-898)
    m.norm3 = nn.LayerNorm(c2)
    return m


def _mk_mixffn(c1: int, c2: int) -> nn.Module:                   # MSTr.py:35-41
    m = nn.Module()
    m.fc1 = nn.Linear(c1, c2)
    m.dwconv = nn.Module()
    m.dwconv.dwconv = nn.Conv2d(c2, c2, 3, 1, 1, groups=c2)
    m.fc2 = nn.Linear(c2, c1)
    return m


def _mk_mhca_encoder(dim: int, layers: int) -> nn.Module:        # MSTr.py:949-978
    m = nn.Module()
    m.cpe = nn.Module()
    m.cpe.proj = nn.Conv2d(dim, dim, 3, 1, 1, groups=dim)
    ch = dim // HEADS
    m.crpe = nn.Module()
    m.crpe.conv_list = nn.ModuleList(
        [nn.Conv2d(nh * ch, nh * ch, k, padding=k // 2, groups=nh * ch) for k, nh in CRPE_WINDOW])
    blks = []
    for _ in range(layers):
        b = nn.Module()
        b.cpe = m.cpe                       # shared modules -> state_dict aliases (MSTr.py:920-921)
        b.crpe = m.crpe
        b.factoratt_crpe = nn.Module()
        b.factoratt_crpe.qkv = nn.Linear(dim, dim * 3, bias=True)
        b.factoratt_crpe.proj = nn.Linear(dim, dim)
        b.factoratt_crpe.crpe = m.crpe
        b.mlp = _mk_mixffn_skip(dim, dim * 4)
        b.norm1 = nn.LayerNorm(dim, eps=1e-6)
        b.norm2 = nn.LayerNorm(dim, eps=1e-6)
        blks.append(b)
    m.MHCA_layers = nn.ModuleList(blks)
    return m


def _mk_resblock(ch: int) -> nn.Module:                          # MSTr.py:996-1039
    m = nn.Module()
    m.conv1 = _mk_conv2d_bn(ch, ch)
    m.dwconv = nn.Conv2d(ch, ch, 3, 1, 1, bias=False, groups=ch)
    m.norm = nn.BatchNorm2d(ch)
    m.conv2 = _mk_conv2d_bn(ch, ch)
    _xavier_convs(m)                        # self.apply(_init_weights): conv1.conv, dwconv, conv2.conv in order
    return m


def _mk_coord_att(inp: int, oup: int) -> nn.Module:              # MSTr.py:1304-1320
    m = nn.Module()
    mip = max(8, inp // 16)
    m.conv1 = nn.Conv2d(inp, mip, 1)
    m.bn1 = nn.BatchNorm2d(mip)
    m.conv_h = nn.Conv2d(mip, inp, 1)
    m.conv_w = nn.Conv2d(mip, inp, 1)
    m.conv_in_out = nn.Conv2d(inp, oup, 1)
    return m


def _mk_mhca_stage(dim: int, out_dim: int, layers: int, concat: str = "coord", use_sa: bool = True, sa_ker: int = 7, npath: int = 3) -> nn.Module:   # MSTr.py:1350-1410
    m = nn.Module()
    m.mhca_blks = nn.ModuleList([_mk_mhca_encoder(dim, layers) for _ in range(npath)])
    m.InvRes = _mk_resblock(dim)
    nb = npath + 1                                              # branch maps the aggregate sees (MSViT_4Stages' first stage: two paths + ResBlock)
    if nb != 4 and concat not in ("coord", "normal", "se", "cbam", "skn"):
        raise NotImplementedError("a two-path MHCA stage exists for concat in {'coord', 'normal', 'se', 'cbam', 'skn'}: the reference's '3d' / 'cam' / 'cam_fact' "
                                  "aggregates hard-code four maps and fail on three")
    # aggregate of the four branch outputs: CoordAtt (IFF, the default, :1402-1403), Conv1x1 + BN + Hardswish ("normal", :1384-1390)
    # or SE_Block ("se", :1396-1397, 571-583)
    if concat == "coord":
        m.aggregate = _mk_coord_att(dim * nb, out_dim)
    elif concat == "3d":                                        # Conv3d_BN_concat, MSTr.py:406-462: Conv3d(C, out, (4, 1, 1)) + ReLU over the stacked maps, BatchNorm
        a = nn.Module()
        a.bn = nn.BatchNorm2d(out_dim)
        a.interact_concat = nn.Sequential(nn.Conv3d(dim, out_dim, kernel_size=(4, 1, 1)), nn.ReLU())
        m.aggregate = a
    elif concat == "cam_fact":                                  # Conv3d_BN_channel_attention_concat(cam="cam_fact") with CAM_Factorized_Module, MSTr.py:512-568
        a = nn.Module()
        a.bn = nn.BatchNorm2d(out_dim)
        a.interact_concat = nn.Sequential(nn.Conv3d(dim, out_dim, kernel_size=(4, 1, 1)), nn.GELU())
        ca = nn.Module()
        ca.gamma = nn.Parameter(torch.zeros(1))
        ca.qkv = nn.Linear(dim, dim * 3)
        ca.proj = nn.Linear(dim, dim)
        ch = dim // HEADS
        ca.crpe = nn.Module()                                   # built, never used by its forward (:551-554): its parameters stay without gradient
        ca.crpe.conv_list = nn.ModuleList([nn.Conv2d(nh * ch, nh * ch, k, padding=k // 2, groups=nh * ch) for k, nh in CRPE_WINDOW])
        a.channelAttention = ca
        a.bn3d = nn.BatchNorm3d(dim)
        m.aggregate = a
    elif concat == "cam":                                       # Conv3d_BN_channel_attention_concat(cam="cam"), MSTr.py:596-668 (constructor order kept)
        a = nn.Module()
        a.bn = nn.BatchNorm2d(out_dim)
        a.interact_concat = nn.Sequential(nn.Conv3d(dim, out_dim, kernel_size=(4, 1, 1)), nn.GELU())
        a.channelAttention = nn.Module()
        a.channelAttention.gamma = nn.Parameter(torch.zeros(1))
        a.bn3d = nn.BatchNorm3d(dim)
        m.aggregate = a
    elif concat == "cbam":                                      # CBAMBlock(channel = 4C, reduction = 16, kernel_size = sa_ker), MSTr.py:1169-1211, 1400-1401
        a = nn.Module()
        a.ca = nn.Module()
        a.ca.se = nn.Sequential(nn.Conv2d(dim * nb, dim * nb // 16, 1, bias=False), nn.ReLU(), nn.Conv2d(dim * nb // 16, dim * nb, 1, bias=False))
        a.sa = nn.Module()
        a.sa.conv = nn.Conv2d(2, 1, kernel_size=sa_ker, stride=1, padding=sa_ker // 2)
        a.conv2d_bn_act = nn.Sequential(nn.Conv2d(dim * nb, out_dim, 1, bias=False), nn.BatchNorm2d(out_dim), nn.ReLU())
        a.use_sa = use_sa
        a.sa_ker = sa_ker
        m.aggregate = a
    elif concat == "skn":                                       # SK_Block(in_ch = C, num_path = 4, reduction = 8, L = 32), MSTr.py:1054-1107, 1398-1399
        a = nn.Module()
        dd = max(32, dim // 8)
        a.fc = nn.Linear(dim, dd)
        a.fcs = nn.ModuleList([nn.Linear(dd, dim) for _ in range(nb)])
        a.softmax = nn.Softmax(dim=0)
        a.conv_bn_ac = nn.Sequential(nn.Conv2d(dim, out_dim, kernel_size=(1, 1)), nn.ReLU(inplace=True), nn.BatchNorm2d(out_dim))
        m.aggregate = a
    elif concat == "se":
        a = nn.Module()
        a.excitation = nn.Sequential(nn.Linear(dim * nb, dim * nb // 16, bias=False), nn.ReLU(inplace=True),
                                     nn.Linear(dim * nb // 16, dim * nb, bias=False), nn.Sigmoid())
        a.conv = nn.Conv2d(dim * nb, out_dim, 1)
        a.bn = nn.BatchNorm2d(out_dim)
        m.aggregate = a
    else:
        m.aggregate = _mk_conv2d_bn(dim * nb, out_dim)
    return m


def _mk_eff_block(dim: int, token_mlp: str = "mix_skip") -> nn.Module:                        # MSTr.py:146-162, 95-103
    m = nn.Module()
    m.norm1 = nn.LayerNorm(dim)
    m.attn = nn.Module()
    m.attn.keys = nn.Conv2d(dim, dim, 1)
    m.attn.queries = nn.Conv2d(dim, dim, 1)
    m.attn.values = nn.Conv2d(dim, dim, 1)
    m.attn.reprojection = nn.Conv2d(dim, dim, 1)
    m.norm2 = nn.LayerNorm(dim)
    m.mlp = _mk_mixffn_skip(dim, dim * 4) if token_mlp == "mix_skip" else _mk_mixffn(dim, dim * 4)
    return m


def _mk_scale_reduce(dim: int) -> nn.Module:                     # MSTr.py:2209-2223
    m = nn.Module()
    m.sr0 = nn.Conv2d(dim, dim, 8, 8)
    m.sr1 = nn.Conv2d(dim * 2, dim * 2, 4, 4)
    m.sr2 = nn.Conv2d(dim * 5, dim * 5, 2, 2)
    m.norm = nn.LayerNorm(dim)
    return m


def _mk_bridge_layer(dim: int, ch_att: bool) -> nn.Module:       # MSTr.py:2356-2371
    m = nn.Module()
    m.norm1 = nn.LayerNorm(dim)
    a = nn.Module()
    if ch_att:
        a.q = nn.Linear(dim, dim, bias=True)
        a.k = nn.Linear(dim, dim, bias=True)
        a.v = nn.Linear(dim, dim, bias=True)
        a.proj = nn.Linear(dim, dim)
    else:
        a.q = nn.Linear(dim, dim, bias=True)
        a.kv = nn.Linear(dim, dim * 2, bias=True)
        a.proj = nn.Linear(dim, dim)
    a.scale_reduce = _mk_scale_reduce(dim)     # unused (grad-less) inside the channel-attention layer
    m.attn = a
    m.norm2 = nn.LayerNorm(dim)
    m.mixffn1 = _mk_mixffn_skip(dim, dim * 4)
    m.mixffn2 = _mk_mixffn_skip(dim * 2, dim * 8)
    m.mixffn3 = _mk_mixffn_skip(dim * 5, dim * 20)
    m.mixffn4 = _mk_mixffn_skip(dim * 8, dim * 32)
    return m


def _mk_spatial_aware_trans(dim: int, num_sp: int) -> nn.Module:             # SpatialAwareTrans, MSTr.py:2586-2615 (constructor order kept)
    m = nn.Module()
    chans = [dim * k for k in MULT]
    for j, c in enumerate(chans):
        setattr(m, f"fc{j + 1}", nn.Linear(c, dim))
    for j, c in enumerate(chans):
        setattr(m, f"fc{j + 1}_back", nn.Linear(dim, c))            # built, never used by its forward (:2658 takes fc_back)
    m.fc_back = nn.ModuleList([nn.Linear(dim, c) for c in chans])
    blocks = []
    for _ in range(num_sp):                                         # InterTransBlock, :2562-2568; MultiScaleAtten :2542-2549; MLP_FFN :63-70
        b = nn.Module()
        b.SlayerNorm_1 = nn.LayerNorm(dim, eps=1e-6)
        b.SlayerNorm_2 = nn.LayerNorm(dim, eps=1e-6)
        b.Attention = nn.Module()
        b.Attention.qkv_linear = nn.Linear(dim, dim * 3)
        b.Attention.proj = nn.Linear(dim, dim)
        b.mlp = nn.Module()
        b.mlp.fc1 = nn.Linear(dim, 4 * dim)
        b.mlp.fc2 = nn.Linear(4 * dim, dim)
        blocks.append(b)
    m.group_attention = nn.Sequential(*blocks)
    return m


def _mk_bridge_layer_sp(dim: int, num_sp: int) -> nn.Module:       # BridgeLayer_new, MSTr.py:2668-2684
    m = _mk_bridge_layer(dim, False)
    m.scale_fuse_att = _mk_spatial_aware_trans(dim, num_sp)
    return m


def _mk_decoder_layer(in_out_chan, n_class: int, is_last: bool, token_mlp: str = "mix_skip") -> nn.Module:   # MSTr.py:230-269
    dims, out_dim = in_out_chan[0], in_out_chan[1]
    m = nn.Module()
    m.concat_linear = nn.Linear(dims * (4 if is_last else 2), out_dim)
    m.layer_up = nn.Module()
    if not is_last:
        m.layer_up.expand = nn.Linear(out_dim, 2 * out_dim, bias=False)
        m.layer_up.norm = nn.LayerNorm(out_dim // 2)
    else:
        m.layer_up.expand = nn.Linear(out_dim, 16 * out_dim, bias=False)
        m.layer_up.norm = nn.LayerNorm(out_dim)
        m.last_layer = nn.Conv2d(out_dim, n_class, 1)
    m.layer_former_1 = _mk_eff_block(out_dim, token_mlp)
    m.layer_former_2 = _mk_eff_block(out_dim, token_mlp)
    for sub in m.modules():                 # MSTr.py:255-269
        if isinstance(sub, (nn.Linear, nn.Conv2d)):
            nn.init.xavier_uniform_(sub.weight)
            if sub.bias is not None:
                nn.init.zeros_(sub.bias)
        elif isinstance(sub, nn.LayerNorm):
            nn.init.ones_(sub.weight)
            nn.init.zeros_(sub.bias)
    return m


def _mk_backbone4(concat: str = "coord", sa_ker: int = 7) -> nn.Module:                                             # MSViT_4Stages, MSTr.py:1746-1920
    m = nn.Module()
    for i, d in enumerate(DIMS):
        setattr(m, f"conv1_1_s{i + 1}", nn.Conv2d(3 * d, d, 1))     # dead parameters, kept for the schema
    stem = []
    for cin, cout in ((3, DIMS[0] // 2), (DIMS[0] // 2, DIMS[0])):                                                   # Conv2d_BN(k3, s2, p1, Hardswish) x 2, :1793-1810
        c = nn.Module()
        c.conv = nn.Conv2d(cin, cout, 3, 2, 1, bias=False)
        c.bn = nn.BatchNorm2d(cout)
        nn.init.xavier_uniform_(c.conv.weight)
        stem.append(c)
    m.stem = nn.Sequential(*stem)
    npath, layers = (2, 3, 3, 3), (1,) + tuple(LAYERS)
    for i in range(4):
        setattr(m, f"patch_embed_stage{i + 1}", _mk_patch_embed_stage(DIMS[max(i - 1, 0)], npath[i], pool=i > 0))
    use_sa = (True, True, True, False)                              # MSTr.py:2778-2779: this list whatever use_sa_config says
    for i in range(4):
        setattr(m, f"mhca_stage{i + 1}", _mk_mhca_stage(DIMS[max(i - 1, 0)], DIMS[i], layers[i], concat, use_sa[i], sa_ker, npath[i]))
    m.cpe = nn.Module()
    m.cpe.proj = nn.Conv2d(DIMS[0], DIMS[0], 3, 1, 1, groups=DIMS[0])   # dead
    m.norm1 = nn.LayerNorm(DIMS[0])                                      # dead
    return m


def _mk_backbone(concat: str = "coord", use_sa_list=(True, True, False), sa_ker: int = 7, token_mlp: str = "mix_skip") -> nn.Module:   # MSTr.py:1536-1671
    m = nn.Module()
    for i, d in enumerate(DIMS):
        setattr(m, f"conv1_1_s{i + 1}", nn.Conv2d(3 * d, d, 1))     # dead parameters, kept for the schema
    m.patch_embed_stage2 = _mk_patch_embed_stage(DIMS[0])
    m.patch_embed_stage3 = _mk_patch_embed_stage(DIMS[1])
    m.patch_embed_stage4 = _mk_patch_embed_stage(DIMS[2])
    m.mhca_stage2 = _mk_mhca_stage(DIMS[0], DIMS[1], LAYERS[0], concat, use_sa_list[0], sa_ker)
    m.mhca_stage3 = _mk_mhca_stage(DIMS[1], DIMS[2], LAYERS[1], concat, use_sa_list[1], sa_ker)
    m.mhca_stage4 = _mk_mhca_stage(DIMS[2], DIMS[3], LAYERS[2], concat, use_sa_list[2], sa_ker)
    m.patch_embed1 = nn.Module()
    m.patch_embed1.proj = nn.Conv2d(3, DIMS[0], 7, 4, 3)
    m.patch_embed1.norm = nn.LayerNorm(DIMS[0])
    m.cpe = nn.Module()
    m.cpe.proj = nn.Conv2d(DIMS[0], DIMS[0], 3, 1, 1, groups=DIMS[0])   # dead
    m.block1 = nn.ModuleList([_mk_eff_block(DIMS[0], token_mlp) for _ in range(2)])
    m.norm1 = nn.LayerNorm(DIMS[0])
    return m


# ----------------------------------------------------------------------------------------------------------
class _StepFn(torch.autograd.Function):
    """Connects the engine's tape to loss.backward().  Parameter gradients are accumulated by the kernels into
    the model's gradient arena (returned grads are None); see MSTransception._attach_grads."""

    @staticmethod
    def forward(ctx, model, x, *params):
        logits, G, out_var = model._run(x, record=True)
        ctx.model, ctx.G, ctx.out_var = model, G, out_var
        return logits

    @staticmethod
    def backward(ctx, dlogits):
        ctx.model._backward(ctx.G, ctx.out_var, dlogits)
        ctx.G = ctx.out_var = None
        return (None, None) + (None,) * len(ctx.model._uniq_params)


class MSTransception(nn.Module):
    def __init__(self, num_classes=9, head_count=8, dil_conv=1, token_mlp_mode="mix_skip", MSViT_config=2, concat='coord',
                 have_bridge='original', use_sa_config=1, sa_ker=7, Stage_3or4=3, inter='res', num_sp=1,
                 br_ch_att_list=[True, False, False, False]):
        super().__init__()
        # Ablation switches of the reference constructor (MSTr.py:2760-2823) that compose from the kernels of the default path:
        #   concat       "coord" (CoordAtt / IFF, default) | "normal" (Conv1x1 + BN + Hardswish over the concatenation, :1384-1390)
        #                | "se" (SE_Block over the concatenation, :571-594: squeeze / excitation gate, Conv1x1 + BN + ReLU)
        #                | "3d" (Conv3d_BN_concat, :406-462: Conv3d(kernel (4, 1, 1)) over the stacked branch maps + ReLU, BatchNorm)
        #                | "skn" (SK_Block, :1054-1107: per-channel softmax over the four branches from their pooled sum, Conv1x1 + ReLU + BN)
        #                | "cbam" (CBAMBlock, :1128-1211: channel attention from max + mean pooling, spatial attention per use_sa_config / sa_ker,
        #                  Conv1x1 + BN + ReLU over out + input)
        #   have_bridge  "original" (default) | "None" (the bridge is built -- its parameters stay in the state_dict -- but skipped, :2840)
        #   br_ch_att_list  which of the four bridge layers use channel attention instead of SR self-attention (:2413-2420)
        # use_sa_config / sa_ker / inter / num_sp only reach the "cbam", "sp" and "para" variants and are accepted and ignored, as in
        #                | "para" (BridgeBlock_para, :2500-2538: channel and spatial layer side by side, Linear(128->64)+LN+GELU, two
        #                  more spatial layers)
        #   Stage_3or4   3 (MSViT, default) | 5 (anything but 3 and 4: MSViT_casa, :1990-2207 = MSViT with MHCA_stage_casa, :1443-1534, which has no
        #                  CoordAtt branch: concat = "coord" -- or any string it does not know -- builds Conv3d_BN_channel_attention_concat with
        #                  CAM_Factorized_Module, i.e. what concat = "cam_fact" builds; concat = "cbam" builds CBAMBlock_casa, :1213-1257, whose
        #                  spatial attention reads the ResBlock branch alone when inter = "res", the gated concatenation when inter = "out", and is
        #                  skipped for any other inter)
        #   token_mlp_mode  "mix_skip" (default) | "mix": the EfficientTransformerBlocks of stage 1 and of the decoder use MixFFN (MSTr.py:35-46: no skip
        #                  around the depthwise convolution, no LayerNorm) instead of MixFFN_skip; the MB blocks and the bridge keep MixFFN_skip.  Any other
        #                  value builds MLP_FFN (:63-77), whose forward(x) the block calls with (x, H, W): the reference raises a TypeError there.
        #   Stage_3or4   ... | 4: MSViT_4Stages with concat in {"coord", "normal", "se", "cbam", "skn"} ("3d" / "cam" / "cam_fact" hard-code four maps
        #                  and fail in the reference itself on the three of its two-path first stage)
        #   have_bridge  ... | "sp" (BridgeBlock_sp, :2728-2757: SpatialAwareTrans -- per-scale Linear to 64 channels, windows of 8 / 4 / 2 / 1 pixels,
        #                  num_sp InterTransBlocks of 8-head attention over the 85 tokens of a window + MLP_FFN with Dropout(0.1), windows back, Linear
        #                  to the scale's width -- ahead of four all-spatial bridge layers)
        # the reference.  Not built: the legacy networks/Transception.py class.
        br = [bool(b) for b in br_ch_att_list]
        if (token_mlp_mode not in ("mix_skip", "mix") or concat not in ("coord", "normal", "se", "3d", "skn", "cbam", "cam", "cam_fact")
                or (Stage_3or4 == 4 and concat not in ("coord", "normal", "se", "cbam", "skn")) or len(br) != 4 or (have_bridge == "sp" and int(num_sp) < 0)):
            raise NotImplementedError("MSTransception: implemented are every concat of the reference ('coord', 'normal', 'se', '3d', 'skn', 'cbam', 'cam', 'cam_fact'), have_bridge in {'original', "
                                      "'None', 'para', 'sp'}, any 4-entry br_ch_att_list, Stage_3or4 in {3, 5} (and 4 with concat in {'coord', 'normal', 'se', 'cbam', 'skn'}), token_mlp_mode in {'mix_skip', 'mix'}")
        self.inter = "out"                              # CBAMBlock (Stage_3or4 = 3) gates with the statistics of the gated concatenation
        if Stage_3or4 not in (3, 4):                    # MSViT_casa (MSTr.py:2788-2791: the else branch of 4 / 3)
            if concat not in ("normal", "3d", "se", "skn", "cbam", "cam"):
                concat = "cam_fact"
            self.inter = inter
        if have_bridge == "para":                       # BridgeBlock_para ignores br_ch_att_list (it receives num_sp, MSTr.py:2806-2807):
            br = [True, False, False, False]            # layer 1 channel, layers 2-4 spatial (MSTr.py:2504-2512)
        if have_bridge == "sp":                         # BridgeBlock_sp (MSTr.py:2728-2757): four BridgeLayer_new, all with spatial attention;
            br = [False, False, False, False]           # SpatialAwareTrans (num_sp InterTransBlocks) runs ahead of the first layer only
        self.num_sp = int(num_sp)
        self.sp_dropout = 0.1                           # nn.Dropout(0.1) of the InterTransBlocks' MLP_FFN (MSTr.py:70); training mode only
        self.concat, self.have_bridge, self.br_ch_att_list = concat, have_bridge, br
        self.num_classes = num_classes
        # use_sa_config (MSTr.py:2766-2775): which of the three stages' CBAM blocks apply the spatial attention; sa_ker its kernel size (3 or 7 here)
        use_sa_list = {1: (True, True, False), 2: (True, False, False), 3: (False, False, False), 4: (True, True, True)}.get(use_sa_config, (True, True, True))
        if concat == "cbam" and sa_ker not in (3, 7):
            raise NotImplementedError("MSTransception(concat='cbam'): sa_ker must be 3 or 7")
        self.token_mlp_mode = token_mlp_mode
        # Stage_3or4 = 4: MSViT_4Stages (MSTr.py:1746-1988) -- a Conv2d_BN stem and a first MHCA stage (two paths, one layer) instead of the
        # OverlapPatchEmbeddings + EfficientTransformerBlocks
        self.backbone = _mk_backbone4(concat, sa_ker) if Stage_3or4 == 4 else _mk_backbone(concat, use_sa_list, sa_ker, token_mlp_mode)
        self.Stage_3or4 = Stage_3or4
        self.bridge = nn.Module()
        if have_bridge == "para":                       # constructor order of BridgeBlock_para: layers 1, 2, proj_act, layers 3, 4
            self.bridge.bridge_layer1 = _mk_bridge_layer(64, True)
            self.bridge.bridge_layer2 = _mk_bridge_layer(64, False)
            self.bridge.proj_act = nn.Sequential(nn.Linear(128, 64), nn.LayerNorm(64), nn.GELU())
            self.bridge.bridge_layer3 = _mk_bridge_layer(64, False)
            self.bridge.bridge_layer4 = _mk_bridge_layer(64, False)
        elif have_bridge == "sp":
            for i in range(4):
                setattr(self.bridge, f"bridge_layer{i + 1}", _mk_bridge_layer_sp(64, self.num_sp))
        else:
            for i, ch in enumerate(br):
                setattr(self.bridge, f"bridge_layer{i + 1}", _mk_bridge_layer(64, ch))
        ioc = [[32, 64, 64, 64], [144, 128, 128, 128], [288, 320, 320, 320], [512, 512, 512, 512]]
        self.decoder_3 = _mk_decoder_layer(ioc[3], num_classes, False, token_mlp_mode)
        self.decoder_2 = _mk_decoder_layer(ioc[2], num_classes, False, token_mlp_mode)
        self.decoder_1 = _mk_decoder_layer(ioc[1], num_classes, False, token_mlp_mode)
        self.decoder_0 = _mk_decoder_layer(ioc[0], num_classes, True, token_mlp_mode)
        self.compute_dtype = torch.float32
        self.use_fused_attention = True
        self.capture_taps = False          # tests: keep copies of the stage outputs of the next forward in self.taps (fp32, reference layouts)
        self.taps = {}
        self._flat: Optional[torch.Tensor] = None
        self._gflat: Optional[torch.Tensor] = None
        self._flat_lp: Optional[torch.Tensor] = None
        self._lp_fresh = False                          # FusedSGD wrote the 16-bit copy of the weights with its last update
        self._index: Dict[str, Tuple[int, Tuple[int, ...]]] = {}
        self._uniq_params: List[nn.Parameter] = []
        self._used: set = set()
        self.last_launches = 0

    # ------------------------------------------------------------------ flat parameter / gradient arenas
    def set_compute_dtype(self, dtype: torch.dtype):
        """Storage type of activations and of the working copy of the weights: float32 (the parity path), bfloat16, or float16 (IEEE
        half, BASELINE config 5; train it with a loss scale -- train.SegLoss(loss_scale=...) -- because activation gradients of
        order 1e-6 underflow in half precision).  Master weights, gradients, statistics and accumulators stay fp32 in every mode."""
        assert dtype in (torch.float32, torch.bfloat16, torch.float16)
        if dtype != self.compute_dtype:
            self._flat_lp = None
        self.compute_dtype = dtype
        return self

    def _tc_dtype(self) -> int:
        return {torch.float32: TC_F32, torch.bfloat16: TC_BF16, torch.float16: TC_F16}[self.compute_dtype]

    def _ensure_flat(self, device):
        uniq = list(self.parameters())
        ok = (self._flat is not None and self._flat.device == device and len(uniq) == len(self._uniq_params)
              and uniq[0].data_ptr() == self._flat.data_ptr() and uniq[-1].device == device)
        if ok:
            return
        offs, total = [], 0
        for p in uniq:
            offs.append(total)
            total += (p.numel() + 7) // 8 * 8
        flat = torch.zeros(total, dtype=torch.float32, device=device)
        for p, o in zip(uniq, offs):
            flat[o:o + p.numel()].copy_(p.data.reshape(-1).to(device=device, dtype=torch.float32))
            p.data = flat[o:o + p.numel()].view(p.shape)
            p.grad = None
        self._flat, self._gflat = flat, torch.zeros_like(flat)
        self._flat_lp = None
        self._uniq_params = uniq
        # the 21 BatchNorm step counters live in one int64 vector (each module's buffer is a view of its element), so a training
        # forward bumps them with one launch instead of 21
        bns = [m for m in self.modules() if isinstance(m, nn.BatchNorm2d) and m.num_batches_tracked is not None]
        nbt = torch.stack([m.num_batches_tracked.detach().to(device=device, dtype=torch.int64).reshape(()) for m in bns]) if bns else None
        for i, m in enumerate(bns):
            m._buffers["num_batches_tracked"] = nbt[i]
        self._nbt_flat, self._nbt_ptrs = nbt, [m.num_batches_tracked.data_ptr() for m in bns]
        off_of = {id(p): o for p, o in zip(uniq, offs)}
        self._index = {n: (off_of[id(p)], tuple(p.shape)) for n, p in self.named_parameters(remove_duplicate=False)}
        self._pid = {n: id(p) for n, p in self.named_parameters(remove_duplicate=False)}

    def load_state_dict(self, *args, **kwargs):
        self._lp_fresh = False                          # weights change under the 16-bit working copy
        r = super().load_state_dict(*args, **kwargs)
        self._recast_working_copy()
        return r

    def _recast_working_copy(self):
        """Refresh the 16-bit working copy IN PLACE (same buffer) from the fp32 master weights.  A captured training step (GraphedStep)
        may have been captured right after a FusedSGD update, i.e. without the cast launch: it reads whatever this buffer holds, so
        every path that changes the master weights behind the optimiser's back re-casts eagerly instead of relying on the next eager
        forward."""
        lp, flat = getattr(self, "_flat_lp", None), getattr(self, "_flat", None)
        if lp is None or flat is None or not flat.is_cuda or lp.dtype != self.compute_dtype or lp.numel() != flat.numel():
            return
        lib().tc_cast(flat.data_ptr(), lp.data_ptr(), flat.numel(), TC_F32, self._tc_dtype(), torch.cuda.current_stream(flat.device).cuda_stream)

    def _apply(self, fn, *args, **kwargs):
        self._lp_fresh = False
        return super()._apply(fn, *args, **kwargs)

    def flat_parameters(self) -> torch.Tensor:
        """The fp32 parameter arena.  A caller that writes into it directly (e.g. a broadcast) must do so before the next forward of a
        step sequence driven by something other than FusedSGD, or call `invalidate_working_copy()`."""
        return self._flat

    def invalidate_working_copy(self):
        self._lp_fresh = False
        self._recast_working_copy()

    def flat_gradients(self) -> torch.Tensor:
        return self._gflat

    def _attach_grads(self):
        """Expose arena slices as .grad for the parameters the step touched (the reference leaves 332 tensors None)."""
        byid = {id(p): p for p in self._uniq_params}
        for pid, (off, shape) in self._used_views.items():
            p = byid[pid]
            if p.grad is None:
                p.grad = self._gflat[off:off + p.numel()].view(shape)

    # ------------------------------------------------------------------ nn.Module surface
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if not x.is_cuda:
            raise RuntimeError("transception_amd runs on MI355X only: move the input to a HIP device "
                               "(there is no CPU fallback; the CPU oracle lives under oracle/ for tests)")
        self._ensure_flat(x.device)
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in self._uniq_params)
        if need_grad:
            return _StepFn.apply(self, x, *self._uniq_params)
        logits, G, _ = self._run(x, record=False)
        self.last_launches = G.n_launch
        return logits

    # ------------------------------------------------------------------ engine plumbing
    def _P(self, G: Graph, name: str, view: Optional[Tuple[int, ...]] = None) -> P:
        off, shape = self._index[name]
        n = math.prod(shape)
        shp = view or shape
        src = self._flat if self.compute_dtype == torch.float32 else self._flat_lp
        data = src[off:off + n].view(shp)
        grad = None
        if G.record:
            grad = self._gflat[off:off + n].view(shp)
            self._used_views[self._pid[name]] = (off, shape)
            if G.ngroups > 1:                       # the same parameter of the other MB paths rides along (+g * stride)
                for g in range(1, G.ngroups):
                    other = name.replace(".mhca_blks.0.", f".mhca_blks.{g}.")
                    o2, s2 = self._index[other]
                    assert o2 == off + g * G.pgs and s2 == shape, (name, other)
                    self._used_views[self._pid[other]] = (o2, s2)
        return P(data, grad, G.pgs if G.ngroups > 1 else 0)

    def _dropout_seed(self, G: Graph) -> torch.Tensor:
        """Device counter that keys the dropout masks of a forward pass (have_bridge = "sp"); advanced once per training forward, inside a captured
        step too (the increment is a captured launch -- the counter itself must exist before the capture: train.GraphedStep's eager warm-up
        steps create it)."""
        if getattr(self, "_drop_ctr", None) is None or self._drop_ctr.device != G.dev:
            #  the key starts at ((seed * world + rank) mod 128) << 24: ranks of a DDP job draw different masks, a user seed
            # (TC_DROPOUT_SEED, default torch's initial seed) moves the sequence, and the low 24 bits count forwards (the kernel keys on
            # 32 bits).  Like the reference -- whose masks come from torch's global generator, which trainer.py:49-237 never saves -- a
            # resumed run restarts the sequence; `dropout_counter` / `set_dropout_counter` let a caller carry it across a checkpoint
            # without touching the state_dict schema (2200 keys, strict-loadable both ways).
            import torch.distributed as dist
            rank, world = (dist.get_rank(), dist.get_world_size()) if dist.is_available() and dist.is_initialized() else (0, 1)
            base = int(os.environ.get("TC_DROPOUT_SEED", torch.initial_seed() & 0x7FFF))
            start = getattr(self, "_drop_ctr_restore", None)
            v0 = start if start is not None else (((base * world + rank) & 0x7F) << 24)
            self._drop_ctr = torch.full((1,), v0, dtype=torch.int64, device=G.dev)
            self._drop_ctr_restore = None
        if G.training:
            self._drop_ctr += 1
        return self._drop_ctr

    def dropout_counter(self) -> Optional[int]:
        c = getattr(self, "_drop_ctr", None)
        return int(c.item()) if c is not None else None

    def set_dropout_counter(self, value: int) -> None:
        self._drop_ctr_restore, self._drop_ctr = int(value), None


    def _path_stride(self, stage: str, npath: int = 3) -> int:
        """Distance in the flat arenas between the parameter blocks of two consecutive MB encoders of a stage."""
        offs = [self._index[f"{stage}.mhca_blks.{g}.cpe.proj.weight"][0] for g in range(npath)]
        d = offs[1] - offs[0]
        assert all(offs[g + 1] - offs[g] == d for g in range(npath - 1)) and d % 8 == 0
        return d

    def _run(self, x: torch.Tensor, record: bool, token_logits: bool = False):
        """token_logits: hand the logits over as the classifier Linear leaves them -- token-major [B*H*W, classes] in the storage type
        (train.GraphedStep: tc_seg_loss_*_tok read / write that layout) -- instead of the module's fp32 NCHW output."""
        dev = x.device
        L = lib()
        stream = torch.cuda.current_stream(dev).cuda_stream
        if self.compute_dtype != torch.float32:
            fresh = self._lp_fresh and self._flat_lp is not None and self._flat_lp.dtype == self.compute_dtype
            self._lp_fresh = False                      # only the step that FusedSGD just finished vouches for the copy
            if not fresh:
                if self._flat_lp is None or self._flat_lp.dtype != self.compute_dtype:
                    self._flat_lp = torch.empty(self._flat.numel(), dtype=self.compute_dtype, device=dev)
                L.tc_cast(self._flat.data_ptr(), self._flat_lp.data_ptr(), self._flat.numel(), TC_F32, self._tc_dtype(), stream)
        if record:
            self._used_views = {}
        G = Graph(self.compute_dtype, dev, self.training, record)
        G.use_fused_attention = self.use_fused_attention
        B, in_ch, H, W = x.shape
        if in_ch not in (1, 3) or H != W or H % 32:
            raise ValueError(f"expected [B, 1|3, S, S] with S a multiple of 32, got {tuple(x.shape)}")
        xin = x.contiguous().float()
        if self.compute_dtype != torch.float32:
            xl = torch.empty(xin.shape, dtype=self.compute_dtype, device=dev)
            L.tc_cast(xin.data_ptr(), xl.data_ptr(), xin.numel(), TC_F32, self._tc_dtype(), stream)
            xin = xl
        self._tok_logits = token_logits
        out_var = _forward(self, G, xin, B, in_ch, H)
        self._tok_logits = False
        logits = out_var.data if token_logits else out_var.data.view(B, self.num_classes, H, W)
        if self.compute_dtype != torch.float32 and not token_logits:
            lf = torch.empty(logits.shape, dtype=torch.float32, device=dev)
            L.tc_cast(logits.data_ptr(), lf.data_ptr(), logits.numel(), self._tc_dtype(), TC_F32, stream)
            logits = lf
        if self.training:
            nbt = getattr(self, "_nbt_flat", None)
            bns = [m for m in self.modules() if isinstance(m, nn.BatchNorm2d)]
            if nbt is not None and [m.num_batches_tracked.data_ptr() for m in bns] == self._nbt_ptrs:
                nbt += 1                                        # all counters, one launch
            else:                                               # buffers were replaced since (_apply / load): per module
                for m in bns:
                    m.num_batches_tracked += 1
        self.last_launches = G.n_launch
        return logits, G, out_var

    def late_gradient_offset(self) -> int:
        """First element of the flat arenas that belongs to the bridge / decoders: parameters are laid out in registration
        order (backbone, bridge, decoder_3..0), so [offset, end) is complete once backward has passed the "encoder_done" mark."""
        late = min(off for name, (off, _) in self._index.items() if not name.startswith("backbone."))
        assert all(off < late for name, (off, _) in self._index.items() if name.startswith("backbone."))
        return late

    def gradient_pieces(self):
        """The stops of a split backward sweep and the arena ranges whose gradients are complete at each, in the order they are reached:
        bridge + decoders at "encoder_done", then stage 4 (63 % of the encoder's parameters), stage 3, and the rest at the end (None).
        A multi-GPU step sends each piece while the sweep continues (train.GraphedStep); only the small last piece is exposed."""
        late, pieces, taken = self.late_gradient_offset(), [], []          # (the flat arenas exist: called after a first step)
        pieces.append(("encoder_done", [(late, self._gflat.numel())]))
        # stages 4 and 3 leave TOGETHER at "stage3_done" by default: their 72 MB travel under the sweep through stages 2 and 1 either way
        # (~2.5 ms against 0.5-1.2 ms of ring time), and every stop is one more graph launch per step (+0.13 ms at N = 1).
        # TC_GRAD_LEGS=split: one piece per stage, as in round 3.
        split = os.environ.get("TC_GRAD_LEGS", "merged") == "split"
        acc = []
        for stage in (4, 3):
            rs = []
            for pre in (f"backbone.patch_embed_stage{stage}.", f"backbone.mhca_stage{stage}."):
                offs = [(o, o + (math.prod(sh) + 7) // 8 * 8) for n, (o, sh) in self._index.items() if n.startswith(pre)]
                lo, hi = min(a for a, _ in offs), max(b for _, b in offs)
                assert all(n.startswith(pre) for n, (o, sh) in self._index.items() if lo <= o < hi), pre      # one module, one contiguous range
                rs.append((lo, hi))
            taken += rs
            acc += rs
            if split or stage == 3:
                pieces.append((f"stage{stage}_done", acc))
                acc = []
        rest, at = [], 0
        for lo, hi in sorted(taken):
            if lo > at:
                rest.append((at, lo))
            at = hi
        if at < late:
            rest.append((at, late))
        pieces.append((None, rest))
        return pieces

    def _backward_continue(self, G: Graph, until: Optional[str]):
        """Next leg of a backward that was stopped at a mark: to the mark `until`, or (None) to the end."""
        if G.backward(until):
            self._attach_grads()

    def _backward_finish(self, G: Graph):
        """Second half of a backward that was stopped at a mark."""
        G.backward()
        self._attach_grads()

    def _backward(self, G: Graph, out_var: Var, dlogits: torch.Tensor, until: Optional[str] = None):
        L = lib()
        first = next((p for p in self._uniq_params if id(p) in self._used_views), None)
        if first is not None and first.grad is None:
            self._gflat.zero_()                     # zero_grad(set_to_none=True) semantics: start from zero
        root = out_var.root
        if not out_var.is_whole:
            # token-major logits with padded rows (Graph.ln_cls(pad_rows=True)): the gradient mirrors that layout.  A caller that allocated
            # dlogits as the column slice of such a buffer (train.GraphedStep does) hands it over as it is; anything else is copied in.
            es = root.data.element_size()
            if (dlogits.dtype == self.compute_dtype and dlogits.dim() == 2 and dlogits.stride() == out_var.data.stride() and dlogits.storage_offset() == 0
                    and dlogits.untyped_storage().nbytes() >= root.data.numel() * es):
                full = torch.as_strided(dlogits, root.data.shape, root.data.stride())
            else:
                full = torch.empty_like(root.data)
                out_var.apply_path(full).copy_(dlogits.view(out_var.rows, out_var.cols))
            root.grad_t, root.whole_written = full, True
            if G.backward(until):
                self._attach_grads()
            return
        d = dlogits.contiguous()
        if d.dtype != self.compute_dtype:
            dl = torch.empty(d.shape, dtype=self.compute_dtype, device=d.device)
            L.tc_cast(d.data_ptr(), dl.data_ptr(), d.numel(), TC_F32, self._tc_dtype(), G.stream)
            d = dl
        out_var.root.grad_t = d.view(out_var.rows, out_var.cols)
        out_var.root.whole_written = True
        if G.backward(until):
            self._attach_grads()


TransCeption = MSTransception


# ----------------------------------------------------------------------------------------------------------
# the forward pass as engine calls (reference lines cited per block; index-level spec in SURVEY.md Appendix C)
# ----------------------------------------------------------------------------------------------------------
def _lin(M: MSTransception, G: Graph, name: str, bias: bool = True):
    off, shape = M._index[name + ".weight"]
    W = M._P(G, name + ".weight", (shape[0], math.prod(shape[1:])))
    b = M._P(G, name + ".bias") if bias else None
    return W, b


_PAIR_GROUPED = os.environ.get("TC_PAIR_GROUPED", "1") != "0"       # A/B switch of _lin_pair


def _lin_pair(M: MSTransception, G: Graph, a: str, b: str):
    """(W, bias, stride) of two same-shaped Linear / 1x1 layers as TWO WEIGHT GROUPS of one launch (Graph.grouped(2, stride)) -- or None when
    their parameters do not sit at one constant, 16-byte-aligned distance in the flat arenas."""
    (oa, sa), (ob, sb) = M._index[a + ".weight"], M._index[b + ".weight"]
    (ba, sba), (bb, sbb) = M._index[a + ".bias"], M._index[b + ".bias"]
    stride = ob - oa
    if sa != sb or sba != sbb or bb - ba != stride or stride <= 0 or stride % 8 or oa % 8 or ba % 8 or not _PAIR_GROUPED:
        return None
    W, bias = _lin(M, G, a)
    if G.record:
        for nm in (b + ".weight", b + ".bias"):
            M._used_views[M._pid[nm]] = M._index[nm]
    return P(W.data, W.grad, stride), P(bias.data, bias.grad, stride), stride


def _ln(M, G, x, name, eps=1e-5, act=ACT_NONE, out=None):
    return G.layernorm(x, M._P(G, name + ".weight"), M._P(G, name + ".bias"), eps, act, out)


def _bn_shift(M, name):
    """Shift of the BatchNorm statistics the producing GEMM leaves behind (engine.Graph.linear(bn_shift=...)): the layer's running
    mean -- any per-channel constant gives the same mean / variance, one near the batch mean keeps the squared sums small."""
    return M.get_submodule(name).running_mean


def _bn(M, G, x, name, act, residual=None, out=None):
    holder = M.get_submodule(name)
    return G.batchnorm(x, M._P(G, name + ".weight"), M._P(G, name + ".bias"), holder.running_mean, holder.running_var, act,
                       residual, out)


FUSED_MIXFFN = os.environ.get("TC_FUSED_MIXFFN", "1") != "0"


def _mixffn_site(M, G, x, name, B, H, W, residual, out=None, pre_ln=None) -> dict:
    return dict(pre_ln=pre_ln, x=x, fc1=_lin(M, G, name + ".fc1"), dw=(M._P(G, name + ".dwconv.dwconv.weight"), M._P(G, name + ".dwconv.dwconv.bias")),
                ln=(M._P(G, name + ".norm1.weight"), M._P(G, name + ".norm1.bias")), fc2=_lin(M, G, name + ".fc2"), geo=(B, H, W),
                residual=residual, out=out)


def _mixffn(M, G, x, name, B, H, W, residual, out=None, pre_ln=None):
    """MixFFN_skip, MSTr.py:889-902 (fc1 evaluated once): fc2(GELU(LN(dw3x3(h) + h))) + residual.
    pre_ln = (norm name, eps): the site's input is LayerNorm(x) -- the block's norm2 -- which the tiled kernels apply themselves."""
    pl = (M._P(G, pre_ln[0] + ".weight"), M._P(G, pre_ln[0] + ".bias"), pre_ln[1]) if pre_ln is not None else None
    if FUSED_MIXFFN and not G.use_streams and x.data.is_contiguous():
        return G.mixffn([_mixffn_site(M, G, x, name, B, H, W, residual, out, pl)])[0]     # 3 + 3 launches (engine.Graph.mixffn)
    if pl is not None:
        x = G.layernorm(x, *pl)
    h = G.linear(x, *_lin(M, G, name + ".fc1"))                 # B = images per weight group (G.ngroups groups are stacked)
    d = G.dwconv(h, M._P(G, name + ".dwconv.dwconv.weight"), M._P(G, name + ".dwconv.dwconv.bias"), B, H, W, 3, 1, True)
    a = _ln(M, G, d, name + ".norm1", act=ACT_GELU)
    return G.linear(a, *_lin(M, G, name + ".fc2"), out=out, residual=residual)


def _mixffn_plain(M, G, x, name, B, H, W, residual):
    """MixFFN (token_mlp_mode = "mix"), MSTr.py:35-46: fc2(GELU(dw3x3(fc1(x)))) + residual -- an ablation variant, run op by op."""
    h = G.linear(x, *_lin(M, G, name + ".fc1"))
    d = G.dwconv(h, M._P(G, name + ".dwconv.dwconv.weight"), M._P(G, name + ".dwconv.dwconv.bias"), B, H, W, 3, 1, False)
    return G.linear(G.gelu(d), *_lin(M, G, name + ".fc2"), residual=residual)


def _eff_attention(M, G, n1: Var, name: str, B: int, N: int, residual: Optional[Var] = None, ln=None):
    """EfficientAttention with one head, MSTr.py:106-143 (Appendix C.6), + the block's residual add.  Returns (tx, LayerNorm(tx) | None):
    ln = (norm name, eps) asks for the block's norm2 from the same launch as `reprojection` (_proj_ln)."""
    C = n1.cols
    rows = B * N
    kqv = G.new(rows, 3 * C, covered=True)         # k, q, v gradients (softmax / softmax / bmm backward) cover it
    k, q, v = kqv.colslice(0, C), kqv.colslice(C, 2 * C), kqv.colslice(2 * C, 3 * C)
    lk, lq, lv = _lin(M, G, name + ".keys"), _lin(M, G, name + ".queries"), _lin(M, G, name + ".values")
    if MULTI_QKV and C % 64 == 0 and G.ngroups == 1:
        G.linear_multi(n1, [lk[0], lq[0], lv[0]], [lk[1], lq[1], lv[1]], kqv)       # the three 1x1 convs in one batched GEMM
    else:
        G.linear(n1, *lk, out=k)
        G.linear(n1, *lq, out=q)
        G.linear(n1, *lv, out=v)
    ksm = G.softmax(k, B, 0)                       # over tokens, per channel
    qsm = G.softmax(q, 1, 1)                       # over channels, per token
    ctx = G.new(B * C, C)
    G.bmm(ksm, v, ctx, C, C, N, 1, 0, nb1=B, sA=(N * ksm.ld, 0), sB=(N * v.ld, 0), sC=(C * C, 0))
    att = G.new(rows, C)
    G.bmm(qsm, ctx, att, N, C, C, 0, 0, nb1=B, sA=(N * qsm.ld, 0), sB=(C * C, 0), sC=(N * C, 0))
    return _proj_ln(M, G, att, name + ".reprojection", residual, ln)


def _eff_block(M, G, t: Var, name: str, B: int, H: int, W: int) -> Var:
    """EfficientTransformerBlock, MSTr.py:164-173."""
    a = name + ".attn"
    blk = ((M._P(G, name + ".norm1.weight"), M._P(G, name + ".norm1.bias")), _lin(M, G, a + ".keys"), _lin(M, G, a + ".queries"),
           _lin(M, G, a + ".values"), _lin(M, G, a + ".reprojection"))
    if G.effatt_supported(t, tuple(p for pair in blk for p in pair)):
        tx = G.eff_attention_block(t, *blk, B, H * W)
        if M.token_mlp_mode == "mix":
            return _mixffn_plain(M, G, _ln(M, G, tx, name + ".norm2"), name + ".mlp", B, H, W, residual=tx)
        return _mixffn(M, G, tx, name + ".mlp", B, H, W, residual=tx, pre_ln=(name + ".norm2", 1e-5))
    tx, n2 = _eff_attention(M, G, _ln(M, G, t, name + ".norm1"), name + ".attn", B, H * W, residual=t, ln=(name + ".norm2", 1e-5))
    if M.token_mlp_mode == "mix":
        return _mixffn_plain(M, G, n2, name + ".mlp", B, H, W, residual=tx)
    return _mixffn(M, G, n2, name + ".mlp", B, H, W, residual=tx)


def _ripm(M, G, m: Var, name: str, B: int, side: int, npath: int = 3, pool: bool = True) -> Tuple[Var, int]:
    """Patch_Embed_stage of DWConv2d_BN, MSTr.py:725-732, 355-362.  Returns the npath chained maps stacked [npath*rows, C] (npath = 3 and a pooling
    first step everywhere but in the first stage of MSViT_4Stages: two paths, no pooling)."""
    so = (side - 1) // 2 + 1 if pool else side
    rows = B * so * so
    stack = G.new(npath * rows, m.cols)
    if npath == 3 and pool and G.ripm_supported(m):              # a step per launch, BatchNorm + Hardswish applied by the consumer
        steps = []
        for i in range(3):
            pre = f"{name}.patch_embeds.{i}.patch_conv"
            holder = M.get_submodule(pre + ".bn")
            steps.append(dict(dw=M._P(G, pre + ".dwconv.weight"), pw=_lin(M, G, pre + ".pwconv", bias=False)[0], gamma=M._P(G, pre + ".bn.weight"),
                              beta=M._P(G, pre + ".bn.bias"), rmean=holder.running_mean, rvar=holder.running_var))
        return stack, G.ripm_stage(m, steps, B, side, stack)
    x = m
    for i in range(npath):
        stride = 2 if (i == 0 and pool) else 1
        pre = f"{name}.patch_embeds.{i}.patch_conv"
        y = G.dwconv(x, M._P(G, pre + ".dwconv.weight"), None, B, side, side, 3, stride)
        side = (side - 1) // stride + 1
        z = G.linear(y, *_lin(M, G, pre + ".pwconv", bias=False), bn_shift=_bn_shift(M, pre + ".bn"))
        x = _bn(M, G, z, pre + ".bn", ACT_HSWISH, out=stack.rowslice(i * rows, (i + 1) * rows))
    return stack, side


def _resblock(M, G, x: Var, name: str, B: int, side: int, out: Var) -> Var:
    """ResBlock, MSTr.py:1042-1050."""
    f = _bn(M, G, G.linear(x, *_lin(M, G, name + ".conv1.conv", bias=False), bn_shift=_bn_shift(M, name + ".conv1.bn")), name + ".conv1.bn", ACT_HSWISH)
    f = G.dwconv(f, M._P(G, name + ".dwconv.weight"), None, B, side, side, 3, 1)
    f = _bn(M, G, f, name + ".norm", ACT_HSWISH)
    f = G.linear(f, *_lin(M, G, name + ".conv2.conv", bias=False), bn_shift=_bn_shift(M, name + ".conv2.bn"))
    return _bn(M, G, f, name + ".conv2.bn", ACT_NONE, residual=x, out=out)


MANY_MIXFFN = os.environ.get("TC_MANY_MIXFFN", "1") != "0"
MULTI_QKV = os.environ.get("TC_MULTI_QKV", "1") != "0"
MULTI_CRPE = os.environ.get("TC_MULTI_CRPE", "1") != "0"
FUSED_FACTOR_ATT = os.environ.get("TC_FACTOR_ATT_FUSED", "1") != "0"
SHUFFLE_IN_LN = os.environ.get("TC_SHUFFLE_IN_LN", "1") != "0"


def _factor_att(M, G, n: Var, blk: str, enc: str, B: int, side: int, residual: Optional[Var] = None, ln=None):
    """FactorAtt_ConvRelPosEnc + ConvRelPosEnc, MSTr.py:852-886, 801-823 (Appendix C.1-2), + residual add.
    ln = (norm name, eps): also return LayerNorm(result) -- the block's norm2 -- from the same launch as `proj` where the library has it
    (engine.Graph.linear_ln); returns (t2, LayerNorm(t2) or None)."""
    C, N = n.cols, side * side
    Bt = B * G.ngroups                              # images of all stacked weight groups
    rows, h, Ch = Bt * N, HEADS, n.cols // HEADS
    if FUSED_FACTOR_ATT and MULTI_CRPE and Ch % 8 == 0 and G.mhca_att_supported(n, N, h, list(CRPE_WINDOW)):     # qkv + crpe + attention core: one launch
        o = G.mhca_attention(n, *_lin(M, G, blk + ".factoratt_crpe.qkv"), [M._P(G, f"{enc}.crpe.conv_list.{i}.weight") for i in range(3)],
                             [M._P(G, f"{enc}.crpe.conv_list.{i}.bias") for i in range(3)], B, side, h, Ch ** -0.5, list(CRPE_WINDOW))
        return _proj_ln(M, G, o, blk + ".factoratt_crpe.proj", residual, ln)
    qkv = G.linear(n, *_lin(M, G, blk + ".factoratt_crpe.qkv"), out=G.new(n.rows, 3 * C, covered=FUSED_FACTOR_ATT))
    q, k, v = qkv.colslice(0, C), qkv.colslice(C, 2 * C), qkv.colslice(2 * C, 3 * C)
    convv = G.new(rows, C)
    c0, xs, outs, wts, bss, kss = 0, [], [], [], [], []
    for i, (ksz, nh) in enumerate(CRPE_WINDOW):
        w = nh * Ch
        xs.append(v.colslice(c0, c0 + w)); outs.append(convv.colslice(c0, c0 + w)); kss.append(ksz)
        wts.append(M._P(G, f"{enc}.crpe.conv_list.{i}.weight")); bss.append(M._P(G, f"{enc}.crpe.conv_list.{i}.bias"))
        c0 += w
    if MULTI_CRPE:
        G.dwconv_multi(xs, wts, bss, (B, side, side), kss, outs)           # the three window sizes in one launch
    else:
        for x_, w_, b_, k_, o_ in zip(xs, wts, bss, kss, outs):
            G.dwconv(x_, w_, b_, B, side, side, k_, 1, False, out=o_)
    if FUSED_FACTOR_ATT and Ch % 8 == 0 and 16 * N * Ch + 8 * Ch * Ch + 1024 <= 150 * 1024:     # a head's q, k, v, do tiles fit LDS in fp32
        o = G.factor_att_core(q, k, v, convv, Bt, N, h, Ch ** -0.5)
    else:                                           # unfused composition: larger inputs (384^2: 2304 tokens at stage 2), A/B tests
        ksm = G.softmax(k, Bt, 0)
        ctx = G.new(Bt * h * Ch, Ch)
        G.bmm(ksm, v, ctx, Ch, Ch, N, 1, 0, nb1=Bt, nb2=h, sA=(N * C, Ch), sB=(N * 3 * C, Ch), sC=(h * Ch * Ch, Ch * Ch))
        fa = G.new(rows, C)
        G.bmm(q, ctx, fa, N, Ch, Ch, 0, 0, nb1=Bt, nb2=h, sA=(N * 3 * C, Ch), sB=(h * Ch * Ch, Ch * Ch), sC=(N * C, Ch))
        o = G.fma3(fa, q, convv, Ch ** -0.5)
    return _proj_ln(M, G, o, blk + ".factoratt_crpe.proj", residual, ln)


def _proj_ln(M, G, o: Var, proj: str, residual: Optional[Var], ln, out: Optional[Var] = None, ln_out: Optional[Var] = None):
    """(t, LayerNorm(t) | None) with t = proj(o) + residual: one launch for both where the library supports the width (tc_linear_ln_fwd)."""
    W, b = _lin(M, G, proj)
    if ln is not None and b is not None:
        g, beta = M._P(G, ln[0] + ".weight"), M._P(G, ln[0] + ".bias")
        if G.linear_ln_supported(o, W, residual, b, g, beta):
            return G.linear_ln(o, W, b, residual, g, beta, ln[1], out=out, ln_out=ln_out)
    t = G.linear(o, W, b, residual=residual, out=out)
    return t, (_ln(M, G, t, ln[0], ln[1], out=ln_out) if ln is not None else None)


def _mhca_block(M, G, t: Var, blk: str, enc: str, B: int, side: int, out: Optional[Var] = None) -> Var:
    """MHCABlock, MSTr.py:935-946: shared cpe (dw3x3 + identity) in every block, LN eps 1e-6."""
    if G.dw_ln_supported(t):                                     # cpe + norm1: one launch
        t1, n1 = G.dw_ln(t, M._P(G, enc + ".cpe.proj.weight"), M._P(G, enc + ".cpe.proj.bias"), M._P(G, blk + ".norm1.weight"),
                         M._P(G, blk + ".norm1.bias"), B, side, side, 1e-6)
    else:
        t1 = G.dwconv(t, M._P(G, enc + ".cpe.proj.weight"), M._P(G, enc + ".cpe.proj.bias"), B, side, side, 3, 1, True)
        n1 = _ln(M, G, t1, blk + ".norm1", 1e-6)
    t2, n2 = _factor_att(M, G, n1, blk, enc, B, side, residual=t1, ln=(blk + ".norm2", 1e-6))     # proj + skip + norm2: one launch
    return _mixffn(M, G, n2, blk + ".mlp", B, side, side, residual=t2, out=out)


def _coord_att(M, G, x: Var, name: str, B: int, side: int, out: Var) -> Var:
    """CoordAtt (IFF), MSTr.py:1322-1348."""
    pooled = G.coord_pool(x, B, side, side)
    y = G.linear(pooled, *_lin(M, G, name + ".conv1"), bn_shift=_bn_shift(M, name + ".bn1"))
    y = _bn(M, G, y, name + ".bn1", ACT_COORD)
    att = G.new(2 * B * side, x.cols)
    half = B * side
    pair = _lin_pair(M, G, name + ".conv_h", name + ".conv_w")
    if pair is not None:                                  # the two gate convolutions: one launch, two weight groups (rows [0, half) / [half, 2 half))
        with G.grouped(2, pair[2]):
            G.linear(y, pair[0], pair[1], out=att, act=ACT_SIGMOID)
    else:
        G.linear(y.rowslice(0, half), *_lin(M, G, name + ".conv_h"), out=att.rowslice(0, half), act=ACT_SIGMOID)
        G.linear(y.rowslice(half, 2 * half), *_lin(M, G, name + ".conv_w"), out=att.rowslice(half, 2 * half), act=ACT_SIGMOID)
    gated = G.coord_gate(x, att, B, side, side)
    return G.linear(gated, *_lin(M, G, name + ".conv_in_out"), out=out)


def _mhca_stage(M, G, stack: Var, name: str, layers: int, B: int, side: int, out: Var, npath: int = 3) -> Var:
    """MHCA_stage, MSTr.py:1412-1441.  `stack` holds the three RIPM maps one after the other ([3*rows, C]); the three MB
    encoders have identical shapes and their weights sit at a constant stride in the flat arena, so every kernel of the
    MB blocks runs ONCE for all three paths (grouped weights) instead of three times.  The four branch outputs are written
    side by side into one [rows, 4C] buffer (no torch.cat)."""
    C = stack.cols
    rows = B * side * side
    cat = G.new(rows, (npath + 1) * C)
    gs = M._path_stride(name, npath)
    enc = f"{name}.mhca_blks.0"
    stg = name[-1]
    with G.parallel(2) as par:
        with par.branch(0):
            G.segment("mb" + stg)
            with G.grouped(npath, gs):
                t = stack
                for l in range(layers):
                    t = _mhca_block(M, G, t, f"{enc}.MHCA_layers.{l}", enc, B, side, cat.colslice(C, (npath + 1) * C) if l == layers - 1 else None)
        with par.branch(1):
            G.segment("res" + stg)
            _resblock(M, G, stack.rowslice(0, rows), name + ".InvRes", B, side, cat.colslice(0, C))
    G.segment("iff" + stg)
    if M.concat == "coord":
        return _coord_att(M, G, cat, name + ".aggregate", B, side, out)
    if M.concat == "3d":                                                         # Conv3d_BN_concat, MSTr.py:447-462
        agg = name + ".aggregate"
        off, shape = M._index[agg + ".interact_concat.0.weight"]                 # [O, C, 4, 1, 1]
        Wp = G.permuted_weight(M._P(G, agg + ".interact_concat.0.weight", (shape[0], C * 4)), shape[0], C, 4)
        z = G.relu(G.linear(cat, Wp, M._P(G, agg + ".interact_concat.0.bias")))
        return _bn(M, G, z, agg + ".bn", ACT_NONE, out=out)
    if M.concat in ("cam", "cam_fact"):                                          # Conv3d_BN_channel_attention_concat, MSTr.py:642-668
        agg = name + ".aggregate"
        N = side * side
        # BatchNorm3d over (image, path, token) per channel = BatchNorm over the rows of the [4 * rows, C] view of the concatenation (a row of the
        # concatenation is path-major).  The reference also runs bn3d on the partial stacks of 1, 2, 3 paths and discards the results (:651-655):
        # only its running statistics see those passes; they are not reproduced.
        x3 = _bn(M, G, cat.reshape(4 * rows, C), agg + ".bn3d", ACT_NONE).reshape(rows, 4 * C)
        if M.concat == "cam":
            x3 = G.cam(x3, M._P(G, agg + ".channelAttention.gamma"), B, N)
        else:
            # CAM_Factorized_Module, MSTr.py:528-568: the 4 N tokens of an image (every path's) through one factorized attention -- softmax of k
            # over the tokens, k^T v per head, q times that, scaled -- then proj and gamma * out + x.  Sums over tokens do not care about their
            # order, so the (token, path) rows of the concatenation's [4 rows, C] view stand for the reference's (path, token) ones.
            ca = agg + ".channelAttention"
            t = x3.reshape(4 * rows, C)
            N4, Ch = 4 * N, C // HEADS
            qkv = G.linear(t, *_lin(M, G, ca + ".qkv"))
            q, k, v = qkv.colslice(0, C), qkv.colslice(C, 2 * C), qkv.colslice(2 * C, 3 * C)
            ksm = G.softmax(k, B, 0)
            ctx = G.new(B * HEADS * Ch, Ch)
            G.bmm(ksm, v, ctx, Ch, Ch, N4, 1, 0, nb1=B, nb2=HEADS, sA=(N4 * C, Ch), sB=(N4 * 3 * C, Ch), sC=(HEADS * Ch * Ch, Ch * Ch))
            fa = G.new(4 * rows, C)
            G.bmm(q, ctx, fa, N4, Ch, Ch, 0, 0, nb1=B, nb2=HEADS, sA=(N4 * 3 * C, Ch), sB=(HEADS * Ch * Ch, Ch * Ch), sC=(N4 * C, Ch), alpha=Ch ** -0.5)
            o = G.linear(fa, *_lin(M, G, ca + ".proj"))
            x3 = G.gamma_residual(o, t, M._P(G, ca + ".gamma")).reshape(rows, 4 * C)
        x3 = _bn(M, G, x3.reshape(4 * rows, C), agg + ".bn3d", ACT_NONE).reshape(rows, 4 * C)
        off, shape = M._index[agg + ".interact_concat.0.weight"]                 # [O, C, 4, 1, 1]
        Wp = G.permuted_weight(M._P(G, agg + ".interact_concat.0.weight", (shape[0], C * 4)), shape[0], C, 4)
        z = G.gelu(G.linear(x3, Wp, M._P(G, agg + ".interact_concat.0.bias")))
        return _bn(M, G, z, agg + ".bn", ACT_NONE, out=out)
    if M.concat == "cbam":                                                       # CBAMBlock, MSTr.py:1198-1211
        agg = name + ".aggregate"
        holder = M.get_submodule(agg)
        N = side * side
        pooled = G.chan_pool2(cat, B, N)                                         # [2B, 4C]: max rows, then mean rows -- the shared MLP runs on both at once
        hid = G.relu(G.linear(pooled, *_lin(M, G, agg + ".ca.se.0", bias=False)))
        ca = G.linear(G.add(hid.rowslice(0, B), hid.rowslice(B, 2 * B)), *_lin(M, G, agg + ".ca.se.2", bias=False), act=ACT_SIGMOID)   # se(max) + se(avg): the last layer is linear
        o = G.chan_gate(cat, ca, B, N)
        if holder.use_sa and M.inter in ("res", "out"):                           # CBAMBlock_casa, MSTr.py:1243-1251: "res" reads the ResBlock branch (x[0])
            k = holder.sa_ker
            src = o if M.inter == "out" else cat.colslice(0, C)
            g = G.sa_conv(G.pix_stats(src), M._P(G, agg + ".sa.conv.weight"), M._P(G, agg + ".sa.conv.bias"), B, side, side, k)
            o = G.pix_gate(o, g)
        Wc, _ = _lin(M, G, agg + ".conv2d_bn_act.0", bias=False)
        z = G.new(rows, Wc.data.shape[0])
        G.linear(o, Wc, None, out=z)                                             # conv(out + residual) = conv(out) + conv(x)
        G.linear(cat, Wc, None, out=z, accumulate=True)
        return _bn(M, G, z, agg + ".conv2d_bn_act.1", ACT_RELU, out=out)
    if M.concat == "skn":                                                        # SK_Block, MSTr.py:1076-1107
        agg = name + ".aggregate"
        N = side * side
        pooled = G.chan_pool(cat, B, N)                                          # [B, 4C]: S = mean(sum_k x_k) = sum_k mean(x_k)
        Wf, bf = _lin(M, G, agg + ".fc")
        Z = G.new(B, Wf.data.shape[0])
        nbr = cat.cols // C                                                      # branch maps: four (three in the first stage of MSViT_4Stages)
        for k in range(nbr):                                                     # fc(S): the same weight on the column blocks, accumulated
            G.linear(pooled.colslice(k * C, (k + 1) * C), Wf, bf if k == 0 else None, out=Z, accumulate=k > 0)
        A = G.new(B, nbr * C)
        for k in range(nbr):
            G.linear(Z, *_lin(M, G, f"{agg}.fcs.{k}"), out=A.colslice(k * C, (k + 1) * C))
        att = G.softmax(A.reshape(B * nbr, C), B, 0).reshape(B, nbr * C)         # softmax over the paths, per image and channel
        gated = G.chan_gate(cat, att, B, N)
        Wc, bc = _lin(M, G, agg + ".conv_bn_ac.0")
        z = G.new(rows, Wc.data.shape[0])
        for k in range(nbr):                                                     # conv(sum_k a_k x_k)
            G.linear(gated.colslice(k * C, (k + 1) * C), Wc, bc if k == 0 else None, out=z, accumulate=k > 0)
        return _bn(M, G, G.relu(z), agg + ".conv_bn_ac.2", ACT_NONE, out=out)
    if M.concat == "se":                                                         # SE_Block, MSTr.py:584-593
        agg = name + ".aggregate"
        y = G.relu(G.linear(G.chan_pool(cat, B, side * side), *_lin(M, G, agg + ".excitation.0", bias=False)))
        y = G.linear(y, *_lin(M, G, agg + ".excitation.2", bias=False), act=ACT_SIGMOID)
        z = G.linear(G.chan_gate(cat, y, B, side * side), *_lin(M, G, agg + ".conv"), bn_shift=_bn_shift(M, agg + ".bn"))
        return _bn(M, G, z, agg + ".bn", ACT_RELU, out=out)
    y = G.linear(cat, *_lin(M, G, name + ".aggregate.conv", bias=False), bn_shift=_bn_shift(M, name + ".aggregate.bn"))   # "normal": Conv2d_BN with Hardswish, MSTr.py:1384-1390
    return _bn(M, G, y, name + ".aggregate.bn", ACT_HSWISH, out=out)


def _channel_att(M, G, n: Var, X: Optional[Var], name: str, B: int, ntok: List[int], R: List[int], N6: int) -> Var:
    """M_EfficientChannelAtten, MSTr.py:2309-2353: needs each image's tokens contiguous (flat [C, N] re-view), so the
    q/k/v GEMMs scatter their rows from the stage-major buffer into image-major ones, and proj gathers back."""
    Cd = 64
    offs = [sum(ntok[:i]) for i in range(4)]
    imgs, many = {}, []
    for key in ("k", "q", "v"):
        buf = G.new(B * N6, Cd, covered=True)
        Wb = _lin(M, G, f"{name}.{key}")
        for s in range(4):
            xs_, os_ = n.rowslice(R[s], R[s] + ntok[s]), buf.rowslice(offs[s], offs[s] + ntok[s])
            if MANY_MIXFFN and not G.use_streams:
                many.append((xs_, Wb[0], Wb[1], os_, None, (B, ntok[s] * Cd, N6 * Cd, 0)))
            else:
                G.linear(xs_, *Wb, out=os_, batch=(B, ntok[s] * Cd, N6 * Cd, 0))
        imgs[key] = buf.reshape(B * Cd, N6)
    if many:
        G.linear_many(many)                                     # 3 projections x 4 scales: one launch (and two for the gradients)
    ksm = G.softmax(imgs["k"], 1, 1)
    qsm = G.softmax(imgs["q"], B, 0)
    ctx = G.new(B * Cd, Cd)
    G.bmm(ksm, imgs["v"], ctx, Cd, Cd, N6, 0, 1, nb1=B, sA=(Cd * N6, 0), sB=(Cd * N6, 0), sC=(Cd * Cd, 0))
    Op = G.new(B * Cd, N6)
    G.bmm(ctx, qsm, Op, Cd, N6, Cd, 1, 0, nb1=B, sA=(Cd * Cd, 0), sB=(Cd * N6, 0), sC=(Cd * N6, 0))
    o_img = G.transpose(Op, B)                                  # [B*N6, 64], image-major
    tx1 = G.new(B * N6, Cd, covered=True)                      # (see _self_att)
    Wp = _lin(M, G, name + ".proj")
    many = []
    for s in range(4):
        xs_, os_ = o_img.rowslice(offs[s], offs[s] + ntok[s]), tx1.rowslice(R[s], R[s] + ntok[s])
        rs_ = X.rowslice(R[s], R[s] + ntok[s]) if X is not None else None
        bt = (B, N6 * Cd, ntok[s] * Cd, ntok[s] * Cd)
        if MANY_MIXFFN and not G.use_streams:
            many.append((xs_, Wp[0], Wp[1], os_, rs_, bt))
        else:
            G.linear(xs_, *Wp, out=os_, residual=rs_, batch=bt)
    if many:
        G.linear_many(many)
    return tx1


def _scale_reduce(M, G, n: Var, name: str, B: int, sides: List[int], ntok: List[int], R: List[int], extra=None):
    """Scale_reduce, MSTr.py:2225-2249 (Appendix C.4): k=s patchify convs + channel de-interleave + LN -> [B*Nk, 64].
    extra = (x, W, b, post_scale): one more independent Linear (the attention's q projection) that shares the launch of the three
    patchified convolutions; its output is returned second."""
    Cd = 64
    Pn = sides[3] * sides[3]
    Nk = Pn * 8 + ntok[3]
    red = G.new(B * Nk, Cd)                                     # image-major K/V source
    pitems = [(R[s] * Cd, sides[s] * sides[s] * Cd * MULT[s], B, sides[s], sides[s], Cd * MULT[s], SR_K[s]) for s in range(3)]
    merged = MANY_MIXFFN and not G.use_streams and G.ngroups == 1     # the layout moves of the three scales share launches
    cols = G.patchify_many(n, pitems) if merged else [G.patchify(n, *it) for it in pitems]
    lins = [_lin(M, G, f"{name}.sr{s}") for s in range(3)]
    xo = None
    if MANY_MIXFFN and not G.use_streams and G.ngroups == 1:
        items = [(cols[s], lins[s][0], lins[s][1], G.new(cols[s].rows, lins[s][0].data.shape[0]), None) for s in range(3)]
        if extra is not None:
            items.append((extra[0], extra[1], extra[2], G.new(extra[0].rows, extra[1].data.shape[0]), None, None, extra[3]))
        outs = G.linear_many(items)
        os_, xo = outs[:3], (outs[3] if extra is not None else None)
    else:
        os_ = [G.linear(cols[s], *lins[s]) for s in range(3)]
        if extra is not None:
            xo = G.linear(extra[0], extra[1], extra[2], post_scale=extra[3])
    roffs = [0, MULT[0] * Pn, (MULT[0] + MULT[1]) * Pn, (MULT[0] + MULT[1] + MULT[2]) * Pn]
    if merged:
        G.sr_gather([(os_[s], roffs[s] * Cd, Nk * Cd, B, Pn, Cd, MULT[s]) for s in range(3)],
                    (n, R[3] * Cd, ntok[3] * Cd, roffs[3] * Cd, Nk * Cd, B, ntok[3], Cd), red)
    else:
        for s in range(3):
            G.sr_deinterleave(os_[s], red, roffs[s] * Cd, Nk * Cd, B, Pn, Cd, MULT[s])
        G.copy_rows(n, R[3] * Cd, ntok[3] * Cd, red, roffs[3] * Cd, Nk * Cd, B, ntok[3], Cd)
    rn = _ln(M, G, red, name + ".norm")
    return rn if extra is None else (rn, xo)


LOG2E = 1.4426950408889634


def _self_att(M, G, n: Var, X: Optional[Var], name: str, B: int, sides: List[int], ntok: List[int], R: List[int], N6: int, ln=None):
    """M_EfficientSelfAtten, MSTr.py:2267-2292: one head, d = 64, keys/values from the reduced token set.  Returns (tx1, LayerNorm(tx1) | None):
    ln = (norm name, eps) asks for the layer's norm2 from the same launch as `proj` (_proj_ln)."""
    Cd = 64
    Nk = sides[3] * sides[3] * 8 + ntok[3]
    # 16-bit fused path: the q projection stores q * scale * log2(e), rounded once from its fp32 accumulator (the attention kernels then
    # spend no multiply per score and Q is not rounded a second time)
    pre = G.use_fused_attention and G.dtype != torch.float32
    rn, q = _scale_reduce(M, G, n, name + ".scale_reduce", B, sides, ntok, R,
                          extra=(n, *_lin(M, G, name + ".q"), (Cd ** -0.5) * LOG2E if pre else None))
    kv = G.linear(rn, *_lin(M, G, name + ".kv"))
    k, v = kv.colslice(0, Cd), kv.colslice(Cd, 2 * Cd)
    att = G.new(B * N6, Cd)
    if G.use_fused_attention:
        G.attention_seg(q, k, v, B, list(ntok), Nk, Cd ** -0.5, out=att, q_prescaled=pre)      # all four scales, one launch
    else:
        for s in range(4):
            G.attention(q.rowslice(R[s], R[s + 1]), k, v, B, ntok[s], Nk, Cd ** -0.5, out=att.rowslice(R[s], R[s + 1]))
    # (tx1's gradient is covered: the four per-scale MixFFN residuals write its four row groups before LayerNorm 2's backward adds to it;
    # tx = norm2(tx1) likewise: its gradient = the four MixFFN input gradients, row group by row group)
    return _proj_ln(M, G, att, name + ".proj", X, ln, out=G.new(B * N6, Cd, covered=True), ln_out=G.new(B * N6, Cd, covered=True) if ln is not None else None)


def _bridge_layer(M, G, X: Var, li: int, B: int, sides, ntok, R, N6) -> Var:
    """BridgLayer_4, MSTr.py:2373-2409, on the stage-major token buffer [sum_s B*ntok_s, 64]."""
    name = f"bridge.bridge_layer{li}"
    n = _ln(M, G, X, name + ".norm1")
    if M.br_ch_att_list[li - 1]:
        tx1 = _channel_att(M, G, n, X, name + ".attn", B, ntok, R, N6)
        tx = _ln(M, G, tx1, name + ".norm2", out=G.new(B * N6, 64, covered=True))    # gradient = the four MixFFN input gradients, row group by row group
    else:
        tx1, tx = _self_att(M, G, n, X, name + ".attn", B, sides, ntok, R, N6, ln=(name + ".norm2", 1e-5))
    tx2 = G.new(B * N6, 64)
    geo = [(B * sides[s] * sides[s], 64 * MULT[s]) for s in range(4)]
    view = lambda v, s: v.rowslice(R[s], R[s + 1]).reshape(*geo[s])
    if FUSED_MIXFFN and MANY_MIXFFN and not G.use_streams:
        # the four per-scale MixFFNs as ONE fused site list: 3 forward + 3 backward launches for all four scales
        G.mixffn([_mixffn_site(M, G, view(tx, s), f"{name}.mixffn{s + 1}", B, sides[s], sides[s], view(tx1, s), view(tx2, s)) for s in range(4)])
        return tx2
    if MANY_MIXFFN and not G.use_streams:
        # the four per-scale MixFFNs level by level: fc1 x4 in one launch, dw x4, LN x4, fc2 x4 in one launch (and the eight
        # gradient GEMMs of each level in one launch): four independent chains of small kernels share the CUs
        nm = [f"{name}.mixffn{s + 1}" for s in range(4)]
        hs = G.linear_many([(view(tx, s), *_lin(M, G, nm[s] + ".fc1"), G.new(geo[s][0], 4 * geo[s][1]), None) for s in range(4)])
        ds = G.dwconv_multi(hs, [M._P(G, nm[s] + ".dwconv.dwconv.weight") for s in range(4)],
                            [M._P(G, nm[s] + ".dwconv.dwconv.bias") for s in range(4)], [(B, sides[s], sides[s]) for s in range(4)],
                            [3] * 4, [None] * 4, add_input=True)
        acts = [_ln(M, G, ds[s], nm[s] + ".norm1", act=ACT_GELU) for s in range(4)]
        G.linear_many([(acts[s], *_lin(M, G, nm[s] + ".fc2"), view(tx2, s), view(tx1, s)) for s in range(4)])
        return tx2
    with G.parallel(4, shared=(tx, tx1)) as par:    # the four per-scale MixFFNs are independent
        for s in range(4):
            with par.branch(s):
                _mixffn(M, G, view(tx, s), f"{name}.mixffn{s + 1}", B, sides[s], sides[s], residual=view(tx1, s), out=view(tx2, s))
    return tx2


def _patch_expand(M, G, t: Var, name: str, B: int, side: int, p: int) -> Var:
    """PatchExpand / FinalPatchExpand_X4, MSTr.py:184-201, 213-227."""
    if t.rows != B * side * side:
        raise AssertionError("input feature has wrong size")
    y = G.linear(t, *_lin(M, G, name + ".expand", bias=False))
    if SHUFFLE_IN_LN:                            # the rearrange is an address computation inside the LayerNorm kernels
        return G.layernorm_shuffled(y, M._P(G, name + ".norm.weight"), M._P(G, name + ".norm.bias"), B, side, side, p)
    return _ln(M, G, G.pixel_shuffle(y, B, side, side, p), name + ".norm")


def _decoder(M, G, x1: Var, skip: Var, name: str, B: int, side: int, last: bool) -> Var:
    """MyDecoderLayer, MSTr.py:271-290; cat([x1, skip]) @ W^T is evaluated as two accumulating GEMMs."""
    W, b = _lin(M, G, name + ".concat_linear")
    c1 = x1.cols
    t = G.linear(x1, W, b, wcols=(0, c1))
    G.linear(skip, W, None, out=t, wcols=(c1, c1 + skip.cols), accumulate=True)
    t = _eff_block(M, G, t, name + ".layer_former_1", B, side, side)
    t = _eff_block(M, G, t, name + ".layer_former_2", B, side, side)
    if not last:
        return _patch_expand(M, G, t, name + ".layer_up", B, side, 2)
    tok = bool(getattr(M, "_tok_logits", False))
    up = name + ".layer_up"
    if t.rows != B * side * side:
        raise AssertionError("input feature has wrong size")
    Wc, bc = _lin(M, G, name + ".last_layer")
    gn, bn = M._P(G, up + ".norm.weight"), M._P(G, up + ".norm.bias")
    ye = G.linear(t, *_lin(M, G, up + ".expand", bias=False))       # FinalPatchExpand_X4.expand, MSTr.py:219-221
    if SHUFFLE_IN_LN and G.ln_cls_supported(ye, 4, gn, bn, Wc, bc, B, side, side):
        # rearrange + norm of FinalPatchExpand_X4 and last_layer in one launch each way (csrc/lncls.hip): the normalised 224^2 x 64 map is never stored
        lg = G.ln_cls(ye, gn, bn, Wc, bc, B, side, side, 4, pad_rows=tok)
    else:
        y = G.layernorm_shuffled(ye, gn, bn, B, side, side, 4) if SHUFFLE_IN_LN else _ln(M, G, G.pixel_shuffle(ye, B, side, side, 4), up + ".norm")
        lg = G.linear(y, Wc, bc)                                # [B*16*side^2, classes]
    if tok:
        return lg                                               # token-major for the captured step's loss kernels
    return G.transpose(lg, B)                                   # NCHW logits [B*classes, H*W]


def _forward(M: MSTransception, G: Graph, x: torch.Tensor, B: int, in_ch: int, S: int) -> Var:
    """MSTransception.forward, MSTr.py:2826-2852 / MSViT.forward :1709-1744 / BridgeBlock_4 :2422-2442."""
    sides = [S // 4, S // 8, S // 16, S // 32]
    ntok = [sides[i] * sides[i] * MULT[i] for i in range(4)]      # 64-wide tokens per image and stage
    N6 = sum(ntok)
    R = [0]
    for nt in ntok:
        R.append(R[-1] + B * nt)
    Xb = G.new(B * N6, 64)                                        # stage-major bridge buffer = the encoder outputs

    def stage_map(buf: Var, s: int) -> Var:
        return buf.rowslice(R[s], R[s + 1]).reshape(B * sides[s] * sides[s], 64 * MULT[s])

    tap = getattr(M, "capture_taps", False)
    if M.Stage_3or4 == 4:
        # MSViT_4Stages.forward, MSTr.py:1956-1988: Conv2d_BN stem (3x3 stride 2 + BatchNorm + Hardswish, twice: im2col + GEMM), then FOUR
        # RIPM + MHCA stages -- the first with two paths, one layer and no pooling
        G.segment("stage1")
        m = x
        for i, (cin, hw) in enumerate(((3, S), (DIMS[0] // 2, S // 2))):
            cols = G.im2col3s2(m, B, cin, hw, hw, src_ch=in_ch if i == 0 else 0)
            z = G.linear(cols, *_lin(M, G, f"backbone.stem.{i}.conv", bias=False), bn_shift=_bn_shift(M, f"backbone.stem.{i}.bn"))
            m = _bn(M, G, z, f"backbone.stem.{i}.bn", ACT_HSWISH)
        if tap:
            M.taps = {}
        side = sides[0]
        for s4, (layers, npath) in enumerate(zip((1,) + tuple(LAYERS), (2, 3, 3, 3))):
            if s4 >= 2:
                G.mark(f"stage{s4 + 1}_done")
            G.segment(f"ripm{s4 + 1}")
            stack, side = _ripm(M, G, m, f"backbone.patch_embed_stage{s4 + 1}", B, side, npath, pool=s4 > 0)
            m = _mhca_stage(M, G, stack, f"backbone.mhca_stage{s4 + 1}", layers, B, side, stage_map(Xb, s4), npath)
        return _bridge_and_decoder(M, G, Xb, B, S, sides, ntok, R, N6, tap)
    # stage 1 -- OverlapPatchEmbeddings + 2 EfficientTransformerBlocks (MSTr.py:1714-1721)
    G.segment("stage1")
    cols = G.stem_im2col(x, B, in_ch, S, S)
    W, b = _lin(M, G, "backbone.patch_embed1.proj")
    t = G.linear(cols.colslice(0, 147), W, b)
    t = _ln(M, G, t, "backbone.patch_embed1.norm")
    if tap:
        M.taps = {"patch_embed1": t.data.float().view(B, sides[0] * sides[0], 64).clone()}
    for i in range(2):
        t = _eff_block(M, G, t, f"backbone.block1.{i}", B, sides[0], sides[0])
    m = _ln(M, G, t, "backbone.norm1", out=stage_map(Xb, 0))
    # stages 2-4 -- RIPM + MB transformer + IFF (MSTr.py:1728-1742)
    for s in (1, 2, 3):
        if s >= 2:
            G.mark(f"stage{s + 1}_done")                          # a backward sweep that stops here has finished stage s + 1 (its RIPM included)
        G.segment(f"ripm{s + 1}")
        stack, side = _ripm(M, G, m, f"backbone.patch_embed_stage{s + 1}", B, sides[s - 1])
        m = _mhca_stage(M, G, stack, f"backbone.mhca_stage{s + 1}", LAYERS[s - 1], B, side, stage_map(Xb, s))
    return _bridge_and_decoder(M, G, Xb, B, S, sides, ntok, R, N6, tap)


def _spatial_aware_trans(M, G, Xb: Var, name: str, B: int, sides, R, stage_map) -> Var:
    """SpatialAwareTrans, MSTr.py:2617-2664 (an ablation variant, run op by op): per scale a Linear to 64 channels, the window partition (windows of
    8 / 4 / 2 / 1 pixels: 64 + 16 + 4 + 1 = 85 tokens per window of the common H/8 x W/8 grid), num_sp InterTransBlocks (:2562-2583) -- LayerNorm,
    8-head attention over a window's tokens with UNSCALED scores (:2555), proj, skip; LayerNorm, MLP_FFN with Dropout(0.1) after GELU and after
    fc2 (:63-77), skip -- the windows put back and a Linear to the scale's width (fc_back), written as the bridge's stage-major buffer."""
    wins, ntw, nb = (8, 4, 2, 1), 85, sides[3]
    assert all(sides[j] == nb * wins[j] for j in range(4)), "SpatialAwareTrans needs the four scales on a common window grid"
    nw, d, h = B * nb * nb, 64, 8
    win = G.new(nw * ntw, d)
    off = 0
    for j in range(4):
        t = G.linear(stage_map(Xb, j), *_lin(M, G, f"{name}.fc{j + 1}"))
        G.window_rows(t, win, B, sides[j], sides[j], wins[j], ntw, off, to_map=False)
        off += wins[j] * wins[j]
    x = win
    seed = M._dropout_seed(G)
    for i in range(M.num_sp):
        blk = f"{name}.group_attention.{i}"
        qkv = G.linear(_ln(M, G, x, blk + ".SlayerNorm_1", 1e-6), *_lin(M, G, blk + ".Attention.qkv_linear"))
        q, k, v = qkv.colslice(0, d), qkv.colslice(d, 2 * d), qkv.colslice(2 * d, 3 * d)
        sc = G.new(nw * h * ntw, ntw)                             # scores per (window, head): [85, 85]
        G.bmm(q, k, sc, ntw, ntw, d // h, 0, 1, nb1=nw, nb2=h, sA=(ntw * qkv.ld, d // h), sB=(ntw * qkv.ld, d // h), sC=(h * ntw * ntw, ntw * ntw))
        p = G.softmax(sc, 1, 1)
        o = G.new(nw * ntw, d)
        G.bmm(p, v, o, ntw, d // h, ntw, 0, 0, nb1=nw, nb2=h, sA=(h * ntw * ntw, ntw * ntw), sB=(ntw * qkv.ld, d // h), sC=(ntw * d, d // h))
        x = G.linear(o, *_lin(M, G, blk + ".Attention.proj"), residual=x)
        y = G.gelu(G.linear(_ln(M, G, x, blk + ".SlayerNorm_2", 1e-6), *_lin(M, G, blk + ".mlp.fc1")))
        y = G.dropout(y, M.sp_dropout, seed, 2 * i)
        y = G.dropout(G.linear(y, *_lin(M, G, blk + ".mlp.fc2")), M.sp_dropout, seed, 2 * i + 1)
        x = G.add(x, y)
    X = G.new(Xb.rows, Xb.cols)
    off = 0
    for j in range(4):
        t = G.new(B * sides[j] * sides[j], d)
        G.window_rows(x, t, B, sides[j], sides[j], wins[j], ntw, off, to_map=True)
        off += wins[j] * wins[j]
        G.linear(t, *_lin(M, G, f"{name}.fc_back.{j}"), out=stage_map(X, j))
    return X


def _bridge_and_decoder(M: MSTransception, G: Graph, Xb: Var, B: int, S: int, sides, ntok, R, N6: int, tap: bool) -> Var:
    """BridgeBlock_4 (MSTr.py:2422-2442) and the four decoder layers (:2843-2850) over the stage-major encoder buffer."""
    def stage_map(buf: Var, s: int) -> Var:
        return buf.rowslice(R[s], R[s + 1]).reshape(B * sides[s] * sides[s], 64 * MULT[s])

    # Dual Transformer Bridge.  The mark lets a multi-GPU step stop its backward sweep here -- bridge and decoder gradients (72 % of
    # the live gradient bytes) are complete and can travel while the encoder's backward runs.
    G.mark("encoder_done")
    X = Xb

    def image_major(buf: Var) -> torch.Tensor:                     # stage-major [stage][B][tokens][64] -> the reference's [B, 6076, 64]
        return torch.cat([buf.data[R[s]:R[s + 1]].float().view(B, ntok[s], 64) for s in range(4)], dim=1)
    if tap:
        for s in range(4):
            M.taps[f"enc{s}"] = stage_map(Xb, s).data.float().view(B, sides[s], sides[s], 64 * MULT[s]).clone()
    if M.have_bridge == "para":                                   # BridgeBlock_para.forward, MSTr.py:2514-2524
        b1 = _bridge_layer(M, G, X, 1, B, sides, ntok, R, N6)      # channel attention and ...
        b2 = _bridge_layer(M, G, X, 2, B, sides, ntok, R, N6)      # ... spatial attention on the same input
        Wp, bp = _lin(M, G, "bridge.proj_act.0")                   # Linear over cat([b1, b2], channels) = two accumulating products
        y = G.linear(b1, Wp, bp, wcols=(0, 64))
        G.linear(b2, Wp, None, wcols=(64, 128), out=y, accumulate=True)
        X = _ln(M, G, y, "bridge.proj_act.1", act=ACT_GELU)
        for li in (3, 4):
            X = _bridge_layer(M, G, X, li, B, sides, ntok, R, N6)
        if tap:
            M.taps["bridge4"] = image_major(X)
    elif M.have_bridge != "None":                                 # MSTr.py:2840
        if M.have_bridge == "sp" and M.num_sp > 0:                # BridgeLayer_new, MSTr.py:2686-2704: only the first layer is handed the maps
            X = _spatial_aware_trans(M, G, Xb, "bridge.bridge_layer1.scale_fuse_att", B, sides, R, stage_map)
        for li in range(1, 5):
            G.segment(f"bridge{li}")
            X = _bridge_layer(M, G, X, li, B, sides, ntok, R, N6)
            if tap:
                M.taps[f"bridge{li}"] = image_major(X)
    # decoder
    G.segment("dec3")
    d3 = _patch_expand(M, G, stage_map(X, 3), "decoder_3.layer_up", B, sides[3], 2)
    G.segment("dec2")
    d2 = _decoder(M, G, d3, stage_map(X, 2), "decoder_2", B, sides[2], False)
    G.segment("dec1")
    d1 = _decoder(M, G, d2, stage_map(X, 1), "decoder_1", B, sides[1], False)
    if tap:
        M.taps["dec1"] = d1.data.float().view(B, sides[0] * sides[0], d1.cols).clone()
    G.segment("dec0")
    out = _decoder(M, G, d1, stage_map(X, 0), "decoder_0", B, sides[0], True)
    G.segment("loss")
    return out
