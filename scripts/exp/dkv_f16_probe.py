"""Which of the two dK/dV implementations agrees with a straight fp32 emulation on the one fp16 case where they differ in the last place
(B=2, nq=[100,37], Nk=80: query chunk 1 of image 1, key 29)?"""
import os, sys, ctypes as C, torch
sys.path.insert(0, "/root/repo")
from transception_amd._lib import TC_F16, lib
from transception_amd.engine import ATTN_DKV_SPLITS
from transception_amd.seeded_init import seeded_tensor
DEV, L = "cuda:0", lib()
d, scale, l2e = 64, 0.125, 1.4426950408889634
dtype, dt = torch.float16, TC_F16
st = torch.cuda.current_stream().cuda_stream
T = lambda tag, shape: torch.from_numpy(seeded_tensor("ops/" + tag, shape, 1.0))
ci, B, nq, Nk = 2, 2, [100, 37], 80
rows = B * sum(nq)
q = (T(f"dv.q{ci}", (rows, d)).to(DEV) * (scale * l2e)).to(dtype)
kv = T(f"dv.kv{ci}", (B * Nk, 2 * d)).to(DEV).to(dtype)
k, v = kv[:, :d], kv[:, d:]
do = T(f"dv.g{ci}", (rows, d)).to(DEV).to(dtype)
nqc = (C.c_int * 4)(*(nq + [0, 0]))
o = torch.empty((rows, d), device=DEV, dtype=dtype); lse = torch.empty((rows,), device=DEV)
L.tc_attn_fwd_seg(q.data_ptr(), d, k.data_ptr(), 2 * d, v.data_ptr(), 2 * d, Nk * 2 * d, o.data_ptr(), d, lse.data_ptr(), B, 2, nqc, Nk, scale, 1, dt, st)
part = {}
for impl in ("1", "0"):
    os.environ["TC_ATTN_DKV_ASM"] = impl; os.environ["TC_ATTN_DQ_ASM"] = "0"
    dq = torch.zeros((rows, d), device=DEV).to(dtype); dkv = torch.zeros((B * Nk, 2 * d), device=DEV).to(dtype)
    delta = torch.empty((rows,), device=DEV); dkv32 = torch.zeros((ATTN_DKV_SPLITS * B * Nk * 128,), device=DEV)
    L.tc_attn_bwd_seg(q.data_ptr(), d, k.data_ptr(), 2 * d, v.data_ptr(), 2 * d, Nk * 2 * d, o.data_ptr(), d, do.data_ptr(), d, lse.data_ptr(), delta.data_ptr(),
                      dkv32.data_ptr(), dq.data_ptr(), d, dkv.data_ptr(), 2 * d, dkv.data_ptr() + 2 * d, 2 * d, Nk * 2 * d, B, 2, nqc, Nk, scale, 1, dt, st)
    torch.cuda.synchronize()
    part[impl] = dkv32[:4 * B * Nk * 128].view(4, B, Nk, 128)[1, 1, 29, :64].clone()      # dK partial of chunk 1, image 1, key 29
    dl = delta.clone()
# emulation: image 1, segment 0 rows 64..99 (tiles 2, 3 of segment 0: rows 100 + 64 .. 100 + 99 of the stage-major buffer)
r0 = 1 * 100 + 64
qs = q[r0:r0 + 36].float(); dos = do[r0:r0 + 36].float()
K29 = k[1 * Nk + 29].float(); V29 = v[1 * Nk + 29].float()
S = qs @ K29                                       # fp32 (the MFMA accumulates in fp32 in another order)
P = torch.exp2(S - lse[r0:r0 + 36] * l2e)
dP = dos @ V29 - dl[r0:r0 + 36]
dS = (P * dP)
dS16 = dS.to(dtype).float()
ref = (dS16[:, None] * qs).sum(0) * 0.6931471805599453
print("asm  - emulation: max", (part["1"].cpu() - ref.cpu()).abs().max().item())
print("hip  - emulation: max", (part["0"].cpu() - ref.cpu()).abs().max().item())
print("asm  - hip      : max", (part["1"] - part["0"]).abs().max().item())
# how close are the dS values to an fp16 rounding tie?
ulp = torch.where(dS.abs() > 0, 2.0 ** (torch.floor(torch.log2(dS.abs())) - 10), torch.zeros_like(dS))
frac = ((dS / ulp) % 1.0 - 0.5).abs()
i = int(frac.argmin())
print("closest-to-tie dS element: row", i, "dS", dS[i].item(), "distance to tie (ulps)", frac[i].item(), " its Q row max", qs[i].abs().max().item())
alt = dS16.clone(); alt[i] = alt[i] + (ulp[i] if dS[i] > dS16[i] else -ulp[i]) * 1.0
ref2 = (alt[:, None] * qs).sum(0) * 0.6931471805599453
print("with that element rounded the other way: asm - emu'", (part["1"].cpu() - ref2.cpu()).abs().max().item(), " hip - emu'", (part["0"].cpu() - ref2.cpu()).abs().max().item())
