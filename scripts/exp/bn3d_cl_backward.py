"""torch 2.10 (CPU): batch-norm backward of a channels-last-3d input with a dense incoming gradient returns a wrong input gradient.

Why this matters here: the reference's CAM_Factorized_Module hands BatchNorm3d exactly that layout, so gradients of concat="cam_fact" taken from
the reference on this container's CPU are off by more than their norm.  tests/golden/make_golden.py gives BatchNorm3d a dense copy of its input
while it makes that variant's vectors (dense_batchnorm3d_input)."""
import torch

torch.manual_seed(0)
B, C, D, H, W = 1, 320, 4, 7, 7
x = torch.randn(B, D * H * W, C).permute(0, 2, 1).reshape(B, C, D, H, W)           # a view with channels-last-3d strides
w, b, g = torch.rand(C) + 0.5, torch.rand(C), torch.randn(B, C, D, H, W)


def grad_in(xx, gg):
    xi = xx.clone().requires_grad_(True)
    torch.nn.functional.batch_norm(xi, None, None, w, b, True, 0.1, 1e-5).backward(gg)
    return xi.grad


dense = grad_in(x.contiguous(), g)
print("torch", torch.__version__, "| strides", x.stride())
print("|dense - channels_last_3d| with a dense gradient       ", (dense - grad_in(x, g)).norm().item(), "of", dense.norm().item())
print("|dense - channels_last_3d| with a channels-last gradient", (dense - grad_in(x, g.contiguous(memory_format=torch.channels_last_3d))).norm().item())
