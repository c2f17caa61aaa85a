import sys, os, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from bitnetmcu_amd import qat
import util
GM = util.qat_model_golden()
m = qat.CNNMNIST(96, 64, 0, cnn_width=64, QuantType="4bitsym", WScale="PerTensor", NormType="RMS", num_classes=10).cuda()
layers = [x for x in list(m.model) + [m.classifier] if hasattr(x, "weight_quant")]
with torch.no_grad():
    for l, layer in enumerate(layers):
        layer.weight.copy_(torch.from_numpy(GM[f"cnn/w{l}"]))
        layer.s = torch.nn.Parameter(torch.from_numpy(GM[f"cnn/s{l}"]).reshape(()).cuda(), requires_grad=False)
x = torch.from_numpy(GM["cnn/x"]).cuda().reshape(-1, 1, 16, 16)
ref_f = GM["cnn/features"]
with torch.no_grad():
    assert m.front_fused(x)
    f = m.front(x)
    y = m(x)
def layerwise(mod, x):
    with torch.no_grad():
        for k in list(mod.model)[:9]:
            x = k(x)
    return x
fl = layerwise(m, x)
for name, got in (("fused", f), ("layerwise", fl)):
    ef = np.abs(got.cpu().numpy() - ref_f).max(axis=1) / np.abs(ref_f).max(axis=1)
    print(name, "features vs ref: frac<=1e-5", (ef <= 1e-5).mean(), "frac<=1e-4", (ef <= 1e-4).mean(), "max", ef.max())
ef = (f - fl).abs().max(dim=1).values / fl.abs().max(dim=1).values
print("fused vs layerwise: ", float((ef <= 1e-5).float().mean()), float(ef.max()))
ref = GM["cnn/logits"]
err = np.abs(y.cpu().numpy() - ref).max(axis=1) / np.abs(ref).max(axis=1)
print("logits: frac<=5e-4", (err <= 5e-4).mean(), "max", err.max())
# other channel counts, random weights
for C in (16, 32, 48, 64, 100, 128):
    torch.manual_seed(C)
    mm = qat.CNNMNIST(64, 64, 0, cnn_width=C, QuantType="4bitsym").cuda()
    for n in (1, 3, 64, 1001):
        xx = torch.randn(n, 1, 16, 16, device="cuda") * (torch.rand(n, 1, 1, 1, device="cuda") * 2 + 0.05)
        with torch.no_grad():
            assert mm.front_fused(xx), C
            a = mm.front(xx)
        b_ = layerwise(mm, xx)
        e = (a - b_).abs().max(dim=1).values / b_.abs().max(dim=1).values.clamp(min=1e-30)
        print(C, n, "frac<=1e-5", float((e <= 1e-5).float().mean()), "frac<=1e-4", float((e <= 1e-4).float().mean()), "max", float(e.max()))
