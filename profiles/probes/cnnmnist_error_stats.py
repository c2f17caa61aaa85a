import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
from bitnetmcu_amd import qat
GM = np.load(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests/golden/qat_fc_model.npz"))
m = qat.CNNMNIST(96, 64, 0, cnn_width=64, QuantType="4bitsym").cuda()
layers = [x for x in list(m.model) + [m.classifier] if hasattr(x, "weight_quant")]
with torch.no_grad():
    for l, layer in enumerate(layers):
        layer.weight.copy_(torch.from_numpy(GM[f"cnn/w{l}"])); layer.s = torch.nn.Parameter(torch.from_numpy(GM[f"cnn/s{l}"]).reshape(()).cuda(), requires_grad=False)
x = torch.from_numpy(GM["cnn/x"]).cuda().reshape(-1, 1, 16, 16).requires_grad_(True)
with torch.no_grad(): feats = m.front(x.detach())
ef = np.abs(feats.cpu().numpy() - GM["cnn/features"]).max(axis=1) / np.abs(GM["cnn/features"]).max(axis=1)
y = m(x); ref = GM["cnn/logits"]
err = np.abs(y.detach().cpu().numpy() - ref).max(axis=1) / np.abs(ref).max(axis=1)
(y * torch.from_numpy(GM["cnn/gy"]).cuda()).sum().backward()
print("features: frac<=1e-3", (ef <= 1e-3).mean(), "frac<=1e-5", (ef<=1e-5).mean(), "max", ef.max())
print("logits: frac<=2e-3", (err <= 2e-3).mean(), "frac<=5e-4", (err<=5e-4).mean(), "max", err.max())
def g(got, want, what):
    e = np.abs(got - want) / np.abs(want).max(); print(what, "median", np.median(e), "max", e.max())
g(x.grad.reshape(-1, 256).cpu().numpy(), GM["cnn/gx"].reshape(-1, 256), "gx")
for l, layer in enumerate(layers): g(layer.weight.grad.cpu().numpy(), GM[f"cnn/gw{l}"], f"gw{l}")
