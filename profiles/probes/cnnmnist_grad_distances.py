import sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from bitnetmcu_amd import qat
import util
GM = util.qat_model_golden()
for tag, nt in (("cnn32", "Lin"), ("cnn48", "RMS"), ("cnn", "RMS")):
    if tag == "cnn":
        cw, w1, w2, ncls = 64, 96, 64, 10
    else:
        cw, w1, w2, ncls = (int(v) for v in GM[f"{tag}/cfg"])
    for fused in (True, False):
        m = qat.CNNMNIST(w1, w2, 0, cnn_width=cw, QuantType="4bitsym", WScale="PerTensor", NormType=nt, num_classes=ncls).cuda()
        layers = [x for x in list(m.model) + [m.classifier] if hasattr(x, "weight_quant")]
        with torch.no_grad():
            for l, layer in enumerate(layers):
                layer.weight.copy_(torch.from_numpy(GM[f"{tag}/w{l}"]))
                layer.s = torch.nn.Parameter(torch.from_numpy(GM[f"{tag}/s{l}"]).reshape(()).cuda(), requires_grad=False)
        x = torch.from_numpy(GM[f"{tag}/x"]).cuda().reshape(-1, 1, 16, 16).requires_grad_(True)
        if fused:
            y = m(x)
        else:
            y = x
            for k in list(m.model):
                y = k(y)
            y = m.classifier(y)
        ref = GM[f"{tag}/logits"]
        err = np.abs(y.detach().cpu().numpy() - ref).max(axis=1) / np.abs(ref).max(axis=1)
        (y * torch.from_numpy(GM[f"{tag}/gy"]).cuda()).sum().backward()
        out = [f"{tag} fused={fused} logits frac<=5e-4 {(err <= 5e-4).mean():.3f} max {err.max():.2e} |"]
        def gc(got, want, what):
            e = np.abs(got - want) / np.abs(want).max()
            out.append(f"{what} med {np.median(e):.1e} max {e.max():.1e}")
        gc(x.grad.reshape(-1, 256).cpu().numpy(), GM[f"{tag}/gx"].reshape(-1, 256), "gx")
        for l, layer in enumerate(layers):
            gc(layer.weight.grad.cpu().numpy(), GM[f"{tag}/gw{l}"], f"gw{l}")
        print(" ".join(out))
