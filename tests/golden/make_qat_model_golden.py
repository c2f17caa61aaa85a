#!/usr/bin/env python3
"""Golden vectors for the whole-model QAT forward, generated HERE by importing the
reference's own FCMNIST (/root/reference/models.py:56-90 over BitNetMCU.py's BitLinear, CPU PyTorch) - the reference cannot
travel to the GPU box, the vectors can.

  python tests/golden/make_qat_model_golden.py      -> tests/golden/qat_fc_model.npz

Per configuration: the inputs, every layer's weights / clipping scalars / weight levels and scale, the reference's logits and
every hidden layer's output after ReLU; for configuration "a" also the gradients of all weights and of the input.
Inputs look like the reference's training images (transforms.Normalize((0.1307,), (0.3081,)) of [0, 1] pixels: background
-0.4242) plus a tiny row, a huge row and an ALL-ZERO row (the reference's Normalize divides 0 / 0: its logits are NaN).
"""
import os
import sys

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
# tag: (width1, width2, width3, QuantType, WScale, NormType, classes, rows, clipping algorithm)
CONFIGS = {
    "a": (64, 64, 64, "4bitsym", "PerTensor", "RMS", 10, 333, "octav"),       # the reference's headline FC model
    "b": (96, 64, 0, "4bitsym", "PerTensor", "RMS", 10, 70, "octav"),         # trainingparameters.yaml's widths: three layers
    "c": (96, 96, 96, "Ternary", "PerTensor", "Lin", 10, 45, "octav"),
    "d": (128, 128, 128, "8bit", "PerTensor", "RMS", 47, 97, "octav"),        # four tiles per layer, two class tiles (EMNIST balanced)
    "e": (80, 40, 72, "2bitsym", "PerOutput", "RMS", 10, 41, "prop"),         # widths off the 32-row tiles, per-output clipping scalars
    "f": (64, 64, 64, "Binary", "PerTensor", "RMS", 10, 33, "octav"),
    "g": (64, 64, 64, "4bitsym", "PerTensor", "LayerNorm", 10, 90, "octav"),  # LayerNorm: no NaN for the all-zero row (epsilon)
    "h": (80, 40, 72, "8bit", "PerTensor", "LayerNorm", 47, 50, "octav"),     # ... with widths off the tiles (padding columns masked)
}
# the six-tile class (hidden widths 129 .. 192; the reference's documented 12 KB family ends at binary 160-160-160,
# docs/documentation.md:169-183) -> tests/golden/qat_fc_model_wide.npz
CONFIGS_WIDE = {
    "i": (160, 160, 160, "Binary", "PerTensor", "RMS", 10, 40, "octav"),
    "j": (192, 144, 0, "4bitsym", "PerOutput", "Lin", 26, 37, "prop"),
    "k": (136, 192, 160, "8bit", "PerTensor", "LayerNorm", 47, 34, "octav"),
}
ZERO_ROW = 5


def images(n, gen):
    x = torch.full((n, 256), -0.4242)
    on = torch.rand(n, 256, generator=gen) < 0.25
    x[on] = (torch.rand(int(on.sum()), generator=gen) * 3.2 - 0.4)
    x[1] *= 1e-3
    x[2] *= 1e3
    x[ZERO_ROW] = 0.0
    return x


def main():
    sys.path.insert(0, REF)
    import models as ref          # the reference module, unmodified
    out_main, wide = {}, {}
    for tag, (w1, w2, w3, qt, wscale, nt, ncls, n, algo) in list(CONFIGS.items()) + list(CONFIGS_WIDE.items()):
        out = wide if tag in CONFIGS_WIDE else out_main
        torch.manual_seed(20240324 + ord(tag))
        gen = torch.Generator().manual_seed(7 + ord(tag))
        model = ref.FCMNIST(w1, w2, w3, QuantType=qt, WScale=wscale, NormType=nt, num_classes=ncls)
        layers = [m for m in model.model if hasattr(m, "weight_quant")] + [model.classifier]
        for m in layers:
            m.update_clipping_scalar(m.weight.data, algo, 0.25)          # as training.py does before the first epoch
        x = images(n, gen)
        xr = x.clone().requires_grad_(True)
        hidden = []
        hooks = [m.register_forward_hook(lambda mod, i, o: hidden.append(o.detach().numpy().copy()))
                 for m in model.model if isinstance(m, torch.nn.ReLU)]
        logits = model(xr.reshape(n, 1, 16, 16))
        [h.remove() for h in hooks]
        out[f"{tag}/x"] = x.numpy()
        out[f"{tag}/logits"] = logits.detach().numpy()
        out[f"{tag}/hidden"] = np.concatenate(hidden, axis=1)
        out[f"{tag}/cfg"] = np.array([w1, w2, w3, ncls], dtype=np.int64)
        for l, m in enumerate(layers):
            u, wsc, _ = m.weight_quant(m.weight.data)
            out[f"{tag}/w{l}"] = m.weight.detach().numpy()
            out[f"{tag}/s{l}"] = m.s.detach().numpy().reshape(-1).astype(np.float32)
            out[f"{tag}/w_int{l}"] = u.numpy()
            out[f"{tag}/w_scale{l}"] = np.asarray(torch.as_tensor(wsc).detach().numpy(), dtype=np.float32).reshape(-1)
        if tag == "a":
            # gradients over the rows without the all-zero one (its NaNs would poison every gradient): a second forward pass
            keep = torch.ones(n, dtype=torch.bool)
            keep[ZERO_ROW] = False
            gy = torch.randn(n - 1, ncls, generator=gen)
            xg = x[keep].clone().requires_grad_(True)
            loss = (model(xg.reshape(n - 1, 1, 16, 16)) * gy).sum()
            grads = torch.autograd.grad(loss, [xg] + [m.weight for m in layers])
            out[f"{tag}/gy"] = gy.numpy()
            out[f"{tag}/gx"] = grads[0].numpy()
            for l, g in enumerate(grads[1:]):
                out[f"{tag}/gw{l}"] = g.numpy()
    out = out_main
    # ---- CNNMNIST (models.py:93-139; the model trainingparameters.yaml names), its widths: 96-64-0, 64 channels, 4bitsym / RMS ----
    torch.manual_seed(20240419)
    gen = torch.Generator().manual_seed(419)
    cnn = ref.CNNMNIST(96, 64, 0, cnn_width=64, QuantType="4bitsym", WScale="PerTensor", NormType="RMS", num_classes=10)
    qlayers = [m for m in list(cnn.model) + [cnn.classifier] if hasattr(m, "weight_quant")]
    for m in qlayers:
        m.update_clipping_scalar(m.weight.data, "octav", 0.25)
    n = 45
    x = images(n, gen)
    x[ZERO_ROW] = x[ZERO_ROW + 1]                      # (no all-zero image here: its NaNs would poison the gradients)
    xr = x.reshape(n, 1, 16, 16).clone().requires_grad_(True)
    feats = []
    hook = cnn.model[8].register_forward_hook(lambda mod, i, o: feats.append(o.detach().numpy().copy()))
    logits = cnn(xr)
    hook.remove()
    gy = torch.randn(n, 10, generator=gen)
    grads = torch.autograd.grad((logits * gy).sum(), [xr] + [m.weight for m in qlayers])
    out["cnn/x"], out["cnn/logits"], out["cnn/features"], out["cnn/gy"], out["cnn/gx"] = x.numpy(), logits.detach().numpy(), feats[0], gy.numpy(), grads[0].numpy()
    for l, (m, g) in enumerate(zip(qlayers, grads[1:])):
        out[f"cnn/w{l}"], out[f"cnn/s{l}"], out[f"cnn/gw{l}"] = m.weight.detach().numpy(), m.s.detach().numpy().reshape(-1).astype(np.float32), g.numpy()
    out["cnn/state_keys"] = np.array(sorted(cnn.state_dict().keys()))
    # ---- a CNNMNIST whose FC stack has fewer than 256 inputs: 32 channels (128 features), 'Lin' norm; and 48 channels (192), RMS ----
    for tag, (cw, w1, w2, nt, seed) in {"cnn32": (32, 64, 48, "Lin", 3201), "cnn48": (48, 80, 64, "RMS", 4801)}.items():
        torch.manual_seed(seed)
        gen = torch.Generator().manual_seed(seed)
        cnn = ref.CNNMNIST(w1, w2, 0, cnn_width=cw, QuantType="4bitsym", WScale="PerTensor", NormType=nt, num_classes=10)
        qlayers = [m for m in list(cnn.model) + [cnn.classifier] if hasattr(m, "weight_quant")]
        for m in qlayers:
            m.update_clipping_scalar(m.weight.data, "octav", 0.25)
        n = 40
        x = images(n, gen)
        x[ZERO_ROW] = x[ZERO_ROW + 1]
        xr = x.reshape(n, 1, 16, 16).clone().requires_grad_(True)
        feats = []
        hook = cnn.model[8].register_forward_hook(lambda mod, i, o: feats.append(o.detach().numpy().copy()))
        logits = cnn(xr)
        hook.remove()
        gy = torch.randn(n, 10, generator=gen)
        grads = torch.autograd.grad((logits * gy).sum(), [xr] + [m.weight for m in qlayers])
        wide[f"{tag}/cfg"] = np.array([cw, w1, w2, 10], dtype=np.int64)
        wide[f"{tag}/x"], wide[f"{tag}/logits"], wide[f"{tag}/features"], wide[f"{tag}/gy"], wide[f"{tag}/gx"] = x.numpy(), logits.detach().numpy(), feats[0], gy.numpy(), grads[0].numpy()
        for l, (m, g) in enumerate(zip(qlayers, grads[1:])):
            wide[f"{tag}/w{l}"], wide[f"{tag}/s{l}"], wide[f"{tag}/gw{l}"] = m.weight.detach().numpy(), m.s.detach().numpy().reshape(-1).astype(np.float32), g.numpy()
    path = os.path.join(HERE, "qat_fc_model.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes;", {t: (out[f"{t}/logits"].shape, bool(np.isnan(out[f"{t}/logits"][ZERO_ROW]).all()), bool(np.isnan(out[f"{t}/logits"]).any())) for t in CONFIGS})
    path = os.path.join(HERE, "qat_fc_model_wide.npz")
    np.savez_compressed(path, **wide)
    print(path, os.path.getsize(path), "bytes;", {t: (wide[f"{t}/logits"].shape, bool(np.isnan(wide[f"{t}/logits"][ZERO_ROW]).all())) for t in CONFIGS_WIDE})


if __name__ == "__main__":
    main()
