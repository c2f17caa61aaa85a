"""C-exact evaluation of a quantised model on a float dataset, on the GPU.

The reference evaluates an exported model by looping `QuantizedModel.inference_quantized` over the test set in
numpy (exportquant.py:537-559, BitNetMCU.py:420-535) — a float emulation that is not bit-exact to the C engine — and
by calling the DLL once per image (test_inference.py:136-168).  This does the same job in two kernel launches: the
reference's input quantisation (test_inference.py:140-141) and the whole-model inference, both on device, with the C
engine's exact integer arithmetic."""
import numpy as np


def predict(ctx, images_float, batch=1 << 20):
    """images_float: array-like [n,256] or [n,16,16] float32 (already normalised as the reference's dataloader does).
    Returns uint32 class ids identical to calling the reference DLL image by image."""
    import torch
    x = np.ascontiguousarray(images_float, dtype=np.float32).reshape(-1, 256)
    out = np.empty(len(x), np.uint32)
    dev = torch.device("cuda", ctx._lib.bnm_ctx_device(ctx._h))
    for s in range(0, len(x), batch):
        xb = torch.from_numpy(x[s:s + batch]).to(dev)
        cls = torch.empty(len(xb), dtype=torch.int32, device=dev)
        ctx.infer_device(ctx.quantize_device(xb), cls)
        out[s:s + batch] = cls.cpu().numpy().astype(np.uint32)
    return out


def accuracy(ctx, images_float, labels, batch=1 << 20):
    """Counterpart of the 'Overall accuracy C' line of test_inference.py:171."""
    pred = predict(ctx, images_float, batch)
    labels = np.asarray(labels)
    return float((pred == labels).mean()), pred
