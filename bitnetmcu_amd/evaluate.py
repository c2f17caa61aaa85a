"""C-exact, GPU-backed evaluation for the reference's own tools.

The reference evaluates a quantised model by looping `QuantizedModel.inference_quantized` over the test set in numpy
(exportquant.py:537-559, BitNetMCU.py:420-535; the conv emulation is a Python triple loop) — a float emulation that is not
bit-exact to the C engine — and by calling the DLL once per image (test_inference.py:136-168).  exportquant.py calls
`inference_quantized` BEFORE any header exists, so this module builds the model in memory from the reference's intermediate
representation, `QuantizedModel.quantized_model` (the list of dicts that `export_to_hfile` consumes, exportquant.py:49-263):

    ev = bitnetmcu_amd.evaluate.from_quantized_model(qm)         # qm: a reference QuantizedModel (or its list of dicts)
    logits = ev.inference_quantized(x)                           # int32 [n, classes] — the C engine's layer_out, exactly
    bitnetmcu_amd.evaluate.attach(qm)                            # or: patch qm.inference_quantized in place

The packing rules are restated from exportquant.py:104-187 (vectorised), emitted as exporter-dialect header text and parsed by
the library's run-time loader like any BitNetMCU_model.h — so what is evaluated is what `export_to_hfile` would write.
Inputs are quantised per image exactly as both reference call sites do (test_inference.py:140-141, BitNetMCU.py:435-436 on
[n, 256] inputs): scale = 127 / max(max|x|, 1e-5), round half to even, clip — on the GPU (bnm_quantize_input_device).

Note the reference's own numpy logits are NOT what this returns where the two engines differ by construction (SURVEY.md §4:
the symmetric codecs' C weights are 2x the Python half-integers, so C logits = 2x Python logits; rounding and shift rules
differ in rare cases): this evaluator is the C engine.  argmax agrees wherever the engines agree.
"""
import numpy as np

from . import _lib as L
from .model import Model, Context

# Lk_bitperweight ids written by the exporter (exportquant.py:105-177)
_QUANT_ID = {"Binary": 1, "2bitsym": 2, "4bitsym": 4, "4bit": 12, "8bit": 16, "FP130": 20, "NF4": 36, "Ternary": 64}


def _encode_fields(qtype, w):
    """weights (float array [n_out, n_in] as the reference's weight_quant leaves them) -> per-weight field codes
    (exportquant.py:105-126)."""
    if qtype == "Binary":
        return np.where(w == -1, 0, 1).astype(np.uint32), 1
    if qtype == "2bitsym":
        return ((w < 0).astype(np.uint32) << 1) | np.floor(np.abs(w)).astype(np.uint32), 2
    if qtype == "4bitsym":
        return ((w < 0).astype(np.uint32) << 3) | np.floor(np.abs(w)).astype(np.uint32), 4
    if qtype == "4bit":
        return np.floor(w).astype(np.int64).astype(np.uint32) & 15, 4
    if qtype == "8bit":
        return np.floor(w).astype(np.int64).astype(np.uint32) & 255, 8
    if qtype == "FP130":
        return ((w < 0).astype(np.uint32) << 3) | np.floor(np.log2(np.abs(w))).astype(np.uint32), 4
    if qtype == "NF4":
        levels = np.array([-1.0, -0.6962, -0.5251, -0.3949, -0.2844, -0.1848, -0.0911, 0.0,
                           0.0796, 0.1609, 0.2461, 0.3379, 0.4407, 0.5626, 0.723, 1.0])
        return np.argmin(np.abs(w[:, :, None] - levels), axis=2).astype(np.uint32), 4
    raise ValueError(f"quantization type {qtype!r} cannot be exported (exportquant.py:178-180 skips it)")


def pack_bitlinear(layer):
    """One BitLinear dict -> (bitperweight id, declared incoming_weights, C type, packed words) as export_to_hfile writes them."""
    w = np.asarray(layer["quantized_weights"], dtype=np.float64)
    qtype = layer["quantization_type"]
    n_out, n_in = w.shape
    if qtype == "Ternary":
        # 10 trits per uint16, base 3, most significant first, +1 -> 0, -1 -> 1, 0 -> 2; rows padded with zero weights to a
        # multiple of 10; stored as ceil(value * 65536 / 59049) so that the C engine's multiply-by-3 pops the digits
        # (exportquant.py:127-157, BitNetMCU_inference.c:116-136)
        pad = (-n_in) % 10
        if pad:
            w = np.pad(w, ((0, 0), (0, pad)), constant_values=0)
        trits = np.where(w == 1, 0, np.where(w == -1, 1, 2)).astype(np.int64).reshape(n_out, -1, 10)
        value = np.zeros(trits.shape[:2], np.int64)
        for t in range(10):
            value = value * 3 + trits[:, :, t]
        return 64, w.shape[1], "uint16_t", ((value * 65536 + 59048) // 59049).astype(np.uint16).ravel()
    fields, fb = _encode_fields(qtype, w)
    if (fb * n_in) % 32:
        raise ValueError(f"L{layer['layer_order']}: incoming weights x bits = {fb * n_in} is not a multiple of 32 (exportquant.py:97-98)")
    per = 32 // fb
    shifts = (32 - fb - np.arange(per, dtype=np.uint32) * fb).astype(np.uint32)          # first weight in the topmost bits
    words = np.bitwise_or.reduce(fields.reshape(-1, per) << shifts, axis=1).astype(np.uint32)
    return _QUANT_ID[qtype], n_in, "uint32_t", words


def header_text(quantized_model, modelname=None):
    """The list of dicts -> BitNetMCU_model.h text in the exporter's layout (exportquant.py:67-259).  Conv / pool geometry
    (`incoming_x`, which the reference only fills in while running inference_quantized, BitNetMCU.py:479-483) is derived from
    the 16x16 input when it is still 0."""
    layers = quantized_model
    if modelname is None:
        modelname = "CNNMNIST" if any(l["layer_type"] == "BitConv2d" for l in layers) else "FCMNIST"
    max_act = max(l["incoming_weights"] for l in layers if "incoming_weights" in l)
    o = ["// Automatically generated header file", "// Generated in memory by bitnetmcu_amd.evaluate (layout of exportquant.py)", "",
         "#include <stdint.h>", "", "#ifndef BITNETMCU_MODEL_H", "#define BITNETMCU_MODEL_H", "", f"#define MODEL_{modelname}", "",
         f"#define NUM_LAYERS {len(layers)}", "", f"#define MAX_N_ACTIVATIONS {max_act}", ""]
    xy = 16
    for l in layers:
        p = f"L{l['layer_order']}"
        if l["layer_type"] == "BitLinear":
            qid, n_decl, ctype, words = pack_bitlinear(l)
            o += [f"// Layer: {p}", f"// QuantType: {l['quantization_type']}", f"#define {p}_active", f"#define {p}_bitperweight {qid}",
                  f"#define {p}_incoming_weights {n_decl}", f"#define {p}_outgoing_weights {l['outgoing_weights']}",
                  f"const {ctype} {p}_weights[] = {{" + ",".join(hex(int(v)) for v in words) + "};", ""]
        elif l["layer_type"] == "BitConv2d":
            k = l["kernel_size"][0] if isinstance(l["kernel_size"], (tuple, list)) else int(l["kernel_size"])
            inx = int(l.get("incoming_x") or xy)
            outx = inx - k + 1
            w = np.asarray(l["quantized_weights"]).ravel()
            o += [f"// Layer: {p} (Convolutional)", f"#define {p}_active", f"#define {p}_type BitConv2d",
                  f"#define {p}_in_channels {l['in_channels']}", f"#define {p}_out_channels {l['out_channels']}",
                  f"#define {p}_incoming_x {inx}", f"#define {p}_incoming_y {inx}", f"#define {p}_outgoing_x {outx}",
                  f"#define {p}_outgoing_y {outx}", f"#define {p}_kernel_size {k}", f"#define {p}_stride 1", f"#define {p}_padding 0",
                  f"#define {p}_groups {l['groups']}", f"#define {p}_bitperweight {l['bpw']}",
                  f"const int8_t {p}_weights[] = {{" + ",".join(str(int(v)) for v in w) + "};", ""]
            xy = outx
        elif l["layer_type"] == "MaxPool2d":
            inx = int(l.get("incoming_x") or xy)
            o += [f"#define {p}_active", f"#define {p}_type MaxPool2d", f"#define {p}_pool_size {l['kernel_size']}",
                  f"#define {p}_incoming_x {inx}", f"#define {p}_incoming_y {inx}", f"#define {p}_outgoing_x {inx // 2}",
                  f"#define {p}_outgoing_y {inx // 2}", ""]
            xy = inx // 2
        else:
            raise ValueError(f"unknown layer_type {l['layer_type']!r}")
    o.append("#endif")
    return "\n".join(o) + "\n"


def record_geometry(layers):
    """Write incoming_x/y and outgoing_x/y into the BitConv2d / MaxPool2d dicts, as the reference's own inference_quantized does
    while it runs (BitNetMCU.py:479-483, 509-513; the fields start at 0): export_to_hfile writes them out
    (exportquant.py:222-259), so an evaluator that replaces the reference method must leave them filled in - a header exported
    after attach() would otherwise say `#define Lk_incoming_x 0`.  Same arithmetic as the reference: valid convolution, 2x2 pool."""
    xy = 16
    for l in layers:
        if l["layer_type"] == "BitConv2d":
            k = l["kernel_size"][0] if isinstance(l["kernel_size"], (tuple, list)) else int(l["kernel_size"])
            l["incoming_x"] = l["incoming_y"] = xy
            xy = xy - k + 1
            l["outgoing_x"] = l["outgoing_y"] = xy
        elif l["layer_type"] == "MaxPool2d":
            l["incoming_x"] = l["incoming_y"] = xy
            xy //= 2
            l["outgoing_x"] = l["outgoing_y"] = xy


class QuantizedEvaluator:
    """A reference `quantized_model` resident on one GPU, evaluated with the C engine's exact integer arithmetic."""

    def __init__(self, quantized_model, device=-1, modelname=None, lib=None):
        layers = getattr(quantized_model, "quantized_model", quantized_model)
        if not layers:
            raise ValueError("quantized_model is empty or None")          # BitNetMCU.py:432-433
        record_geometry(layers)
        if any(l.get("WScale") == "PerOutput" for l in layers):
            # the reference method multiplies its float logits by quantized_scale for per-output scales (BitNetMCU.py:529-531);
            # the exporter drops those scales and the C engine never sees them - this evaluator returns what the C engine computes
            import warnings
            warnings.warn("bitnetmcu_amd.evaluate: a layer uses WScale='PerOutput'; the per-output scales are not part of the exported "
                          "header, so the C engine's int32 logits (returned here) can rank classes differently from "
                          "QuantizedModel.inference_quantized, which applies them", stacklevel=3)
        self.text = header_text(layers, modelname)
        self.model = Model.from_header_text(self.text, lib)
        self.ctx = Context(self.model, device)

    def _device(self):
        import torch
        return torch.device("cuda", self.ctx._lib.bnm_ctx_device(self.ctx._h))

    def infer_int8(self, images_int8, logits=True, batch=1 << 20):
        """already-quantised int8 [n,256] images (what test_inference.py hands to the DLL) -> (class ids, logits)"""
        x = np.ascontiguousarray(images_int8, dtype=np.int8).reshape(-1, 256)
        return self.ctx.infer(x, logits=logits)

    def inference_quantized(self, input_data, batch=1 << 20):
        """Drop-in for QuantizedModel.inference_quantized (BitNetMCU.py:420-535): float inputs [n,256] (numpy or torch; any
        trailing shape with 256 elements per sample) -> the last layer's outputs [n, classes], here the C engine's int32."""
        import torch
        if hasattr(input_data, "detach"):
            input_data = input_data.detach().cpu().numpy()
        x = np.ascontiguousarray(input_data, dtype=np.float32).reshape(-1, 256)
        dev = self._device()
        out = np.empty((len(x), self.model.num_classes), np.int32)
        for s in range(0, len(x), batch):
            xb = torch.from_numpy(x[s:s + batch]).to(dev)
            cls = torch.empty(len(xb), dtype=torch.int32, device=dev)
            lg = torch.empty((len(xb), self.model.num_classes), dtype=torch.int32, device=dev)
            self.ctx.infer_device(self.ctx.quantize_device(xb), cls, lg)
            out[s:s + batch] = lg.cpu().numpy()
        return out

    def predict(self, input_data, batch=1 << 20):
        """class ids as the DLL returns them (ReLUNorm's first maximum, BitNetMCU_inference.c:25-37) — not argmax of the logits
        array, which breaks ties the same way (first index) and therefore agrees."""
        return predict(self.ctx, input_data, batch)

    def close(self):
        self.ctx.close()


def from_quantized_model(qm, device=-1, modelname=None, lib=None):
    """qm: a reference QuantizedModel (BitNetMCU.py:323) or its `.quantized_model` list of dicts."""
    return QuantizedEvaluator(qm, device, modelname, lib)


def attach(qm, device=-1, modelname=None):
    """Replace qm.inference_quantized by the GPU evaluator's (the reference's scripts then call it unchanged,
    exportquant.py:537-559, test_inference.py:153).  Returns the evaluator; the original method stays at
    qm.inference_quantized_reference.  Like the method it replaces, it leaves the conv / pool geometry (incoming_x ...) filled in
    in qm.quantized_model, which export_to_hfile depends on.  It returns the C engine's int32 logits: per-output weight scales
    (WScale='PerOutput'), which only the Python method applies and the exporter drops, are NOT applied (a warning says so)."""
    ev = from_quantized_model(qm, device, modelname)
    qm.inference_quantized_reference = qm.inference_quantized
    qm.inference_quantized = ev.inference_quantized
    return ev


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
