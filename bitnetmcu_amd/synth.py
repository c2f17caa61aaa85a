"""Synthetic 16x16 int8 workload.  numpy statement of the generator whose host-C
statement is oracle/synth.h and whose HIP statement is synth_fill_kernel; tests compare all three."""
import numpy as np

from . import _lib as L

_M = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(z):
    z = (np.asarray(z, dtype=np.uint64) + np.uint64(0x9E3779B97F4A7C15)) & _M
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M
    return z ^ (z >> np.uint64(31))


def images(first, count, dist=L.DIST_U, seed=None):
    """int8 [count,256] for global image indices [first, first+count)."""
    if seed is None:
        seed = L.SEED_DIST_U if dist == L.DIST_U else L.SEED_DIST_M
    with np.errstate(over="ignore"):
        idx = np.arange(first, first + count, dtype=np.uint64)[:, None] * np.uint64(32) + \
            np.arange(32, dtype=np.uint64)[None, :] + np.uint64(seed)
        x = splitmix64(idx)
        sh = (np.arange(8, dtype=np.uint64) * np.uint64(8))[None, None, :]
        b = ((x[:, :, None] >> sh) & np.uint64(0xFF)).astype(np.int64)
        if dist == L.DIST_U:
            v = b.astype(np.uint8).view(np.int8)
        else:
            x2 = splitmix64(x)
            b2 = ((x2[:, :, None] >> sh) & np.uint64(0xFF)).astype(np.int64)
            v = np.where(b < 169, -20, (b2 % 148) - 20).astype(np.int8)
    return np.ascontiguousarray(v.reshape(count, 256))


# The float workload of the float-input path: every synthetic int8 image as float32 pixels v / 127 (the scale MNIST-style data
# has after ToTensor / Normalize is of this order).  Quantising it back (test_inference.py:140-141) is NOT the identity: an image
# that holds -128 has max|x| = 128/127, so its values are rescaled by 127/128 and rounded.
FLOAT_PIXEL = np.float32(1.0 / 127.0)


def float_images(first, count, dist=L.DIST_U, seed=None):
    """float32 [count,256]: images(first, count, dist) * (1/127), multiplied in float32."""
    return images(first, count, dist, seed).astype(np.float32) * FLOAT_PIXEL


def float_images_device(images_i8, out=None, chunk=1 << 22):
    """The same on the device from a resident torch.int8 [n,256] tensor (workload generation: plain torch, in chunks)."""
    import torch
    n = images_i8.shape[0]
    if out is None:
        out = torch.empty((n, 256), dtype=torch.float32, device=images_i8.device)
    for i in range(0, n, chunk):
        j = min(n, i + chunk)
        torch.mul(images_i8[i:j].to(torch.float32), float(FLOAT_PIXEL), out=out[i:j])
    return out


def class_digest(cls, first=0):
    """sum_i splitmix64((first+i)*64 + cls[i]) mod 2^64 — the order-independent digest the device
    computes with bnm_class_digest_device."""
    cls = np.asarray(cls, dtype=np.uint64)
    with np.errstate(over="ignore"):
        idx = (np.arange(first, first + len(cls), dtype=np.uint64) * np.uint64(64) + cls) & _M
        return int(np.sum(splitmix64(idx), dtype=np.uint64))


def fill_device(tensor, first=0, dist=L.DIST_U, seed=None, stream=None, lib=None):
    """Fill a torch.int8 cuda tensor [count,256] on the GPU (asynchronous, torch's current stream)."""
    import torch
    lib = lib or L.load()
    if seed is None:
        seed = L.SEED_DIST_U if dist == L.DIST_U else L.SEED_DIST_M
    assert tensor.is_cuda and tensor.is_contiguous() and tensor.dtype == torch.int8
    count = tensor.numel() // 256
    s = stream if stream is not None else torch.cuda.current_stream(tensor.device)
    L.check(lib, lib.bnm_synth_fill_device(tensor.data_ptr(), first, count, seed, dist, s.cuda_stream),
            "bnm_synth_fill_device")
    return tensor


def digest_device(cls, first=0, n_bins=10, stream=None, lib=None):
    """Returns a torch.int64 cuda tensor [1+n_bins]: digest, then the class histogram."""
    import torch
    lib = lib or L.load()
    out = torch.zeros(1 + n_bins, dtype=torch.int64, device=cls.device)
    s = stream if stream is not None else torch.cuda.current_stream(cls.device)
    L.check(lib, lib.bnm_class_digest_device(cls.data_ptr(), first, cls.numel(), out.data_ptr(), n_bins,
                                             s.cuda_stream), "bnm_class_digest_device")
    return out


def stream_rw_device(images, out, out_bytes_per_row, mode=0, stream=None, lib=None):
    """The mixed stream probe (bnm_stream_rw_device): 32-row tiles of `images` (int8 [n, 256]) read, out_bytes_per_row bytes per row
    written into `out` (a cuda tensor of >= n x out_bytes_per_row bytes); asynchronous."""
    import torch
    lib = lib or L.load()
    assert images.is_cuda and images.is_contiguous() and out.is_cuda and out.is_contiguous()
    n = images.shape[0]
    assert out.numel() * out.element_size() >= n * out_bytes_per_row
    s = stream if stream is not None else torch.cuda.current_stream(images.device)
    L.check(lib, lib.bnm_stream_rw_device(images.data_ptr(), n, out.data_ptr(), out_bytes_per_row, mode, s.cuda_stream), "bnm_stream_rw_device")
    return out


def stream_read_device(tensor, sink=None, stream=None, lib=None):
    """Plain 16 B/lane nontemporal read of a cuda tensor's bytes (bnm_stream_read_device): the box's read rate, asynchronous."""
    import torch
    lib = lib or L.load()
    assert tensor.is_cuda and tensor.is_contiguous()
    if sink is None:
        sink = torch.zeros(1, dtype=torch.int32, device=tensor.device)
    s = stream if stream is not None else torch.cuda.current_stream(tensor.device)
    L.check(lib, lib.bnm_stream_read_device(tensor.data_ptr(), tensor.numel() * tensor.element_size(), sink.data_ptr(), s.cuda_stream),
            "bnm_stream_read_device")
    return sink
