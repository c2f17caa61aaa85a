"""Multi-GPU sharding of the image stream: one process per GPU, torch.distributed (backend "nccl"
is RCCL on ROCm; "gloo" in the CPU tests).

Images are independent (BitMnistInference is a pure function, BitNetMCU_MNIST_dll.c:95-121), so rank g
of G owns the contiguous index range [g*N/G, (g+1)*N/G) and generates its images on its own GPU from
the global index: no image byte crosses xGMI.  The only collectives are the model broadcast from rank 0
at setup (~13 KB blob) and an optional all-reduce of the (digest, histogram) vector at the end.
"""
import numpy as np


def init_process_group_or_exit(backend, rank, world, device=None, timeout_s=60.0, exit_code=3):
    """torch.distributed.init_process_group + one barrier, under a watchdog: when the group is not up within timeout_s - a rank
    that never started, a rendezvous that cannot be reached, RCCL's communicator setup stuck below Python (the IPC transport of a
    driver this code has never met: no multi-GPU node has been available to it) - the process prints ONE line saying so to stderr
    and exits with exit_code, instead of hanging until an outer limit kills it without a reason.  backend "nccl" is RCCL."""
    import datetime
    import os
    import sys
    import threading
    import torch.distributed as td
    stage = ["rendezvous (TCP store at %s:%s)" % (os.environ.get("MASTER_ADDR", "?"), os.environ.get("MASTER_PORT", "?"))]

    def give_up():
        sys.stderr.write(f"bitnetmcu_amd.dist: rank {rank} of {world}: the '{backend}' process group did not come up within {timeout_s:g} s "
                         f"(stuck in: {stage[0]}; HSA_ENABLE_IPC_MODE_LEGACY={os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', 'unset')}) - giving up\n")
        sys.stderr.flush()
        os._exit(exit_code)

    dog = threading.Timer(timeout_s, give_up)
    dog.daemon = True
    dog.start()
    try:
        kw = {"device_id": device} if (backend == "nccl" and device is not None) else {}
        # (the group's own timeout also bounds every later collective: ten minutes; the watchdog above is what bounds the start)
        td.init_process_group(backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=max(timeout_s, 600.0)), **kw)
        stage[0] = "first barrier (communicator setup)"
        td.barrier()
    except Exception as e:      # the store's own timeout, a refused connection ...: the same one line, then out
        dog.cancel()
        sys.stderr.write(f"bitnetmcu_amd.dist: rank {rank} of {world}: the '{backend}' process group did not come up "
                         f"({stage[0]}): {type(e).__name__}: {str(e).splitlines()[0] if str(e) else ''}\n")
        sys.stderr.flush()
        os._exit(exit_code)
    dog.cancel()


def shard_range(n, rank, world):
    """[first, last) of rank's contiguous shard; shards differ by at most one image."""
    base, rem = divmod(int(n), int(world))
    first = rank * base + min(rank, rem)
    return first, first + base + (1 if rank < rem else 0)


def broadcast_blob(blob, src=0, device=None):
    """Broadcast a bytes object from `src` to every rank; returns bytes on all ranks.
    blob is ignored on ranks != src.  device: torch device for the staging tensors (cuda for nccl)."""
    import torch
    import torch.distributed as td
    rank = td.get_rank()
    dev = device if device is not None else torch.device("cpu")
    n = torch.tensor([len(blob) if rank == src else 0], dtype=torch.int64, device=dev)
    td.broadcast(n, src)
    size = int(n.item())
    if rank == src:
        buf = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(dev)
    else:
        buf = torch.empty(size, dtype=torch.uint8, device=dev)
    td.broadcast(buf, src)
    return bytes(buf.cpu().numpy().tobytes())


def broadcast_model(model, src=0, device=None, lib=None):
    """Rank `src` passes a Model (others pass None); every rank gets an equal Model back."""
    import torch.distributed as td
    from .model import Model
    blob = model.to_blob() if td.get_rank() == src else b""
    blob = broadcast_blob(blob, src, device)
    return model if td.get_rank() == src else Model.from_blob(blob, lib)


def allreduce_digest(vec):
    """Sum the int64 (digest, histogram...) vectors of all ranks (wrap-around add for the digest)."""
    import torch.distributed as td
    td.all_reduce(vec, op=td.ReduceOp.SUM)
    return vec


def combine_digests(vecs):
    """Host-side equivalent of allreduce_digest for tests: list of int64 arrays -> summed array."""
    acc = np.zeros_like(np.asarray(vecs[0], dtype=np.uint64))
    with np.errstate(over="ignore"):
        for v in vecs:
            acc = acc + np.asarray(v).astype(np.uint64)
    return acc
