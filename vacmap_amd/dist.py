"""Multi-GPU plumbing: one process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm).

The path has no exchange step (reads are independent), so the only collectives are
  * `broadcast_index`: rank `src` builds or loads the index once, every other rank allocates an empty replica from the metadata
    (`vm_index_from_meta`) and receives the four HBM-resident pieces (`vm_index_blob`: codes, positions, hash table, contig offsets)
    straight into its own HBM by `ncclBroadcast` — the counterpart of the reference's forked workers sharing one `mp.Aligner`
    copy-on-write (/root/reference/src/vacmap/vacmap:414-420);
  * `gather_lines`: SAM text of every rank to rank 0 (variable-length host data: an object gather).
"""
import ctypes as C
import time

import numpy as np

from .lib import Index


class _DevView:
    """zero-copy view of a raw device allocation for torch (`__cuda_array_interface__`, version 2)"""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {'shape': (int(nbytes),), 'typestr': '|u1', 'data': (int(ptr), False), 'version': 2, 'strides': None}


def blob_tensor(ptr, nbytes, device):
    """uint8 torch tensor aliasing `nbytes` bytes at device address `ptr` (no copy)"""
    import torch
    if nbytes == 0:
        return torch.empty(0, dtype=torch.uint8, device=device)
    if torch.device(device).type == 'cuda':
        return torch.as_tensor(_DevView(ptr, nbytes), device=device)
    # host-backed contexts (the CPU emulator of the test-suite): plain memory
    return torch.frombuffer((C.c_uint8 * int(nbytes)).from_address(int(ptr)), dtype=torch.uint8)


def index_blobs(index, device):
    return [blob_tensor(p, n, device) for p, n in index.blobs()]


def broadcast_index(ctx, index, src=0, device=None, group=None, chunk_bytes=1 << 30, self_replica=False):
    """every rank calls this; `index` is the built index on rank `src` and None elsewhere. Returns (index on this rank, seconds).
    Pieces larger than `chunk_bytes` go in slices so that one collective never exceeds a few hundred ms of link time.
    self_replica (the world-1 test of this code on ONE GPU, VMX_FORCE_DIST=1): rank `src` also builds a replica from the metadata the way a
    receiving rank does, the collectives run on the source's pieces (zero-copy views of raw hipMalloc blocks handed to RCCL), the replica's
    pieces — the same kind of view — are filled from them by a torch copy, and the REPLICA is returned."""
    import torch
    import torch.distributed as dist
    rank = dist.get_rank(group)
    if device is None:
        device = torch.device('cuda', torch.cuda.current_device()) if torch.cuda.is_available() else torch.device('cpu')
    t0 = time.time()
    meta = [index.meta() if rank == src else None]
    dist.broadcast_object_list(meta, src=src, group=group)
    replica = None
    if rank != src:
        index = Index.from_meta(ctx, meta[0])
    elif self_replica:
        replica = Index.from_meta(ctx, meta[0])
    for t in index_blobs(index, device):
        for a in range(0, t.numel(), chunk_bytes):
            dist.broadcast(t[a:a + chunk_bytes], src=src, group=group)
    if replica is not None:
        for t, r in zip(index_blobs(index, device), index_blobs(replica, device)):
            r.copy_(t)
        index = replica
    if torch.device(device).type == 'cuda':
        torch.cuda.synchronize()
    dist.barrier(group=group)
    return index, time.time() - t0


def gather_lines(lines, dst=0, group=None):
    """list of strings per rank -> list of lists on rank `dst` (None elsewhere)"""
    import torch.distributed as dist
    out = [None] * dist.get_world_size(group) if dist.get_rank(group) == dst else None
    dist.gather_object(lines, out, dst=dst, group=group)
    return out


def shard(n_items, rank, world):
    """static sharding of the batch list: item i -> rank i mod world"""
    return np.arange(rank, n_items, world, dtype=np.int64)
