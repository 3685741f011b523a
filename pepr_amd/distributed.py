"""Gene sharding across the GPUs of one node + the single gather of per-gene results.

The path shards embarrassingly: genes (or jackknife replicates,
PhylogenomicPipeline2.java:1599-1631) are independent, so there is NO data-path collective; the
only communication is one gather of fixed-size result records per batch to rank 0
(torch.distributed backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in CPU tests).
"""
import os

import numpy as np
import torch
import torch.distributed as dist

RECORD_HEADER = 5      # gene id, status, lnL, alpha, tree length   (float64 each)


def init_from_env(backend=None):
    """One process per GPU: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from torch.distributed.run."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def shard(n_items, rank, world):
    """Block-cyclic gene -> rank assignment: item i belongs to rank i % world (costs are i.i.d.)."""
    return list(range(rank, n_items, world))


def _device():
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


def gather_results(gene_ids, lnl, alpha=None, tree_length=None, newicks=None, newick_bytes=0, dst=0):
    """Gathers per-gene records to rank `dst`.  Every rank must hold the same number of genes
    (pad with gene id -1).  Returns on dst a dict gene_id -> record, elsewhere None."""
    n = len(gene_ids)
    width = RECORD_HEADER + (newick_bytes + 7) // 8
    rec = np.zeros((n, width))
    rec[:, 0] = gene_ids
    rec[:, 2] = lnl
    rec[:, 3] = alpha if alpha is not None else 0.0
    rec[:, 4] = tree_length if tree_length is not None else 0.0
    if newicks is not None and newick_bytes:
        raw = np.zeros((n, (newick_bytes + 7) // 8 * 8), dtype=np.uint8)
        for i, s in enumerate(newicks):
            b = (s or "").encode()[:newick_bytes]
            raw[i, :len(b)] = np.frombuffer(b, dtype=np.uint8)
        rec[:, RECORD_HEADER:] = raw.view(np.float64)
    if not dist.is_initialized() or dist.get_world_size() == 1:
        parts = [rec]
    else:
        t = torch.from_numpy(rec).to(_device())
        world, rank = dist.get_world_size(), dist.get_rank()
        bufs = [torch.empty_like(t) for _ in range(world)] if rank == dst else None
        dist.gather(t, bufs, dst=dst)
        if rank != dst:
            return None
        parts = [b.cpu().numpy() for b in bufs]
    out = {}
    for part in parts:
        for row in part:
            gid = int(row[0])
            if gid < 0:
                continue
            d = {"status": int(row[1]), "lnl": float(row[2]), "alpha": float(row[3]), "tree_length": float(row[4])}
            if newick_bytes:
                d["newick"] = row[RECORD_HEADER:].copy().view(np.uint8).tobytes().split(b"\0", 1)[0].decode()
            out[gid] = d
    return out
