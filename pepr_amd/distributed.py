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


def shard_by_cost(costs, rank, world):
    """static block-cyclic by DESCENDING cost estimate (taxa x patterns).  Item ids sorted by cost
    (ties by id), dealt round-robin: every rank gets one of the `world` most expensive genes, and so on down."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    return order[rank::world]


def _device():
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


def _world_max(x):
    """MAX of a small non-negative integer over all ranks (1 element; latency-bound like the gather)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return int(x)
    t = torch.tensor([float(x)], dtype=torch.float64, device=_device())
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return int(t[0])


def gather_results(gene_ids, lnl, alpha=None, tree_length=None, newicks=None, newick_bytes=None, dst=0, status=None):
    """Gathers per-gene records {gene id, status, lnL, alpha, tree length, Newick} to rank `dst` (ONE gather).
    Every rank must hold the same number of records (pad with gene id -1).  `status`: per-record PML_* code
    (default 0).  `newick_bytes`: None/0 = sized by an all-reduce MAX of the longest Newick over all ranks; an explicit
    size that a Newick does not fit in raises (never truncates silently).  Returns on dst a dict gene_id -> record,
    elsewhere None."""
    n = len(gene_ids)
    enc = [(s or "").encode() for s in newicks] if newicks is not None else None
    if enc is not None:
        # the longest Newick over ALL ranks, always: with an explicit size every rank learns of a string that does not
        # fit and raises together, before the collective (one rank raising alone would leave the others in the gather)
        longest = _world_max(max([len(b) for b in enc], default=0))
        if not newick_bytes:
            newick_bytes = (longest + 1 + 7) // 8 * 8
        elif longest > newick_bytes:
            raise ValueError("gather_results: a %d-byte Newick (longest over all ranks) does not fit newick_bytes=%d" % (longest, newick_bytes))
    else:
        newick_bytes = 0
    width = RECORD_HEADER + (newick_bytes + 7) // 8
    rec = np.zeros((n, width))
    rec[:, 0] = gene_ids
    rec[:, 1] = status if status is not None else 0.0
    rec[:, 2] = lnl
    rec[:, 3] = alpha if alpha is not None else 0.0
    rec[:, 4] = tree_length if tree_length is not None else 0.0
    if enc is not None and newick_bytes:
        raw = np.zeros((n, (newick_bytes + 7) // 8 * 8), dtype=np.uint8)
        for i, b in enumerate(enc):
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


def jackknife(ctx, genes, reps=100, seed=0, newick_bytes=None, **kw):
    """Gene-wise jackknife over all ranks (PhylogenomicPipeline2.java:994-1126 across GPUs): rank r searches the
    replicates r, r+world, ... (pml_jackknife_opts.shard_*; every rank draws the same subsets from `seed`), rank 0
    also searches the full tree; ONE gather of the support trees, then the supports are counted on rank 0.
    A rank whose engine call fails still joins the gather (status < 0 in its records) so that nobody hangs; the
    failure is then raised on that rank and on rank 0.  Returns the single-GPU result dict on rank 0, None elsewhere."""
    from . import engine
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    err = None
    try:
        part = ctx.jackknife(genes, reps=reps, seed=seed, shard=(rank, world), **kw)
    except Exception as e:              # noqa: BLE001 -- reported through the gather, re-raised below
        if world == 1:
            raise
        err, part = e, {"newick": None, "support_trees": []}
    if world == 1:
        return part
    mine = part["support_trees"]
    per = (reps + world - 1) // world
    nmine = len(range(rank, reps, world))
    ids = [rank + world * i if i < nmine else -1 for i in range(per)]
    # a replicate the engine returned no tree for (without raising) is a failed record, never an empty Newick counted as a tree
    st = [(-5 if (err is not None or i >= len(mine) or not mine[i]) else 0) if i < nmine else 0 for i in range(per)]
    recs = gather_results(ids, np.zeros(per), newicks=(mine + [""] * per)[:per], newick_bytes=newick_bytes, status=st)
    if err is not None:
        raise err
    if rank != 0:
        return None
    bad = sorted(i for i in recs if recs[i]["status"] != 0)
    if bad:
        raise RuntimeError("jackknife: replicates %s failed on their rank (status %d)" % (bad[:8], recs[bad[0]]["status"]))
    sup = [recs[i]["newick"] for i in sorted(recs)]
    import re
    plain = re.sub(r"\)\d+:", "):", part["newick"])
    part["support_trees"] = sup
    leaves = lambda t: sorted(re.findall(r"[(,]([^(),:;]+)", t))
    # a replicate that lacks taxa of the full tree cannot contain its bipartitions (counted as not supporting)
    part["newick"] = engine.support_tree(plain, [t for t in sup if t and leaves(t) == leaves(plain)], digits=6)
    return part
