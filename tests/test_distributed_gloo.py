"""world_size-2 gloo test of the N>1 plumbing (sharding + the one gather), CPU only."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from pepr_amd import distributed as pd
    r, _, w = pd.init_from_env(backend="gloo")
    ids = pd.shard(7, r, w)
    ids = ids + [-1] * (4 - len(ids))                       # equal record counts per rank
    lnl = np.array([-100.0 - i if i >= 0 else 0.0 for i in ids])
    nws = ["(a:%d,b:1,c:1);" % i for i in ids]
    out = pd.gather_results(ids, lnl, alpha=0.5 + np.array(ids), tree_length=np.ones(4), newicks=nws, newick_bytes=40)
    if r == 0:
        q.put(out)
    dist.barrier()
    dist.destroy_process_group()


def test_shard_partition():
    from pepr_amd import distributed as pd
    for n in (0, 1, 7, 128, 500):
        for w in (1, 2, 8):
            parts = [pd.shard(n, r, w) for r in range(w)]
            assert sorted(sum(parts, [])) == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def test_gather_world2_gloo():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(out) == list(range(7))
    for gid, rec in out.items():
        assert rec["lnl"] == -100.0 - gid and rec["alpha"] == 0.5 + gid
        assert rec["newick"] == "(a:%d,b:1,c:1);" % gid


def test_gather_single_process():
    from pepr_amd import distributed as pd
    out = pd.gather_results([3, 5], np.array([-1.0, -2.0]), newicks=["x;", "y;"], newick_bytes=8)
    assert out[3]["lnl"] == -1.0 and out[5]["newick"] == "y;"


class _FakeCtx:
    """stands in for the GPU engine: returns canned replicate trees for the shard it is asked for"""
    TREES = ["((a:1,b:1):1,(c:1,d:1):1,e:1);", "((a:1,c:1):1,(b:1,d:1):1,e:1);", "((a:1,b:1):1,(c:1,e:1):1,d:1);",
             "((a:1,b:1):1,(c:1,d:1):1,e:1);", "((a:1,b:1):1,c:1,d:1);"]

    def jackknife(self, genes, reps, seed, shard, **kw):
        r, w = shard
        return {"newick": "((a:0.1,b:0.1)0:0.1,(c:0.1,d:0.1)0:0.1,e:0.1);" if r == 0 else None, "lnl": -1.0,
                "support_trees": self.TREES[r:reps:w]}


def _jk_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from pepr_amd import distributed as pd
    pd.init_from_env(backend="gloo")
    out = pd.jackknife(_FakeCtx(), [(["a", "b", "c", "d", "e"], None)], reps=5, seed=1)
    if rank == 0:
        q.put(out)
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_jackknife_world2_gloo():
    """replicates split over 2 ranks, support trees gathered once, supports counted on rank 0
    (the 4-taxon replicate lacks a taxon of the full tree and supports nothing)"""
    import re
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_jk_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert out["support_trees"] == _FakeCtx.TREES
    # the printed tree is re-rooted at taxon a's neighbour: split ab|cde appears as clade (cde); compare label multisets
    assert sorted(int(x) for x in re.findall(r"\)(\d+):", out["newick"])) == [2, 3]
    assert re.search(r"\(c:[0-9.]+,d:[0-9.]+\)2:", out["newick"])
