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


def _run_bench(extra_args, env_extra, timeout=600):
    import json
    import subprocess
    env = dict(os.environ, **env_extra)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        if k not in env_extra:
            env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra_args, env=env, capture_output=True, text=True, timeout=timeout)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    return p.returncode, [json.loads(l) for l in lines], p.stderr


def test_bench_spawns_its_own_ranks_cpu():
    """`python bench.py --gpus 2` with no WORLD_SIZE starts 2 ranks itself (before any GPU call), the ranks shard and
    gather over gloo, rank 0 prints ONE line with n_gpus = 2 (BENCH_PLUMBING_ONLY: the engine is not touched)"""
    rc, out, err = _run_bench(["--gpus", "2"], {"BENCH_PLUMBING_ONLY": "1"})
    assert rc == 0, err
    assert len(out) == 1 and out[0]["n_gpus"] == 2 and out[0]["ok"] and out[0]["gathered"] == 13


def test_bench_refuses_world_mismatch_cpu():
    rc, out, err = _run_bench(["--gpus", "8"], {"BENCH_PLUMBING_ONLY": "1", "WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert rc == 2 and not out and "WORLD_SIZE" in err


def test_gather_sizes_newicks_and_refuses_overflow():
    from pepr_amd import distributed as pd
    import pytest
    long_nw = "(" + ",".join("taxon_with_a_very_long_name_%04d:0.123456" % i for i in range(60)) + ");"
    out = pd.gather_results([0], np.array([-1.0]), newicks=[long_nw], status=[-5])          # sized automatically
    assert out[0]["newick"] == long_nw and out[0]["status"] == -5
    with pytest.raises(ValueError):
        pd.gather_results([0], np.array([-1.0]), newicks=[long_nw], newick_bytes=64)


def test_shard_by_cost():
    from pepr_amd import distributed as pd
    costs = [5, 9, 1, 9, 7, 3, 8]
    parts = [pd.shard_by_cost(costs, r, 3) for r in range(3)]
    assert sorted(sum(parts, [])) == list(range(7))
    assert [p[0] for p in parts] == [1, 3, 6]                # the three most expensive genes lead the three ranks
    loads = [sum(costs[i] for i in p) for p in parts]
    assert max(loads) - min(loads) <= max(costs)


import pytest as _pytest


@_pytest.mark.gpu
def test_bench_two_ranks_on_one_gpu():
    """the real N = 2 path end to end: bench.py starts its own two ranks, both drive the one GPU of the test box
    (BENCH_FORCE_DEVICE=0) and talk over gloo; weak and strong scaling"""
    import torch
    # two visible devices: one rank per GPU over RCCL (the `nccl` branch of pepr_amd/distributed.py); the one-GPU test box:
    # both ranks on device 0 over gloo
    env = {} if torch.cuda.device_count() >= 2 else {"BENCH_BACKEND": "gloo", "BENCH_FORCE_DEVICE": "0"}
    # BENCH_FORCE_C4 + a small shape: the c4_strong record of the N > 1 line (the 500-gene job dealt by cost, one gather) in miniature
    env_c4 = dict(env, BENCH_FORCE_C4="1", BENCH_C4_GENES="7", BENCH_C4_SHAPE="14,260")
    rc, out, err = _run_bench(["--gpus", "2", "--workload", "tiny", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], env_c4)
    assert rc == 0, err[-2000:]
    assert len(out) == 1 and out[0]["n_gpus"] == 2 and out[0]["scaling"] == "weak" and out[0]["config"]["genes_total"] == 16
    assert out[0]["search"]["genes"] == 16 and out[0]["search"]["finite"]
    assert 0 < out[0]["roofline"]["frac"] < 1 and out[0]["stored_traversal"]["value"] > 0 and out[0]["cold"]["value"] > 0
    c4 = out[0]["c4_strong"]
    assert c4["search_nni"]["records_gathered"] == 7 and c4["search_nni_spr5"]["records_gathered"] == 7 and c4["score"]["value"] > 0
    assert c4["search_nni"]["rank_seconds_max"] >= c4["search_nni"]["rank_seconds_min"] > 0
    rc, out, err = _run_bench(["--gpus", "2", "--workload", "tiny", "--scaling", "strong", "--genes", "11", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], env)
    assert rc == 0, err[-2000:]
    assert out[0]["n_gpus"] == 2 and out[0]["scaling"] == "strong" and out[0]["config"]["genes_total"] == 11 and out[0]["search"]["genes"] == 11


def _jk_gpu_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from pepr_amd import distributed as pd, engine, synth
    pd.init_from_env(backend="gloo")
    names = ["t%d" % i for i in range(9)]
    genes = []
    for g in range(6):
        n, rows, _ = synth.simulate_alignment(9, 120 + 10 * g, 9000 + g, names=names)
        genes.append((n, rows))
    ctx = engine.Context(0)                       # both ranks on the one GPU of the test box
    out = pd.jackknife(ctx, genes, reps=7, seed=3)
    if rank == 0:
        q.put({"newick": out["newick"], "support_trees": out["support_trees"]})
    dist.barrier()
    ctx.close()
    dist.destroy_process_group()


@_pytest.mark.gpu
def test_real_jackknife_world2_matches_world1(gpu_ctx):
    """pml_jackknife sharded over two ranks (gloo, both on GPU 0): the gathered support trees and the decorated full
    tree equal the single-process result"""
    from pepr_amd import synth
    names = ["t%d" % i for i in range(9)]
    genes = []
    for g in range(6):
        n, rows, _ = synth.simulate_alignment(9, 120 + 10 * g, 9000 + g, names=names)
        genes.append((n, rows))
    ref = gpu_ctx.jackknife(genes, reps=7, seed=3)
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_jk_gpu_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    out = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    import re
    from pepr_amd import engine
    assert out["support_trees"] == ref["support_trees"]
    assert sorted(re.findall(r"\)(\d+):", out["newick"])) == sorted(re.findall(r"\)(\d+):", ref["newick"]))
    assert engine.rf_distance(out["newick"], ref["newick"]) == 0
