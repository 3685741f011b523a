#!/usr/bin/env python
"""bench.py -- BASELINE.json metric on synthetic data.

Step = one full pass of the hot path over one batch: for every gene, transition matrices for all
branches, the complete post-order CLV pass (newview), root evaluation and the weighted lnL
reduction -- inputs (encoded alignments, trees) already resident in HBM.

Workload (genes sharded over ranks, no data-path collective):
  c3 (default): BASELINE config[2], 128 genes x (50 taxa x 1000 AA sites) per GPU, WAG+G4, f64
  c4:           BASELINE config[3] shard, 63 genes x (200 x 5000) per GPU (500 genes / 8 GPUs)
  --scaling weak (default): the per-GPU gene count is fixed as N grows;
  --scaling strong: a FIXED gene list (--genes, default 500 for c4 = BASELINE config[3] itself, 128 for c3) is dealt
                    over the N ranks block-cyclically by descending cost; a rank whose share does not fit in
                    HBM at once scores it in consecutive resident sub-batches.
`value` = M site-lnL/s = 1e-6 x alignment patterns x full-tree likelihood evaluations per second,
summed over all ranks.  Launch: python bench.py --gpus N --steps K --warmup W.  With N > 1 and no
WORLD_SIZE in the environment this process only SPAWNS the N ranks (one per GPU, RANK/LOCAL_RANK/WORLD_SIZE/MASTER_*
set) before anything touches the GPU and exits with their code; under torch.distributed.run the ranks come from the
environment and --gpus must equal WORLD_SIZE.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def _cpu_worker(args):
    """one host core: full-tree lnL of one gene, repeated for `budget_s` seconds (oracle, no GPU)"""
    import time as _t
    from oracle import po
    (names, rows, nw), alpha, budget_s = args
    m = po.Model(0)
    a = po.Alignment(names, rows); t = po.Tree(nw, a); e = po.Engine(a, m, 4, alpha)
    reps, t0 = 0, _t.time()
    while _t.time() - t0 < budget_s:
        e.set_alpha(alpha)              # invalidates every CLV: a full traversal, as on the GPU
        e.lnl(t)
        reps += 1
    return a.npat * reps, _t.time() - t0


def _cpu_search_worker(args):
    """one host core: one complete tree inference (NJ start, model optimisation, NNI search) with the oracle"""
    import time as _t
    from oracle import po
    names, rows, _ = args
    a = po.Alignment(names, rows); e = po.Engine(a, po.Model(0), 4, 1.0)
    t0 = _t.time()
    e.search(None, 0, 1e-3)
    return _t.time() - t0


def cpu_baseline(genes, alpha, budget_s=12.0):
    """Oracle (C port, oracle/pml_oracle.c) on a bounded sample of the same workload, one gene per
    host core (genes are independent, as PEPR runs one FastTree process per core), in separate
    spawned processes that never touch the GPU."""
    import multiprocessing as mp
    cores = max(1, min(len(genes), 16, os.cpu_count() or 1))
    ctx = mp.get_context("spawn")
    t0 = time.time()
    with ctx.Pool(cores) as pool:
        res = pool.map(_cpu_worker, [(genes[i], alpha, budget_s) for i in range(cores)])
        wall = time.time() - t0
        # second metric: the same search the GPU leg runs (one gene per core, one inference each); only for the small
        # workload, where a gene takes ~10-20 s of one core
        search = None
        if os.environ.get("BENCH_CPU_SEARCH", "1") != "0" and len(genes[0][0]) <= 64 and len(genes[0][1][0]) <= 1200:
            ts = time.time()
            st = pool.map(_cpu_search_worker, [genes[i] for i in range(cores)])
            swall = time.time() - ts
            search = {"gene_trees_per_sec": cores / swall, "seconds": swall, "genes": cores,
                      "sample": "the first %d genes of the workload, one complete inference each on its own core (slowest %.1f s)" % (cores, max(st))}
    units = sum(r[0] for r in res); span = max(r[1] for r in res)
    out = {"value": units / span / 1e6, "unit": "M site-lnL/s", "cores": cores, "kind": "port",
           "sample": "%d genes of the workload (one per core) x full-tree evaluations for %.0f s each (wall %.1f s), oracle/pml_oracle.c gcc -O2" % (cores, budget_s, wall)}
    if search is not None:
        out["search"] = search
    return out


class _Trees:
    """result trees of a one-shot search, with the two methods bench uses of a resident batch"""
    def __init__(self, nw):
        self.nw = nw

    def newick(self, i):
        return self.nw[i]

    def close(self):
        pass


def launch_ranks(ngpus):
    """Parent of `python bench.py --gpus N` (N > 1, no WORLD_SIZE): starts one child per GPU with the torch.distributed
    environment and waits.  It never imports torch or touches the GPU, and nothing is exec'ed from a GPU process."""
    import socket
    import subprocess
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(ngpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(ngpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    live = list(procs)
    while live:                          # a rank that dies would leave the others waiting in a collective: stop them
        time.sleep(0.2)
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0:
                rc = rc or code
                for q in live:
                    q.terminate()
    return rc


def plumbing_only(world_expected):
    """BENCH_PLUMBING_ONLY=1 (CPU rehearsal of the N > 1 path, tests/test_distributed_gloo.py): ranks, sharding and the
    one gather run for real over gloo, the engine is not touched.  Prints the same JSON shape with n_gpus."""
    import numpy as np
    from pepr_amd import distributed as pd
    rank, local, world = pd.init_from_env("gloo")
    ids = pd.shard_by_cost([1000] * 13, rank, world)
    per = (13 + world - 1) // world
    pad = ids + [-1] * (per - len(ids))
    out = pd.gather_results(pad, np.array([-100.0 - i for i in pad]), newicks=["(a:%d,b:1,c:1);" % i for i in pad],
                            status=[0] * per)
    if rank == 0:
        ok = sorted(out) == list(range(13)) and all(out[i]["lnl"] == -100.0 - i and out[i]["status"] == 0 for i in out)
        print(json.dumps({"metric": "plumbing only", "n_gpus": world, "genes_total": 13, "gathered": len(out), "ok": bool(ok)}), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier(); dist.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c3", choices=["c3", "c4", "tiny"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-search", action="store_true", help="skip the gene-trees/s leg (NJ + NNI search of every gene)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--genes", type=int, default=0, help="strong scaling: total genes of the job (default 500 for c4, 128 for c3, 16 for tiny)")
    args = ap.parse_args()

    # N ranks: either torch.distributed.run provides them (WORLD_SIZE set) or this process starts them itself, strictly
    # before any GPU call
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            sys.exit(launch_ranks(args.gpus))
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        print("bench.py: --gpus %d but WORLD_SIZE=%s" % (args.gpus, os.environ["WORLD_SIZE"]), file=sys.stderr)
        sys.exit(2)
    if os.environ.get("BENCH_PLUMBING_ONLY"):
        sys.exit(plumbing_only(args.gpus))

    import numpy as np
    import torch
    import torch.distributed as dist
    from pepr_amd import distributed as pd, engine, synth

    # rehearsal knobs (not used by the driver): BENCH_BACKEND=gloo + BENCH_FORCE_DEVICE=0 let two
    # ranks share the one GPU of a test box to exercise the N>1 path
    rank, local, world = pd.init_from_env(os.environ.get("BENCH_BACKEND"))
    if "BENCH_FORCE_DEVICE" in os.environ:
        local = int(os.environ["BENCH_FORCE_DEVICE"])
    ntax, nsites, per_gpu = {"c3": (50, 1000, 128), "c4": (200, 5000, 63), "tiny": (12, 200, 8)}[args.workload]
    alpha = 0.8
    strong = args.scaling == "strong"
    if strong:
        total = args.genes or {"c3": 128, "c4": 500, "tiny": 16}[args.workload]
        gene_ids = pd.shard_by_cost([ntax * nsites] * total, rank, world)   # synthetic genes: equal estimates -> round robin
        per_gpu = (total + world - 1) // world
    else:
        total = per_gpu * world
        gene_ids = [rank * per_gpu + i for i in range(per_gpu)]            # contiguous shard of the global list
    genes = [synth.simulate_alignment(ntax, nsites, 1 + gid, alpha) for gid in gene_ids]
    # CPU baseline first (rank 0, N=1 only): worker processes are spawned before this process
    # touches the GPU
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(genes, alpha, float(os.environ.get("BENCH_CPU_SECONDS", "12")))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP engine has no CPU fallback)")
    torch.cuda.set_device(local)
    ctx = engine.Context(local, profile=True)
    # resident sub-batches: the rank's genes are scored from HBM-resident batches; a share that does not fit at once
    # (strong scaling at small N: 500 C4 genes = 950 GB of CLVs) goes through in consecutive resident sub-batches, every
    # rank the same number of them so that the barriers pair up; the default workloads are ONE sub-batch
    free_b = torch.cuda.mem_get_info()[0]
    gene_bytes = (3 * ntax + 12) * 640 * (nsites + 32)
    nchunks = max(1, -(-len(genes) * gene_bytes // int(0.8 * free_b)))
    if world > 1:
        nchunks = pd._world_max(nchunks)
    chunks = [list(c) for c in np.array_split(np.arange(len(genes)), nchunks)]
    dt, npat, lnl_all = 0.0, 0, []
    stats = None
    CLOCK_WARMUP_S = float(os.environ.get("BENCH_CLOCK_WARMUP_S", "1.5"))
    clock_warmup_steps = 0
    for ci, idx in enumerate(chunks):
        sub = [genes[i] for i in idx]
        batch = engine.Batch(ctx, [(g[0], g[1]) for g in sub], [g[2] for g in sub], alpha=alpha) if sub else None
        npat += sum(batch.npatterns()) if batch else 0
        ids = [gene_ids[i] for i in idx] + [-1] * (len(chunks[0]) - len(idx))     # equal record counts per rank

        def step():
            return batch.score() if batch else np.zeros(0)          # ends with the results on this rank's host (synchronised)

        def gather(lnls):
            # THE one gather of per-gene results of a job (RCCL over xGMI): the K passes of the timed region are
            # one job, their K x genes records go to rank 0 in ONE collective, inside the timed region.  (A gather per
            # 1-ms pass would price the plumbing a thousand times higher than the real job does: one gather per tree build.)
            if world > 1:
                lnls = lnls or [np.zeros(0)]
                k = len(lnls)
                pd.gather_results(ids * k, np.concatenate([np.concatenate([l, np.zeros(len(ids) - len(l))]) for l in lnls]))

        for _ in range(args.warmup):
            lnl = step()
        gather([lnl] if args.warmup else [])              # also brings the communicator up before the clock starts
        # a fresh box needs a second or so of load before the chip and the host hold their clocks (the first processes after
        # box start measured 1.07 s for the search that later takes 0.83 s, profiles/r02_ab_search_warmup.txt): the same
        # untimed step, repeated until CLOCK_WARMUP_S have passed
        t_w = time.perf_counter()
        while ci == 0 and batch and time.perf_counter() - t_w < CLOCK_WARMUP_S:
            batch.score(); clock_warmup_steps += 1          # no collective here: every rank loops on its own clock
        if ci == 0:
            ctx.kernel_stats(reset=True)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        timed = []
        for _ in range(args.steps):
            lnl = step(); timed.append(lnl)
        gather(timed)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        dt += time.perf_counter() - t0
        lnl_all.append(lnl)
        if batch:
            batch.close()
    stats = ctx.kernel_stats()
    assert np.all(np.isfinite(np.concatenate(lnl_all)))

    tot_pat = npat
    if world > 1:
        t = torch.tensor([dt, float(npat)], dtype=torch.float64, device=pd._device())
        tmax = t.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt, tot_pat = float(tmax[0]), float(tsum[1])

    # second metric of BASELINE.json: complete tree inferences per second (NJ start, model
    # optimisation, NNI hill climbing), every gene of the shard, one batched call
    search = None
    if not args.no_search:
        sctx = engine.Context(local)           # no per-kernel HIP events in the timed search
        G = [(g[0], g[1]) for g in genes]

        def infer():
            """one complete inference of every gene of the shard: encode + pattern compression, NJ start trees,
            device arena, model optimisation, NNI search; returns (seconds, setup seconds, lnl, batch)"""
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if nchunks > 1:                 # does not fit at once: the one-shot call walks it in HBM-sized sub-batches
                out = sctx.search(G, None, nni=True, spr_radius=0, epsilon=1e-3)
                b, tset, l = _Trees([o["newick"] for o in out]), 0.0, np.array([o["lnl"] for o in out])
            else:
                b = engine.Batch(sctx, G, None, alpha=1.0)
                tset = time.perf_counter() - t0
                l, _ = b.search(True, True, 0, 1e-3)
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            return time.perf_counter() - t0, tset, l, b

        # first pass = cold (first touch of fresh HBM pages costs ~20 ms/GiB in the driver), second = steady state
        # of a long-lived context, which keeps the arena; both are whole inferences and both are reported
        cold, _, _, sb = infer()
        sb.close()
        warm1, _, _, sb = infer()              # second call: the context keeps its arena, clocks are up
        sb.close()
        sctx.kernel_stats(reset=True)
        sdt, tc, slnl, sb = infer()            # third call = the reported steady state
        sst = sctx.kernel_stats()
        rf = [engine.rf_distance(genes[i][2], sb.newick(i)) for i in range(len(genes))]
        sb.close()
        # the RAxML-path search (`raxmlHPC -f d`): parsimony start trees + NNI + lazy SPR radius 5, one one-shot call
        spr = None
        if ntax <= 64 and not os.environ.get("BENCH_NO_SPR"):
            if world > 1:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            sout = sctx.search(G, None, nni=True, spr_radius=5, epsilon=1e-3, seed=12345)
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
            tspr = time.perf_counter() - t0
            if world > 1:
                t = torch.tensor([tspr], dtype=torch.float64, device=pd._device())
                dist.all_reduce(t, op=dist.ReduceOp.MAX); tspr = float(t[0])
            spr = {"gene_trees_per_sec": total / tspr, "seconds": tspr,
                   "algorithm": "randomised stepwise-addition parsimony start + model optimisation + NNI + lazy SPR (radius 5), eps 1e-3",
                   "rf_to_generating_tree_mean_rank0": float(np.mean([engine.rf_distance(genes[i][2], sout[i]["newick"]) for i in range(len(genes))]))}
        sctx.close()
        if world > 1:
            t = torch.tensor([sdt, cold], dtype=torch.float64, device=pd._device())
            dist.all_reduce(t, op=dist.ReduceOp.MAX); sdt, cold = float(t[0]), float(t[1])
        search = {"gene_trees_per_sec": total / sdt, "seconds": sdt, "setup_seconds_rank0": tc,
                  "cold_first_call_seconds": cold, "cold_gene_trees_per_sec": total / cold, "second_call_seconds_rank0": warm1, "genes": total,
                  "algorithm": "NJ start + WAG+G4 model optimisation + NNI hill climbing (eps 1e-3); timed from host char rows to Newick",
                  "rf_to_generating_tree_mean_rank0": float(np.mean(rf)), "finite": bool(np.all(np.isfinite(slnl))),
                  # no closed form for a search -> measured call counts x the per-pattern byte figures / time
                  "work_rank0": {"launches": {k: v["launches"] for k, v in sst.items() if v["launches"] and not k.startswith("host")},
                                 "algorithmic_GB": {k: v["algo_bytes"] / 1e9 for k, v in sst.items() if v["algo_bytes"]},
                                 "algorithmic_TB_per_s": sum(v["algo_bytes"] for k, v in sst.items() if k in ("newview", "newton")) / sdt / 1e12}}

    if rank == 0:
        nv = stats["newview"]
        avg_ms = nv["ms"] / max(nv["launches"], 1)
        achieved = nv["algo_bytes"] / max(nv["launches"], 1) / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        # HBM bytes per launch: rocprofv3 --pmc cannot run inside this process, so this is the COMMITTED counter
        # measurement of this same command (profiles/, named in traffic_source), not a value measured in this run
        traffic, traffic_src = None, None
        for name in ("r02_pmc_traffic_%s.json" % args.workload, "r01_pmc_traffic_%s.json" % args.workload):
            pmc = os.path.join(ROOT, "profiles", name)
            if os.path.exists(pmc) and not strong:
                traffic = json.load(open(pmc))["traffic_bytes_per_launch"]
                traffic_src = "committed PMC passes, profiles/" + name
                break
        algo_flops = float(npat) / max(nchunks, 1) * (6480.0 * (ntax - 2) + 3360.0)      # one launch = one resident sub-batch of rank 0
        pipe_busy = None
        pb = os.path.join(ROOT, "profiles", "r02_pmc_k_oplist_chained_%s.json" % args.workload)
        if os.path.exists(pb) and not strong:
            pipe_busy = json.load(open(pb))["derived"]["mfma_pipe_busy_frac"]
        out = {
            "metric": "M site-lnL/sec (WAG+G4 full-tree likelihood evaluations x alignment patterns); the gene-trees/sec half of BASELINE.json's metric is search.gene_trees_per_sec",
            "value": tot_pat * args.steps / dt / 1e6, "unit": "M site-lnL/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "clock_warmup_steps": clock_warmup_steps,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s: %s x %d taxa x %d AA sites, WAG+G4 (RAxML PROTGAMMAWAG conventions), seeds 1..G" % (
                args.workload, ("%d genes in all, block-cyclic by cost" % total) if strong else ("%d genes/GPU" % per_gpu), ntax, nsites),
                "patterns_rank0": npat, "genes_total": total, "resident_sub_batches_per_rank": nchunks,
                "parallelism": "gene-sharded x%d" % world},
            # Since register chaining a scoring launch moves a fraction of SURVEY 8d's bytes (a child that
            # its parent consumes next never leaves the registers), so the pass sits under SURVEY 8d's OTHER roofline, the
            # f64 matrix pipe ("the fused scorer ... then the bound is flops"): achieved = 8d's flops per site-lnL
            # (6480 (n-2) + 3360) x the patterns of a launch / the launch's HIP-event time; peak = 78.6 TFLOP/s (vendor FP64
            # matrix figure SURVEY 8d uses; MI355X_MICROARCH.md lists none for f64; v_mfma_f64_4x4x4 measured 62-67 on the
            # box, profiles/r01_ubench_f64.txt).  "hbm" keeps the previous accounting beside it: frac = 8d's bytes / time /
            # 8 TB/s (above 1: bytes that are no longer moved), frac_traffic = counter bytes / time / 8 TB/s.
            "roofline": {"bound": "mfma", "kernel": "k_oplist (newview+evaluate)", "achieved": algo_flops / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0,
                         "peak": 78.6, "unit": "TFLOP/s", "frac": (algo_flops / (avg_ms * 1e-3) / 78.6e12) if avg_ms > 0 else 0.0,
                         "traffic": traffic, "traffic_source": traffic_src, "avg_launch_ms": avg_ms,
                         "algo_flops_per_launch": algo_flops, "matrix_pipe_busy_frac_pmc": pipe_busy,
                         "hbm": {"achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                                 "frac_traffic": (traffic / (avg_ms * 1e-3) / 8e12) if (traffic and avg_ms > 0) else None,
                                 "algo_bytes_per_launch": nv["algo_bytes"] / max(nv["launches"], 1)}},
            "kernels_ms_per_step": {k: v["ms"] / args.steps for k, v in stats.items() if v["launches"]},
        }
        if search is not None:
            out["search"] = search
            if spr is not None:
                out["search_raxml_path"] = spr
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out), flush=True)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
