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

Beside `value` the line carries
  stored_traversal   the same pass with every CLV written to HBM (what a search runs; second timed leg),
  cold               the K steps timed right after the W warm-up steps, before the extra clock warm-up,
  search / search_raxml_path   complete tree inferences per second (the other half of BASELINE.json's metric),
  c4_shard (N = 1) / c4_strong (N > 1)   BASELINE config[3]: this GPU's 63-gene share, resp. the 500-gene job dealt by cost
                     over the N ranks with the one gather of result records (BENCH_NO_C4=1 skips it).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F64_MATRIX_TFLOPS = 78.6      # vendor FP64 matrix figure SURVEY 8d uses (v_mfma_f64_4x4x4 measured 62-67 on the box, profiles/r01_ubench_f64.txt)
PEAK_HBM_GBS = 8000.0
SHAPES = {"c3": (50, 1000, 128), "c4": (200, 5000, 63), "tiny": (12, 200, 8)}


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


def _sim_worker(args):
    from pepr_amd import synth
    return synth.simulate_alignment(*args)


def simulate(ntax, nsites, gene_ids, alpha):
    """seeded synthetic genes (seed = 1 + global gene id); large ones on host worker processes that never touch the GPU"""
    from pepr_amd import synth
    jobs = [(ntax, nsites, 1 + gid, alpha) for gid in gene_ids]
    if ntax * nsites * len(jobs) < 20_000_000 or (os.cpu_count() or 1) < 4:
        return [synth.simulate_alignment(*j) for j in jobs]
    import multiprocessing as mp
    with mp.get_context("spawn").Pool(max(1, min(len(jobs), 16, os.cpu_count() or 1))) as pool:
        return pool.map(_sim_worker, jobs, chunksize=1)


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


def git_head():
    """commit of the build: git when the checkout is here, else the stamp __graft_entry__.build() left (.git does not travel to the GPU box)"""
    try:
        h = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True, timeout=10).stdout.strip()
        if h:
            return h
    except Exception:       # noqa: BLE001
        pass
    try:
        return open(os.path.join(ROOT, "pepr_amd", "BUILD_COMMIT")).read().strip() or None
    except OSError:
        return None


def committed_pmc_stored(workload):
    """the same for the stored traversal (its own counter passes: bench.py --stored-only)"""
    out = {"traffic": None, "traffic_source": None, "traffic_commit": None, "pipe_busy": None, "pipe_busy_source": None}
    f = os.path.join(ROOT, "profiles", "r03_pmc_k_oplist_stored_%s.json" % workload)
    if os.path.exists(f):
        d = json.load(open(f))
        out["traffic"] = d["derived"].get("hbm_traffic_GB") and d["derived"]["hbm_traffic_GB"] * 1e9
        out["traffic_source"] = "committed PMC passes, profiles/" + os.path.basename(f); out["traffic_commit"] = d.get("commit")
        out["pipe_busy"] = d["derived"].get("mfma_pipe_busy_frac"); out["pipe_busy_source"] = "profiles/%s @ %s" % (os.path.basename(f), d.get("commit"))
    return out


def committed_pmc(workload):
    """HBM bytes per k_oplist launch and matrix-pipe busy share from the COMMITTED rocprofv3 --pmc passes of this same command
    (separate passes per counter group, tools/pmc_collect.sh; counters cannot be read from inside the run).  Every file
    carries the commit it was measured at, which is handed through so that a stale file is visible in the line."""
    out = {"traffic": None, "traffic_source": None, "traffic_commit": None, "pipe_busy": None, "pipe_busy_source": None}
    for rnd in ("r03", "r02"):
        pmc = os.path.join(ROOT, "profiles", "%s_pmc_traffic_%s.json" % (rnd, workload))
        if out["traffic"] is None and os.path.exists(pmc):
            d = json.load(open(pmc))
            out["traffic"] = d["traffic_bytes_per_launch"]
            out["traffic_source"] = "committed PMC passes, profiles/" + os.path.basename(pmc)
            out["traffic_commit"] = d.get("commit", "not stamped (before round 3)")
        pb = os.path.join(ROOT, "profiles", "%s_pmc_k_oplist_chained_%s.json" % (rnd, workload))
        if out["pipe_busy"] is None and os.path.exists(pb):
            d = json.load(open(pb))
            out["pipe_busy"] = d["derived"]["mfma_pipe_busy_frac"]
            out["pipe_busy_source"] = "profiles/%s @ %s" % (os.path.basename(pb), d.get("commit", "not stamped (before round 3)"))
    return out


def roofline_of(nv, pmc, upper_bound_flops, stored=False):
    """roofline object for the k_oplist launches counted in the kernel statistics `nv` (HIP events on the engine's stream).
    achieved = SURVEY 8d's PER-OPERATION flops of a launch (inner-inner 6480, tip-inner 3280, tip-tip 80, evaluate 3360 per
    pattern: pml_kernel_flops) / the average launch time; peak = the f64 matrix figure.  SURVEY 8d's closed form
    6480 (n-2) + 3360 ('upper bound ignoring tip savings') is printed beside it under its own name, never as `frac`."""
    n = max(nv["launches"], 1)
    avg_ms = nv["ms"] / n
    sec = avg_ms * 1e-3
    flops = nv["algo_flops"] / n
    algo_bytes = nv["algo_bytes"] / n
    ach = flops / sec / 1e12 if sec > 0 else 0.0
    r = {"bound": "mfma", "kernel": "k_oplist (newview + evaluate)", "achieved": ach, "peak": PEAK_F64_MATRIX_TFLOPS, "unit": "TFLOP/s",
         "frac": ach / PEAK_F64_MATRIX_TFLOPS, "avg_launch_ms": avg_ms, "algo_flops_per_launch": flops,
         "traffic": pmc["traffic"], "traffic_source": pmc["traffic_source"], "traffic_commit": pmc["traffic_commit"],
         "matrix_pipe_busy_frac_pmc": pmc["pipe_busy"], "matrix_pipe_busy_source": pmc["pipe_busy_source"],
         "survey_8d_closed_form": {"flops_per_launch": upper_bound_flops, "note": "6480 (n-2) + 3360 flop per site-lnL, 8d's upper bound ignoring tip savings; NOT what the kernel executes",
                                   "tflops": upper_bound_flops / sec / 1e12 if sec > 0 else 0.0},
         # the HBM side: 8d's per-operation bytes of the op list (what an unfused traversal would move) next to what the
         # counters saw; register chaining keeps a child its parent consumes next in VGPRs, so the first can exceed the
         # chip's 8 TB/s -- it is an accounting figure, not a transfer rate, and carries no `frac`
         "hbm": {"algorithmic_bytes_per_launch": algo_bytes, "algorithmic_GB_per_s": algo_bytes / sec / 1e9 if sec > 0 else 0.0,
                 "peak_GB_per_s": PEAK_HBM_GBS,
                 "traffic_GB_per_s": (pmc["traffic"] / sec / 1e9) if (pmc["traffic"] and sec > 0) else None,
                 "traffic_frac_of_peak": (pmc["traffic"] / sec / 1e9 / PEAK_HBM_GBS) if (pmc["traffic"] and sec > 0) else None}}
    return r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c3", choices=["c3", "c4", "tiny"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-search", action="store_true", help="skip the gene-trees/s leg (NJ + NNI search of every gene)")
    ap.add_argument("--no-c4", action="store_true", help="skip the c4_shard / c4_strong record (also BENCH_NO_C4=1)")
    ap.add_argument("--no-stored", action="store_true", help="skip the stored-traversal leg (counter passes: every k_oplist launch of the run is then of one kind)")
    ap.add_argument("--stored-only", action="store_true", help="every scoring step of the run is the stored traversal (counter passes of that leg)")
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
    from pepr_amd import distributed as pd, engine

    # rehearsal knobs (not used by the driver): BENCH_BACKEND=gloo + BENCH_FORCE_DEVICE=0 let two
    # ranks share the one GPU of a test box to exercise the N>1 path
    rank, local, world = pd.init_from_env(os.environ.get("BENCH_BACKEND"))
    if "BENCH_FORCE_DEVICE" in os.environ:
        local = int(os.environ["BENCH_FORCE_DEVICE"])
    ntax, nsites, per_gpu = SHAPES[args.workload]
    alpha = 0.8
    strong = args.scaling == "strong"
    if strong:
        total = args.genes or {"c3": 128, "c4": 500, "tiny": 16}[args.workload]
        gene_ids = pd.shard_by_cost([ntax * nsites] * total, rank, world)   # synthetic genes: equal estimates -> round robin
        per_gpu = (total + world - 1) // world
    else:
        total = per_gpu * world
        gene_ids = [rank * per_gpu + i for i in range(per_gpu)]            # contiguous shard of the global list
    want_c4 = not (args.no_c4 or os.environ.get("BENCH_NO_C4") or strong) and (args.workload == "c3" or bool(os.environ.get("BENCH_FORCE_C4")))
    # everything that needs host worker processes happens first, before this process touches the GPU: the synthetic genes
    # (both workloads) and the CPU baseline (rank 0, N=1 only)
    genes = simulate(ntax, nsites, gene_ids, alpha)
    c4_ids, c4_genes = [], []
    if want_c4:
        c4_ids = pd.shard_by_cost([200 * 5000] * 500, rank if world > 1 else 0, world if world > 1 else 8)
        if os.environ.get("BENCH_C4_GENES"):                 # rehearsal knob: a smaller job (tests)
            c4_ids = pd.shard_by_cost([200 * 5000] * int(os.environ["BENCH_C4_GENES"]), rank, world)
        c4_genes = simulate(*[int(x) for x in os.environ.get("BENCH_C4_SHAPE", "200,5000").split(",")], c4_ids, alpha)
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(genes, alpha, float(os.environ.get("BENCH_CPU_SECONDS", "12")))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP engine has no CPU fallback)")
    torch.cuda.set_device(local)

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def rank_max(*vals):
        if world == 1:
            return [float(v) for v in vals]
        t = torch.tensor(list(vals), dtype=torch.float64, device=pd._device())
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return [float(x) for x in t]

    def score_legs(ctx, G, ids, steps, warmup, clock_warmup_s, stored_leg):
        """The timed scoring region over this rank's genes G (resident sub-batches when they do not fit at once; every rank the
        same number so that the barriers pair up).  Returns seconds (max over ranks), patterns of this rank, and the kernel
        statistics of the timed steps -- for the default (register-resident) pass, for the K steps before the clock warm-up,
        and for the stored traversal."""
        free_b = torch.cuda.mem_get_info()[0]
        nt, ns = len(G[0][0]), len(G[0][1][0])
        gene_bytes = (3 * nt + 12) * 640 * (ns + 32)
        nchunks = max(1, -(-len(G) * gene_bytes // int(0.8 * free_b)))
        if world > 1:
            nchunks = pd._world_max(nchunks)
        chunks = [list(c) for c in np.array_split(np.arange(len(G)), nchunks)]
        res = {"dt": 0.0, "dt_cold": 0.0, "dt_stored": 0.0, "npat": 0, "nchunks": nchunks, "clock_warmup_steps": 0, "first_step_ms": None,
               "lnl": [], "stats": None, "stats_stored": None}
        for ci, idx in enumerate(chunks):
            sub = [G[i] for i in idx]
            t_first = time.perf_counter()
            batch = engine.Batch(ctx, [(g[0], g[1]) for g in sub], [g[2] for g in sub], alpha=alpha) if sub else None
            res["npat"] += sum(batch.npatterns()) if batch else 0
            rec_ids = [ids[i] for i in idx] + [-1] * (len(chunks[0]) - len(idx))     # equal record counts per rank

            def step(stored=False):
                return batch.score(stored or args.stored_only) if batch else np.zeros(0)          # ends with the results on this rank's host (synchronised)

            def gather(lnls):
                # THE one gather of per-gene results of a job (RCCL over xGMI): the K passes of the timed region are
                # one job, their K x genes records go to rank 0 in ONE collective, inside the timed region.  (A gather per
                # 1-ms pass would price the plumbing a thousand times higher than the real job does: one gather per tree build.)
                if world > 1:
                    lnls = lnls or [np.zeros(0)]
                    pd.gather_results(rec_ids * len(lnls), np.concatenate([np.concatenate([l, np.zeros(len(rec_ids) - len(l))]) for l in lnls]))

            def timed(k, stored=False):
                sync_all()
                t0 = time.perf_counter()
                out = [step(stored) for _ in range(k)]
                gather(out)
                torch.cuda.synchronize()
                if world > 1:
                    dist.barrier()
                return time.perf_counter() - t0, out

            lnl = None
            for w in range(warmup):
                lnl = step()
                if ci == 0 and w == 0:
                    res["first_step_ms"] = (time.perf_counter() - t_first) * 1e3      # batch creation (encode, arena, H2D) + the first pass
            gather([lnl] if warmup else [])              # also brings the communicator up before the clock starts
            # cold: exactly the contract's region -- K steps right behind the W warm-up steps
            d, _ = timed(steps)
            res["dt_cold"] += d
            # a fresh box needs a second or so of load before the chip and the host hold their clocks (the first processes after
            # box start measured 1.07 s for the search that later takes 0.83 s, profiles/r02_ab_search_warmup.txt): the same
            # untimed step, repeated until clock_warmup_s have passed; `value` is the steady state behind it, `cold` the K steps before
            t_w = time.perf_counter()
            while ci == 0 and batch and time.perf_counter() - t_w < clock_warmup_s:
                step(); res["clock_warmup_steps"] += 1          # no collective here: every rank loops on its own clock
            if ci == 0:
                ctx.kernel_stats(reset=True)
            d, out = timed(steps)
            res["dt"] += d
            res["lnl"].append(out[-1])
            if ci == 0:
                res["stats"] = ctx.kernel_stats(reset=True)
            if stored_leg:
                for _ in range(max(warmup, 1)):
                    lnl_s = step(True)                   # records the stored plan
                assert np.array_equal(lnl_s, out[-1]), "the stored traversal must return the bits of the register-resident one"
                if ci == 0:
                    ctx.kernel_stats(reset=True)
                d, _ = timed(steps, True)
                res["dt_stored"] += d
                if ci == 0:
                    res["stats_stored"] = ctx.kernel_stats(reset=True)
            if batch:
                batch.close()
        assert np.all(np.isfinite(np.concatenate(res["lnl"])))
        res["dt"], res["dt_cold"], res["dt_stored"] = rank_max(res["dt"], res["dt_cold"], res["dt_stored"])
        if world > 1:
            t = torch.tensor([float(res["npat"])], dtype=torch.float64, device=pd._device())
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            res["npat_all"] = float(t[0])
        else:
            res["npat_all"] = float(res["npat"])
        return res

    ctx = engine.Context(local, profile=True)
    S = score_legs(ctx, genes, gene_ids, args.steps, args.warmup, float(os.environ.get("BENCH_CLOCK_WARMUP_S", "1.5")), not (args.no_stored or args.stored_only))
    ctx.close()
    nchunks, npat, tot_pat, dt = S["nchunks"], S["npat"], S["npat_all"], S["dt"]

    # second metric of BASELINE.json: complete tree inferences per second (NJ start, model
    # optimisation, NNI hill climbing), every gene of the shard, one batched call
    search, spr = None, None
    if not args.no_search:
        sctx = engine.Context(local)           # no per-kernel HIP events in the timed search
        G = [(g[0], g[1]) for g in genes]

        def infer():
            """one complete inference of every gene of the shard: encode + pattern compression, NJ start trees,
            device arena, model optimisation, NNI search; returns (seconds, setup seconds, lnl, batch)"""
            sync_all()
            t0 = time.perf_counter()
            # the one-shot call PEPR's tree builder would issue (pml_search_batch): it deals the genes over a few groups, each a
            # device batch on its own stream with its own host thread (api.cpp), and walks gene lists that do not fit in HBM at
            # once in sub-batches
            out = sctx.search(G, None, nni=True, spr_radius=0, epsilon=1e-3)
            b, tset, l = _Trees([o["newick"] for o in out]), 0.0, np.array([o["lnl"] for o in out])
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
        # calls three to five = the steady state; the MEDIAN is reported (the groups a search call is dealt over share the device
        # through hardware queues whose interleaving differs from call to call: 188-212 gene-trees/s on C3), all three are listed
        steady = []
        for rep in range(3):
            sctx.kernel_stats(reset=True)
            sdt_i, tc, slnl, sb = infer()
            sst_i = sctx.kernel_stats()
            steady.append((sdt_i, sst_i))
            if rep < 2:
                sb.close()
        order = sorted(range(3), key=lambda i: steady[i][0])
        sdt, sst = steady[order[1]]
        rf = [engine.rf_distance(genes[i][2], sb.newick(i)) for i in range(len(genes))]
        sb.close()
        # the RAxML-path search (`raxmlHPC -f d`): parsimony start trees + NNI + lazy SPR radius 5, one one-shot call
        if ntax <= 64 and not os.environ.get("BENCH_NO_SPR"):
            tsprs = []
            for rep in range(2):             # first call: the parsimony kernels' first use; the second is reported, both are listed
                sync_all()
                t0 = time.perf_counter()
                sout = sctx.search(G, None, nni=True, spr_radius=5, epsilon=1e-3, seed=12345)
                torch.cuda.synchronize()
                if world > 1:
                    dist.barrier()
                tsprs.append(rank_max(time.perf_counter() - t0)[0])
            tspr = tsprs[1]
            spr = {"gene_trees_per_sec": total / tspr, "seconds": tspr, "seconds_of_both_calls": tsprs,
                   "algorithm": "randomised stepwise-addition parsimony start + model optimisation + NNI + lazy SPR (radius 5), eps 1e-3",
                   "rf_to_generating_tree_mean_rank0": float(np.mean([engine.rf_distance(genes[i][2], sout[i]["newick"]) for i in range(len(genes))]))}
        nfb = sctx.newton_fallbacks()
        sctx.close()
        sdt, cold = rank_max(sdt, cold)              # (N > 1: the slowest rank's median call)
        search = {"gene_trees_per_sec": total / sdt, "seconds": sdt, "seconds_of_the_three_steady_calls_rank0": [x[0] for x in steady], "setup_seconds_rank0": tc,
                  "cold_first_call_seconds": cold, "cold_gene_trees_per_sec": total / cold, "second_call_seconds_rank0": warm1, "genes": total,
                  "algorithm": "NJ start + WAG+G4 model optimisation + NNI hill climbing (eps 1e-3); timed from host char rows to Newick",
                  "rf_to_generating_tree_mean_rank0": float(np.mean(rf)), "finite": bool(np.all(np.isfinite(slnl))),
                  # k_newton exchange waits that gave up and were re-issued through the no-exchange form (0 on a GPU the rank has to itself)
                  "newton_fallbacks_rank0": nfb,
                  # no closed form for a search -> measured call counts x the per-pattern byte figures / time
                  "work_rank0": {"launches": {k: v["launches"] for k, v in sst.items() if v["launches"] and not k.startswith("host")},
                                 "algorithmic_GB": {k: v["algo_bytes"] / 1e9 for k, v in sst.items() if v["algo_bytes"]},
                                 "algorithmic_TB_per_s": sum(v["algo_bytes"] for k, v in sst.items() if k in ("newview", "newton")) / sdt / 1e12}}

    # BASELINE config[3] (200 taxa x 5000 sites x 500 genes over 8 GPUs): at N = 1 this GPU's share of it (the 63 genes rank 0
    # of 8 gets), at N > 1 the whole 500-gene job dealt by cost over the N ranks, with the ONE gather of the result records
    c4 = None
    if want_c4 and c4_genes:
        c4 = c4_record(engine, pd, dist, torch, np, c4_genes, c4_ids, rank, local, world, alpha, sync_all, rank_max, score_legs)

    if rank == 0:
        pmc = committed_pmc(args.workload) if not strong else committed_pmc("none")
        upper = float(npat) / max(nchunks, 1) * (6480.0 * (ntax - 2) + 3360.0)      # one launch = one resident sub-batch of rank 0
        stats = S["stats"]
        out = {
            "metric": "M site-lnL/sec (WAG+G4 full-tree likelihood evaluations x alignment patterns); the gene-trees/sec half of BASELINE.json's metric is search.gene_trees_per_sec",
            "value": tot_pat * args.steps / dt / 1e6, "unit": "M site-lnL/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "clock_warmup_steps": S["clock_warmup_steps"],
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f64", "data": "synthetic", "commit": git_head(),
            "config": {"workload": "%s: %s x %d taxa x %d AA sites, WAG+G4 (RAxML PROTGAMMAWAG conventions), seeds 1..G" % (
                args.workload, ("%d genes in all, block-cyclic by cost" % total) if strong else ("%d genes/GPU" % per_gpu), ntax, nsites),
                "patterns_rank0": npat, "genes_total": total, "resident_sub_batches_per_rank": nchunks,
                "parallelism": "gene-sharded x%d" % world},
            # the K steps right behind the W warm-up steps (the contract's region to the letter); `value` is the steady state after
            # `clock_warmup_steps` more untimed steps (a fresh box holds its clocks only after ~1 s of load)
            "cold": {"value": tot_pat * args.steps / S["dt_cold"] / 1e6, "ms_per_step": S["dt_cold"] / args.steps * 1e3,
                     "first_step_ms_rank0": S["first_step_ms"], "note": "first_step = batch creation (encode, arena, upload) + plan recording + one pass"},
            "roofline": roofline_of(stats["newview"], pmc, upper),
            "kernels_ms_per_step": {k: v["ms"] / args.steps for k, v in stats.items() if v["launches"]},
        }
        if S["stats_stored"] is not None and S["dt_stored"] > 0:
            # second timed leg: the traversal a search runs -- chained children are READ from registers but every CLV is written
            # (later partial traversals read them back); same lnL bits (asserted above)
            out["stored_traversal"] = {"value": tot_pat * args.steps / S["dt_stored"] / 1e6, "unit": "M site-lnL/s",
                                       "ms_per_step": S["dt_stored"] / args.steps * 1e3,
                                       "roofline": roofline_of(S["stats_stored"]["newview"], committed_pmc_stored(args.workload) if not strong else committed_pmc("none"), upper, stored=True),
                                       "kernels_ms_per_step": {k: v["ms"] / args.steps for k, v in S["stats_stored"].items() if v["launches"]}}
        if search is not None:
            out["search"] = search
            if spr is not None:
                out["search_raxml_path"] = spr
        if c4 is not None:
            out["c4_strong" if world > 1 else "c4_shard"] = c4
        if cpu is not None:
            out["cpu_baseline"] = cpu
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def c4_record(engine, pd, dist, torch, np, genes, ids, rank, local, world, alpha, sync_all, rank_max, score_legs):
    """BASELINE config[3] at this N: scoring (M site-lnL/s), NNI search and NNI + lazy SPR (gene-trees/s) of this rank's
    share of the 500-gene job, per-rank seconds (max / min) and the time of the one gather of {id, status, lnL, alpha, tree
    length, Newick} records to rank 0."""
    nt, ns = len(genes[0][0]), len(genes[0][1][0])
    ctx = engine.Context(local, profile=True)
    S = score_legs(ctx, genes, ids, 10, 2, 0.0, False)
    ctx.close()
    total = len(ids)
    if world > 1:
        t = torch.tensor([float(total)], dtype=torch.float64, device=pd._device()); dist.all_reduce(t, op=dist.ReduceOp.SUM); total = int(t[0])
    rec = {"config": "BASELINE configs[3]: %d genes x %d taxa x %d sites %s" % (total, nt, ns, "dealt by cost over %d ranks (strong scaling)" % world if world > 1 else
                     "= the share rank 0 of 8 gets of the 500-gene job, on this one GPU"),
           "genes_rank0": len(ids), "patterns_rank0": S["npat"], "resident_sub_batches_per_rank": S["nchunks"],
           "score": {"value": S["npat_all"] * 10 / S["dt"] / 1e6, "unit": "M site-lnL/s", "ms_per_step": S["dt"] / 10 * 1e3, "steps": 10,
                     "roofline": roofline_of(S["stats"]["newview"], committed_pmc("c4"), float(S["npat"]) / S["nchunks"] * (6480.0 * (nt - 2) + 3360.0))}}
    sctx = engine.Context(local)
    G = [(g[0], g[1]) for g in genes]
    per = max(pd._world_max(len(ids)), 1) if world > 1 else len(ids)
    pad_ids = ids + [-1] * (per - len(ids))

    def job(radius, seed):
        sync_all()
        t0 = time.perf_counter()
        out = sctx.search(G, None, nni=True, spr_radius=radius, epsilon=1e-3, seed=seed) if G else []
        torch.cuda.synchronize()
        mine = time.perf_counter() - t0
        tg = time.perf_counter()
        pad = per - len(out)
        got = pd.gather_results(pad_ids, np.array([o["lnl"] for o in out] + [0.0] * pad), alpha=np.array([o["alpha"] for o in out] + [0.0] * pad),
                                tree_length=np.array([o["tree_length"] for o in out] + [0.0] * pad), newicks=[o["newick"] for o in out] + [""] * pad,
                                status=[0] * per)
        gather_ms = (time.perf_counter() - tg) * 1e3
        if world > 1:
            dist.barrier()
        wall = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([mine], dtype=torch.float64, device=pd._device())
            allt = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(allt, t)
            secs = [float(x[0]) for x in allt]
        else:
            secs = [mine]
        wall, = rank_max(wall)
        r = {"gene_trees_per_sec": total / wall, "seconds": wall, "rank_seconds_max": max(secs), "rank_seconds_min": min(secs), "gather_ms_rank0": gather_ms,
             "rf_to_generating_tree_mean_rank0": float(np.mean([engine.rf_distance(genes[i][2], out[i]["newick"]) for i in range(len(out))])) if out else None}
        if rank == 0 and got is not None:
            r["records_gathered"] = len(got)
            assert len(got) == total and all(np.isfinite(v["lnl"]) and v["status"] == 0 for v in got.values())
        return r
    rec["search_nni"] = job(0, 0)
    rec["search_nni"]["algorithm"] = "NJ start + model optimisation + NNI (eps 1e-3), host char rows -> Newick, first call of the context (cold arena)"
    if len(ids) <= 64 and not os.environ.get("BENCH_NO_WARM"):
        # the same call again: the context keeps its arena (a fresh 111 GiB arena costs ~3.7 s of driver zero-fill), clocks are up
        rec["search_nni_second_call"] = job(0, 0)
    if (len(ids) <= 64 or os.environ.get("BENCH_C4_SPR")) and not os.environ.get("BENCH_NO_SPR"):
        rec["search_nni_spr5"] = job(5, 12345)
        rec["search_nni_spr5"]["algorithm"] = "parsimony start + model optimisation + NNI + lazy SPR radius 5 (eps 1e-3), the RAxML path"
    else:
        rec["search_nni_spr5"] = None
        rec["search_nni_spr5_skipped"] = "more than 64 genes per rank: several HBM sub-batches of ~21 s each; BENCH_C4_SPR=1 runs it"
    sctx.close()
    return rec


if __name__ == "__main__":
    main()
