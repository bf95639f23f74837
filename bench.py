#!/usr/bin/env python3
"""bench.py — headline benchmark of the LP-relaxation hot path (BASELINE.json `metric`).

One "step" = one complete pass of the hot path over one synthetic input: a full dense-simplex solve of the
metric workload (2048x4096 fp64 dense LP, SURVEY.md §8d generator, seed 2 + rank) from HBM-resident inputs
(c, A, b uploaded before the timed region), through the C-ABI of include/gomilp_lp.h.
value = simplex pivots per second, whole job (all ranks' pivots / max-over-ranks wall time).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload M|C2|C4|C3]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line.  `roofline` prices the HBM-streaming kernel of the pipeline that ran (default: the rank-8
update of the tableau, k_bt_update_tiled) against the HBM roof with HIP-event timings of sampled launches inside the
timed region; `roofline.detail` gives the latency-bound single-workgroup kernel beside it; `cpu_baseline` times the
CPU oracle (the reference algorithm: 3 fresh LU per pivot) on a bounded sample of the same workload on the host cores.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="M")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-pivots", type=int, default=32, help="Phase-II pivots timed on the CPU oracle (about 12 s at the metric size)")
    ap.add_argument("--cpu-threads", type=int, default=16, help="OpenMP threads of the CPU oracle: a 1-GPU box owns 16 host cores; more "
                    "threads only add fork/join time to the 64-column panels (8: 2.8 pivots/s, 64: 1.2, 256: 0.09 at the metric size)")
    ap.add_argument("--sample-events", type=int, default=64, help="time the kernels of every k-th pivot (every k/K-th block) with HIP events")
    ap.add_argument("--chunk", type=int, default=64)
    ap.add_argument("--refresh", type=int, default=0)
    ap.add_argument("--frontier-vars", type=int, default=8, help="C5: 2^k children from the k highest fractional integer vars (0 = skip)")
    ap.add_argument("--workers", type=int, default=16, help="engine contexts (HIP streams) per GPU for the frontier")
    ap.add_argument("--frontier-cpu-children", type=int, default=8, help="children timed on the CPU oracle")
    ap.add_argument("--concurrent", type=int, default=4, help="extra figure: independent 2048x4096 LPs solved concurrently on one GPU (0 = skip)")
    ap.add_argument("--milp-nodes", type=int, default=127, help="C3: node budget of the host B&B over GPU relaxations (0 = skip)")
    args = ap.parse_args()

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        print("warning: --gpus %d but WORLD_SIZE %d" % (args.gpus, world), file=sys.stderr)
    if not torch.cuda.is_available():
        print("bench.py needs a GPU: the product path has no CPU fallback", file=sys.stderr)
        return 2
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist_mod.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        dist = dist_mod

    from gomilp_amd import lp, synth

    m, seed = synth.CONFIGS[args.workload]
    seed = seed + rank  # weak scaling: every rank owns an independent relaxation of the same shape
    c, A, b = synth.dense_lp_standard_form(m, seed)
    n = A.shape[1]
    ctx = lp.Context(device=local_rank, chunk=args.chunk, refresh=args.refresh, sample_events=args.sample_events)
    prob = ctx.upload(c, A, b)  # inputs resident in HBM before the timed region

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    incumbent = torch.full((1,), float("inf"), dtype=torch.float64, device="cuda")

    def step():
        r = prob.solve(0.0)
        if r.status != lp.OK:
            raise RuntimeError("solve failed: %s" % lp.STATUS_NAMES.get(r.status, r.status))
        if dist is not None:
            # the only exchange of the frontier-parallel path: incumbent bound, one all-reduce(min) over xGMI
            incumbent[0] = min(float(incumbent[0]), r.z)
            dist.all_reduce(incumbent, op=dist.ReduceOp.MIN)
        return r

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    pivots = 0
    ksec = [0.0, 0.0, 0.0, 0.0]
    loop_s = final_s = final_dev = final_host = 0.0
    last = None
    for _ in range(args.steps):
        last = step()
        pivots += last.stats["pivots_phase1"] + last.stats["pivots_phase2"]
        for i in range(4):
            ksec[i] += last.stats["pivot_kernel_seconds"][i]
        loop_s += last.stats["seconds_pivot_loop"]
        final_s += last.stats["seconds_final_solve"]
        final_dev += last.stats["seconds_final_device"]
        final_host += last.stats["seconds_final_host"]
    barrier()
    dt = time.perf_counter() - t0
    tt = torch.tensor([dt, float(pivots)], dtype=torch.float64, device="cuda")
    if dist is not None:
        tmax = tt.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tt, op=dist.ReduceOp.SUM)
        dt_max, piv_all = float(tmax[0]), float(tt[1])
    else:
        dt_max, piv_all = dt, float(pivots)

    # ---- extra: B independent LPs of the metric shape solved concurrently on this GPU (one context = one stream each).
    # The single-workgroup inner kernel of the blocked pipeline leaves most CUs idle, so independent relaxations overlap;
    # `value` above stays the single-LP figure.
    batched_out = None
    if args.concurrent > 1 and rank == 0:
        import threading
        cps = []
        for i in range(args.concurrent):
            ci, Ai, bi = synth.dense_lp_standard_form(m, synth.CONFIGS[args.workload][1] + 100 + i)
            cxi = lp.Context(device=local_rank, chunk=args.chunk)
            cps.append((cxi, cxi.upload(ci, Ai, bi)))
        resb = [None] * len(cps)

        def solve_one(i):
            resb[i] = cps[i][1].solve(0.0)

        for rep in range(2):  # first round = warm-up
            ths = [threading.Thread(target=solve_one, args=(i,)) for i in range(len(cps))]
            torch.cuda.synchronize()
            tb0 = time.perf_counter()
            for t_ in ths:
                t_.start()
            for t_ in ths:
                t_.join()
            torch.cuda.synchronize()
            tb = time.perf_counter() - tb0
        pb = sum(r.stats["pivots_phase1"] + r.stats["pivots_phase2"] for r in resb)
        batched_out = {"concurrent_lps": len(cps), "pivots": int(pb), "seconds": tb, "pivots_per_s": pb / tb,
                       "all_ok": all(r.status == lp.OK for r in resb),
                       "note": "independent %dx%d LPs (seeds +100..), one engine context per LP on one GPU" % (m, n)}
        for cxi, _ in cps:
            cxi.close()

    # ---- C5: one 256-wide B&B wave of 512x1024 relaxations, sharded over the ranks ----
    frontier_out = None
    if args.frontier_vars > 0:
        from gomilp_amd import frontier as fr
        m5, seed5 = synth.CONFIGS["C5"]
        c5, A5, b5 = synth.dense_lp_standard_form(m5, seed5)
        mask5 = synth.integrality_mask(m5, m5)
        ctx5 = lp.Context(device=local_rank)
        root5 = ctx5.upload(c5, A5, b5).solve(0.0)          # every rank solves the root (tree.go:72), outside the timing
        ctx5.close()
        children = synth.frontier_children(root5.x, mask5, args.frontier_vars)
        pool = lp.FrontierPool(device=local_rank, workers=args.workers)
        pool.set_root(c5, A5, b5)                            # root resident on every GPU before the timed region
        holder = {}

        def solve_shard(chs):
            r = pool.solve(chs)
            holder["stats"] = r.stats
            return r.status, r.z, r.x, r.has_x

        dev = torch.device("cuda", local_rank)
        fr.solve_wave(solve_shard, children, mask5, rank, world, dist, dev)  # warm-up: a full wave (first-touch allocations of every worker)
        tfs = []
        for _rep in range(3):   # median of three waves: a single wave occasionally catches a 2x outlier (host scheduling)
            barrier()
            tf0 = time.perf_counter()
            wave = fr.solve_wave(solve_shard, children, mask5, rank, world, dist, dev)
            barrier()
            tfs.append(time.perf_counter() - tf0)
        tf = sorted(tfs)[1]
        tft = torch.tensor([tf], dtype=torch.float64, device="cuda")
        st5 = holder["stats"]
        agg = torch.tensor([float(st5["pivots_phase1"] + st5["pivots_phase2"]), float(st5["phase1_runs"]),
                            float(st5["bland_steps"]), float(sum(1 for s in wave["status"] if s == lp.OK))],
                           dtype=torch.float64, device="cuda")
        if dist is not None:
            dist.all_reduce(tft, op=dist.ReduceOp.MAX)
            dist.all_reduce(agg, op=dist.ReduceOp.SUM)
        frontier_out = {
            "workload": "C5: %d children of the %dx%d root (seed %d), %d bnb rows each, dealt round-robin over a fixed shuffle to %d rank(s)"
                        % (len(children), m5, 2 * m5, seed5, args.frontier_vars, world),
            "relaxations_per_s": len(children) / float(tft[0]), "wave_seconds": float(tft[0]), "waves_timed": 3, "n_gpus": world,
            "workers_per_gpu": args.workers, "pivots": int(agg[0]), "phase1_runs": int(agg[1]), "bland_steps": int(agg[2]),
            "feasible_children": int(agg[3]), "incumbent_z": wave["incumbent_z"], "incumbent_child": wave["incumbent_index"],
            "collective": "2 x all_reduce(min) of one scalar per wave (RCCL)" if dist is not None else "none (1 rank)",
        }
        pool.close()

    # ---- C3: host branch-and-bound (tree.go semantics, gomilp_amd/bnb.py) driving GPU relaxations, rank 0 only ----
    milp_out = None
    if args.milp_nodes > 0 and rank == 0:
        from gomilp_amd import bnb
        m3, seed3 = synth.CONFIGS["C3"]
        c3, G3, h3 = synth.dense_lp_inequality_form(m3, seed3)
        int3 = [j % 4 == 0 for j in range(m3)]
        bnb.solve_milp(c3, None, None, G3, h3, int3, max_nodes=15, workers=args.workers, device=local_rank)  # warm-up
        tm0 = time.perf_counter()
        mres = bnb.solve_milp(c3, None, None, G3, h3, int3, max_nodes=args.milp_nodes, workers=args.workers, device=local_rank)
        tm = time.perf_counter() - tm0
        milp_out = {"workload": "C3: random MILP %dx%d (seed %d), 25%% integer vars, FIFO B&B, node budget %d"
                                % (m3, 2 * m3, seed3, args.milp_nodes),
                    "relaxations": mres.relaxations, "waves": mres.waves, "pivots": mres.pivots, "seconds": tm,
                    "relaxations_per_s": mres.relaxations / tm, "result": mres.error or "optimal",
                    "incumbent_z": None if mres.x is None else mres.z}
    if dist is not None:
        dist.barrier()

    if rank == 0:
        value = piv_all / dt_max
        nn = n - m
        bytes_pivot = 8.0 * (m * nn + 3.0 * m * m)      # pricing m(n-m) + FTRAN m^2 + update 2 m^2
        pipeline = last.stats["pipeline"]
        nsamp = max(ksec[3], 1.0)
        extra = {}
        if pipeline == "blocked":
            # K pivots per k_bt_inner2 launch (one workgroup, latency-bound: no HBM roofline applies to it) followed by ONE
            # streaming launch k_bt_update_tiled that reads and writes T = B^-1 A_N once: 16*m*(n-m) bytes, HBM-bound.
            # (Shapes whose block terms do not fit in registers run k_bt_inner / k_bt_update on row-major T instead.)
            nblocks = max(ksec[1], 1.0)
            kernel_name, bytes_update, bytes_moved = ("k_bt_update_tiled" if m <= 2048 and nn <= 2048 else "k_bt_update"), 16.0 * m * nn, 16.0 * m * nn
            t_upd = ksec[2] / nblocks
            t_inner = ksec[0] / nblocks
            extra = {"block_pivots": nsamp / nblocks, "k_bt_inner_us_per_launch": 1e6 * t_inner,
                     "k_bt_inner_us_per_pivot": 1e6 * ksec[0] / nsamp, "k_bt_update_us_per_pivot": 1e6 * ksec[2] / nsamp,
                     "time_share_inner": ksec[0] / max(ksec[0] + ksec[2], 1e-30),
                     "note": "k_bt_inner is a single-workgroup latency-bound kernel (it touches one column and one row of T "
                             "per pivot); the roofline entry prices the streaming kernel"}
        elif pipeline == "tableau":
            # one launch = one whole pivot: the launch is credited with the SURVEY per-pivot figure; what the
            # single-kernel formulation really moves is 16*m*(n-m) bytes (read + write T = B^-1 A_N)
            kernel_name, bytes_update, bytes_moved = "k_tableau_pivot", bytes_pivot, 16.0 * m * nn
            t_upd = ksec[2] / nsamp
        else:
            kernel_name, bytes_update, bytes_moved = "k_update", 16.0 * m * m, 16.0 * m * m   # read + write B^-1
            t_upd = ksec[2] / nsamp
        achieved = bytes_update / t_upd / 1e9 if t_upd > 0 else 0.0
        # HBM bytes per launch from the PMC counters (FETCH_SIZE doubled per the gfx950 correction + WRITE_SIZE), collected
        # with rocprofv3 --pmc in separate passes on the same command and committed under profiles/
        traffic = None
        if args.workload == "M":
            import glob
            for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json"))):   # newest tag last
                try:
                    pmc = json.load(open(path))
                except Exception:
                    continue
                if pmc.get("kernel") == kernel_name:
                    traffic = pmc["traffic_bytes_per_launch"]
        out = {
            "metric": "simplex pivots/sec on %dx%d fp64 dense LP" % (m, n),
            "value": value,
            "unit": "pivots/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * dt_max / max(args.steps, 1),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "%s: %dx%d dense LP, splitmix64 seed %d (+rank), one full solve per step"
                                   % (args.workload, m, n, synth.CONFIGS[args.workload][1]),
                       "pivots_per_solve": int(last.stats["pivots_phase2"]), "parallelism": "1 relaxation per GPU", "pipeline": last.stats["pipeline"],
                       "chunk": args.chunk, "refresh": args.refresh},
            "roofline": {"bound": "hbm", "kernel": kernel_name, "pipeline": pipeline, "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "bytes_per_launch": bytes_update, "avg_launch_us": 1e6 * t_upd,
                         "bytes_moved_model": bytes_moved, "moved_GBs": bytes_moved / t_upd / 1e9 if t_upd > 0 else 0.0,
                         "sampled_pivots": int(ksec[3]), "detail": extra,
                         "per_pivot": {"bytes": bytes_pivot, "achieved_GBs": value / world * bytes_pivot / 1e9,
                                       "frac": value / world * bytes_pivot / 1e9 / HBM_PEAK_GBS}},
            "breakdown": {"pivot_loop_s": loop_s, "final_solve_s": final_s, "final_device_s": final_dev, "final_host_s": final_host, "wall_s": dt, "drift_xb": last.stats["drift_xb"],
                          "z": last.z},
        }
        if not args.no_cpu_baseline and world == 1:   # reported at N = 1 only
            from oracle import oracle as O   # the checker, timed as the CPU baseline (never the product path)
            cores = max(1, min(args.cpu_threads, os.cpu_count() or 1))
            O.set_threads(cores)
            tcb = time.perf_counter()
            ro = O.simplex(c, A, b, 0.0, None, fast_initial_basis=True, stop_after_pivots=args.cpu_pivots)
            tcb = time.perf_counter() - tcb
            out["cpu_baseline"] = {
                "value": ro.pivots_phase2 / ro.seconds_loop if ro.seconds_loop > 0 else 0.0,
                "unit": "pivots/s", "cores": cores, "kind": "port",
                "sample": "first %d Phase-II pivots of the same %dx%d LP from the slack basis (reference algorithm: "
                          "3 fresh LU + cond estimate per pivot, gonum order); unit-column initial-basis fast path; "
                          "%.1f s wall" % (ro.pivots_phase2, m, n, tcb)}
            # a size where both engines finish the whole solve: the truncated sample above is not an artefact
            c256, A256, b256 = synth.dense_lp_standard_form(256, 7)
            O.set_threads(1)   # 256 OpenMP threads on 256x256 panels only add fork/join time
            t1 = time.perf_counter()
            r256 = O.simplex(c256, A256, b256, 0.0, None, fast_initial_basis=True)
            t_cpu = time.perf_counter() - t1
            p256 = ctx.upload(c256, A256, b256)
            p256.solve(0.0)
            t1 = time.perf_counter()
            g256 = p256.solve(0.0)
            t_gpu = time.perf_counter() - t1
            p256.free()
            npv = r256.pivots_phase1 + r256.pivots_phase2
            out["cpu_baseline"]["full_solve_256x512"] = {
                "pivots": int(npv), "cpu_cores": 1, "cpu_seconds": t_cpu, "cpu_pivots_per_s": npv / t_cpu, "gpu_seconds": t_gpu,
                "gpu_pivots_per_s": npv / t_gpu, "same_x_bits": bool(g256.status == 0 and r256.x is not None and np.array_equal(g256.x, r256.x))}
        if frontier_out is not None:
            if not args.no_cpu_baseline and world == 1 and args.frontier_cpu_children > 0:
                from concurrent.futures import ThreadPoolExecutor
                from oracle import oracle as O
                O.set_threads(1)
                sample = children[: args.frontier_cpu_children]

                def cpu_child(cons):
                    cc, AA, bb = O.child_standard_form(c5, A5, b5, cons)
                    return O.simplex(cc, AA, bb, 0.0, None, fast_initial_basis=True).status

                tcb = time.perf_counter()
                with ThreadPoolExecutor(max_workers=len(sample)) as ex:
                    list(ex.map(cpu_child, sample))
                tcb = time.perf_counter() - tcb
                frontier_out["cpu_baseline"] = {"value": len(sample) / tcb, "unit": "relaxations/s", "cores": len(sample),
                                                "kind": "port", "sample": "first %d children of the same wave, one oracle "
                                                "solve per host thread (mirrors Problem.SetWorkers), %.1f s wall" % (len(sample), tcb)}
            out["frontier"] = frontier_out
        if milp_out is not None:
            out["milp_c3"] = milp_out
        if batched_out is not None:
            out["batched"] = batched_out
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    prob.free()
    ctx.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
