#!/usr/bin/env python3
"""bench.py — headline benchmark of the LP-relaxation hot path (BASELINE.json `metric`:
"simplex pivots/sec on 2048x4096 fp64 dense LP; relaxations/sec at 1/2/4/8 GPUs").

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload M|C2|C4|C3]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
    (python bench.py --gpus N without a torchrun environment starts that launcher itself.)

N = 1: one "step" = one complete pass of the hot path over one synthetic input: a full dense-simplex solve of the metric
workload (2048x4096 fp64 dense LP, SURVEY.md §8d generator, seed 2) from HBM-resident inputs through the C-ABI of
include/gomilp_lp.h; value = simplex pivots per second.  The same line carries the 1-GPU figure of the frontier metric
(`frontier.relaxations_per_s`) the N > 1 lines scale from.

N > 1: one "step" = one 256-wide branch-and-bound wave of 512x1024 relaxations (BASELINE config 5) sharded over the N
ranks (one process per GPU, device-batched pivot loops per rank), closed by ONE RCCL all-reduce(min) of the incumbent
through the C-ABI (gomilp_incumbent_allreduce); value = relaxations per second of the whole job, scaling "strong".

Rank 0 prints ONE JSON line.  `roofline` prices the kernel with the largest share of GPU time in the timed region — the
persistent loop kernel k_bt_loop (pivot workgroups + update workgroups in one launch) — by its algorithmic bytes per full launch
(64 blocks of 8 pivots: the rank-8 update's pass over the tableau + the pivot workgroups' columns, rows and terms) over the
HIP-event time of that launch against the HBM peak; `limited_by` names what sets its pace (the pivot chain, not bandwidth); `roofline.loop` does the same with the wall time of the whole pivot loop (launch boundaries and host waits included).  `cpu_baseline` times the CPU oracle (the reference algorithm: 3 fresh LU per pivot) on a
bounded sample of the same workload on the host cores.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
LOOP_CHUNK = 512       # pivots per launch of the persistent loop kernel: handed to the engine (knob "loop_chunk") AND used by the byte model below


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="M", help="N = 1 headline workload: M (metric), C2, C3, C4")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-pivots", type=int, default=32, help="Phase-II pivots timed on the CPU oracle (about 12 s at the metric size)")
    ap.add_argument("--cpu-threads", type=int, default=16, help="OpenMP threads of the CPU oracle: a 1-GPU box owns 16 host cores; more "
                    "threads only add fork/join time to the 64-column panels (8: 2.8 pivots/s, 64: 1.2, 256: 0.09 at the metric size)")
    ap.add_argument("--sample-events", type=int, default=64, help="time the kernels of every k-th pivot (every k/K-th block) with HIP events")
    ap.add_argument("--chunk", type=int, default=64)
    ap.add_argument("--frontier-vars", type=int, default=8, help="C5: 2^k children from the k highest fractional integer vars (0 = skip at N = 1)")
    ap.add_argument("--workers", type=int, default=4, help="worker contexts per GPU (final solves / fall-backs of the batched frontier): 4 measured best — the pivot loops run in ONE batched schedule, and every further spinning host thread only delays it (16 workers: 1 wave in 6 takes 11 ms instead of 6)")
    ap.add_argument("--frontier-wide-vars", type=int, default=11, help="second frontier line at N = 1: 2^k children (0 = skip)")
    ap.add_argument("--frontier-xwide-vars", type=int, default=13, help="third frontier line: 2^k children (8192: the width at which >= 6x from 8 GPUs is arithmetically "
                    "possible — one GPU needs more than 6x the heaviest child's time for the wave) (0 = skip)")
    ap.add_argument("--milp-cpu-nodes", type=int, default=7, help="C3: nodes behind the root that the CPU oracle solves too (baseline + check; 0 = skip)")
    ap.add_argument("--frontier-cpu-children", type=int, default=8, help="children solved on the CPU oracle too (baseline + check)")
    ap.add_argument("--concurrent", type=int, default=4, help="extra figure: independent LPs of the headline shape solved together on one GPU (0 = skip)")
    ap.add_argument("--milp-nodes", type=int, default=127, help="C3: node budget of the host B&B over GPU relaxations (0 = skip)")
    ap.add_argument("--debug-one-gpu", action="store_true", help="rehearsal of the N > 1 control flow on a one-GPU box: every rank uses GPU 0, "
                    "torch.distributed over gloo, incumbent table over gloo + gomilp_incumbent_pick (RCCL refuses two ranks on one device)")
    ap.add_argument("--general", type=int, default=1, help="extra figure: an equality-constrained LP (no slack basis: the findLinearlyIndependent path) (0 = skip)")
    ap.add_argument("--pool-knob", action="append", default=[], help="key=value set on the frontier pools (A/B runs of a schedule knob, e.g. batch_virt=0); recorded in the line")
    ap.add_argument("--c4", type=int, default=1, help="extra figure: one timed solve of the 4096x8192 LP (BASELINE config 4) (0 = skip)")
    return ap.parse_args()


def newest_pmc(kernel_substr):
    """HBM bytes per launch of a kernel from the committed rocprofv3 --pmc passes (profiles/*_pmc_traffic.json, made by
    tools/pmc_traffic.py from separate FETCH_SIZE / WRITE_SIZE runs of this command).  Counters cannot be read from inside
    the run they measure: this is the most recent committed measurement, named in `traffic_source`."""
    import glob
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json"))):
        try:
            doc = json.load(open(path))
        except Exception:
            continue
        for ent in (doc if isinstance(doc, list) else [doc]):
            if kernel_substr in ent.get("kernel", "") and ent.get("traffic_bytes_per_launch") is not None:
                best = (ent.get("traffic_bytes_per_launch"), os.path.relpath(path, ROOT))
    return best or (None, None)


def mfma_object(m, nn, pipeline, ksec):
    """Where the matrix cores are used on this path and how busy they are.  The only GEMM-shaped step is the rank-16 update of
    the tableau (k_bt_update_mfma16: T += U V'^T, v_mfma_f64_16x16x4_f64, shapes beyond 1024 rows); it moves 16 bytes per 32
    flop and runs at the MALL / HBM rate, so its MFMA pipes are mostly idle by construction."""
    if pipeline != "blocked" or max(m, nn) <= 1024 or ksec[1] <= 0:
        return {"util": 0.0, "why": "this shape runs the rank-8 VALU update (1 flop per byte moved); no MFMA instruction is issued"}
    if ksec[2] <= 0:
        # persistent loop kernel: the rank-8 update of a block (two v_mfma_f64_16x16x4_f64 per 16 x 16 block of the tableau) is
        # applied by the update workgroups of the SAME launch beside the next block's pivots: busy share over the block time
        t_blk = ksec[0] / ksec[1]
        n_mfma = (m // 16) * (nn // 16) * 2.0
        return {"kernel": "k_bt_loop (update role)", "instructions_per_block": n_mfma, "tflops": n_mfma * 2048.0 / t_blk / 1e12,
                "util": (n_mfma * 64.0) / (1024 * t_blk * 2.4e9),
                "util_model": "MFMA instructions x 64 busy cycles (f64 16x16x4, counter-checked in round 2) / (1024 SIMDs x block time x 2.4 GHz)",
                "why_low": "the update moves 16 bytes per 16 flop (rank 8) and is hidden behind the latency-bound pivot chain of the same launch: "
                           "the matrix cores only have to keep it off the VALU; pricing is one row update per pivot in the tableau form, and sibling "
                           "relaxations do not share a matrix once their bases differ: there is no batched pricing GEMM"}
    t_upd = ksec[2] / ksec[1]
    n_mfma = (m // 16) * (nn // 16) * 4.0                 # wave-level v_mfma_f64_16x16x4_f64 per update launch
    flops = n_mfma * 2048.0
    simd_cycles = 1024 * t_upd * 2.4e9                     # 256 CUs x 4 SIMDs at the 2.4 GHz peak clock
    counters, src = None, None
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json"))):
        try:
            for ent in json.load(open(path)):
                if "k_bt_update_mfma16" in ent.get("kernel", "") and "mfma_counters_mean_per_launch" in ent:
                    counters, src = ent["mfma_counters_mean_per_launch"], os.path.relpath(path, ROOT)
        except Exception:
            pass
    return {"kernel": "k_bt_update_mfma16", "instructions_per_launch": n_mfma, "tflops": flops / t_upd / 1e12 if t_upd > 0 else 0.0,
            "util": (n_mfma * 64.0) / simd_cycles if t_upd > 0 else 0.0,
            "util_model": "MFMA instructions x 64 issue cycles (f64 16x16x4) / (1024 SIMDs x launch duration x 2.4 GHz)",
            "time_share_of_kernel": (ksec[2] / max(ksec[0] + ksec[2], 1e-30)),
            "counters_per_launch": counters, "counters_source": src,
            "why_low": "the update is bound by the 16 bytes it moves per element (2 flop/byte; the f64 MFMA ridge of gfx950 is ~10 flop/byte); "
                       "the matrix cores replace 32 VALU multiply-adds + 16 LDS reads per 16 bytes, which had made the VALU form of the rank-16 update "
                       "issue-bound (13.9 us against 12.5 us). Pricing itself is one row update per pivot in the tableau form, and sibling "
                       "relaxations do not share a matrix once their bases differ: there is no batched pricing GEMM"}


def main() -> int:
    args = parse_args()
    # ONE JSON line on stdout: libraries that print banners there (RCCL at communicator set-up) go to stderr instead
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    def summary_of(o):
        """The handful of figures a reader of the line's first or last 2000 characters must find (the driver keeps a tail)."""
        def g(*path):
            cur = o
            for k in path:
                if not isinstance(cur, dict) or k not in cur or cur[k] is None:
                    return None
                cur = cur[k]
            return cur
        sm = {"value": o.get("value"), "unit": o.get("unit"), "ms_per_step": o.get("ms_per_step"),
              "roofline_frac": g("roofline", "frac"), "us_per_pivot": g("roofline", "us_per_pivot"),
              "final_solve_ms": None if g("breakdown", "final_solve_s") is None else 1e3 * g("breakdown", "final_solve_s") / max(o.get("steps", 1), 1),
              "cpu_pivots_per_s": g("cpu_baseline", "value"),
              "frontier_relaxations_per_s": g("frontier", "relaxations_per_s"), "frontier_wave_ms": None if g("frontier", "wave_seconds") is None else 1e3 * g("frontier", "wave_seconds"),
              "frontier_wave_median_ms": None if g("frontier", "wave_seconds_median_rank0") is None else 1e3 * g("frontier", "wave_seconds_median_rank0"),   # (rank 0's waves: one slow wave moves the mean above by 10 %)
              "frontier_scaling_bound_per_s": g("frontier", "scaling_bound", "max_relaxations_per_s"),
              "heaviest_child_ms": None if g("frontier", "scaling_bound", "heaviest_child", "seconds_alone") is None else 1e3 * g("frontier", "scaling_bound", "heaviest_child", "seconds_alone"),
              "frontier_wide_relaxations_per_s": g("frontier_wide", "relaxations_per_s"),
              "frontier_xwide_relaxations_per_s": g("frontier_xwide", "relaxations_per_s"),
              "c4_pivots_per_s": g("c4", "pivots_per_s"), "c2_pivots_per_s": g("c2", "pivots_per_s"),
              "batched_vs_single": g("batched", "vs_single"), "batched_pivots_per_s": g("batched", "pivots_per_s"),
              "milp_c3_relaxations_per_s": g("milp_c3", "relaxations_per_s"),
              "degenerate_trees_relaxations_per_s": g("degenerate_trees", "relaxations_per_s"),
              "warm_start_c3_pivots_per_node": g("warm_start", "c3", "warm_pivots_per_node"),
              "warm_start_c3_relaxations_per_s": g("warm_start", "c3", "warm_relaxations_per_s")}
        return {k: (round(v, 4) if isinstance(v, float) else v) for k, v in sm.items() if v is not None}

    def emit(obj):
        # `summary` sits right behind the contract keys AND closes the line (`summary_tail`): whichever end a log keeps, the figures are there
        sm = summary_of(obj)
        head_keys = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config")
        line = {k: obj[k] for k in head_keys if k in obj}
        line["summary"] = sm
        for k, v in obj.items():
            if k not in line:
                line[k] = v
        line["summary_tail"] = sm
        os.write(real_stdout, (json.dumps(line) + "\n").encode())

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher environment: start it (before anything touches the GPU) and hand back its exit code
        import socket
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        os.dup2(real_stdout, 1)   # the launcher's rank 0 prints the line
        return subprocess.call(cmd)

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        print("bench.py: --gpus %d but the launcher started %d rank(s)" % (args.gpus, world), file=sys.stderr)
        return 2
    if not torch.cuda.is_available():
        print("bench.py needs a GPU: the product path has no CPU fallback", file=sys.stderr)
        return 2
    one_gpu = args.debug_one_gpu
    if one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_gpu:
            dist_mod.init_process_group(backend="gloo")
        else:
            dist_mod.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        dist = dist_mod
    red_dev = "cpu" if one_gpu else "cuda"

    from gomilp_amd import frontier as fr
    from gomilp_amd import lp, synth
    # a generation-2 collection walks every object torch's import created (~5 ms): one wave in ~8 took 10.9 ms instead of 5.6
    import gc
    gc.freeze()
    gc.disable()

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def allmax(x):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t[0])

    def allsum(vals):
        if dist is None:
            return list(vals)
        t = torch.tensor(list(vals), dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return [float(v) for v in t]

    # ------------------------------------------------------------------------------------------------------------------
    # the frontier wave (BASELINE config 5): headline for N > 1, `frontier` object of the N = 1 line
    # ------------------------------------------------------------------------------------------------------------------
    def frontier_leg(steps, warmup, nvars=None, light=False):
        nvars = args.frontier_vars if nvars is None else nvars
        m5, seed5 = synth.CONFIGS["C5"]
        c5, A5, b5 = synth.dense_lp_standard_form(m5, seed5)
        mask5 = synth.integrality_mask(m5, m5)
        ctx5 = lp.Context(device=local_rank)
        root5 = ctx5.upload(c5, A5, b5).solve(0.0)          # every rank solves the root (tree.go:72), outside the timing
        ctx5.close()
        children = synth.frontier_children(root5.x, mask5, nvars)
        pool = lp.FrontierPool(device=local_rank, workers=args.workers)   # kernel sampling (HIP events per launch) only in the extra waves below
        for kv in args.pool_knob:
            pool.set(kv.split("=")[0], int(kv.split("=")[1]))
        pool.set_root(c5, A5, b5)                            # root resident on every GPU before the timed region
        # the incumbent exchange goes through the C-ABI (RCCL), also at N = 1 (a 1-rank communicator)
        comm = None
        if not (one_gpu and world > 1):
            uid = [lp.comm_unique_id() if rank == 0 else None]
            if dist is not None:
                dist.broadcast_object_list(uid, src=0)
            comm = lp.Comm(rank, world, uid[0], device=local_rank)
        wave_dist = dist if comm is None else None   # rehearsal only: the table travels over gloo
        holder = {}

        # The wave's inputs are built ONCE, explicitly, before the timed region: this rank's shard and its flat description (koff / var /
        # sign / rhs — what a Go caller hands to gomilp_frontier_solve).  What that costs on the host is reported beside the wave
        # (host_pack_seconds), and one wave is also timed from the Python lists (wave_seconds_unpacked_rank0): like-for-like with rounds 1-3.
        tp = time.perf_counter()
        shard = fr.make_shard(children, rank, world)
        shard_packed = lp.pack_children(shard[1])
        host_pack_s = time.perf_counter() - tp

        def solve_shard(chs):
            ts = time.perf_counter()
            r = pool.solve(shard_packed if chs is shard[1] else chs)
            holder["stats"] = r.stats
            holder["solve_s"] = time.perf_counter() - ts
            return r.status, r.z, r.x, r.has_x

        dev = None if comm is None else torch.device("cuda", local_rank)
        for _ in range(max(2 if light else 6, warmup)):   # first-touch allocations and code loading of every worker end inside the first waves
            wave = fr.solve_wave(solve_shard, children, mask5, rank, world, wave_dist, dev, comm=comm, shard=shard)
        acc = dict(inner=0.0, update=0.0, blocks=0, blocks_sampled=0, batch=0.0, pivots=0, phase1=0, bland=0, fallbacks=0, batched=0)
        per_wave = []
        split = [0.0, 0.0, 0.0]   # this rank's solve / incumbent scan / exchange seconds over the timed waves
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            tw = time.perf_counter()
            wave = fr.solve_wave(solve_shard, children, mask5, rank, world, wave_dist, dev, comm=comm, shard=shard)
            per_wave.append(time.perf_counter() - tw)
            split[0] += wave["seconds_solve"]; split[1] += wave["seconds_scan"]; split[2] += wave["seconds_exchange"]
            st = holder["stats"]
            if os.environ.get("GOMILP_BENCH_DEBUG"):
                print("wave %.2f ms: solve %.2f (batch %.2f) exchange+rest %.2f" % (1e3 * per_wave[-1], 1e3 * holder["solve_s"], 1e3 * st["seconds_batch"],
                                                                                 1e3 * (per_wave[-1] - holder["solve_s"])), file=sys.stderr)
            acc["blocks"] += st["blocks"]; acc["batch"] += st["seconds_batch"]
            acc["pivots"] += st["pivots_phase1"] + st["pivots_phase2"]; acc["phase1"] += st["phase1_runs"]; acc["bland"] += st["bland_steps"]
            acc["fallbacks"] += st["host_fallbacks"]; acc["batched"] += st["batched_relaxations"]
        barrier()
        dt = allmax(time.perf_counter() - t0)
        # kernel durations: two more waves with a HIP event pair around every batched launch (outside the timed region: the
        # event records cost ~10 % of a wave)
        pool.set("sample_batch", 1)
        for _ in range(0 if light else 2):
            fr.solve_wave(solve_shard, children, mask5, rank, world, wave_dist, dev, comm=comm, shard=shard)
            st = holder["stats"]
            acc["inner"] += st["seconds_inner_kernels"]; acc["update"] += st["seconds_update_kernels"]; acc["blocks_sampled"] += st["blocks_sampled"]
        pool.set("sample_batch", 0)
        # one wave from the Python lists (shard and flat description rebuilt inside the call), outside the timed region
        tu = time.perf_counter()
        fr.solve_wave(solve_shard, children, mask5, rank, world, wave_dist, dev, comm=comm)
        unpacked_s = time.perf_counter() - tu
        tot = allsum([acc["pivots"], acc["phase1"], acc["bland"], acc["fallbacks"], acc["batched"],
                      float(sum(1 for s in wave["status"] if s == lp.OK))])
        # per rank: mean wave time and its split (one slot per rank, summed over the ranks: every rank fills its own)
        slots = [0.0] * (4 * world)
        slots[4 * rank: 4 * rank + 4] = [sum(per_wave) / max(steps, 1), split[0] / max(steps, 1), split[1] / max(steps, 1), split[2] / max(steps, 1)]
        slots = allsum(slots)
        per_rank = [{"rank": rk, "children": len(fr.shard_indices(len(children), rk, world)), "wave_ms": 1e3 * slots[4 * rk], "solve_ms": 1e3 * slots[4 * rk + 1],
                     "incumbent_scan_ms": 1e3 * slots[4 * rk + 2], "exchange_ms": 1e3 * slots[4 * rk + 3]} for rk in range(world)]
        # scaling bound: the heaviest child alone through the same batched path (a wave can never be faster than that)
        solo = None
        if rank == 0 and not light:
            res_all = pool.solve(children)
            t_best, heavy = 0.0, 0
            for i in [i for i, st_ in enumerate(res_all.status) if st_ == lp.OK]:   # the feasible children are the long ones
                t1 = time.perf_counter(); pool.solve([children[i]]); d1 = time.perf_counter() - t1
                if d1 > t_best:
                    t_best, heavy = d1, i
            for _ in range(2):
                t1 = time.perf_counter(); r1 = pool.solve([children[heavy]]); t_best = min(t_best, time.perf_counter() - t1)
            solo = {"child": heavy, "pivots": int(r1.stats["pivots_phase1"] + r1.stats["pivots_phase2"]), "bland_steps": int(r1.stats["bland_steps"]),
                    "seconds_alone": t_best}
        m_c, n_c = m5 + nvars, 2 * m5 + nvars
        nn_c = n_c - m_c
        alg_bytes = 32.0 * (m_c + nn_c)     # per pivot: column + row of T read, u and v' written (8 B each)
        inner_us = 1e6 * acc["inner"] / max(acc["blocks_sampled"], 1)
        upd_us = 1e6 * acc["update"] / max(acc["blocks_sampled"], 1)
        out = {
            "workload": "C5: %d children of the %dx%d root (seed %d), %d bnb rows each, dealt round-robin over a fixed shuffle to %d rank(s)"
                        % (len(children), m5, 2 * m5, seed5, nvars, world),
            "relaxations_per_s": steps * len(children) / dt, "wave_seconds": dt / steps, "waves_timed": steps, "n_gpus": world,
            "pool_knobs": list(args.pool_knob), "host_pack_seconds": host_pack_s, "wave_seconds_unpacked_rank0": unpacked_s,
            "per_rank": per_rank,   # exchange_ms = the ONE all-reduce(min) of the wave + the wait for the slowest rank's solve
            "wave_seconds_rank0": per_wave, "wave_seconds_median_rank0": float(np.median(per_wave)) if len(per_wave) else None, "pivots_per_wave": int(tot[0] / steps), "phase1_runs_per_wave": int(tot[1] / steps),
            "bland_steps_per_wave": int(tot[2] / steps), "host_fallbacks_per_wave": tot[3] / steps, "device_batched_per_wave": tot[4] / steps,
            "feasible_children": int(tot[5]), "incumbent_z": wave["incumbent_z"], "incumbent_child": wave["incumbent_index"],
            "collective": "gomilp_incumbent_allreduce: 1 x ncclAllReduce(min) of %d doubles per wave over %d rank(s) (RCCL, C-ABI)" % (2 * world, world),
            "schedule": "device-batched, two schedules side by side (relaxations that start feasible: the long chains, on the higher-priority stream | relaxations that need "
                        "Phase I): one launch per kernel type per block step for a whole schedule while it is wide (grid.x = relaxation), ONE persistent launch per superstep "
                        "(k_b_loop: per relaxation a pivot workgroup + 7 update workgroups, update of block t beside block t + 1) once <= 24 relaxations are active (12 per schedule while two run); "
                        "a wide schedule of slack-start relaxations runs its set-up pivot and first block on VIRTUAL tableaus (entries computed from the root's A + branch rows + the set-up term) and writes out only what is alive behind that block; "
                        "%d block steps (of 8 pivots) and %.1f host looks per wave on rank 0" % (acc["blocks"] // max(steps, 1), holder["stats"]["supersteps"]),
            "kernels_rank0": {"inner": "k_b_loop<512,2,4,7> (narrow) / k_bt_inner2_virt_batch<512,2,2,8,0> (set-up pivot + first block of a wide schedule, on computed tableau entries) / k_bt_inner2_batch<512,2,2,8,0> (wide)", "inner_us_per_launch": inner_us, "update_us_per_launch": upd_us,
                              "time_share_inner": acc["inner"] / max(acc["inner"] + acc["update"], 1e-30),
                              "algorithmic_bytes_per_pivot": alg_bytes,
                              "note": "HIP events of the sampled waves: a k_b_loop launch (up to 32 blocks of 8 pivots for every active relaxation, updates inside) counts as ONE inner launch"},
        }
        if solo is not None:
            out["scaling_bound"] = {"heaviest_child": solo, "note": "a wave cannot finish before its heaviest child: relaxations/s at any GPU count "
                                    "<= %d / %.2f ms; more GPUs only remove what the other children add to that" % (len(children), 1e3 * solo["seconds_alone"]),
                                    "max_relaxations_per_s": len(children) / solo["seconds_alone"]}
        roof = {"bound": "hbm", "bound_kind": "latency", "limited_by": "latency: one workgroup per relaxation, two workgroup-wide argmins and two dependent tableau reads per pivot",
                "kernel": "k_b_loop<512,2,4,7> + k_bt_inner2_virt_batch<512,2,2,8,0> / k_bt_inner2_batch<512,2,2,8,0>", "achieved": acc["pivots"] * alg_bytes / max(acc["inner"], 1e-30) / 1e9 if acc["inner"] > 0 else 0.0,
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": None,
                "note": "one workgroup per relaxation, two workgroup-wide argmins and two dependent tableau reads per pivot: bound by "
                        "latency, not bandwidth; achieved = rank 0's pivots x algorithmic bytes per pivot / HIP-event time of its batched inner launches"}
        roof["frac"] = roof["achieved"] / HBM_PEAK_GBS
        # the same launches with the UPDATE role's bytes: inside k_b_loop seven update workgroups read and write the relaxation's tableau once per
        # block of 4 pivots (the model the metric-size `roofline` includes); and what the counters saw (C5-only --pmc passes: tools/prof_cmds.sh step 4)
        m4_c = (m_c + 3) & ~3
        ldt_c = ((nn_c + 63) // 64) * 64
        upd_bytes = 16.0 * m4_c * ldt_c / 4.0
        roof["algorithmic_bytes_per_pivot"] = {"pivot_role": alg_bytes, "with_update_role": alg_bytes + upd_bytes,
                                               "note": "pivot role: a column and a row of T read, u and v' written; update role: T (%d x %d) read + written once per block of 4 pivots" % (m4_c, ldt_c)}
        roof["achieved_with_update_role"] = acc["pivots"] * (alg_bytes + upd_bytes) / max(acc["inner"], 1e-30) / 1e9 if acc["inner"] > 0 else 0.0
        roof["frac_with_update_role"] = roof["achieved_with_update_role"] / HBM_PEAK_GBS
        try:
            import glob
            src = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic_C5.json")))[-1]
            doc = json.load(open(src))
            loop_k = [k for k in doc["kernels"] if "k_b_loop" in k["kernel"]]
            if loop_k and nvars == 8:
                roof["traffic"] = loop_k[0]["traffic_bytes_per_wave"]
                roof["traffic_unit"] = "bytes per 256-wide wave, all k_b_loop launches of the wave (FETCH_SIZE x 2 + WRITE_SIZE, separate --pmc passes of the C5 wave alone)"
                roof["traffic_source"] = os.path.relpath(src, ROOT)
                roof["algorithmic_bytes_per_wave_with_update_role"] = (acc["pivots"] / max(steps, 1)) * (alg_bytes + upd_bytes)
        except Exception:
            pass
        cpu = None
        if not args.no_cpu_baseline and world == 1 and args.frontier_cpu_children > 0 and not light:
            from concurrent.futures import ThreadPoolExecutor
            from oracle import oracle as O   # the checker, timed as the CPU baseline (never the product path)
            O.set_threads(1)
            # the first children are the heavy feasible ones, the rest infeasible: take both kinds
            idx = list(range(args.frontier_cpu_children // 2)) + list(range(len(children) - (args.frontier_cpu_children - args.frontier_cpu_children // 2), len(children)))
            res_all = pool.solve(children)

            def cpu_child(i):
                cc, AA, bb = O.child_standard_form(c5, A5, b5, children[i])
                return O.simplex(cc, AA, bb, 0.0, None, fast_initial_basis=True)

            tcb = time.perf_counter()
            with ThreadPoolExecutor(max_workers=len(idx)) as ex:
                ores = list(ex.map(cpu_child, idx))
            tcb = time.perf_counter() - tcb
            same = all(o.status == res_all.status[i] and (o.x is None or (np.array_equal(o.x[: res_all.x.shape[1]], res_all.x[i]) and o.z == res_all.z[i]))
                       for i, o in zip(idx, ores))
            cpu = {"value": len(idx) / tcb, "unit": "relaxations/s", "cores": len(idx), "kind": "port",
                   "sample": "children %s of the same wave, one oracle solve per host thread (mirrors Problem.SetWorkers), %.1f s wall" % (idx, tcb),
                   "gpu_results_identical": bool(same)}
        if comm is not None:
            comm.close()
        pool.close()
        return out, roof, cpu

    # ------------------------------------------------------------------------------------------------------------------
    # N > 1: the sharded frontier is the headline
    # ------------------------------------------------------------------------------------------------------------------
    if world > 1:
        fout, roof, _ = frontier_leg(args.steps, args.warmup)
        wide = None
        if args.frontier_wide_vars > args.frontier_vars:
            # second line: a frontier wider than one GPU's 256 CUs (2^11 = 2048 children of the same root), sharded the same way — the
            # shape where more GPUs can pay; compare with `frontier_wide` of the --gpus 1 line
            wout, _, _ = frontier_leg(3, 2, nvars=args.frontier_wide_vars, light=True)
            wide = {k: wout[k] for k in ("workload", "relaxations_per_s", "wave_seconds", "waves_timed", "pivots_per_wave", "feasible_children",
                                         "host_fallbacks_per_wave", "device_batched_per_wave", "schedule", "per_rank")}
        xwide = None
        if args.frontier_xwide_vars > args.frontier_wide_vars:
            # third line: 2^13 = 8192 children — one GPU needs ~4x the 2048-wide wave for it, more than 6x its heaviest child: the width at
            # which the >= 6x of BASELINE.json is arithmetically reachable from 8 GPUs; compare with `frontier_xwide` of the --gpus 1 line
            xout, _, _ = frontier_leg(2, 1, nvars=args.frontier_xwide_vars, light=True)
            xwide = {k: xout[k] for k in ("workload", "relaxations_per_s", "wave_seconds", "waves_timed", "pivots_per_wave", "feasible_children",
                                          "host_fallbacks_per_wave", "device_batched_per_wave", "per_rank")}
        if rank == 0:
            out = {
                "metric": "LP relaxations/sec on the 256-wide B&B frontier of 512x1024 relaxations", "value": fout["relaxations_per_s"],
                "unit": "relaxations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": 1e3 * fout["wave_seconds"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                "dtype": "f64", "data": "synthetic",
                "config": {"workload": fout["workload"], "parallelism": "frontier sharded over %d GPUs, device-batched pivot loops per GPU" % world,
                           "scale_from": "frontier.relaxations_per_s of the --gpus 1 line"},
                "roofline": roof, "frontier": fout, "frontier_wide": wide, "frontier_xwide": xwide,
                "mfma": {"util": 0.0, "why": "no GEMM-shaped step on this path: every relaxation has its own tableau (no shared operand for a batched "
                         "pricing GEMM) and the rank-8 update is HBM-bound at 1 flop/byte (f64 MFMA ridge ~10 flop/byte)"},
            }
            emit(out)
        dist.barrier()
        dist.destroy_process_group()
        return 0

    # ------------------------------------------------------------------------------------------------------------------
    # N = 1: pivots/s on the metric LP
    # ------------------------------------------------------------------------------------------------------------------
    m, seed = synth.CONFIGS[args.workload]
    c, A, b = synth.dense_lp_standard_form(m, seed)
    n = A.shape[1]
    ctx = lp.Context(device=local_rank, chunk=args.chunk, sample_events=args.sample_events, loop_chunk=LOOP_CHUNK)
    prob = ctx.upload(c, A, b)  # inputs resident in HBM before the timed region

    def step():
        r = prob.solve(0.0)
        if r.status != lp.OK:
            raise RuntimeError("solve failed: %s" % lp.STATUS_NAMES.get(r.status, r.status))
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
    value = pivots / dt
    nn = n - m
    pipeline = last.stats["pipeline"]

    # ---- roofline: the kernel with the largest share of the timed region, then the loop, then the streaming kernel alone
    bytes_pivot_survey = 8.0 * (m * nn + 3.0 * m * m)      # SURVEY.md §8d per-unit figure (explicit-inverse model)
    if pipeline == "blocked":
        nblocks = max(ksec[1], 1.0)
        K = ksec[3] / nblocks if ksec[1] > 0 else 8.0   # pivots per block
        t_inner, t_upd = ksec[0] / nblocks, ksec[2] / nblocks
        need = max(m, nn)
        inner_name = ("k_bt_inner2<512,2,2,8,0>" if need <= 1024 else "k_bt_innerG<8,256,1,16>" if need <= 2048 else
                      "k_bt_innerG<8,512,%d,16>" % (1 if need <= 4096 else 2))
        upd_name = "k_bt_update_tiled<8>" if need <= 1024 else "k_bt_update_mfma16"
        # byte model of THIS pipeline, per block of K pivots: the inner kernel reads one column and one row of T per pivot and
        # writes u_k, v_k' (8 B each) + loads / stores r, x_B and the index lists once per launch; the update reads and
        # writes T once
        bytes_inner = K * 16.0 * (m + nn) + 2 * 12.0 * (m + nn)
        bytes_update = 16.0 * m * nn
        block_s = loop_s / max(pivots / K, 1.0)   # wall time of the loop per block: kernels + boundaries + the host's chunk waits
        if t_upd <= 0 and need > 1024:
            # persistent loop kernel: ONE kernel carries the whole pivot loop — the pivot workgroups' chain and, beside it, the
            # update workgroups' streaming pass over the tableau (read + written once per block of K = 8 pivots)
            loop_name = "k_bt_loop<16,128,1,8>" if need <= 2048 else "k_bt_loop<16,256,1,16>"
            traffic, tsrc = newest_pmc("k_bt_loop")
            bytes_block = bytes_inner + bytes_update
            bpl = max(4.0, float(LOOP_CHUNK // int(K)))   # sampled launches are full ones: loop_chunk / K blocks (engine_tableau.cpp: blocks_per_chunk)
            roofline = {
                "bound": "hbm", "bound_kind": "latency", "kernel": loop_name, "time_share": loop_s / dt,
                "achieved": bytes_block / t_inner / 1e9 if t_inner > 0 else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "traffic": traffic, "traffic_source": tsrc,
                "bytes_per_launch": bpl * bytes_block, "avg_launch_us": 1e6 * t_inner * bpl, "blocks_per_launch": int(bpl),
                "sampled_launches": int(ksec[1] / bpl),
                "bytes_per_block": bytes_block, "block_us": 1e6 * t_inner, "us_per_pivot": 1e6 * t_inner / K, "pivots_per_block": K,
                "limited_by": "the pivot workgroups' dependent-latency chain (two exchanges + two tableau reads per pivot), not by bandwidth: "
                              "the update workgroups finish a block's %.0f MB in less than the chain needs for its 8 pivots" % (bytes_update / 1e6),
                "note": "persistent kernel, one workgroup per CU or fewer: 16 pivot workgroups on one XCD run the blocks (two exchanges through that XCD's L2 and two "
                        "dependent tableau reads per pivot: the latency chain that sets the pace), the other workgroups apply the rank-8 update of block t "
                        "(tableau read + written once, matrix cores) beside block t+1.  achieved = algorithmic bytes per block (16 m (n-m) update + the pivot "
                        "workgroups' columns, rows and terms) x 64 blocks of a full launch / HIP-event time of that launch (only full launches are sampled; "
                        "rocprofv3's average also counts the short last launch of each phase)",
                "loop": {"bytes_per_block": bytes_block, "block_us": 1e6 * block_s, "kernel_us_per_block": 1e6 * t_inner,
                         "achieved_GBs": bytes_block / block_s / 1e9, "frac": bytes_block / block_s / 1e9 / HBM_PEAK_GBS, "us_per_pivot_end_to_end": 1e6 * block_s / K,
                         "model": "per block of K pivots: 16*m*(n-m) (rank-K update: T read + written once) + K*16*(m+n-m) + 24*(m+n-m) (pivot workgroups); wall time of the whole loop / blocks"},
                "sampled_blocks": int(ksec[1]), "sampled_pivots": int(ksec[3]),
                "per_pivot_survey_model": {"bytes": bytes_pivot_survey, "achieved_GBs": value * bytes_pivot_survey / 1e9, "frac": value * bytes_pivot_survey / 1e9 / HBM_PEAK_GBS,
                                           "note": "SURVEY §8d explicit-inverse model (134 MB per pivot at the metric size); the blocked tableau moves ~%.1f MB per pivot, so this ratio can exceed 1 and is not a roofline" % (bytes_block / K / 1e6)},
            }
            roofline["frac"] = roofline["achieved"] / HBM_PEAK_GBS
        else:
            share_inner = t_inner / max(t_inner + t_upd, 1e-30)
            traffic, tsrc = newest_pmc(inner_name.split("<")[0] + "<")
            roofline = {
                "bound": "hbm", "bound_kind": "latency", "limited_by": "latency (two argmins over all columns / rows and two dependent tableau reads per pivot)", "kernel": inner_name,
                "time_share": share_inner * loop_s / dt,
                "achieved": bytes_inner / t_inner / 1e9 if t_inner > 0 else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "traffic": traffic, "traffic_source": tsrc,
                "bytes_per_launch": bytes_inner, "avg_launch_us": 1e6 * t_inner, "us_per_pivot": 1e6 * t_inner / K, "pivots_per_launch": K,
                "note": ("single-workgroup kernel (1 of 256 CUs)" if need <= 1024 else "8 workgroups of one XCD (8 of 256 CUs), two exchanges through that XCD's L2 per pivot") +
                        ": two first-index argmins over all columns / rows and two dependent tableau reads per "
                        "pivot; its roof is the dependent-latency chain, not HBM bandwidth — `frac` is reported against the HBM peak all the same",
                "loop": {"bytes_per_block": bytes_inner + bytes_update, "block_us": 1e6 * block_s,
                         "kernel_us_per_block": 1e6 * (t_inner + t_upd), "achieved_GBs": (bytes_inner + bytes_update) / block_s / 1e9,
                         "frac": (bytes_inner + bytes_update) / block_s / 1e9 / HBM_PEAK_GBS,
                         "model": "per block of K pivots: 16*m*(n-m) (rank-K update: T read + written once) + K*16*(m+n-m) + 24*(m+n-m) (block kernel)"},
                "streaming_kernel": {"bound": "hbm", "kernel": upd_name, "bytes_per_launch": bytes_update, "avg_launch_us": 1e6 * t_upd,
                                     "achieved": bytes_update / t_upd / 1e9 if t_upd > 0 else 0.0, "frac": bytes_update / t_upd / 1e9 / HBM_PEAK_GBS if t_upd > 0 else 0.0,
                                     "traffic": newest_pmc(upd_name.split("<")[0])[0], "traffic_source": newest_pmc(upd_name.split("<")[0])[1],
                                     "time_share": (1 - share_inner) * loop_s / dt,
                                     "note": "the %.1f MB tableau stays in the 256 MB Infinity Cache between launches: a MALL rate where it exceeds the ~6.3 TB/s HBM copy rate" % (8e-6 * m * nn)},
                "sampled_blocks": int(ksec[1]), "sampled_pivots": int(ksec[3]),
                "per_pivot_survey_model": {"bytes": bytes_pivot_survey, "achieved_GBs": value * bytes_pivot_survey / 1e9, "frac": value * bytes_pivot_survey / 1e9 / HBM_PEAK_GBS,
                                           "note": "SURVEY §8d explicit-inverse model (134 MB per pivot at the metric size); the blocked tableau moves ~%.1f MB per pivot, so this ratio can exceed 1 and is not a roofline" % ((bytes_inner + bytes_update) / K / 1e6)},
            }
            roofline["frac"] = roofline["achieved"] / HBM_PEAK_GBS
    else:
        nsamp = max(ksec[3], 1.0)
        t_upd = ksec[2] / nsamp
        kname, bts = ("k_tableau_pivot", 16.0 * m * nn) if pipeline == "tableau" else ("k_update", 16.0 * m * m)
        roofline = {"bound": "hbm", "kernel": kname, "achieved": bts / t_upd / 1e9 if t_upd > 0 else 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "traffic": None, "bytes_per_launch": bts, "avg_launch_us": 1e6 * t_upd}
        roofline["frac"] = roofline["achieved"] / HBM_PEAK_GBS

    out = {
        "metric": "simplex pivots/sec on %dx%d fp64 dense LP" % (m, n), "value": value, "unit": "pivots/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / max(args.steps, 1), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "%s: %dx%d dense LP, splitmix64 seed %d, one full solve per step" % (args.workload, m, n, seed),
                   "pivots_per_solve": int(last.stats["pivots_phase2"]), "parallelism": "1 relaxation on 1 GPU", "pipeline": pipeline, "chunk": args.chunk},
        "roofline": roofline,
        "mfma": mfma_object(m, nn, pipeline, ksec),
        "breakdown": {"pivot_loop_s": loop_s, "final_solve_s": final_s, "final_device_s": final_dev, "final_host_s": final_host, "wall_s": dt,
                      "drift_xb": last.stats["drift_xb"], "z": last.z},
    }

    # ---- CPU baseline (the oracle = the reference algorithm), bounded sample of the same workload.  Timed LAST (deferred below): 15-25 s
    # of 16 busy host threads in front of the latency-bound GPU legs cost those legs 5-10 % (frontier wave 3.7 -> 4.1 ms on the same box)
    deferred = []
    def _cpu_headline():
        from oracle import oracle as O   # the checker, timed as the CPU baseline (never the product path)
        cores = max(1, min(args.cpu_threads, os.cpu_count() or 1))
        O.set_threads(cores)
        tcb = time.perf_counter()
        ro = O.simplex(c, A, b, 0.0, None, fast_initial_basis=True, stop_after_pivots=args.cpu_pivots, trace=True)
        tcb = time.perf_counter() - tcb
        gtr = prob.solve(0.0, trace=True)
        same_prefix = [(p[0], p[2], p[3], p[4], p[5]) for p in ro.pivots] == [(p[0], p[2], p[3], p[4], p[5]) for p in gtr.pivots[: len(ro.pivots)]]
        out["cpu_baseline"] = {
            "value": ro.pivots_phase2 / ro.seconds_loop if ro.seconds_loop > 0 else 0.0, "unit": "pivots/s", "cores": cores, "kind": "port",
            "sample": "first %d Phase-II pivots of the same %dx%d LP from the slack basis (reference algorithm: 3 fresh LU + cond estimate "
                      "per pivot, gonum order); unit-column initial-basis fast path; %.1f s wall" % (ro.pivots_phase2, m, n, tcb),
            "gpu_pivots_identical_on_sample": bool(same_prefix)}
        # a size where both engines finish the whole solve: the truncated sample above is not an artefact
        c256, A256, b256 = synth.dense_lp_standard_form(256, 7)
        O.set_threads(1)
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
    if not args.no_cpu_baseline:
        deferred.append(_cpu_headline)

    # ---- extra: B independent LPs of the headline shape on this GPU, advancing TOGETHER through the device-batched schedule
    # (children with no branch rows of B roots in one pool: one launch per kernel type per block step for all of them)
    if args.concurrent > 1:
        poolb = lp.FrontierPool(device=local_rank, workers=min(args.workers, args.concurrent))
        lps = [synth.dense_lp_standard_form(m, seed + 100 + i) for i in range(args.concurrent)]
        poolb.set_root(*lps[0])
        roots = [0] + [poolb.add_root(*q) for q in lps[1:]]
        tbs, resb = [], None
        for rep in range(6):  # first round = warm-up; the MEDIAN of the other five is reported (one sample swung between 24 and 38 ms from run to run: r5b / r5e / r5f)
            torch.cuda.synchronize()
            tb0 = time.perf_counter()
            resb = poolb.solve([[] for _ in roots], roots=roots)
            torch.cuda.synchronize()
            if rep: tbs.append(time.perf_counter() - tb0)
        tb = sorted(tbs)[len(tbs) // 2]
        pb = resb.stats["pivots_phase1"] + resb.stats["pivots_phase2"]
        out["batched"] = {"concurrent_lps": len(roots), "pivots": int(pb), "seconds": tb, "seconds_all": tbs, "pivots_per_s": pb / tb, "vs_single": pb / tb / value,
                          "all_ok": bool((resb.status == lp.OK).all()), "device_batched": int(resb.stats["batched_relaxations"]),
                          "host_round_trips": int(resb.stats["supersteps"]),
                          "note": ("independent %dx%d LPs (seeds +100..) through gomilp_frontier_solve_roots on one GPU: " % (m, n)) +
                                  ("the pool's workers run them on persistent loop kernels side by side (up to four launches share the device, each with its pivot "
                                   "workgroups on an XCD of its own; device_batched = 0)" if int(resb.stats["batched_relaxations"]) == 0 else
                                   "one device-batched schedule: k_bt_innerG_batch, one XCD per LP, + the batched MFMA rank-16 update")}
        poolb.close()

    # ---- BASELINE config 2: the 1024x2048 LP (one relaxation, no B&B)
    if args.c4 and args.workload == "M":
        m2, seed2 = synth.CONFIGS["C2"]
        c2, A2, b2 = synth.dense_lp_standard_form(m2, seed2)
        cx2 = lp.Context(device=local_rank)
        p2 = cx2.upload(c2, A2, b2)
        p2.solve(0.0)
        t1 = time.perf_counter()
        r2 = p2.solve(0.0)
        t2 = time.perf_counter() - t1
        out["c2"] = {"workload": "C2: %dx%d dense LP (seed %d), one full solve" % (m2, 2 * m2, seed2), "status": int(r2.status),
                     "pivots": int(r2.stats["pivots_phase2"]), "seconds": t2, "pivots_per_s": r2.stats["pivots_phase2"] / t2,
                     "loop_us_per_pivot": 1e6 * r2.stats["seconds_pivot_loop"] / max(r2.stats["pivots_phase2"], 1),
                     "note": "persistent loop kernel (16 x 128-thread pivot workgroups, blocks of 8), as at the metric size"}
        p2.free()
        cx2.close()
        del c2, A2, b2

    # ---- BASELINE config 4: one solve of the 4096x8192 LP
    if args.c4 and args.workload != "C4":
        m4, seed4 = synth.CONFIGS["C4"]
        c4, A4, b4 = synth.dense_lp_standard_form(m4, seed4)
        cx4 = lp.Context(device=local_rank, chunk=args.chunk, sample_events=args.sample_events)
        p4 = cx4.upload(c4, A4, b4)
        p4.solve(0.0)
        t1 = time.perf_counter()
        r4 = p4.solve(0.0)
        t4 = time.perf_counter() - t1
        k4 = r4.stats["pivot_kernel_seconds"]
        nb4 = max(k4[1], 1.0)
        out["c4"] = {"workload": "C4: %dx%d dense LP (seed %d), one full solve" % (m4, 2 * m4, seed4), "status": int(r4.status),
                     "pivots": int(r4.stats["pivots_phase2"]), "seconds": t4, "pivots_per_s": r4.stats["pivots_phase2"] / t4,
                     "loop_us_per_pivot": 1e6 * r4.stats["seconds_pivot_loop"] / max(r4.stats["pivots_phase2"], 1),
                     "block_GBs": (16.0 * m4 * m4 + (k4[3] / nb4) * 32.0 * m4) / (k4[0] / nb4) / 1e9 if k4[0] > 0 else 0.0,
                     "inner_us_per_launch": 1e6 * k4[0] / nb4, "update_us_per_launch": 1e6 * k4[2] / nb4, "pivots_per_launch": k4[3] / nb4,
                     "update_GBs": 16.0 * m4 * m4 / (k4[2] / nb4) / 1e9 if k4[2] > 0 else 0.0,
                     "update_frac_of_hbm_peak": 16.0 * m4 * m4 / (k4[2] / nb4) / 1e9 / HBM_PEAK_GBS if k4[2] > 0 else 0.0,
                     "inner_kernel": "k_bt_loop<16,256,1,16>", "us_per_block": 1e6 * k4[0] / nb4,
                     "note": "persistent loop kernel: blocks of 16 pivots on 16 pivot workgroups of 256 threads on one XCD (16 current + 16 lagging terms per row / column in "
                             "registers), the rank-16 update of block t (matrix cores) on the other workgroups beside block t+1; inner_us_per_launch = time per block.  "
                             "The two 134 MB tableau buffers do not fit the Infinity Cache: the update streams at the HBM rate (~51 us per block) and would set the block "
                             "time with blocks of 8; with 16 the pivot chain does (block_GBs = update + chain bytes per block / block time)"}
        p4.free()
        cx4.close()
        del c4, A4, b4

    # ---- a standard form WITHOUT a slack basis (equality rows): findLinearlyIndependent (simplex.go:611-637) decides the start
    if args.general:
        rng = np.random.default_rng(5)
        mg, ng = 500, 700
        x0 = np.abs(rng.standard_normal(ng))
        Ae = rng.standard_normal((mg, ng)); Ge = rng.standard_normal((mg, ng))
        ce = np.abs(rng.standard_normal(ng))
        # [Ae 0; Ge I] [x; s] = [Ae x0; Ge x0 + slack]: 1000 rows, 1200 columns, feasible and bounded by construction
        A0 = np.zeros((2 * mg, ng + mg)); A0[:mg, :ng] = Ae; A0[mg:, :ng] = Ge; A0[mg:, ng:] = np.eye(mg)
        b0 = np.concatenate([Ae @ x0, Ge @ x0 + np.abs(rng.standard_normal(mg))])
        c0 = np.concatenate([ce, np.zeros(mg)])
        gres = {}
        for devsearch, blocked in ((1, 1), (1, 0), (0, 0)):
            cxg = lp.Context(device=local_rank, general_device=devsearch, general_block=blocked)
            pg = cxg.upload(c0, A0, b0)
            pg.solve(0.0)
            tg = 1e9
            for _ in range(3):
                t1 = time.perf_counter(); rg = pg.solve(0.0); tg = min(tg, time.perf_counter() - t1)
            gres[(devsearch, blocked)] = (tg, rg)
            cxg.close()
        (tdev, rdev), (tone, rone), (thost, rhost) = gres[(1, 1)], gres[(1, 0)], gres[(0, 0)]
        out["general_basis"] = {"workload": "%dx%d standard form with %d equality rows and no slack basis (seed 5), one full solve" % (2 * mg, ng + mg, mg),
                                "status": int(rdev.status), "pivots": int(rdev.stats["pivots_phase1"] + rdev.stats["pivots_phase2"]),
                                "seconds": tdev, "seconds_pivot_loops": rdev.stats["seconds_pivot_loop"], "seconds_final_solve": rdev.stats["seconds_final_solve"],
                                "seconds_with_one_candidate_per_launch_set": tone, "seconds_with_host_search": thost,
                                "same_result_as_host_search": bool(rdev.status == rhost.status and rdev.z == rhost.z and np.array_equal(rdev.x, rhost.x)
                                                                   and rone.status == rhost.status and rone.z == rhost.z and np.array_equal(rone.x, rhost.x)),
                                "note": "initial-basis search on the device (general_block.hip: explicit Q^T, 16 candidate columns per four launches, the two products on the matrix cores); "
                                        "the reference form of the search is O(m^4)"}

    # ---- BASELINE config 5 on one GPU (the figure the N > 1 lines scale from)
    if args.frontier_vars > 0:
        fout, froof, fcpu = frontier_leg(12, 3)
        fout["roofline"] = froof
        if fcpu is not None:
            fout["cpu_baseline"] = fcpu
        out["frontier"] = fout
        if args.frontier_wide_vars > args.frontier_vars:
            # a frontier wider than one GPU's 256 CUs (2^11 = 2048 children of the same root): where sharding over GPUs can pay
            wout, wroof, _ = frontier_leg(3, 2, nvars=args.frontier_wide_vars, light=True)
            out["frontier_wide"] = {k: wout[k] for k in ("workload", "relaxations_per_s", "wave_seconds", "waves_timed", "pivots_per_wave", "feasible_children",
                                                       "host_fallbacks_per_wave", "device_batched_per_wave", "schedule")}
        if args.frontier_xwide_vars > args.frontier_wide_vars > args.frontier_vars:
            # 8192 children: the 1-GPU figure the N > 1 `frontier_xwide` lines scale from (the width at which >= 6x is reachable)
            xout, _, _ = frontier_leg(2, 1, nvars=args.frontier_xwide_vars, light=True)
            out["frontier_xwide"] = {k: xout[k] for k in ("workload", "relaxations_per_s", "wave_seconds", "waves_timed", "pivots_per_wave", "feasible_children",
                                                         "host_fallbacks_per_wave", "device_batched_per_wave")}

    # ---- BASELINE config 3: host branch-and-bound (tree.go semantics, gomilp_amd/bnb.py) driving GPU relaxations
    if args.milp_nodes > 0:
        from gomilp_amd import bnb
        m3, seed3 = synth.CONFIGS["C3"]
        c3, G3, h3 = synth.dense_lp_inequality_form(m3, seed3)
        int3 = [j % 4 == 0 for j in range(m3)]
        pool3 = lp.FrontierPool(device=local_rank, workers=args.workers)   # one pool for the process, like a Go host would keep
        bnb.solve_milp(c3, None, None, G3, h3, int3, max_nodes=15, pool=pool3)  # warm-up
        tm0 = time.perf_counter()
        mres = bnb.solve_milp(c3, None, None, G3, h3, int3, max_nodes=args.milp_nodes, pool=pool3)   # root upload + root solve + 33 waves
        tm = time.perf_counter() - tm0
        pool3.close()
        out["milp_c3"] = {"workload": "C3: random MILP %dx%d (seed %d), 25%% integer vars, FIFO B&B, node budget %d" % (m3, 2 * m3, seed3, args.milp_nodes),
                          "relaxations": mres.relaxations, "waves": mres.waves, "pivots": mres.pivots, "seconds": tm,
                          "relaxations_per_s": mres.relaxations / tm, "result": mres.error or "optimal",
                          "incumbent_z": None if mres.x is None else mres.z}
        # opt-in warm start (gomilp_frontier_solve_warm): the same tree with every child started from its parent's kept basis
        pool3w = lp.FrontierPool(device=local_rank, workers=args.workers)
        bnb.solve_milp(c3, None, None, G3, h3, int3, max_nodes=15, pool=pool3w, warm=True)  # warm-up
        tw0 = time.perf_counter()
        wres = bnb.solve_milp(c3, None, None, G3, h3, int3, max_nodes=args.milp_nodes, pool=pool3w, warm=True)
        tw = time.perf_counter() - tw0
        pool3w.close()
        cs = [nd for nd in mres.nodes if nd.status != -1]
        ws = [nd for nd in wres.nodes if nd.status != -1]
        same_dec = len(cs) == len(ws) and all(a.status == b_.status and a.decision == b_.decision and list(a.constraints) == list(b_.constraints) for a, b_ in zip(cs, ws))
        zdiff = max([abs(a.z - b_.z) / max(1.0, abs(a.z)) for a, b_ in zip(cs, ws) if a.status == 0 and b_.status == 0] or [0.0])
        out["warm_start"] = {"c3": {"cold_relaxations_per_s": mres.relaxations / tm, "warm_relaxations_per_s": wres.relaxations / tw,
                                    "cold_pivots_per_node": mres.pivots / max(mres.relaxations, 1), "warm_pivots_per_node": wres.pivots / max(wres.relaxations, 1),
                                    "dual_pivots": int(wres.pivots_dual), "warm_started": int(wres.warm_started), "handed_back_to_cold": int(wres.warm_fallbacks),
                                    "nodes": len(ws), "identical_status_and_decisions": bool(same_dec), "max_relative_z_difference": zdiff},
                             "c5": {"note": "a relaxation starts warm only from a kept parent it extends by ONE branch row; the 256 children of the C5 wave extend the root by 8 "
                                            "rows each and start cold by rule (the frontier figures above are theirs): the worst case of the mode is cold + the dual-pivot budget"},
                             "note": "opt-in (gomilp_frontier_solve_warm): parity on status / decision / z <= 1e-9, not on the pivot path"}
        def _cpu_c3():
            from oracle import oracle as O   # the checker, timed as the CPU baseline (never the product path)
            O.set_threads(max(1, min(args.cpu_threads, os.cpu_count() or 1)))
            tc0 = time.perf_counter()
            ores = O.solve_milp(c3, None, None, G3, h3, int3, max_nodes=args.milp_cpu_nodes,
                                simplex_fn=lambda cc, AA, bb: O.simplex(cc, AA, bb, 0.0, None, fast_initial_basis=True))
            tc = time.perf_counter() - tc0
            osolved = [nd for nd in ores.nodes if nd.status != -1]
            gsolved = [nd for nd in mres.nodes if nd.status != -1][: len(osolved)]
            same = all(g.status == o.status and g.decision == o.decision and (o.status != 0 or (g.z == o.z and np.array_equal(g.x[: len(o.x)], o.x)))
                       for g, o in zip(gsolved, osolved))
            out["milp_c3"]["cpu_baseline"] = {"value": len(osolved) / tc, "unit": "relaxations/s", "cores": max(1, min(args.cpu_threads, os.cpu_count() or 1)), "kind": "port",
                                              "sample": "root + the first %d nodes of the same tree in the reference's FIFO order, one oracle solve after the other "
                                                        "(LU panels threaded over the host cores), %.1f s wall" % (len(osolved) - 1, tc),
                                              "gpu_nodes_identical_on_sample": bool(same)}
        if not args.no_cpu_baseline and args.milp_cpu_nodes > 0:
            deferred.append(_cpu_c3)
    # ---- degenerate trees: the 240 integer-data MILPs of the parity suite (duplicate rows, stacked branch rows: every node sits on a
    # degenerate vertex, decided on fresh gonum-order solves — Engine::exact_step): what the exact steps cost
    if args.milp_nodes > 0:
        from gomilp_amd import bnb
        poold = lp.FrontierPool(device=local_rank, workers=args.workers)
        fam = [synth.degenerate_integer_milp(sd) for sd in range(240)]
        for cd, Gd, hd, intd in fam[:8]:
            bnb.solve_milp(cd, None, None, Gd, hd, intd, max_nodes=15, pool=poold)   # warm-up
        td0 = time.perf_counter()
        nrel = 0
        for cd, Gd, hd, intd in fam:
            rd = bnb.solve_milp(cd, None, None, Gd, hd, intd, max_nodes=15, pool=poold)
            nrel += rd.relaxations
        td = time.perf_counter() - td0
        poold.close()
        out["degenerate_trees"] = {"workload": "240 integer-data MILPs (2-5 rows + branch rows, duplicate rows; gomilp_amd/synth.py degenerate_integer_milp), FIFO B&B, 15 nodes each",
                                   "trees": len(fam), "relaxations": int(nrel), "seconds": td, "relaxations_per_s": nrel / td,
                                   "note": "bases of up to 64 rows: every relaxation on a worker's single-relaxation engine with the pivot-by-pivot condition replay and the "
                                           "exact-degenerate steps (two gonum-order device LUs per degenerate pivot: ab^T, and ab once for x_B / the entering column / the Bland candidates); tests/test_gpu_golden.py checks every node of these trees "
                                           "bit for bit against the oracle tree"}
    for fn in deferred:   # the CPU baselines (oracle), behind every GPU leg
        fn()
    emit(out)
    prob.free()
    ctx.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
