#!/usr/bin/env python3
"""bench.py — env-steps/sec of the batched WindGym step() on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run)

One "step" = one batched step() over all envs of this rank (actions already resident in HBM).  Prints ONE
JSON line on rank 0 with `value` = whole-job env-steps/s, `roofline` for the dominant kernel (k_flow) and
`cpu_baseline` (the oracle's C port timed on the host cores, rank 0, N=1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


WORKLOADS = {
    # name: (default envs per GPU, description)            -- BASELINE.json `configs`
    "cfg2": (4096, "4x4 16-turbine grid (V80, 5.33D pitch), yaw-only action, Env1 sensors, inflow None"),
    "cfg3": (512, "Horns Rev 1 layout (80 turbines), yaw action, Env1 sensors, inflow None"),
    "cfg4": (2048, "3x3 farm, WindFarmEnvMulti per-turbine-agent observations [B,9,o_t+o_f], inflow None"),
    "cfg5": (1024, "4x4 16-turbine grid, frozen Mann box 2048x512x64 @ 3 m + DWM meandering"),
}


def make_cfg(n_envs, autoreset=True, farms2=True, workload="cfg2"):
    from windgym_amd import presets
    from windgym_amd.config import EnvConfig
    from windgym_amd.turbine import V80
    kw = {}
    turbtype = "None"
    if workload == "cfg3":
        d = presets.horns_rev_config()
        kw["x_pos"], kw["y_pos"] = presets.horns_rev1_layout()
    elif workload == "cfg4":
        d = presets.multi_3x3_config()
        kw["extra_timestep_inc"] = True
    else:
        d = presets.bench_cfg2_config()
        if workload == "cfg5":
            turbtype = "MannGenerate"
    if not farms2:
        d["power_def"]["Power_reward"] = "Power_avg"
    return EnvConfig(turbine=V80(), yaml_dict=d, turbtype=turbtype, n_envs=n_envs, autoreset=autoreset,
                     n_passthrough=5, n_rotor_pts=16, **kw)


def cpu_baseline(args):
    """The oracle's C port (OpenMP over envs) on a bounded sample of the same workload."""
    import numpy as np
    from oracle import oracle as om
    om.build()
    n = args.cpu_envs
    cfg = make_cfg(n, autoreset=True, farms2=not args.one_farm, workload=args.workload)
    orc = om.Oracle(cfg, "f32" if args.cpu_f32 else "f64")
    if args.workload == "cfg5":
        from windgym_amd.mann import generate_mann_box
        orc.set_turbulence_box(generate_mann_box((512, 128, 32), (3.0, 3.0, 3.0), seed=1234), (3.0, 3.0, 3.0))
    cores = orc.max_threads()
    orc.reset(seeds=1234 + np.arange(n))
    rng = np.random.default_rng(0)
    acts = rng.uniform(-1, 1, size=(8, n, cfg.n_turb)).astype(np.float32)
    for i in range(3):
        orc.step(acts[i % 8])
    t0 = time.perf_counter()
    steps = 0
    while True:
        orc.step(acts[steps % 8])
        steps += 1
        el = time.perf_counter() - t0
        if el > args.cpu_seconds or steps >= args.cpu_max_steps:
            break
    return {"value": n * steps / el, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"{n} envs x {steps} steps of the same {args.workload} workload (oracle C port, "
                      f"{'fp32' if args.cpu_f32 else 'fp64'}, OpenMP over envs, after reset)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--envs", type=int, default=None, help="envs per GPU (weak scaling); default per workload")
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS),
                    help="cfg2 is the headline metric; the others are BASELINE.json's remaining GPU configs")
    ap.add_argument("--one-farm", action="store_true", help="F=1 (no baseline farm)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--full-chains", action="store_true", help="no chain pruning: advect all P slots of every chain")
    ap.add_argument("--no-autoreset", action="store_true",
                    help="diagnostic only: no background episodes (the run must stay shorter than the shortest episode)")
    ap.add_argument("--cpu-envs", type=int, default=64)
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--cpu-max-steps", type=int, default=400)
    ap.add_argument("--cpu-f32", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # under torch.distributed.run (RANK / MASTER_PORT in the env) the RCCL group is always created — also for one
    # rank, so that the single-GPU box exercises exactly the collectives the 8-GPU run uses
    dist_on = world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    dev = torch.device("cuda", local_rank if dist_on else 0)

    from windgym_amd import binding
    from windgym_amd.parallel import ShardedMetrics
    B = args.envs if args.envs else WORKLOADS[args.workload][0]
    cfg = make_cfg(B, autoreset=not args.no_autoreset, farms2=not args.one_farm, workload=args.workload)
    cfg.advect_full_chains = bool(args.full_chains)
    env = binding.HipBatch(cfg, device=dev.index)
    if args.workload == "cfg5":
        from windgym_amd.mann import generate_mann_box_torch, reference_box_spec
        spec = reference_box_spec("MannFixed", cfg.D)
        env.set_turbulence_box(generate_mann_box_torch(device=dev, **spec), spec["dxyz"])
    # env i of the global batch is seeded 1234 + i regardless of the number of GPUs
    seeds = 1234 + rank * B + np.arange(B)
    env.reset(seeds=seeds)
    gen = torch.Generator(device="cpu").manual_seed(0 + rank)
    n_act = 16
    actions = (torch.rand((n_act, B, cfg.n_turb), generator=gen) * 2 - 1).to(dev).contiguous()
    metrics = ShardedMetrics(env)

    def barrier():
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
            torch.cuda.synchronize()

    # cfg4: the PettingZoo facade needs the per-agent observations [B, N, o_t + o_f] every step — written by the
    # step's own glue kernel into a registered buffer (no separate packing launch)
    multi_out = env.fuse_obs_multi() if args.workload == "cfg4" else None

    def one_step(i):
        env.step(actions[i % n_act])

    for i in range(args.warmup):
        one_step(i)
    env.check()
    env.kernel_timing(4)          # HIP events around every 4th step() of the timed region
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        one_step(i)
    barrier()
    el = time.perf_counter() - t0
    flow_ms, glue_ms, n_launch, flow_steps, particles = env.kernel_timing(False)
    env.check()
    if multi_out is not None:          # the fused buffer holds what an explicit wg_obs_multi returns
        assert torch.equal(multi_out, env.obs_multi())
    m = metrics.all_reduce()            # the only collective on the path: 8 floats

    t = torch.tensor([el], dtype=torch.float64, device=dev)
    if dist_on:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    el_max = float(t.item())
    total_envs = B * world
    value = total_envs * args.steps / el_max

    if rank == 0:
        F = cfg.to_c().n_farms
        # algorithmic bytes of one k_flow launch, from two counts made on the device: the wake
        # particles the advection passes streamed (chain pruning: a particle behind the last turbine is not touched)
        # and the farm flow-steps executed (live farms + background development of the next episodes).
        # per particle: py read+write (8) + packed record ct|k, eps|hv read (8); per turbine: state r/w + positions;
        # box: + pz,vlp,wlp r/w (24) + 8 corners x (v, w) of the meandering box per particle (64), 8 corners x
        # (u, v, w) of the fine box per rotor point (96)
        per_particle = 16.0 + (24.0 + 64.0 if args.workload == "cfg5" else 0.0)
        per_farm_step = cfg.n_turb * 72.0 + (cfg.n_turb * cfg.n_rotor_pts * 96.0 if args.workload == "cfg5" else 0.0)
        alg_bytes_flow = particles * per_particle + flow_steps * per_farm_step
        bytes_per_flow_step = alg_bytes_flow / flow_steps if flow_steps > 0 else 0.0
        achieved = alg_bytes_flow / (flow_ms * 1e-3) / 1e9 if flow_ms > 0 else 0.0
        # HBM bytes per k_flow launch from the rocprofv3 PMC passes of this same command (separate runs:
        # tools/profile_kflow.sh -> profiles/r01_kflow_traffic.json); only quoted for the profiled workload
        traffic = None
        tf = os.path.join(ROOT, "profiles", "r01_kflow_traffic.json")
        if os.path.exists(tf) and B == 4096 and F == 2 and world == 1 and args.workload == "cfg2":
            try:
                traffic = json.load(open(tf))["hbm_bytes_per_launch"]
            except Exception:
                traffic = None
        out = {
            "metric": "env-steps/sec (whole node), 16-turbine farm x 4096 envs" if args.workload == "cfg2"
                      else f"env-steps/sec (whole node), {args.workload}",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": el_max / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {WORKLOADS[args.workload][1]} "
                                   f"(O={env.obs_dim}), {B} envs/GPU, F={F} farms/env "
                                   f"({'Baseline reward' if F == 2 else 'Power_avg reward'}), P={cfg.n_particles}, "
                                   f"S={cfg.n_rotor_pts}, same-step autoreset on",
                       "envs_per_gpu": B, "n_turb": cfg.n_turb, "farms_per_env": F,
                       "parallelism": f"env-axis shard x{world}"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": achieved / 8000.0, "traffic": traffic, "kernel": "k_flow",
                         "kernel_ms": flow_ms, "glue_kernel_ms": glue_ms, "launches_timed": n_launch,
                         "algorithmic_bytes_per_launch": alg_bytes_flow, "farm_flow_steps_per_launch": flow_steps,
                         "particles_streamed_per_launch": particles,
                         "particle_slots_per_launch": flow_steps * cfg.n_turb * cfg.n_particles,
                         "bytes_per_farm_flow_step": bytes_per_flow_step},
            "episode_metrics": {k: float(v) for k, v in m.items()},
        }
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(args)
        print(json.dumps(out))
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
