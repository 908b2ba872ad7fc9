#!/usr/bin/env python3
"""bench.py — env-steps/sec of the batched WindGym step() on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

One "step" = one batched step() over all envs of a rank (actions already resident in HBM).  Rank 0 prints ONE JSON
line: `value` = whole-job env-steps/s, `roofline` for the dominant kernel (k_flow), `cpu_baseline` (the oracle's C
port timed on the host cores; rank 0, N = 1 only).

Launch: for N > 1 the driver starts this file under `python -m torch.distributed.run --nproc-per-node N`; when it is
started directly with --gpus N > 1 it re-executes itself that way.  A world size that differs from --gpus, or fewer
visible GPUs than ranks, is an error — the line never reports a GPU count it did not run on.

Timed region: after `--preroll` untimed steps (the batch leaves its synchronised start: episodes truncate and reset at
their own times, background episodes are in their steady-state schedule) and W warm-up steps, EXACTLY K steps are timed
between barrier + synchronize; that is repeated `--reps` times and the MEDIAN repetition is reported (with its
inter-quartile range `ms_per_step_iqr`; the repetitions are listed in `ms_per_step_reps`), so that one descheduled host
thread does not decide the number.  When K is small the repetition count is raised until the timed regions add up to
`--min-timed-seconds` (0.25 s): repetitions are free, a 1.5 ms sample is not a measurement.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# host threads must not spin: idle OpenMP workers (torch's pool, the oracle's) busy-waiting inside a CPU-quota'd
# container starve the one thread that launches kernels
os.environ.setdefault("OMP_WAIT_POLICY", "passive")
os.environ.setdefault("GOMP_SPINCOUNT", "0")
os.environ.setdefault("KMP_BLOCKTIME", "0")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


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


def host_cores():
    """Cores this process may really use: the affinity mask, capped by the cgroup CPU quota."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, q // per))
            break
        except Exception:
            continue
    return max(1, n)


def _cpu_rate(orc, acts, seconds, max_steps):
    """env-steps/s of the oracle over >= `seconds` of wall time (3 untimed steps first)."""
    n_act = len(acts)
    for i in range(3):
        orc.step(acts[i % n_act])
    t0 = time.perf_counter()
    steps = 0
    while True:
        orc.step(acts[steps % n_act])
        steps += 1
        el = time.perf_counter() - t0
        if el >= seconds or (max_steps and steps >= max_steps):
            break
    return orc.B * steps / el, steps, el


def cpu_baseline(args):
    """The oracle's C port timed on the host cores of this box, in this run (SURVEY.md §8d "CPU reference timing"):
      * the SAME workload as the GPU line, envs spread over all usable cores with OpenMP (one thread per core, passive
        waits), fp64 (`value`) and fp32 (`f32_value`);
      * cfg1 (BASELINE.json configs[0]: 2 turbines, 2turb.yaml sensors, B = 1) on ONE core, fp64 and fp32 — the analogue
        of one reference env process.
    Each leg runs for a fixed wall time (no step cap), so the sample is seconds of CPU work, and the env count is large
    enough (>= 8 per thread) that the synchronous resets of single envs average out."""
    import numpy as np
    from oracle import oracle as om
    om.build()
    cores = min(host_cores(), 64)
    n = args.cpu_envs if args.cpu_envs else max(128, 8 * cores)
    rng = np.random.default_rng(0)
    legs = {}
    for prec in ("f64", "f32"):
        cfg = make_cfg(n, autoreset=True, farms2=not args.one_farm, workload=args.workload)
        orc = om.Oracle(cfg, prec)
        orc.set_threads(cores)
        if args.workload == "cfg5":
            from windgym_amd.mann import generate_mann_box
            orc.set_turbulence_box(generate_mann_box((512, 128, 32), (3.0, 3.0, 3.0), seed=1234), (3.0, 3.0, 3.0))
        orc.reset(seeds=1234 + np.arange(n))
        acts = rng.uniform(-1, 1, size=(8, n, cfg.n_turb)).astype(np.float32)
        legs[prec] = _cpu_rate(orc, acts, args.cpu_seconds if prec == "f64" else 0.6 * args.cpu_seconds,
                               args.cpu_max_steps)
        orc.close()
    # cfg1: the reference's own CPU-runnable case, one env on one core
    from windgym_amd import presets
    from windgym_amd.config import EnvConfig
    from windgym_amd.turbine import V80
    cfg1 = {}
    for prec in ("f64", "f32"):
        c1 = EnvConfig(turbine=V80(), yaml_dict=presets.two_turb_config(), turbtype="None", n_envs=1, autoreset=True,
                       n_passthrough=5, n_rotor_pts=16)
        orc = om.Oracle(c1, prec)
        orc.set_threads(1)
        orc.reset(seeds=[1234])
        acts = rng.uniform(-1, 1, size=(64, 1, c1.n_turb)).astype(np.float32)
        cfg1[prec] = _cpu_rate(orc, acts, 0.25 * args.cpu_seconds, 0)
        orc.close()
    v64, s64, e64 = legs["f64"]
    v32, s32, e32 = legs["f32"]
    return {"value": v64, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "f32_value": v32,
            "cfg1_f64": cfg1["f64"][0], "cfg1_f32": cfg1["f32"][0], "cfg1_cores": 1,
            "seconds": {"f64": e64, "f32": e32, "cfg1_f64": cfg1["f64"][2], "cfg1_f32": cfg1["f32"][2]},
            "sample": f"{n} envs x {s64} steps ({e64:.1f} s, fp64) and x {s32} steps ({e32:.1f} s, fp32) of the same "
                      f"{args.workload} workload: oracle C port, OpenMP over envs on {cores} threads = usable cores "
                      f"(affinity mask capped by the cgroup quota), passive waits, after reset, synchronous per-env "
                      f"resets inside the sample; cfg1 = 2 turbines / 2turb.yaml sensors (O=200) / B=1 on one core, "
                      f"{cfg1['f64'][1]} + {cfg1['f32'][1]} steps, incl. one ctypes call per step"}


def api_legs(args, dev, abi_env, acts):
    """Throughput through the API the reference's users call, beside the bare C-ABI rate, on the same
    workload and batch: `WindFarmVecEnv.step` on CUDA tensors with the lazy info dict (examples/longer_steps_example.py:194-219
    drives a VecEnv), the same under `RecordEpisodeVals`, which reads infos["Power agent"] on the host every step
    (wrappers/recordEpisodeVals.py:31-64), and stable-baselines3's VecEnv protocol with numpy in / out (`as_sb3()`).
    Per leg: env-steps/s over `--api-steps` steps between two device synchronisations (what a training loop sees), and the host
    time of one step() call — the mean duration of the first 200 calls after a synchronisation, while the launch queue is
    far from full, i.e. the Python + ctypes + launch cost the GPU work hides behind (or not)."""
    import numpy as np
    import torch
    from windgym_amd import presets
    from windgym_amd.envs import RecordEpisodeVals, WindFarmVecEnv
    from windgym_amd.turbine import V80
    B, n_act = abi_env.B, len(acts)
    steps, warm = args.api_steps, 50

    def measure(step_fn, sync):
        for i in range(warm):
            step_fn(i)
        sync()
        host = 0.0
        for i in range(200):
            t0 = time.perf_counter()
            step_fn(i)
            host += time.perf_counter() - t0
        sync()
        t0 = time.perf_counter()
        for i in range(steps):
            step_fn(i)
        sync()
        el = time.perf_counter() - t0
        return {"value": B * steps / el, "unit": "env-steps/s", "ms_per_step": el / steps * 1e3, "host_us_per_call": host / 200 * 1e6}

    sync = lambda: torch.cuda.synchronize(dev)          # noqa: E731
    out = {"steps": steps, "envs": B}
    out["abi"] = measure(lambda i: abi_env.step(acts[i % n_act]), sync)
    kw = dict(turbtype="None", n_passthrough=5, n_rotor_pts=16)
    d = presets.bench_cfg2_config()
    if args.one_farm:
        d["power_def"]["Power_reward"] = "Power_avg"
    venv = WindFarmVecEnv(V80(), B, yaml_dict=d, seed=1234, device=dev.index, as_torch=True, **kw)
    venv.reset(seed=1234)
    for i in range(300):          # (out of the synchronised start, like the headline's pre-roll — shorter: these legs are ratios)
        venv.step(acts[i % n_act])
    out["vecenv_torch"] = measure(lambda i: venv.step(acts[i % n_act]), sync)
    rec = RecordEpisodeVals(venv)
    out["vecenv_torch_record_episode_vals"] = measure(lambda i: rec.step(acts[i % n_act]), sync)
    sb3 = venv.as_sb3()
    acts_np = [a.cpu().numpy() for a in acts]
    r = measure(lambda i: sb3.step(acts_np[i % n_act]), sync)
    o_dim = venv.batch.obs_dim
    r["h2d_bytes_per_step"] = B * venv.n_turb * 4
    r["d2h_bytes_per_step"] = B * (o_dim * 4 + 4 + 1 + 4)
    r["note"] = ("numpy actions in (one H2D copy), observations / rewards / dones / farm powers out through pinned buffers with ONE "
                 "stream synchronisation per step + a host copy (SB3 keeps the previous observations while it steps); final "
                 "observations cross only on steps with a truncation; infos = a persistent list of lazy per-env dicts")
    out["sb3_numpy"] = r
    venv.close()
    for k in ("vecenv_torch", "vecenv_torch_record_episode_vals", "sb3_numpy"):
        out[k]["frac_of_abi"] = out[k]["value"] / out["abi"]["value"]
    return out


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--timing-period", type=int, default=16, help="HIP events bracket the kernels of every n-th step() "
                    "of the timed region (roofline.kernel_ms)")
    ap.add_argument("--reps", type=int, default=5, help="repetitions of the K-step timed region; the median is reported")
    ap.add_argument("--min-timed-seconds", type=float, default=0.25,
                    help="raise the repetition count until the timed regions add up to this much (0 = exactly --reps)")
    ap.add_argument("--preroll", type=int, default=None,
                    help="untimed steps before the warm-up that take the batch out of its synchronised start "
                         "(default: 600 for cfg2/cfg4/cfg5, 100 for cfg3)")
    ap.add_argument("--envs", type=int, default=None, help="envs per GPU (weak scaling); default per workload")
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS),
                    help="cfg2 is the headline metric; the others are BASELINE.json's remaining GPU configs")
    ap.add_argument("--one-farm", action="store_true", help="F=1 (no baseline farm)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--graph", type=int, default=None, help="1: step() as one hipGraphLaunch, 0: direct launches "
                                                             "(default: the library's default)")
    ap.add_argument("--full-chains", action="store_true", help="no chain pruning: advect all P slots of every chain")
    ap.add_argument("--no-autoreset", action="store_true",
                    help="diagnostic only: no background episodes (the run must stay shorter than the shortest episode)")
    ap.add_argument("--cpu-envs", type=int, default=0, help="envs of the CPU sample (default: max(128, 8 x cores))")
    ap.add_argument("--cpu-seconds", type=float, default=10.0,
                    help="wall time of the fp64 leg of the CPU sample (fp32: 0.6 x, cfg1 legs: 0.25 x each)")
    ap.add_argument("--cpu-max-steps", type=int, default=0, help="optional step cap of the CPU legs (0 = none)")
    ap.add_argument("--api", default=None, choices=("all", "none"),
                    help="also time the facades the reference's users call (WindFarmVecEnv, RecordEpisodeVals, SB3 VecEnv) beside the "
                         "C ABI and report them under \"api\" (extra keys; the headline is unchanged).  Default: all for the single-GPU "
                         "cfg2 run unless --no-cpu asks for a quick line")
    ap.add_argument("--api-steps", type=int, default=2000, help="timed steps of each --api leg")
    ap.add_argument("--scaling", default="weak", choices=("weak", "strong"),
                    help="weak: --envs (default: the workload's) per GPU; strong: that many envs in TOTAL, split over "
                         "the N ranks (BASELINE.json's '4096 envs sharded across 8')")
    return ap.parse_args()


def main():
    args = parse()
    # stdout carries exactly ONE line (the JSON): everything a library prints there (RCCL's version banner, ...) is
    # sent to stderr instead; the line itself is written to the saved descriptor at the very end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if args.gpus < 1 or args.steps < 1 or args.reps < 1:
        sys.exit("bench.py: --gpus, --steps and --reps must be >= 1")
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    import torch
    n_vis = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if n_vis < args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} needs {args.gpus} visible GPUs, found {n_vis}")
    if args.gpus > 1 and not launched:
        # started directly: become the N-rank job the flag asks for (one process per GPU over RCCL)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd, stdout=json_fd))

    import numpy as np
    import torch.distributed as dist
    torch.set_num_threads(1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a GPU count that did not run")

    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # The RCCL group exists for every N — also N = 1, so that the single-GPU run exercises exactly the collective
    # the 8-GPU run uses (ShardedMetrics.all_reduce over "nccl").
    if not launched:
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(free_port())
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    rccl = True
    try:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    except Exception as ex:          # pragma: no cover — only tolerated for a single rank
        if world > 1:
            raise
        print(f"bench.py: RCCL process group unavailable for the single rank ({ex}); metrics stay local", file=sys.stderr)
        rccl = False

    from windgym_amd import binding
    from windgym_amd.parallel import ShardedMetrics
    from windgym_amd.parallel import shard_range
    B_flag = args.envs if args.envs else WORKLOADS[args.workload][0]
    if args.scaling == "strong":
        # a fixed global batch, rank g owns the contiguous shard [lo, hi) of the env axis
        env_lo, env_hi = shard_range(B_flag, rank, world)
        if env_hi <= env_lo:
            sys.exit(f"bench.py: --scaling strong: {B_flag} envs cannot be split over {world} ranks")
    else:
        env_lo, env_hi = rank * B_flag, (rank + 1) * B_flag
    B = env_hi - env_lo
    cfg = make_cfg(B, autoreset=not args.no_autoreset, farms2=not args.one_farm, workload=args.workload)
    cfg.advect_full_chains = bool(args.full_chains)
    env = binding.HipBatch(cfg, device=dev.index)
    if args.graph is not None:
        env.set_step_graph(bool(args.graph))
    if args.workload == "cfg5":
        from windgym_amd.mann import generate_mann_box_torch, reference_box_spec
        spec = reference_box_spec("MannFixed", cfg.D)
        env.set_turbulence_box(generate_mann_box_torch(device=dev, **spec), spec["dxyz"])
    # env i of the global batch is seeded 1234 + i regardless of the number of GPUs
    seeds = 1234 + env_lo + np.arange(B)
    env.reset(seeds=seeds)
    gen = torch.Generator(device="cpu").manual_seed(0 + rank)
    n_act = 16
    actions = (torch.rand((n_act, B, cfg.n_turb), generator=gen) * 2 - 1).to(dev).contiguous()
    acts = [actions[i] for i in range(n_act)]
    metrics = ShardedMetrics(env)

    def barrier():
        torch.cuda.synchronize()
        if rccl and world > 1:       # (one rank: there is nobody to wait for — the RCCL barrier would only add its own kernel
            dist.barrier()           # and host sync, ~70 us, to every timed region: 6 % of a 20-step region)
            torch.cuda.synchronize()

    # cfg4: the PettingZoo facade needs the per-agent observations [B, N, o_t + o_f] every step — written by the
    # step's own glue kernel into a registered buffer (no separate packing launch)
    multi_out = env.fuse_obs_multi() if args.workload == "cfg4" else None

    step = env.step
    it = 0
    preroll = args.preroll if args.preroll is not None else (0 if args.no_autoreset else
                                                             (100 if args.workload == "cfg3" else 600))
    for _ in range(preroll):
        step(acts[it % n_act]); it += 1
    env.check()
    metrics.all_reduce()                 # RCCL warm-up of the one collective + reset of the episode-metric sums
    env.kernel_timing(4)                 # creates the event pool (outside every timed region)
    env.kernel_timing(0)
    rep_s = []
    flow_ms = glue_ms = 0.0
    n_launch = 0
    # two events around each repetition's whole run of step() calls, on the stream the kernels are launched on: with ONE launch
    # per step, (interval / steps) is the step kernel's duration plus the gap to the next launch — an upper bound that has no
    # per-launch event packets in it (the events around single launches add ~5 us to a 50 us kernel)
    ev_a, ev_b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    region_ms = 0.0
    region_steps = 0
    flow_steps = particles = added = 0.0
    for _ in range(args.warmup):
        step(acts[it % n_act]); it += 1
    env.check()
    n_reps, rep = args.reps, 0
    while rep < n_reps:
        env.kernel_timing(args.timing_period)   # HIP events around every n-th step() of the timed region
        barrier()
        t0 = time.perf_counter()
        ev_a.record()
        for _ in range(args.steps):
            step(acts[it % n_act]); it += 1
        ev_b.record()
        barrier()
        el = time.perf_counter() - t0
        region_ms += ev_a.elapsed_time(ev_b); region_steps += args.steps
        f, g, n, fs, pt = env.kernel_timing(0)
        t = torch.tensor([el], dtype=torch.float64, device=dev)
        if rccl:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)        # slowest rank
        rep_s.append(float(t.item()))
        flow_ms += f * n; glue_ms += g * n; n_launch += n
        flow_steps += fs; particles += pt; added += env.added_lookups()
        rep += 1
        if rep == 1 and args.min_timed_seconds > 0:
            # (decided from the slowest rank's first repetition, so every rank runs the same count)
            n_reps = max(args.reps, min(2000, int(args.min_timed_seconds / max(rep_s[0], 1e-6)) + 1))
    env.check()
    if n_launch:
        flow_ms /= n_launch; glue_ms /= n_launch
    flow_steps /= n_reps; particles /= n_reps; added /= n_reps
    if multi_out is not None:          # the fused buffer holds what an explicit wg_obs_multi returns (to summation order)
        assert (multi_out - env.obs_multi()).abs().max().item() <= 2e-6
    m = metrics.all_reduce()            # the only collective on the path: 8 floats

    rs = sorted(rep_s)
    el_med = rs[len(rs) // 2]
    el_iqr = rs[(3 * len(rs)) // 4] - rs[len(rs) // 4]
    total_envs = B_flag if args.scaling == "strong" else B * world
    value = total_envs * args.steps / el_med

    if rank == 0:
        F = cfg.to_c().n_farms
        # algorithmic bytes of one k_flow launch, from two counts made on the device: the wake
        # particles the advection passes streamed (chain pruning: a particle behind the last turbine is not touched)
        # and the farm flow-steps executed (live farms + background development of the next episodes).
        # per particle: py read+write (8) + packed record ct|k, u_e|hv read (8); per turbine: state r/w + positions;
        # box: + pz,vlp,wlp r/w (24) + 8 corners x (v, w) of the meandering box per particle (64), 8 corners x
        # (u, v, w) of the fine box per rotor point (96); wake-added turbulence (row a7): 8 corners x (u, v, w) of the
        # isotropic box at the rotor points of every target with a candidate source wake (96 each), counted on the device
        per_particle = 16.0 + (24.0 + 64.0 if args.workload == "cfg5" else 0.0)
        per_farm_step = cfg.n_turb * 72.0 + (cfg.n_turb * cfg.n_rotor_pts * 96.0 if args.workload == "cfg5" else 0.0)
        alg_bytes_flow = particles * per_particle + flow_steps * per_farm_step + added * 96.0
        bytes_per_flow_step = alg_bytes_flow / flow_steps if flow_steps > 0 else 0.0
        achieved = alg_bytes_flow / (flow_ms * 1e-3) / 1e9 if flow_ms > 0 else 0.0
        # HBM bytes per k_flow launch: NOT measured in this run — read from the committed summary of the rocprofv3 PMC
        # passes of this same command (separate runs, as the counters require: tools/profile_kflow.sh ->
        # profiles/r03_<cfgN>_kflow_traffic.json); `traffic_source` names the file.  Only quoted for the profiled
        # workloads at their profiled size (the workload's default env count, baseline farm on, one GPU).
        traffic = traffic_source = None
        for rnd in ("r06", "r05", "r04", "r03", "r02"):
            tf = os.path.join(ROOT, "profiles", f"{rnd}_kflow_traffic.json" if args.workload == "cfg2" and rnd == "r02"
                              else f"{rnd}_{args.workload}_kflow_traffic.json")
            if os.path.exists(tf) and args.envs is None and F == 2 and world == 1:
                try:
                    traffic = json.load(open(tf))["hbm_bytes_per_launch"]
                    traffic_source = (f"{os.path.relpath(tf, ROOT)}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes of "
                                      f"this command, FETCH_SIZE x 2 (gfx950); not collected in this run")
                    if rnd != "r06":
                        traffic_source = (f"WARNING: no profile of the current round's kernels for {args.workload} — STALE figure from "
                                          + traffic_source)
                    break
                except Exception:
                    traffic = traffic_source = None
        variant = env.flow_variant()[2]
        fused = variant == 2 and glue_ms == 0.0 and flow_ms > 0
        per_launch_event_ms = flow_ms
        if fused and region_steps > 0:
            # one launch per step: the kernel cannot last longer than the stream's time per step
            flow_ms = min(flow_ms, region_ms / region_steps)
        if fused:
            # step() is ONE launch (k_flow_env with the env's glue as its tail): the timed kernel carries the glue's bytes too
            # (20 N + 12 O + 20 per env step)
            alg_bytes_flow += B * (20.0 * cfg.n_turb + 12.0 * env.obs_dim + 20.0)
            achieved = alg_bytes_flow / (flow_ms * 1e-3) / 1e9
        # bytes of the particles the advection passes actually touched (k_flow_env counts them on the device; the `needed`
        # particles of the algorithmic figure include the chains that rest and are not read at all)
        touched = added if (variant == 2 and args.workload != "cfg5") else None
        if touched is not None:
            added = 0.0
            alg_bytes_flow = particles * per_particle + flow_steps * per_farm_step + (B * (20.0 * cfg.n_turb + 12.0 * env.obs_dim + 20.0) if fused else 0.0)
            achieved = alg_bytes_flow / (flow_ms * 1e-3) / 1e9 if flow_ms > 0 else 0.0
            bytes_per_flow_step = alg_bytes_flow / flow_steps if flow_steps > 0 else 0.0
        touched_bytes = (touched * per_particle + (alg_bytes_flow - particles * per_particle)) if touched is not None else None
        kernel_name = {0: "k_flow", 2: "k_flow_envb" if args.workload == "cfg5" else "k_flow_env"}[variant] + \
            (" (one launch per step: flow + glue tail)" if fused else "")
        out = {
            "metric": "env-steps/sec (whole node), 16-turbine farm x 4096 envs" if args.workload == "cfg2"
                      else f"env-steps/sec (whole node), {args.workload}",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": el_med / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "storage": "state fp32; the frozen emission record of a wake particle (ct, k, u_e / U, hv) is stored as 4 x 16-bit "
                       "fixed point (8 B per particle; eps is recomputed from ct), positions in fp64",
            "config": {"workload": f"{args.workload}: {WORKLOADS[args.workload][1]} "
                                   f"(O={env.obs_dim}), {B} envs/GPU, F={F} farms/env "
                                   f"({'Baseline reward' if F == 2 else 'Power_avg reward'}), P={cfg.n_particles}, "
                                   f"S={cfg.n_rotor_pts}, same-step autoreset on; f32 arithmetic, u16 emission record",
                       "envs_per_gpu": B, "n_turb": cfg.n_turb, "farms_per_env": F,
                       "parallelism": f"env-axis shard x{world}"},
            "reps": n_reps, "preroll": preroll,
            "ms_per_step_iqr": el_iqr / args.steps * 1e3,
            "timed_seconds": sum(rep_s),
            "ms_per_step_reps": [round(r / args.steps * 1e3, 6) for r in (rep_s if len(rep_s) <= 32 else rs[::max(1, len(rs) // 32)])],
            "gpu_ms_per_step": flow_ms + glue_ms,
            "rccl": rccl,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": achieved / 8000.0, "traffic": traffic, "traffic_source": traffic_source,
                         # the same launch on the bytes of the particles it actually touched, and on the HBM bytes the
                         # counters saw (committed profile of this command): wasted re-reads show as frac_traffic > frac
                         "frac_touched": (touched_bytes / (flow_ms * 1e-3) / 1e9 / 8000.0) if (touched_bytes and flow_ms > 0) else None,
                         "frac_traffic": (traffic / (flow_ms * 1e-3) / 1e9 / 8000.0) if (traffic and flow_ms > 0) else None,
                         "particles_touched_per_launch": touched,
                         "kernel": kernel_name,
                         "kernel_ms": flow_ms, "glue_kernel_ms": glue_ms, "launches_timed": n_launch,
                         "kernel_ms_how": ("HIP events around each repetition's run of step() calls / steps (one launch per step: kernel + gap to "
                                           "the next launch); events around single launches measured %.4f ms" % per_launch_event_ms) if fused
                                          else "HIP events around every n-th launch of the timed region",
                         "algorithmic_bytes_per_launch": alg_bytes_flow, "farm_flow_steps_per_launch": flow_steps,
                         "particles_needed_per_launch": particles,
                         "added_turbulence_rotor_points_per_launch": added,
                         "particle_slots_per_launch": flow_steps * cfg.n_turb * cfg.n_particles,
                         "bytes_per_farm_flow_step": bytes_per_flow_step},
            "episode_metrics": {k: float(v) for k, v in m.items()},
        }
        want_api = args.api == "all" or (args.api is None and not args.no_cpu)
        if world == 1 and want_api and args.workload == "cfg2" and args.envs is None and not args.no_autoreset:
            out["api"] = api_legs(args, dev, env, acts)
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(args)
        line = json.dumps(out)
    else:
        line = None
    if rccl:
        dist.destroy_process_group()
    sys.stdout.flush()
    if line is not None:
        os.write(json_fd, (line + "\n").encode())
    os.close(json_fd)


if __name__ == "__main__":
    main()
