"""Pretty-print the key fields of bench.py's JSON line (stdin)."""
import json
import sys

for ln in sys.stdin:
    ln = ln.strip()
    if not ln.startswith("{"):
        continue
    d = json.loads(ln)
    r = d["roofline"]
    print(f"{sys.argv[1] if len(sys.argv) > 1 else ''} value {d['value'] / 1e6:.2f} M/s  ms/step {d['ms_per_step']:.4f} "
          f"(reps {[round(x, 4) for x in d.get('ms_per_step_reps', [])]})  gpu_ms {d.get('gpu_ms_per_step', 0):.4f}  "
          f"flow {r['kernel_ms'] * 1e3:.1f} us glue {r['glue_kernel_ms'] * 1e3:.1f} us  frac {r['frac']:.3f} "
          f" farm-steps/launch {r['farm_flow_steps_per_launch']:.0f}  "
          f"episodes {d['episode_metrics']['n_episodes']:.0f}  rccl {d.get('rccl')}  "
          f"cpu {d.get('cpu_baseline', {}).get('value', 0):.0f} on {d.get('cpu_baseline', {}).get('cores', 0)} cores")
