# usage (GPU box): WL=cfg2 bash tools/timeline_env_lib.sh <variant built with tools/envvariant.sh <name> -DWG_TIMELINE> [extra bench args]
# -> per-phase cycles of k_flow_env's waves (wave 0 of every env: the main wave of context 0), from a prebuilt variant library
cd $GRAFT_REPO_ROOT
export WG_DEBUG_HOOKS=1 WG_FLOW_ENV=1
V=$1; shift
WG_LIB=$PWD/windgym_amd/variants/lib_$V.so WG_TIMELINE_OUT=gpurun_out/timeline_env.bin python bench.py --workload ${WL:-cfg2} --steps 60 --warmup 10 --reps 1 --no-cpu "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('value', d['value'], 'kflow ms', d['roofline']['kernel_ms'])"
python - <<'PY'
import numpy as np
raw = np.fromfile('gpurun_out/timeline_env.bin', dtype=np.int64).reshape(-1, 32)
ok = (raw[:, 8] > raw[:, 0]) & (raw[:, 0] > 0) & (raw[:, 7] > raw[:, 6]) & (raw[:, 6] > raw[:, 3]) & (raw[:, 4] > raw[:, 1])
a = raw[ok]
print('waves with a consistent single round:', len(a), 'of', len(raw))
seq = [(0, 1, 'prologue: headers + state loads + LDS set-up'), (1, 4, 'roles, clocks published'), (4, 5, 'candidate pass + list offsets'),
       (5, 11, 'first batch of bracket gathers issued'), (11, 2, 'records'), (2, 9, 'evaluation batches + sums'),
       (9, 10, 'quad list (+ direct emission stores)'), (10, 3, 'advection pass'), (3, 6, 'clock advance'),
       (6, 7, 'tail (power, measurement, ring push, farm sums, schedule)'), (7, 8, 'epilogue (state stores, accounting)'),
       (8, 12, 'store drain before the glue'), (12, 13, 'glue (lean_step)'), (12, 28, '  glue: header'), (28, 29, '  glue: loads'),
       (29, 30, '  glue: reward, metrics'), (30, 31, '  glue: observation'), (31, 13, '  glue: plan + write-back')]
fin = np.where(a[:, 13] > 0, a[:, 13], a[:, 8])
tot = fin - a[:, 0]
print('total: mean %.0f median %.0f p10 %.0f p90 %.0f' % (tot.mean(), np.median(tot), np.percentile(tot, 10), np.percentile(tot, 90)))
for i, j, n in seq:
    if (a[:, j] <= 0).all() or (a[:, i] <= 0).all():
        continue
    d = a[:, j] - a[:, i]
    print(f'{n:62s} mean {d.mean():8.0f}  median {np.median(d):8.0f}  p90 {np.percentile(d, 90):8.0f}')
w0, w1 = a[:, 14], a[:, 15]
t0 = w0.min()
print('wall clock (100 MHz ticks -> us): wave starts median %.2f p90 %.2f max %.2f ; wave ends median %.2f p90 %.2f max %.2f ; wave life median %.2f us' % (
    np.median(w0 - t0) / 100, np.percentile(w0 - t0, 90) / 100, (w0 - t0).max() / 100, np.median(w1 - t0) / 100, np.percentile(w1 - t0, 90) / 100, (w1 - t0).max() / 100, np.median(w1 - w0) / 100))
life = (w1 - w0) / 100
for name, sel in (('episode set-up at the head of the launch', a[:, 16] != 0), ('two or more flow rounds', a[:, 17] >= 2), ('first observation built', a[:, 18] != 0),
                  ('swapped (timestep 0 after the glue)', a[:, 19] == 0), ('none of these', (a[:, 16] == 0) & (a[:, 17] < 2) & (a[:, 18] == 0) & (a[:, 19] != 0))):
    if sel.any():
        print('  waves with %-44s n %5d  life us: median %.2f max %.2f' % (name, sel.sum(), np.median(life[sel]), life[sel].max()))
c = a[:, 20:24].sum(0).astype(float)
if c[0] > 0:
    print('candidates per wave: listed %.1f fetched %.1f inside the 5-sigma cut %.1f (inside 3 sigma %.1f)' % tuple(c / len(a)))
if c[0] > 0:
    nc, nq = a[:, 20].astype(float), a[:, 24].astype(float)
    print('quads listed per wave: mean %.1f p90 %.0f max %.0f ; candidates p90 %.0f max %.0f' % (nq.mean(), np.percentile(nq, 90), nq.max(), np.percentile(nc, 90), nc.max()))
    print('correlation of wave life with candidates %.2f, with quads %.2f, with start time %.2f' % (np.corrcoef(life, nc)[0, 1], np.corrcoef(life, nq)[0, 1], np.corrcoef(life, (w0 - t0))[0, 1]))
    A = np.stack([nc, nq, np.ones_like(nc)], 1)
    co, *_ = np.linalg.lstsq(A, life, rcond=None)
    print('least squares: life us = %.4f x candidates + %.4f x quads + %.2f ; residual std %.2f us' % (co[0], co[1], co[2], (life - A @ co).std()))
    slow = life >= np.percentile(life, 99)
    print('slowest 1 %% of the waves: life median %.1f, candidates median %.0f (all: %.0f), quads median %.0f (all: %.0f)' % (np.median(life[slow]), np.median(nc[slow]), np.median(nc), np.median(nq[slow]), np.median(nq)))
    for i, j, n in seq:
        if (a[:, j] <= 0).all() or (a[:, i] <= 0).all():
            continue
        d = (a[:, j] - a[:, i])
        print('   slowest 1 %%: %-52s median %8.0f (all %8.0f)' % (n, np.median(d[slow]), np.median(d)))
ini = a[:, 16] != 0
if ini.any():
    print('episode set-up (cycles; the waves that ran it): wind + yaw draws (lane 0) %s | rated power, rotation, ring layout %s | slots, turbine state %s' % (a[ini, 25].tolist(), a[ini, 26].tolist(), a[ini, 27].tolist()))
order = np.argsort(-life)[:12]
print('slowest waves: life us | candidates quads rounds | set-up first-obs swapped | advection eval glue prologue (cycles)')
for i in order:
    print('   %.1f | %4d %4d %d | %d %d %d | %6d %6d %6d %6d' % (life[i], a[i, 20], a[i, 24], a[i, 17], a[i, 16] != 0, a[i, 18] != 0, a[i, 19] == 0,
          a[i, 3] - a[i, 10], a[i, 9] - a[i, 2], a[i, 13] - a[i, 12], a[i, 1] - a[i, 0]))
print('core clock (cycles per us of wave life): %.0f' % np.median(tot / np.maximum((w1 - w0) / 100, 1e-9)))
PY
