"""Drop-in host classes: the reference's Gymnasium / PettingZoo surface over the batched HIP transition.

* :class:`WindFarmVecEnv`   — the native batched env (thousands of farms per GPU, same-step autoreset);
                              gymnasium ``VectorEnv``-style ``reset/step``; ``as_sb3()`` gives the SB3 ``VecEnv`` protocol.
* :class:`WindFarmEnv`      — single-env facade with the reference's constructor, attributes, ``reset`` /
                              ``step`` return types and info-dict keys (WindGym/Wind_Farm_Env.py:47-1034).
* :class:`FarmEval`         — evaluation subclass (WindGym/FarmEval.py:10-90).
* :class:`WindFarmEnvMulti` — PettingZoo ``ParallelEnv`` facade (WindGym/WindEnvMulti.py:17-249).
* :class:`RecordEpisodeVals`— episode-mean-power statistics (WindGym/wrappers/recordEpisodeVals.py:8-64).

All physics and sensor work happens in libwindgym_hip.so; nothing here computes flow values.
"""
from __future__ import annotations

import copy
from collections import deque
from typing import Optional

import numpy as np

from .binding import HipBatch
from .config import EnvConfig
from .spaces import Box

try:  # pragma: no cover
    import gymnasium as _gym
    _EnvBase = _gym.Env
except Exception:
    _EnvBase = object

try:  # pragma: no cover
    from pettingzoo import ParallelEnv as _ParallelEnvBase
except Exception:
    _ParallelEnvBase = object


def _np(t):
    return t.detach().cpu().numpy()


MAX_BOX_POOL = 16


def _resolve_turbulence(cfg: EnvConfig, turbulence_box=None):
    """Decide, BEFORE the device batch is created, where the frozen turbulence comes from (Wind_Farm_Env.py:197-213,
    :611-659).  Returns None, ("one", box, spacing) or ("pool", [boxes], spacing).

    turbtype "MannLoad": `turbulence_box` (a (box, spacing) pair or a ([boxes], spacing) pool) or the TurbBox file /
    directory of TF_* files — all readable files of equal shape (at most MAX_BOX_POOL) become a device-resident pool and
    every reset draws one like np_random.choice(self.TF_files) (:614).  Without files the reference switches to
    turbtype "MannGenerate" (:207-212), and so does this build: `cfg.turbtype` is changed, so the per-episode draw is
    the seed of :623 again.  Generated boxes (MannFixed / MannGenerate) are made on the GPU when the batch exists."""
    if cfg.turbtype not in ("MannFixed", "MannGenerate", "MannLoad"):
        return None
    if turbulence_box is not None:
        first = turbulence_box[0]
        if isinstance(first, (list, tuple)):
            return ("pool", list(first), turbulence_box[1])
        return ("one", first, turbulence_box[1])
    if cfg.turbtype == "MannLoad":
        from .mann import find_box_files, load_box
        pool, spacing = [], None
        files = find_box_files(getattr(cfg, "TurbBox", None))
        # The reference draws np_random.choice(self.TF_files) over ALL files (:614): a pool of a different size changes
        # every later draw of the episode generator, so a pool that cannot be held as found is an error, not a silent
        # truncation.
        if len(files) > MAX_BOX_POOL:
            raise ValueError(f"TurbBox: {len(files)} turbulence files found, the device pool holds at most {MAX_BOX_POOL}; "
                             f"point TurbBox at a directory with fewer files (the reference draws one of ALL files per reset)")
        for f in files:
            b, sp = load_box(f)
            if pool and (np.shape(b) != np.shape(pool[0]) or tuple(sp) != tuple(spacing)):
                raise ValueError(f"{f}: shape {np.shape(b)} / spacing {tuple(sp)} differs from the first box of the pool "
                                 f"({np.shape(pool[0])} / {tuple(spacing)}); the device pool needs boxes of one shape")
            pool.append(b)
            spacing = sp
        if pool:
            return ("pool", pool, spacing)
        print("Coudnt find the turbulence box file(s), so we switch to generated turbulence")
        cfg.turbtype = "MannGenerate"
    return ("generate",)


def _attach_box(batch: HipBatch, cfg: EnvConfig, source):
    """Put the resolved turbulence source on the device (wg_set_turbulence_box / wg_set_turbulence_boxes); returns the
    source with a generated box filled in, so that a rebuild of the batch reuses it."""
    if source is None:
        return None
    if source[0] == "generate":
        # generated on the device (wg_generate_mann_box).  turbtype "MannGenerate" with cfg.mann_pool = K > 1: K
        # realisations (seeds 1234 ... 1234 + K - 1); every episode's seed (:623) then picks one of them AND an offset into
        # it, instead of an offset into a single box
        from .mann import generate_mann_box_hip, reference_box_spec
        spec = reference_box_spec(cfg.turbtype, cfg.D)
        K = int(getattr(cfg, "mann_pool", 1) or 1) if cfg.turbtype == "MannGenerate" else 1
        if K > MAX_BOX_POOL:
            raise ValueError(f"mann_pool = {K}: the device pool holds at most {MAX_BOX_POOL} boxes "
                             f"(each realisation of the reference's box is ~1 GB on the device)")
        if K > 1:
            base = spec.pop("seed")
            source = ("pool", [generate_mann_box_hip(device=batch.device, seed=base + k, **spec) for k in range(K)], spec["dxyz"])
        else:
            source = ("one", generate_mann_box_hip(device=batch.device, **spec), spec["dxyz"])
    if source[0] == "pool":
        batch.set_turbulence_boxes(source[1], source[2])
    else:
        batch.set_turbulence_box(source[1], source[2])
    return source


class _TurbinesProxy:
    """Duck type of ``fs.windTurbines`` for the call sites listed in SURVEY.md Appendix A."""

    def __init__(self, batch: HipBatch, farm: str, env_index: int = 0):
        self._b, self._farm, self._i = batch, farm, env_index

    @property
    def yaw(self):
        return _np(self._b.info(f"yaw_{self._farm}"))[self._i].astype(np.float64)

    @property
    def rotor_avg_windspeed(self):
        return _np(self._b.info(f"rotor_uvw_{self._farm}"))[self._i].astype(np.float64)

    def power(self):
        return _np(self._b.info(f"power_turb_{self._farm}"))[self._i].astype(np.float64)

    @property
    def positions_xyz(self):
        x = _np(self._b.info("turb_x"))[self._i]
        y = _np(self._b.info("turb_y"))[self._i]
        return np.array([x, y, np.full_like(x, self._b.cfg.tab.hub_height())], dtype=np.float64)

    rotor_positions_xyz = positions_xyz

    def yaw_tilt(self):
        return self.yaw, np.zeros(len(self.yaw))


class _FlowProxy:
    """Duck type of the DYNAMIKS flow-simulation object (`env.fs`, `env.fs_baseline`)."""

    def __init__(self, batch: HipBatch, farm: str, env_index: int = 0):
        self._b, self._i = batch, env_index
        self.windTurbines = _TurbinesProxy(batch, farm, env_index)

    @property
    def time(self):
        return float(_np(self._b.info("fs_time"))[self._i])

    @property
    def wind_direction(self):
        return float(_np(self._b.info("wd_global"))[self._i])

    def get_windspeed(self, view, include_wakes=True, xarray=False):
        """fs.get_windspeed(XYView(z=, x=, y=), include_wakes=) (Wind_Farm_Env.py:1056; AgentEval.py:220-228):
        np.float32 [3, nx, ny]; ``view`` is anything with x, y, z attributes or keys."""
        g = (lambda k: view[k]) if isinstance(view, dict) else (lambda k: getattr(view, k))
        farm = 1 if self.windTurbines._farm == "base" else 0
        return _np(self._b.windspeed(self._i, g("x"), g("y"), z=g("z"), farm=farm, include_wakes=include_wakes))


# ======================================================================================================
def _gym_vector_base():
    """gymnasium's ``VectorEnv`` when gymnasium is installed — isinstance checks of vector wrappers and of training
    libraries then pass (the reference's RecordEpisodeVals extends gymnasium.wrappers.vector.RecordEpisodeStatistics,
    wrappers/recordEpisodeVals.py:8, which requires a VectorEnv) — a plain object otherwise."""
    try:
        from gymnasium.vector import VectorEnv
        return VectorEnv
    except Exception:
        return object


def _gym_autoreset_mode():
    try:
        from gymnasium.vector import AutoresetMode
        return AutoresetMode.SAME_STEP
    except Exception:
        return "SameStep"


class WindFarmVecEnv(_gym_vector_base()):
    """``n_envs`` independent farms on one GPU behind one handle.

    ``step(actions)`` takes a float32 array / CUDA tensor ``[n_envs, n_turb]`` in [-1, 1] and returns
    ``(obs, rewards, terminations, truncations, infos)``; with ``as_torch=True`` these stay CUDA tensors
    (no host synchronisation on the step path).  Truncated envs are reset in the same step from an
    episode that was developed in the background; ``infos["final_obs"]`` holds their last observation.
    """

    def __init__(self, turbine, n_envs: int, yaml_path=None, *, seed: Optional[int] = 0, device: Optional[int] = None,
                 as_torch: bool = False, autoreset: bool = True, turbulence_box=None, sample_site=None, **kwargs):
        self.cfg = EnvConfig(turbine=turbine, yaml_path=yaml_path, n_envs=int(n_envs), autoreset=autoreset,
                             seed=seed, **kwargs)
        source = _resolve_turbulence(self.cfg, turbulence_box)
        self.batch = HipBatch(self.cfg, device=device)
        _attach_box(self.batch, self.cfg, source)
        # site-based wind sampling (Wind_Farm_Env.py:569-594) for the whole batch: a device-resident override table
        # refreshed with torch ops before every step, so every episode initialisation sees an independent draw
        self._site = None
        if sample_site is not None:
            from .site import DeviceSiteSampler
            c = self.cfg
            self._site = DeviceSiteSampler(sample_site, self.batch, (c.wd_min, c.wd_max), (c.ws_min, c.ws_max),
                                           seed=0 if seed is None else seed)
        self.num_envs = self.n_envs = int(n_envs)
        self.n_turb = self.cfg.n_turb
        self.as_torch = as_torch
        self.torch = self.batch.torch
        self.single_observation_space = Box(-1.0, 1.0, (self.batch.obs_dim,), np.float32)
        self.single_action_space = Box(-1.0, 1.0, (self.n_turb,), np.float32)
        self.observation_space = Box(-1.0, 1.0, (self.num_envs, self.batch.obs_dim), np.float32)
        self.action_space = Box(-1.0, 1.0, (self.num_envs, self.n_turb), np.float32)
        self._base_seed = seed
        # gymnasium >= 1.0 vector API: truncated envs are reset in the SAME step (final_obs in the info dict)
        self.metadata = {"autoreset_mode": _gym_autoreset_mode(), "render_modes": []}
        self.render_mode = None
        self.spec = None
        self.closed = False
        self._global_offset = 0          # first global env index of this shard (set by shard())
        self._actions = self.torch.zeros((self.num_envs, self.n_turb), dtype=self.torch.float32,
                                         device=self.batch.device)
        self._term = None

    # -- sharding: env i of the *global* batch is always seeded base_seed + i ------------------------
    def shard(self, rank: int, world: int, n_envs_total: Optional[int] = None):
        from .parallel import shard_range
        total = n_envs_total if n_envs_total is not None else self.num_envs * world
        lo, hi = shard_range(total, rank, world)
        assert hi - lo == self.num_envs, "construct the env with this rank's share of the env axis"
        self._global_offset = lo
        return self

    def _seeds(self, seed):
        if seed is None:
            return None
        if np.isscalar(seed):
            return (int(seed) + self._global_offset + np.arange(self.num_envs)).astype(np.uint64)
        return np.asarray(seed, dtype=np.uint64)

    def _out(self, t):
        return t if self.as_torch else _np(t).copy()

    def reset(self, *, seed=None, options=None, mask=None):
        seeds = self._seeds(seed if seed is not None else (self._base_seed if not getattr(self, "_was_reset", False) else None))
        self._was_reset = True
        obs = self.batch.reset(seeds=seeds, mask=mask)
        return self._out(obs), self.infos()

    def _step_device(self, actions):
        """The step on device tensors: actions uploaded into the persistent buffer if they are not a CUDA tensor already, the
        site-sampling table refreshed, ONE wg_step.  Returns HipBatch.step's views (obs, reward, truncated u8, final_obs)."""
        t = self.torch
        if not isinstance(actions, t.Tensor):
            # (np.array copies: a read-only view — np.broadcast_to in eval_sweep — must not be wrapped as a writable tensor)
            self._actions.copy_(t.from_numpy(np.array(actions, dtype=np.float32, order="C")).reshape(self.num_envs, self.n_turb))
            actions = self._actions
        elif not (actions.is_cuda and actions.dtype == t.float32 and actions.is_contiguous()):
            self._actions.copy_(actions.reshape(self.num_envs, self.n_turb))
            actions = self._actions
        if self._site is not None:
            self._site.refresh()
        return self.batch.step(actions)

    def step(self, actions):
        t = self.torch
        obs, rew, trunc, fin = self._step_device(actions)
        # No torch kernels on the step path (measured: bench.py --api, the facade at >= 90 % of the bare ABI rate): `terminated`
        # is always False (:1029) — one persistent all-False tensor; `truncated` is the ABI's 0 / 1 byte tensor viewed as bool.
        if self._term is None:
            self._term = t.zeros((self.num_envs,), dtype=t.bool, device=self.batch.device)
        term = self._term
        infos = self.infos(step=True)
        trunc_b = self._out(trunc.view(t.bool))
        # gymnasium's vector info convention: a value array plus a "_key" mask of the envs it is valid for.  final_obs stays
        # a dense [B, obs_dim] array (rows of envs that did not truncate are unspecified) instead of gymnasium's per-env
        # object array: no per-env host objects on the step path
        infos["final_obs"] = self._out(fin)
        infos["_final_obs"] = trunc_b
        return self._out(obs), self._out(rew), self._out(term), trunc_b, infos

    def infos(self, step=False):
        """Lazy info dict: values are fetched from the device on first access (keys of _get_info).  After a step(),
        "Power agent" / "Power baseline" are the farm powers of the step just taken — for an env that truncated, the
        terminal step of the episode that ended, not the first state of the swapped-in episode — which is what
        RecordEpisodeVals adds to the finished episode (wrappers/recordEpisodeVals.py:43-46)."""
        return _LazyInfo(self, step=step)

    def metrics(self, reset_after=True):
        from .parallel import ShardedMetrics
        return ShardedMetrics(self.batch).all_reduce(reset_after=reset_after)

    def close(self):
        self.batch.close()
        self.closed = True          # gymnasium.vector.VectorEnv's flag (its own close() sets it after close_extras)

    def as_sb3(self):
        """The same batch behind stable-baselines3's ``VecEnv`` protocol (see :class:`SB3VecEnv`)."""
        return SB3VecEnv(self)

    def seed(self, seed=None):
        self._base_seed = seed
        self._was_reset = False
        return [seed] * self.num_envs


def _sb3_base():
    try:                                           # subclass the real thing when it is installed, so that SB3's
        from stable_baselines3.common.vec_env import VecEnv      # isinstance checks pass and nothing re-wraps the env
        return VecEnv
    except Exception:
        return object


class SB3VecEnv(_sb3_base()):
    """stable-baselines3 ``VecEnv`` over a :class:`WindFarmVecEnv` — what ``make_vec_env(..., n_envs=n)`` +
    ``SubprocVecEnv`` give the reference's training scripts (examples/longer_steps_example.py:194-209), as one GPU handle.

    SB3's protocol, not gymnasium's: ``reset()`` returns the observations only; ``step_wait()`` returns
    ``(obs, rewards, dones, infos)`` with ``infos`` a list of per-env dicts, ``dones = truncations`` (the env never
    terminates, :1029), ``infos[i]["TimeLimit.truncated"]`` and — for an env that was reset in the same step —
    ``infos[i]["terminal_observation"]``.  Arrays are numpy on the host (SB3's buffers are).
    """

    def __init__(self, venv: "WindFarmVecEnv"):
        self.venv = venv
        self.num_envs = venv.num_envs
        self.observation_space = venv.single_observation_space
        self.action_space = venv.single_action_space
        self.render_mode = None
        self._actions = None
        # the step's host arrays (what the per-env info views read) and, on a real batch, pinned staging buffers
        self._power = self._trunc = self._fin = self._fin_dev = None
        self._pin = None
        self._infos = [_SB3Info(self, i) for i in range(self.num_envs)]
        base = type(self).__mro__[1]
        if base is not object:
            base.__init__(self, venv.num_envs, self.observation_space, self.action_space)

    def reset(self):
        obs, _ = self.venv.reset()
        return np.asarray(_np(obs) if self.venv.as_torch else obs)

    def step_async(self, actions):
        self._actions = actions

    def _step_wait_device(self):
        """A real WindFarmVecEnv: drive the batch on device tensors and cross PCIe once per step — observations, rewards,
        truncation flags and farm powers through pinned buffers with ONE stream synchronisation; the final observations only
        on a step in which some env truncated (bench.py --api sb3 states the cost of this path)."""
        v = self.venv
        t = v.torch
        obs, rew, trunc, fin = v._step_device(self._actions)
        power = v.batch.info("step_power_agent")
        if self._pin is None:
            self._pin = tuple(t.empty(x.shape, dtype=x.dtype, pin_memory=True) for x in (obs, rew, trunc, power))
        for h, d in zip(self._pin, (obs, rew, trunc, power)):
            h.copy_(d, non_blocking=True)
        t.cuda.current_stream(v.batch.device).synchronize()
        # (fresh host arrays: SB3 keeps the previous step's observations while it calls step() again)
        o, r, tr, pw = (h.numpy().copy() for h in self._pin)
        self._power, self._trunc = pw, tr.astype(bool)
        self._fin, self._fin_dev = None, (fin if self._trunc.any() else None)
        return o, r, self._trunc, self._infos

    def _fin_row(self, i):
        if self._fin is None:
            self._fin = _np(self._fin_dev)
        return self._fin[i]

    def step_wait(self):
        v = self.venv
        if hasattr(v, "_step_device") and hasattr(v, "batch"):
            return self._step_wait_device()
        obs, rew, term, trunc, infos = v.step(self._actions)
        obs, rew, trunc = (np.asarray(_np(x) if v.as_torch else x) for x in (obs, rew, trunc))
        fin = infos["final_obs"]
        self._fin, self._fin_dev = (_np(fin) if v.as_torch else np.asarray(fin)), None
        power = infos["Power agent"]
        self._power = _np(power) if v.as_torch else np.asarray(power)
        self._trunc = trunc.astype(bool)
        return obs, rew, self._trunc, self._infos

    def step(self, actions):
        self.step_async(actions)
        return self.step_wait()

    def close(self):
        self.venv.close()

    def seed(self, seed=None):
        return self.venv.seed(seed)

    def _indices(self, indices):
        if indices is None:
            return range(self.num_envs)
        return [indices] if isinstance(indices, int) else indices

    def get_attr(self, attr_name, indices=None):
        return [getattr(self.venv, attr_name) for _ in self._indices(indices)]

    def set_attr(self, attr_name, value, indices=None):
        raise NotImplementedError("the farms of a batch share one configuration; construct a new WindFarmVecEnv")

    def env_method(self, method_name, *method_args, indices=None, **method_kwargs):
        raise NotImplementedError("per-env method calls do not exist on a batched GPU env; use the WindFarmVecEnv API")

    def env_is_wrapped(self, wrapper_class, indices=None):
        return [False for _ in self._indices(indices)]

    def get_images(self):
        return [None] * self.num_envs

    def render(self, mode=None):
        return None


class _SB3Info(dict):
    """``infos[i]`` of SB3's list-of-dicts protocol: a dict whose three standard entries — "Power agent",
    "TimeLimit.truncated" and, for an env that truncated in this step, "terminal_observation" — are read from the adapter's
    arrays of the step just taken when they are asked for.  The list and its 4096 dicts are built ONCE: rebuilding them cost
    ~1.5 ms of Python per step against a 55 us GPU step (bench.py --api sb3).  Like SB3's own buffers the views are valid until
    the next ``step()``; ``copy()`` / ``dict(info)`` give a plain snapshot (what VecMonitor and friends do before they add keys)."""
    __slots__ = ("_a", "_i")
    _LAZY = ("Power agent", "TimeLimit.truncated", "terminal_observation")

    def __init__(self, adapter, i):
        super().__init__()
        self._a, self._i = adapter, i

    def __missing__(self, key):
        a, i = self._a, self._i
        if key == "Power agent":
            return float(a._power[i])
        if key == "TimeLimit.truncated":
            return bool(a._trunc[i])
        if key == "terminal_observation" and a._trunc is not None and a._trunc[i]:
            return a._fin_row(i)
        raise KeyError(key)

    def __contains__(self, key):
        if dict.__contains__(self, key):
            return True
        if key == "terminal_observation":
            return self._a._trunc is not None and bool(self._a._trunc[self._i])
        return key in self._LAZY and self._a._power is not None

    def get(self, key, default=None):
        try:
            return self[key]
        except KeyError:
            return default

    def keys(self):
        return [k for k in self._LAZY if k in self] + [k for k in dict.keys(self) if k not in self._LAZY]

    def __iter__(self):
        return iter(self.keys())

    def __len__(self):
        return len(self.keys())

    def items(self):
        return [(k, self[k]) for k in self.keys()]

    def values(self):
        return [self[k] for k in self.keys()]

    def copy(self):
        return dict(self.items())

    def __repr__(self):
        return repr(self.copy())


_INFO_KEYS = {
    # reference key (Wind_Farm_Env.py:527-554)     -> wg_info_field name
    "yaw angles agent": "yaw_agent",
    "Wind speed Global": "ws_global",
    "Wind speed at turbines": "ws_turb",
    "Wind direction Global": "wd_global",
    "Wind direction at turbines": "wd_turb",
    "Turbulence intensity": "ti_global",
    "Power agent": "power_agent",
    "Power pr turbine agent": "power_turb_agent",
    "Turbine x positions": "turb_x",
    "Turbine y positions": "turb_y",
    "yaw angles base": "yaw_base",
    "Power baseline": "power_base",
    "Power pr turbine baseline": "power_turb_base",
    "Wind speed at turbines baseline": "ws_turb_base",
}
_BASE_ONLY = {"yaw angles base", "Power baseline", "Power pr turbine baseline", "Wind speed at turbines baseline"}


class _LazyInfo(dict):
    """Batched info dict with the reference's keys; each value is copied from the device when first read."""

    _STEP_FIELDS = {"Power agent": "step_power_agent", "Power baseline": "step_power_base"}

    def __init__(self, venv: WindFarmVecEnv, step=False):
        super().__init__()
        self._v = venv
        self._two = venv.cfg.baseline_comp
        self._step = step

    def _keys(self):
        return [k for k in _INFO_KEYS if self._two or k not in _BASE_ONLY]

    def __missing__(self, key):
        if key not in _INFO_KEYS or (key in _BASE_ONLY and not self._two):
            raise KeyError(key)
        field = self._STEP_FIELDS.get(key, _INFO_KEYS[key]) if self._step else _INFO_KEYS[key]
        val = self._v.batch.info(field)
        val = val if self._v.as_torch else _np(val)
        self[key] = val
        return val

    def __contains__(self, key):
        return dict.__contains__(self, key) or key in self._keys()

    def keys(self):
        return list(dict.keys(self)) + [k for k in self._keys() if not dict.__contains__(self, k)]

    def get(self, key, default=None):
        try:
            return self[key]
        except KeyError:
            return default


# ======================================================================================================
class WindFarmEnv(_EnvBase):
    """Single farm with the reference's API (WindGym/Wind_Farm_Env.py:47-1034), backed by a batch of 1.

    Differences, all documented in DESIGN.md §3: the flow physics is model M0 (DYNAMIKS is not available);
    ``HTC_path`` (HAWC2 turbines) is not part of the step() path and raises ``NotImplementedError``.
    ``sample_site`` takes any object with py_wake's ``Site.local_wind`` duck type (windgym_amd.site.WeibullSite).  Rendering draws the flow field evaluated on the device (k_windspeed) off-screen.
    """

    metadata = {"render_modes": ["human", "rgb_array"]}
    _extra_timestep_inc = False
    _never_truncate = False

    def __init__(self, turbine, n_passthrough=5, TI_min_mes: float = 0.0, TI_max_mes: float = 0.50,
                 TurbBox="Default", turbtype="MannLoad", yaml_path=None, Baseline_comp=False, yaw_init=None,
                 render_mode=None, seed=None, dt_sim=1, dt_env=1, yaw_step=1, fill_window=True, sample_site=None,
                 HTC_path=None, reset_init=True, *, device=None, yaml_dict=None, n_particles=None,
                 n_rotor_pts=16, x_pos=None, y_pos=None, turbulence_box=None, **model_options):
        # model_options: switches of the flow model that have no counterpart in the reference's constructor — EnvConfig's
        # ``deficit`` ("gaussian" | "super_gaussian" | "ainslie"), ``model_constants``, ``added_turbulence``, ``wake_ti_fold``,
        # ``mann_pool``
        unknown = set(model_options) - {"deficit", "model_constants", "added_turbulence", "wake_ti_fold", "mann_pool"}
        if unknown:
            raise TypeError(f"unexpected keyword argument(s): {sorted(unknown)}")
        if HTC_path is not None:
            raise NotImplementedError("HAWC2 turbines (HTC_path) are outside the MI355X step() path")
        self.sample_site = sample_site
        self._site_rng = np.random.default_rng(seed)     # the reference draws from numpy's global, unseeded state
        assert render_mode is None or render_mode in self.metadata["render_modes"]
        self.render_mode = render_mode
        self.turbine = turbine
        self.seed = seed
        self._device = device
        self._kw = dict(turbine=turbine, n_passthrough=n_passthrough, TI_min_mes=TI_min_mes, TI_max_mes=TI_max_mes,
                        TurbBox=TurbBox, turbtype=turbtype, yaml_path=yaml_path, Baseline_comp=Baseline_comp,
                        yaw_init=yaw_init, seed=seed, dt_sim=dt_sim, dt_env=dt_env, yaw_step=yaw_step,
                        fill_window=fill_window, yaml_dict=yaml_dict, n_particles=n_particles,
                        n_rotor_pts=n_rotor_pts, x_pos=x_pos, y_pos=y_pos, n_envs=1, autoreset=False,
                        advect_full_chains=True,     # single envs render: keep the far wake exact (no chain pruning)
                        never_truncate=self._never_truncate, extra_timestep_inc=self._extra_timestep_inc, **model_options)
        self.yaw_initial = [0]
        self._overrides = {}
        self._turbulence_box = turbulence_box
        self._build()
        self.timestep = 0
        self._torn_down = True
        if reset_init:
            self.reset(seed=seed)

    # -- construction ---------------------------------------------------------------------------------
    def _build(self):
        kw = dict(self._kw)
        if self.yaw_initial is not None and len(self.yaw_initial) and kw.get("yaw_init") == "Defined":
            kw["yaw_defined"] = self.yaw_initial
        cfg = EnvConfig(**kw)
        for k, v in self._overrides.items():      # FarmEval.set_wind_vals
            setattr(cfg, k, v)
        old = getattr(self, "_batch", None)
        if old is not None:
            old.close()
        self.cfg = cfg
        if getattr(self, "_turb_source", None) is None:
            self._turb_source = _resolve_turbulence(cfg, self._turbulence_box)
            self._turbtype_resolved = cfg.turbtype
        else:
            cfg.turbtype = self._turbtype_resolved                              # (MannLoad -> MannGenerate fallback)
        self._batch = HipBatch(cfg, device=self._device)
        self._turb_source = _attach_box(self._batch, cfg, self._turb_source)   # generated once, reused on rebuilds
        c = cfg
        # attributes callers read
        self.n_turb, self.x_pos, self.y_pos = c.n_turb, c.x_pos, c.y_pos
        self.yaw_min, self.yaw_max, self.yaw_step = c.yaw_min, c.yaw_max, c.yaw_step
        self.ws_min, self.ws_max, self.TI_min, self.TI_max = c.ws_min, c.ws_max, c.TI_min, c.TI_max
        self.wd_min, self.wd_max = c.wd_min, c.wd_max
        self.Baseline_comp = c.baseline_comp
        self.power_reward, self.ActionMethod = c.power_reward, c.ActionMethod
        self.action_penalty, self.action_penalty_type = c.action_penalty, c.action_penalty_type
        self.hist_max, self.steps_on_reset = c.hist_max, c.steps_on_reset
        self.obs_var = self._batch.obs_dim
        self.D, self.maxturbpower = c.D, c.maxturbpower
        self.dt = self.dt_sim = c.dt_sim
        self.dt_env, self.sim_steps_per_env_step = c.dt_env, c.sim_steps_per_env_step
        self.n_passthrough = c.n_passthrough
        self.mes_level = c.mes_level
        self.fs = _FlowProxy(self._batch, "agent")
        self.fs_baseline = _FlowProxy(self._batch, "base") if c.baseline_comp else None
        self._init_spaces()
        self._dirty = False

    def _init_spaces(self):
        self.observation_space = Box(low=-1.0, high=1.0, shape=(self.obs_var,), dtype=np.float32)
        self.action_space = Box(low=-1, high=1, shape=(self.n_turb,), dtype=np.float32)

    # -- gymnasium API ---------------------------------------------------------------------------------
    def reset(self, seed: Optional[int] = None, options: Optional[dict] = None):
        if self._dirty:
            self._build()
        seeds = None if seed is None else np.array([seed], dtype=np.uint64)
        if self.sample_site is not None:                 # _set_windconditions with a site (:569-584)
            if seed is not None:
                self._site_rng = np.random.default_rng(seed)
            from .site import sample_site
            wd, ws = sample_site(self.sample_site, 1, self._site_rng, (self.wd_min, self.wd_max), (self.ws_min, self.ws_max))
            self._batch.set_wind(ws=ws, wd=wd, ti=None)  # TI stays uniform, drawn on the device
            self._site_active = True
        elif getattr(self, "_site_active", False):
            self._batch.set_wind()
            self._site_active = False
        obs = self._batch.reset(seeds=seeds)
        self._batch.check()
        self._torn_down = False
        self.timestep = 0
        self._refresh_episode_attrs()
        return _np(obs)[0].copy(), self._get_info()

    def _set_windconditions(self):
        """Host-side restatement of Wind_Farm_Env.py:557-584 for callers that probe it (the episode's own draw is
        made inside reset(): on the device without a site, from the site's wind rose with one)."""
        rng = self._site_rng
        if self.sample_site is None:
            self.ws = float(rng.uniform(self.ws_min, self.ws_max))
            self.ti = float(rng.uniform(self.TI_min, self.TI_max))
            self.wd = float(rng.uniform(self.wd_min, self.wd_max))
        else:
            from .site import sample_site
            wd, ws = sample_site(self.sample_site, 1, rng, (self.wd_min, self.wd_max), (self.ws_min, self.ws_max))
            self.wd, self.ws = float(wd[0]), float(ws[0])
            self.ti = float(rng.uniform(self.TI_min, self.TI_max))

    def _refresh_episode_attrs(self):
        b = self._batch
        self.ws, self.wd, self.ti = (float(v) for v in _np(b.info("wind_f64"))[0])
        self.time_max = int(_np(b.info("time_max"))[0])
        self.rated_power = float(_np(b.info("rated_power"))[0])

    def step(self, action):
        if self._torn_down:
            # the reference deletes fs/site/farm_measurements at truncation (:1003-1023)
            raise RuntimeError("step() called on a truncated environment: call reset() first")
        a = np.ascontiguousarray(action, dtype=np.float32).reshape(1, self.n_turb)
        self.old_yaws = self.fs.windTurbines.yaw                        # :932
        obs, rew, trunc, _ = self._batch.step(self._batch.torch.as_tensor(a, device=self._batch.device))
        self._batch.check()               # Exception("NaN Power") (:980-981)
        observation = _np(obs)[0].copy()
        reward = float(_np(rew)[0])
        truncated = bool(_np(trunc)[0])
        info = self._get_info()
        self.timestep = int(_np(self._batch.info("timestep"))[0])
        self.fs_time = self.fs.time
        if truncated:
            self._torn_down = True
        return observation, reward, False, truncated, info

    def _get_info(self):
        """Keys and shapes of WindFarmEnv._get_info (Wind_Farm_Env.py:522-555)."""
        b = self._batch
        g = lambda k: _np(b.info(k))[0].astype(np.float64)   # noqa: E731
        self._raw_cache = None
        d = {
            "yaw angles agent": g("yaw_agent"),
            "yaw angles measured": self._measured("yaw"),
            "Wind speed Global": float(g("ws_global")),
            "Wind speed at turbines": g("ws_turb"),
            "Wind speed at turbines measured": self._measured("ws"),
            "Wind speed at farm measured": self._measured("ws", farm=True),
            "Wind direction Global": float(g("wd_global")),
            "Wind direction at turbines": g("wd_turb"),
            "Wind direction at turbines measured": self._measured("wd"),
            "Wind direction at farm measured": self._measured("wd", farm=True),
            "Turbulence intensity": float(g("ti_global")),
            "Power agent": float(g("power_agent")),
            "Power pr turbine agent": g("power_turb_agent"),
            "Turbine x positions": g("turb_x"),
            "Turbine y positions": g("turb_y"),
        }
        if self.Baseline_comp:
            d["yaw angles base"] = g("yaw_base")
            d["Power baseline"] = float(g("power_base"))
            d["Power pr turbine baseline"] = g("power_turb_base")
            d["Wind speed at turbines baseline"] = g("ws_turb_base")
        return d

    def _measured(self, which, farm=False):
        """farm_measurements.get_{ws,wd,yaw}_turb() / get_{ws,wd}_farm() (Wind_Farm_Env.py:529-537): the unscaled
        sensor windows, sliced out of wg_get_measurements' vector."""
        raw = getattr(self, "_raw_cache", None)
        if raw is None:
            raw = self._raw_cache = _np(self._batch.measurements())[0]
        lay = self.cfg.obs_layout()
        if farm:
            o, n = lay["farm"][which]
            return raw[o:o + n].copy()
        o, n = lay["turb"][which]
        blk = lay["turb_block"]
        return np.concatenate([raw[t * blk + o: t * blk + o + n] for t in range(self.n_turb)]) if n else np.array([], dtype=np.float32)

    # -- helpers callers use -----------------------------------------------------------------------------
    def _get_num_raw_features(self):
        m = self.mes_level
        f = sum(self.n_turb for k in ("turb_ws", "turb_wd", "turb_TI", "turb_power") if m[k])
        return f + sum(1 for k in ("farm_ws", "farm_wd", "farm_TI", "farm_power") if m[k])

    def _action_penalty(self):
        """Wind_Farm_Env.py:804-820 (host-side restatement for callers that probe it; the reward itself is
        computed on the device)."""
        if self.action_penalty < 0.001:
            return 0
        yaw = self.fs.windTurbines.yaw
        if self.action_penalty_type == "Change":
            pen_val = np.mean(np.abs(getattr(self, "old_yaws", yaw) - yaw))
        else:
            pen_val = np.mean(np.abs(yaw)) / self.yaw_max
        return self.action_penalty * pen_val

    # -- flow-field view / rendering (Wind_Farm_Env.py:464-476, :1036-1083) --------------------------------
    def init_render(self):
        """The reference's view: x from 200 m upstream of the first to 1000 m behind the last turbine, y +-200 m
        around the farm, 250 x 250 points at hub height (:464-476); flow-frame coordinates."""
        x_turb, y_turb = self.fs.windTurbines.positions_xyz[:2]
        self.a = np.linspace(-200 + min(x_turb), 1000 + max(x_turb), 250)
        self.b = np.linspace(-200 + min(y_turb), 200 + max(y_turb), 250)
        self.view = dict(z=self.turbine.hub_height(), x=self.a, y=self.b)

    def get_windspeed(self, x=None, y=None, z=None, baseline=False, include_wakes=True):
        """fs.get_windspeed(XYView(z, x, y), include_wakes) -> np.float32 [3, nx, ny] (u, v, w), evaluated by
        k_windspeed on the device from the env's current particle state."""
        if x is None or y is None:
            if not hasattr(self, "view"):
                self.init_render()
            x = self.view["x"] if x is None else x
            y = self.view["y"] if y is None else y
        if baseline and not self.Baseline_comp:
            raise ValueError("no baseline farm in this env (Baseline_comp is off)")
        return _np(self._batch.windspeed(0, x, y, z=z, farm=1 if baseline else 0, include_wakes=include_wakes))

    def _render_frame(self, baseline=False):
        """Flow field (u component) + turbines, like the reference's _render_frame (:1040-1083); returns the frame as
        uint8 [H, W, 3] (off-screen Agg canvas — no display / IPython on a GPU box)."""
        from matplotlib.backends.backend_agg import FigureCanvasAgg
        from matplotlib.figure import Figure
        if not hasattr(self, "view"):
            self.init_render()
        uvw = self.get_windspeed(baseline=baseline)
        fs_use = self.fs_baseline if baseline else self.fs
        wt = fs_use.windTurbines
        x_turb, y_turb = wt.positions_xyz[:2]
        fig = Figure(figsize=(10, 4), dpi=100)
        canvas = FigureCanvasAgg(fig)
        ax = fig.add_subplot(111)
        pc = ax.pcolormesh(self.view["x"], self.view["y"], uvw[0].T, shading="nearest")
        fig.colorbar(pc, ax=ax, label="u [m/s]")
        R = 0.5 * self.turbine.diameter()
        for xt, yt, g in zip(x_turb, y_turb, np.deg2rad(np.asarray(wt.yaw, dtype=float))):
            # rotor disc seen from above: a line normal to the (yawed) rotor axis
            ax.plot([xt + R * np.sin(g), xt - R * np.sin(g)], [yt - R * np.cos(g), yt + R * np.cos(g)], "k-", lw=2)
        ax.set_aspect("equal")
        ax.set_xlabel("x [m]"), ax.set_ylabel("y [m]")
        ax.set_title("Flow field at {} s".format(fs_use.time))
        canvas.draw()
        return np.asarray(canvas.buffer_rgba())[..., :3].copy()

    def render(self):
        if self.render_mode == "rgb_array":
            return self._render_frame()
        return None

    def plot_frame(self, baseline=False):
        """Plots a single frame of the flow field and the wind turbines (:1098-1103); returns the image."""
        self.init_render()
        return self._render_frame(baseline=baseline)

    def close(self):
        b = getattr(self, "_batch", None)
        if b is not None:
            b.close()
            self._batch = None


class FarmEval(WindFarmEnv):
    """WindGym/FarmEval.py:10-90: fixed wind values, never truncates, optional initial yaws."""

    _never_truncate = True

    def __init__(self, turbine, TI_min_mes: float = 0.0, TI_max_mes: float = 0.50, yaw_init="Zeros",
                 TurbBox="Default", yaml_path=None, Baseline_comp=False, render_mode=None, turbtype="MannLoad",
                 seed=None, dt_sim=1, dt_env=1, yaw_step=1, n_passthrough=5, HTC_path=None, reset_init=True, **kw):
        super().__init__(turbine=turbine, n_passthrough=n_passthrough, TI_min_mes=TI_min_mes, TI_max_mes=TI_max_mes,
                         TurbBox=TurbBox, turbtype=turbtype, yaml_path=yaml_path, Baseline_comp=Baseline_comp,
                         yaw_init=yaw_init, render_mode=render_mode, seed=seed, dt_sim=dt_sim, dt_env=dt_env,
                         yaw_step=yaw_step, HTC_path=HTC_path, reset_init=reset_init, **kw)

    def reset(self, seed=None, options=None):
        observation, info = super().reset(seed=seed, options=options)
        self.time_max = 9999999                                         # FarmEval.py:54-61
        return observation, info

    def set_wind_vals(self, ws=None, ti=None, wd=None):                # FarmEval.py:63-78
        if ws is not None:
            self.ws = ws
            self._overrides.update(ws_min=ws, ws_max=ws)
        if ti is not None:
            self.ti = ti
            self._overrides.update(TI_min=ti, TI_max=ti)
        if wd is not None:
            self.wd = wd
            self._overrides.update(wd_min=wd, wd_max=wd)
        self._dirty = True

    def set_yaw_vals(self, yaw_vals):                                  # FarmEval.py:80-84
        self.yaw_initial = yaw_vals
        self._kw["yaw_init"] = "Defined"
        self._dirty = True

    def update_tf(self, path):                                         # FarmEval.py:86-90
        self.TF_files = [path]


# ======================================================================================================
class WindFarmEnvMulti(_ParallelEnvBase):
    """PettingZoo ``ParallelEnv`` facade, one agent per turbine (WindGym/WindEnvMulti.py:17-249).

    Reproduced quirks: ``timestep`` advances twice per step (:219), the per-agent vector is the turbine block
    followed by ``farm_mes.farm_mes``' block and is *shorter* than the declared ``obs_var`` whenever yaw is
    observed.  Fixed (the reference raises there, tests/golden/make_golden.py): the
    constructor works with the real pettingzoo base class, and truncation returns instead of raising.
    """

    metadata = {"name": "MultiFarm_environment_v0"}

    def __init__(self, turbine, n_passthrough=20, TI_min_mes: float = 0.0, TI_max_mes: float = 0.50,
                 TurbBox="Default", turbtype="MannLoad", yaml_path=None, Baseline_comp=False, yaw_init=None,
                 render_mode=None, seed=None, dt_sim=1, dt_env=1, yaw_step=1, fill_window=True, sample_site=None,
                 **kw):
        class _Inner(WindFarmEnv):
            _extra_timestep_inc = True

        self._env = _Inner(turbine=turbine, n_passthrough=n_passthrough, TI_min_mes=TI_min_mes,
                           TI_max_mes=TI_max_mes, TurbBox=TurbBox, turbtype=turbtype, yaml_path=yaml_path,
                           Baseline_comp=Baseline_comp, yaw_init=yaw_init, render_mode=render_mode, seed=seed,
                           dt_sim=dt_sim, dt_env=dt_env, yaw_step=yaw_step, fill_window=fill_window,
                           sample_site=sample_site, reset_init=False, **kw)
        self.act_var = 1
        self.n_turb = self._env.n_turb
        self.obs_var = self._env.cfg.multi_declared_obs_var()           # as declared (:65-69)
        self.obs_len = self._env._batch.obs_dim_multi                   # as produced (:79-103)
        self.possible_agents = ["turbine_" + str(r) for r in range(self.n_turb)]
        self.agent_name_mapping = dict(zip(self.possible_agents, range(self.n_turb)))
        self.agents = []
        self.timestep = 0
        self._seed = seed
        self._was_reset = False

    def __getattr__(self, name):            # n_turb, ws, wd, fs, ... of the wrapped env
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self._env, name)

    def _get_obs_multi(self):
        om = _np(self._env._batch.obs_multi())[0]
        return {a: om[i].copy() for i, a in enumerate(self.agents)}

    def _get_infos(self):
        info = self._env._get_info()
        out = {}
        for a in self.agents:
            i = self.agent_name_mapping[a]
            out[a] = {
                "yaw angles agent": info["yaw angles agent"][i],
                "Wind speed Global": info["Wind speed Global"],
                "Wind speed at turbine": info["Wind speed at turbines"][i],
                "Wind direction Global": info["Wind direction Global"],
                "Wind direction at turbine": info["Wind direction at turbines"][i],
                "Turbulence intensity": info["Turbulence intensity"],
                "Power agent": info["Power agent"],
                "Power turbine agent": info["Power pr turbine agent"][i],
                "Turbine x positions": info["Turbine x positions"][i],
                "Turbine y positions": info["Turbine y positions"][i],
            }
        return out

    def reset(self, seed=None, options=None):
        if seed is None and not self._was_reset:
            seed = self._seed
        self._was_reset = True
        self._env.reset(seed, options)
        self.agents = copy.copy(self.possible_agents)
        self.timestep = 0
        return self._get_obs_multi(), self._get_infos()

    def step(self, actions):
        all_action = np.array([np.asarray(actions[a]).reshape(-1)[0] for a in self.agents], dtype=np.float32)
        _, reward, _, truncated, _ = self._env.step(all_action)
        observations = self._get_obs_multi()
        infos = self._get_infos()
        rewards = {a: reward for a in self.agents}
        truncations = {a: bool(truncated) for a in self.agents}
        terminations = {a: False for a in self.agents}
        self.timestep = self._env.timestep
        if all(truncations.values()):
            self.agents = []
        return observations, rewards, terminations, truncations, infos

    def observation_space(self, agent):
        return Box(low=-1.0, high=1.0, shape=(self.obs_var,), dtype=np.float32)

    def action_space(self, agent):
        return Box(low=-1.0, high=1.0, shape=(self.act_var,), dtype=np.float32)

    def close(self):
        self._env.close()


# ======================================================================================================
class RecordEpisodeVals:
    """Episode statistics of a :class:`WindFarmVecEnv` (wrappers/recordEpisodeVals.py:8-64): per-env running
    sum of ``infos["Power agent"]``, pushed as ``sum / episode_length`` into ``mean_power_queue`` when the
    episode ends; also ``return_queue`` / ``length_queue`` like gymnasium's RecordEpisodeStatistics.

    On a CUDA-tensor env (``as_torch=True``) the wrapper does not synchronise with the device every step (that alone made
    a 55 us step take 135 us, bench.py --api): the step's rewards, truncation flags and farm powers are parked in a device
    ring of ``flush_every`` steps and replayed on the host, in order, when the ring is full or a queue is read — the same
    arithmetic on the same values, a bounded number of steps later."""

    def __init__(self, env: WindFarmVecEnv, buffer_length=100, flush_every=64):
        self.env = env
        self.num_envs = env.num_envs
        self._mean_power_queue = deque(maxlen=buffer_length)
        self._return_queue = deque(maxlen=buffer_length)
        self._length_queue = deque(maxlen=buffer_length)
        self.episode_powers = np.zeros(self.num_envs)
        self.episode_returns = np.zeros(self.num_envs)
        self.episode_lengths = np.zeros(self.num_envs, dtype=np.int64)
        self._ring = None                    # device ring [flush_every, 3, B] float32 (reward, truncated, power) + fill count
        self._n_parked = 0
        self._flush_every = int(flush_every)

    def __getattr__(self, name):
        return getattr(self.env, name)

    # the queues: reading one first replays whatever is still parked on the device
    @property
    def mean_power_queue(self):
        self._flush()
        return self._mean_power_queue

    @property
    def return_queue(self):
        self._flush()
        return self._return_queue

    @property
    def length_queue(self):
        self._flush()
        return self._length_queue

    def reset(self, **kw):
        self._flush()
        out = self.env.reset(**kw)
        self.episode_powers[:] = 0
        self.episode_returns[:] = 0
        self.episode_lengths[:] = 0
        return out

    def _account(self, r, d, p):
        """one step of the reference wrapper's bookkeeping (recordEpisodeVals.py:43-56) on host arrays"""
        self.episode_powers += p
        self.episode_returns += r
        self.episode_lengths += 1
        for i in np.nonzero(d)[0]:
            self._mean_power_queue.append(self.episode_powers[i] / self.episode_lengths[i])
            self._return_queue.append(self.episode_returns[i])
            self._length_queue.append(int(self.episode_lengths[i]))
        self.episode_powers[d] = 0
        self.episode_returns[d] = 0
        self.episode_lengths[d] = 0

    def _flush(self):
        if self._n_parked:
            blk = _np(self._ring[:self._n_parked])          # ONE device-to-host copy for all parked steps
            n, self._n_parked = self._n_parked, 0
            for k in range(n):
                self._account(blk[k, 0].astype(np.float64), blk[k, 1] != 0, blk[k, 2].astype(np.float64))

    def step(self, actions):
        obs, rew, term, trunc, infos = self.env.step(actions)
        if getattr(self.env, "as_torch", False) and hasattr(rew, "is_cuda") and rew.is_cuda:
            t = self.env.torch
            if self._ring is None:
                self._ring = t.empty((self._flush_every, 3, self.num_envs), dtype=t.float32, device=rew.device)
            row = self._ring[self._n_parked]
            row[0].copy_(rew); row[1].copy_(trunc); row[2].copy_(infos["Power agent"])
            self._n_parked += 1
            if self._n_parked == self._flush_every:
                self._flush()
            return obs, rew, term, trunc, infos
        to_np = (lambda x: _np(x)) if self.env.as_torch else np.asarray
        self._account(to_np(rew), to_np(trunc).astype(bool), to_np(infos["Power agent"]))
        return obs, rew, term, trunc, infos
