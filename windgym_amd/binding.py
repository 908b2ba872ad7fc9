"""ctypes binding of libwindgym_hip.so (C ABI in include/windgym_hip.h) on PyTorch-ROCm tensors.

PyTorch is plumbing here: it owns the I/O device buffers and the HIP stream; all compute happens in the
hand-written kernels behind the C ABI.  There is NO CPU fallback: constructing a :class:`HipBatch` without
the built library or without a GPU raises.
"""
from __future__ import annotations

import ctypes as C
import os
import sys

import numpy as np

from .config import INFO, INFO_INT, WG_N_METRICS, CConfig, EnvConfig

_HERE = os.path.dirname(os.path.abspath(__file__))
# Measurement hooks, honoured ONLY with WG_DEBUG_HOOKS=1 (tools/ab.sh sets it) and announced on stderr when active:
#   WG_LIB      alternative build of the same library (same-box A/B measurements of kernel variants)
#   WG_NOCHECK  HipBatch.check() only synchronises (profiling builds that ablate parts of the kernels)
_HOOKS = os.environ.get("WG_DEBUG_HOOKS") == "1"
LIB_PATH = os.path.join(_HERE, "libwindgym_hip.so")
if _HOOKS and os.environ.get("WG_LIB"):
    LIB_PATH = os.environ["WG_LIB"]
    print(f"[windgym] WG_DEBUG_HOOKS: loading {LIB_PATH} instead of the in-tree library", file=sys.stderr)
_NOCHECK = bool(_HOOKS and os.environ.get("WG_NOCHECK"))
if _NOCHECK:
    print("[windgym] WG_DEBUG_HOOKS: WG_NOCHECK set — HipBatch.check() does not read the device error word", file=sys.stderr)
UINT64_MAX = 0xFFFFFFFFFFFFFFFF

# every symbol include/windgym_hip.h declares (tests check the built library exports all of them)
ABI_SYMBOLS = (
    "wg_last_error", "wg_abi_version", "wg_create", "wg_destroy", "wg_obs_dim", "wg_hist_max",
    "wg_set_turbulence_box", "wg_set_turbulence_boxes", "wg_set_added_turbulence_box", "wg_set_deficit_table", "wg_set_box_ids", "wg_set_wind", "wg_set_wind_device", "wg_set_flow_script", "wg_reset", "wg_step", "wg_set_step_graph", "wg_check", "wg_obs_multi", "wg_set_obs_multi_buffer",
    "wg_get_info", "wg_get_measurements", "wg_get_windspeed", "wg_metrics", "wg_get_state", "wg_set_state", "wg_generate_mann_box", "wg_mann_beta_table", "wg_steady_power", "wg_kernel_timing", "wg_added_lookups", "wg_algorithmic_bytes", "wg_flow_variant",
)

_lib = None


class WindGymHipError(RuntimeError):
    pass


def load_library():
    """Load libwindgym_hip.so; raises (never falls back) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise WindGymHipError(
            f"{LIB_PATH} is missing: build it with `python -m windgym_amd.build` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the step() path.")
    # torch first: its bundled HIP runtime must be the one the process initialises — loading ours against the
    # system libamdhip64 before `import torch` leaves two runtimes in the process and hipSetDevice then fails with
    # "no ROCm-capable device is detected"
    import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    L.wg_last_error.restype = C.c_char_p
    L.wg_create.argtypes = [C.POINTER(CConfig), C.c_int, C.POINTER(C.c_void_p)]
    L.wg_destroy.argtypes = [C.c_void_p]
    L.wg_obs_dim.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.wg_hist_max.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    L.wg_set_turbulence_box.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double,
                                        C.c_double, C.c_double]
    L.wg_set_box_ids.argtypes = [C.c_void_p, C.c_void_p]
    L.wg_set_deficit_table.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_int, C.c_double, C.c_double,
                                       C.c_int, C.c_double, C.c_int, C.c_double]
    L.wg_set_added_turbulence_box.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double,
                                              C.c_double, C.c_double]
    L.wg_set_turbulence_boxes.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int,
                                          C.c_double, C.c_double, C.c_double]
    L.wg_set_flow_script.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    L.wg_set_wind.argtypes = [C.c_void_p, C.c_void_p]
    L.wg_set_wind_device.argtypes = [C.c_void_p, C.c_void_p]
    L.wg_reset.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.wg_step.argtypes = [C.c_void_p] + [C.c_void_p] * 6
    L.wg_set_step_graph.argtypes = [C.c_void_p, C.c_int]
    L.wg_check.argtypes = [C.c_void_p, C.c_void_p]
    L.wg_obs_multi.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.wg_set_obs_multi_buffer.argtypes = [C.c_void_p, C.c_void_p]
    L.wg_get_info.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.wg_get_measurements.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.wg_get_windspeed.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_float,
                                   C.c_int, C.c_void_p, C.c_void_p]
    L.wg_metrics.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    L.wg_get_state.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t)]
    L.wg_set_state.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.wg_kernel_timing.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double),
                                   C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.wg_added_lookups.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    L.wg_generate_mann_box.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double,
                                       C.c_double, C.c_double, C.c_double, C.c_uint64, C.c_void_p, C.c_void_p]
    L.wg_steady_power.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 6
    L.wg_mann_beta_table.argtypes = [C.c_double, C.c_int, C.c_double, C.c_double, C.POINTER(C.c_double)]
    L.wg_algorithmic_bytes.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    L.wg_flow_variant.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    _lib = L
    return L


def _chk(rc, what):
    if rc != 0:
        msg = load_library().wg_last_error().decode(errors="replace")
        if rc == -4:
            raise Exception("NaN Power")                       # Wind_Farm_Env.py:980-981
        if rc == -2:
            raise NotImplementedError(msg)
        if rc == -1:
            raise ValueError(msg)
        raise WindGymHipError(f"{what} failed (rc={rc}): {msg}")


class HipBatch:
    """A batch of ``cfg.n_envs`` farms resident on one MI355X."""

    def __init__(self, cfg: EnvConfig, device: int | None = None):
        import torch
        if not torch.cuda.is_available():
            raise WindGymHipError("no HIP device visible: the step() path only runs on the GPU "
                                  "(no CPU fallback; the CPU restatement under oracle/ is test-only)")
        self.torch = torch
        self.L = load_library()
        self.cfg = cfg
        self.device_index = torch.cuda.current_device() if device is None else int(device)
        self.device = torch.device("cuda", self.device_index)
        self._c = cfg.to_c()
        h = C.c_void_p()
        _chk(self.L.wg_create(C.byref(self._c), self.device_index, C.byref(h)), "wg_create")
        self._h = h
        o, om = C.c_int(), C.c_int()
        self.L.wg_obs_dim(self._h, C.byref(o), C.byref(om))
        self.obs_dim, self.obs_dim_multi = o.value, om.value
        self.B, self.N = cfg.n_envs, cfg.n_turb
        f32 = dict(dtype=torch.float32, device=self.device)
        self.obs = torch.zeros((self.B, self.obs_dim), **f32)
        self.final_obs = torch.zeros((self.B, self.obs_dim), **f32)
        self.reward = torch.zeros(self.B, **f32)
        self.truncated = torch.zeros(self.B, dtype=torch.uint8, device=self.device)
        self._metrics = torch.zeros(WG_N_METRICS, **f32)
        # step() hot path: the persistent output pointers and the bound C function, prepared once
        self._out_ptrs = (self.obs.data_ptr(), self.reward.data_ptr(), self.truncated.data_ptr(),
                          self.final_obs.data_ptr())
        self._wg_step = self.L.wg_step
        self._cur_stream = torch.cuda.current_stream
        self._script = None
        self._box = None

    # -- lifetime -----------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self.L.wg_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    # -- API ----------------------------------------------------------------------------------------
    def reset(self, seeds=None, mask=None):
        sp = mp = None
        if seeds is not None:
            s = np.ascontiguousarray(np.asarray(seeds, dtype=np.uint64).reshape(self.B))
            sp = s.ctypes.data_as(C.c_void_p)
        if mask is not None:
            m = np.ascontiguousarray(np.asarray(mask, dtype=np.uint8).reshape(self.B))
            mp = m.ctypes.data_as(C.c_void_p)
        if self._c.added_turbulence and not getattr(self, "_abox_set", False):
            from .mann import default_added_box
            self.set_added_turbulence_box(*default_added_box())
        if self._c.deficit_model == 2 and getattr(self, "_dtab", None) is None:
            from .ainslie import deficit_table
            self.set_deficit_table(*deficit_table())
        _chk(self.L.wg_reset(self._h, mp, sp, C.c_void_p(self.obs.data_ptr()), self._stream()), "wg_reset")
        return self.obs

    def step(self, actions):
        """actions: float32 CUDA tensor [B, N].  Returns views of the persistent output tensors."""
        if not (actions.is_cuda and actions.dtype == self.torch.float32 and actions.is_contiguous()
                and actions.numel() == self.B * self.N):
            raise ValueError("step(): actions must be a contiguous float32 CUDA tensor [B, N]")
        o, r, t, f = self._out_ptrs
        rc = self._wg_step(self._h, actions.data_ptr(), o, r, t, f, self._cur_stream(self.device).cuda_stream)
        if rc:
            _chk(rc, "wg_step")
        return self.obs, self.reward, self.truncated, self.final_obs

    def set_step_graph(self, enable=True):
        """step() as one hipGraphLaunch (captured per distinct set of I/O pointers) instead of direct launches."""
        _chk(self.L.wg_set_step_graph(self._h, int(bool(enable))), "wg_set_step_graph")

    def check(self):
        if _NOCHECK:          # profiling builds that ablate parts of the kernels (tools/ab.sh)
            self.torch.cuda.synchronize()
            return
        _chk(self.L.wg_check(self._h, self._stream()), "wg_check")

    def obs_multi(self):
        out = self.torch.zeros((self.B, self.N, self.obs_dim_multi), dtype=self.torch.float32, device=self.device)
        _chk(self.L.wg_obs_multi(self._h, C.c_void_p(out.data_ptr()), self._stream()), "wg_obs_multi")
        return out

    def fuse_obs_multi(self, enable=True):
        """Let every following step() / reset() also write the per-agent observations (PettingZoo facade) into a
        persistent tensor [B, N, obs_dim_multi] — returned here, updated in place — instead of a separate
        obs_multi() launch per step."""
        if not enable:
            _chk(self.L.wg_set_obs_multi_buffer(self._h, None), "wg_set_obs_multi_buffer")
            self._multi_buf = None
            return None
        t = self.torch
        self._multi_buf = t.zeros((self.B, self.N, self.obs_dim_multi), dtype=t.float32, device=self.device)
        _chk(self.L.wg_set_obs_multi_buffer(self._h, C.c_void_p(self._multi_buf.data_ptr())), "wg_set_obs_multi_buffer")
        return self._multi_buf

    def measurements(self):
        """Unscaled sensor values in the layout of the observation, f32[B, O]."""
        out = self.torch.zeros((self.B, self.obs_dim), dtype=self.torch.float32, device=self.device)
        _chk(self.L.wg_get_measurements(self._h, C.c_void_p(out.data_ptr()), self._stream()), "wg_get_measurements")
        return out

    def windspeed(self, env, x, y, z=None, farm=0, include_wakes=True):
        """(u, v, w) of one farm of one env on the grid x[nx] x y[ny] at height z (default: hub height), flow frame:
        f32[3, nx, ny] — fs.get_windspeed(XYView(...)) of the reference (Wind_Farm_Env.py:1040-1083)."""
        t = self.torch
        xs = t.as_tensor(x, dtype=t.float32, device=self.device).contiguous()
        ys = t.as_tensor(y, dtype=t.float32, device=self.device).contiguous()
        out = t.empty((3, xs.numel(), ys.numel()), dtype=t.float32, device=self.device)
        zz = float(self.cfg.tab.hub_height() if z is None else z)
        _chk(self.L.wg_get_windspeed(self._h, C.c_int(int(env)), C.c_int(int(farm)), C.c_void_p(xs.data_ptr()),
                                     C.c_int(xs.numel()), C.c_void_p(ys.data_ptr()), C.c_int(ys.numel()),
                                     C.c_float(zz), C.c_int(1 if include_wakes else 0), C.c_void_p(out.data_ptr()),
                                     self._stream()), "wg_get_windspeed")
        return out

    def info(self, name):
        t = self.torch
        per_turb = name in ("yaw_agent", "yaw_base", "ws_turb", "wd_turb", "power_turb_agent",
                            "power_turb_base", "ws_turb_base", "turb_x", "turb_y")
        if name == "wind_f64":
            out = t.zeros((self.B, 3), dtype=t.float64, device=self.device)
            _chk(self.L.wg_get_info(self._h, INFO[name], C.c_void_p(out.data_ptr()), self._stream()), "wg_get_info")
            return out
        if name.startswith("rotor_uvw"):
            shape = (self.B, self.N, 3)
        elif per_turb:
            shape = (self.B, self.N)
        else:
            shape = (self.B,)
        out = t.zeros(shape, dtype=t.int32 if name in INFO_INT else t.float32, device=self.device)
        _chk(self.L.wg_get_info(self._h, INFO[name], C.c_void_p(out.data_ptr()), self._stream()), "wg_get_info")
        return out

    def metrics(self, reset_after=False):
        _chk(self.L.wg_metrics(self._h, C.c_void_p(self._metrics.data_ptr()), int(reset_after), self._stream()),
             "wg_metrics")
        return self._metrics

    def set_wind(self, ws=None, wd=None, ti=None):
        """Fix the wind conditions per env (arrays of length B or scalars; None keeps the sampled value)."""
        if ws is None and wd is None and ti is None:
            _chk(self.L.wg_set_wind(self._h, None), "wg_set_wind")
            return
        w = np.full((self.B, 3), np.nan)
        for k, v in enumerate((ws, wd, ti)):
            if v is not None:
                w[:, k] = np.broadcast_to(np.asarray(v, dtype=np.float64), (self.B,))
        w = np.ascontiguousarray(w)
        _chk(self.L.wg_set_wind(self._h, w.ctypes.data_as(C.c_void_p)), "wg_set_wind")

    def set_box_ids(self, ids):
        """Box of the pool each env uses from the next reset on (FarmEval.update_tf per env); None = draw again."""
        if ids is None:
            _chk(self.L.wg_set_box_ids(self._h, None), "wg_set_box_ids")
            return
        a = np.ascontiguousarray(np.broadcast_to(np.asarray(ids, dtype=np.int32), (self.B,)))
        _chk(self.L.wg_set_box_ids(self._h, a.ctypes.data_as(C.c_void_p)), "wg_set_box_ids")

    def set_wind_device(self, wind):
        """Borrow a CUDA float64 tensor [B, 3] = (ws, wd, ti; NaN = keep the sampled value) as the per-env wind
        override; the caller may rewrite it between steps (stream-ordered).  None removes it."""
        if wind is None:
            self._wind_dev = None
            _chk(self.L.wg_set_wind_device(self._h, None), "wg_set_wind_device")
            return
        t = self.torch
        assert wind.is_cuda and wind.dtype == t.float64 and wind.is_contiguous() and tuple(wind.shape) == (self.B, 3)
        self._wind_dev = wind                                   # keep it alive
        _chk(self.L.wg_set_wind_device(self._h, C.c_void_p(wind.data_ptr())), "wg_set_wind_device")

    def set_flow_script(self, uvw, power):
        """Replay mode (test hook).  uvw [F,T,B,N,3], power [F,T,B,N] (array-likes)."""
        t = self.torch
        if uvw is None:
            self._script = None
            _chk(self.L.wg_set_flow_script(self._h, None, None, 0), "wg_set_flow_script")
            return
        u = t.as_tensor(np.ascontiguousarray(uvw, dtype=np.float32), device=self.device).contiguous()
        p = t.as_tensor(np.ascontiguousarray(power, dtype=np.float32), device=self.device).contiguous()
        self._script = (u, p)
        _chk(self.L.wg_set_flow_script(self._h, C.c_void_p(u.data_ptr()), C.c_void_p(p.data_ptr()),
                                       int(u.shape[1])), "wg_set_flow_script")

    def set_turbulence_box(self, box, spacing):
        t = self.torch
        b = box if isinstance(box, t.Tensor) else t.as_tensor(np.ascontiguousarray(box, dtype=np.float32))
        b = b.to(self.device, dtype=t.float32).contiguous()
        assert b.ndim == 4 and b.shape[0] == 3
        self._box = b
        _chk(self.L.wg_set_turbulence_box(self._h, C.c_void_p(b.data_ptr()), int(b.shape[1]), int(b.shape[2]),
                                          int(b.shape[3]), float(spacing[0]), float(spacing[1]),
                                          float(spacing[2])), "wg_set_turbulence_box")

    def set_added_turbulence_box(self, box, spacing):
        """Isotropic unit-variance box of the wake-added turbulence (wg_config.added_turbulence); default:
        ``mann.default_added_box()`` installed at the first reset."""
        t = self.torch
        b = box if isinstance(box, t.Tensor) else t.as_tensor(np.ascontiguousarray(box, dtype=np.float32))
        b = b.to(self.device, dtype=t.float32).contiguous()
        assert b.ndim == 4 and b.shape[0] == 3
        _chk(self.L.wg_set_added_turbulence_box(self._h, C.c_void_p(b.data_ptr()), int(b.shape[1]), int(b.shape[2]),
                                                int(b.shape[3]), float(spacing[0]), float(spacing[1]),
                                                float(spacing[2])), "wg_set_added_turbulence_box")
        self._abox_set = True

    def set_deficit_table(self, table, spec):
        """Deficit table of ``deficit_model`` 2 (``ainslie.deficit_table()`` is installed at the first reset by default):
        table [n_ct, n_ti, n_x, n_r] of 1 - U / U0, ``spec`` its axes (ainslie.TABLE_SPEC)."""
        t = self.torch
        tb = t.as_tensor(np.ascontiguousarray(table, dtype=np.float32)).to(self.device).contiguous()
        assert tb.ndim == 4 and tb.shape[2] == spec["n_x"] and tb.shape[3] == spec["n_r"]
        _chk(self.L.wg_set_deficit_table(self._h, C.c_void_p(tb.data_ptr()), int(tb.shape[0]), float(spec["ct"][0]),
                                         float(spec["ct"][-1]), int(tb.shape[1]), float(spec["ti"][0]), float(spec["ti"][-1]),
                                         int(spec["n_x"]), float(spec["x_max_D"]), int(spec["n_r"]), float(spec["r_max_R"])),
             "wg_set_deficit_table")
        self._dtab = tb              # borrowed by the handle

    def set_turbulence_boxes(self, boxes, spacing):
        """Pool of K boxes of equal shape (turbtype "MannLoad": one TF_* file drawn per reset, :611-618)."""
        t = self.torch
        bs = []
        for box in boxes:
            b = box if isinstance(box, t.Tensor) else t.as_tensor(np.ascontiguousarray(box, dtype=np.float32))
            b = b.to(self.device, dtype=t.float32).contiguous()
            assert b.ndim == 4 and b.shape[0] == 3 and (not bs or b.shape == bs[0].shape)
            bs.append(b)
        ptrs = (C.c_void_p * len(bs))(*[b.data_ptr() for b in bs])
        _chk(self.L.wg_set_turbulence_boxes(self._h, ptrs, len(bs), int(bs[0].shape[1]), int(bs[0].shape[2]),
                                            int(bs[0].shape[3]), float(spacing[0]), float(spacing[1]),
                                            float(spacing[2])), "wg_set_turbulence_boxes")
        self._box = bs[0]      # (the library keeps its own copies; the caller's tensors may go)

    def get_state(self) -> bytes:
        n = C.c_size_t(0)
        _chk(self.L.wg_get_state(self._h, None, C.byref(n)), "wg_get_state")
        buf = (C.c_char * n.value)()
        _chk(self.L.wg_get_state(self._h, buf, C.byref(n)), "wg_get_state")
        return bytes(buf)

    def set_state(self, blob: bytes):
        _chk(self.L.wg_set_state(self._h, blob, len(blob)), "wg_set_state")

    def kernel_timing(self, enable=True):
        """-> (flow kernel ms/launch, glue kernel ms/launch, launches timed, farm flow-steps per launch, wake particles
        streamed per launch).  enable: False/0 = stop, True/1 = time every step(), n > 1 = time every n-th step()."""
        f, g, n, fs, pt = C.c_double(), C.c_double(), C.c_int(), C.c_double(), C.c_double()
        _chk(self.L.wg_kernel_timing(self._h, int(enable), C.byref(f), C.byref(g), C.byref(n), C.byref(fs),
                                     C.byref(pt)), "wg_kernel_timing")
        return f.value, g.value, n.value, fs.value, pt.value

    def steady_power(self, ws, wd, ti, yaw, model="m0"):
        """Steady-state power per turbine [n_cases, N] (W) for the wind conditions ws / wd / ti [n_cases] and yaw vectors
        [n_cases, N] (degrees): ONE launch of k_steady (wg_steady_power).  model "m0" = the env's own flow model,
        "blondel_jimenez" = the reference PyWakeAgent's model."""
        t = self.torch
        f = lambda a: t.as_tensor(np.array(a, dtype=np.float32) if not isinstance(a, t.Tensor) else a, dtype=t.float32,
                                  device=self.device).contiguous()          # noqa: E731
        yaw = f(yaw).reshape(-1, self.cfg.n_turb)
        n = yaw.shape[0]
        ws, wd, ti = (f(a).reshape(-1).expand(n).contiguous() for a in (ws, wd, ti))
        out = t.empty((n, self.cfg.n_turb), dtype=t.float32, device=self.device)
        _chk(self.L.wg_steady_power(self._h, {"m0": 0, "blondel_jimenez": 1}[model], n, ws.data_ptr(), wd.data_ptr(),
                                    ti.data_ptr(), yaw.data_ptr(), out.data_ptr(), self._stream()), "wg_steady_power")
        return out

    def added_lookups(self):
        """Rotor points per flow launch at which the wake-added turbulence box was looked up (window of the last
        kernel_timing call)."""
        v = C.c_double()
        _chk(self.L.wg_added_lookups(self._h, C.byref(v)), "wg_added_lookups")
        return v.value

    def algorithmic_bytes(self):
        v = C.c_double()
        _chk(self.L.wg_algorithmic_bytes(self._h, C.byref(v)), "wg_algorithmic_bytes")
        return v.value

    def flow_variant(self):
        """(threads per k_flow workgroup, compact rings / pair-major phases?, farm slots per wave: 0 = one (k_flow),
        2 = every slot of an env or of one of its contexts (k_flow_env / k_flow_envb: one or two waves per env))"""
        b, r, d = C.c_int(), C.c_int(), C.c_int()
        _chk(self.L.wg_flow_variant(self._h, C.byref(b), C.byref(r), C.byref(d)), "wg_flow_variant")
        return b.value, bool(r.value), d.value
