"""ORACLE / TEST INFRASTRUCTURE — ctypes binding of oracle/liboracle.so (see wg_oracle.c header).

Only tests/, ``__graft_entry__.smoke()`` and bench.py's ``cpu_baseline`` leg may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
UINT64_MAX = np.uint64(0xFFFFFFFFFFFFFFFF)


def build(force=False):
    """Compile the C restatement (gcc; seconds)."""
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("wg_oracle.c", "rng_api.c", "pcg64.h", "philox.h")]
    srcs.append(os.path.join(_HERE, "..", "include", "windgym_hip.h"))
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.run(["make", "-C", _HERE, "-s"], check=True)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())      # (rebuilds when a source or include/windgym_hip.h — the ABI version — is newer than the library)
    return _LIB


_dp = C.POINTER(C.c_double)


def _d(a):
    return a.ctypes.data_as(_dp)


class Oracle:
    """Batched CPU env with the reference's semantics; ``precision`` 'f64' (default) or 'f32'."""

    def __init__(self, cfg, precision="f64"):
        self.cfg = cfg
        self._c = cfg.to_c()
        L = lib()
        pre = "wgo_" if precision == "f64" else "wgof_"
        self._f = {n: getattr(L, pre + n) for n in (
            "create", "destroy", "obs_dim", "hist_max", "set_flow_script", "set_turbulence_box", "set_turbulence_boxes", "set_added_turbulence_box", "set_deficit_table", "set_box_ids", "reset",
            "step", "obs_multi", "get_obs", "get_info", "metrics", "get_chain", "set_threads", "max_threads",
            "get_windspeed")}
        self._f["create"].restype = C.c_void_p
        self._f["create"].argtypes = [C.c_void_p]
        self._h = C.c_void_p(self._f["create"](C.byref(self._c)))
        if not self._h:
            raise RuntimeError("oracle create failed")
        o, om = C.c_int(), C.c_int()
        self._f["obs_dim"](self._h, C.byref(o), C.byref(om))
        self.obs_dim, self.obs_dim_multi = o.value, om.value
        self.B, self.N = cfg.n_envs, cfg.n_turb
        self._script = None
        self._box = None
        self._abox = None
        self._dtab = None

    def close(self):
        if self._h:
            self._f["destroy"](self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_threads(self, n):
        self._f["set_threads"](C.c_int(n))

    def max_threads(self):
        return int(self._f["max_threads"]())

    def set_flow_script(self, uvw, power):
        """uvw [F,T,B,N,3], power [F,T,B,N] (float64)."""
        if uvw is None:
            self._f["set_flow_script"](self._h, None, None, C.c_long(0))
            self._script = None
            return
        uvw = np.ascontiguousarray(uvw, dtype=np.float64)
        power = np.ascontiguousarray(power, dtype=np.float64)
        self._script = (uvw, power)
        self._f["set_flow_script"](self._h, _d(uvw), _d(power), C.c_long(uvw.shape[1]))

    def set_turbulence_box(self, box, spacing):
        box = np.ascontiguousarray(box, dtype=np.float32)
        assert box.ndim == 4 and box.shape[0] == 3
        self._box = box
        self._f["set_turbulence_box"](self._h, box.ctypes.data_as(C.c_void_p), C.c_int(box.shape[1]),
                                      C.c_int(box.shape[2]), C.c_int(box.shape[3]), C.c_double(spacing[0]),
                                      C.c_double(spacing[1]), C.c_double(spacing[2]))

    def set_added_turbulence_box(self, box, spacing):
        box = np.ascontiguousarray(box, dtype=np.float32)
        assert box.ndim == 4 and box.shape[0] == 3
        self._abox = box
        self._f["set_added_turbulence_box"](self._h, box.ctypes.data_as(C.c_void_p), C.c_int(box.shape[1]),
                                            C.c_int(box.shape[2]), C.c_int(box.shape[3]), C.c_double(spacing[0]),
                                            C.c_double(spacing[1]), C.c_double(spacing[2]))

    def set_deficit_table(self, table, spec):
        tb = np.ascontiguousarray(table, dtype=np.float32)
        self._dtab = tb
        self._f["set_deficit_table"](self._h, tb.ctypes.data_as(C.c_void_p), C.c_int(tb.shape[0]), C.c_double(spec["ct"][0]),
                                     C.c_double(spec["ct"][-1]), C.c_int(tb.shape[1]), C.c_double(spec["ti"][0]),
                                     C.c_double(spec["ti"][-1]), C.c_int(spec["n_x"]), C.c_double(spec["x_max_D"]),
                                     C.c_int(spec["n_r"]), C.c_double(spec["r_max_R"]))

    def set_box_ids(self, ids):
        self._box_ids = None if ids is None else np.ascontiguousarray(np.broadcast_to(np.asarray(ids, dtype=np.int32), (self.B,)))
        self._f["set_box_ids"](self._h, None if ids is None else self._box_ids.ctypes.data_as(C.c_void_p))

    def set_turbulence_boxes(self, boxes, spacing):
        bs = [np.ascontiguousarray(b, dtype=np.float32) for b in boxes]
        assert all(b.shape == bs[0].shape and b.ndim == 4 and b.shape[0] == 3 for b in bs)
        self._box = bs
        ptrs = (C.c_void_p * len(bs))(*[b.ctypes.data for b in bs])
        rc = self._f["set_turbulence_boxes"](self._h, ptrs, C.c_int(len(bs)), C.c_int(bs[0].shape[1]), C.c_int(bs[0].shape[2]),
                                             C.c_int(bs[0].shape[3]), C.c_double(spacing[0]), C.c_double(spacing[1]),
                                             C.c_double(spacing[2]))
        assert rc == 0

    def windspeed(self, env, x, y, z=None, farm=0, include_wakes=True):
        """(u, v, w)[3, nx, ny] of one farm of one env on an XY grid at height z (flow frame)."""
        xs = np.ascontiguousarray(x, dtype=np.float64)
        ys = np.ascontiguousarray(y, dtype=np.float64)
        out = np.zeros((3, xs.size, ys.size))
        zz = float(self.cfg.tab.hub_height() if z is None else z)
        rc = self._f["get_windspeed"](self._h, C.c_int(int(env)), C.c_int(int(farm)), _d(xs), C.c_int(xs.size), _d(ys),
                                      C.c_int(ys.size), C.c_double(zz), C.c_int(1 if include_wakes else 0), _d(out))
        if rc != 0:
            raise RuntimeError("get_windspeed failed")
        return out

    def reset(self, seeds=None, mask=None):
        if self._c.added_turbulence and self._abox is None:
            # the default isotropic field of the wake-added turbulence: the same array the HIP batch installs
            from windgym_amd.mann import default_added_box
            self.set_added_turbulence_box(*default_added_box())
        if self._c.deficit_model == 2 and self._dtab is None:
            from windgym_amd.ainslie import deficit_table
            self.set_deficit_table(*deficit_table())
        obs = np.zeros((self.B, self.obs_dim))
        sp = None
        if seeds is not None:
            s = np.ascontiguousarray(seeds, dtype=np.uint64)
            sp = s.ctypes.data_as(C.c_void_p)
        mp = None
        if mask is not None:
            m = np.ascontiguousarray(mask, dtype=np.uint8)
            mp = m.ctypes.data_as(C.c_void_p)
        rc = self._f["reset"](self._h, mp, sp, _d(obs))
        if rc:
            raise RuntimeError(f"oracle reset rc={rc}")
        return obs

    def step(self, actions):
        a = np.ascontiguousarray(actions, dtype=np.float64).reshape(self.B, self.N)
        obs = np.zeros((self.B, self.obs_dim))
        fin = np.zeros((self.B, self.obs_dim))
        rew = np.zeros(self.B)
        tr = np.zeros(self.B, dtype=np.uint8)
        rc = self._f["step"](self._h, _d(a), _d(obs), _d(rew), tr.ctypes.data_as(C.c_void_p), _d(fin))
        if rc == -5:
            raise RuntimeError("step() on a truncated env without reset")
        if rc == -4:
            raise Exception("NaN Power")
        return obs, rew, tr.astype(bool), fin

    def get_obs(self):
        obs = np.zeros((self.B, self.obs_dim))
        self._f["get_obs"](self._h, _d(obs))
        return obs

    def obs_multi(self):
        obs = np.zeros((self.B, self.N, self.obs_dim_multi))
        self._f["obs_multi"](self._h, _d(obs))
        return obs

    def info(self, name):
        from windgym_amd.config import INFO
        f = INFO[name]
        per_turb = name in ("yaw_agent", "yaw_base", "ws_turb", "wd_turb", "power_turb_agent",
                            "power_turb_base", "ws_turb_base", "turb_x", "turb_y")
        if name.startswith("rotor_uvw"):
            out = np.zeros((self.B, self.N, 3))
        elif per_turb:
            out = np.zeros((self.B, self.N))
        else:
            out = np.zeros(self.B)
        rc = self._f["get_info"](self._h, C.c_int(f), _d(out))
        assert rc == 0
        return out

    def metrics(self, reset_after=False):
        out = np.zeros(8)
        self._f["metrics"](self._h, _d(out), C.c_int(int(reset_after)))
        return out

    def chain(self, b, farm, t):
        P = self.cfg.n_particles
        py, ue, ct = np.zeros(P), np.zeros(P), np.zeros(P)
        s = C.c_double()
        n = self._f["get_chain"](self._h, C.c_int(b), C.c_int(farm), C.c_int(t), _d(py), _d(ue), _d(ct),
                                 C.byref(s))
        return py[:n], ue[:n], ct[:n], s.value


def mann_noise(seed, Nxyz):
    """The complex white noise wg_generate_mann_box draws for ``seed`` (Philox stream of wg_mann.hip, restated in
    rng_api.c): complex64 [3, Nx, Ny, Nz], E|n|^2 = 1."""
    L = lib()
    cells = int(np.prod(Nxyz))
    out = np.empty((3, cells, 2), dtype=np.float32)
    L.wgo_mann_noise.argtypes = [C.c_uint64, C.c_uint64, C.c_void_p]
    L.wgo_mann_noise.restype = None
    L.wgo_mann_noise(C.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF), C.c_uint64(cells), out.ctypes.data_as(C.c_void_p))
    return (out[..., 0] + 1j * out[..., 1]).reshape((3,) + tuple(int(n) for n in Nxyz))
