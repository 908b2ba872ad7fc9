"""Eddy-viscosity (Ainslie-type) wake deficit of the Dynamic Wake Meandering model, tabulated on the host.

The reference builds its flow simulation with ``particleDeficitGenerator=jDWMAinslieGenerator()``
(WindGym/Wind_Farm_Env.py:706, :774) from DYNAMIKS, which is not in the reference tree.  This module restates the
quasi-steady deficit of the published DWM model — the thin-shear-layer equations of Ainslie (1988) with the two-term eddy
viscosity, filter functions and rotor boundary condition of Madsen, Larsen, Larsen, Troldborg & Mikkelsen (2010, J. Sol.
Energy Eng. 132) as calibrated by Larsen et al. (2013, Wind Energy 16) and adopted by IEC 61400-1 ed. 4, Annex E:

    U dU/dx + V dU/dr = (1/r) d/dr (nu_T r dU/dr),      (1/r) d(r V)/dr + dU/dx = 0          (lengths / R, speeds / U0)
    nu_T = F1(x) k1 TI_amb + F2(x) k2 (b / R) (1 - U_min),      k1 = 0.1, k2 = 0.008
    F1 = (x/8)^1.5 - sin(2 pi (x/8)^1.5) / (2 pi)  (x < 8), 1 beyond;
    F2 = 0.0625 (x < 4), 0.025 x - 0.0375 (4 <= x < 12), 0.00105 (x - 12)^3 + 0.025 x - 0.0375 (12 <= x < 20), 1 beyond
    inflow at the rotor plane: U = 1 - 2 a over r <= r_w, a = (1 - sqrt(1 - Ct)) / 2 (uniform induction),
    r_w = sqrt((1 - a) / (1 - 2 a)) (1 - 0.45 a^2)                                            (pressure-expanded stream tube)

**Constants and filter functions are restated from the publications as recalled; neither DYNAMIKS nor the papers are
available here — physics parity stays unpinned.**  What IS checked (tests/test_ainslie.py): the scheme
conserves the momentum-deficit flux of the inflow profile, reduces to the analytic self-similar decay for constant nu_T,
and the HIP kernels / the C oracle sample the same table.

``deficit_table`` solves the equations for a grid of (Ct, TI_amb) and returns the deficit fraction 1 - U on
(x / D, r / R) — the 4-D table the flow kernels sample per (wake particle, rotor point) for ``EnvConfig(deficit="ainslie")``.
"""
from __future__ import annotations

import hashlib
import os

import numpy as np

K1, K2 = 0.1, 0.008


def _f1(x):
    s = np.clip(x / 8.0, 0.0, 1.0) ** 1.5
    return np.where(x < 8.0, s - np.sin(2.0 * np.pi * s) / (2.0 * np.pi), 1.0)


def _f2(x):
    return np.where(x < 4.0, 0.0625,
                    np.where(x < 12.0, 0.025 * x - 0.0375,
                             np.where(x < 20.0, 0.00105 * (x - 12.0) ** 3 + 0.025 * x - 0.0375, 1.0)))


def _thomas(a, b, c, d):
    """Tridiagonal solve along the last axis for a batch of systems (a: sub-, b: main, c: super-diagonal; a[..., 0] and
    c[..., -1] are 0, so the batch is ONE long tridiagonal system for LAPACK)."""
    try:
        from scipy.linalg import solve_banded
        ab = np.zeros((3, b.size))
        ab[0, 1:] = c.reshape(-1)[:-1]
        ab[1] = b.reshape(-1)
        ab[2, :-1] = a.reshape(-1)[1:]
        return solve_banded((1, 1), ab, d.reshape(-1), overwrite_ab=True, check_finite=False).reshape(d.shape)
    except ImportError:
        pass
    n = b.shape[-1]
    cp = np.empty_like(b)
    dp = np.empty_like(d)
    cp[..., 0] = c[..., 0] / b[..., 0]
    dp[..., 0] = d[..., 0] / b[..., 0]
    for i in range(1, n):
        m = b[..., i] - a[..., i] * cp[..., i - 1]
        cp[..., i] = c[..., i] / m
        dp[..., i] = (d[..., i] - a[..., i] * dp[..., i - 1]) / m
    x = np.empty_like(d)
    x[..., -1] = dp[..., -1]
    for i in range(n - 2, -1, -1):
        x[..., i] = dp[..., i] - cp[..., i] * x[..., i + 1]
    return x


def solve(ct, ti, x_out, r_max=6.0, nr=240, dx=0.02, dx_growth=0.01, k1=K1, k2=K2, nu_const=None):
    """March the thin-shear-layer equations downstream for every (Ct, TI) pair of the broadcast arrays ``ct``, ``ti``.

    x_out: ascending distances in rotor RADII at which the profile is stored; marching step dx + dx_growth x.  Returns (r [nr] cell centres in R,
    deficit [..., len(x_out), nr] = 1 - U).  ``nu_const``: constant eddy viscosity instead of the DWM closure (tests)."""
    ct, ti = np.broadcast_arrays(np.asarray(ct, dtype=np.float64), np.asarray(ti, dtype=np.float64))
    shp = ct.shape
    ct, ti = ct.reshape(-1), ti.reshape(-1)
    nb = ct.size
    dr = r_max / nr
    r = (np.arange(nr) + 0.5) * dr                       # cell centres
    rf = np.arange(nr + 1) * dr                          # cell faces
    a = 0.5 * (1.0 - np.sqrt(1.0 - ct))
    rw = np.sqrt((1.0 - a) / (1.0 - 2.0 * a)) * (1.0 - 0.45 * a * a)
    # inflow profile: top hat of depth 2a over r <= r_w, the partly covered cell shares it by area (conserves the deficit)
    cover = np.clip((rw[:, None] ** 2 - rf[None, :-1] ** 2) / (rf[None, 1:] ** 2 - rf[None, :-1] ** 2), 0.0, 1.0)
    U = 1.0 - 2.0 * a[:, None] * cover
    x_out = np.asarray(x_out, dtype=np.float64)
    out = np.empty((nb, len(x_out), nr))
    k_out, x = 0, 0.0
    while k_out < len(x_out) and x_out[k_out] <= 1e-12:
        out[:, k_out] = 1.0 - U
        k_out += 1
    V = np.zeros_like(U)
    while k_out < len(x_out):
        h = min(dx + dx_growth * x, x_out[k_out] - x)     # the profile smooths downstream: the step grows with x
        # eddy viscosity of the current profile
        dfc = 1.0 - U
        dmax = dfc.max(axis=1)
        # wake half width b: the radius where the deficit has fallen to exp(-3.56) = 2.84 % of its maximum (Ainslie 1988)
        above = dfc >= 0.0284 * dmax[:, None]
        b = (np.where(above, r[None, :], 0.0).max(axis=1) + 0.5 * dr)
        xm = x + 0.5 * h
        nu = (_f1(xm) * k1 * ti + _f2(xm) * k2 * b * dmax) if nu_const is None else np.full(nb, float(nu_const))
        nu = np.maximum(nu, 1e-6)
        # implicit step: U_j (U_j' - U_j) / h + V_j (U_j+1' - U_j-1') / (2 dr) = nu / (r_j dr^2) [rf_j+1 (U_j+1' - U_j') - rf_j (U_j' - U_j-1')]
        lo = -V / (2.0 * dr) - nu[:, None] * rf[None, :-1] / (r[None, :] * dr * dr)
        up = V / (2.0 * dr) - nu[:, None] * rf[None, 1:] / (r[None, :] * dr * dr)
        di = U / h + nu[:, None] * (rf[None, 1:] + rf[None, :-1]) / (r[None, :] * dr * dr)
        rhs = U * U / h
        # axis: symmetry (rf_0 = 0 kills the lower flux; V_0 = 0); outer edge: U = 1 beyond the grid
        lo[:, 0] = 0.0
        rhs[:, -1] -= up[:, -1] * 1.0
        up[:, -1] = 0.0
        Un = _thomas(lo, di, up, rhs)
        # radial velocity from continuity: r V = - int_0^r r' dU/dx dr' (face values averaged to the centres)
        dUdx = (Un - U) / h
        cell = dUdx * r[None, :] * dr
        V = -(np.cumsum(cell, axis=1) - 0.5 * cell) / r[None, :]
        U = Un
        x += h
        if abs(x - x_out[k_out]) < 1e-9:
            out[:, k_out] = 1.0 - U
            k_out += 1
    return r, out.reshape(shp + (len(x_out), nr))


# the table the kernels sample (EnvConfig(deficit="ainslie")): nodes in Ct, TI, x / D and r / R
# (Ct uniform, TI log-uniform, x / D and r / R uniform: every axis is indexed arithmetically by the kernels)
TABLE_SPEC = dict(ct=np.linspace(0.04, 0.96, 24), ti=0.01 * (0.70 / 0.01) ** (np.arange(14) / 13.0),
                  n_x=96, x_max_D=28.5, n_r=48, r_max_R=4.7)
_TABLE = None


def _cache_dir_ok(cdir):
    """The cache directory exists, belongs to this user and nobody else can write to it."""
    try:
        st = os.stat(cdir)
    except OSError:
        return False
    return st.st_uid == os.getuid() and (st.st_mode & 0o022) == 0


def deficit_table():
    """(table float32 [n_ct, n_ti, n_x, n_r], spec) — deficit fraction 1 - U / U0 at x / D = i x_max / (n_x - 1), r / R =
    j r_max / (n_r - 1).  Solved once per process (a few seconds)."""
    global _TABLE
    if _TABLE is None:
        s = TABLE_SPEC
        # the solve takes a few seconds: the result is cached in a per-user directory (WINDGYM_AMD_CACHE overrides the place,
        # WINDGYM_AMD_NO_CACHE=1 disables it), keyed by this file's text and the numpy version.  The checksum stored with the
        # table detects truncation / corruption only (whoever can write the file can write a matching digest); what keeps
        # someone else from replacing the physics — the oracle would load the same file and parity would still pass — is the
        # directory: the cache is used only if it belongs to this user and is not writable by group / others (_cache_dir_ok).
        with open(__file__, "rb") as fh:
            key = hashlib.sha1(fh.read() + np.__version__.encode()).hexdigest()[:16]
        shape = (len(s["ct"]), len(s["ti"]), s["n_x"], s["n_r"])
        use_cache = os.environ.get("WINDGYM_AMD_NO_CACHE", "0") in ("", "0")
        cdir = os.environ.get("WINDGYM_AMD_CACHE") or os.path.join(os.path.expanduser("~"), ".cache", "windgym_amd")
        path = os.path.join(cdir, f"ainslie_{key}.npz")
        use_cache = use_cache and (_cache_dir_ok(cdir) or not os.path.exists(cdir))
        if use_cache:
            try:
                with np.load(path) as z:
                    tab, digest = z["table"], str(z["sha1"])
                if tab.shape == shape and tab.dtype == np.float32 and hashlib.sha1(tab.tobytes()).hexdigest() == digest:
                    _TABLE = (np.ascontiguousarray(tab), s)
                    return _TABLE
            except (OSError, ValueError, KeyError):
                pass
        xs = np.linspace(0.0, s["x_max_D"], s["n_x"]) * 2.0                 # in R
        r, dfc = solve(s["ct"][:, None], s["ti"][None, :], xs)
        rn = np.linspace(0.0, s["r_max_R"], s["n_r"])
        tab = np.empty(dfc.shape[:3] + (s["n_r"],), dtype=np.float32)
        for i in range(dfc.shape[0]):
            for j in range(dfc.shape[1]):
                for k in range(dfc.shape[2]):
                    tab[i, j, k] = np.interp(rn, r, dfc[i, j, k], left=dfc[i, j, k, 0])
        tab = np.ascontiguousarray(np.maximum(tab, 0.0))
        if use_cache:
            try:
                os.makedirs(cdir, mode=0o700, exist_ok=True)
                if not _cache_dir_ok(cdir):          # (an existing directory keeps its mode: makedirs does not enforce 0700)
                    raise OSError("cache directory is shared")
                tmp = f"{path}.{os.getpid()}.tmp.npz"
                np.savez(tmp, table=tab, sha1=np.array(hashlib.sha1(tab.tobytes()).hexdigest()))
                os.replace(tmp, path)
            except OSError:
                pass
        _TABLE = (tab, s)
    return _TABLE
