"""Site-based wind sampling — the `sample_site` option of WindFarmEnv (Wind_Farm_Env.py:569-594).

The reference asks a py_wake ``Site`` for the sector frequencies and Weibull parameters at 1-degree resolution
(``site.local_wind(x=0, y=0, wd=arange(360), ws=arange(3, 25))`` -> ``Sector_frequency_ilk``, ``Weibull_A_ilk``,
``Weibull_k_ilk``), draws a direction with those frequencies, a speed ``A * weibull(k)`` from that direction's
Weibull, clips both to the env's ranges and keeps TI uniform.  py_wake is not installable here, so any object with
that ``local_wind`` duck type works; ``WeibullSite`` is a minimal one (sector table, linear interpolation between
sector centres like py_wake's ``UniformWeibullSite``) and ``hornsrev1_site()`` restates py_wake's Horns Rev 1 wind
rose (12 sectors; values from recall of py_wake/examples/data/hornsrev1.py — re-verify when py_wake is at hand).
"""
from __future__ import annotations

import numpy as np


class _LocalWind:
    def __init__(self, f, A, k):
        self.Sector_frequency_ilk = f[None, :, None]
        self.Weibull_A_ilk = A[None, :, None]
        self.Weibull_k_ilk = k[None, :, None]


class WeibullSite:
    """Sector-wise Weibull wind rose: ``f`` (any positive weights), ``A`` [m/s], ``k`` per sector, sector i centred at
    i * 360 / n_sectors degrees."""

    def __init__(self, f, A, k):
        self.f = np.asarray(f, dtype=np.float64) / np.sum(f)
        self.A = np.asarray(A, dtype=np.float64)
        self.k = np.asarray(k, dtype=np.float64)
        assert self.f.shape == self.A.shape == self.k.shape and self.f.ndim == 1

    def _interp(self, v, wd):
        n = v.size
        pos = (np.asarray(wd, dtype=np.float64) % 360.0) / (360.0 / n)
        i0 = np.floor(pos).astype(int) % n
        w = pos - np.floor(pos)
        return v[i0] * (1 - w) + v[(i0 + 1) % n] * w

    def local_wind(self, x=0, y=0, wd=None, ws=None, **_):
        wd = np.arange(360) if wd is None else np.atleast_1d(wd)
        f = self._interp(self.f, wd)
        f = f / f.sum()                                   # frequency per requested direction bin
        return _LocalWind(f, self._interp(self.A, wd), self._interp(self.k, wd))


def hornsrev1_site() -> WeibullSite:
    f = [3.597152, 3.948682, 5.167395, 7.000154, 8.364547, 6.43485, 8.643194, 11.77051, 15.15757, 14.73792,
         10.01205, 5.165975]
    A = [9.176929, 9.782334, 9.531809, 9.909545, 10.04269, 9.593921, 9.584007, 10.51499, 11.39895, 11.68746,
         11.63732, 10.08803]
    k = [2.392578, 2.447266, 2.412109, 2.591797, 2.755859, 2.595703, 2.583984, 2.548828, 2.470703, 2.607422,
         2.626953, 2.326172]
    return WeibullSite(f, A, k)


def site_tables(site):
    """(dirs, freqs, As, ks) exactly as WindFarmEnv._set_windconditions reads them (:570-577)."""
    dirs = np.arange(0, 360, 1)
    ws = np.arange(3, 25, 1)
    lw = site.local_wind(x=0, y=0, wd=dirs, ws=ws)
    freqs = np.asarray(lw.Sector_frequency_ilk)[0, :, 0].astype(np.float64)
    As = np.asarray(lw.Weibull_A_ilk)[0, :, 0].astype(np.float64)
    ks = np.asarray(lw.Weibull_k_ilk)[0, :, 0].astype(np.float64)
    return dirs.astype(np.float64), freqs / freqs.sum(), As, ks


def sample_site(site, n, rng, wd_range=None, ws_range=None):
    """n draws of (wd, ws) like _sample_site (:586-594) + the clipping of :579-580.  ``rng``: numpy Generator."""
    dirs, freqs, As, ks = site_tables(site)
    idx = rng.choice(dirs.size, size=n, p=freqs)
    wd = dirs[idx]
    ws = As[idx] * rng.weibull(ks[idx])
    if wd_range is not None:
        wd = np.clip(wd, *wd_range)
    if ws_range is not None:
        ws = np.clip(ws, *ws_range)
    return wd, ws


class DeviceSiteSampler:
    """The same sampling for a whole batch on the device (torch ops on the current stream, no host sync): fills the
    f64[B, 3] override tensor of ``HipBatch.set_wind_device`` with fresh (ws, wd, NaN) rows."""

    def __init__(self, site, batch, wd_range, ws_range, seed=0):
        import torch
        self.t = torch
        dirs, freqs, As, ks = site_tables(site)
        dev = batch.device
        self.dirs = torch.as_tensor(dirs, device=dev)
        self.freqs = torch.as_tensor(freqs, device=dev)
        self.As = torch.as_tensor(As, device=dev)
        self.ks = torch.as_tensor(ks, device=dev)
        self.wd_range, self.ws_range = wd_range, ws_range
        self.gen = torch.Generator(device=dev).manual_seed(int(seed))
        self.buf = torch.full((batch.B, 3), float("nan"), dtype=torch.float64, device=dev)
        self.B = batch.B
        self.refresh()
        batch.set_wind_device(self.buf)

    def refresh(self):
        t = self.t
        idx = t.multinomial(self.freqs, self.B, replacement=True, generator=self.gen)
        u = t.rand(self.B, dtype=t.float64, device=self.buf.device, generator=self.gen)
        ws = self.As[idx] * (-t.log1p(-u)) ** (1.0 / self.ks[idx])
        self.buf[:, 0] = ws.clamp(self.ws_range[0], self.ws_range[1])
        self.buf[:, 1] = self.dirs[idx].clamp(self.wd_range[0], self.wd_range[1])
