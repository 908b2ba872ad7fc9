"""Tabular wind-turbine model: the only part of py_wake the step() path needs.

The reference passes a py_wake ``WindTurbine`` (``V80()`` in tests/test_basics.py:4,17) and uses
``turbine.power(ws)`` (Wind_Farm_Env.py:112, :700), ``turbine.diameter()`` (:244) and ``hub_height()``
(:475); DYNAMIKS' ``PyWakeWindTurbines`` additionally interpolates the Ct curve.  py_wake is not
installable here, so the V80 table below is restated from py_wake's ``examples/data/hornsrev1.py``
(SURVEY.md Appendix D).  Any object exposing the same methods (including a real py_wake turbine) is
accepted by the env classes; :func:`as_tabular` extracts a table from it.
"""
from __future__ import annotations

import numpy as np


class TabularTurbine:
    """Linear-interpolated power / Ct curves; zero outside [ws[0], ws[-1]] (cut-in / cut-out)."""

    def __init__(self, name, diameter, hub_height, ws, power_w, ct):
        self.name = name
        self._d = float(diameter)
        self._h = float(hub_height)
        self.ws_tab = np.ascontiguousarray(ws, dtype=np.float64)
        self.power_tab = np.ascontiguousarray(power_w, dtype=np.float64)
        self.ct_tab = np.ascontiguousarray(ct, dtype=np.float64)
        assert self.ws_tab.shape == self.power_tab.shape == self.ct_tab.shape

    def diameter(self, *_, **__):
        return self._d

    def hub_height(self, *_, **__):
        return self._h

    def power(self, ws, **_):
        return np.interp(ws, self.ws_tab, self.power_tab, left=0.0, right=0.0)

    def ct(self, ws, **_):
        return np.interp(ws, self.ws_tab, self.ct_tab, left=0.0, right=0.0)


def V80():
    """Vestas V80-2MW as tabulated in py_wake's Horns Rev 1 example (D = 80 m, hub height 70 m)."""
    ws = np.arange(3.0, 26.0, 1.0)
    power_kw = [0, 66.6, 154, 282, 460, 696, 996, 1341, 1661, 1866, 1958, 1988, 1997, 1999] + [2000] * 9
    ct = [0, .818, .806, .804, .805, .806, .807, .793, .739, .709, .409, .314, .249, .202, .167, .140,
          .119, .102, .088, .077, .067, .060, .053]
    return TabularTurbine("V80", 80.0, 70.0, ws, 1e3 * np.asarray(power_kw, dtype=float), ct)


def as_tabular(turbine, ws_grid=None) -> TabularTurbine:
    """Accept a TabularTurbine or any py_wake-like object with power()/ct()/diameter()/hub_height()."""
    if isinstance(turbine, TabularTurbine):
        return turbine
    if ws_grid is None:
        ws_grid = np.arange(0.0, 40.5, 0.5)
    p = np.asarray(turbine.power(ws_grid), dtype=float)
    if hasattr(turbine, "ct"):
        ct = np.asarray(turbine.ct(ws_grid), dtype=float)
    else:
        raise ValueError("turbine object must expose ct(ws) (py_wake WindTurbine does)")
    return TabularTurbine(getattr(turbine, "name", "turbine"), turbine.diameter(), turbine.hub_height(),
                          ws_grid, p, ct)
