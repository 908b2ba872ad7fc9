"""Mann (1994/1998) spectral-tensor turbulence boxes — host-side generator (numpy FFT).

Stands in for ``MannTurbulenceField.generate(alphaepsilon, L, Gamma, Nxyz, dxyz, seed)`` of
dynamiks/hipersim (Wind_Farm_Env.py:624-637, :649-658; tests/test_basics.py:38-45), which are not
installable here.  The algorithm is the published one: an isotropic von Karman field with random complex
Gaussian amplitudes is distorted by rapid-distortion shear with the eddy-lifetime parameter Gamma and
transformed back with an inverse FFT.  The box is returned normalised to unit standard deviation of the u
component (the reference rescales with ``scale_TI(TI, U)``, :617/:637/:658 — the kernels multiply by TI*U per
env instead).  Layout: float32 [3, Nx, Ny, Nz], z fastest — what ``wg_set_turbulence_box`` expects.

One box is generated per process and shared by every env of the GPU (a per-env 0.8 GB box is not an option at
thousands of envs); episodes differ by a random horizontal offset (WG_TURB_BOX_SHIFT).  The product path generates
the box on the device (``generate_mann_box_hip`` -> wg_generate_mann_box: HIP spectral-tensor kernel + hipFFT, row f2 of
SURVEY.md §8); the numpy version below is its restatement for tests and for the CPU oracle.
"""
from __future__ import annotations

import numpy as np


def _eddy_lifetime_beta(kL, Gamma):
    if Gamma == 0:                       # isotropic field: no shear distortion
        return np.zeros_like(kL)
    from scipy.special import hyp2f1
    kL = np.maximum(kL, 1e-12)
    return Gamma * kL ** (-2.0 / 3.0) / np.sqrt(hyp2f1(1.0 / 3.0, 17.0 / 6.0, 4.0 / 3.0, -kL ** (-2.0)))


BETA_TABLE = dict(n=4096, log10_lo=-6.0, log10_hi=6.0)    # the table wg_generate_mann_box interpolates (wg_mann.hip)


def _beta_from_table(kL, table):
    """beta(kL) by linear interpolation in log10(kL) of a log-spaced table — what the HIP kernel does (float64 here)."""
    lo, hi, n = BETA_TABLE["log10_lo"], BETA_TABLE["log10_hi"], len(table)
    lk = np.log10(np.clip(kL, 10.0 ** lo, 10.0 ** hi))
    pos = (lk - lo) * ((n - 1) / (hi - lo))
    i0 = np.clip(np.floor(pos).astype(np.int64), 0, n - 2)
    w = pos - i0
    return table[i0] * (1.0 - w) + table[i0 + 1] * w


def mann_field_from_noise(noise, dxyz=(3.0, 3.0, 3.0), alphaepsilon=0.1, L=33.6, Gamma=3.9, beta_table=None):
    """The spectral part of the generator for GIVEN complex white noise ``noise`` [3, Nx, Ny, Nz] (E|n|^2 = 1): sheared
    von Karman tensor times the noise, inverse FFT, unit standard deviation of u.  float64 numpy — the CPU restatement
    the HIP generator (``generate_mann_box_hip(noise=...)``) is pinned against cell by cell (tests/test_mann_generator.py).
    ``beta_table``: interpolate the eddy lifetime in this table (as the kernel does) instead of evaluating 2F1 per cell."""
    noise = np.asarray(noise)
    _, Nx, Ny, Nz = noise.shape
    dx, dy, dz = (float(d) for d in dxyz)
    k1 = 2 * np.pi * np.fft.fftfreq(Nx, dx)[:, None, None]
    k2 = 2 * np.pi * np.fft.fftfreq(Ny, dy)[None, :, None]
    k3 = 2 * np.pi * np.fft.fftfreq(Nz, dz)[None, None, :]
    k1 = np.broadcast_to(k1, (Nx, Ny, Nz)).astype(np.float64)
    k2 = np.broadcast_to(k2, (Nx, Ny, Nz)).astype(np.float64)
    k3 = np.broadcast_to(k3, (Nx, Ny, Nz)).astype(np.float64)
    kk = np.sqrt(k1 ** 2 + k2 ** 2 + k3 ** 2)
    beta = _eddy_lifetime_beta(kk * L, Gamma) if beta_table is None else _beta_from_table(kk * L, np.asarray(beta_table, np.float64))
    k30 = k3 + beta * k1
    k0 = np.sqrt(k1 ** 2 + k2 ** 2 + k30 ** 2)
    k0 = np.where(k0 == 0, 1e-12, k0)
    kk_s = np.where(kk == 0, 1e-12, kk)
    # von Karman energy spectrum at the undistorted wave number
    E0 = alphaepsilon * L ** (5.0 / 3.0) * (k0 * L) ** 4 / (1.0 + (k0 * L) ** 2) ** (17.0 / 6.0)
    amp = np.sqrt(E0 / (4.0 * np.pi)) / k0 ** 2
    # rapid-distortion coefficients (Mann 1998, eqs. 3.16-3.18)
    k12 = k1 ** 2 + k2 ** 2
    k12s = np.where(k12 == 0, 1e-12, k12)
    C1 = beta * k1 ** 2 * (k0 ** 2 - 2 * k30 ** 2 + beta * k1 * k30) / (kk_s ** 2 * k12s)
    C2 = k2 * k0 ** 2 / k12s ** 1.5 * np.arctan2(beta * k1 * np.sqrt(k12s), k0 ** 2 - k30 * k1 * beta)
    k1s = np.where(k1 == 0, 1e-12, k1)
    zeta1 = C1 - k2 / k1s * C2
    zeta2 = k2 / k1s * C1 + C2
    zeta1 = np.where(k1 == 0, -beta, zeta1)
    zeta2 = np.where(k1 == 0, 0.0, zeta2)
    n = noise
    # isotropic incompressible field dZ_iso = amp * (k0 x n), then sheared
    a1 = amp * (k2 * n[2] - k30 * n[1])
    a2 = amp * (k30 * n[0] - k1 * n[2])
    a3 = amp * (k1 * n[1] - k2 * n[0])
    r = (k0 / kk_s) ** 2
    dZ1 = a1 + zeta1 * a3
    dZ2 = a2 + zeta2 * a3
    dZ3 = r * a3
    dV = (2 * np.pi) ** 3 / (Nx * dx * Ny * dy * Nz * dz)
    out = np.empty((3, Nx, Ny, Nz), dtype=np.float64)
    for c, dZ in enumerate((dZ1, dZ2, dZ3)):
        dZ = dZ.copy()
        dZ[0, 0, 0] = 0.0
        out[c] = (np.fft.ifftn(dZ) * (Nx * Ny * Nz) * np.sqrt(dV)).real
    out /= float(out[0].std())
    return out


def generate_mann_box(Nxyz=(1024, 128, 32), dxyz=(3.0, 3.0, 3.0), alphaepsilon=0.1, L=33.6, Gamma=3.9, seed=1234):
    Nx, Ny, Nz = (int(n) for n in Nxyz)
    rng = np.random.default_rng(seed)
    # random complex Gaussian white noise
    n = (rng.standard_normal((3, Nx, Ny, Nz)) + 1j * rng.standard_normal((3, Nx, Ny, Nz))) / np.sqrt(2.0)
    return np.ascontiguousarray(mann_field_from_noise(n, dxyz, alphaepsilon, L, Gamma).astype(np.float32))


def mann_beta_table(Gamma, n=None, log10_lo=None, log10_hi=None):
    """The eddy-lifetime table of the HIP generator (wg_mann_beta_table: 2F1 by adaptive quadrature of its Euler integral,
    host code of libwindgym_hip.so — no GPU needed)."""
    import ctypes as C
    from .binding import _chk, load_library
    n = n or BETA_TABLE["n"]
    lo = BETA_TABLE["log10_lo"] if log10_lo is None else log10_lo
    hi = BETA_TABLE["log10_hi"] if log10_hi is None else log10_hi
    out = np.empty(n, dtype=np.float64)
    _chk(load_library().wg_mann_beta_table(float(Gamma), int(n), float(lo), float(hi), out.ctypes.data_as(C.POINTER(C.c_double))),
         "wg_mann_beta_table")
    return out


def generate_mann_box_hip(Nxyz=(2048, 512, 64), dxyz=(3.0, 3.0, 3.0), alphaepsilon=0.1, L=33.6, Gamma=3.9, seed=1234,
                          device="cuda", noise=None):
    """The box generated on the MI355X by libwindgym_hip.so (wg_generate_mann_box: spectral-tensor kernel + hipFFT; the
    reference's sizes — 2048 x 512 x 64 = 0.8 GB, Wind_Farm_Env.py:654; 4096 x 512 x 64, :629-633 — take a fraction of a
    second and never leave the device).  ``noise``: optional complex white noise [3, Nx, Ny, Nz] (numpy complex or a CUDA
    float tensor [3, Nx, Ny, Nz, 2]) used instead of the built-in Philox stream keyed by ``seed``.
    Returns a float32 CUDA tensor [3, Nx, Ny, Nz], unit std of u."""
    import ctypes as C
    import torch
    from .binding import _chk, load_library
    L_ = load_library()
    if not torch.cuda.is_available():
        raise RuntimeError("generate_mann_box_hip needs a HIP device (the numpy restatement is generate_mann_box)")
    Nx, Ny, Nz = (int(n) for n in Nxyz)
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    out = torch.empty((3, Nx, Ny, Nz), dtype=torch.float32, device=torch.device("cuda", idx))
    nz = None
    if noise is not None:
        if not isinstance(noise, torch.Tensor):
            noise = np.asarray(noise)
            noise = torch.from_numpy(np.stack([noise.real, noise.imag], axis=-1).astype(np.float32))
        nz = noise.to(device=out.device, dtype=torch.float32).contiguous()
        assert tuple(nz.shape) == (3, Nx, Ny, Nz, 2)
    st = torch.cuda.current_stream(out.device).cuda_stream
    _chk(L_.wg_generate_mann_box(idx, C.c_void_p(out.data_ptr()), Nx, Ny, Nz, float(dxyz[0]), float(dxyz[1]), float(dxyz[2]),
                                 float(alphaepsilon), float(L), float(Gamma), C.c_uint64(int(seed) & 0xFFFFFFFFFFFFFFFF),
                                 C.c_void_p(nz.data_ptr()) if nz is not None else None, C.c_void_p(st)), "wg_generate_mann_box")
    return out


generate_mann_box_torch = generate_mann_box_hip      # (rounds 1-3 generated the box with torch.fft; kept as a name only)


# Wake-added turbulence (row a7): the isotropic small-scale box the reference's addedTurbulenceModel =
# [Synchronized]AutoScalingIsotropicMannTurbulence() scales inside the wakes (Wind_Farm_Env.py:618, :638, :644, :659).
# Its default field is hipersim's "Hipersim_mann_l5.0_ae1.0000_g0.0_h0_128x128x128_3.000x3.00x3.00_s0001.nc"
# (examples/longer_steps_example.py:153): L = 5 m, Gamma = 0 (isotropic), alpha-epsilon 1, 128^3 cells of 3 m, seed 1 —
# regenerated here with the same parameters (hipersim's own random stream is not reproducible without hipersim), unit
# variance.  One array per process, shared by the HIP batch and the oracle so that both see the same field.
ADDED_BOX_SPEC = dict(Nxyz=(128, 128, 128), dxyz=(3.0, 3.0, 3.0), alphaepsilon=1.0, L=5.0, Gamma=0.0, seed=1)
_ADDED_BOX = None


def default_added_box():
    """(box float32 [3, 128, 128, 128], spacing) of the wake-added isotropic turbulence."""
    global _ADDED_BOX
    if _ADDED_BOX is None:
        _ADDED_BOX = generate_mann_box(**ADDED_BOX_SPEC)
    return _ADDED_BOX, ADDED_BOX_SPEC["dxyz"]


# the reference's box definitions per turbtype (Wind_Farm_Env.py:624-637, :649-658)
def reference_box_spec(turbtype: str, D: float):
    if turbtype == "MannFixed":
        return dict(Nxyz=(2048, 512, 64), dxyz=(3.0, 3.0, 3.0), seed=1234)
    return dict(Nxyz=(4096, 512, 64), dxyz=(D / 20, D / 10, D / 10), seed=1234)


# ---------------------------------------------------------------------------------------------------
# on-disk boxes (row f2): what `TurbBox=` may point to for turbtype "MannLoad" (Wind_Farm_Env.py:197-213, :611-618)
# ---------------------------------------------------------------------------------------------------
def save_box(path, box, dxyz):
    """Write a box [3, Nx, Ny, Nz] with its spacing as .npz (keys uvw, dxyz)."""
    np.savez_compressed(path, uvw=np.asarray(box, dtype=np.float32), dxyz=np.asarray(dxyz, dtype=np.float64))


def _dims_from_name(name):
    """hipersim-style file names carry the grid: ..._128x128x128_3.000x3.00x3.00_s0001.nc"""
    import re
    m = re.search(r"_(\d+)x(\d+)x(\d+)_(\d+(?:\.\d+)?)x(\d+(?:\.\d+)?)x(\d+(?:\.\d+)?)", name)
    if not m:
        return None, None
    return tuple(int(m.group(i)) for i in (1, 2, 3)), tuple(float(m.group(i)) for i in (4, 5, 6))


def _is_hdf5(path):
    with open(path, "rb") as f:
        return f.read(8) == b"\x89HDF\r\n\x1a\n"


def load_box(path, dxyz=None, Nxyz=None):
    """Read a turbulence box from disk -> (float32 [3, Nx, Ny, Nz] normalised to unit std of u, (dx, dy, dz)).

    * ``.npz`` written by :func:`save_box` (or with keys u, v, w + dxyz);
    * ``.npy`` [3, Nx, Ny, Nz] (spacing from ``dxyz`` or from a hipersim-style file name);
    * raw float32 binaries (HAWC2 / Mann-generator convention): one file holding u, v, w one after the other, or the
      ``*_u.bin`` of a ``_u / _v / _w`` triplet; grid from ``Nxyz`` / ``dxyz`` or from the file name;
    * netCDF: NetCDF-4 / HDF5 containers (hipersim's ``to_netcdf``: the reference's ``TF_*.nc`` files) through the
      package's own minimal HDF5 reader (:mod:`windgym_amd.hdf5_min`: contiguous / chunked, deflate + shuffle), classic
      NetCDF3 through ``scipy.io.netcdf_file``.
    """
    import os
    name = os.path.basename(path)
    ext = os.path.splitext(name)[1].lower()
    n_name, d_name = _dims_from_name(name)
    Nxyz = Nxyz or n_name
    dxyz = dxyz or d_name
    if ext == ".npz":
        z = np.load(path)
        box = z["uvw"] if "uvw" in z else np.stack([z["u"], z["v"], z["w"]])
        dxyz = tuple(float(v) for v in z["dxyz"]) if "dxyz" in z else dxyz
    elif ext == ".npy":
        box = np.load(path)
    elif ext in (".bin", ".dat", ".raw"):
        if Nxyz is None:
            raise ValueError("raw box: pass Nxyz (and dxyz) or use a file name that carries NxXNyXNz_dxXdyXdz")
        n = int(np.prod(Nxyz))
        if name.endswith("_u" + ext):
            box = np.stack([np.fromfile(path[: -len("_u" + ext)] + f"_{c}{ext}", dtype=np.float32, count=n).reshape(Nxyz)
                            for c in "uvw"])
        else:
            box = np.fromfile(path, dtype=np.float32, count=3 * n).reshape((3,) + tuple(Nxyz))
    elif ext in (".nc", ".nc4", ".cdf", ".h5", ".hdf5") and _is_hdf5(path):
        # NetCDF-4 = HDF5 container: what hipersim's MannTurbulenceField.to_netcdf writes and the reference loads
        # (Wind_Farm_Env.py:616) — read with the package's own minimal HDF5 reader (hdf5_min.py)
        from .hdf5_min import read_turbulence_box
        box, sp = read_turbulence_box(path)
        dxyz = sp or dxyz
    elif ext in (".nc", ".cdf"):
        from scipy.io import netcdf_file
        f = netcdf_file(path, "r", mmap=False)
        keys = [k for k in ("uvw", "turb", "u") if k in f.variables]
        if "uvw" in f.variables or "turb" in f.variables:
            box = np.array(f.variables[keys[0]][:], dtype=np.float32)
        else:
            box = np.stack([np.array(f.variables[c][:], dtype=np.float32) for c in "uvw"])
        if dxyz is None and all(c in f.variables for c in "xyz"):
            dxyz = tuple(float(f.variables[c][1] - f.variables[c][0]) for c in "xyz")
        f.close()
    else:
        raise ValueError(f"unknown turbulence box format: {name}")
    box = np.ascontiguousarray(box, dtype=np.float32)
    if box.ndim != 4 or box.shape[0] != 3:
        raise ValueError(f"{name}: expected [3, Nx, Ny, Nz], got {box.shape}")
    if dxyz is None:
        raise ValueError(f"{name}: grid spacing unknown (pass dxyz)")
    box = box / float(box[0].std())
    return box, tuple(float(v) for v in dxyz)


def find_box_files(turb_box):
    """The reference's TurbBox rule (:197-213): a file is used as it is, a directory contributes its ``TF_*`` files."""
    import os
    if turb_box is None or not isinstance(turb_box, (str, os.PathLike)):
        return []
    if os.path.isfile(turb_box):
        return [str(turb_box)]
    if os.path.isdir(turb_box):
        return sorted(os.path.join(turb_box, f) for f in os.listdir(turb_box) if f.split("_")[0] == "TF")
    return []
