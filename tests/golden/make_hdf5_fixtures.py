#!/usr/bin/env python3
"""Fixture generator for windgym_amd/hdf5_min.py (row f2): small turbulence boxes written by the REAL HDF5 library
(h5py 3.3 / libhdf5 1.10, /opt/conda/bin/python3.9 in the build container — h5py is NOT available to the package or its
tests) in the layouts a hipersim / xarray / netCDF-4 file can have.  Data only: every file holds the array
`expected.npz` holds.  Run:  /opt/conda/bin/python3.9 tests/golden/make_hdf5_fixtures.py

Variable naming follows xarray's DataArray.to_netcdf (what hipersim's MannTurbulenceField.to_netcdf calls): dims
('uvw', 'x', 'y', 'z'), coordinate variables x, y, z (float64), the unnamed array as `__xarray_dataarray_variable__`,
dimension scales attached like the netCDF-4 library does.
"""
import os

import h5py
import numpy as np

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hdf5")
os.makedirs(HERE, exist_ok=True)
rng = np.random.default_rng(42)
Nx, Ny, Nz = 12, 6, 5
dx, dy, dz = 3.0, 2.5, 2.0
box = rng.standard_normal((3, Nx, Ny, Nz))
np.savez(os.path.join(HERE, "expected.npz"), box=box.astype(np.float32), box64=box, dxyz=np.array([dx, dy, dz]))


def write(name, dtype="f4", libver=None, track_order=False, var_kw=None, uvw_last=False, scales=True):
    kw = {}
    if libver:
        kw["libver"] = libver
    with h5py.File(os.path.join(HERE, name), "w", track_order=track_order, **kw) as f:
        f.attrs["_NCProperties"] = np.string_("version=2,netcdf=4.7.4,hdf5=1.10.6")
        x = f.create_dataset("x", data=np.arange(Nx) * dx)
        y = f.create_dataset("y", data=np.arange(Ny) * dy)
        z = f.create_dataset("z", data=np.arange(Nz) * dz)
        c = f.create_dataset("uvw", data=np.array([b"u", b"v", b"w"], dtype="S1"))
        data = np.moveaxis(box, 0, -1) if uvw_last else box
        v = f.create_dataset("__xarray_dataarray_variable__", data=data.astype(dtype), **(var_kw or {}))
        v.attrs["alphaepsilon"] = 1.0
        v.attrs["L"] = 33.6
        v.attrs["Gamma"] = 3.9
        if scales:
            dims = [x, y, z, c] if uvw_last else [c, x, y, z]
            for d in (x, y, z, c):
                d.make_scale(d.name.lstrip("/"))
            for i, d in enumerate(dims):
                v.dims[i].attach_scale(d)


# classic: superblock 0, symbol-table group, version-1 object headers, contiguous float32
write("TF_contiguous_f32.nc")
# float64, chunked + shuffle + deflate through a version-1 chunk B-tree (netCDF-4 with zlib=True)
write("TF_chunked_deflate_f64.nc", dtype="f8", var_kw=dict(chunks=(1, 5, 4, 5), compression="gzip", compression_opts=4, shuffle=True))
# chunked without filters, chunk grid that does not divide the shape (edge chunks)
write("TF_chunked_plain_f32.nc", var_kw=dict(chunks=(2, 5, 4, 3)))
# netCDF-4 style group: creation-order tracking -> link messages in a version-2 object header, superblock 2
write("TF_trackorder_v2.nc", libver=("v108", "v108"), track_order=True)
# newest format the library writes: superblock 3, layout message version 4 (single chunk / fixed array / implicit index)
write("TF_latest_contiguous.nc", libver="latest", track_order=True)
write("TF_latest_fixed_array.nc", libver="latest", track_order=True, var_kw=dict(chunks=(1, 4, 3, 5)))
write("TF_latest_fixed_array_deflate.nc", libver="latest", track_order=True, dtype="f8",
      var_kw=dict(chunks=(1, 4, 3, 5), compression="gzip", shuffle=True))
write("TF_latest_single_chunk.nc", libver="latest", var_kw=dict(chunks=(3, Nx, Ny, Nz), compression="gzip"))
# big-endian float32, component axis last, no dimension scales
with h5py.File(os.path.join(HERE, "TF_bigendian_uvw_last.nc"), "w") as f:
    f.create_dataset("x", data=np.arange(Nx) * dx)
    f.create_dataset("y", data=np.arange(Ny) * dy)
    f.create_dataset("z", data=np.arange(Nz) * dz)
    f.create_dataset("uvw_field", data=np.moveaxis(box, 0, -1).astype(">f4"))
for n in sorted(os.listdir(HERE)):
    print(n, os.path.getsize(os.path.join(HERE, n)))
