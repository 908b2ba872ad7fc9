"""Row f2: the reference's on-disk turbulence format.  ``MannTurbulenceField.from_netcdf(filename=tf_file)``
(Wind_Farm_Env.py:616; file discovery :197-213) reads hipersim's NetCDF-4 = HDF5 files; windgym_amd/hdf5_min.py reads
them without netCDF4 / h5py.  The fixtures under tests/golden/hdf5/ were written by the real HDF5 library
(tests/golden/make_hdf5_fixtures.py, h5py 3.3 / libhdf5 1.10) in every container variant such a file can have."""
import glob
import os

import numpy as np
import pytest

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hdf5")
FILES = sorted(glob.glob(os.path.join(HERE, "*.nc")))


def test_fixture_set_is_complete():
    names = {os.path.basename(f) for f in FILES}
    assert {"TF_contiguous_f32.nc", "TF_chunked_deflate_f64.nc", "TF_chunked_plain_f32.nc", "TF_trackorder_v2.nc",
            "TF_latest_contiguous.nc", "TF_latest_fixed_array.nc", "TF_latest_fixed_array_deflate.nc",
            "TF_latest_single_chunk.nc", "TF_bigendian_uvw_last.nc"} <= names


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f) for f in FILES])
def test_reads_what_the_hdf5_library_wrote(path):
    from windgym_amd.hdf5_min import Hdf5File, read_turbulence_box
    exp = np.load(os.path.join(HERE, "expected.npz"))
    box, spacing = read_turbulence_box(path)
    assert box.dtype == np.float32 and box.shape == (3, 12, 6, 5)
    np.testing.assert_array_equal(box, exp["box"])                  # float32 files: bit-exact; float64 files: rounded once
    assert spacing == tuple(exp["dxyz"])
    f = Hdf5File(path)
    dss = f.datasets()
    for ax, n, d in zip("xyz", (12, 6, 5), exp["dxyz"]):
        np.testing.assert_array_equal(dss[ax].read(), np.arange(n) * d)
    if "f64" in path or "deflate" in path:
        v = [d for d in dss.values() if d.shape is not None and len(d.shape) == 4][0]
        assert v.dtype.itemsize == 8
        np.testing.assert_array_equal(v.read(), exp["box64"])       # float64 payload bit-exact


def test_format_variants_covered():
    """The fixtures really exercise the branches the reader claims: superblock 0 / 2 / 3, object headers 1 / 2, both
    group styles, layout versions 3 / 4, contiguous / chunked (B-tree v1, fixed array, single chunk), filters."""
    from windgym_amd.hdf5_min import Hdf5File
    seen = set()
    for path in FILES:
        f = Hdf5File(path)
        b = f.buf
        p = f._addr(f.root_addr)
        seen.add(("superblock", b[f.sb_off + 8]))
        seen.add(("object header", 2 if b[p:p + 4] == b"OHDR" else b[p]))
        seen.add(("group", "symtab" if f.root_btree else "links"))
        v = [d for d in f.datasets().values() if d.shape is not None and len(d.shape) == 4][0]
        seen.add(("layout", v.layout[0], v.layout[1]))
        if v.layout[0] == 4 and v.layout[1] == 2:
            q = 5 + v.layout[4] * v.layout[3]
            seen.add(("chunk index", v.layout[q]))
        for fid, _ in v.filters:
            seen.add(("filter", fid))
        seen.add(("byte order", v.dtype.byteorder if v.dtype.byteorder in "<>" else "<"))
    for want in [("superblock", 0), ("superblock", 2), ("superblock", 3), ("object header", 1), ("object header", 2),
                 ("group", "symtab"), ("group", "links"), ("layout", 3, 1), ("layout", 3, 2), ("layout", 4, 1),
                 ("layout", 4, 2), ("chunk index", 1), ("chunk index", 3), ("filter", 1), ("filter", 2),
                 ("byte order", ">")]:
        assert want in seen, want


def test_load_box_and_turbbox_directory(tmp_path):
    """WindFarmEnv(turbtype="MannLoad", TurbBox=<dir of TF_*.nc>) resolves and loads NetCDF-4 files (:197-213, :616)."""
    import shutil
    from windgym_amd.mann import find_box_files, load_box
    exp = np.load(os.path.join(HERE, "expected.npz"))
    d = tmp_path / "TurbBoxes"
    d.mkdir()
    for n in ("TF_contiguous_f32.nc", "TF_chunked_deflate_f64.nc", "TF_trackorder_v2.nc"):
        shutil.copy(os.path.join(HERE, n), d / n)
    files = find_box_files(str(d))
    assert len(files) == 3
    for f in files:
        box, dxyz = load_box(f)
        assert dxyz == (3.0, 2.5, 2.0) and box.shape == (3, 12, 6, 5)
        np.testing.assert_allclose(box, exp["box"] / exp["box"][0].std(), rtol=1e-6)
    # the env-level resolution (no GPU needed): a pool of the three boxes
    from windgym_amd import presets
    from windgym_amd.config import EnvConfig
    from windgym_amd.envs import _resolve_turbulence
    from windgym_amd.turbine import V80
    cfg = EnvConfig(turbine=V80(), yaml_dict=presets.env1_config(), turbtype="MannLoad", TurbBox=str(d), n_envs=2)
    kind, pool, spacing = _resolve_turbulence(cfg)
    assert kind == "pool" and len(pool) == 3 and tuple(spacing) == (3.0, 2.5, 2.0)


def test_unsupported_features_fail_loudly(tmp_path):
    from windgym_amd.hdf5_min import Hdf5File, Hdf5Unsupported
    p = tmp_path / "future.h5"
    p.write_bytes(b"\x89HDF\r\n\x1a\n" + bytes([9]) + b"\0" * 100)
    with pytest.raises(Hdf5Unsupported):
        Hdf5File(str(p))
    q = tmp_path / "not_hdf5.nc"
    q.write_bytes(b"CDF\x01" + b"\0" * 100)
    with pytest.raises(ValueError):
        Hdf5File(str(q))
