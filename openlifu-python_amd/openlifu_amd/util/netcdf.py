"""NetCDF persistence of the simulation result (the `.nc` half of ``Solution.to_files`` and the
base64 blob of ``Solution.to_json(include_simulation_data=True)``, plan/solution.py:411-533).

The reference writes the embedded blob with ``xarray.Dataset.to_netcdf(engine='scipy')`` — NetCDF-3
classic — and reads it back with ``xa.open_dataset(bytes, engine='scipy')`` (:424, :471-476); this
module writes and reads the same container through ``scipy.io.netcdf_file`` following xarray's
conventions (one dimension per coordinate, 1-D coordinate variables named after their dimension,
variable attributes ``units`` / ``long_name``, dataset attributes as global attributes), so a blob
made here opens in xarray and vice versa.  The reference's ``to_files`` uses ``engine='h5netcdf'``
(NetCDF-4 / HDF5, :507); h5py is not a dependency of this package, so ``write`` always produces
NetCDF-3 (xarray's ``open_dataset`` auto-detects it) and ``read`` refuses an HDF5 file by name.
NetCDF-3 has no 64-bit integers: integer coordinates such as ``focal_point_index`` are stored as
int32 and restored as int64 (xarray does the same on its scipy backend).
"""
from __future__ import annotations

import io
from collections import OrderedDict
from pathlib import Path

import numpy as np

from . import dataset as ds

_HDF5_MAGIC = b"\x89HDF\r\n\x1a\n"


def _nc3_array(a):
    a = np.asarray(a)
    if a.dtype.kind in "iu" and a.dtype.itemsize > 4 or a.dtype.kind == "b":
        if a.size and (a.min() < np.iinfo(np.int32).min or a.max() > np.iinfo(np.int32).max):
            raise ValueError("integer values do not fit NetCDF-3 int32")
        return a.astype(np.int32)
    if a.dtype.kind == "u":
        return a.astype(np.int32 if a.dtype.itemsize >= 2 else np.int16)
    if a.dtype.kind == "f" and a.dtype.itemsize not in (4, 8):
        return a.astype(np.float32)
    return a


def _set_attrs(obj, attrs):
    for k, v in attrs.items():
        if v is None:
            continue
        if isinstance(v, (bool, np.bool_)):
            v = int(v)
        setattr(obj, k, v)


def _write(f, dataset):
    from scipy.io import netcdf_file
    nc = netcdf_file(f, "w", version=2)
    sizes = OrderedDict()
    for name in dataset:
        da = dataset[name]
        for d, n in zip(da.dims, np.shape(da.data)):
            if sizes.setdefault(d, n) != n:
                raise ValueError(f"dimension {d!r} has conflicting sizes")
    for d in dataset.coords:
        c = dataset.coords[d]
        if np.ndim(c.data) == 1:
            sizes.setdefault(d, len(c.data))
    for d, n in sizes.items():
        nc.createDimension(d, int(n))
    for d in sizes:
        if d in dataset.coords:
            c = dataset.coords[d]
            a = _nc3_array(c.data)
            v = nc.createVariable(d, a.dtype, (d,))
            v[:] = a
            _set_attrs(v, dict(c.attrs))
    for name in dataset:
        da = dataset[name]
        a = _nc3_array(da.data)
        v = nc.createVariable(name, a.dtype, tuple(da.dims))
        if a.ndim == 0:
            v.assignValue(a[()])
        else:
            v[:] = a
        _set_attrs(v, dict(da.attrs))
    _set_attrs(nc, dict(dataset.attrs))
    nc.flush()
    return nc


def write(dataset, path) -> None:
    """Dataset -> NetCDF-3 file at ``path``."""
    nc = _write(str(path), dataset)
    nc.close()


def to_bytes(dataset) -> bytes:
    buf = io.BytesIO()
    nc = _write(buf, dataset)
    raw = buf.getvalue()
    nc.close()
    return raw


def _attrs_of(obj):
    out = {}
    for k, v in getattr(obj, "_attributes", {}).items():
        if isinstance(v, bytes):
            v = v.decode("utf-8")
        elif isinstance(v, np.ndarray) and v.size == 1:
            v = v.reshape(()).item()
        out[k] = v
    return out


def _native(a):
    a = np.array(a)  # own, writable copy (Solution.scale multiplies the volumes in place)
    if a.dtype.byteorder not in ("=", "|"):
        a = a.astype(a.dtype.newbyteorder("="))
    if a.dtype == np.int32:
        a = a.astype(np.int64)
    return a


def _read_hdf5(source):
    """NetCDF-4 / HDF5 file (what the reference's Solution.to_files writes with engine='h5netcdf', plan/solution.py:515) ->
    Dataset, through h5py when the host has it (this build's image does not: untested there, exercised by
    tests/test_host_api.py::test_netcdf4_files_read_when_h5py_is_present only where h5py imports).  Dimension names come from
    the HDF5 dimension scales h5netcdf / netCDF4 attach to every variable."""
    try:
        import h5py
    except ImportError as e:
        raise ValueError("simulation result is NetCDF-4/HDF5 (written with engine='h5netcdf'); reading it needs h5py, which this "
                         "host lacks -- re-save it with Dataset.to_netcdf(path, engine='scipy') (NetCDF-3), the format this build "
                         "writes and the reference's own JSON blob embeds") from e

    def attrs_of(obj):
        out = {}
        for k, v in obj.attrs.items():
            if k in ("DIMENSION_LIST", "REFERENCE_LIST", "CLASS", "NAME", "_Netcdf4Dimid", "_Netcdf4Coordinates", "_NCProperties"):
                continue
            if isinstance(v, bytes):
                v = v.decode("utf-8")
            elif isinstance(v, np.ndarray) and v.size == 1:
                v = v.reshape(()).item()
                if isinstance(v, bytes):
                    v = v.decode("utf-8")
            out[k] = v
        return out

    fh = io.BytesIO(bytes(source)) if isinstance(source, (bytes, bytearray, memoryview)) else source
    with h5py.File(fh, "r") as f:
        names = [k for k, v in f.items() if isinstance(v, h5py.Dataset)]

        def dims_of(name):
            v = f[name]
            out = []
            for i in range(v.ndim):
                scales = list(v.dims[i].values()) if len(v.dims[i]) else []
                out.append(scales[0].name.split("/")[-1] if scales else (name if v.ndim == 1 and v.attrs.get("CLASS") == b"DIMENSION_SCALE" else f"dim_{i}"))
            return tuple(out)

        dims = {n: dims_of(n) for n in names}
        coord_names = [n for n in names if dims[n] == (n,) and not str(f[n].attrs.get("NAME", b"")).startswith("b'This is a netCDF dimension but not")]
        vecs = {d: _native(f[d][()]) for d in coord_names}
        cattrs = {d: attrs_of(f[d]) for d in coord_names}
        data_vars = OrderedDict()
        for n in names:
            if n in coord_names or (f[n].ndim == 1 and dims[n] == (n,)):
                continue
            coords = ds.make_coords(OrderedDict((d, vecs[d]) for d in dims[n] if d in vecs), {d: cattrs[d] for d in dims[n] if d in vecs})
            data_vars[n] = ds.make_dataarray(_native(f[n][()]), coords, dims=dims[n], name=n, attrs=attrs_of(f[n]))
        return ds.make_dataset(data_vars, attrs=attrs_of(f))


def read(source):
    """NetCDF-3 file path, bytes or file object -> Dataset (all arrays loaded, nothing mapped).  NetCDF-4 / HDF5 input (the
    reference's on-disk format) is read through h5py where that is installed."""
    from scipy.io import netcdf_file
    if isinstance(source, (bytes, bytearray, memoryview)):
        raw = bytes(source)
        if raw[:8] == _HDF5_MAGIC:
            return _read_hdf5(raw)
        f = io.BytesIO(raw)
    else:
        p = Path(source) if isinstance(source, (str, Path)) else None
        if p is not None:
            with p.open("rb") as fh:
                if fh.read(8) == _HDF5_MAGIC:
                    return _read_hdf5(str(p))
            f = str(p)
        else:
            f = source
    nc = netcdf_file(f, "r", mmap=False)
    try:
        coord_names = [d for d in nc.dimensions if d in nc.variables and nc.variables[d].dimensions == (d,)]
        vecs = {d: _native(nc.variables[d].data) for d in coord_names}
        cattrs = {d: _attrs_of(nc.variables[d]) for d in coord_names}
        data_vars = OrderedDict()
        for name, v in nc.variables.items():
            if name in coord_names:
                continue
            coords = ds.make_coords(OrderedDict((d, vecs[d]) for d in v.dimensions if d in vecs),
                                    {d: cattrs[d] for d in v.dimensions if d in vecs})
            data_vars[name] = ds.make_dataarray(_native(v.data), coords, dims=v.dimensions, name=name,
                                                attrs=_attrs_of(v))
        return ds.make_dataset(data_vars, attrs=_attrs_of(nc))
    finally:
        nc.close()
