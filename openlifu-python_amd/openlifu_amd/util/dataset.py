"""Labelled-array containers for the hot path's inputs and outputs.

The reference passes ``xarray`` objects across the seams this package plugs into
(``params`` Dataset in, ``Dataset{p_max, p_min, intensity}`` out,
sim/kwave_if.py:131-146; plan/protocol.py:341-347).  xarray is not part of this
image, so a minimal stand-in with the attribute surface those call sites use is
provided; when xarray is importable (and ``OPENLIFU_AMD_USE_XARRAY`` is not "0")
the factory functions return real xarray objects instead.

Only what the path touches is implemented: ``.data .dims .coords .attrs .sizes
.shape``, name lookup, integer indexing that returns WRITABLE VIEWS (Solution.scale
multiplies ``simulation_result['p_min'][i].data`` in place, plan/solution.py:332-337),
``max/mean(dim=...)``, ``isel``, ``assign_coords``, ``drop_dims``, ``copy``.
"""
from __future__ import annotations

import os
from collections import OrderedDict

import numpy as np

try:  # pragma: no cover - xarray is absent in the build image
    if os.environ.get("OPENLIFU_AMD_USE_XARRAY", "1") == "0":
        raise ImportError
    import xarray as _xa
    HAVE_XARRAY = True
except ImportError:
    _xa = None
    HAVE_XARRAY = False


class DataArray:
    def __init__(self, data, coords=None, dims=None, name=None, attrs=None):
        self.data = np.asarray(data)
        self._init_labels(self.data.ndim, coords, dims, name, attrs)

    def _init_labels(self, ndim, coords, dims, name, attrs):
        if dims is None:
            dims = tuple(coords.keys())[:ndim] if coords is not None else tuple(f"dim_{i}" for i in range(ndim))
        self.dims = tuple(dims)
        if len(self.dims) != ndim:
            raise ValueError(f"dims {self.dims} do not match data of rank {ndim}")
        self.coords = OrderedDict()
        if coords is not None:
            for k, v in coords.items():
                self.coords[k] = v if isinstance(v, DataArray) else DataArray(np.asarray(v), dims=(k,), name=k)
        self.name = name
        self.attrs = dict(attrs) if attrs else {}

    # numpy-ish surface
    @property
    def shape(self):
        return self.data.shape

    @property
    def values(self):
        return self.data

    @property
    def sizes(self):
        return OrderedDict(zip(self.dims, self.shape))

    @property
    def ndim(self):
        return len(self.shape)

    @property
    def size(self):
        return int(np.prod(self.shape, dtype=np.int64))

    def __len__(self):
        return self.shape[0]

    def __array__(self, dtype=None, copy=None):
        return np.asarray(self.data, dtype=dtype)

    def __iter__(self):
        return iter(self.data)

    def __float__(self):
        return float(self.data)

    def item(self):
        return self.data.item()

    def to_numpy(self):
        return self.data

    def __getitem__(self, key):
        """Positional indexing along the leading axes; returns a VIEW-backed DataArray."""
        if isinstance(key, str):
            return self.coords[key]
        sub = self.data[key]
        keys = key if isinstance(key, tuple) else (key,)
        dims = []
        for i, d in enumerate(self.dims):
            if i < len(keys) and isinstance(keys[i], (int, np.integer)):
                continue
            dims.append(d)
        coords = OrderedDict()
        for i, d in enumerate(self.dims):
            if d in self.coords and d in dims:
                c = self.coords[d]
                coords[d] = DataArray(c.data[keys[i]] if i < len(keys) else c.data, dims=(d,), name=d, attrs=c.attrs)
        return DataArray(sub, coords=coords, dims=tuple(dims), name=self.name, attrs=self.attrs)

    def isel(self, **indexers):
        key = tuple(indexers.get(d, slice(None)) for d in self.dims)
        return self[key]

    def _reduce(self, fn, dim, keep_attrs):
        if dim is None:
            return DataArray(fn(self.data), dims=(), name=self.name, attrs=self.attrs if keep_attrs else None)
        ax = self.dims.index(dim)
        dims = tuple(d for d in self.dims if d != dim)
        coords = OrderedDict((k, v) for k, v in self.coords.items() if k != dim)
        return DataArray(fn(self.data, axis=ax), coords=coords, dims=dims, name=self.name,
                         attrs=self.attrs if keep_attrs else None)

    def max(self, dim=None, keep_attrs=False):
        return self._reduce(np.max, dim, keep_attrs)

    def min(self, dim=None, keep_attrs=False):
        return self._reduce(np.min, dim, keep_attrs)

    def mean(self, dim=None, keep_attrs=False):
        return self._reduce(np.mean, dim, keep_attrs)

    def copy(self, deep=True):
        return DataArray(self.data.copy() if deep else self.data,
                         coords=OrderedDict((k, v if k == self.name and v is self else
                                             DataArray(v.data.copy() if deep else v.data, dims=v.dims, name=v.name, attrs=v.attrs))
                                            for k, v in self.coords.items()),
                         dims=self.dims, name=self.name, attrs=dict(self.attrs))

    def assign_coords(self, **kw):
        out = self.copy(deep=False)
        for k, v in kw.items():
            out.coords[k] = DataArray(np.asarray(v), dims=() if np.ndim(v) == 0 else (k,), name=k)
        return out

    def __repr__(self):
        return f"<openlifu_amd.DataArray {self.name!r} dims={self.dims} shape={self.shape} attrs={list(self.attrs)}>"


class LazyDataArray(DataArray):
    """A DataArray whose values still live in HBM: shape / dims / coords / attrs are known, ``.data`` brings the values
    to the host on FIRST access (``fetch()`` -> a fresh, writable, caller-owned ndarray) and is an ordinary NumPy array
    from then on.  ``Protocol.calc_solution`` returns the per-focus volumes this way: scaling, aggregation and analysis
    run on the device, so a caller that only looks at the analysis never pays for 3 x F volumes of PCIe traffic
    (DESIGN.md section 6).  ``materialized`` tells whether the host copy exists (and may have been edited)."""

    def __init__(self, shape, dtype, fetch, coords=None, dims=None, name=None, attrs=None, uniform_value=None):
        self._shape = tuple(int(v) for v in shape)
        self._dtype = np.dtype(dtype)
        self._fetch = fetch
        self._host = None
        self._uniform = uniform_value
        self._on_materialize = []      # callables run once, right after the host copy exists and before the reader can edit it (snapshots taken from this array)
        self._init_labels(len(self._shape), coords, dims, name, attrs)

    @property
    def materialized(self) -> bool:
        return self._host is not None

    @property
    def uniform_value(self):
        """The single value every entry holds, for arrays declared constant (``uniform``) that nobody has read -- and
        thereby possibly edited -- yet; None otherwise."""
        return self._uniform if self._host is None else None

    @classmethod
    def uniform(cls, shape, value, coords=None, dims=None, name=None, attrs=None, dtype=np.float64):
        """A constant volume that is only allocated if somebody reads it (the parameter volumes of a uniform medium:
        5 x 134 MB of identical numbers at 256^3, seg/seg_method.py:84-115)."""
        shape = tuple(int(v) for v in shape)
        return cls(shape, dtype, lambda: np.full(shape, value, dtype=dtype), coords=coords, dims=dims, name=name,
                   attrs=attrs, uniform_value=value)

    @property
    def data(self):
        if self._host is None:
            host = np.asarray(self._fetch())
            if host.shape != self._shape:
                raise ValueError(f"device result of shape {host.shape}, expected {self._shape}")
            self._host, self._fetch = host, None
            hooks, self._on_materialize = self._on_materialize, []
            for h in hooks:
                h()
        return self._host

    @data.setter
    def data(self, value):
        self._host, self._fetch = np.asarray(value), None
        self._shape = self._host.shape

    @property
    def shape(self):
        return self._shape

    @property
    def dtype(self):
        return self._dtype

    # copies and pickles must not share the fetch closure: the device result behind it is retired (and unreadable) after the
    # next launch or upload, and only arrays registered with it are rescued to the host first.  A copy therefore takes the
    # values NOW (one device read, which also materialises this array); declared-constant volumes stay lazy.
    def _detached(self, memo=None):
        import copy as _copy
        kw = dict(coords=_copy.deepcopy(self.coords, memo), dims=self.dims, name=self.name, attrs=_copy.deepcopy(self.attrs, memo))
        if self._host is None and self._uniform is not None:
            return LazyDataArray.uniform(self._shape, self._uniform, dtype=self._dtype, **kw)
        return DataArray(np.array(self.data, copy=True), **kw)

    def __deepcopy__(self, memo):
        return self._detached(memo)

    def __copy__(self):
        return self._detached()

    def __reduce__(self):
        if self._host is None and self._uniform is not None:
            return (_rebuild_uniform, (self._shape, self._uniform, self._dtype.str, dict(self.coords), self.dims, self.name, self.attrs))
        return (DataArray, (np.array(self.data, copy=True), dict(self.coords), self.dims, self.name, self.attrs))

    def __repr__(self):
        where = "host" if self.materialized else "device"
        return f"<openlifu_amd.LazyDataArray {self.name!r} dims={self.dims} shape={self.shape} on {where}>"


def _rebuild_uniform(shape, value, dtype, coords, dims, name, attrs):
    return LazyDataArray.uniform(shape, value, coords=coords, dims=dims, name=name, attrs=attrs, dtype=np.dtype(dtype))


class Coordinates(OrderedDict):
    """Mapping dim -> 1-D coordinate DataArray (stand-in for xarray.Coordinates)."""

    def __init__(self, coords=None):
        super().__init__()
        for k, v in (coords or {}).items():
            self[k] = v if isinstance(v, DataArray) else DataArray(np.asarray(v), dims=(k,), name=k)

    @property
    def dims(self):
        return tuple(k for k, v in self.items() if v.ndim == 1)

    @property
    def sizes(self):
        return OrderedDict((k, v.shape[0]) for k, v in self.items() if v.ndim == 1)


class Dataset:
    def __init__(self, data_vars=None, coords=None, attrs=None):
        self.data_vars = OrderedDict()
        self.coords = Coordinates(coords)
        self.attrs = dict(attrs) if attrs else {}
        for k, v in (data_vars or {}).items():
            self[k] = v

    def __setitem__(self, name, da):
        if not isinstance(da, DataArray):
            raise TypeError("Dataset values must be DataArray")
        da.name = name
        for k, c in da.coords.items():
            if k not in self.coords:
                self.coords[k] = c
        self.data_vars[name] = da

    def __getitem__(self, name):
        if name in self.data_vars:
            return self.data_vars[name]
        return self.coords[name]

    def __contains__(self, name):
        return name in self.data_vars or name in self.coords

    def __iter__(self):
        return iter(self.data_vars)

    def keys(self):
        return self.data_vars.keys()

    def values(self):
        return self.data_vars.values()

    def items(self):
        return self.data_vars.items()

    def __len__(self):
        return len(self.data_vars)

    @property
    def dims(self):
        seen = []
        for da in self.data_vars.values():
            for d in da.dims:
                if d not in seen:
                    seen.append(d)
        if not seen:
            seen = list(self.coords.dims)
        return tuple(seen)

    @property
    def sizes(self):
        out = OrderedDict()
        for da in self.data_vars.values():
            out.update(da.sizes)
        if not out:
            out.update(self.coords.sizes)
        return out

    def isel(self, **indexers):
        return Dataset({k: v.isel(**{d: i for d, i in indexers.items() if d in v.dims}) for k, v in self.items()},
                       attrs=self.attrs)

    def assign_coords(self, **kw):
        out = Dataset({k: v for k, v in self.items()}, coords=self.coords, attrs=self.attrs)
        for k, v in kw.items():
            out.coords[k] = DataArray(np.asarray(v), dims=() if np.ndim(v) == 0 else (k,), name=k)
        return out

    def drop_dims(self, dim):
        keep = OrderedDict((k, v) for k, v in self.items() if dim not in v.dims)
        coords = OrderedDict((k, v) for k, v in self.coords.items() if k != dim)
        return Dataset(keep, coords=coords, attrs=self.attrs)

    def copy(self, deep=True):
        return Dataset({k: v.copy(deep=deep) for k, v in self.items()}, coords=self.coords, attrs=self.attrs)

    def __repr__(self):
        return f"<openlifu_amd.Dataset vars={list(self.data_vars)} dims={dict(self.sizes)}>"


# ---- factories: real xarray when present, the stand-in otherwise -------------------------------
def make_coords(vectors: dict, attrs: dict):
    """vectors: dim -> 1-D values; attrs: dim -> attrs dict (units, long_name)."""
    if HAVE_XARRAY:  # pragma: no cover
        c = _xa.Coordinates({d: np.asarray(v) for d, v in vectors.items()})
        for d in vectors:
            c[d].attrs.update(attrs.get(d, {}))
        return c
    return Coordinates({d: DataArray(np.asarray(v), dims=(d,), name=d, attrs=attrs.get(d, {}))
                        for d, v in vectors.items()})


def make_dataarray(data, coords, dims=None, name=None, attrs=None):
    if HAVE_XARRAY:  # pragma: no cover
        return _xa.DataArray(data, coords=coords, dims=dims, name=name, attrs=attrs)
    dims = tuple(dims) if dims is not None else tuple(coords.dims if hasattr(coords, "dims") else coords.keys())
    return DataArray(data, coords=OrderedDict((d, coords[d]) for d in dims if d in coords), dims=dims,
                     name=name, attrs=attrs)


def make_dataset(data_vars=None, attrs=None):
    if HAVE_XARRAY:  # pragma: no cover
        return _xa.Dataset(data_vars or {}, attrs=attrs)
    return Dataset(data_vars or {}, attrs=attrs)


def stack_foci(datasets, dim="focal_point_index"):
    """plan/protocol.py:341-347: concat per-focus Datasets along a new leading dim.

    ``datasets`` may also be a dict name -> (stacked ndarray [F,...], coords, attrs) built
    directly from one batched device result (no per-focus copies)."""
    if HAVE_XARRAY and not isinstance(datasets, dict):  # pragma: no cover
        return _xa.concat([d.assign_coords(**{dim: i}) for i, d in enumerate(datasets)], dim=dim)
    if isinstance(datasets, dict):
        out = {}
        for name, (arr, coords, attrs) in datasets.items():
            c = OrderedDict([(dim, np.arange(arr.shape[0]))])
            c.update(coords)
            da_dims = (dim,) + tuple(coords.dims if hasattr(coords, "dims") else coords.keys())
            if HAVE_XARRAY:  # pragma: no cover
                out[name] = _xa.DataArray(arr, coords=c, dims=da_dims, name=name, attrs=attrs)
            else:
                out[name] = DataArray(arr, coords=c, dims=da_dims, name=name, attrs=attrs)
        return make_dataset(out)
    first = datasets[0]
    out = {}
    for name in first.keys():
        arr = np.stack([np.asarray(d[name].data) for d in datasets], axis=0)
        c = OrderedDict([(dim, np.arange(len(datasets)))])
        c.update(first[name].coords)
        out[name] = DataArray(arr, coords=c, dims=(dim,) + tuple(first[name].dims), name=name,
                              attrs=first[name].attrs)
    return Dataset(out)
