"""HDF5 files of the reference's data path without h5py: a ctypes binding of the system's libhdf5 (1.10 / 1.12 C API).

What the reference reads and writes with ``h5py.File`` (rho_diffusion/data/synthetic.py:186-189, 258-261, 285-289, 307-333:
datasets ``density`` / ``l`` / ``m`` + attribute ``seed``; scripts/inference.py:168-169: dataset ``data`` = the generated samples)
is whole-array dataset writes, whole / row reads and scalar attributes - five calls of the C library each.  h5py is not installed in
this image, libhdf5 is (``/opt/conda/lib/libhdf5.so.103``); files written here open in h5py and vice versa (same library, default
creation properties, native little-endian types).  Host-side I/O only: nothing here touches the GPU.

    write("samples.h5", {"data": array})                      # h5py: f["data"] = array
    read("cache.h5", "density", index=7)                      # h5py: f["density"][7]
    shape("cache.h5", "density"); read_attr("cache.h5", "seed")
"""
from __future__ import annotations

import ctypes as C
import ctypes.util
import glob
import os
from typing import Dict, Optional, Sequence, Tuple, Union

import numpy as np

__all__ = ["H5Error", "available", "write", "read", "shape", "read_attr", "datasets"]

hid_t = C.c_int64
hsize_t = C.c_uint64
H5F_ACC_RDONLY, H5F_ACC_TRUNC, H5F_ACC_EXCL = 0, 2, 4
H5P_DEFAULT = 0
H5S_ALL = 0
H5S_SELECT_SET = 0
H5S_SCALAR = 0

_NATIVE = {"float32": "H5T_NATIVE_FLOAT_g", "float64": "H5T_NATIVE_DOUBLE_g", "int32": "H5T_NATIVE_INT32_g", "int64": "H5T_NATIVE_INT64_g",
           "uint8": "H5T_NATIVE_UINT8_g", "int8": "H5T_NATIVE_INT8_g", "int16": "H5T_NATIVE_INT16_g", "uint16": "H5T_NATIVE_UINT16_g",
           "uint32": "H5T_NATIVE_UINT32_g", "uint64": "H5T_NATIVE_UINT64_g"}
# H5T_class_t / size / sign -> numpy dtype of a stored dataset
H5T_INTEGER, H5T_FLOAT = 0, 1


class H5Error(RuntimeError):
    pass


_lib: Optional[C.CDLL] = None


def _candidates():
    env = os.environ.get("RHO_HDF5_LIB")
    if env:
        yield env
    found = ctypes.util.find_library("hdf5")
    if found:
        yield found
    for pat in ("/opt/conda/lib/libhdf5.so*", "/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so*", "/usr/lib/x86_64-linux-gnu/libhdf5*.so*",
                "/usr/lib64/libhdf5.so*", "/usr/local/lib/libhdf5.so*"):
        for p in sorted(glob.glob(pat)):
            base = os.path.basename(p)
            if base.startswith("libhdf5.so") or base.startswith("libhdf5_serial.so"):
                yield p


def _load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    last = None
    for cand in _candidates():
        try:
            lib = C.CDLL(cand)
            lib.H5open.restype = C.c_int
            if lib.H5open() < 0:
                raise OSError("H5open failed")
        except OSError as exc:
            last = exc
            continue
        sig = {
            "H5Fcreate": (hid_t, [C.c_char_p, C.c_uint, hid_t, hid_t]), "H5Fopen": (hid_t, [C.c_char_p, C.c_uint, hid_t]),
            "H5Fclose": (C.c_int, [hid_t]), "H5Screate_simple": (hid_t, [C.c_int, C.POINTER(hsize_t), C.POINTER(hsize_t)]),
            "H5Screate": (hid_t, [C.c_int]), "H5Sclose": (C.c_int, [hid_t]),
            "H5Sselect_hyperslab": (C.c_int, [hid_t, C.c_int, C.POINTER(hsize_t), C.POINTER(hsize_t), C.POINTER(hsize_t), C.POINTER(hsize_t)]),
            "H5Sget_simple_extent_ndims": (C.c_int, [hid_t]),
            "H5Sget_simple_extent_dims": (C.c_int, [hid_t, C.POINTER(hsize_t), C.POINTER(hsize_t)]),
            "H5Dcreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t]), "H5Dopen2": (hid_t, [hid_t, C.c_char_p, hid_t]),
            "H5Dclose": (C.c_int, [hid_t]), "H5Dget_space": (hid_t, [hid_t]), "H5Dget_type": (hid_t, [hid_t]),
            "H5Dwrite": (C.c_int, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
            "H5Dread": (C.c_int, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
            "H5Tget_class": (C.c_int, [hid_t]), "H5Tget_size": (C.c_size_t, [hid_t]), "H5Tget_sign": (C.c_int, [hid_t]), "H5Tclose": (C.c_int, [hid_t]),
            "H5Acreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t]), "H5Aopen": (hid_t, [hid_t, C.c_char_p, hid_t]),
            "H5Awrite": (C.c_int, [hid_t, hid_t, C.c_void_p]), "H5Aread": (C.c_int, [hid_t, hid_t, C.c_void_p]),
            "H5Aget_type": (hid_t, [hid_t]), "H5Aclose": (C.c_int, [hid_t]),
            "H5Lexists": (C.c_int, [hid_t, C.c_char_p, hid_t]),
            "H5Gget_num_objs": (C.c_int, [hid_t, C.POINTER(hsize_t)]),
            "H5Gget_objname_by_idx": (C.c_ssize_t, [hid_t, hsize_t, C.c_char_p, C.c_size_t]),
            "H5Eset_auto2": (C.c_int, [hid_t, C.c_void_p, C.c_void_p]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        lib.H5Eset_auto2(0, None, None)          # errors are reported through return codes -> H5Error, not printed stacks
        _lib = lib
        return lib
    raise H5Error("libhdf5 not found (set RHO_HDF5_LIB to its path): HDF5 files are read and written through the system's C library, "
                  f"h5py is not required; last error: {last}")


def available() -> bool:
    try:
        _load()
        return True
    except H5Error:
        return False


def _native(dtype: np.dtype) -> int:
    key = np.dtype(dtype).name
    if key not in _NATIVE:
        raise H5Error(f"unsupported dtype {key} (supported: {sorted(_NATIVE)})")
    return hid_t.in_dll(_load(), _NATIVE[key]).value


def _chk(v: int, what: str) -> int:
    if v < 0:
        raise H5Error(f"{what} failed")
    return v


class _File:
    def __init__(self, path, mode: str):
        L = _load()
        p = os.fspath(path).encode()
        if mode == "r":
            if not os.path.exists(os.fspath(path)):
                raise FileNotFoundError(os.fspath(path))
            self.id = L.H5Fopen(p, H5F_ACC_RDONLY, H5P_DEFAULT)
        elif mode in ("w", "x"):
            if mode == "x" and os.path.exists(os.fspath(path)):
                raise FileExistsError(os.fspath(path))       # h5py mode "x": fail if the file exists (synthetic.py:322)
            self.id = L.H5Fcreate(p, H5F_ACC_TRUNC if mode == "w" else H5F_ACC_EXCL, H5P_DEFAULT, H5P_DEFAULT)
        else:
            raise ValueError("mode must be 'r', 'w' or 'x'")
        _chk(self.id, f"opening {os.fspath(path)} ({mode})")

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        _load().H5Fclose(self.id)


def write(path, arrays: Dict[str, np.ndarray], attrs: Optional[Dict[str, Union[int, float]]] = None, mode: str = "w") -> None:
    """Create ``path`` with one contiguous dataset per entry (``h5f[name] = array``) and scalar root attributes."""
    L = _load()
    with _File(path, mode) as f:
        for name, arr in arrays.items():
            a = np.ascontiguousarray(arr)
            tid = _native(a.dtype)
            dims = (hsize_t * max(a.ndim, 1))(*a.shape)
            sid = _chk(L.H5Screate_simple(a.ndim, dims, None) if a.ndim else L.H5Screate(H5S_SCALAR), "H5Screate")
            did = _chk(L.H5Dcreate2(f.id, name.encode(), tid, sid, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT), f"creating dataset {name}")
            rc = L.H5Dwrite(did, tid, H5S_ALL, H5S_ALL, H5P_DEFAULT, a.ctypes.data)
            L.H5Dclose(did)
            L.H5Sclose(sid)
            _chk(rc, f"writing dataset {name}")
        for name, val in (attrs or {}).items():
            v = np.asarray(val)
            if v.dtype.kind in "iu":
                v = v.astype(np.int64)
            elif v.dtype.kind == "f":
                v = v.astype(np.float64)
            tid = _native(v.dtype)
            sid = _chk(L.H5Screate(H5S_SCALAR), "H5Screate")
            aid = _chk(L.H5Acreate2(f.id, name.encode(), tid, sid, H5P_DEFAULT, H5P_DEFAULT), f"creating attribute {name}")
            buf = np.ascontiguousarray(v).reshape(1)
            rc = L.H5Awrite(aid, tid, buf.ctypes.data)
            L.H5Aclose(aid)
            L.H5Sclose(sid)
            _chk(rc, f"writing attribute {name}")


def _np_dtype_of(tid: int) -> np.dtype:
    L = _load()
    cls, size = L.H5Tget_class(tid), int(L.H5Tget_size(tid))
    if cls == H5T_FLOAT and size in (4, 8):
        return np.dtype(f"f{size}")
    if cls == H5T_INTEGER and size in (1, 2, 4, 8):
        return np.dtype(("i" if L.H5Tget_sign(tid) == 1 else "u") + str(size))
    raise H5Error(f"unsupported stored type (class {cls}, {size} bytes)")


def _open_dataset(f: _File, name: str):
    L = _load()
    if L.H5Lexists(f.id, name.encode(), H5P_DEFAULT) <= 0:
        raise KeyError(f"Unable to open object (object '{name}' doesn't exist)")       # h5py's wording
    did = _chk(L.H5Dopen2(f.id, name.encode(), H5P_DEFAULT), f"opening dataset {name}")
    sid = _chk(L.H5Dget_space(did), "H5Dget_space")
    nd = _chk(L.H5Sget_simple_extent_ndims(sid), "ndims")
    dims = (hsize_t * max(nd, 1))()
    if nd:
        L.H5Sget_simple_extent_dims(sid, dims, None)
    return did, sid, tuple(int(dims[i]) for i in range(nd))


def shape(path, name: str) -> Tuple[int, ...]:
    L = _load()
    with _File(path, "r") as f:
        did, sid, shp = _open_dataset(f, name)
        L.H5Sclose(sid)
        L.H5Dclose(did)
        return shp


def read(path, name: str, index: Optional[Union[int, slice]] = None) -> np.ndarray:
    """The whole dataset, or ``dataset[index]`` along axis 0 (an int: that row with the axis dropped, as h5py; a slice with step 1:
    the rows) read through a hyperslab selection - a replayed training set is never loaded whole."""
    L = _load()
    with _File(path, "r") as f:
        did, sid, shp = _open_dataset(f, name)
        try:
            ftype = _chk(L.H5Dget_type(did), "H5Dget_type")
            dt = _np_dtype_of(ftype)
            L.H5Tclose(ftype)
            tid = _native(dt)
            if index is None or len(shp) == 0:
                out = np.empty(shp, dtype=dt)
                _chk(L.H5Dread(did, tid, H5S_ALL, H5S_ALL, H5P_DEFAULT, out.ctypes.data), f"reading {name}")
                return out
            if isinstance(index, (int, np.integer)):
                i = int(index) + (shp[0] if index < 0 else 0)
                if not 0 <= i < shp[0]:
                    raise IndexError(f"index {index} out of range for dataset {name} of length {shp[0]}")
                lo, n, drop = i, 1, True
            else:
                lo, hi, step = index.indices(shp[0])
                if step != 1:
                    raise H5Error("only unit-stride row slices are supported")
                n, drop = max(hi - lo, 0), False
            out = np.empty((n,) + shp[1:], dtype=dt)
            if n:
                nd = len(shp)
                start = (hsize_t * nd)(lo, *([0] * (nd - 1)))
                count = (hsize_t * nd)(n, *shp[1:])
                _chk(L.H5Sselect_hyperslab(sid, H5S_SELECT_SET, start, None, count, None), "H5Sselect_hyperslab")
                msid = _chk(L.H5Screate_simple(nd, count, None), "H5Screate_simple")
                rc = L.H5Dread(did, tid, msid, sid, H5P_DEFAULT, out.ctypes.data)
                L.H5Sclose(msid)
                _chk(rc, f"reading {name}[{index}]")
            return out[0] if drop else out
        finally:
            L.H5Sclose(sid)
            L.H5Dclose(did)


def read_attr(path, name: str):
    L = _load()
    with _File(path, "r") as f:
        aid = L.H5Aopen(f.id, name.encode(), H5P_DEFAULT)
        if aid < 0:
            raise KeyError(f"attribute {name!r} not found")
        ftype = _chk(L.H5Aget_type(aid), "H5Aget_type")
        dt = _np_dtype_of(ftype)
        L.H5Tclose(ftype)
        buf = np.empty(1, dtype=dt)
        rc = L.H5Aread(aid, _native(dt), buf.ctypes.data)
        L.H5Aclose(aid)
        _chk(rc, f"reading attribute {name}")
        return buf[0].item()


def datasets(path) -> Sequence[str]:
    """Names of the root group's members."""
    L = _load()
    with _File(path, "r") as f:
        n = hsize_t()
        _chk(L.H5Gget_num_objs(f.id, C.byref(n)), "H5Gget_num_objs")
        out = []
        for i in range(int(n.value)):
            buf = C.create_string_buffer(256)
            L.H5Gget_objname_by_idx(f.id, i, buf, 256)
            out.append(buf.value.decode())
        return out
