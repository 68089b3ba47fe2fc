"""HDF5 access without h5py: a ctypes binding of the system's libhdf5 that reads -- and, for
tests and synthetic fixtures, writes -- the feature files of the reference's input pipeline: datasets `features [N,(1,)D]` and
`coords [N,2]`, chunked and resizable along axis 0 as madeleine/preprocessing/conch_patch_embedder.py:16-66 (`save_hdf5`)
creates them, consumed by madeleine/datasets/wsi_dataset.py:14-19 (`load_features`).  libhdf5 converts the stored element
type (float16 / float32 / float64) to float32 while reading.  Host glue: no numerics here."""
import ctypes
import ctypes.util
import glob
import os
import threading

import numpy as np

_hid = ctypes.c_int64        # hid_t is 64-bit since HDF5 1.10
_hsize = ctypes.c_uint64
_LIB = None
_LOCK = threading.Lock()     # libhdf5 is not built thread-safe everywhere: one call sequence at a time


def _find():
    cands = []
    env = os.environ.get("MADELEINE_LIBHDF5")
    if env:
        cands.append(env)
    found = ctypes.util.find_library("hdf5")
    if found:
        cands.append(found)
    for pat in ("/opt/conda/lib/libhdf5.so*", "/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so*",
                "/usr/lib/x86_64-linux-gnu/libhdf5*.so*", "/usr/lib64/libhdf5.so*", "/usr/local/lib/libhdf5.so*"):
        cands += sorted(glob.glob(pat))
    for c in cands:
        try:
            return ctypes.CDLL(c)
        except OSError:
            continue
    raise ImportError("madeleine_amd.h5io: no libhdf5 shared library found (set MADELEINE_LIBHDF5=/path/to/libhdf5.so, "
                      "or install h5py)")


def lib():
    global _LIB
    if _LIB is None:
        L = _find()
        sig = {
            "H5open": (ctypes.c_int, []),
            "H5Fopen": (_hid, [ctypes.c_char_p, ctypes.c_uint, _hid]),
            "H5Fcreate": (_hid, [ctypes.c_char_p, ctypes.c_uint, _hid, _hid]),
            "H5Fclose": (ctypes.c_int, [_hid]),
            "H5Dopen2": (_hid, [_hid, ctypes.c_char_p, _hid]),
            "H5Dclose": (ctypes.c_int, [_hid]),
            "H5Dget_space": (_hid, [_hid]),
            "H5Dread": (ctypes.c_int, [_hid, _hid, _hid, _hid, _hid, ctypes.c_void_p]),
            "H5Dwrite": (ctypes.c_int, [_hid, _hid, _hid, _hid, _hid, ctypes.c_void_p]),
            "H5Dcreate2": (_hid, [_hid, ctypes.c_char_p, _hid, _hid, _hid, _hid, _hid]),
            "H5Sget_simple_extent_ndims": (ctypes.c_int, [_hid]),
            "H5Sget_simple_extent_dims": (ctypes.c_int, [_hid, ctypes.POINTER(_hsize), ctypes.POINTER(_hsize)]),
            "H5Screate_simple": (_hid, [ctypes.c_int, ctypes.POINTER(_hsize), ctypes.POINTER(_hsize)]),
            "H5Sclose": (ctypes.c_int, [_hid]),
            "H5Pcreate": (_hid, [_hid]),
            "H5Pset_chunk": (ctypes.c_int, [_hid, ctypes.c_int, ctypes.POINTER(_hsize)]),
            "H5Pclose": (ctypes.c_int, [_hid]),
            "H5Lexists": (ctypes.c_int, [_hid, ctypes.c_char_p, _hid]),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = res, args
        if L.H5open() < 0:
            raise ImportError("madeleine_amd.h5io: H5open() failed")
        L.native_float = _hid.in_dll(L, "H5T_NATIVE_FLOAT_g").value
        L.ieee_f32le = _hid.in_dll(L, "H5T_IEEE_F32LE_g").value
        L.ieee_f16 = None
        L.dcpl_class = _hid.in_dll(L, "H5P_CLS_DATASET_CREATE_ID_g").value
        _LIB = L
    return _LIB


def read_dataset(path, name="features") -> np.ndarray:
    """The whole dataset `name` of the file as a float32 array (any stored float type, any layout libhdf5 reads)."""
    L = lib()
    with _LOCK:
        fid = L.H5Fopen(os.fsencode(path), 0, 0)             # H5F_ACC_RDONLY, H5P_DEFAULT
        if fid < 0:
            raise OSError("cannot open HDF5 file %s" % path)
        try:
            if L.H5Lexists(fid, name.encode(), 0) <= 0:
                raise KeyError("%s has no dataset %r" % (path, name))
            did = L.H5Dopen2(fid, name.encode(), 0)
            if did < 0:
                raise OSError("cannot open dataset %r of %s" % (name, path))
            try:
                sid = L.H5Dget_space(did)
                nd = L.H5Sget_simple_extent_ndims(sid)
                dims = (_hsize * max(nd, 1))()
                L.H5Sget_simple_extent_dims(sid, dims, None)
                L.H5Sclose(sid)
                out = np.empty(tuple(int(d) for d in dims[:nd]), dtype=np.float32)
                if out.size and L.H5Dread(did, L.native_float, 0, 0, 0, out.ctypes.data_as(ctypes.c_void_p)) < 0:
                    raise OSError("H5Dread failed on %r of %s" % (name, path))
                return out
            finally:
                L.H5Dclose(did)
        finally:
            L.H5Fclose(fid)


def write_datasets(path, arrays: dict, chunk_rows: int = 256):
    """Writes float32 arrays as chunked datasets, resizable along axis 0 -- the layout save_hdf5 produces
    (conch_patch_embedder.py:38-42: chunks, maxshape=(None,) + shape[1:]).  For tests / synthetic feature files."""
    L = lib()
    unlimited = _hsize(0xFFFFFFFFFFFFFFFF)                    # H5S_UNLIMITED
    with _LOCK:
        fid = L.H5Fcreate(os.fsencode(path), 2, 0, 0)         # H5F_ACC_TRUNC
        if fid < 0:
            raise OSError("cannot create %s" % path)
        try:
            for name, a in arrays.items():
                a = np.ascontiguousarray(a, dtype=np.float32)
                nd = a.ndim
                dims = (_hsize * nd)(*a.shape)
                maxd = (_hsize * nd)(unlimited.value, *a.shape[1:])
                chunk = (_hsize * nd)(max(1, min(chunk_rows, a.shape[0])), *a.shape[1:])
                sid = L.H5Screate_simple(nd, dims, maxd)
                pl = L.H5Pcreate(L.dcpl_class)
                L.H5Pset_chunk(pl, nd, chunk)
                did = L.H5Dcreate2(fid, name.encode(), L.ieee_f32le, sid, 0, pl, 0)
                if did < 0:
                    raise OSError("cannot create dataset %r" % name)
                if a.size and L.H5Dwrite(did, L.native_float, 0, 0, 0, a.ctypes.data_as(ctypes.c_void_p)) < 0:
                    raise OSError("H5Dwrite failed on %r" % name)
                L.H5Dclose(did)
                L.H5Pclose(pl)
                L.H5Sclose(sid)
        finally:
            L.H5Fclose(fid)
    return path
