"""ctypes binding of libnbe.so (C ABI: include/nbe.h).

The library is built in-tree by `build()` (hipcc, gfx950 only).  There is no CPU
fallback anywhere in this package: if the shared object is missing, or no MI355X
is visible, the product path raises.
"""

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NBE_LIB") or os.path.join(_HERE, "libnbe.so")   # NBE_LIB: timing-probe builds only
CSRC = os.path.join(_HERE, "csrc")
SOURCES = ("nbe_kernels.hip", "nbe_kernels_h3.hip", "nbe_engine.cpp")


class NBEError(RuntimeError):
    pass


class NBERangeError(NBEError):
    """A call on an f16-based context produced non-finite values from a finite input (include/nbe.h, "Range")."""


class LayerDesc(C.Structure):
    _fields_ = [
        ("block", C.c_char_p), ("layer", C.c_char_p),
        ("cout", C.c_int), ("cin", C.c_int), ("k", C.c_int),
        ("weight", C.c_void_p), ("bias", C.c_void_p),
        ("style_weight", C.c_void_p), ("style_bias", C.c_void_p),
        ("dweight", C.c_void_p),
    ]


PROGRESS_CB = C.CFUNCTYPE(None, C.c_int, C.c_int, C.c_void_p)

# name -> (restype, argtypes); kept in one table so tests can check every symbol of include/nbe.h
SIGNATURES = {
    "nbe_last_error": (C.c_char_p, []),
    "nbe_version": (C.c_int, []),
    "nbe_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "nbe_destroy": (C.c_int, [C.c_void_p]),
    "nbe_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "nbe_use_own_stream": (C.c_int, [C.c_void_p]),
    "nbe_synchronize": (C.c_int, [C.c_void_p]),
    "nbe_set_arch": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int]),
    "nbe_load_style_weights": (C.c_int, [C.c_void_p, C.POINTER(LayerDesc), C.c_int]),
    "nbe_load_premod_weights": (C.c_int, [C.c_void_p, C.POINTER(LayerDesc), C.c_int]),
    "nbe_set_cosmology": (C.c_int, [C.c_void_p, C.c_float, C.c_float]),
    "nbe_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                              C.c_void_p, C.c_void_p]),
    "nbe_process_box": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int),
                                  C.POINTER(C.c_int), C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_int,
                                  PROGRESS_CB, C.c_void_p]),
    "nbe_process_region": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                     C.POINTER(C.c_int64), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int,
                                     C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_int,
                                     C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "nbe_brick_halo_bytes": (C.c_int64, [C.c_void_p, C.POINTER(C.c_int64), C.c_int]),
    "nbe_brick_plan": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "nbe_brick_interior": (C.c_int, [C.c_void_p]),
    "nbe_brick_exchange": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "nbe_brick_encode": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "nbe_brick_finish": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_int]),
    "nbe_plan_tiles": (C.c_int, [C.POINTER(C.c_int64), C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int)]),
    "nbe_plan_tiles_ctx": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_int)]),
    "nbe_set_max_tile": (C.c_int, [C.c_void_p, C.c_int]),
    "nbe_set_slab": (C.c_int, [C.c_void_p, C.c_int]),
    "nbe_set_periodic": (C.c_int, [C.c_void_p, C.c_int]),
    "nbe_set_precision": (C.c_int, [C.c_void_p, C.c_int]),
    "nbe_host_alloc": (C.c_void_p, [C.c_size_t]),
    "nbe_host_free": (C.c_int, [C.c_void_p]),
    "nbe_host_trim": (C.c_int, []),
    "nbe_check_finite": (C.c_int, [C.c_void_p]),
    "nbe_set_input_range": (C.c_int, [C.c_void_p, C.c_float]),
    "nbe_query": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double)]),
    "nbe_growth_factor": (C.c_double, [C.c_double, C.c_double]),
    "nbe_vel_norm": (C.c_double, [C.c_double, C.c_double]),
    "nbe_test_layer": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                 C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                 C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "nbe_test_layer_gauged": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "nbe_test_layer_gauged_res": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_void_p]),
    "nbe_test_modulate": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                    C.c_float, C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_void_p]),
    "nbe_probe_begin": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.c_int]),
    "nbe_probe_slots": (C.c_int, [C.c_void_p]),
    "nbe_probe_layout": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int64)]),
    "nbe_probe_read": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "nbe_probe_end": (C.c_int, [C.c_void_p]),
    "nbe_profile_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "nbe_profile_reset": (C.c_int, [C.c_void_p]),
    "nbe_profile_count": (C.c_int, [C.c_void_p]),
    "nbe_profile_entry": (C.c_int, [C.c_void_p, C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_double),
                                    C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "nbe_workspace_bytes": (C.c_int64, [C.c_void_p]),
    "nbe_debug_phase_cycles": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
}

_lib = None


def source_hash():
    """sha256 (16 hex digits) over the kernel / engine sources: identifies what a measurement was taken on."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".hip", ".cpp", ".h")):
            h.update(f.encode())
            h.update(open(os.path.join(CSRC, f), "rb").read())
    return h.hexdigest()[:16]


def build(force=False, verbose=False):
    """Compile the HIP sources for gfx950 into libnbe.so (in-tree)."""
    srcs = [os.path.join(CSRC, s) for s in SOURCES]
    deps = srcs + [os.path.join(CSRC, "nbe_kernels.h"), os.path.join(CSRC, "nbe_kernels_internal.h"), os.path.join(CSRC, "nbe_kernels_wino.h"), os.path.join(CSRC, "nbe_kernels_head.h"), os.path.join(_HERE, "..", "include", "nbe.h")]
    if not force and os.path.exists(LIB_PATH):
        if all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(d) for d in deps):
            return LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", LIB_PATH] + srcs
    if os.environ.get("NBE_BUILD_DBG") == "1":      # timing-experiment switches in the kernels (never for results)
        cmd.insert(1, "-DNBE_DBG=1")
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return LIB_PATH


def lib():
    """The loaded library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NBEError(
                "libnbe.so is missing at %s. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc, gfx950). There is no CPU fallback." % LIB_PATH)
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


class _PinnedBlock:
    """Owner of one pooled pinned buffer: returns it to the library's pool when the last NumPy view is gone."""

    def __init__(self, ptr, nbytes):
        self.ptr, self.nbytes = ptr, nbytes

    def __del__(self):
        try:
            if self.ptr:
                lib().nbe_host_free(C.c_void_p(self.ptr))
                self.ptr = 0
        except Exception:
            pass


def pinned_empty(shape, dtype):
    """Uninitialised NumPy array in pinned host memory from the library's pool (include/nbe.h, nbe_host_alloc): the GPU
    copies straight into it, asynchronously.  Falls back to ordinary memory if pinning fails."""
    import numpy as np
    dtype = np.dtype(dtype)
    n = int(np.prod(shape)) * dtype.itemsize
    ptr = lib().nbe_host_alloc(max(n, 1)) if n >= (1 << 20) else None   # small arrays are not worth a pinned block
    if not ptr:
        return np.empty(shape, dtype)
    blk = _PinnedBlock(int(ptr), n)
    buf = (C.c_char * n).from_address(int(ptr))
    buf._nbe_block = blk                       # the ctypes view (kept alive by the array's buffer) keeps the block alive
    return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)


def check(rc):
    if rc != 0:
        msg = lib().nbe_last_error().decode("utf-8", "replace")
        raise (NBERangeError if rc == 2 else NBEError)(msg)
