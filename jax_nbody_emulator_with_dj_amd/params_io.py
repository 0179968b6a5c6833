"""Parameter files: a code-free format of our own and a restricted reader for the reference's.

The reference stores its weights as an .npz whose single member `params` is a pickled nested dict
(reference nbody_emulator.py:124-129, `np.load(..., allow_pickle=True)['params'].item()`): loading such a file
executes whatever the pickle names.  Here

  * `save_parameters` / `load_parameters` use a FLAT .npz -- one float array per leaf, key "block/layer/leaf" --
    read with `allow_pickle=False`: nothing in the file is executed;
  * a file in the reference's format is read with an unpickler that resolves NumPy's array / dtype / scalar
    constructors and nothing else; a pickle that names any other global (e.g. `os.system`) is refused;
  * `python -m jax_nbody_emulator_with_dj_amd.params_io SRC DST` converts the reference's blob to the flat format.

Leaves pickled as JAX arrays need JAX to reconstruct and are refused too: convert such a tree to NumPy where it was
made (`jax.tree_util.tree_map(np.asarray, params)`), then save it with `save_parameters`.
"""

import pickle
import sys
import zipfile

import numpy as np
from numpy.lib import format as _npf

SEP = "/"

# (module, name) pairs a reference-format pickle may name: how NumPy pickles ndarray, dtype and scalars
_NUMPY_GLOBALS = {
    ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
    ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
    ("numpy.core.numeric", "_frombuffer"), ("numpy._core.numeric", "_frombuffer"),
    ("numpy", "ndarray"), ("numpy", "dtype"),
}


class _RestrictedUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if (module, name) not in _NUMPY_GLOBALS:
            raise pickle.UnpicklingError(
                "refusing to load global %s.%s from a parameter file: only NumPy arrays in nested dicts are accepted"
                % (module, name))
        import numpy._core.multiarray as ma
        import numpy._core.numeric as nu
        if name == "_reconstruct":
            return ma._reconstruct
        if name == "scalar":
            return ma.scalar
        if name == "_frombuffer":
            return nu._frombuffer
        return getattr(np, name)


def _read_member(zf, member):
    """One .npy member of an .npz without ever calling pickle.load: plain arrays through NumPy's reader
    (allow_pickle=False), object arrays through the restricted unpickler."""
    with zf.open(member) as f:
        version = _npf.read_magic(f)
        if version == (1, 0):
            shape, fortran, dtype = _npf.read_array_header_1_0(f)
        elif version in ((2, 0), (3, 0)):
            shape, fortran, dtype = _npf.read_array_header_2_0(f)
        else:
            raise ValueError("unsupported .npy version %s in %s" % (version, member))
        if dtype.hasobject:
            return _RestrictedUnpickler(f).load()
    with zf.open(member) as f:
        return _npf.read_array(f, allow_pickle=False)


def _check_tree(tree):
    if not isinstance(tree, dict):
        raise ValueError("parameter tree must be a dict of blocks")
    out = {}
    for b, layers in tree.items():
        if not isinstance(layers, dict):
            raise ValueError("block %r is not a dict of layers" % (b,))
        out[str(b)] = {}
        for l, leaves in layers.items():
            if not isinstance(leaves, dict):
                raise ValueError("layer %r/%r is not a dict of arrays" % (b, l))
            out[str(b)][str(l)] = {str(k): np.asarray(v) for k, v in leaves.items()}
            for k, v in out[str(b)][str(l)].items():
                if v.dtype.kind not in "fiub":
                    raise ValueError("leaf %s/%s/%s is not a numeric array (dtype %s)" % (b, l, k, v.dtype))
    return out


def load_parameters(path):
    """-> {'params': {block: {layer: {leaf: ndarray}}}} from a flat .npz (this package's format) or from the
    reference's pickled .npz (restricted unpickler).  Nothing from the file is executed in either case."""
    with zipfile.ZipFile(path) as zf:
        names = [n[:-4] for n in zf.namelist() if n.endswith(".npy")]
        if names and all(SEP in n for n in names):                       # flat format
            tree = {}
            for n in names:
                parts = n.split(SEP)
                if len(parts) != 3:
                    raise ValueError("flat parameter key %r is not block/layer/leaf" % n)
                arr = _read_member(zf, n + ".npy")
                tree.setdefault(parts[0], {}).setdefault(parts[1], {})[parts[2]] = np.asarray(arr)
            return {"params": _check_tree(tree)}
        if "params" not in names:
            raise ValueError("%s holds neither block/layer/leaf arrays nor a 'params' member" % path)
        obj = _read_member(zf, "params.npy")
    if isinstance(obj, np.ndarray) and obj.dtype.hasobject:
        obj = obj.item()                                                  # 0-d object array holding the dict
    if isinstance(obj, dict) and set(obj) == {"params"}:
        obj = obj["params"]
    return {"params": _check_tree(obj)}


def save_parameters(params, path):
    """Write a parameter tree as a flat .npz (one array per leaf, key block/layer/leaf, no pickle)."""
    tree = params["params"] if "params" in params else params
    flat = {}
    for b, layers in tree.items():
        for l, leaves in layers.items():
            for k, v in leaves.items():
                flat[SEP.join((b, l, k))] = np.ascontiguousarray(np.asarray(v))
    with open(path, "wb") as f:
        np.savez(f, **flat)


def convert_parameters(src, dst):
    """Reference-format (or flat) .npz -> flat .npz."""
    save_parameters(load_parameters(src), dst)


if __name__ == "__main__":
    if len(sys.argv) != 3:
        sys.exit("usage: python -m jax_nbody_emulator_with_dj_amd.params_io SRC.npz DST.npz")
    convert_parameters(sys.argv[1], sys.argv[2])
    print("wrote", sys.argv[2])
