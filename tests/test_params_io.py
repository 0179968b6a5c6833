"""CPU: parameter files and the identity of a loaded tree.

* the reference stores weights as an .npz with a pickled nested dict (nbody_emulator.py:124-129); both that format
  and the package's pickle-free flat format must load WITHOUT executing anything from the file;
* `load_default_parameters()` / `create_emulator(load_params=True)` are exercised on a temporary blob
  (the pretrained one is absent, .MISSING_LARGE_BLOBS);
* the engine's fingerprint of a loaded tree is content-based: an in-place edit and a recycled `id` are noticed."""

import os
import pickle
import zipfile

import numpy as np
import pytest

import jax_nbody_emulator_with_dj_amd as J
from jax_nbody_emulator_with_dj_amd import params_io
from jax_nbody_emulator_with_dj_amd.engine import params_fingerprint


def _tree(seed=0, mid=8):
    return J.StyleNBodyEmulatorVelCore(mid_chan=mid).init(seed)


def _same(a, b):
    ta, tb = a["params"], b["params"]
    assert sorted(ta) == sorted(tb)
    for blk in ta:
        assert sorted(ta[blk]) == sorted(tb[blk])
        for lay in ta[blk]:
            assert sorted(ta[blk][lay]) == sorted(tb[blk][lay])
            for k in ta[blk][lay]:
                x, y = np.asarray(ta[blk][lay][k]), np.asarray(tb[blk][lay][k])
                assert x.dtype == y.dtype and np.array_equal(x, y), (blk, lay, k)


def test_flat_format_round_trip_has_no_pickle(tmp_path):
    p = _tree(1)
    f = tmp_path / "flat.npz"
    params_io.save_parameters(p, f)
    with np.load(f, allow_pickle=False) as z:                 # every member is a plain numeric array
        assert len(z.files) == 33 * 4 and all(k.count("/") == 2 for k in z.files)
        assert z["conv_l00/conv_0/weight"].shape == (8, 3, 3, 3, 3)
    _same(params_io.load_parameters(f), p)


def test_reference_format_is_read_with_the_restricted_unpickler(tmp_path):
    p = _tree(2)
    f = tmp_path / "ref.npz"
    np.savez(f, params=p["params"])                            # what the reference ships: a pickled nested dict
    with pytest.raises(ValueError):                            # NumPy itself refuses it without allow_pickle
        np.load(f, allow_pickle=False)["params"]
    _same(params_io.load_parameters(f), p)
    # {'params': {...}} one level deeper is accepted too
    np.savez(f, params=p)
    _same(params_io.load_parameters(f), p)
    # ... and the converter writes the flat format
    g = tmp_path / "flat.npz"
    params_io.convert_parameters(f, g)
    with np.load(g, allow_pickle=False) as z:
        assert "conv_r01/skip/style_bias" in z.files
    _same(params_io.load_parameters(g), p)


class _Evil:
    def __reduce__(self):
        return (os.system, ("echo pwned > /dev/null",))


def test_foreign_global_in_a_blob_is_refused(tmp_path):
    f = tmp_path / "evil.npz"
    obj = np.empty((), dtype=object)
    obj[()] = {"conv_l00": {"conv_0": {"weight": _Evil()}}}
    np.savez(f, params=obj)
    with pytest.raises(pickle.UnpicklingError, match="refusing to load global"):
        params_io.load_parameters(f)
    # a raw pickle smuggled in as the member is refused the same way
    payload = pickle.dumps(_Evil(), protocol=4)
    hdr = {"descr": "|O", "fortran_order": False, "shape": ()}
    import io
    from numpy.lib import format as npf
    buf = io.BytesIO()
    npf.write_array_header_1_0(buf, hdr)
    with zipfile.ZipFile(f, "w") as z:
        z.writestr("params.npy", buf.getvalue() + payload)
    with pytest.raises(pickle.UnpicklingError, match="refusing to load global"):
        params_io.load_parameters(f)


def test_non_array_leaves_are_refused(tmp_path):
    f = tmp_path / "odd.npz"
    np.savez(f, params={"conv_l00": {"conv_0": {"weight": "not an array"}}})
    with pytest.raises(ValueError, match="not a numeric array"):
        params_io.load_parameters(f)


@pytest.mark.parametrize("fmt", ["flat", "reference"])
def test_load_default_parameters_and_factory_on_a_temporary_blob(tmp_path, monkeypatch, fmt):
    p = J.StyleNBodyEmulatorVelCore().init(7)                  # production width: the factory's default architecture
    f = tmp_path / "nbody_emulator_params.npz"
    if fmt == "flat":
        params_io.save_parameters(p, f)
    else:
        np.savez(f, params=p["params"])
    monkeypatch.setenv("NBE_PARAMS", str(f))
    _same(J.load_default_parameters(), p)
    emu = J.create_emulator(premodulate=False, compute_vel=True,
                            processor_config=J.SubboxConfig(size=(128,) * 3, ndiv=(1, 1, 1)))
    _same(emu.params, p)
    assert emu.processor.params is emu.params
    # premodulated factory path: the walker output replaces the style leaves (nbody_emulator.py:353-358)
    emu2 = J.create_emulator(premodulate=True, premodulate_z=0.5, premodulate_Om=0.3)
    leaf = emu2.params["params"]["conv_l1"]["conv_0"]
    assert sorted(leaf) == ["bias", "dweight", "weight"] and leaf["dweight"].shape == leaf["weight"].shape
    monkeypatch.setenv("NBE_PARAMS", str(tmp_path / "missing.npz"))
    with pytest.raises(FileNotFoundError):
        J.load_default_parameters()


def test_fingerprint_is_content_based():
    p = _tree(3)
    fp = params_fingerprint(p)
    assert params_fingerprint(p) == fp                          # stable
    assert params_fingerprint({"params": {b: {l: dict(v) for l, v in ls.items()} for b, ls in p["params"].items()}}) == fp
    w = p["params"]["conv_l1"]["conv_1"]["weight"]
    keep = w[3, 2, 1, 1, 1]
    w[3, 2, 1, 1, 1] += 1e-3                                    # in-place edit: same id, different content
    assert params_fingerprint(p) != fp
    w[3, 2, 1, 1, 1] = keep
    assert params_fingerprint(p) == fp
    q = _tree(4)                                                # a different tree never matches, whatever its ids
    assert params_fingerprint(q) != fp
    p["params"]["conv_l1"]["conv_1"]["bias"] = p["params"]["conv_l1"]["conv_1"]["bias"].astype(np.float64)
    assert params_fingerprint(p) != fp                          # dtype is part of the identity
