"""CPU: the host side of the drop-in boundary -- API surface, index tables, error behaviour,
cosmology, premodulation walkers, and that libnbe.so loads and exports every symbol of include/nbe.h.
No GPU compute is attempted here."""

import os
import re

import numpy as np
import pytest

import jax_nbody_emulator_with_dj_amd as J
from jax_nbody_emulator_with_dj_amd import _lib
from oracle import cosmology as OC, params as OP, layers as OL

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---- C ABI ---------------------------------------------------------------------------------------------
def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "nbe.h")).read()
    declared = set(re.findall(r"\b(nbe_[a-z0-9_]+)\s*\(", hdr)) - {"nbe_progress_cb"}
    assert len(declared) >= 20
    _lib.build()                                   # no-op when libnbe.so is up to date
    lib = _lib.lib()
    for name in sorted(declared):
        assert hasattr(lib, name), "libnbe.so does not export %s" % name
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert lib.nbe_version() >= 100


def test_no_gpu_means_loud_failure():
    """The product path has no CPU fallback: without a device, context creation raises."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    from jax_nbody_emulator_with_dj_amd.engine import Engine
    with pytest.raises(_lib.NBEError, match="no HIP device|no CPU fallback"):
        Engine()
    m = J.StyleNBodyEmulatorVelCore()
    p = m.init(0)
    with pytest.raises(_lib.NBEError):
        m.apply(p, np.zeros((1, 3, 104, 104, 104), np.float32), 0.3, 1.0, 1.0)


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "jax_nbody_emulator_with_dj_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f


# ---- public surface (reference __init__.py:73-95) ---------------------------------------------------------
def test_public_names():
    expected = ["create_emulator", "NBodyEmulator", "SubboxConfig", "SubboxProcessor", "load_default_parameters",
                "modulate_emulator_parameters", "modulate_emulator_parameters_vel", "growth_factor", "hubble_rate",
                "growth_rate", "dlogH_dloga", "vel_norm", "acc_norm", "StyleNBodyEmulatorCore",
                "StyleNBodyEmulatorVelCore", "NBodyEmulatorCore", "NBodyEmulatorVelCore"]
    assert sorted(J.__all__) == sorted(expected)
    for n in expected:
        assert hasattr(J, n)


def test_model_constructors_and_param_trees():
    m = J.StyleNBodyEmulatorVelCore()
    assert (m.style_size, m.in_chan, m.out_chan, m.mid_chan, m.eps) == (2, 3, 3, 64, 1e-8)
    p = m.init(42)["params"]
    assert len(p) == 15
    for blk in ("conv_l00", "conv_l01", "conv_l1", "conv_l2", "conv_c", "conv_r2", "conv_r1", "conv_r00", "conv_r01"):
        assert sorted(p[blk]) == ["conv_0", "conv_1", "skip"]
    for blk in ("down_l0", "down_l1", "down_l2", "up_r2", "up_r1", "up_r0"):
        assert sorted(p[blk]) == ["conv_0"]
    leaf = p["conv_r1"]["conv_0"]
    assert sorted(leaf) == ["bias", "style_bias", "style_weight", "weight"]
    assert leaf["weight"].shape == (128, 128, 3, 3, 3) and leaf["style_weight"].shape == (128, 2)
    assert np.all(leaf["style_bias"] == 1) and np.all(leaf["bias"] == 0)
    # vel and non-vel style models share the parameter structure (tests/test_style_nbody_emulator_vel_core.py:421-446)
    q = J.StyleNBodyEmulatorCore().init(42)["params"]
    assert {(b, l, k): v.shape for b in p for l in p[b] for k, v in p[b][l].items()} == \
           {(b, l, k): v.shape for b in q for l in q[b] for k, v in q[b][l].items()}
    pm = J.NBodyEmulatorVelCore(mid_chan=8).init(1)["params"]
    assert sorted(pm["conv_l1"]["conv_0"]) == ["bias", "dweight", "weight"]
    assert sorted(J.NBodyEmulatorCore(mid_chan=8).init(1)["params"]["conv_l1"]["conv_0"]) == ["bias", "weight"]


# ---- SubboxConfig (tests/test_subbox.py:31-204) -----------------------------------------------------------
def test_subbox_config_tables_match_oracle():
    from oracle import subbox as OS
    cfg = J.SubboxConfig(size=(256, 128, 64), ndiv=(2, 1, 1))
    assert cfg.NDIM == 3 and cfg.n_subboxes == 2 and cfg.crop_size == (128, 128, 64)
    assert cfg.in_chan == 3 and cfg.padding == ((48, 48),) * 3
    assert cfg.dtype == np.float32 and cfg.output_dtype == np.float32
    assert len(cfg.all_crop_inds) == 2 and len(cfg.all_add_inds) == 2
    for idx in range(2):
        assert cfg._get_anchor(idx) == OS.get_anchor(idx, cfg.ndiv, cfg.crop_size)
        oc, oa = OS.compute_indices(idx, cfg.size, cfg.ndiv)
        for a, b in zip(cfg.all_crop_inds[idx][1:], oc[1:]):
            np.testing.assert_array_equal(a, b)
        for a, b in zip(cfg.all_add_inds[idx][1:], oa[1:]):
            np.testing.assert_array_equal(a, b)
    crop = cfg.all_crop_inds[0]
    assert crop[0] == slice(None) and crop[1].shape == (224, 1, 1) and crop[3].shape == (160,)
    assert crop[3].min() >= 0 and crop[3].max() < 64                           # wraps a 64-wide axis several times
    box = np.arange(3 * 256 * 128 * 64, dtype=np.float32).reshape(3, 256, 128, 64)
    assert box[crop].shape == (3, 224, 224, 160)


def test_subbox_config_floor_division_and_coverage():
    cfg = J.SubboxConfig(size=(100, 64, 64), ndiv=(3, 2, 2))
    assert cfg.crop_size == (33, 32, 32)                                       # subbox.py:49
    cover = np.zeros(cfg.size, np.int32)
    for add in cfg.all_add_inds:
        cover[add[1:]] += 1
    assert cover[:99].min() == 1 and cover[:99].max() == 1 and np.all(cover[99:] == 0)


# ---- factory / errors (tests/test_nbody_emulator.py:82-105, :381-410) -------------------------------------
def test_create_emulator_variants_and_errors():
    cfg = J.SubboxConfig(size=(128, 128, 128), ndiv=(1, 1, 1))
    table = {(False, True): J.StyleNBodyEmulatorVelCore, (False, False): J.StyleNBodyEmulatorCore,
             (True, True): J.NBodyEmulatorVelCore, (True, False): J.NBodyEmulatorCore}
    for (pre, vel), cls in table.items():
        e = J.create_emulator(premodulate=pre, compute_vel=vel, load_params=False, processor_config=cfg)
        assert type(e.model) is cls and e.params is None and e.premodulate == pre and e.compute_vel == vel
        assert e.processor.premodulate == pre and e.processor.compute_vel == vel and e.dtype == np.float32
    e = J.create_emulator(load_params=False, mid_chan=16)
    assert e.model.mid_chan == 16 and e.processor is None
    with pytest.raises(ValueError, match="No parameters loaded"):
        e.apply(np.zeros((1, 3, 128, 128, 128), np.float32), 0.0, 0.3)
    with pytest.raises(ValueError, match="No processor created"):
        e.process_box(np.zeros((3, 128, 128, 128), np.float32), 0.0, 0.3)
    with pytest.raises(ValueError, match="premodulate_z and premodulate_Om are required"):
        J.create_emulator(premodulate=True, load_params=True)
    with pytest.raises(ValueError, match="premodulate_z and premodulate_Om are required"):
        J.create_emulator(premodulate=True, load_params=True, premodulate_z=0.0)
    assert J.create_emulator(load_params=False, dtype=np.float16).dtype == np.float16
    cfg16 = J.SubboxConfig(size=(128, 128, 128), ndiv=(1, 1, 1), dtype=np.float16)
    assert J.create_emulator(load_params=False, processor_config=cfg16, dtype=np.float32).dtype == np.float16


# ---- cosmology (reference tests/test_cosmology.py pins + oracle) -------------------------------------------
def test_cosmology_matches_oracle_and_c_library():
    zs = np.array([0.0, 0.3, 0.5, 1.0, 2.0, 3.0])
    for Om in (0.1, 0.3, 0.5):
        for name in ("growth_factor", "hubble_rate", "growth_rate", "dlogH_dloga", "vel_norm", "acc_norm"):
            got = getattr(J, name)(zs, Om)
            assert got.dtype == np.float32 and got.shape == zs.shape
            np.testing.assert_allclose(got, getattr(OC, name)(zs, Om), rtol=3e-7)
    lib = _lib.lib()
    for z, Om in ((0.0, 0.3), (0.5, 0.3), (2.0, 0.12), (1.0, 0.5)):
        assert abs(lib.nbe_growth_factor(z, Om) / float(OC.growth_factor(z, Om)) - 1) < 1e-12
        assert abs(lib.nbe_vel_norm(z, Om) / float(OC.vel_norm(z, Om)) - 1) < 1e-12
    assert J.growth_factor(0.0, 0.3) == np.float32(1.0) and J.hubble_rate(0.0, 0.3) == np.float32(100.0)
    assert J.growth_factor(0.5, 0.3).shape == ()


# ---- premodulation walkers (nbody_emulator.py:150-266) -------------------------------------------------------
def test_premodulation_walkers_match_oracle():
    p = OP.synthetic_params(seed=5, mid_chan=8)
    pv = J.modulate_emulator_parameters_vel(p, 0.5, 0.3)
    ov = OP.premodulate_vel(p, 0.5, 0.3)
    pn = J.modulate_emulator_parameters(p, 0.5, 0.3)
    for blk in p["params"]:
        for lay in p["params"][blk]:
            a, b = pv["params"][blk][lay], ov["params"][blk][lay]
            assert sorted(a) == ["bias", "dweight", "weight"]
            np.testing.assert_allclose(a["weight"], b["weight"], rtol=2e-5, atol=2e-7)
            np.testing.assert_allclose(a["dweight"], b["dweight"], rtol=2e-4, atol=2e-6)
            assert sorted(pn["params"][blk][lay]) == ["bias", "weight"]
            np.testing.assert_allclose(pn["params"][blk][lay]["weight"], b["weight"], rtol=2e-5, atol=2e-7)
    # first-layer rule only on conv_l00/{conv_0, skip} (tests/test_nbody_emulator.py:644-664)
    s = OL.style_vector(0.3, float(OC.growth_factor(0.5, 0.3)))
    lp = p["params"]["conv_l00"]["conv_1"]
    _, dw_later = OL.modulate_weights_vel(lp["style_weight"], lp["style_bias"], lp["weight"], s, False)
    np.testing.assert_allclose(pv["params"]["conv_l00"]["conv_1"]["dweight"], dw_later, rtol=2e-4, atol=2e-6)
