"""Batch driver (reference examples/run_jax_emulator.py:117-139, :196-355): argument parsing, file-format
validation and ranges on the CPU; an end-to-end run over two tiny simulations on the GPU."""

import argparse

import numpy as np
import pytest

from jax_nbody_emulator_with_dj_amd import run_emulator as R


def test_divisions_and_precision_parsing():
    assert R.divisions("4") == (4, 4, 4)
    assert R.divisions("2,4,4") == (2, 4, 4) and R.divisions("(2, 4, 4)") == (2, 4, 4)
    with pytest.raises(argparse.ArgumentTypeError):
        R.divisions("2,4")
    assert R.precision("f16") is np.float16 and R.precision("f32") is np.float32
    with pytest.raises(argparse.ArgumentTypeError):
        R.precision("bf16")
    ap = R.build_parser()
    for opt in ("--cosmo_param_files", "--displacement_files", "--output_dirs", "--ndiv", "--vel", "--no-vel",
                "--style", "--no-style", "--precision", "--output-precision", "--quiet", "-q"):
        assert any(opt in a.option_strings for a in ap._actions), opt


def test_file_validation(tmp_path):
    good = tmp_path / "params.npy"
    np.save(good, np.array([0.3, 0.05, 0.7, 0.96, 0.8, 0.5]))
    assert R.read_cosmology(good) == (0.3, 0.5)
    for bad in ([0.05, 0, 0, 0, 0, 0.5], [0.6, 0, 0, 0, 0, 0.5], [0.3, 0, 0, 0, 0, 3.5], [0.3, 0, 0, 0, 0, -0.1]):
        f = tmp_path / "bad.npy"
        np.save(f, np.array(bad, dtype=float))
        with pytest.raises(SystemExit, match="out of valid range"):
            R.read_cosmology(f)
    d = tmp_path / "dis.npy"
    np.save(d, np.zeros((3, 8, 8, 8), np.float32))
    assert R.displacement_shape(d, None) == (3, 8, 8, 8)
    with pytest.raises(SystemExit, match="differs from first file shape"):
        R.displacement_shape(d, (3, 16, 8, 8))
    np.save(d, np.zeros((2, 8, 8, 8), np.float32))
    with pytest.raises(SystemExit, match="is not 3"):
        R.displacement_shape(d, None)
    np.save(d, np.zeros((3, 8, 8), np.float32))
    with pytest.raises(SystemExit, match="ndim 3 is not 4"):
        R.displacement_shape(d, None)
    with pytest.raises(argparse.ArgumentTypeError, match="No files match"):
        R.files_matching(str(tmp_path / "nothing*.npy"))
    assert R.dirs_matching(str(tmp_path)) == [tmp_path]
    # mismatching numbers of files
    with pytest.raises(SystemExit, match="Number of files must match"):
        R.run(argparse.Namespace(cosmo_param_files=[good], displacement_files=[d, d], output_dirs=[tmp_path],
                                 ndiv=(1, 1, 1), precision=np.float32, output_precision=np.float16, vel=True,
                                 style=True, quiet=True, params=None))


@pytest.mark.gpu
@pytest.mark.parametrize("style", [True, False])
def test_batch_run_end_to_end(tmp_path, style):
    import jax_nbody_emulator_with_dj_amd as J
    from oracle import params as P
    p = P.synthetic_params(seed=51, mid_chan=8)
    np.savez(tmp_path / "weights.npz", params=p["params"])
    rng = np.random.default_rng(52)
    boxes, cosmos = [], [(0.3, 0.5), (0.2, 1.5)]
    for i, (Om, z) in enumerate(cosmos):
        sim = tmp_path / ("sim%d" % i)
        sim.mkdir()
        np.save(sim / "params.npy", np.array([Om, 0.05, 0.7, 0.96, 0.8, z]))
        b = rng.standard_normal((3, 16, 8, 8)).astype(np.float32)
        np.save(sim / "dis.npy", b)
        boxes.append(b)
    argv = ["--cosmo_param_files", str(tmp_path / "sim*/params.npy"), "--displacement_files", str(tmp_path / "sim*/dis.npy"),
            "--output_dirs", str(tmp_path / "sim*/"), "--ndiv", "2,1,1", "--quiet", "--params", str(tmp_path / "weights.npz")]
    if not style:
        argv.append("--no-style")
    R.main(argv)
    cfg = J.SubboxConfig(size=(16, 8, 8), ndiv=(2, 1, 1))
    emu = J.create_emulator(load_params=False, processor_config=cfg, mid_chan=8)
    emu.processor.params = p
    for i, (Om, z) in enumerate(cosmos):
        dis = np.load(tmp_path / ("sim%d" % i) / "emu_dis.npy")
        vel = np.load(tmp_path / ("sim%d" % i) / "emu_vel.npy")
        assert dis.dtype == np.float16 and dis.shape == (3, 16, 8, 8) and vel.shape == dis.shape
        d_ref, v_ref = emu.process_box(boxes[i], z, Om, show_progress=False)
        np.testing.assert_allclose(dis.astype(np.float32), d_ref, rtol=2e-3, atol=2e-3)
        np.testing.assert_allclose(vel.astype(np.float32), v_ref, rtol=2e-3, atol=5e-2)
