#!/usr/bin/env python
"""Batch driver: many (cosmology, displacement) file pairs through one resident engine.

Same command line, file formats, range checks and output names as the reference's
`examples/run_jax_emulator.py` (:196-262 arguments, :117-139 cosmology files and ranges,
:265-355 loop):

    python -m jax_nbody_emulator_with_dj_amd.run_emulator \\
        --cosmo_param_files '/path/to/sims/*/params.npy' \\
        --displacement_files '/path/to/sims/*/dis.npy' \\
        --output_dirs '/path/to/sims/*/' \\
        --ndiv 4,2,2 --precision f16 --vel

    cosmology file   (6,)  [Omega_m, Omega_b, h, n_s, sigma_8, redshift]; Omega_m in [0.1, 0.5], z in [0, 3]
    displacement     (3, N0, N1, N2) z = 0 linear (ZA) displacement field
    outputs          <output_dir>/emu_dis.npy  and, with --vel, <output_dir>/emu_vel.npy

What differs from the reference: the engine, its weights and its ~100 GB workspace stay resident on the
GPU for the whole batch, and disk I/O overlaps compute -- the next displacement file is read and the
previous results are written by a background thread while the GPU works on the current box.
`--params FILE.npz` names the parameter tree ({'params': {...}} saved with np.savez, the reference's own
format); without it the default blob is looked up as in `load_default_parameters()`.
"""

import argparse
import sys
import time
from concurrent.futures import ThreadPoolExecutor
from glob import glob
from pathlib import Path

import numpy as np

OM_RANGE = (0.1, 0.5)
Z_RANGE = (0.0, 3.0)


def _die(msg):
    sys.exit(msg)


def _check_file(path):
    if not path.is_file():
        _die(f'Input file path is not a readable file: {path}')
    try:
        with path.open('rb'):
            pass
    except Exception as e:
        _die(f'Input file cannot be read: {path} ({e})')


def _check_dir(path):
    if not path.is_dir():
        _die(f'Output directory path is not a directory: {path}')
    probe = path / '.write_test'
    try:
        with probe.open('w'):
            pass
        probe.unlink()
    except Exception as e:
        _die(f'Output directory is not writable: {path} ({e})')


def files_matching(pattern):
    paths = sorted(Path(p) for p in glob(pattern))
    if not paths:
        raise argparse.ArgumentTypeError(f'No files match pattern: {pattern}')
    for p in paths:
        _check_file(p)
    return paths


def dirs_matching(pattern):
    paths = sorted(Path(p) for p in glob(pattern))
    if not paths:
        raise argparse.ArgumentTypeError(f'No directories match pattern: {pattern}')
    for p in paths:
        _check_dir(p)
    return paths


def divisions(text):
    """'4' -> (4,4,4); '2,4,4' or '(2, 4, 4)' -> (2,4,4)."""
    vals = [int(t) for t in text.strip('()').split(',')]
    if len(vals) == 1:
        return (vals[0],) * 3
    if len(vals) == 3:
        return tuple(vals)
    raise argparse.ArgumentTypeError(f'Expected 1 or 3 values, got {len(vals)}')


def precision(text):
    table = {'f16': np.float16, 'f32': np.float32}
    if text not in table:
        raise argparse.ArgumentTypeError(f"precision must be 'f32' or 'f16', got '{text}'")
    return table[text]


def read_cosmology(path):
    """(Omega_m, z) from a (6,) array [Om, Ob, h, ns, s8, z], with the reference's validity ranges."""
    data = np.load(path)
    Om, z = float(data[0]), float(data[-1])
    if not OM_RANGE[0] <= Om <= OM_RANGE[1]:
        _die(f'in file {path}: Om={Om:.4f} out of valid range [0.1, 0.5]')
    if not Z_RANGE[0] <= z <= Z_RANGE[1]:
        _die(f'in file {path}: z={z:.4f} out of valid range [0.0, 3.0]')
    return Om, z


def displacement_shape(path, expected):
    shape = np.load(path, mmap_mode='r').shape
    if len(shape) != 4:
        _die(f'in file {path}: input array ndim {len(shape)} is not 4')
    if shape[0] != 3:
        _die(f'in file {path}: first dimension {shape[0]} is not 3 (expected 3 displacement components)')
    if expected is not None and shape != expected:
        _die(f'in file {path}: input array shape {shape} differs from first file shape {expected}')
    return shape


def build_parser():
    ap = argparse.ArgumentParser(
        description='Batch process displacement fields with the MI355X N-body emulator engine.',
        formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument('--cosmo_param_files', type=files_matching, required=True,
                    help='Glob pattern for cosmology parameter files (numpy arrays with [Om, Ob, h, ns, s8, z])')
    ap.add_argument('--displacement_files', type=files_matching, required=True,
                    help='Glob pattern for input displacement files (numpy arrays with shape [3, N, N, N])')
    ap.add_argument('--output_dirs', type=dirs_matching, required=True, help='Glob pattern for output directories')
    ap.add_argument('--ndiv', type=divisions, required=True,
                    help='Number of subbox divisions: single int (e.g., 4) or tuple (e.g., 2,4,4)')
    ap.add_argument('--vel', action=argparse.BooleanOptionalAction, default=True,
                    help='Compute velocity field in addition to displacement (default: True)')
    ap.add_argument('--style', action=argparse.BooleanOptionalAction, default=True,
                    help='Use style modulation for flexible cosmology; if False, premodulate parameters '
                         'for each cosmology (default: True)')
    ap.add_argument('--precision', type=precision, default=np.float32,
                    help='Model precision: f16 (half) or f32 (full) (default: f32)')
    ap.add_argument('--output-precision', type=precision, default=np.float16, dest='output_precision',
                    help='Output file precision: f16 (half) or f32 (full) (default: f16).')
    ap.add_argument('--quiet', '-q', action='store_true', help='Suppress progress bars (useful for batch jobs)')
    ap.add_argument('--params', type=Path, default=None,
                    help="Parameter tree: flat .npz (block/layer/leaf arrays, see params_io.py) or the reference's "
                         ".npz with a pickled {'params': ...} dict (read with a restricted unpickler); "
                         'default: the packaged pretrained blob')
    return ap


def load_params(path):
    from .nbody_emulator import load_default_parameters
    if path is None:
        return load_default_parameters()
    from .params_io import load_parameters
    return load_parameters(path)                       # flat .npz or the reference's format; nothing is executed


def run(args):
    from . import create_emulator, SubboxConfig
    from . import modulate_emulator_parameters, modulate_emulator_parameters_vel

    n = len(args.cosmo_param_files)
    if not (n == len(args.displacement_files) == len(args.output_dirs)):
        _die('Number of files must match:\n'
             f'  cosmo_param_files: {n}\n'
             f'  displacement_files: {len(args.displacement_files)}\n'
             f'  output_dirs: {len(args.output_dirs)}')
    print(f'Processing {n} simulation(s)')
    print(f'  Precision: {args.precision}')
    print(f'  Output precision: {args.output_precision}')
    print(f'  Compute velocity: {args.vel}')
    print(f'  Style modulation: {args.style}')
    print(f'  Subbox divisions: {args.ndiv}')
    print()

    shape = None
    for f in args.displacement_files:
        shape = displacement_shape(f, shape)
    box = tuple(shape[1:])
    print(f'  Box size: {box}')
    cosmologies = [read_cosmology(f) for f in args.cosmo_param_files]

    params = load_params(args.params)
    mid = int(params['params']['conv_l01']['conv_0']['weight'].shape[0])
    config = SubboxConfig(size=box, ndiv=args.ndiv, dtype=args.precision, output_dtype=args.output_precision)
    emu = create_emulator(premodulate=not args.style, compute_vel=args.vel, load_params=False,
                          processor_config=config, mid_chan=mid)
    if args.style:
        emu.params = emu.processor.params = params

    def save(out_dir, result):
        if args.vel:
            np.save(out_dir / 'emu_dis.npy', result[0])
            np.save(out_dir / 'emu_vel.npy', result[1])
        else:
            np.save(out_dir / 'emu_dis.npy', result)

    # one reader and one writer thread: disk I/O of the neighbours overlaps the GPU work on the current box
    with ThreadPoolExecutor(max_workers=2) as pool:
        nxt = pool.submit(np.load, args.displacement_files[0])
        pending = None
        for i, (cosmo, out_dir) in enumerate(zip(cosmologies, args.output_dirs)):
            Om, z = cosmo
            dis_in = nxt.result()
            if i + 1 < n:
                nxt = pool.submit(np.load, args.displacement_files[i + 1])
            if not args.style:
                tree = (modulate_emulator_parameters_vel if args.vel else modulate_emulator_parameters)(params, z, Om)
                emu.params = emu.processor.params = tree
            t0 = time.time()
            result = emu.process_box(dis_in, z=z, Om=Om, show_progress=not args.quiet)
            dt = time.time() - t0
            if pending is not None:
                pending.result()
            pending = pool.submit(save, out_dir, result)
            print(f'[{i + 1}/{n}] z={z:.4f}, Om={Om:.4f}: {dt:.2f}s -> {out_dir}')
        if pending is not None:
            pending.result()
    print('\nDone!')


def main(argv=None):
    run(build_parser().parse_args(argv))


if __name__ == '__main__':
    main()
