"""Kink-aware parity checks of the velocity (test infrastructure; uses the CPU oracle).

The tangent of LeakyReLUVel jumps by a factor 100 at zero (reference layers_vel.py:184-185), so the velocity is a
DISCONTINUOUS function of the input: two floating-point evaluations differ wherever a pre-activation is zero to within
their rounding, and with ~10^9 activations behind a 128^3 input such activations always exist (the float32 NumPy oracle
itself differs from the float64 one in 0.6 % of the velocity voxels of BASELINE config 1).  Statistical allowances for
those voxels cannot tell a branch flip from a localised kernel bug.  These checks are causal instead:

  1. the library records which branch it took at EVERY LeakyReLU in the dependency cone of a block of output voxels
     (nbe_probe_*, include/nbe.h; Engine.probe_begin / probe_read);
  2. the float64 oracle is evaluated on the cone's input with THOSE branches for the tangent (oracle.model.forward_single,
     `branch_hook`; the primal follows the reference);
  3. the fields of the block must then agree with the oracle at the PLAIN tolerances on every voxel -- there is nothing
     left that may legitimately differ -- and
  4. every branch that differs from the oracle's own must belong to a pre-activation that is zero to within the
     rounding error of that tensor (|x| <= FLIP_TOL * RMS of the tensor): a flip anywhere else is an error.

The cone of an n^3 block is an (n + 96)^3 input (all-VALID U-Net, core :105-195); sub-box origins that are multiples of 8
keep the phase of the three stride-2 levels, so the cone can be cut out of any larger tile.
"""

import numpy as np

from oracle import layers as L, model as M

FLIP_TOL = 2e-5          # a legitimate flip has |pre-activation| <= FLIP_TOL * RMS(tensor): a tenth of the per-layer max-norm tolerance
                         # (measured at production width: <= 8.6e-7 f16x3, <= 1.9e-6 strict float32, 53 / 94 flips in 5.4e8 activations)


def cone_input_periodic(box, origin, nout):
    """Input of the cone of output block [origin, origin + nout)^3 of process_box: the box's periodic crop with the
    reference's 48-voxel padding (subbox.py:81-97)."""
    box = np.asarray(box)
    ix = [np.arange(o - 48, o + nout + 48) % s for o, s in zip(origin, box.shape[1:])]
    return np.ascontiguousarray(box[:, ix[0][:, None, None], ix[1][None, :, None], ix[2][None, None, :]])


def cone_input_valid(x, origin, nout):
    """... of model.apply on one padded input x (C, D, H, W): output voxel p reads x[p : p + 97)."""
    o = origin
    return np.ascontiguousarray(np.asarray(x)[:, o[0]:o[0] + nout + 96, o[1]:o[1] + nout + 96, o[2]:o[2] + nout + 96])


def oracle_cone(params, xin, Om, Dz, vel_fac, branches, premodulated=False, backend='torch'):
    """Float64 oracle on the cone input with the tangent branches of the evaluation under test.
    Returns disp, vel (3, n, n, n) and per-layer statistics {name: (flips, max |x| / rms at a flip, activations)}."""
    stats = {}
    seen = []

    def hook(name, x):
        b = branches[name]
        assert b.shape == x.shape, (name, b.shape, x.shape)
        flips = b != (x > 0)
        nf = int(np.count_nonzero(flips))
        rms = float(np.sqrt(np.mean(x * x)))
        stats[name] = (nf, float(np.abs(x[flips]).max() / rms) if nf else 0.0, int(x.size))
        seen.append(name)
        return b

    with L.backend(backend):
        d, v = M.forward_single(params, np.asarray(xin, dtype=np.float64), Om, Dz, vel_fac, premodulated, True,
                                np.float64, branch_hook=hook)
    assert sorted(seen) == sorted(branches), "the probe and the oracle disagree about the activations of the network"
    return d, v, stats


def assert_cone(tag, d_got, v_got, d_o, v_o, stats, rms_d=None, rms_v=None, tol_d=(2e-5, 2e-4), tol_v=(5e-5, 2e-4),
                flip_tol=FLIP_TOL, max_flip_frac=1e-3):
    """Plain tolerances on every voxel of the block (relative L2, max|delta| / RMS of the field), and every flipped branch
    at a pre-activation within flip_tol of zero.  Returns the measured figures."""
    d_got, v_got = np.asarray(d_got, np.float64), np.asarray(v_got, np.float64)
    assert d_got.shape == d_o.shape and v_got.shape == v_o.shape, (tag, d_got.shape, d_o.shape)
    assert np.all(np.isfinite(d_got)) and np.all(np.isfinite(v_got)), tag
    rms_d = float(np.sqrt(np.mean(d_o * d_o))) if rms_d is None else rms_d
    rms_v = float(np.sqrt(np.mean(v_o * v_o))) if rms_v is None else rms_v
    ed = (float(np.linalg.norm(d_got - d_o) / np.linalg.norm(d_o)), float(np.abs(d_got - d_o).max() / rms_d))
    ev = (float(np.linalg.norm(v_got - v_o) / np.linalg.norm(v_o)), float(np.abs(v_got - v_o).max() / rms_v))
    nflip = sum(s[0] for s in stats.values())
    nact = sum(s[2] for s in stats.values())
    worst = max(stats.items(), key=lambda kv: kv[1][1])
    print("%s: disp %.2e / %.2e  vel %.2e / %.2e (plain tolerances, every voxel); %d of %.2e branches differ from the "
          "oracle's own, largest |pre-activation| among them %.1e RMS (%s)" % (tag, *ed, *ev, nflip, nact, worst[1][1], worst[0]))
    assert ed[0] <= tol_d[0] and ed[1] <= tol_d[1], "%s: displacement %.3e / %.3e" % (tag, *ed)
    assert ev[0] <= tol_v[0] and ev[1] <= tol_v[1], "%s: velocity with the library's branches %.3e / %.3e" % (tag, *ev)
    assert worst[1][1] <= flip_tol, "%s: a branch flipped at |pre-activation| = %.2e RMS in %s" % (tag, worst[1][1], worst[0])
    assert nflip <= max_flip_frac * nact, "%s: %d of %d branches flipped" % (tag, nflip, nact)
    return {"disp": ed, "vel": ev, "flips": nflip, "activations": nact, "worst_flip": worst[1][1], "worst_layer": worst[0]}
