"""Flat-LambdaCDM scalars with the reference's names and signatures
(reference src/jax_nbody_emulator/cosmology.py:34-155).

Host-side NumPy; no SciPy/JAX dependency: 2F1(1, 1/3; 11/6; x) is summed as a power
series after the same Pfaff transformation the reference uses for x < 0
(cosmology.py:24-31).  The reference differentiates with jax.jvp; here the
derivatives are analytic.  Inputs may be scalars or arrays; results are float32
arrays of the broadcast shape (JAX's default precision), computed in float64.
The same series is implemented in C inside libnbe.so (nbe_growth_factor, nbe_vel_norm).
"""

import numpy as np

_A, _B, _C = 1.0, 1.0 / 3.0, 11.0 / 6.0


def _series(a, b, c, z):
    z = np.asarray(z, dtype=np.float64)
    term = np.ones_like(z)
    total = np.ones_like(z)
    for n in range(200000):
        term = term * ((a + n) * (b + n) / ((c + n) * (n + 1.0))) * z
        total = total + term
        if np.all(np.abs(term) <= 1e-17 * np.abs(total)):
            break
    return total


def _hyp2f1(a, b, c, x):
    x = np.asarray(x, dtype=np.float64)
    neg = x < 0
    xs = np.where(neg, x, 0.0)
    pf = np.power(1.0 - xs, -a) * _series(a, c - b, c, xs / (xs - 1.0))      # Pfaff, x < 0
    ps = _series(a, b, c, np.where(neg, 0.0, x))
    return np.where(neg, pf, ps)


def _f64(z, Om):
    z, Om = np.broadcast_arrays(np.asarray(z, dtype=np.float64), np.asarray(Om, dtype=np.float64))
    return z, Om


def _growth_factor64(z, Om):
    a = 1.0 / (1.0 + z)
    OL = 1.0 - Om
    return a * _hyp2f1(_A, _B, _C, -OL * a ** 3 / Om) / _hyp2f1(_A, _B, _C, -OL / Om)


def _hubble64(z, Om):
    return 100.0 * np.sqrt(Om * (1.0 + z) ** 3 + (1.0 - Om))


def _growth_rate64(z, Om):
    a = 1.0 / (1.0 + z)
    x = -(1.0 - Om) * a ** 3 / Om
    F = _hyp2f1(_A, _B, _C, x)
    dF = (_A * _B / _C) * _hyp2f1(_A + 1.0, _B + 1.0, _C + 1.0, x)
    return 1.0 + 3.0 * x * dF / F


def _dlogH_dloga64(z, Om):
    E2 = Om * (1.0 + z) ** 3 + (1.0 - Om)
    return -1.5 * Om * (1.0 + z) ** 3 / E2


def _out(v):
    return np.asarray(v, dtype=np.float32)


def growth_factor(z, Om):
    """Linear growth function for flat LambdaCDM, normalized to 1 at redshift zero."""
    return _out(_growth_factor64(*_f64(z, Om)))


def hubble_rate(z, Om):
    """Hubble parameter in [h km/s/Mpc] for flat LambdaCDM."""
    return _out(_hubble64(*_f64(z, Om)))


def growth_rate(z, Om):
    """Linear growth rate f = d log D / d log a."""
    return _out(_growth_rate64(*_f64(z, Om)))


def dlogH_dloga(z, Om):
    """Log-log derivative of Hubble w.r.t. scale factor."""
    return _out(_dlogH_dloga64(*_f64(z, Om)))


def vel_norm(z, Om):
    """Velocity normalization factor [km/s]: D f H / (1+z)."""
    z, Om = _f64(z, Om)
    return _out(_growth_factor64(z, Om) * _growth_rate64(z, Om) * _hubble64(z, Om) / (1.0 + z))


def acc_norm(z, Om):
    """Acceleration normalization factor [km/s^2]."""
    z, Om = _f64(z, Om)
    return _out(_growth_factor64(z, Om) * _growth_rate64(z, Om) * _hubble64(z, Om) ** 2
                * _dlogH_dloga64(z, Om) / (1.0 + z))
