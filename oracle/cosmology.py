"""Oracle: flat-LambdaCDM scalars (test infrastructure only).

Follows reference `src/jax_nbody_emulator/cosmology.py`:
  _growth_2f1     :24-31   2F1(1, 1/3; 11/6; x), Pfaff form for x < 0
  growth_factor   :34-40   D = a F(-OL a^3/Om) / F(-OL/Om)
  hubble_rate     :43-46   H = 100 sqrt(Om (1+z)^3 + OL)
  growth_rate     :101-113 f = -(1+z) dlnD/dz   (reference: jax.jvp)
  dlogH_dloga     :116-127
  vel_norm        :130-141 D f H / (1+z)
  acc_norm        :144-155 D f H^2 dlogH_dloga / (1+z)

The reference differentiates with jax.jvp; here the derivative of
2F1 is taken analytically, d/dx 2F1(a,b;c;x) = (ab/c) 2F1(a+1,b+1;c+1;x),
in float64 with scipy.special.hyp2f1.
"""

import numpy as np
from scipy.special import hyp2f1

_A, _B, _C = 1.0, 1.0 / 3.0, 11.0 / 6.0


def _growth_2f1(x):
    x = np.asarray(x, dtype=np.float64)
    return hyp2f1(_A, _B, _C, x)


def _dgrowth_2f1(x):
    x = np.asarray(x, dtype=np.float64)
    return (_A * _B / _C) * hyp2f1(_A + 1.0, _B + 1.0, _C + 1.0, x)


def growth_factor(z, Om):
    z = np.asarray(z, dtype=np.float64)
    Om = np.asarray(Om, dtype=np.float64)
    a = 1.0 / (1.0 + z)
    OL = 1.0 - Om
    aa3 = -OL * a ** 3 / Om
    aa30 = -OL / Om
    return a * _growth_2f1(aa3) / _growth_2f1(aa30)


def hubble_rate(z, Om):
    z = np.asarray(z, dtype=np.float64)
    Om = np.asarray(Om, dtype=np.float64)
    return 100.0 * np.sqrt(Om * (1.0 + z) ** 3 + (1.0 - Om))


def growth_rate(z, Om):
    """f = dlnD/dlna = 1 + 3 x F'(x)/F(x), x = -OL a^3 / Om."""
    z = np.asarray(z, dtype=np.float64)
    Om = np.asarray(Om, dtype=np.float64)
    a = 1.0 / (1.0 + z)
    x = -(1.0 - Om) * a ** 3 / Om
    return 1.0 + 3.0 * x * _dgrowth_2f1(x) / _growth_2f1(x)


def dlogH_dloga(z, Om):
    z = np.asarray(z, dtype=np.float64)
    Om = np.asarray(Om, dtype=np.float64)
    E2 = Om * (1.0 + z) ** 3 + (1.0 - Om)
    return -1.5 * Om * (1.0 + z) ** 3 / E2


def vel_norm(z, Om):
    z = np.asarray(z, dtype=np.float64)
    return growth_factor(z, Om) * growth_rate(z, Om) * hubble_rate(z, Om) / (1.0 + z)


def acc_norm(z, Om):
    z = np.asarray(z, dtype=np.float64)
    return (growth_factor(z, Om) * growth_rate(z, Om) * hubble_rate(z, Om) ** 2
            * dlogH_dloga(z, Om) / (1.0 + z))
