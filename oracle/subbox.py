"""Oracle: sub-box index tables and the serial process_box loop (test infrastructure only).

Follows reference `src/jax_nbody_emulator/subbox.py`:
  SubboxConfig.__post_init__  :45-58   crop_size = size // ndiv, tables for every idx
  _get_anchor                 :60-66   row-major over ndiv, last axis fastest
  _compute_indices            :68-79
  _get_crop_inds              :81-97   arange(a - p0, a + c + p1) % size, broadcast shapes
  SubboxProcessor.process_box :139-219 gather -> model -> ASSIGN into the outputs
"""

import numpy as np

from . import cosmology
from . import model as M

DEFAULT_PADDING = ((48, 48), (48, 48), (48, 48))


def crop_size(size, ndiv):
    return tuple(s // d for s, d in zip(size, ndiv))


def get_anchor(idx, ndiv, csize):
    return ((idx // (ndiv[1] * ndiv[2])) * csize[0],
            ((idx // ndiv[2]) % ndiv[1]) * csize[1],
            (idx % ndiv[2]) * csize[2])


def get_crop_inds(anchor, crop, pad, size):
    ind = [slice(None)]
    for d, (a, c, (p0, p1), s) in enumerate(zip(anchor, crop, pad, size)):
        i = np.arange(a - p0, a + c + p1) % s
        ind.append(i.reshape((-1,) + (1,) * (3 - d - 1)))
    return tuple(ind)


def compute_indices(idx, size, ndiv, padding=DEFAULT_PADDING):
    cs = crop_size(size, ndiv)
    anchor = get_anchor(idx, ndiv, cs)
    crop_inds = get_crop_inds(anchor, cs, padding, size)
    add_inds = get_crop_inds(anchor, cs, ((0, 0),) * 3, size)
    return crop_inds, add_inds


def process_box(params, input_box, z, Om, size, ndiv, premodulated=False, compute_vel=True,
                padding=DEFAULT_PADDING, dtype=np.float64, output_dtype=np.float64, eps=1e-8,
                in_chan=3):
    """subbox.py:139-219.  Cosmology scalars are computed once; outputs are
    assigned (not accumulated); trailing voxels stay zero when size % ndiv != 0."""
    size = tuple(size)
    dis_out = np.zeros((in_chan,) + size, dtype=output_dtype)
    vel_out = np.zeros((in_chan,) + size, dtype=output_dtype) if compute_vel else None
    Dz = float(cosmology.growth_factor(z, Om))
    vel_fac = float(cosmology.vel_norm(z, Om)) if compute_vel else None
    for idx in range(int(np.prod(ndiv))):
        crop_inds, add_inds = compute_indices(idx, size, ndiv, padding)
        x = np.asarray(input_box[crop_inds], dtype=dtype)[None]
        res = M.forward(params, x, None if premodulated else Om, Dz, vel_fac,
                        premodulated=premodulated, compute_vel=compute_vel, dtype=dtype, eps=eps)
        if compute_vel:
            dis_out[add_inds] = res[0][0].astype(output_dtype)
            vel_out[add_inds] = res[1][0].astype(output_dtype)
        else:
            dis_out[add_inds] = res[0].astype(output_dtype)
    return (dis_out, vel_out) if compute_vel else dis_out
