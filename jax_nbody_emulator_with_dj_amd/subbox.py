"""SubboxConfig / SubboxProcessor with the reference's names, fields and behaviour
(reference src/jax_nbody_emulator/subbox.py:25-233).

The index tables (`all_crop_inds`, `all_add_inds`) are kept because callers and tests
read them (tests/test_subbox.py:86-204), but the product path never gathers on the
CPU: `process_box` hands the whole box to libnbe.so once (nbe_process_box), which
crops with periodic wrap, runs the network and pastes on the GPU.
"""

from dataclasses import dataclass

import numpy as np

from . import cosmology
from . import engine as _engine
from . import models as _models
from .models import (NBodyEmulatorCore, NBodyEmulatorVelCore, StyleNBodyEmulatorCore,
                     StyleNBodyEmulatorVelCore)


@dataclass
class SubboxConfig:
    """
    Configuration for subbox processing.

    Attributes:
        size: Full box size (D, H, W)
        ndiv: Number of divisions along each dimension
        dtype: Precision of the model (float32 -> float32-equivalent engine; float16 -> float16 engine)
        output_dtype: Precision for output arrays (np.float16 or np.float32)
        in_chan: Number of input channels (default: 3 for displacement)
        padding: Padding on each side for each dimension
    """
    size: tuple
    ndiv: tuple
    dtype: object = np.float32
    output_dtype: object = np.float32
    in_chan: int = 3
    padding: tuple = ((48, 48), (48, 48), (48, 48))

    def __post_init__(self):
        self.NDIM = 3
        self.n_subboxes = np.prod(self.ndiv)
        self.crop_size = tuple(s // d for s, d in zip(self.size, self.ndiv))
        self.all_crop_inds = []
        self.all_add_inds = []
        for idx in range(self.n_subboxes):
            crop_inds, add_inds = self._compute_indices(idx)
            self.all_crop_inds.append(crop_inds)
            self.all_add_inds.append(add_inds)

    def _get_anchor(self, idx):
        """Anchor of sub-box `idx`: row-major over ndiv, last axis fastest."""
        return (
            (idx // (self.ndiv[1] * self.ndiv[2])) * self.crop_size[0],
            ((idx // self.ndiv[2]) % self.ndiv[1]) * self.crop_size[1],
            (idx % self.ndiv[2]) * self.crop_size[2],
        )

    def _compute_indices(self, idx):
        anchor = self._get_anchor(idx)
        crop_inds = self._get_crop_inds(anchor, self.crop_size, self.padding)
        add_inds = self._get_crop_inds(anchor, self.crop_size, ((0, 0),) * self.NDIM)
        return crop_inds, add_inds

    def _get_crop_inds(self, anchor, crop, pad):
        """Periodic index vectors, shaped for NumPy broadcasting (channel axis first)."""
        ind = [slice(None)]
        for d, (a, c, (p0, p1), s) in enumerate(zip(anchor, crop, pad, self.size)):
            i = np.arange(a - p0, a + c + p1) % s
            ind.append(i.reshape((-1,) + (1,) * (self.NDIM - d - 1)))
        return tuple(ind)


class SubboxProcessor:
    """
    Unified subbox processor for all model variants (reference subbox.py:99-233).

    `params` may be re-assigned after construction; the engine re-reads it lazily.
    """

    def __init__(self, model, params, config):
        self.model = model
        self.params = params
        self.config = config
        model_type = type(model)
        if model_type in (NBodyEmulatorCore, NBodyEmulatorVelCore):
            self.premodulate = True
        elif model_type in (StyleNBodyEmulatorCore, StyleNBodyEmulatorVelCore):
            self.premodulate = False
        else:
            raise TypeError("unsupported model type %r" % (model_type,))
        self.compute_vel = model_type in (NBodyEmulatorVelCore, StyleNBodyEmulatorVelCore)
        self.apply_fn = model.apply          # the reference stores jax.jit(model.apply) here

    def process_box(self, input_box, z, Om, desc="Processing subboxes", show_progress=True):
        """
        Process entire box through subboxes.

        Args:
            input_box: Input displacement field (C, D, H, W): NumPy array (host) or CUDA torch tensor
            z: Redshift
            Om: Omega_matter
            desc: Progress bar description
            show_progress: Whether to show progress bar

        Returns:
            If compute_vel=False: displacement (C, D, H, W)
            If compute_vel=True: (displacement, velocity) tuple
            NumPy arrays of config.output_dtype for NumPy input, CUDA tensors for tensor input.
        """
        cfg = self.config
        is_t = _engine._is_torch(input_box)
        if is_t and not input_box.is_cuda:
            input_box, is_t = input_box.numpy(), False
        device = (input_box.device.index or 0) if is_t else None
        try:
            return self._process_on(_models.get_engine(self.model, device, _models.precision_for(cfg.dtype)),
                                    input_box, is_t, z, Om, desc, show_progress)
        except _engine.NBERangeError as e:
            # an activation left the f16 range (include/nbe.h, "Range"): the strict float32 engine has float32's range
            import warnings
            warnings.warn("%s -- recomputing this box with the strict float32 engine" % e, RuntimeWarning)
            return self._process_on(_models.get_engine(self.model, device, "f32"), input_box, is_t, z, Om, desc,
                                    show_progress)

    def _process_on(self, eng, input_box, is_t, z, Om, desc, show_progress):
        cfg = self.config
        eng.ensure_params(self.params, self.premodulate)

        # cosmology once per box (subbox.py:173-178), float32 like the reference
        Dz = np.float32(cosmology.growth_factor(z, Om))
        vel_fac = np.float32(cosmology.vel_norm(z, Om)) if self.compute_vel else np.float32(0)
        if not self.premodulate:
            eng.set_cosmology(np.float32(Om), Dz)

        box = input_box
        cdt = np.dtype(cfg.dtype) if not _is_torch_dtype(cfg.dtype) else None
        if not is_t:
            box = np.asarray(box)
            if cdt is not None and cdt != np.float32:
                box = box.astype(cdt)            # the reference casts the crop to config.dtype (subbox.py:200-201)
            box = box.astype(np.float32, copy=False)

        bar = None
        cb = None
        if show_progress:
            try:
                from tqdm import tqdm
                bar = tqdm(total=int(cfg.n_subboxes), desc=desc, ncols=80,
                           bar_format='{desc}: {percentage:3.0f}%|{bar:30}| {n_fmt}/{total_fmt} [{elapsed}<{remaining}]')
                nsub = int(cfg.n_subboxes)      # the engine may merge sub-boxes into fewer, larger tiles
                cb = lambda done, total, user: bar.update(int(round(done * nsub / max(total, 1))) - bar.n)
            except Exception:
                bar = None
        try:
            out_np = np.dtype(cfg.output_dtype)
            eng_dtype = np.float16 if out_np == np.float16 else np.float32
            res = eng.process_box(box, cfg.size, cfg.ndiv, cfg.padding, Dz, vel_fac, out_dtype=eng_dtype, progress=cb)
        finally:
            if bar is not None:
                bar.close()
        if is_t:
            return res
        fin = lambda a: a.astype(out_np, copy=False)
        if cdt is not None and cdt != np.float32 and out_np == np.float32:
            fin = lambda a: a.astype(cdt).astype(out_np)     # results of a reduced-precision model, widened
        if self.compute_vel:
            return fin(res[0]), fin(res[1])
        return fin(res)

    def _apply_model(self, x, Om, Dz, vel_fac):
        """Dispatch to model with correct signature (reference subbox.py:221-233)."""
        if self.premodulate:
            if self.compute_vel:
                return self.apply_fn(self.params, x, Dz, vel_fac)
            return self.apply_fn(self.params, x, Dz)
        if self.compute_vel:
            return self.apply_fn(self.params, x, Om, Dz, vel_fac)
        return self.apply_fn(self.params, x, Om, Dz)


def _is_torch_dtype(dt):
    return _engine.torch is not None and isinstance(dt, _engine.torch.dtype)
