"""Factory and bundle with the reference's names, defaults and error behaviour
(reference src/jax_nbody_emulator/nbody_emulator.py:23-384).

`modulate_emulator_parameters[_vel]` return the same trees as the reference's walkers
(:150-187, :221-266) so existing user code keeps working; they are host-side float32
NumPy (a one-off 13 MB of weights).  The per-call modulation of the Style* models runs
in the HIP library instead (nbe_set_cosmology).
"""

from dataclasses import dataclass
from pathlib import Path

import numpy as np

from .cosmology import growth_factor, vel_norm
from .subbox import SubboxConfig, SubboxProcessor


@dataclass
class NBodyEmulator:
    """
    Container for emulator components with convenient access methods.

    Attributes:
        model: The underlying model
        params: Model parameters (None if not loaded)
        processor: SubboxProcessor for large volumes (None if not created)
        premodulate: Whether params were premodulated (True=fixed cosmology, False=runtime cosmology)
        compute_vel: Whether model returns velocity field, default=True
    """
    model: object
    params: object
    processor: object
    premodulate: bool = False
    compute_vel: bool = True
    dtype: object = np.float32

    def apply(self, x, z, Om):
        """Apply model directly to input tensor (B, C, D, H, W)."""
        if self.params is None:
            raise ValueError("No parameters loaded. Use load_params=True in create_emulator.")
        z = np.atleast_1d(np.asarray(z, dtype=np.float32))
        Om = np.atleast_1d(np.asarray(Om, dtype=np.float32))
        Dz = growth_factor(z, Om)
        if self.compute_vel:
            vel_fac = vel_norm(z, Om)
        if hasattr(x, 'astype'):
            x = x.astype(self.dtype)
        if self.premodulate:
            if self.compute_vel:
                return self.model.apply(self.params, x, Dz, vel_fac)
            return self.model.apply(self.params, x, Dz)
        if self.compute_vel:
            return self.model.apply(self.params, x, Om, Dz, vel_fac)
        return self.model.apply(self.params, x, Om, Dz)

    def process_box(self, input_box, z, Om, desc="Processing subboxes", show_progress=True):
        """Process large volume through subbox decomposition."""
        if self.processor is None:
            raise ValueError("No processor created. Use create_processor=True in create_emulator.")
        return self.processor.process_box(input_box, z, Om, desc=desc, show_progress=show_progress)

    def __call__(self, x, z, Om):
        """Alias for apply()."""
        return self.apply(x, z, Om)


def load_default_parameters():
    """
    Load default pretrained model parameters (reference nbody_emulator.py:115-129).

    The reference ships them as an .npz holding a pickled nested dict.  The blob is not part of
    this repository (it is absent from the reference checkout too); place it -- as it is, or converted to the
    pickle-free flat format with `python -m jax_nbody_emulator_with_dj_amd.params_io` -- at
    `jax_nbody_emulator_with_dj_amd/model_parameters/nbody_emulator_params.npz`
    (or point the environment variable NBE_PARAMS at it).  Either format is read without executing anything
    from the file (params_io.py).
    """
    import os
    params_path = Path(os.environ.get("NBE_PARAMS") or
                       Path(__file__).parent / "model_parameters" / "nbody_emulator_params.npz")
    if not params_path.exists():
        raise FileNotFoundError(
            "pretrained parameters not found at %s (the blob is not distributed with this repository); "
            "use create_emulator(load_params=False) and assign emulator.params / processor.params" % params_path)
    from .params_io import load_parameters
    return load_parameters(params_path)                    # flat .npz or the reference's pickled dict; no code is executed


def _style_vector(z, Om):
    Dz = np.float32(growth_factor(z, Om))
    return np.array([(np.float32(Om) - np.float32(0.3)) * np.float32(5.0), Dz - np.float32(1.0)], dtype=np.float32)


def _modulate_weights(style_weight, style_bias, weight, s, eps=1.e-8):
    sw, sb, w0 = (np.asarray(a, dtype=np.float32) for a in (style_weight, style_bias, weight))
    s_mod = sw @ s + sb
    w = w0 * s_mod[None, :, None, None, None]
    norm = np.sqrt(np.sum(w ** 2, axis=(1, 2, 3, 4), keepdims=True) + np.float32(eps))
    return w / norm


def _modulate_weights_vel(style_weight, style_bias, weight, s, dx=None, eps=1.e-8):
    sw, sb, w0 = (np.asarray(a, dtype=np.float32) for a in (style_weight, style_bias, weight))
    s_mod = sw @ s + sb
    ds_mod = sw[:, 1]
    w = w0 * s_mod[None, :, None, None, None]
    dw_style = w0 * ds_mod[None, :, None, None, None]
    norm = np.sqrt(np.sum(w ** 2, axis=(1, 2, 3, 4), keepdims=True) + np.float32(eps))
    dnorm = -np.sum(w * dw_style, axis=(1, 2, 3, 4), keepdims=True) / (norm ** 3)
    w_n = w / norm
    dw_n = dw_style / norm + w * dnorm
    if dx is None:      # first layer: input is linear in Dz
        dw_n = dw_n + w_n / (s[1] + np.float32(1.0))
    return w_n, dw_n


def modulate_emulator_parameters(params, z, Om, eps=1.e-8):
    """Preprocess all network parameters for fixed (z, Om); returns {'params': {block: {layer: {weight, bias}}}}."""
    s = _style_vector(z, Om)
    out = {'params': {}}
    for block_name, block_params in params['params'].items():
        out['params'][block_name] = {}
        for layer_name, lp in block_params.items():
            if 'style_weight' in lp:
                w = _modulate_weights(lp['style_weight'], lp['style_bias'], lp['weight'], s, eps=eps)
                out['params'][block_name][layer_name] = {'weight': w, 'bias': np.asarray(lp['bias'], dtype=np.float32)}
            else:
                print(f'skipping {block_name} {layer_name}')
                out['params'][block_name][layer_name] = lp
    return out


def modulate_emulator_parameters_vel(params, z, Om, eps=1.e-8):
    """As above with `dweight` = d(weight)/dDz; conv_l00/{conv_0,skip} take the first-layer rule."""
    s = _style_vector(z, Om)
    out = {'params': {}}
    for block_name, block_params in params['params'].items():
        out['params'][block_name] = {}
        for layer_name, lp in block_params.items():
            if 'style_weight' in lp:
                first = block_name == 'conv_l00' and layer_name in ('conv_0', 'skip')
                w, dw = _modulate_weights_vel(lp['style_weight'], lp['style_bias'], lp['weight'], s,
                                              dx=None if first else 1, eps=eps)
                out['params'][block_name][layer_name] = {'weight': w, 'dweight': dw,
                                                         'bias': np.asarray(lp['bias'], dtype=np.float32)}
            else:
                print(f'skipping {block_name} {layer_name}')
                out['params'][block_name][layer_name] = lp
    return out


def create_emulator(premodulate=False, compute_vel=True, load_params=True, processor_config=None,
                    premodulate_z=None, premodulate_Om=None, dtype=None, **model_kwargs):
    """
    Factory function to create emulator, optionally with params and processor
    (same arguments, defaults and errors as the reference, nbody_emulator.py:268-384).
    """
    from .models import (NBodyEmulatorCore, NBodyEmulatorVelCore, StyleNBodyEmulatorCore,
                         StyleNBodyEmulatorVelCore)
    if premodulate:
        model = NBodyEmulatorVelCore(**model_kwargs) if compute_vel else NBodyEmulatorCore(**model_kwargs)
    else:
        model = StyleNBodyEmulatorVelCore(**model_kwargs) if compute_vel else StyleNBodyEmulatorCore(**model_kwargs)

    params = None
    if load_params:
        # checked before touching the file so the argument error does not depend on the blob being present
        if premodulate and (premodulate_z is None or premodulate_Om is None):
            raise ValueError(
                "premodulate_z and premodulate_Om are required "
                "when premodulate=True and load_params=True"
            )
        params = load_default_parameters()
        if premodulate:
            if compute_vel:
                params = modulate_emulator_parameters_vel(params, premodulate_z, premodulate_Om)
            else:
                params = modulate_emulator_parameters(params, premodulate_z, premodulate_Om)

    processor = None
    if processor_config is not None:
        processor = SubboxProcessor(model, params, processor_config)

    if processor_config is not None:
        dtype = processor_config.dtype
    elif dtype is None:
        dtype = np.float32

    return NBodyEmulator(model=model, params=params, processor=processor, premodulate=premodulate,
                         compute_vel=compute_vel, dtype=dtype)
