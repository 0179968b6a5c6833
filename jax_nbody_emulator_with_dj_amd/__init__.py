"""MI355X-native N-body emulator: the reference's Python API over hand-written HIP (gfx950).

Quick start (same as the reference `jax_nbody_emulator/__init__.py:4-20`):

    from jax_nbody_emulator_with_dj_amd import create_emulator, SubboxConfig

    config = SubboxConfig(size=(512, 512, 512), ndiv=(4, 4, 4))
    emulator = create_emulator(processor_config=config)
    displacement, velocity = emulator.process_box(input_box, z=0.0, Om=0.3)

Exports mirror reference `src/jax_nbody_emulator/__init__.py:30-47, :73-95`.
Everything numerical runs in libnbe.so (include/nbe.h); there is no CPU fallback.
"""

from .nbody_emulator import (
    NBodyEmulator,
    create_emulator,
    load_default_parameters,
    modulate_emulator_parameters,
    modulate_emulator_parameters_vel,
)
from .subbox import SubboxConfig, SubboxProcessor
from .cosmology import growth_factor, hubble_rate, growth_rate, dlogH_dloga, vel_norm, acc_norm
from .models import (
    StyleNBodyEmulatorCore,
    StyleNBodyEmulatorVelCore,
    NBodyEmulatorCore,
    NBodyEmulatorVelCore,
)

__version__ = "0.1.0"

__all__ = [
    "create_emulator",
    "NBodyEmulator",
    "SubboxConfig",
    "SubboxProcessor",
    "load_default_parameters",
    "modulate_emulator_parameters",
    "modulate_emulator_parameters_vel",
    "growth_factor",
    "hubble_rate",
    "growth_rate",
    "dlogH_dloga",
    "vel_norm",
    "acc_norm",
    "StyleNBodyEmulatorCore",
    "StyleNBodyEmulatorVelCore",
    "NBodyEmulatorCore",
    "NBodyEmulatorVelCore",
]
