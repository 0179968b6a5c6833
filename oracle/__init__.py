"""CPU oracle for the N-body emulator hot path -- TEST INFRASTRUCTURE ONLY.

This package is a NumPy/SciPy restatement of the reference algorithm
(`/root/reference/src/jax_nbody_emulator/`), written from the reference's
source text.  The reference itself is pure Python on JAX+Flax, and neither is
installed in the build container or on the GPU box (ordinary
ModuleNotFoundError, no network), so it can be neither imported nor run.

Status of the pin ("what proves the oracle right"):
  * The reference's tests hold NO golden vectors and the pretrained weights
    are absent (/root/reference/.MISSING_LARGE_BLOBS).  The only value-level
    known answers it offers are pinned in
    tests/test_oracle_pins.py: LeakyReLU values/tangents, vel primal ==
    non-vel primal, velocity proportional to vel_fac, the cosmology identities
    (D(0)=1, H(0)=100, EdS limits, f ~ Om(z)^0.55) and the README table.
  * Beyond those pins the numerical parity of the whole network is
    **parity unpinned** against JAX outputs; it is anchored instead on
    self-consistency that needs no reference: the tangent equals a float64
    central finite difference d(disp)/d(Dz), the style path equals the
    premodulated path, the conv core equals torch.nn.functional.conv3d
    (an independent implementation of VALID cross-correlation).

Rules: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this package.  The product (jax_nbody_emulator_with_dj_amd) never does.
"""

from . import cosmology, layers, model, subbox, params  # noqa: F401
