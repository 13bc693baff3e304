"""CPU oracle for the CMCD annealed-Langevin hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this package.  The product (``cmcd_amd``) never does.

PARITY UNPINNED: the reference (shreyaspadhy/CMCD) is JAX-only, cannot be
imported in the build container (no jax/jaxlib/numpyro/haiku/distrax) and ships
no tests or golden vectors.  The oracle is therefore pinned only by
  * JAX's public Threefry PRNG known answers (tests/test_oracle_prng.py),
  * analytic identities (tests/test_oracle_identities.py),
  * closed-form gradients checked against torch.autograd / finite differences.
"""
