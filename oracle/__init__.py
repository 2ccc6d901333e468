"""CPU oracle for the adversarial G+D train step.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product
path: only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg
of ``bench.py`` may import it, and only as the checker.  The product path
(``gan-image-captioning_amd``) never imports this package and fails loudly when
the HIP library is missing.

Parity status
-------------
* Decoder.sample / add_gumbel / Discriminator.forward / get_losses /
  get_fixed_temperature / clip+Adam step order: PINNED.  ``cpu_step.py`` is a
  plain-torch restatement that is checked (``tests/test_oracle_golden.py``)
  against golden vectors produced by ``make_golden.py``, which imports the
  reference's own classes from /root/reference/src (behind inert stubs for the
  unused ``torchvision.models`` / ``SummaryWriter`` imports).
* Encoder ResNet trunk: PARITY UNPINNED.  The reference builds it from
  ``torchvision.models.resnet18`` (generator.py:12); torchvision is absent and
  no version is pinned anywhere in the reference, so ``cpu_encoder.py`` is a
  build-owned restatement of the published ResNet-18/50 architecture.
"""
