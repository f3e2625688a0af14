"""CPU oracle for the RGBD -> phase-only-hologram hot path.

TEST INFRASTRUCTURE ONLY.  This package is a pure-torch, CPU, fp32 restatement of
the reference algorithm (WeijieXie/learned_hologram_gan @ 2025-02-17).  Only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and there only as the checker / reported baseline.
Nothing under ``learned_hologram_gan_amd/`` imports it; the product path raises
when its HIP library is missing instead of falling back to this code.

Parity pinning: every function here is checked against fixtures under
``tests/golden/`` that were produced by the *real* reference modules imported
from ``/root/reference`` in the build container (``oracle/make_golden.py`` is
the generating script) and against the reference's only known-answer artefact,
``output/test_output/terminalTest/poh.pt`` + ``0..9.png`` (copied as data).

Citations ``ref: <file>:<lines>`` are into ``/root/reference/``.

Layout
    optics.py   A1 A2 A5 A8 A9  constants, pad/crop, angular-spectrum propagation
    nets.py     A3 A4 A6 A7 A10 functional UNet / generator / critic on a state_dict
    losses.py   A13             focal sin/cos phase-gradient, TV, pixel losses
    step.py     A11 A12 A14     gradient penalty, D/G step sequence, Adam, all-planes validation
    pretrain.py     N4          stand-alone pre-training loops of the two generator halves
    perceptual.py   N1          VGG19 perceptual loss (parity unpinned: the reference downloads its weights)
    seeded.py                   deterministic per-key weights with the reference's
                                state_dict names and shapes
"""

from . import optics, nets, losses, step, seeded, perceptual, pretrain  # noqa: F401
