"""MI355X-native (gfx950) implementation of the RGBD -> phase-only-hologram hot path of
WeijieXie/learned_hologram_gan: same module / class / method names and checkpoint schema as the
reference package ``learnedMethodForHologram``, computed by hand-written HIP kernels behind the
C ABI of ``include/lhg_hip.h``.

    import learned_hologram_gan_amd as learnedMethodForHologram
"""

from . import native  # noqa: F401  (ctypes binding; load() raises if the library is absent)
from . import angular_spectrum_method, neural_network_components, utilities, watermelon_hologram  # noqa: F401

__all__ = ["angular_spectrum_method", "neural_network_components", "utilities", "watermelon_hologram", "native"]
