"""MI355X-native DIS-YOLO hot path: hand-written gfx950 kernels behind a C ABI
(``include/disyolo.h``, built from ``csrc/``) plus the host-side mirror of the
reference's ``YOLONet`` / ``Solver`` interface.  Import as ``disyolo_amd``."""
from . import config  # noqa: F401

__all__ = ["config"]
