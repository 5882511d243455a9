"""MI355X-native DIS-YOLO hot path: hand-written gfx950 kernels behind a C ABI
(``include/disyolo.h``, built from ``csrc/``) plus the host-side mirror of the
reference's ``YOLONet`` / ``Solver`` interface.  Import as ``disyolo_amd``."""
import os as _os

# the step executor's lanes need hardware queues of their own even after RCCL has created its streams (ROCm
# default: 4 queues; 8 collide with four lanes + two communicators -- profiles/r05_hw_queues.txt); read by the HIP
# runtime at initialisation
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "12")

from . import config  # noqa: F401,E402

__all__ = ["config"]
