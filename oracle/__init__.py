"""CPU oracle for the capsule/YOLO training hot path.

TEST INFRASTRUCTURE ONLY.  This package is a plain PyTorch-CPU restatement of
the reference's algorithm (models.py / loss_fns.py / utils.py of
Cranial-XIX/cs231-capsule-yolo-traffic-sign-detection).  It exists so that the
hand-written HIP path can be checked against something that runs anywhere.

Rules (enforced by tests/test_layout.py):
  * only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
    ``cpu_baseline`` leg may import it, and only as the checker;
  * the product package (``cs231-capsule-yolo-traffic-sign-detection_amd/``,
    importable as ``capsyolo_amd``) never imports it and has no CPU fallback.

Pinning: the reference ships no tests, golden vectors or fixtures of its own
(SURVEY.md section 4), so the oracle is pinned by outputs of the reference
itself, executed in the build container on torch-CPU: see
``tests/golden/make_golden.py`` (imports /root/reference, writes the ``.npz``
fixtures under ``tests/golden/``) and ``tests/test_oracle_golden.py``.
"""
from . import models, loss_fns  # noqa: F401
