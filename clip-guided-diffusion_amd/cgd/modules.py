"""Plugin surface `cgd.modules` (reference: /root/reference/cgd/modules.py:5-66): MakeCutouts backed by the HIP
crop+adaptive-pool kernel (csrc/guidance.hip via cgd_cutouts_fwd)."""
from cgd_amd.guidance import MakeCutouts  # noqa: F401
