"""MI355X-native conditional-VAE normative-modeling hot path (HIP kernels behind the
reference's own cVAE class surface).  See DESIGN.md."""
from . import _lib
from ._lib import NmError
from .layout import ModelSpec, ParamLayout
from .engine import Table, Job, JobSet, adam_step
from .api import (cVAE, cVAE_multimodal, cVAE_multimodal_regression, cVAE_multimodal_endtoend, mmJSD, DMVAE, WeightedDMVAE,
                  mmVAEPlus, mvtCAE, NormalLike)

__all__ = ["ModelSpec", "ParamLayout", "Table", "Job", "JobSet", "adam_step", "cVAE", "cVAE_multimodal", "cVAE_multimodal_regression",
           "cVAE_multimodal_endtoend", "mmJSD", "DMVAE", "WeightedDMVAE", "mmVAEPlus", "mvtCAE", "NormalLike",
           "_lib", "NmError"]
