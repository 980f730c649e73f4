"""Weight transforms that run between calibration and deployment (dipoorlet/weight_transform/)."""
from .adaround import adaround
from .bias_correction import bias_correction
from .brecq import brecq
from .sparse_quant import sparse_quant
from .update_bn import update_bn
from .weight_equalization import weight_equalization
from .weight_trans_base import weight_calibration

__all__ = ["adaround", "bias_correction", "brecq", "sparse_quant", "update_bn", "weight_calibration", "weight_equalization"]
