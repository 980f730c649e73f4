from .tensor_cali_base import tensor_calibration  # noqa: F401
from .basic_algorithm import (find_clip_val_hist, find_clip_val_minmax, find_clip_val_minmax_weight,  # noqa: F401
                              find_clip_val_octav, tensor_cali_dispatcher)
