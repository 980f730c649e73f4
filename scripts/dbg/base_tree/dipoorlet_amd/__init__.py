"""dipoorlet_amd — MI355X-native activation-calibration core behind Dipoorlet's calibration API.

Product code: HIP kernels + C ABI (csrc/, include/dipoorlet_hip.h) and the Python host mirror of the
reference's tensor_cali / forward_net / quantize interfaces.  Never imports oracle/.
"""
__version__ = "0.1.0"
