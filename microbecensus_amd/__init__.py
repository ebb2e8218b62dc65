"""microbecensus_amd - MI355X-native hot path of MicrobeCensus (translated search + hit classification)."""
import os as _os

__version__ = "0.1.0"

# Eight hardware queues for the process instead of HIP's four: the library keeps seven streams busy (csrc/mc_hip.hip, open_impl) and
# streams that share a queue wait for each other.  Read by the HIP runtime at its first call - so set here, at import, before
# anything (torch included) touches the GPU; an explicit setting of the environment wins.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
