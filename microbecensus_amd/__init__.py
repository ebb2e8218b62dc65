"""microbecensus_amd - MI355X-native hot path of MicrobeCensus (translated search + hit classification)."""
import os as _os

__version__ = "0.1.0"


def configure_process_env():
    """For programs that OWN their process (scripts/run_microbe_census.py, scripts/rapsearch_mi355x, bench.py, the test suite):
    eight hardware queues instead of HIP's four - the library keeps seven streams busy per handle (csrc/mc_hip.hip, open_impl) and
    streams that share a queue wait for each other (51 instead of 54 M reads/s).  The HIP runtime reads the variable at its first
    call, so this must run before anything (torch included) touches the GPU; an explicit setting of the environment wins.
    Importing the package does NOT do this (a process-global side effect on every other HIP user of an embedding
    application, dependent on import order) - an application that embeds run_pipeline() exports the variable itself if it wants
    the last 5 %."""
    _os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
