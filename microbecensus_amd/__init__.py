"""microbecensus_amd - MI355X-native hot path of MicrobeCensus (translated search + hit classification)."""
__version__ = "0.1.0"
