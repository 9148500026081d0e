"""deepmimic_amd: MI355X-native vectorised DeepMimic imitation environment (hot path only)."""
__version__ = "0.1.0"
