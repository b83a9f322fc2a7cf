"""plen_ml_walk_amd: MI355X-native vectorised PLEN walking environment + TD3."""
__version__ = "0.1.0"
