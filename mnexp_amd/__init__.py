"""mnexp_amd: MI355X-native LSTUR training path behind nvagus/mnexp's task/settings entry points."""
__version__ = "0.1.0"
