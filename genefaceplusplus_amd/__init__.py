"""genefaceplusplus_amd -- MI355X-native motion2video NeRF renderer behind GeneFace++'s modules/radnerfs API."""
__version__ = "0.1.0"
