"""robo_amd -- MI355X-native GP-posterior + acquisition hot path behind RoBO's plugin API."""
__version__ = "0.1.0"
