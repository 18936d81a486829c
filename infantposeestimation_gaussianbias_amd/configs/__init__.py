"""Configs module (same exports as the reference's configs/__init__.py)."""
from .config import Config, DataConfig, ModelConfig, TrainConfig, get_config

__all__ = ['Config', 'DataConfig', 'ModelConfig', 'TrainConfig', 'get_config']
