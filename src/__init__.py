"""Reference-compatible import path: ``from src.core import NeuralField`` etc. resolve to the
MI355X-native implementation in ``project-nerf_amd/``."""
