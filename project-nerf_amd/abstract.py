"""Operator surface kept from the reference (src/abstract.py:4-26): every spatial
representation maps coordinates to features and reports its width; every decoder maps
features to field outputs."""
from abc import ABC, abstractmethod

import torch.nn as nn


class BaseRepresentation(nn.Module, ABC):
    """coords [N, D] -> features [N, out_dim]."""

    @abstractmethod
    def forward(self, x):
        ...

    @property
    @abstractmethod
    def out_dim(self):
        """Feature width; decoders size their first layer from it."""
        ...


class BaseDecoder(nn.Module, ABC):
    """features -> rgb / sigma (and whatever else the mode defines)."""

    @abstractmethod
    def forward(self, x):
        ...
