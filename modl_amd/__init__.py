"""modl_amd — MI355X-native implementation of MODL's SOMF hot path
(DictFact.partial_fit: code solve, surrogate statistics, block-coordinate
dictionary update) behind the reference's estimator API."""
from .dict_fact import DictFact, Coder  # noqa: F401

__all__ = ['DictFact', 'Coder']
