import numpy as np


def get_sub_slice(indices, sub_indices):
    """Indices of a batch inside an index array / slice / None
    (reference: modl/utils/__init__.py:4-27)."""
    if indices is None:
        if isinstance(sub_indices, slice):
            return np.arange(sub_indices.start, sub_indices.stop)
        return sub_indices
    if isinstance(indices, slice):
        return np.arange(indices.start + sub_indices.start, indices.start + sub_indices.stop)
    return indices[sub_indices]
