"""Rating-matrix splitting helpers of the recommender examples (reference:
modl/utils/recsys/cross_validation.py:8-50).  Host-side data plumbing: a random
partition of the stored ratings of a sparse matrix into a train and a test matrix of
the same shape, drawn with numpy's legacy RandomState so that the split of a given
seed is the reference's."""
import numpy as np
import scipy.sparse as sp


def _take(coo, sel):
    return sp.coo_matrix((coo.data[sel], (coo.row[sel], coo.col[sel])), shape=coo.shape)


class ShuffleSplit:
    """n_iter random (train, test) partitions of the stored entries; train_size is the
    fraction of entries kept for training (cross_validation.py:8-36)."""

    def __init__(self, n_iter=5, train_size=0.75, random_state=None):
        self.n_iter = n_iter
        self.train_size = train_size
        self.random_state = random_state

    def __len__(self):
        return self.n_iter

    def split(self, X):
        coo = sp.coo_matrix(X)
        rng = np.random.RandomState(self.random_state)
        n_entries = coo.data.shape[0]
        cut = int(self.train_size * n_entries)
        for _ in range(self.n_iter):
            shuffled = rng.permutation(n_entries)           # one draw per split, as the reference
            yield _take(coo, shuffled[:cut]), _take(coo, shuffled[cut:])


def train_test_split(X, train_size=0.75, random_state=None):
    """A single split (cross_validation.py:39-42)."""
    return next(ShuffleSplit(n_iter=1, train_size=train_size, random_state=random_state).split(X))


def cross_val_score(estimator, X, cv):
    """fit on each train part, score on the matching test part (cross_validation.py:45-50)."""
    return np.array([estimator.fit(train).score(test) for train, test in cv.split(X)])
