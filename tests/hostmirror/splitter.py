"""DataSplitter.splitFolds / getKthFold (reference src/carskit/data/processor/DataSplitter.java:68-133):
every stored entry of the rating matrix draws one uniform from the seeded stream, entry i (CRS order) starts in
fold floor(i / (n/k)) + 1, the draws are sorted ascending carrying the fold labels along, and the sorted labels are
dealt back to the entries in CRS order.  Fold k's entries form the test matrix, the rest the train matrix."""
import numpy as np

from .javarand import JavaRandom


def split_folds(n_entries, k_fold, seed):
    """fold label (1..k) of every matrix entry in CRS order."""
    num_fold = min(k_fold, n_entries)
    rdm = JavaRandom(seed).doubles(n_entries)
    indv = (n_entries + 0.0) / num_fold
    fold = (np.arange(n_entries) / indv).astype(np.int64) + 1      # (int)(i / indvCount) + 1
    order = np.argsort(rdm, kind="stable")                          # Sortor.quickSort(rdm, fold, ..., ascending)
    return fold[order], num_fold


def kth_fold(data, labels, k):
    """(train, test) RatingData of fold k (entries with a zero rating vanish in reshape(), as in the reference)."""
    nz = data.r != 0.0
    idx = np.arange(data.n)
    return data.subset(idx[(labels != k) & nz]), data.subset(idx[(labels == k) & nz])
