from .cindex import concordance_index, concordance_index_censored, NoComparablePairException  # noqa: F401
