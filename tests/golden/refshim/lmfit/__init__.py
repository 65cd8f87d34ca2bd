"""Placeholder for lmfit so the reference's dynspec.py / scint_models.py import.
The secondary-spectrum / theta-theta path never calls into lmfit."""
class _Missing:  # pragma: no cover
    def __init__(self, *a, **k):
        raise NotImplementedError("lmfit shim")
class Minimizer(_Missing): pass
class Parameters(_Missing): pass
def fit_report(*a, **k):  # pragma: no cover
    raise NotImplementedError("lmfit shim")
