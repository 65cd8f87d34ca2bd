"""Placeholder: the hot path never touches coordinates."""
class SkyCoord:  # pragma: no cover
    def __init__(self, *a, **k):
        raise NotImplementedError("astropy.coordinates shim")
def get_body_barycentric(*a, **k):  # pragma: no cover
    raise NotImplementedError("astropy.coordinates shim")
