def disk(*a, **k):  # pragma: no cover - never invoked by the decode path
    raise NotImplementedError("skimage stub: visualisation only")
