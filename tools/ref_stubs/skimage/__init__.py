"""Import stub (no arithmetic): LFG/modules/util.py imports `skimage.draw.disk` for a visualisation helper
that the inference path never calls."""
