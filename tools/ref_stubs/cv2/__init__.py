"""Import stub for golden generation (build container only; OpenCV is absent offline).  NO ARITHMETIC: the only
function the pinned code path calls is cvtColor(frame, COLOR_RGB2BGR) on a uint8 HxWx3 array, which is a channel-order
permutation (UVG:548)."""
import numpy as np

COLOR_RGB2BGR = 4
COLOR_BGR2RGB = 4
IMREAD_COLOR = 1


def cvtColor(img, code):
    assert code == COLOR_RGB2BGR and img.ndim == 3 and img.shape[2] == 3
    return np.ascontiguousarray(img[:, :, ::-1])


def __getattr__(name):          # anything else is outside the pinned path
    raise AttributeError(f"cv2 stub: {name} is not available (import stub for tools/gen_goldens_egress.py)")
