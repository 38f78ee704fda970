"""``sh2rgb`` as in the reference's ``src/utils/graphics_utils.py:72-73`` (C0 = 0.28209479177387814)."""
C0 = 0.28209479177387814


def sh2rgb(sh):
    return sh * C0 + 0.5
