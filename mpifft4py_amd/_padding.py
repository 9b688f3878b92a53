"""Host-side (numpy) spectrum padding helpers with the names the reference's classes carry
(slab.py:516-536, 803-825; pencil.py:351-379; line.py:164-175).  The device path never uses them -- the
3/2-rule copies are fused into the transforms (csrc/plan.hip) -- they exist for callers of the
reference's helper API.  Written once, generically: a spectrum axis of n modes sits in a padded axis of
m >= n modes with its non-negative frequencies at the front and its negative ones at the back."""
import numpy as np


def _ax(nd, axis, sl):
    idx = [slice(None)] * nd
    idx[axis] = sl
    return tuple(idx)


def spread(fu, fp, n, axis):
    """fp <- fu along `axis`: lower half to the front, upper half to the back (the rest of fp is left as it is)."""
    h = int(n) // 2
    nd = fp.ndim
    fp[_ax(nd, axis, slice(0, h))] = fu[_ax(nd, axis, slice(0, h))]
    fp[_ax(nd, axis, slice(fp.shape[axis] - h, None))] = fu[_ax(nd, axis, slice(h, None))]
    return fp


def gather_fold(fp, fu, n, axis):
    """fu <- fp along `axis` with the Nyquist fold: rows 0..n/2 taken as they are, the last n/2 rows of fp ADDED
    to rows n/2.. of fu (so row n/2 holds the sum of both Nyquist images)."""
    h = int(n) // 2
    nd = fp.ndim
    fu[_ax(nd, axis, slice(0, h + 1))] = fp[_ax(nd, axis, slice(0, h + 1))]
    fu[_ax(nd, axis, slice(h, None))] += fp[_ax(nd, axis, slice(fp.shape[axis] - h, None))]
    return fu


# ---- slab.R2C (slab.py:516-536) ------------------------------------------------------------------
def r2c_copy_to_padded(fu, fp, N, axis=0):
    if axis in (0, 1):
        return spread(fu, fp, N[axis], axis)
    if axis == 2:
        fp[:, :, :int(N[2]) // 2 + 1] = fu
    return fp


def r2c_copy_from_padded(fp, fu, N, axis=0):
    nf = int(N[2]) // 2 + 1
    if axis == 1:
        fu.fill(0)
        return gather_fold(fp[:, :, :nf], fu, N[1], 1)
    if axis == 2:
        fu[:] = fp[:, :, :nf]
    return fu


# ---- slab.C2C (slab.py:803-825) ------------------------------------------------------------------
def c2c_copy_to_padded(fu, fp, N, axis=0):
    return spread(fu, fp, N[axis], axis) if axis in (0, 1, 2) else fp


def c2c_copy_from_padded(fp, fu, N, axis=0):
    if axis == 1:          # y and z truncated together, both with the fold
        tmp = np.zeros(fp.shape[:2] + (int(N[2]),), dtype=fp.dtype)
        gather_fold(fp, tmp, N[2], 2)
        fu.fill(0)
        gather_fold(tmp, fu, N[1], 1)
    return fu
