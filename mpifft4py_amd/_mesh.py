"""Physical coordinates, wave vectors and the 2/3-rule filter of ONE rank's block of a distributed mesh.

The reference builds these in every class separately (slab.py:146-197, pencil.py:289-349, 945-969, line.py:105-134).
Here there is one description of a rank's block -- for every axis the window `[start, start + length)` it owns of the
physical mesh and of the spectrum, as the device-free layout query of the C ABI reports them (`mfft_layout_query`) --
and the classes only choose the dtype / rounding policy their reference counterpart has.  Nothing here touches a
device, so the helpers work on a machine without a GPU.

Rounding policies (they decide the last bit, and the fixtures under tests/golden/helpers_*.npz were written by the
reference): a physical coordinate is `(index * L) / N` evaluated in double and then cast ("sparse" form: slab and
y-aligned pencil), or `dtype(index) * (L / N)` ("dense" form: x-aligned pencil and the 2-D class); a scaled wave
number is `dtype(k) * dtype(2 pi / L)` (slab, 2-D class) or `dtype(k * (2 pi / L))` (pencils).
"""
import numpy as np


def dft_modes(n, half=False):
    """Integer wave numbers of a length-n DFT in storage order.  Full axis: 0 .. ceil(n/2)-1 followed by the negative
    ones (for even n the Nyquist mode is stored as -n/2); `half`: the n//2 + 1 non-negative ones a real transform
    keeps (its Nyquist mode is stored as +n/2)."""
    n = int(n)
    if half:
        return np.arange(n // 2 + 1, dtype=np.int64)
    k = np.arange(n, dtype=np.int64)
    k[(n + 1) // 2:] -= n
    return k


def _open(vec, axis, nd):
    """1-D vector -> array that broadcasts along `axis` of an nd-dimensional block."""
    shape = [1] * nd
    shape[axis] = vec.shape[0]
    return vec.reshape(shape)


class Block(object):
    """One rank's part of the mesh.

    N, L              global extents (nd entries each)
    real_window       per axis (start, length) of the physical block
    spectral_window   per axis (start, length) of the spectral block
    half_axis         index of the axis whose spectrum is the non-negative half (the real transform's axis), or None
    """

    def __init__(self, N, L, real_window, spectral_window, half_axis):
        self.N = [int(n) for n in N]
        self.L = L
        self.nd = len(self.N)
        self.real_window = [(int(s), int(l)) for s, l in real_window]
        self.spectral_window = [(int(s), int(l)) for s, l in spectral_window]
        self.half_axis = half_axis

    # ---- physical space -----------------------------------------------------------------------------------------
    def real_shape(self):
        return tuple(l for _, l in self.real_window)

    def spectral_shape(self):
        return tuple(l for _, l in self.spectral_window)

    def _indices(self, axis):
        s, l = self.real_window[axis]
        return np.arange(s, s + l, dtype=np.int64)

    def coordinates_sparse(self, dtype):
        """List of nd read-only views of the block's shape: coordinate i varies along axis i only."""
        shape = self.real_shape()
        out = []
        for i in range(self.nd):
            x = ((self._indices(i) * float(self.L[i])) / self.N[i]).astype(dtype)
            out.append(np.broadcast_to(_open(x, i, self.nd), shape))
        return out

    def coordinates_dense(self, dtype):
        """(nd, *block shape) array of coordinates, each index cast to `dtype` before it is scaled."""
        shape = self.real_shape()
        X = np.empty((self.nd,) + shape, dtype=dtype)
        for i in range(self.nd):
            step = np.float64(self.L[i]) / self.N[i]          # L carries the class's own rounding already
            x = (self._indices(i).astype(dtype) * step).astype(dtype)
            X[i] = _open(x, i, self.nd)
        return X

    # ---- spectral space -----------------------------------------------------------------------------------------
    def mode_vectors(self, zero_nyquist=False):
        """Per axis the integer wave numbers of the block's spectral window.  zero_nyquist: the mode n/2 of every even
        axis is reported as 0 (`eliminate_highest_freq` of the reference)."""
        out = []
        for i in range(self.nd):
            k = dft_modes(self.N[i], half=(i == self.half_axis))
            if zero_nyquist and self.N[i] % 2 == 0:
                k[self.N[i] // 2] = 0
            s, l = self.spectral_window[i]
            out.append(k[s:s + l])
        return out

    def wavenumber_grid(self, dtype=None, factors=None, cast_first=True, zero_nyquist=False, dense=False):
        """Sparse (open) grid of the block's wave numbers: a list of nd arrays, entry i of shape (1, .., len_i, .., 1).

        dtype       None keeps the integers (unless factors are given and cast_first is False: then the products)
        factors     per-axis scale (2 pi / L), or None
        cast_first  True: cast the integers to dtype, then multiply; False: multiply, then cast the product
        dense       broadcast every entry to the block's spectral shape (read-only views)
        """
        grid = []
        for i, k in enumerate(self.mode_vectors(zero_nyquist)):
            if cast_first and dtype is not None:
                k = k.astype(dtype)
            if factors is not None:
                k = k * factors[i]
                if not cast_first and dtype is not None:
                    k = k.astype(dtype)
            grid.append(_open(k, i, self.nd))
        if dense:
            shape = self.spectral_shape()
            grid = [np.broadcast_to(g, shape) for g in grid]
        return grid

    def two_thirds_filter(self, grid=None):
        """uint8 mask of the block's spectral shape: 1 where |k_i| < 2/3 (N_i // 2 + 1) on every axis (the reference's
        `get_dealias_filter`).  `grid`: the (open) wave-number grid to test, default the integer one."""
        if grid is None:
            grid = self.wavenumber_grid()
        keep = None
        for i, g in enumerate(grid):
            kmax = 2. / 3. * (self.N[i] // 2 + 1)
            cond = np.abs(g) < kmax
            keep = cond if keep is None else keep & cond
        return np.ascontiguousarray(np.broadcast_to(keep, self.spectral_shape())).astype(np.uint8)
