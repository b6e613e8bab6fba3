"""2D Cartesian mesh metadata (reference: src/Grids/CartesianGrid.jl, src/Grids/mask_utils.jl,
src/custom_structures.jl:51-57).  Only what the time step needs: Nx, Ny (boundary-typed),
dx, dy, node coordinates and the 0/1/2/3 mask.  angle must be 0."""
from __future__ import annotations

from dataclasses import dataclass
from types import SimpleNamespace

import numpy as np


@dataclass(frozen=True)
class N_Periodic:
    N: int

    def __int__(self):
        return self.N


@dataclass(frozen=True)
class N_NonPeriodic:
    N: int

    def __int__(self):
        return self.N


def interior_boundary(mask: np.ndarray) -> np.ndarray:
    """mask_utils.jl:14-22: land points adjacent (4-neighbourhood, circular) to ocean"""
    mask = mask.astype(bool)
    bmask = np.zeros(mask.shape, dtype=int)
    for dims in [(1, 0), (-1, 0), (0, 1), (0, -1)]:
        bmask += (np.roll(mask, dims, axis=(0, 1)) & ~mask)
    return bmask != 0


def make_boundaries(mask: np.ndarray, Nx, Ny) -> np.ndarray:
    """mask_utils.jl:38-55 -> total mask 0 land / 1 ocean / 2 land boundary / 3 grid boundary"""
    mask = mask.astype(bool)
    bmask = interior_boundary(mask)
    total = mask.astype(np.int8) + 2 * bmask.astype(np.int8)
    if isinstance(Nx, N_NonPeriodic):
        total[0, :] = 3
        total[-1, :] = 3
    if isinstance(Ny, N_NonPeriodic):
        total[:, 0] = 3
        total[:, -1] = 3
    return total


def make_boundary_lists(total_mask: np.ndarray):
    """mask_utils.jl:71-82: column-major findall lists (0-based (i,j) tuples)"""
    def findall(v):
        jj, ii = np.nonzero((total_mask == v).T)  # column-major order: i fastest
        return list(zip(ii.tolist(), jj.tolist()))
    return SimpleNamespace(ocean=findall(1), land_boundary=findall(2), grid_boundary=findall(3))


def mask_circle(mask, xx, yy, pp_ij, radius):
    """mask_utils.jl:118-133 (in place)"""
    px, py = xx[pp_ij], yy[pp_ij]
    mask[(xx - px) ** 2 + (yy - py) ** 2 < radius ** 2] = False
    return mask


class TwoDCartesianGridStatistics:
    """CartesianGrid.jl:26-64"""

    def __init__(self, xmin, xmax, Nx: int, ymin, ymax, Ny: int, angle=0.0, periodic_boundary=(False, False)):
        if angle != 0.0:
            raise NotImplementedError("rotated ProjetionKernel is not supported (SURVEY Appendix B.11)")
        self.dimx, self.dimy = xmax - xmin, ymax - ymin
        self.Ndx, self.Ndy = Nx - 1, Ny - 1
        self.dx, self.dy = self.dimx / self.Ndx, self.dimy / self.Ndy
        self.area = self.dx * self.dy
        self.xmin, self.xmax, self.ymin, self.ymax = xmin, xmax, ymin, ymax
        self.Nx = N_Periodic(Nx) if periodic_boundary[0] else N_NonPeriodic(Nx)
        self.Ny = N_Periodic(Ny) if periodic_boundary[1] else N_NonPeriodic(Ny)
        self.angle_dx = angle


class TwoDCartesianGridMesh:
    """CartesianGrid.jl:67-112.  Call forms: (dimx, nx, dimy, ny; ...) or
    (xmin, xmax, Nx, ymin, ymax, Ny; mask=...)."""

    def __init__(self, *args, mask=None, angle=0.0, periodic_boundary=(False, False)):
        if len(args) == 4:
            dimx, nx, dimy, ny = args
            xmin, xmax, ymin, ymax = 0.0, dimx, 0.0, dimy
        elif len(args) == 6:
            xmin, xmax, nx, ymin, ymax, ny = args
        else:
            raise TypeError("TwoDCartesianGridMesh(dimx, nx, dimy, ny) or (xmin, xmax, Nx, ymin, ymax, Ny)")
        self.stats = TwoDCartesianGridStatistics(xmin, xmax, nx, ymin, ymax, ny, angle=angle,
                                                 periodic_boundary=periodic_boundary)
        x = xmin + self.stats.dx * np.arange(nx)
        y = ymin + self.stats.dy * np.arange(ny)
        XX, YY = np.meshgrid(x, y, indexing="ij")  # XX[i,j] = x[i]
        if mask is None:
            mask = np.ones(XX.shape, dtype=bool)
        total = make_boundaries(np.asarray(mask, dtype=bool), self.stats.Nx, self.stats.Ny)
        self.data = SimpleNamespace(x=XX, y=YY, mask=total)

    def ProjetionKernel(self):
        """CartesianGrid.jl:115-121"""
        return np.array([[1 / self.stats.dx, 0.0], [0.0, 1 / self.stats.dy]])
