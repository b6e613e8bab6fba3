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


@dataclass(frozen=True)
class N_TripolarNorth:
    """custom_structures.jl:59: the y axis of a tripolar grid — open at the south edge, folded onto itself (mirrored in x)
    at the north edge (ParticleInCell.jl:353-361, 409-428).  The reference builds it only in TripolarGridMOM6.jl, whose
    MOM6 file reader is out of scope; here any regular mesh can carry it (periodic_boundary = (True, "tripolar_north"))."""
    N: int

    def __int__(self):
        return self.N


def _axis_type(flag, N):
    if isinstance(flag, str):
        if flag != "tripolar_north":
            raise ValueError(f"unknown boundary type {flag!r}")
        return N_TripolarNorth(N)
    return N_Periodic(N) if flag else N_NonPeriodic(N)


def interior_boundary(mask: np.ndarray) -> np.ndarray:
    """land nodes with an ocean node among their four neighbours, the mesh read as circular in both axes
    (the rule of mask_utils.jl:14-22)"""
    ocean = np.asarray(mask, dtype=bool)
    wet_neighbour = np.zeros_like(ocean)
    for axis in (0, 1):
        wet_neighbour |= np.roll(ocean, 1, axis=axis) | np.roll(ocean, -1, axis=axis)
    return wet_neighbour & ~ocean


LAND, OCEAN, LAND_BOUNDARY, GRID_BOUNDARY = 0, 1, 2, 3      # classes of the total mask (mask_utils.jl:38-55)


def make_boundaries(mask: np.ndarray, Nx, Ny) -> np.ndarray:
    """node classes of a mesh: LAND / OCEAN from the mask, LAND_BOUNDARY = coast (interior_boundary), and the outer rows of every
    axis that is not periodic = GRID_BOUNDARY, whatever they were"""
    ocean = np.asarray(mask, dtype=bool)
    total = np.where(ocean, OCEAN, LAND).astype(np.int8)
    total[interior_boundary(ocean)] = LAND_BOUNDARY
    for axis, N in ((0, Nx), (1, Ny)):
        if isinstance(N, N_NonPeriodic):
            edge = [slice(None), slice(None)]
            edge[axis] = [0, -1]
            total[tuple(edge)] = GRID_BOUNDARY
    return total


def make_boundary_lists(total_mask: np.ndarray):
    """the nodes of each class as 0-based (i, j) pairs, i running fastest (the order of a column-major `findall`,
    mask_utils.jl:71-82: it is the order particles are numbered in)"""
    def nodes(cls):
        j, i = np.nonzero(total_mask.T == cls)
        return list(zip(i.tolist(), j.tolist()))
    return SimpleNamespace(ocean=nodes(OCEAN), land_boundary=nodes(LAND_BOUNDARY), grid_boundary=nodes(GRID_BOUNDARY))


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
        self.Nx = _axis_type(periodic_boundary[0], Nx)
        self.Ny = _axis_type(periodic_boundary[1], Ny)
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


# ---------------------------------------------------------------------------------------------
# Spherical (lon/lat) mesh — reference: src/Grids/SphericalGrid.jl, spherical_grid_corrections.jl
# ---------------------------------------------------------------------------------------------
R_EARTH_MESH = 6371.0e3      # SphericalGrid.jl:54,72
R_EARTH_PC = 6.3710e6        # spherical_grid_corrections.jl:3


def cal_dx_degree(XX):
    """SphericalGrid.jl:25-31: centred differences, one-sided at the ends"""
    dx = np.zeros(XX.shape)
    dx[1:-1, :] = (XX[2:, :] - XX[:-2, :]) / 2
    dx[0, :] = XX[1, :] - XX[0, :]
    dx[-1, :] = XX[-1, :] - XX[-2, :]
    return dx


def cal_dy_degree(YY):
    """SphericalGrid.jl:33-39"""
    dy = np.zeros(YY.shape)
    dy[:, 1:-1] = (YY[:, 2:] - YY[:, :-2]) / 2
    dy[:, 0] = YY[:, 1] - YY[:, 0]
    dy[:, -1] = YY[:, -1] - YY[:, -2]
    return dy


def cal_dx_meters(XX, YY):
    """SphericalGrid.jl:52-57"""
    return cal_dx_degree(XX) * np.pi / 180 * (R_EARTH_MESH * np.cos(YY * np.pi / 180))


def cal_dy_meters(YY):
    """SphericalGrid.jl:70-73"""
    return cal_dy_degree(YY) * np.pi / 180 * R_EARTH_MESH


class TwoDSphericalGridStatistics:
    """SphericalGrid.jl:99-135"""

    def __init__(self, xmin, xmax, Nx: int, ymin, ymax, Ny: int, mask_value=1, angle=0.0, periodic_boundary=(False, False)):
        self.dimx, self.dimy = xmax - xmin, ymax - ymin
        self.Ndx, self.Ndy = Nx - 1, Ny - 1
        self.Nx = _axis_type(periodic_boundary[0], Nx)
        self.Ny = _axis_type(periodic_boundary[1], Ny)
        self.dx_deg, self.dy_deg = self.dimx / self.Ndx, self.dimy / self.Ndy
        self.xmin, self.xmax, self.ymin, self.ymax = xmin, xmax, ymin, ymax
        self.angle_dx, self.mask_value = angle, mask_value
        # the kernels take the projection from the per-node metric; dx, dy are placeholders
        self.dx = self.dy = 1.0


class TwoDSphericalGridMesh:
    """SphericalGrid.jl:154-203: node lon/lat (degrees), dx/dy/area in metres, total mask.
    `metric()` gives what the time step needs: the diagonal of ProjetionKernel (:225-237, with the
    reference's `cos.(Gi.dy * pi / 180)` as written) and the PropagationCorrection coefficient
    (spherical_grid_corrections.jl:3-21)."""

    def __init__(self, xmin, xmax, Nx: int, ymin, ymax, Ny: int, mask=None, angle=0.0, periodic_boundary=(False, False)):
        self.stats = TwoDSphericalGridStatistics(xmin, xmax, Nx, ymin, ymax, Ny, angle=angle, periodic_boundary=periodic_boundary)
        x = xmin + self.stats.dx_deg * np.arange(Nx)
        y = ymin + self.stats.dy_deg * np.arange(Ny)
        XX, YY = np.meshgrid(x, y, indexing="ij")
        dx = cal_dx_meters(XX, YY)
        dy = cal_dy_meters(YY)
        if mask is None:
            mask = np.ones(XX.shape, dtype=bool)
        total = make_boundaries(np.asarray(mask, dtype=bool), self.stats.Nx, self.stats.Ny)
        self.data = SimpleNamespace(x=XX, y=YY, dx=dx, dy=dy, area=dx * dy, mask=total)

    def metric(self):
        d = self.data
        cos_lat = np.cos(d.dy * np.pi / 180)                      # (sic) SphericalGrid.jl:228
        with np.errstate(divide="ignore"):
            m11 = 1.0 / (cos_lat * d.dx)
            m22 = 1.0 / d.dy
        phi = d.y
        sg = np.sign(phi)
        coef = (sg * np.minimum(sg * np.tan(np.deg2rad(phi)), 60.0)) / R_EARTH_PC
        return m11, m22, coef
