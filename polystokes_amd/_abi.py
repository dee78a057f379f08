"""ctypes mirror of include/polystokes.h (POD structs and enums only).

The C header is the contract; this file restates it for Python harness code (tests, bench.py).
"""
import ctypes as C

import numpy as np

REDUCED_DOF = 26

# ps_result (exec/HDK_PolyStokesSolver.h:61-70)
UNSUPPORTED_SOLVER, INCOMPLETE, INVALID, FAILED, NOCONVERGE, SUCCESS, NOCHANGE = -4, -3, -2, -1, 0, 1, 2
# ps_label (exec/HDK_PolyStokesSolver.h:71-82)
UNASSIGNED, UNSOLVED, GENERICFLUID, ACTIVEFLUID, SOLID, REDUCED, UNVISITED, VISITED, BOUNDARY = (
    -1, -2, -3, -4, -5, -6, -7, -8, -9)
PCG_MATRIX_VECTOR_PRODUCTS, EIGEN = 0, 1
PRE_IDENTITY, PRE_DIAGONAL, PRE_CHEBYSHEV, PRE_CHEBYSHEV_F32 = 1, 5, 6, 7
ORDER_VOXEL_TILES, ORDER_LINEAR = 0, 1

STAGE_NAMES = ["weights", "classify", "regions", "indices", "tile_matrices", "blocks", "assemble",
               "precond", "solve", "recover", "writeback"]

SAMPLE_NAMES = ["center", "faceX", "faceY", "faceZ", "edgeYZ", "edgeXZ", "edgeXY"]


class Params(C.Structure):
    _fields_ = [
        ("mindensity", C.c_double), ("maxdensity", C.c_double),
        ("matrixSetup", C.c_int32), ("solverType", C.c_int32),
        ("doSolve", C.c_int32), ("keepNonConvergedResults", C.c_int32),
        ("exportMatrices", C.c_int32), ("exportComponentMatrices", C.c_int32),
        ("exportStats", C.c_int32), ("useWarmStart", C.c_int32),
        ("tolerance", C.c_double),
        ("maxSolverIterations", C.c_int32), ("useInputSurfaceWeights", C.c_int32),
        ("useInputCollisionWeights", C.c_int32), ("activeLiquidBoundaryLayerSize", C.c_int32),
        ("activeSolidBoundaryLayerSize", C.c_int32), ("doReducedRegions", C.c_int32),
        ("doTile", C.c_int32), ("tileSize", C.c_int32), ("tilePadding", C.c_int32),
        ("preconditioner", C.c_int32), ("indexOrder", C.c_int32), ("negateCollision", C.c_int32),
        ("preconditionerDegree", C.c_int32),
        ("exportDataPrefix", C.c_char_p),
    ]


class FieldsIn(C.Structure):
    _fields_ = [
        ("nx", C.c_int32), ("ny", C.c_int32), ("nz", C.c_int32),
        ("dx", C.c_double), ("dt", C.c_double), ("orig", C.c_double * 3),
        ("density", C.c_float), ("reserved", C.c_int32),
        ("vel", C.c_void_p * 3), ("surface", C.c_void_p), ("collision", C.c_void_p),
        ("viscosity", C.c_void_p), ("collisionvel", C.c_void_p * 3),
        ("weights", C.c_void_p * 14),
    ]


class FieldsOut(C.Structure):
    _fields_ = [("vel", C.c_void_p * 3), ("valid", C.c_void_p * 3)]


class Stats(C.Structure):
    _fields_ = [
        ("dimData", C.c_double * 27), ("solveData", C.c_double * 6),
        ("result", C.c_int32), ("usedBiCGStab", C.c_int32),
        ("stage_ms", C.c_double * 16),
    ]


class SlabStruct(C.Structure):
    _fields_ = [("rank", C.c_int32), ("world", C.c_int32), ("zLoOwned", C.c_int32), ("zHiOwned", C.c_int32),
                ("hasLower", C.c_int32), ("hasUpper", C.c_int32), ("zGlobalOwned", C.c_int32)]


class BrickStruct(C.Structure):
    _fields_ = [("rank", C.c_int32), ("world", C.c_int32), ("dims", C.c_int32 * 3), ("lo", C.c_int32 * 3), ("hi", C.c_int32 * 3),
                ("hasLower", C.c_int32 * 3), ("hasUpper", C.c_int32 * 3), ("globalLo", C.c_int32 * 3)]


def default_params(**kw):
    """Defaults of the reference's PRM template (exec/HDK_PolyStokes.C:88-208)."""
    p = Params()
    p.mindensity, p.maxdensity = 1.0, 100000.0
    p.matrixSetup, p.solverType = 0, PCG_MATRIX_VECTOR_PRODUCTS
    p.doSolve, p.keepNonConvergedResults = 1, 1
    p.exportMatrices = p.exportComponentMatrices = p.exportStats = 0
    p.useWarmStart = 1
    p.tolerance, p.maxSolverIterations = 1e-3, 5000
    p.useInputSurfaceWeights = p.useInputCollisionWeights = 1
    p.activeLiquidBoundaryLayerSize = p.activeSolidBoundaryLayerSize = 2
    p.doReducedRegions, p.doTile, p.tileSize, p.tilePadding = 1, 1, 16, 2
    p.preconditioner, p.indexOrder, p.negateCollision = PRE_IDENTITY, ORDER_VOXEL_TILES, 1
    p.exportDataPrefix = None
    for k, v in kw.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def grid_shapes(nx, ny, nz):
    """numpy shapes (z, y, x) — x fastest — of the 7 sample grids, keyed by SAMPLE_NAMES."""
    return {
        "center": (nz, ny, nx),
        "faceX": (nz, ny, nx + 1), "faceY": (nz, ny + 1, nx), "faceZ": (nz + 1, ny, nx),
        "edgeYZ": (nz + 1, ny + 1, nx), "edgeXZ": (nz + 1, ny, nx + 1), "edgeXY": (nz, ny + 1, nx + 1),
    }


class Scene:
    """Host-side input bundle: numpy float32 arrays laid out as the C ABI expects."""

    def __init__(self, nx, ny, nz, dx, dt, density, vel, surface, collision, viscosity,
                 collisionvel=None, weights=None, name="scene"):
        self.nx, self.ny, self.nz, self.dx, self.dt, self.density = nx, ny, nz, float(dx), float(dt), float(density)
        sh = grid_shapes(nx, ny, nz)
        f32 = lambda a, s: np.array(np.broadcast_to(np.asarray(a, dtype=np.float32), s), dtype=np.float32, order="C", copy=True)
        self.vel = [f32(vel[a], sh["face" + "XYZ"[a]]) for a in range(3)]
        self.surface = f32(surface, sh["center"])
        self.collision = f32(collision, sh["center"])
        self.viscosity = f32(viscosity, sh["center"])
        if collisionvel is None:
            collisionvel = [0.0, 0.0, 0.0]
        self.collisionvel = [f32(collisionvel[a], sh["face" + "XYZ"[a]]) for a in range(3)]
        self.weights = None
        if weights is not None:
            self.weights = [f32(weights[i], sh[SAMPLE_NAMES[i % 7]]) for i in range(14)]
        self.name = name

    def fields_in(self):
        fi = FieldsIn()
        fi.nx, fi.ny, fi.nz = self.nx, self.ny, self.nz
        fi.dx, fi.dt = self.dx, self.dt
        fi.orig[0] = fi.orig[1] = fi.orig[2] = 0.0
        fi.density = self.density
        for a in range(3):
            fi.vel[a] = self.vel[a].ctypes.data
            fi.collisionvel[a] = self.collisionvel[a].ctypes.data
        fi.surface = self.surface.ctypes.data
        fi.collision = self.collision.ctypes.data
        fi.viscosity = self.viscosity.ctypes.data
        for i in range(14):
            fi.weights[i] = self.weights[i].ctypes.data if self.weights is not None else None
        return fi
