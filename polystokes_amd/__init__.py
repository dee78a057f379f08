"""polystokes_amd — MI355X-native drop-in for the PolyStokes per-step reduced-viscosity Stokes solve.

The product is the C-ABI shared library `libpolystokes_hip.so` (include/polystokes.h), built from the HIP
sources in `csrc/`.  This module is only the Python harness over that ABI used by tests and bench.py.
There is no CPU fallback: loading fails loudly if the library is missing, and creating a context fails
loudly if no HIP device is present.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from . import _abi
from ._abi import (FieldsIn, FieldsOut, Params, Scene, Stats, default_params)  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PS_LIB") or os.path.join(_HERE, "libpolystokes_hip.so")   # PS_LIB: A/B builds (scripts/build_variant.sh)
_lib = None

EXPORTED_SYMBOLS = [
    "ps_abi_version", "ps_reduced_dof", "ps_context_create", "ps_context_destroy", "ps_last_error", "ps_params_default",
    "ps_upload_fields", "ps_step_device", "ps_setup_device", "ps_solve_device", "ps_download_fields",
    "polystokes_step", "ps_apply_operator", "ps_apply_preconditioner", "ps_query_array", "ps_read_array",
    "ps_export_component_matrices", "ps_export_matrices", "ps_export_stats", "ps_bench_kernel", "ps_memory_stats", "ps_set_interrupt", "ps_solve_exported_system",
    "ps_set_slab", "ps_set_brick", "ps_comm_unique_id", "ps_comm_init_rccl", "ps_comm_selftest", "ps_comm_init_tcp", "ps_dist_stats",
    "ps_group_create", "ps_group_destroy", "ps_group_rank", "ps_group_step",
]


def build(force=False):
    """Compile every HIP source for gfx950 (hipcc cross-compiles without a GPU)."""
    csrc = os.path.join(_HERE, "csrc")
    if force:
        subprocess.check_call(["make", "-C", csrc, "clean"])
    subprocess.check_call(["make", "-C", csrc, "-j4", "-s"])
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `make -C polystokes_amd/csrc` "
                "(__graft_entry__.build()).  There is no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        L.ps_abi_version.restype = C.c_int32
        L.ps_reduced_dof.restype = C.c_int32
        L.ps_context_create.argtypes = [C.c_int32]
        L.ps_context_create.restype = C.c_void_p
        L.ps_context_destroy.argtypes = [C.c_void_p]
        L.ps_last_error.argtypes = [C.c_void_p]
        L.ps_last_error.restype = C.c_char_p
        L.ps_params_default.argtypes = [C.POINTER(Params)]
        L.ps_upload_fields.argtypes = [C.c_void_p, C.POINTER(Params), C.POINTER(FieldsIn)]
        L.ps_upload_fields.restype = C.c_int32
        for fn in (L.ps_step_device, L.ps_setup_device, L.ps_solve_device):
            fn.argtypes = [C.c_void_p, C.POINTER(Stats)]
            fn.restype = C.c_int32
        L.ps_download_fields.argtypes = [C.c_void_p, C.POINTER(FieldsOut)]
        L.ps_download_fields.restype = C.c_int32
        L.polystokes_step.argtypes = [C.c_void_p, C.POINTER(Params), C.POINTER(FieldsIn), C.POINTER(FieldsOut),
                                      C.POINTER(Stats)]
        L.polystokes_step.restype = C.c_int32
        L.ps_apply_operator.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.ps_apply_operator.restype = C.c_int32
        L.ps_apply_preconditioner.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.ps_apply_preconditioner.restype = C.c_int32
        L.ps_query_array.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_int32)]
        L.ps_query_array.restype = C.c_int64
        L.ps_read_array.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64]
        L.ps_read_array.restype = C.c_int32
        L.ps_export_component_matrices.argtypes = [C.c_void_p, C.c_char_p]
        L.ps_export_component_matrices.restype = C.c_int32
        L.ps_export_matrices.argtypes = [C.c_void_p, C.c_char_p]
        L.ps_export_matrices.restype = C.c_int32
        L.ps_export_stats.argtypes = [C.c_void_p, C.POINTER(Stats), C.c_char_p]
        L.ps_export_stats.restype = C.c_int32
        L.ps_bench_kernel.argtypes = [C.c_void_p, C.c_char_p, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.ps_bench_kernel.restype = C.c_int32
        L.ps_memory_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
        L.ps_memory_stats.restype = C.c_int32
        L.ps_solve_exported_system.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(Params), C.c_double, C.c_void_p, C.c_int64, C.POINTER(Stats)]
        L.ps_solve_exported_system.restype = C.c_int32
        L.ps_set_interrupt.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.ps_set_interrupt.restype = C.c_int32
        L.ps_set_slab.argtypes = [C.c_void_p, C.POINTER(_abi.SlabStruct)]
        L.ps_set_slab.restype = C.c_int32
        L.ps_set_brick.argtypes = [C.c_void_p, C.POINTER(_abi.BrickStruct)]
        L.ps_set_brick.restype = C.c_int32
        L.ps_comm_unique_id.argtypes = [C.c_void_p]
        L.ps_comm_unique_id.restype = C.c_int32
        L.ps_comm_init_rccl.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]
        L.ps_comm_init_rccl.restype = C.c_int32
        L.ps_comm_init_tcp.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_char_p, C.c_int32]
        L.ps_comm_init_tcp.restype = C.c_int32
        L.ps_comm_selftest.argtypes = [C.c_void_p]
        L.ps_comm_selftest.restype = C.c_int32
        L.ps_dist_stats.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        L.ps_dist_stats.restype = C.c_int32
        L.ps_group_create.argtypes = [C.c_int32, C.c_int32]
        L.ps_group_create.restype = C.c_void_p
        L.ps_group_destroy.argtypes = [C.c_void_p]
        L.ps_group_rank.argtypes = [C.c_void_p, C.c_int32]
        L.ps_group_rank.restype = C.c_void_p
        L.ps_group_step.argtypes = [C.c_void_p, C.POINTER(Stats)]
        L.ps_group_step.restype = C.c_int32
        _lib = L
    return _lib


_DT = {(1, "i"): np.int8, (4, "i"): np.int32, (4, "f"): np.float32, (8, "f"): np.float64, (4, "u"): np.uint32}


def _kind(name):
    if name.endswith(("Labels", "Indices", ".col", ".ptr", "Region", "Perm", ".chunkInfo", ".chunkRep", ".code")) or name.startswith("faceRow"):
        return "i"
    if name in ("valuesCoded", "columns16", "diagonalsCoded", "fusedStep", "streamRuns", "rowPerLane", "chebInner32"):
        return "i"
    if name in ("ownedX", "ownedY", "ownedZ"):
        return "f"
    if False:
        return "i"
    if name == "reducedRowFace":
        return "u"
    return "f"


def process_memory_stats():
    """ps_memory_stats without a context: device bytes held through the library in this process, their peak, live contexts"""
    v = (C.c_int64 * 4)()
    lib().ps_memory_stats(None, v)
    return {"live_bytes": int(v[0]), "peak_bytes": int(v[1]), "contexts": int(v[3])}


class PolyStokesError(RuntimeError):
    pass


class Solver:
    """Thin object wrapper over a `ps_context` — the counterpart of `HDK_PolyStokes::Solver`
    (exec/HDK_PolyStokesSolver.h:27) as seen through the C ABI."""

    def __init__(self, device=0, _handle=None):
        self.L = lib()
        self._owned = _handle is None
        h = self.L.ps_context_create(device) if _handle is None else _handle
        if not h:
            raise PolyStokesError(self.L.ps_last_error(None).decode())
        self.h = C.c_void_p(h)
        self.stats = Stats()
        self.scene = None

    def close(self):
        if getattr(self, "h", None):
            if self._owned:
                self.L.ps_context_destroy(self.h)
            self.h = None

    def set_interrupt(self, fn):
        """fn() -> truthy stops the solve at the next CG batch boundary (UT_Interrupt equivalent)."""
        if fn is None:
            self._cb = None
            self._check(self.L.ps_set_interrupt(self.h, None, None))
            return
        self._cb = C.CFUNCTYPE(C.c_int32, C.c_void_p)(lambda user: 1 if fn() else 0)
        self._check(self.L.ps_set_interrupt(self.h, C.cast(self._cb, C.c_void_p), None))

    def set_slab(self, slab):
        st = _abi.SlabStruct(slab.rank, slab.world, slab.zLoOwned, slab.zHiOwned, slab.hasLower, slab.hasUpper, slab.z0)
        self._check(self.L.ps_set_slab(self.h, C.byref(st)))

    def set_brick(self, b):
        """b: partition.Brick (ps_set_brick: the decomposition along all three axes)."""
        i3 = C.c_int32 * 3
        st = _abi.BrickStruct(b.rank, b.world, i3(*b.dims), i3(*b.lo), i3(*b.hi), i3(*b.hasLower), i3(*b.hasUpper), i3(*b.g0))
        self._check(self.L.ps_set_brick(self.h, C.byref(st)))

    def comm_selftest(self):
        self._check(self.L.ps_comm_selftest(self.h))

    def dist_stats(self):
        """What this rank's last distributed solve did (ps_dist_stats): dict of bytes per iteration over its cuts, owned DOFs,
        whether the exchanges overlapped, sampled transport / all-reduce times, halo cells whose label the owners' exchange changed."""
        v = (C.c_double * 8)()
        self._check(self.L.ps_dist_stats(self.h, v))
        return {"halo_bytes_per_iter": v[0], "owned_dofs": v[1], "overlap": bool(v[2]),
                "exchange_ms_per_transport": (v[3] / v[4]) if v[4] else None, "exchange_samples": int(v[4]),
                "allreduce_ms": (v[5] / v[6]) if v[6] else None, "allreduce_samples": int(v[6]), "halo_label_changes": int(v[7])}

    def memory_stats(self):
        """ps_memory_stats: device bytes held through the library in this process, their peak, bytes this context dropped that still wait for
        release, live contexts"""
        v = (C.c_int64 * 4)()
        self.L.ps_memory_stats(self.h, v)
        return {"live_bytes": int(v[0]), "peak_bytes": int(v[1]), "deferred_bytes": int(v[2]), "contexts": int(v[3])}

    def comm_init(self, uid_bytes, rank, world):
        buf = C.create_string_buffer(bytes(uid_bytes), 128)
        self._check(self.L.ps_comm_init_rccl(self.h, buf, rank, world))

    def comm_init_tcp(self, rank, world, host="127.0.0.1", base_port=29600):
        """host-staged transport (one process per rank, ranks may share a GPU); collective over the world"""
        self._check(self.L.ps_comm_init_tcp(self.h, rank, world, host.encode(), base_port))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, allow=(1,)):
        if rc == _abi.FAILED or rc not in allow:
            raise PolyStokesError(f"rc={rc}: {self.L.ps_last_error(self.h).decode()}")
        return rc

    def upload(self, scene, params):
        self.scene, self.params = scene, params
        fi = scene.fields_in()
        self._check(self.L.ps_upload_fields(self.h, C.byref(params), C.byref(fi)))

    def setup(self):
        return self._check(self.L.ps_setup_device(self.h, C.byref(self.stats)))

    def solve(self):
        return self._check(self.L.ps_solve_device(self.h, C.byref(self.stats)), allow=(0, 1, -3, -4))

    def step_device(self):
        return self._check(self.L.ps_step_device(self.h, C.byref(self.stats)), allow=(0, 1, -3, -4))

    def step(self, scene, params):
        """solveGasSubclass equivalent on host buffers (HDK_PolyStokes.C:222-609)."""
        self.scene, self.params = scene, params
        fi = scene.fields_in()
        out, keep = self._fields_out()
        rc = self._check(self.L.polystokes_step(self.h, C.byref(params), C.byref(fi), C.byref(out), C.byref(self.stats)),
                         allow=(0, 1, -3, -4))
        self.vel, self.valid = keep[:3], keep[3:]
        return rc

    def _fields_out(self):
        sh = _abi.grid_shapes(self.scene.nx, self.scene.ny, self.scene.nz)
        keep = [np.empty(sh["face" + a], np.float32) for a in "XYZ"] + [np.empty(sh["face" + a], np.float32) for a in "XYZ"]
        out = FieldsOut()
        for a in range(3):
            out.vel[a] = keep[a].ctypes.data
            out.valid[a] = keep[3 + a].ctypes.data
        return out, keep

    def download(self):
        out, keep = self._fields_out()
        self._check(self.L.ps_download_fields(self.h, C.byref(out)))
        self.vel, self.valid = keep[:3], keep[3:]
        return self.vel, self.valid

    def array(self, name):
        eb = C.c_int32(0)
        n = self.L.ps_query_array(self.h, name.encode(), C.byref(eb))
        if n < 0:
            raise KeyError(name)
        out = np.empty(n, dtype=_DT[(eb.value, _kind(name))])
        if n:
            self._check(self.L.ps_read_array(self.h, name.encode(), out.ctypes.data, out.nbytes))
        return out

    def apply(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.empty_like(x)
        self._check(self.L.ps_apply_operator(self.h, x.ctypes.data, y.ctypes.data))
        return y

    def precondition(self, r):
        """z = M^-1 r of the configured preconditioner (reference numbering)"""
        r = np.ascontiguousarray(r, dtype=np.float64)
        z = np.empty_like(r)
        self._check(self.L.ps_apply_preconditioner(self.h, r.ctypes.data, z.ctypes.data))
        return z

    def bench_kernel(self, name, iters=20):
        ms, by = C.c_double(0), C.c_double(0)
        self._check(self.L.ps_bench_kernel(self.h, name.encode(), iters, C.byref(ms), C.byref(by)))
        return ms.value, by.value

    def export_component_matrices(self, prefix):
        self._check(self.L.ps_export_component_matrices(self.h, prefix.encode()))

    def export_matrices(self, prefix):
        self._check(self.L.ps_export_matrices(self.h, prefix.encode()))

    def solve_exported_system(self, prefix, params, dt, n):
        """PCG on a component set written by exportComponentMatrices (files <prefix>Mat_*.mtx, Vec_b.mtx)."""
        x = np.zeros(n, np.float64)
        rc = self._check(self.L.ps_solve_exported_system(self.h, prefix.encode(), C.byref(params), dt, x.ctypes.data, n, C.byref(self.stats)),
                         allow=(0, 1))
        return rc, x

    def export_stats(self, prefix):
        self._check(self.L.ps_export_stats(self.h, C.byref(self.stats), prefix.encode()))

    # convenience accessors into dimData (Solver.cpp:578-593)
    @property
    def nP(self):
        return int(self.stats.dimData[12])

    @property
    def nT(self):
        return int(self.stats.dimData[13])

    @property
    def nA(self):
        return int(self.stats.dimData[7])

    @property
    def nRegions(self):
        return int(self.stats.dimData[24])

    def S_matrices(self):
        """(S, St) as scipy CSR in REFERENCE numbering — rows of S are face rows (active faces in reference
        order, then reduced-with-entries), columns [p; tau] as in Solver.h:586-606.  On the device both are
        stored in the block-interleaved internal numbering; `sysPerm` / `rowPerm` map reference -> internal."""
        import scipy.sparse as sp
        n = self.nP + self.nT
        nA = self.nA
        sys_perm, row_perm = self.array("sysPerm"), self.array("rowPerm")
        ptr, col, val = self.array("S.ptr"), self.array("S.col"), self.array("S.val")
        nrows = len(ptr) - 1
        S = sp.csr_matrix((val, col, ptr), shape=(nrows, n))
        rows = np.concatenate([row_perm, np.arange(nA, nrows, dtype=row_perm.dtype)])
        S_ref = S[rows, :][:, sys_perm].tocsr()
        ptr, col, val = self.array("St.ptr"), self.array("St.col"), self.array("St.val")
        St = sp.csr_matrix((val, col, ptr), shape=(n, nrows))
        St_ref = St[sys_perm, :][:, rows].tocsr()
        S_ref.sort_indices()
        St_ref.sort_indices()
        return S_ref, St_ref


def comm_unique_id():
    """128-byte RCCL unique id (rank 0 creates it, the harness broadcasts it)."""
    buf = C.create_string_buffer(128)
    if lib().ps_comm_unique_id(buf) != 1:
        raise PolyStokesError("ncclGetUniqueId failed")
    return bytes(buf.raw)


class Group:
    """`world` ranks inside one process on one GPU (device copies instead of RCCL): the distributed algorithm
    on a single-GPU box.  Same kernels, exchange lists and reduction order as the one-process-per-GPU path."""

    def __init__(self, world, device=0, dims=None):
        """dims = (dx, dy, dz) ranks per axis (bricks, dx * dy * dz == world); None: z-slabs."""
        self.L = lib()
        if dims is not None and dims[0] * dims[1] * dims[2] != world:
            raise ValueError("dims must multiply to world")
        self.dims = tuple(dims) if dims is not None else None
        g = self.L.ps_group_create(device, world)
        if not g:
            raise PolyStokesError(self.L.ps_last_error(None).decode())
        self.g = C.c_void_p(g)
        self.world = world
        self.ranks = [Solver(device, _handle=self.L.ps_group_rank(self.g, r)) for r in range(world)]
        self.stats = Stats()

    def step(self):
        rc = self.L.ps_group_step(self.g, C.byref(self.stats))
        if rc == _abi.FAILED:
            raise PolyStokesError(self.L.ps_last_error(self.ranks[0].h).decode())
        for r in self.ranks:
            r.stats = self.stats
        return rc

    def solve_scene(self, scene, params):
        """Partition `scene` into slabs, run the distributed step, merge the owned faces into global arrays."""
        from . import partition
        if self.dims is not None:
            return self._solve_scene_bricks(scene, params)
        slabs = [partition.make_slab(scene.nz, self.world, r, params.tileSize) for r in range(self.world)]
        for r, sl in enumerate(slabs):
            self.ranks[r].upload(partition.local_scene(scene, sl), params)
            self.ranks[r].set_slab(sl)
        rc = self.step()
        sh = _abi.grid_shapes(scene.nx, scene.ny, scene.nz)
        vel = [np.array(scene.vel[a], copy=True) for a in range(3)]
        valid = [np.zeros(sh["face" + "XYZ"[a]], np.float32) for a in range(3)]
        for r, sl in enumerate(slabs):
            lv, lval = self.ranks[r].download()
            for a in range(3):
                owned = self.ranks[r].array("owned" + "XYZ"[a])
                partition.merge_faces(vel[a], lv[a], owned, sl, a)
                partition.merge_faces(valid[a], lval[a], owned, sl, a)
        self.vel, self.valid, self.slabs = vel, valid, slabs
        return rc

    def _solve_scene_bricks(self, scene, params):
        from . import partition
        bricks = [partition.make_brick((scene.nx, scene.ny, scene.nz), self.dims, r, params.tileSize) for r in range(self.world)]
        for r, b in enumerate(bricks):
            self.ranks[r].upload(partition.local_scene_brick(scene, b), params)
            self.ranks[r].set_brick(b)
        rc = self.step()
        sh = _abi.grid_shapes(scene.nx, scene.ny, scene.nz)
        vel = [np.array(scene.vel[a], copy=True) for a in range(3)]
        valid = [np.zeros(sh["face" + "XYZ"[a]], np.float32) for a in range(3)]
        for r, b in enumerate(bricks):
            lv, lval = self.ranks[r].download()
            for a in range(3):
                owned = self.ranks[r].array("owned" + "XYZ"[a])
                partition.merge_faces_brick(vel[a], lv[a], owned, b, a)
                partition.merge_faces_brick(valid[a], lval[a], owned, b, a)
        self.vel, self.valid, self.slabs, self.bricks = vel, valid, bricks, bricks
        return rc

    def close(self):
        if getattr(self, "g", None):
            for r in self.ranks:
                r.h = None
            self.L.ps_group_destroy(self.g)
            self.g = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
