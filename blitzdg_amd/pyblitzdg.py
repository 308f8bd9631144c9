"""pyblitzdg-shaped front end over the C ABI.

Same class, method and property names as the reference's boost::python module
(src/pyblitzdg/pyblitzdg.cpp:59-201) for the objects the sw2d / advec1d path uses:
``MeshManager``, ``TriangleNodesProvisioner`` (+ ``dgContext()`` -> ``DGContext2D``),
``Nodes1DProvisioner``, ``LSERK4``, ``BCType``. As in the reference every property
access returns a FRESH C-order ndarray (float64 / int32). Objects outside the hot
path (quads, Poisson) are not provided. ``GaussFaceContext2D`` / ``CubatureContext2D`` (the curved,
over-integrated RHS's tables) come from ``buildGaussFaceNodes`` / ``buildCubatureVolumeMesh``.
"""
import numpy as np

from . import _capi as C
from ._capi import POINTER, byref, c_double, c_int, c_void_p, check, lib


class _BCType:
    """reference: struct BCType in src/pyblitzdg/pyblitzdg.cpp:52-56"""
    Dirichlet = 6
    Neuman = 7
    Wall = 3


BCType = _BCType()


class _LSERK4:
    """reference: include/LSERK4.hpp:15-29, exported at src/pyblitzdg/pyblitzdg.cpp:28-50,96-99"""
    numStages = lib.bdg_lserk4_num_stages()

    @property
    def rk4a(self):
        return np.array([lib.bdg_lserk4_a()[i] for i in range(self.numStages)], dtype=np.float64)

    @property
    def rk4b(self):
        return np.array([lib.bdg_lserk4_b()[i] for i in range(self.numStages)], dtype=np.float64)


LSERK4 = _LSERK4()


class VandermondeBuilder:
    """reference: src/pyblitzdg/pyblitzdg.cpp:92-93 (VandermondeBuilders::buildVandermondeMatrix_numpy,
    include/VandermondeBuilders.hpp:76-105). V[i, j] = P_j^(0,0)(r[i]), orthonormal Legendre polynomials; a tuple
    (V, Vinv) with includeInverse, else the 1-tuple (V,) -- the reference's own return shapes."""

    def buildVandermondeMatrix(self, r, includeInverse=True, order=-1):
        r = np.ascontiguousarray(r, dtype=np.float64).reshape(-1)
        m = r.size
        n = order + 1 if order > -1 else m
        V = np.zeros((m, n))
        Vinv = np.zeros((n, n)) if includeInverse else None
        check(lib.bdg_vandermonde1d(C.ptr(r), m, n, C.ptr(V), C.ptr(Vinv) if includeInverse else None))
        return (V, Vinv) if includeInverse else (V,)


class TriangleCubatureRules:
    """reference: include/TriangleCubatureRules.hpp:11-1830 (a C++-only class there): the tabulated symmetric rule of
    degree NCubature = 1..28 on the reference triangle; a computed conical-product rule beyond the table."""

    def __init__(self, NCubature):
        n = lib.bdg_cubature_rule_num_points(int(NCubature))
        if n < 0:
            raise ValueError("TriangleCubatureRules: degree must be >= 1")
        self._NCubature = int(NCubature)
        self._r, self._s, self._w = (np.empty(n, dtype=np.float64) for _ in range(3))
        as_p = lambda a: a.ctypes.data_as(POINTER(c_double))
        check(lib.bdg_cubature_rule(self._NCubature, as_p(self._r), as_p(self._s), as_p(self._w)))

    def NCubature(self):
        return self._NCubature

    def NumCubaturePoints(self):
        return self._r.size

    def rCoord(self):
        return self._r.copy()

    def sCoord(self):
        return self._s.copy()

    def weights(self):
        return self._w.copy()


class MeshManager:
    """reference: include/MeshManager.hpp:23-232; python names at pyblitzdg.cpp:101-112"""

    def __init__(self):
        h = c_void_p()
        check(lib.bdg_mesh_create(byref(h)))
        self._h = h

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            lib.bdg_mesh_destroy(h)

    def readMesh(self, gmshInputFile):
        check(lib.bdg_mesh_read(self._h, str(gmshInputFile).encode()))

    def writeMesh(self, gmshOutputFile):
        """Gmsh 2.2 ASCII of the triangles (the format readMesh takes)."""
        check(lib.bdg_mesh_write(self._h, str(gmshOutputFile).encode()))

    def writeCache(self, cacheFile):
        """Binary cache of the mesh with its connectivity, BC table and partition maps (one checksummed file)."""
        check(lib.bdg_mesh_write_cache(self._h, str(cacheFile).encode()))

    def readCache(self, cacheFile):
        """Restores a mesh written by writeCache without reading ASCII or rebuilding connectivity."""
        check(lib.bdg_mesh_read_cache(self._h, str(cacheFile).encode()))

    def buildMesh(self, EToV, Vert):
        """EToV: (K, 3) vertex ids (any numeric dtype, as the reference accepts float64);
        Vert: (Nv, 2|3) coordinates."""
        e = C.as_i32(np.asarray(EToV).astype(np.int64))
        v = C.as_f64(Vert)
        if e.ndim != 2 or e.shape[1] != 3 or v.ndim != 2:
            raise ValueError("buildMesh: EToV must be (K,3) and Vert (Nv,2|3)")
        check(lib.bdg_mesh_build(self._h, C.ptr(e), e.shape[0], C.ptr(v), v.shape[0], v.shape[1]))

    def buildBoxMesh(self, nx, ny, x0=-1.0, x1=1.0, y0=-1.0, y1=1.0, shuffleSeed=0):
        """Synthetic structured box, K = 2*nx*ny CCW triangles (benchmark configurations)."""
        check(lib.bdg_mesh_build_box(self._h, nx, ny, x0, x1, y0, y1, shuffleSeed))

    def partitionMesh(self, numPartitions):
        check(lib.bdg_mesh_partition(self._h, int(numPartitions)))

    def setBCType(self, bcType):
        b = C.as_i32(bcType).reshape(-1)
        check(lib.bdg_mesh_set_bctype(self._h, C.ptr(b), b.size))

    def _table(self, which):
        t = C.Table()
        check(lib.bdg_mesh_table(self._h, which, byref(t)))
        return C.table_to_numpy(t)

    numElements = property(lambda self: lib.bdg_mesh_num_elements(self._h))
    numVerts = property(lambda self: lib.bdg_mesh_num_verts(self._h))
    vertices = property(lambda self: self._table(C.MESH_VERTICES))
    elements = property(lambda self: self._table(C.MESH_ELEMENTS))
    bcType = property(lambda self: self._table(C.MESH_BCTYPE))
    EToE = property(lambda self: self._table(C.MESH_ETOE))
    EToF = property(lambda self: self._table(C.MESH_ETOF))
    elementPartitionMap = property(lambda self: self._table(C.MESH_EPART))
    vertexPartitionMap = property(lambda self: self._table(C.MESH_NPART))


class DGContext2D:
    """Read-only view of a TriangleNodesProvisioner's tables.
    reference: include/DGContext2D.hpp:9-258; python names at pyblitzdg.cpp:160-187."""

    _TABLES = {
        "filter": C.TRI_FILTER, "r": C.TRI_R, "s": C.TRI_S, "x": C.TRI_X, "y": C.TRI_Y,
        "Fscale": C.TRI_FSCALE, "Fmask": C.TRI_FMASK, "gather": C.TRI_GATHER, "scatter": C.TRI_SCATTER,
        "J": C.TRI_J, "rx": C.TRI_RX, "ry": C.TRI_RY, "sx": C.TRI_SX, "sy": C.TRI_SY, "nx": C.TRI_NX,
        "ny": C.TRI_NY, "Dr": C.TRI_DR, "Ds": C.TRI_DS, "Lift": C.TRI_LIFT, "vmapM": C.TRI_VMAPM,
        "vmapP": C.TRI_VMAPP, "V": C.TRI_V, "Vinv": C.TRI_VINV,
    }

    def __init__(self, nodes):
        self._nodes = nodes  # keeps the provisioner (and its mesh) alive

    def __getattr__(self, name):
        which = DGContext2D._TABLES.get(name)
        if which is None:
            raise AttributeError(name)
        return self._nodes._table(which)

    @property
    def numLocalPoints(self):
        return self._nodes._dims()[1]

    @property
    def numFacePoints(self):
        return self._nodes._dims()[2]

    @property
    def numElements(self):
        return self._nodes._dims()[3]

    @property
    def numFaces(self):
        return 3

    @property
    def order(self):
        return self._nodes._dims()[0]

    @property
    def BCmap(self):
        return self._nodes._bcmap()


class GaussFaceContext2D:
    """Gauss quadrature mesh on the element faces. reference: include/GaussFaceContext2D.hpp:68-104;
    python names at pyblitzdg.cpp:124-140. Owns its tables (a snapshot of the provisioner's coordinates
    at build time); every property access returns a fresh ndarray."""

    _TABLES = {"nx": C.GAUSS_NX, "ny": C.GAUSS_NY, "sJ": C.GAUSS_SJ, "J": C.GAUSS_J, "rx": C.GAUSS_RX,
               "ry": C.GAUSS_RY, "sx": C.GAUSS_SX, "sy": C.GAUSS_SY, "x": C.GAUSS_X, "y": C.GAUSS_Y,
               "W": C.GAUSS_W, "Interp": C.GAUSS_INTERP, "mapM": C.GAUSS_MAPM, "mapP": C.GAUSS_MAPP}

    def __init__(self, handle):
        self._h = handle

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            lib.bdg_gaussctx_destroy(h)

    def __getattr__(self, name):
        which = GaussFaceContext2D._TABLES.get(name)
        if which is None:
            raise AttributeError(name)
        t = C.Table()
        check(lib.bdg_gaussctx_table(self._h, which, byref(t)))
        return C.table_to_numpy(t)

    @property
    def NGauss(self):
        return lib.bdg_gaussctx_ngauss(self._h)

    @property
    def BCmap(self):
        n = lib.bdg_gaussctx_bcmap_num_tags(self._h)
        tags = (c_int * max(n, 1))()
        check(lib.bdg_gaussctx_bcmap_tags(self._h, tags, n))
        out = {}
        for i in range(n):
            p, cnt = C.POINTER(c_int)(), c_int()
            check(lib.bdg_gaussctx_bcmap_nodes(self._h, tags[i], byref(p), byref(cnt)))
            out[int(tags[i])] = np.ctypeslib.as_array(p, shape=(cnt.value,)).tolist() if cnt.value else []
        return out


class CubatureContext2D:
    """Volume cubature mesh of the elements. reference: include/CubatureContext2D.hpp:75-118; python
    names at pyblitzdg.cpp:142-158. MM / MMChol are (Np, Np, K) as in the reference."""

    _TABLES = {"r": C.CUB_R, "s": C.CUB_S, "w": C.CUB_WEIGHTS, "V": C.CUB_V, "rx": C.CUB_RX, "ry": C.CUB_RY,
               "sx": C.CUB_SX, "sy": C.CUB_SY, "J": C.CUB_J, "Dr": C.CUB_DR, "Ds": C.CUB_DS, "MM": C.CUB_MM,
               "MMChol": C.CUB_MMCHOL, "x": C.CUB_X, "y": C.CUB_Y, "W": C.CUB_W}

    def __init__(self, handle):
        self._h = handle

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            lib.bdg_cubctx_destroy(h)

    def __getattr__(self, name):
        which = CubatureContext2D._TABLES.get(name)
        if which is None:
            raise AttributeError(name)
        t = C.Table()
        check(lib.bdg_cubctx_table(self._h, which, byref(t)))
        arr = C.table_to_numpy(t)
        if name in ("MM", "MMChol"):
            n = int(round(np.sqrt(arr.shape[0])))
            arr = arr.reshape(n, n, arr.shape[1])
        return arr

    @property
    def NCubature(self):
        return lib.bdg_cubctx_order(self._h)

    @property
    def NumCubaturePoints(self):
        return lib.bdg_cubctx_num_points(self._h)


class TriangleNodesProvisioner:
    """reference: include/TriangleNodesProvisioner.hpp:32-427; python names at pyblitzdg.cpp:114-119"""

    def __init__(self, NOrder, meshManager):
        h = c_void_p()
        check(lib.bdg_trinodes_create(int(NOrder), meshManager._h, byref(h)))
        self._h = h
        self._mesh = meshManager  # the C++ object borrows the mesh

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            lib.bdg_trinodes_destroy(h)

    #: bumped by every method that changes a table a device solver may have copied (filter, BC lists, coordinates,
    #: repaired metric terms): caches of device images key on it (blitzdg_amd.sw2d.computeRHS)
    _tables_version = 0

    def buildFilter(self, Nc, s):
        check(lib.bdg_trinodes_build_filter(self._h, float(Nc), int(s)))
        self._tables_version += 1

    def buildBCHash(self, bcType):
        b = C.as_i32(bcType).reshape(-1)
        check(lib.bdg_trinodes_build_bchash(self._h, C.ptr(b), b.size))
        self._tables_version += 1

    def bedSlopes(self, H):
        """(Hx, Hy) as the variant-B driver builds them (reference src/sw2d/main.cpp:128-133)."""
        _, Np, _, K = self._dims()
        Hh = C.as_f64(H, (Np, K), "H")
        Hx, Hy = np.empty((Np, K)), np.empty((Np, K))
        check(lib.bdg_trinodes_bed_slopes(self._h, C.ptr(Hh), C.ptr(Hx), C.ptr(Hy)))
        return Hx, Hy

    def buildSpongeCoeff(self, mapO, spongeStrength, radInfl):
        """reference src/sw2d/main.cpp:516-556 (sw2d::buildSpongeCoeff)"""
        _, Np, _, K = self._dims()
        mo = C.as_i32(mapO).reshape(-1)
        out = np.empty((Np, K))
        check(lib.bdg_trinodes_sponge_coeff(self._h, C.ptr(mo) if mo.size else None, mo.size, float(spongeStrength),
                                            float(radInfl), C.ptr(out)))
        return out

    def splitElements(self, field):
        """(xnew, ynew, fieldnew), each (3, N^2*K): the field on N^2 linear triangles per element
        (reference src/TriangleNodesProvisioner.cpp:1154-1264)."""
        _, Np, _, K = self._dims()
        n = lib.bdg_trinodes_split_count(self._h) * K
        f = C.as_f64(field, (Np, K), "field")
        out = [np.empty((3, n)) for _ in range(3)]
        check(lib.bdg_trinodes_split_elements(self._h, C.ptr(f), *[C.ptr(o) for o in out]))
        return tuple(out)

    def splitOperators(self):
        """(IM, tri): (Np, Np) interpolation to the equispaced lattice, (N^2, 3) lattice-point indices."""
        _, Np, _, _ = self._dims()
        n = lib.bdg_trinodes_split_count(self._h)
        IM, tri = np.empty((Np, Np)), np.empty((n, 3), dtype=np.int32)
        check(lib.bdg_trinodes_split_operators(self._h, C.ptr(IM), C.ptr(tri)))
        return IM, tri

    def setCoordinates(self, x, y):
        _, Np, _, K = self._dims()
        xa, ya = C.as_f64(x, (Np, K), "x"), C.as_f64(y, (Np, K), "y")
        check(lib.bdg_trinodes_set_coordinates(self._h, C.ptr(xa), C.ptr(ya)))
        self._tables_version += 1

    def buildGaussFaceNodes(self, NGauss):
        """reference src/TriangleNodesProvisioner.cpp:207-381"""
        h = c_void_p()
        check(lib.bdg_trinodes_build_gauss_face_nodes(self._h, int(NGauss), byref(h)))
        return GaussFaceContext2D(h)

    def buildCubatureVolumeMesh(self, NCubature):
        """reference src/TriangleNodesProvisioner.cpp:81-205 (also recomputes the nodal J, rx, ry, sx, sy)"""
        h = c_void_p()
        check(lib.bdg_trinodes_build_cubature_volume_mesh(self._h, int(NCubature), byref(h)))
        self._tables_version += 1
        return CubatureContext2D(h)

    def dgContext(self):
        return DGContext2D(self)

    def _dims(self):
        o, np_, nfp, k = c_int(), c_int(), c_int(), c_int()
        check(lib.bdg_trinodes_dims(self._h, byref(o), byref(np_), byref(nfp), byref(k)))
        return o.value, np_.value, nfp.value, k.value

    def _table(self, which, copy=True):
        t = C.Table()
        check(lib.bdg_trinodes_table(self._h, which, byref(t)))
        return C.table_to_numpy(t, copy=copy)

    def _bcmap(self):
        n = lib.bdg_trinodes_bcmap_num_tags(self._h)
        tags = (c_int * max(n, 1))()
        check(lib.bdg_trinodes_bcmap_tags(self._h, tags, n))
        out = {}
        for i in range(n):
            p, cnt = C.POINTER(c_int)(), c_int()
            check(lib.bdg_trinodes_bcmap_nodes(self._h, tags[i], byref(p), byref(cnt)))
            out[int(tags[i])] = [int(p[j]) for j in range(cnt.value)]
        return out


class Nodes1DProvisioner:
    """reference: include/Nodes1DProvisioner.hpp:25-302; python names at pyblitzdg.cpp:66-81"""

    def __init__(self, NOrder, K, xLeft, xRight):
        h = c_void_p()
        check(lib.bdg_nodes1d_create(int(NOrder), int(K), float(xLeft), float(xRight), byref(h)))
        self._h = h
        self._order, self._K = int(NOrder), int(K)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            lib.bdg_nodes1d_destroy(h)

    def buildNodes(self):
        check(lib.bdg_nodes1d_build_nodes(self._h))

    def computeJacobian(self):
        check(lib.bdg_nodes1d_compute_jacobian(self._h))

    def _table(self, which):
        t = C.Table()
        check(lib.bdg_nodes1d_table(self._h, which, byref(t)))
        return C.table_to_numpy(t)

    numLocalPoints = property(lambda self: self._order + 1)
    numElements = property(lambda self: self._K)
    rGrid = property(lambda self: self._table(C.N1D_R))
    xGrid = property(lambda self: self._table(C.N1D_X))
    V = property(lambda self: self._table(C.N1D_V))
    Dr = property(lambda self: self._table(C.N1D_DR))
    rx = property(lambda self: self._table(C.N1D_RX))
    J = property(lambda self: self._table(C.N1D_J))
    Fscale = property(lambda self: self._table(C.N1D_FSCALE))
    Fmask = property(lambda self: self._table(C.N1D_FMASK))
    Fx = property(lambda self: self._table(C.N1D_FX))
    Lift = property(lambda self: self._table(C.N1D_LIFT))
    EToV = property(lambda self: self._table(C.N1D_ETOV))
    EToE = property(lambda self: self._table(C.N1D_ETOE))
    EToF = property(lambda self: self._table(C.N1D_ETOF))
    vmapM = property(lambda self: self._table(C.N1D_VMAPM))
    vmapP = property(lambda self: self._table(C.N1D_VMAPP))
    nx = property(lambda self: self._table(C.N1D_NX))
    mapI = property(lambda self: lib.bdg_nodes1d_map_i(self._h))
    mapO = property(lambda self: lib.bdg_nodes1d_map_o(self._h))


def advec1dComputeRHS(u, c, nodes1d):
    """The reference script's ``advec1dComputeRHS(u, c, nodes1d)`` (advec1d.py:12-39; C++:
    src/advec1d/main.cpp:126-188): upwind flux, outflow at mapO, zero inflow at mapI. Host code --
    advec1d is the CPU plumbing configuration."""
    Np, K = nodes1d.numLocalPoints, nodes1d.numElements
    ua = C.as_f64(u, (Np, K), "u")
    out = np.empty((Np, K))
    check(lib.bdg_nodes1d_advec_rhs(nodes1d._h, C.ptr(ua), float(c), C.ptr(out)))
    return out


def advec1dRun(N=4, K=30, xmin=-1.0, xmax=4.0, c=0.1, CFL=0.8, finalTime=20.0):
    """The reference's bin/advec1d (src/advec1d/main.cpp:35-122) with N, K as arguments;
    host-only LSERK4 loop. Returns (max-norm error vs exact, number of steps)."""
    err, steps = c_double(), c_int()
    check(lib.bdg_advec1d_run(N, K, xmin, xmax, c, CFL, finalTime, byref(err), byref(steps)))
    return err.value, steps.value


def burgers1dComputeRHS(u, t, c, alpha, nu, nodes1d):
    """blitzdg::burgers1d::computeRHS (src/burgers1d/main.cpp:129-226) on the provisioner's own grid: viscous Burgers as a first-order
    system, local Lax-Friedrichs flux, the travelling wave as boundary data. Host code (CPU plumbing beside advec1d)."""
    Np, K = nodes1d.numLocalPoints, nodes1d.numElements
    ua = C.as_f64(u, (Np, K), "u")
    out = np.empty((Np, K))
    check(lib.bdg_nodes1d_burgers_rhs(nodes1d._h, C.ptr(ua), float(t), float(c), float(alpha), float(nu), C.ptr(out)))
    return out


def burgers1dRun(N=6, K=40, xmin=-5.0, xmax=5.0, alpha=1.0, nu=0.1, c=0.5, CFL=0.75, finalTime=0.1):
    """The reference's bin/burgers1d (src/burgers1d/main.cpp:28-115, its constants as defaults); host-only LSERK4 loop.
    Returns (max-norm error vs the travelling wave, number of steps)."""
    err, steps = c_double(), c_int()
    check(lib.bdg_burgers1d_run(N, K, xmin, xmax, alpha, nu, c, CFL, finalTime, byref(err), byref(steps)))
    return err.value, steps.value


class VtkOutputter:
    """Writes nodal fields as *.vtu files for Paraview; names of the reference's binding
    (src/pyblitzdg/pyblitzdg.cpp:189-192). No VTK library involved."""

    def __init__(self, TriangleNodesProvisioner):
        self._nodes = TriangleNodesProvisioner

    @staticmethod
    def generateFileName(fieldName, fileNumber):
        return f"{fieldName}{int(fileNumber):07d}.vtu"

    def writeFieldToFile(self, fileName, field, fieldName):
        _, Np, _, K = self._nodes._dims()
        f = C.as_f64(field, (Np, K), "field")
        check(lib.bdg_trinodes_write_vtu(self._nodes._h, str(fileName).encode(), C.ptr(f), str(fieldName).encode()))

    def writeFieldsToFiles(self, fields, tstep):
        for name, field in fields.items():
            self.writeFieldToFile(self.generateFileName(name, tstep), field, name)

    def splitNodes(self):
        """Coordinates of the small triangles' corners, (3, K*N^2) each (third entry: zeros)."""
        _, Np, _, K = self._nodes._dims()
        return self._nodes.splitElements(np.zeros((Np, K)))

    def writeSolverFields(self, solver, tstep, directory="."):
        """eta, u, v of a device-resident ``sw2d.Sw2dSolver`` to ``<directory>/{eta,u,v}NNNNNNN.vtu``:
        primitive variables and lattice interpolation happen on the device, the host only cuts the
        lattice into triangles and writes. Returns the file paths."""
        import os
        order, Np, _, K = self._nodes._dims()
        if not hasattr(self, "_lattice"):
            IM, tri = self._nodes.splitOperators()
            ctx = self._nodes.dgContext()
            if order == 1:
                self._lattice = (None, lambda a: a, ctx.x, ctx.y)
            else:
                cut = lambda a: np.ascontiguousarray(a[tri.T, :].transpose(0, 2, 1).reshape(3, -1))  # noqa: E731
                xt, yt, _ = self.splitNodes()                    # (3, K*N^2), element-major
                self._lattice = (IM, cut, xt, yt)
        IM, cut, xt, yt = self._lattice
        paths = []
        fields = list(zip(("eta", "u", "v"), solver.outputFields(IM)))
        if getattr(solver, "fields", 3) == 4:
            fields.append(("N", solver.outputTracer(IM)))       # the reference script's fourth output field
        for name, lat in fields:
            ft = cut(lat)
            path = os.path.join(directory, self.generateFileName(name, tstep))
            check(lib.bdg_write_vtu_triangles(path.encode(), C.ptr(xt), C.ptr(yt), C.ptr(np.ascontiguousarray(ft)),
                                              ft.shape[1], name.encode()))
            paths.append(path)
        return paths
