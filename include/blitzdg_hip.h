/* blitzdg_hip.h -- C ABI of the MI355X-native blitzdg hot path.
 *
 * One shared library (libblitzdg_hip.so) with two groups of entry points:
 *
 *  (1) Host setup handles (CPU): mesh, 2-D triangle nodes provisioner, 1-D nodes
 *      provisioner. They build exactly the tables the reference's C++ classes
 *      build and export them as borrowed (pointer, rows, cols) views. This is the
 *      seam the reference crosses with boost::python in src/pyblitzdg/pyblitzdg.cpp
 *      :59-201 (MeshManager, TriangleNodesProvisioner/DGContext2D,
 *      Nodes1DProvisioner, LSERK4); blitzdg_amd/pyblitzdg.py binds it with ctypes.
 *
 *  (2) The device-resident sw2d solver (HIP, gfx950): right-hand-side evaluation
 *      and Runge-Kutta stage updates of the 2-D shallow-water nodal DG scheme.
 *      bdg_sw2d_rhs replaces blitzdg::sw2d::computeRHS
 *      (reference src/sw2d-simple/main.cpp:181-356, decl SW2d.hpp:15);
 *      bdg_sw2d_step_lserk4 replaces the LSERK4 stage loop
 *      (src/advec1d/main.cpp:92-102 with include/LSERK4.hpp:15-29);
 *      bdg_sw2d_step_rk2 replaces the midpoint-RK2 + filter loop body
 *      (src/sw2d-simple/main.cpp:132-151); bdg_sw2d_compute_dt replaces the
 *      time-step/blow-up reductions (src/sw2d-simple/main.cpp:98-109,153-167).
 *
 * Conventions: every (rows, K) field/table is row-major fp64 with K (the element
 * index) contiguous, as in the reference (include/Types.hpp:16-18); index tables
 * are int32 and use the reference's column-wise node numbering n + Np*k.
 * Every function returns 0 on success or a BDG_ERR_* code; bdg_last_error()
 * returns a thread-local message. No exceptions cross this boundary. A handle
 * must be used from one host thread at a time.
 */
#ifndef BLITZDG_HIP_H
#define BLITZDG_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BDG_OK 0
#define BDG_ERR_ARGUMENT 1  /* null pointer, bad size, unsupported order ...   */
#define BDG_ERR_RUNTIME 2   /* I/O or setup failure (message has the details) */
#define BDG_ERR_HIP 3       /* a HIP call failed / no usable device            */
#define BDG_ERR_UNSTABLE 4  /* NaN or |eta| > 1e8 detected ("A numerical instability has occurred!") */

typedef struct bdg_mesh bdg_mesh;
typedef struct bdg_trinodes bdg_trinodes;
typedef struct bdg_nodes1d bdg_nodes1d;
typedef struct bdg_gaussctx bdg_gaussctx;
typedef struct bdg_cubctx bdg_cubctx;
typedef struct bdg_sw2d bdg_sw2d;

const char* bdg_last_error(void);
int bdg_version(void);

/* ---------------------------------------------------------------- tables */

/* dtype codes of a table view */
#define BDG_F64 0
#define BDG_I32 1

/* Borrowed view of a host table; valid until the owning handle is destroyed or rebuilt. */
typedef struct bdg_table {
    const void* data;
    int rows;   /* 1-D tables: rows = length, cols = 1 */
    int cols;
    int dtype;  /* BDG_F64 or BDG_I32 */
} bdg_table;

/* ---------------------------------------------------------------- MeshManager
 * reference: include/MeshManager.hpp:23-232, src/MeshManager.cpp */
enum {
    BDG_MESH_VERTICES = 0, /* (Nv, 3) f64 */
    BDG_MESH_ELEMENTS = 1, /* (K, 3) i32, CCW */
    BDG_MESH_ETOE = 2,     /* (K, 3) i32 */
    BDG_MESH_ETOF = 3,     /* (K, 3) i32 */
    BDG_MESH_BCTYPE = 4,   /* (K, 3) i32: 0 interior, 3 wall ... */
    BDG_MESH_EPART = 5,    /* (K) i32, after bdg_mesh_partition */
    BDG_MESH_NPART = 6     /* (Nv) i32, after bdg_mesh_partition */
};
int bdg_mesh_create(bdg_mesh** out);
void bdg_mesh_destroy(bdg_mesh* mesh);
int bdg_mesh_read(bdg_mesh* mesh, const char* gmsh_path);                 /* readMesh   */
int bdg_mesh_write(const bdg_mesh* mesh, const char* gmsh_path);          /* Gmsh 2.2 ASCII, what readMesh takes */
/* binary cache of a mesh with its connectivity, BC table and partition maps (MeshManager::writeCache / readCache; the
 * reference has none: SURVEY 8f.2). read_cache refuses files whose magic, version, sizes, index ranges or checksum do not fit. */
int bdg_mesh_write_cache(const bdg_mesh* mesh, const char* cache_path);
int bdg_mesh_read_cache(bdg_mesh* mesh, const char* cache_path);
int bdg_mesh_build(bdg_mesh* mesh, const int* etov, int num_elements,     /* buildMesh  */
                   const double* vert, int num_verts, int dim);
int bdg_mesh_build_box(bdg_mesh* mesh, int nx, int ny, double x0, double x1, double y0, double y1,
                       unsigned long long shuffle_seed);                  /* synthetic box, K = 2*nx*ny */
int bdg_mesh_set_bctype(bdg_mesh* mesh, const int* bctype, int n);        /* setBCType  */
int bdg_mesh_partition(bdg_mesh* mesh, int num_partitions);               /* partitionMesh */
int bdg_mesh_num_elements(const bdg_mesh* mesh);
int bdg_mesh_num_verts(const bdg_mesh* mesh);
int bdg_mesh_table(const bdg_mesh* mesh, int which, bdg_table* out);

/* ---------------------------------------------------------------- TriangleNodesProvisioner / DGContext2D
 * reference: include/TriangleNodesProvisioner.hpp:32-427, include/DGContext2D.hpp:9-258 */
enum {
    BDG_TRI_R = 0, BDG_TRI_S, BDG_TRI_X, BDG_TRI_Y, BDG_TRI_V, BDG_TRI_VINV, BDG_TRI_DR, BDG_TRI_DS,
    BDG_TRI_DRW, BDG_TRI_DSW, BDG_TRI_LIFT, BDG_TRI_FILTER, BDG_TRI_J, BDG_TRI_RX, BDG_TRI_RY, BDG_TRI_SX,
    BDG_TRI_SY, BDG_TRI_NX, BDG_TRI_NY, BDG_TRI_FSCALE, BDG_TRI_FMASK, BDG_TRI_FX, BDG_TRI_FY,
    BDG_TRI_VMAPM, BDG_TRI_VMAPP, BDG_TRI_MAPP, BDG_TRI_VMAPB, BDG_TRI_MAPB, BDG_TRI_GATHER, BDG_TRI_SCATTER
};
int bdg_trinodes_create(int order, const bdg_mesh* mesh, bdg_trinodes** out); /* mesh must outlive it */
void bdg_trinodes_destroy(bdg_trinodes* nodes);
int bdg_trinodes_build_filter(bdg_trinodes* nodes, double Nc, int s);
int bdg_trinodes_build_bchash(bdg_trinodes* nodes, const int* bctype, int n); /* appends, as the reference */
int bdg_trinodes_set_coordinates(bdg_trinodes* nodes, const double* x, const double* y);
int bdg_trinodes_dims(const bdg_trinodes* nodes, int* order, int* np, int* nfp, int* num_elements);
int bdg_trinodes_table(const bdg_trinodes* nodes, int which, bdg_table* out);
/* BCmap: number of tags; tag list; node list of one tag (borrowed). */
int bdg_trinodes_bcmap_num_tags(const bdg_trinodes* nodes);
int bdg_trinodes_bcmap_tags(const bdg_trinodes* nodes, int* tags, int capacity);
int bdg_trinodes_bcmap_nodes(const bdg_trinodes* nodes, int tag, const int** nodes_out, int* count);

/* Output step after the path (reference TriangleNodesProvisioner::splitElements,
 * src/TriangleNodesProvisioner.cpp:1154-1264, and VtkOutputter, include/VtkOutputter.hpp:30-99).
 * split_count = N^2 linear triangles per element. split_operators: IM (Np, Np) interpolation to the
 * equispaced lattice and 3*N^2 lattice-point indices. split_elements: xnew, ynew, fieldnew as
 * (3, N^2*K). write_vtu: one field to a *.vtu file (raw appended binary; no VTK library). */
int bdg_trinodes_split_count(const bdg_trinodes* nodes);
int bdg_trinodes_split_operators(const bdg_trinodes* nodes, double* IM, int* local_triangles);
int bdg_trinodes_split_elements(const bdg_trinodes* nodes, const double* field, double* xnew, double* ynew,
                                double* fieldnew);
int bdg_trinodes_write_vtu(const bdg_trinodes* nodes, const char* path, const double* field,
                           const char* field_name);

/* ---------------------------------------------------------------- GaussFaceContext2D / CubatureContext2D
 * reference: TriangleNodesProvisioner::buildGaussFaceNodes (src/TriangleNodesProvisioner.cpp:207-381,
 * include/GaussFaceContext2D.hpp:68-84) and ::buildCubatureVolumeMesh (:81-205,
 * include/CubatureContext2D.hpp:75-96); python names at src/pyblitzdg/pyblitzdg.cpp:116-117, 124-158.
 * The contexts own their tables (built from the provisioner's CURRENT coordinates) and outlive it.
 * Gauss tables are (3*NGauss, K), Interp (3*NGauss, Np), mapM/mapP (3*NGauss*K) flat ids g + 3*NGauss*k.
 * Cubature tables are (Ncub, K), V/Dr/Ds (Ncub, Np), r/s/w (Ncub); MM and MMChol are the reference's
 * (Np, Np, K) tensors viewed as (Np*Np, K). build_cubature_volume_mesh also recomputes the provisioner's
 * nodal J, rx, ry, sx, sy from its current coordinates, as the reference does. The cubature rule is the
 * reference's tabulated symmetric rule for degrees 1..28 (36 points at degree 12) and a computed conical-product
 * rule beyond its table (include/blitzdg/TriangleCubatureRules.hpp). */
enum {
    BDG_GAUSS_NX = 0, BDG_GAUSS_NY, BDG_GAUSS_SJ, BDG_GAUSS_J, BDG_GAUSS_RX, BDG_GAUSS_RY, BDG_GAUSS_SX,
    BDG_GAUSS_SY, BDG_GAUSS_X, BDG_GAUSS_Y, BDG_GAUSS_W, BDG_GAUSS_INTERP, BDG_GAUSS_MAPM, BDG_GAUSS_MAPP
};
enum {
    BDG_CUB_R = 0, BDG_CUB_S, BDG_CUB_WEIGHTS, BDG_CUB_V, BDG_CUB_RX, BDG_CUB_RY, BDG_CUB_SX, BDG_CUB_SY,
    BDG_CUB_J, BDG_CUB_DR, BDG_CUB_DS, BDG_CUB_MM, BDG_CUB_MMCHOL, BDG_CUB_X, BDG_CUB_Y, BDG_CUB_W
};
int bdg_trinodes_build_gauss_face_nodes(bdg_trinodes* nodes, int NGauss, bdg_gaussctx** out);
void bdg_gaussctx_destroy(bdg_gaussctx* ctx);
int bdg_gaussctx_ngauss(const bdg_gaussctx* ctx);
int bdg_gaussctx_table(const bdg_gaussctx* ctx, int which, bdg_table* out);
int bdg_gaussctx_bcmap_num_tags(const bdg_gaussctx* ctx);
int bdg_gaussctx_bcmap_tags(const bdg_gaussctx* ctx, int* tags, int capacity);
int bdg_gaussctx_bcmap_nodes(const bdg_gaussctx* ctx, int tag, const int** nodes_out, int* count);
int bdg_trinodes_build_cubature_volume_mesh(bdg_trinodes* nodes, int NCubature, bdg_cubctx** out);
void bdg_cubctx_destroy(bdg_cubctx* ctx);
int bdg_cubctx_num_points(const bdg_cubctx* ctx);
int bdg_cubctx_order(const bdg_cubctx* ctx);
int bdg_cubctx_table(const bdg_cubctx* ctx, int which, bdg_table* out);
/* TriangleCubatureRules (reference include/TriangleCubatureRules.hpp:1807-1830): number of points of the rule of
 * degree NCubature (-1: bad degree), and its rCoord / sCoord / weights copied into caller arrays of that length. */
int bdg_cubature_rule_num_points(int NCubature);
int bdg_cubature_rule(int NCubature, double* r, double* s, double* w);

/* ---------------------------------------------------------------- Nodes1DProvisioner
 * reference: include/Nodes1DProvisioner.hpp:25-302 */
enum {
    BDG_N1D_R = 0, BDG_N1D_X, BDG_N1D_V, BDG_N1D_VINV, BDG_N1D_DR, BDG_N1D_LIFT, BDG_N1D_J, BDG_N1D_RX,
    BDG_N1D_NX, BDG_N1D_FMASK, BDG_N1D_FX, BDG_N1D_FSCALE, BDG_N1D_ETOV, BDG_N1D_ETOE, BDG_N1D_ETOF,
    BDG_N1D_VMAPM, BDG_N1D_VMAPP
};
int bdg_nodes1d_create(int order, int num_elements, double xmin, double xmax, bdg_nodes1d** out);
void bdg_nodes1d_destroy(bdg_nodes1d* nodes);
int bdg_nodes1d_build_nodes(bdg_nodes1d* nodes);
int bdg_nodes1d_compute_jacobian(bdg_nodes1d* nodes);
int bdg_nodes1d_map_i(const bdg_nodes1d* nodes);
int bdg_nodes1d_map_o(const bdg_nodes1d* nodes);
int bdg_nodes1d_table(const bdg_nodes1d* nodes, int which, bdg_table* out);

/* LSERK4 coefficients (reference include/LSERK4.hpp:15-29); 5 entries each. */
int bdg_lserk4_num_stages(void);
const double* bdg_lserk4_a(void);
const double* bdg_lserk4_b(void);

/* VandermondeBuilders::buildVandermondeMatrix_numpy (reference include/VandermondeBuilders.hpp:76-105, Python name
 * VandermondeBuilder.buildVandermondeMatrix, src/pyblitzdg/pyblitzdg.cpp:92-93): V(i, j) = P_j^(0,0)(r_i), the orthonormal
 * Legendre polynomials at the num_points entries of r, j = 0 .. num_cols - 1 (the reference: num_cols = order + 1, or
 * num_points when order < 0). V is (num_points, num_cols) row-major; with Vinv != NULL (needs num_points == num_cols)
 * the inverse is written there too. Host only. */
int bdg_vandermonde1d(const double* r, int num_points, int num_cols, double* V, double* Vinv);

/* advec1d: CPU plumbing config (reference src/advec1d/main.cpp:35-122). Runs the
 * LSERK4 loop on the host to t >= final_time and returns the max-norm error
 * against the translated Gaussian. No GPU involved. */
/* advec1d::computeRHS(u, c, nodes1D, RHS) (reference src/advec1d/main.cpp:126-188; Python twin
 * advec1d.py:12-39) on the host: u, rhs are (Np, K). buildNodes and computeJacobian must have run. */
int bdg_nodes1d_advec_rhs(bdg_nodes1d* nodes, const double* u, double c, double* rhs);
int bdg_advec1d_run(int order, int num_elements, double xmin, double xmax, double c, double cfl,
                    double final_time, double* max_error, int* num_steps);

/* burgers1d: the second 1-D solver on Nodes1DProvisioner + LSERK4 (reference src/burgers1d/main.cpp:28-115 driver, :129-226
 * RHS). Host only. rhs: burgers1d::computeRHS(u, x, t, c, alpha, nu, nodes1D, RHS) with x the provisioner's own grid;
 * run: the driver loop to t >= final_time, max-norm error against the travelling wave Burgers2. */
int bdg_nodes1d_burgers_rhs(bdg_nodes1d* nodes, const double* u, double t, double c, double alpha, double nu, double* rhs);
int bdg_burgers1d_run(int order, int num_elements, double xmin, double xmax, double alpha, double nu, double c, double cfl,
                      double final_time, double* max_error, int* num_steps);

/* ---------------------------------------------------------------- sw2d device solver */

/* Host tables a solver is created from (all borrowed for the duration of the call). */
typedef struct bdg_sw2d_desc {
    int order;          /* N: 1..BDG_SW2D_MAX_ORDER                                   */
    int num_elements;   /* K                                                          */
    const double* Dr;   /* (Np, Np)                                                   */
    const double* Ds;   /* (Np, Np)                                                   */
    const double* Lift; /* (Np, 3*Nfp)                                                */
    const double* Filter; /* (Np, Np) or NULL (no filter available)                   */
    const double* rx; const double* sx; const double* ry; const double* sy; /* (Np, K) */
    const double* nx; const double* ny; const double* Fscale;               /* (3*Nfp, K) */
    const int* vmapM;   /* (3*Nfp*K) volume node of each face node, n + Np*k; may be NULL */
    const int* vmapP;   /* (3*Nfp*K) neighbour node of each face node                 */
    const int* mapW;    /* BCmap[3]: flat face-node indices of reflective-wall nodes  */
    int num_wall;
    double g;           /* gravitational acceleration                                 */
    int device;         /* HIP device ordinal                                         */
    int flags;          /* BDG_SW2D_* bits                                            */
    /* ---- optional "variant D" physics of the reference's Python RHS (swhelpers/rhs.py:178-311);
     * all zero / NULL gives variant A above. Straight-sided elements only. */
    int num_fields;     /* 0 or 3: h, hu, hv;  4: + passive tracer hN (F4 = hN u, G4 = hN v)      */
    int sources;        /* nonzero: RHS2 += f hv - CD|u|u - g h zx;  RHS3 -= f hu - CD|u|v;  RHS3 -= g h zy */
    const double* zx;   /* (Np, K) bed slope or NULL (= 0)                            */
    const double* zy;
    const double* coriolis; /* (Np, K) Coriolis parameter f, or NULL: coriolis_const    */
    double coriolis_const;
    double drag;        /* CD                                                         */
} bdg_sw2d_desc;

#define BDG_SW2D_MAX_ORDER 8 /* 7 and 8: straight-sided (affine) geometry only */
#define BDG_SW2D_REORDER 1u /* renumber elements internally for gather locality (results are
                               returned in the caller's numbering either way)         */
#define BDG_SW2D_KEEP_ORDER 4u /* never renumber (default: renumber when the mean face-neighbour
                               distance exceeds 4*sqrt(K) slots, e.g. a shuffled mesh). Required
                               for bdg_sw2d_set_partition, whose element ranges are in caller order */
#define BDG_SW2D_NODAL_GEOMETRY 2u /* always read rx..sy, nx, ny, Fscale per node (general path).
                               Default: if they are constant per element / per face to round-off
                               (straight-sided elements: everything the reference's provisioner
                               builds), one value per element / face is kept instead.      */

int bdg_device_count(void); /* HIP devices visible to this process (0 if none / no driver) */
int bdg_sw2d_create(const bdg_sw2d_desc* desc, bdg_sw2d** out);
/* Convenience: take every table from a nodes provisioner (wall nodes = BCmap[3]). */
int bdg_sw2d_create_from_nodes(const bdg_trinodes* nodes, double g, int device, int flags, bdg_sw2d** out);
void bdg_sw2d_destroy(bdg_sw2d* s);

/* State I/O: host (Np, K) row-major arrays in the caller's element numbering. */
int bdg_sw2d_set_state(bdg_sw2d* s, const double* h, const double* hu, const double* hv);
int bdg_sw2d_get_state(bdg_sw2d* s, double* h, double* hu, double* hv);
/* Still-water depth H used only by the eta = h - H blow-up check; default: none (check h). */
int bdg_sw2d_set_bathymetry(bdg_sw2d* s, const double* H);

/* Output step: the drivers' primitive fields eta = h - H (h if no bathymetry was set), u = hu/h,
 * v = hv/h of the resident state as host (Np, K) arrays; with IM != NULL ((Np, Np), from
 * bdg_trinodes_split_operators) they are interpolated to each element's equispaced lattice on the
 * device first (what splitElements does before a *.vtu is written). NULL outputs are skipped. */
int bdg_sw2d_output_fields(bdg_sw2d* s, const double* IM, double* eta, double* u, double* v);
/* Four-field solvers: the tracer concentration N = hN / h (what the reference's sw2d.py writes out), same
 * optional lattice interpolation. */
int bdg_sw2d_output_tracer(bdg_sw2d* s, const double* IM, double* tracer);
/* Writes linear triangles ((3, num_triangles) x, y, field; one column per triangle) as a *.vtu. */
int bdg_write_vtu_triangles(const char* path, const double* x, const double* y, const double* field,
                            int num_triangles, const char* field_name);

/* Drop-in for computeRHS: host fields in, host RHS out (upload, one kernel, download).
 * Does not disturb the resident state. filter != 0 applies Filter to the result. */
int bdg_sw2d_rhs(bdg_sw2d* s, const double* h, const double* hu, const double* hv, double* rhs1,
                 double* rhs2, double* rhs3, int filter);

/* Four-field forms for solvers created with num_fields = 4 (hN: tracer). The three-field
 * functions above refuse such a solver and vice versa. */
int bdg_sw2d_set_state4(bdg_sw2d* s, const double* h, const double* hu, const double* hv, const double* hN);
int bdg_sw2d_get_state4(bdg_sw2d* s, double* h, double* hu, double* hv, double* hN);
int bdg_sw2d_rhs4(bdg_sw2d* s, const double* h, const double* hu, const double* hv, const double* hN,
                  double* rhs1, double* rhs2, double* rhs3, double* rhs4, int filter);
int bdg_sw2d_num_fields(const bdg_sw2d* s);

/* ---- "variant B": the physics of the reference's C++ sw2d driver,
 * computeRHS(fields, numParams, physParams, DGContext2D, t) -- src/sw2d/main.cpp:279-484:
 * still-water depth H with star states at the faces, open-boundary nodes (BCmap[2]) driven by
 *   hP = HM + tide_amplitude cos(2 pi t / tide_period) 1/2 (tanh(tide_ramp (t - tide_period)) + 1),
 * one global Lax-Friedrichs speed, and the sources RHS2 += g h Hx - CD u|u| + f hv,
 * RHS3 += g h Hy - CD v|u| - f hu. After this call every RHS / stepping entry point of a 3-field
 * straight-sided solver evaluates variant B at the solver's model time (bdg_sw2d_set_time; the
 * steppers advance it by dt per step, both Heun evaluations at the old level as main.cpp:211-236).
 * sponge: (Np, K) coefficient of hu /= 1 + c hu^2 used by bdg_sw2d_step_ssprk2 instead of the
 * scalar. Hx, Hy are the caller's (bdg_trinodes_bed_slopes builds them as main.cpp:128-133). */
typedef struct bdg_sw2d_vb_desc {
    const double* H;       /* (Np, K)                                    */
    const double* Hx;      /* (Np, K)                                    */
    const double* Hy;      /* (Np, K)                                    */
    const int* mapO;       /* flat face-node indices of open-boundary nodes, or NULL */
    int num_out;
    double drag;           /* physParams.CD                              */
    double coriolis;       /* physParams.f                               */
    double tide_amplitude; /* reference: 3.0                             */
    double tide_period;    /* reference: 3600*12.42                      */
    double tide_ramp;      /* reference: 0.15/3600                       */
    const double* sponge;  /* (Np, K) or NULL                            */
} bdg_sw2d_vb_desc;
int bdg_sw2d_enable_variant_b(bdg_sw2d* s, const bdg_sw2d_vb_desc* desc);
int bdg_sw2d_set_time(bdg_sw2d* s, double t);
int bdg_sw2d_get_time(const bdg_sw2d* s, double* t);
/* The global Lax-Friedrichs speed of the most recent variant-B evaluation. */
int bdg_sw2d_global_speed(bdg_sw2d* s, double* lam);
/* Host helpers for the variant-B driver set-up: Hx, Hy = Filter (rx Dr H + sx Ds H, ry Dr H + sy Ds H)
 * (main.cpp:128-133; buildFilter must have been called), and buildSpongeCoeff (main.cpp:516-556). */
int bdg_trinodes_bed_slopes(const bdg_trinodes* nodes, const double* H, double* Hx, double* Hy);
int bdg_trinodes_sponge_coeff(const bdg_trinodes* nodes, const int* mapO, int num_out, double strength,
                              double radius, double* coeff);

/* Resident time stepping (state stays in HBM). */
int bdg_sw2d_step_lserk4(bdg_sw2d* s, double dt, int num_steps);          /* 5 fused stages per step */
int bdg_sw2d_lserk4_stages(bdg_sw2d* s, double dt, int num_stages);       /* stage i = count % 5      */
int bdg_sw2d_step_rk2(bdg_sw2d* s, double dt, int num_steps, int filter); /* midpoint RK2            */
/* SSP-RK2 (Heun) of the variant-B driver, reference src/sw2d/main.cpp:211-235:
 * q1 = sp(q + dt R(q)); q = sp((q + q1 + dt R(q1))/2), sp(x) = x/(1 + sponge_coeff x^2) on hu, hv
 * (sponge_coeff = 0: plain Heun). */
int bdg_sw2d_step_ssprk2(bdg_sw2d* s, double dt, int num_steps, int filter, double sponge_coeff);
/* dt = CFL / ((N+1)^2 * 0.5 * max_i |Fscale_i| * (|u|+sqrt(g h))[vmapM_i]); also returns
 * max|eta| (or max|h|) and BDG_ERR_UNSTABLE on NaN / > 1e8. */
int bdg_sw2d_compute_dt(bdg_sw2d* s, double cfl, double* dt, double* eta_max);
/* Reference driver loop body (src/sw2d-simple/main.cpp:121-171): RK2 step with
 * filter, blow-up check, adaptive dt; runs until t >= final_time or max_steps. */
int bdg_sw2d_run_adaptive(bdg_sw2d* s, double cfl, double final_time, int max_steps, int filter,
                          double* t_inout, double* dt_inout, int* steps_done);

/* ---- multi-GPU: element partition with a ghost layer (no reference analogue; the reference
 * only computes METIS partition vectors, src/MeshManager.cpp:491-544, and never uses them).
 * The solver is created on a LOCAL mesh whose elements are ordered
 *   [ interior | partition-boundary | ghost ]            (ghost = owned by another rank)
 * Only [0, num_owned) are updated; ghost state is refreshed each stage by the caller:
 *   pack (device buffer of num_send*3*Np doubles, element-major) -> exchange (RCCL, caller's
 *   job) -> unpack into the ghost slots, while the interior elements are already computing. */
int bdg_sw2d_set_partition(bdg_sw2d* s, int num_interior, int num_owned, const int* send_elements,
                           int num_send);
int bdg_sw2d_halo_doubles_per_element(const bdg_sw2d* s);
int bdg_sw2d_halo_pack(bdg_sw2d* s, void* send_buffer_device);
int bdg_sw2d_halo_unpack(bdg_sw2d* s, const void* recv_buffer_device);
/* One LSERK4 stage over part of the owned elements: 0 = interior only (does not advance the
 * stage), 1 = partition-boundary elements then advance, 2 = all owned elements then advance. */
int bdg_sw2d_lserk4_stage_part(bdg_sw2d* s, double dt, int part);
/* Native transport: RCCL point-to-point over xGMI, driven entirely from this library (bound
 * with dlopen on first use). Rank 0 creates a 128-byte id and hands it to every rank by any
 * means; each rank then gives its neighbour list: for peer i, it sends the elements
 * send_elements[send_start[i] .. +send_count[i]) (of bdg_sw2d_set_partition) and receives
 * recv_count[i] ghost elements into ghost slots recv_start[i].. (relative to num_owned).
 * A peer may be this rank itself (loop-back: the send range is delivered to the receive range on
 * the same device) -- used to rehearse the per-stage schedule of an N-way split on one GPU. */
int bdg_comm_unique_id(void* id_out, int capacity);
int bdg_sw2d_comm_init(bdg_sw2d* s, int rank, int world, const void* unique_id, const int* peer_ranks,
                       const int* send_start, const int* send_count, const int* recv_start,
                       const int* recv_count, int num_peers);
/* num_stages LSERK4 stages with the ghost exchange overlapped with the interior elements:
 * pack -> {grouped ncclSend/ncclRecv on the comm stream || interior kernel} -> unpack -> boundary.
 * Ghost elements are inputs of a stage only: after this call (and after bdg_sw2d_group_lserk4_stages)
 * the ghost columns returned by bdg_sw2d_get_state / read by bdg_sw2d_rhs_resident and the output
 * functions are UNDEFINED -- they hold whatever an earlier stage received, or, where the boundary kernel
 * reads the received records directly, their initial values. Owned columns are exact. The next
 * exchanged stage refreshes the ghosts before it reads them. */
int bdg_sw2d_lserk4_stages_exchanged(bdg_sw2d* s, double dt, int num_stages);
/* The drivers' two-evaluation schemes in a partitioned run (bdg_sw2d_step_rk2 / bdg_sw2d_step_ssprk2 with an
 * exchange in front of EACH evaluation, of the state that evaluation reads): midpoint RK2 + filter of the sw2d.py /
 * sw2d-simple drivers (three or four fields, sources) and Heun + sponge of the variant-B driver. With variant B
 * enabled every evaluation -- here and in bdg_sw2d_lserk4_stages_exchanged -- also reduces the one global
 * Lax-Friedrichs speed (reference src/sw2d/main.cpp:414) over all ranks: a speed pass over the owned elements and
 * one 8-byte ncclAllReduce(max) on the solver's stream; these paths are not overlapped with computation, because
 * no element of an evaluation can start before the speed is known. Results equal the single-domain run bit for bit. */
int bdg_sw2d_step_rk2_exchanged(bdg_sw2d* s, double dt, int num_steps, int filter);
int bdg_sw2d_step_ssprk2_exchanged(bdg_sw2d* s, double dt, int num_steps, int filter, double sponge_coeff);
/* In-process alternative to RCCL: all parts of the split are handles of THIS process (one per GPU
 * of the node, or several on one GPU). bdg_sw2d_local_peers takes the same neighbour tables as
 * bdg_sw2d_comm_init; bdg_sw2d_group_lserk4_stages then advances parts[0..n) (parts[r] = rank r)
 * together with the same overlapped schedule, the exchange being device-to-device copies out of the
 * neighbours' send buffers. */
int bdg_sw2d_local_peers(bdg_sw2d* s, int rank, const int* peer_ranks, const int* send_start,
                         const int* send_count, const int* recv_start, const int* recv_count, int num_peers);
int bdg_sw2d_group_lserk4_stages(bdg_sw2d** parts, int num_parts, double dt, int num_stages);
/* bdg_sw2d_compute_dt with the maxima reduced over all ranks (one 8-byte all-reduce each). */
int bdg_sw2d_compute_dt_global(bdg_sw2d* s, double cfl, double* dt, double* eta_max);
int bdg_sw2d_allreduce_max(bdg_sw2d* s, double value, double* out);
int bdg_sw2d_allreduce_sum(bdg_sw2d* s, double value, double* out); /* e.g. total mass over all ranks */
/* Drains this rank's streams, meets every other rank, drains again. */
int bdg_sw2d_barrier(bdg_sw2d* s);
/* RHS of the resident state (owned elements valid; ghosts must be current) to host arrays. */
int bdg_sw2d_rhs_resident(bdg_sw2d* s, double* rhs1, double* rhs2, double* rhs3);

int bdg_sw2d_synchronize(bdg_sw2d* s);
/* Runs num_stages LSERK4 stages bracketed by HIP events on the solver's own stream and
 * returns the average device time per stage-kernel launch in milliseconds. */
int bdg_sw2d_time_lserk4_stages(bdg_sw2d* s, double dt, int num_stages, float* ms_per_launch);
/* Bytes of HBM the solver holds; algorithmic bytes per element per fused stage. */
/* Measurement aids. bdg_probe_stream_triad: STREAM triad (a = b + s*c, 16 B per lane) over three
 * arrays of bytes_per_array each; returns GB/s -- the practical HBM roof of the device.
 * bdg_sw2d_probe_stage_traffic: a launch with the stage kernel's own row accesses and read/write
 * mix but no gathers and no arithmetic; its time is the memory-system floor for that pattern. */
int bdg_probe_stream_triad(int device, size_t bytes_per_array, int repeats, double* gbps);
int bdg_sw2d_probe_stage_traffic(bdg_sw2d* s, int repeats, float* ms_per_launch);
/* 1 if the solver runs the affine-geometry kernels, 0 for the per-node-geometry kernels. */
int bdg_sw2d_uses_affine_geometry(const bdg_sw2d* s);
/* 1 if the solver renumbered the elements internally (I/O stays in the caller's numbering). */
int bdg_sw2d_is_renumbered(const bdg_sw2d* s);
size_t bdg_sw2d_device_bytes(const bdg_sw2d* s);
/* The raw stream (hipStream_t) launches are issued on, for callers that interleave their own work. */
void* bdg_sw2d_stream(bdg_sw2d* s);

/* ---------------------------------------------------------------- sw2d, curved / over-integrated RHS
 * reference: swhelpers.rhs.sw2dComputeRHS_curved(h, hu, hv, hN, zx, zy, g, H, f, CD, ctx, cub_ctx, gauss_ctx,
 * curvedEls, J, gmapM, gmapP) (swhelpers/rhs.py:6-176), driven by sw2d_curved.py:246-277 (RHS, Filter, midpoint
 * RK2). The function is pure in its context arguments, and so is this solver: every table is an input (any
 * cubature rule, any Gauss-node map -- periodic rewiring included), borrowed for the duration of create.
 * Four fields always (h, hu, hv, hN). H is an argument of the reference function that it never uses
 * (rhs.py:16, :73-75 form cub_H / gauss_H and drop them); it is therefore not part of this interface.
 * Host code above this seam: blitzdg_amd/swhelpers/rhs.py (same 17-argument signature). */
typedef struct bdg_sw2d_curved bdg_sw2d_curved;
typedef struct bdg_sw2d_curved_desc {
    int order;            /* N: 1..8                                                          */
    int num_elements;     /* K                                                                */
    int num_cub;          /* cubature points per element (cub_ctx.V.shape[0])                 */
    int num_gauss;        /* Gauss points per face (gauss_ctx.NGauss)                         */
    const double* V;      /* (Np, Np) ctx.V: the straight elements' mass inverse is V V^T     */
    const double* Filter; /* (Np, Np) ctx.filter or NULL                                      */
    const double* J;      /* (Np, K) nodal Jacobian as the driver forms it (sw2d_curved.py:112-118) */
    const double* cubV; const double* cubDr; const double* cubDs;     /* (Ncub, Np) cub_ctx.V, Dr, Ds   */
    const double* cubW; const double* cubrx; const double* cubry;     /* (Ncub, K)  cub_ctx.W, rx, ry   */
    const double* cubsx; const double* cubsy;                         /* (Ncub, K)  cub_ctx.sx, sy      */
    const double* gaussInterp;                                        /* (3 NGauss, Np) gauss_ctx.Interp */
    const double* gaussW; const double* gaussnx; const double* gaussny; /* (3 NGauss, K) gauss_ctx.W, nx, ny */
    const int* gmapM;     /* (3 NGauss K) flat Gauss ids g + 3 NGauss k                       */
    const int* gmapP;
    const int* gmapW;     /* gauss_ctx.BCmap[3]: flat ids of reflective-wall Gauss nodes      */
    int num_wall;
    const int* curvedEls; /* elements that take their own cubature mass matrix                */
    int num_curved;
    const double* MMChol; /* (Np, Np, K) cub_ctx.MMChol; only the columns of curvedEls are read; may be NULL if none */
    const double* zx;     /* (Np, K) or NULL (= 0)                                            */
    const double* zy;
    const double* coriolis; /* (Np, K) f, or NULL: coriolis_const                             */
    double coriolis_const;
    const double* drag;   /* (Np, K) CD (an array in sw2d_curved.py:176-192), or NULL: drag_const */
    double drag_const;
    double g;
    int device;
    int flags;            /* 0                                                                */
} bdg_sw2d_curved_desc;
/* Size limit of one solver: device tables are addressed with 32-bit byte offsets, so max(5 Np, 16 ceil(num_cub / 16),
 * 192 ceil(num_gauss / 16)) * ld * 8 must stay below 4 GiB (ld = num_elements rounded up to 64): 2.8 million elements at
 * N = 4 with the builders' default rules, 2.5 million at N = 8. BDG_ERR_ARGUMENT beyond it: partition the mesh. */
int bdg_sw2d_curved_create(const bdg_sw2d_curved_desc* desc, bdg_sw2d_curved** out);
void bdg_sw2d_curved_destroy(bdg_sw2d_curved* s);
/* The reference function itself: host (Np, K) fields in, host RHS out; filter != 0 returns Filter * RHS
 * (what the driver applies next, sw2d_curved.py:250-253). The resident state is not disturbed. */
int bdg_sw2d_curved_rhs(bdg_sw2d_curved* s, const double* h, const double* hu, const double* hv, const double* hN,
                        double* rhs1, double* rhs2, double* rhs3, double* rhs4, int filter);
int bdg_sw2d_curved_set_state(bdg_sw2d_curved* s, const double* h, const double* hu, const double* hv, const double* hN);
int bdg_sw2d_curved_get_state(bdg_sw2d_curved* s, double* h, double* hu, double* hv, double* hN);
/* num_steps of the driver's loop body (sw2d_curved.py:246-277) on the resident state: RHS, filter, predictor
 * q1 = q + dt/2 RHS, RHS(q1), filter, corrector q += dt RHS. */
int bdg_sw2d_curved_step_rk2(bdg_sw2d_curved* s, double dt, int num_steps, int filter);
/* num_stages fused LSERK4 stages on the resident state (stage index continues across calls). */
int bdg_sw2d_curved_lserk4_stages(bdg_sw2d_curved* s, double dt, int num_stages);
/* As step_rk2, timed with HIP events on the solver's stream: milliseconds per RHS evaluation
 * (Gauss-trace kernel + stage kernel + curved-element kernel), two evaluations per step. */
int bdg_sw2d_curved_time_rk2(bdg_sw2d_curved* s, double dt, int num_steps, int filter, float* ms_per_rhs);
int bdg_sw2d_curved_synchronize(bdg_sw2d_curved* s);
/* Element-partitioned runs (blitzdg_amd.sw2d_curved.DistributedSw2dCurved; no reference analogue, the reference is
 * single-process): the solver is created on a rank-local mesh [owned elements | ghost elements], and the caller refreshes the
 * ghost columns before every RHS evaluation. Columns [first, first + count) of all 4 Np rows of the resident state
 * (which = 0) or of the RK2 intermediate state (which = 1), as a (4 Np, count) row-major host array: */
int bdg_sw2d_curved_get_elements(bdg_sw2d_curved* s, int which, int first, int count, double* out);
int bdg_sw2d_curved_set_elements(bdg_sw2d_curved* s, int which, int first, int count, const double* in);
/* One half of the driver's step (sw2d_curved.py:246-277), so that a ghost exchange fits between the two evaluations:
 * phase 0 -- predictor, intermediate = state + dt/2 RHS(state); phase 1 -- corrector, state += dt RHS(intermediate). */
int bdg_sw2d_curved_rk2_phase(bdg_sw2d_curved* s, double dt, int phase, int filter);
/* The same exchange driven by this library over RCCL (point-to-point over xGMI; bound with dlopen on first use, as for
 * bdg_sw2d_comm_init, whose id -- bdg_comm_unique_id -- and neighbour tables it takes): the solver's mesh is ordered
 * [elements without a ghost neighbour: num_interior | partition-boundary elements: up to num_owned | ghosts: up to K]; peer i is
 * sent the elements send_elements[send_start[i] .. + send_count[i]) and its recv_count[i] elements arrive in ghost slots
 * recv_start[i] .. (relative to num_owned); one record of 4 Np doubles per element. step_rk2_exchanged = num_steps times
 * {exchange(state), predictor, exchange(intermediate), corrector}; on the nodal-trace form the interior elements of an
 * evaluation run on the solver's stream beside the exchange and the partition-boundary elements on a second stream (two chains
 * joined by events, as bdg_sw2d_lserk4_stages_exchanged), ghost elements are not evaluated, and the ghost columns bdg_sw2d_curved_
 * get_state returns afterwards are whatever was last received. exchange: one refresh by itself (which as above). barrier: drains
 * the streams, meets every rank (an 8-byte all-reduce), drains again. */
int bdg_sw2d_curved_set_partition(bdg_sw2d_curved* s, int num_interior, int num_owned, const int* send_elements, int num_send);
int bdg_sw2d_curved_comm_init(bdg_sw2d_curved* s, int rank, int world, const void* unique_id, const int* peer_ranks,
                              const int* send_start, const int* send_count, const int* recv_start, const int* recv_count,
                              int num_peers);
int bdg_sw2d_curved_step_rk2_exchanged(bdg_sw2d_curved* s, double dt, int num_steps, int filter);
int bdg_sw2d_curved_lserk4_stages_exchanged(bdg_sw2d_curved* s, double dt, int num_stages); /* bdg_sw2d_curved_lserk4_stages, an exchange in front of every stage */
int bdg_sw2d_curved_exchange(bdg_sw2d_curved* s, int which);
int bdg_sw2d_curved_barrier(bdg_sw2d_curved* s);
size_t bdg_sw2d_curved_device_bytes(const bdg_sw2d_curved* s);
/* Compulsory HBM bytes of one RHS evaluation with the tables as this solver holds them (per element, averaged). */
double bdg_sw2d_curved_bytes_per_element(const bdg_sw2d_curved* s);
/* Which kernels serve this solver: 1 = the nodal-trace form (one stage launch per evaluation; the neighbours' Gauss traces
 * are products of their face-node values: needs gmapM = identity, a gmapP that pairs whole faces and an Interp whose face
 * rows vanish off the face nodes -- what buildGaussFaceNodes produces, periodic rewiring included), 0 = the general form
 * (Gauss-trace planes; any map). Decided at creation; BDG_SW2D_CURVED_GENERAL=1 forces 0. -1: NULL handle. */
int bdg_sw2d_curved_form(const bdg_sw2d_curved* s);

#ifdef __cplusplus
}
#endif
#endif /* BLITZDG_HIP_H */
