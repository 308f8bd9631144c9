// C ABI, host-setup group (see include/blitzdg_hip.h). Thin handles around the
// C++ classes; tables are exported as borrowed views so the Python front end
// (blitzdg_amd/pyblitzdg.py) can mirror the reference's boost::python module
// (src/pyblitzdg/pyblitzdg.cpp:59-201) without copying on this side.
#include "capi_internal.hpp"
#include "parallel_for.hpp"
#include "blitzdg/Advec1d.hpp"
#include "blitzdg/Burgers1d.hpp"
#include "blitzdg/LSERK4.hpp"
#include "blitzdg/TriangleCubatureRules.hpp"
#include "blitzdg/VtkOutputter.hpp"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

using namespace blitzdg;

namespace bdg_detail {
thread_local std::string g_last_error;
void set_error(const std::string& msg) { g_last_error = msg; }
} // namespace bdg_detail

using bdg_detail::guard;
using bdg_detail::set_error;

namespace {
void view(bdg_table* out, const real_matrix_type& m) { *out = {m.data(), m.rows(), m.cols(), BDG_F64}; }
void view(bdg_table* out, const index_matrix_type& m) { *out = {m.data(), m.rows(), m.cols(), BDG_I32}; }
void view(bdg_table* out, const real_vector_type& v) { *out = {v.data(), v.size(), 1, BDG_F64}; }
void view(bdg_table* out, const index_vector_type& v) { *out = {v.data(), v.size(), 1, BDG_I32}; }
void view(bdg_table* out, const std::vector<index_type>& v) {
    *out = {v.data(), static_cast<int>(v.size()), 1, BDG_I32};
}
void view2(bdg_table* out, const index_vector_type& v, int cols) {
    *out = {v.data(), cols ? v.size() / cols : 0, cols, BDG_I32};
}
} // namespace

extern "C" {

const char* bdg_last_error(void) { return bdg_detail::g_last_error.c_str(); }
int bdg_version(void) { return 100; }

// ------------------------------------------------------------------ mesh

int bdg_mesh_create(bdg_mesh** out) {
    return guard([&] {
        if (!out) throw bdg_detail::arg_error("bdg_mesh_create: out is NULL");
        *out = new bdg_mesh();
    });
}

void bdg_mesh_destroy(bdg_mesh* mesh) { delete mesh; }

int bdg_mesh_read(bdg_mesh* mesh, const char* path) {
    return guard([&] {
        if (!mesh || !path) throw bdg_detail::arg_error("bdg_mesh_read: NULL argument");
        mesh->mgr.readMesh(path);
    });
}

int bdg_mesh_write(const bdg_mesh* mesh, const char* gmsh_path) {
    return guard([&] {
        if (!mesh || !gmsh_path) throw bdg_detail::arg_error("bdg_mesh_write: NULL argument");
        mesh->mgr.writeMesh(gmsh_path);
    });
}

int bdg_mesh_write_cache(const bdg_mesh* mesh, const char* cache_path) {
    return guard([&] {
        if (!mesh || !cache_path) throw bdg_detail::arg_error("bdg_mesh_write_cache: NULL argument");
        mesh->mgr.writeCache(cache_path);
    });
}

int bdg_mesh_read_cache(bdg_mesh* mesh, const char* cache_path) {
    return guard([&] {
        if (!mesh || !cache_path) throw bdg_detail::arg_error("bdg_mesh_read_cache: NULL argument");
        mesh->mgr.readCache(cache_path);
    });
}

int bdg_mesh_build(bdg_mesh* mesh, const int* etov, int K, const double* vert, int Nv, int dim) {
    return guard([&] {
        if (!mesh || !etov || !vert || K < 1 || Nv < 3) throw bdg_detail::arg_error("bdg_mesh_build: bad argument");
        mesh->mgr.buildMesh(etov, K, vert, Nv, dim);
    });
}

int bdg_mesh_build_box(bdg_mesh* mesh, int nx, int ny, double x0, double x1, double y0, double y1,
                       unsigned long long seed) {
    return guard([&] {
        if (!mesh) throw bdg_detail::arg_error("bdg_mesh_build_box: mesh is NULL");
        mesh->mgr.buildBoxMesh(nx, ny, x0, x1, y0, y1, seed);
    });
}

int bdg_mesh_set_bctype(bdg_mesh* mesh, const int* bctype, int n) {
    return guard([&] {
        if (!mesh || !bctype) throw bdg_detail::arg_error("bdg_mesh_set_bctype: NULL argument");
        mesh->mgr.set_BCType(bctype, n);
    });
}

int bdg_mesh_partition(bdg_mesh* mesh, int nparts) {
    return guard([&] {
        if (!mesh) throw bdg_detail::arg_error("bdg_mesh_partition: mesh is NULL");
        mesh->mgr.partitionMesh(nparts);
    });
}

int bdg_mesh_num_elements(const bdg_mesh* mesh) { return mesh ? mesh->mgr.get_NumElements() : -1; }
int bdg_mesh_num_verts(const bdg_mesh* mesh) { return mesh ? mesh->mgr.get_NumVerts() : -1; }

int bdg_mesh_table(const bdg_mesh* mesh, int which, bdg_table* out) {
    return guard([&] {
        if (!mesh || !out) throw bdg_detail::arg_error("bdg_mesh_table: NULL argument");
        const MeshManager& m = mesh->mgr;
        switch (which) {
        case BDG_MESH_VERTICES: *out = {m.get_Vertices().data(), m.get_NumVerts(), m.get_Dim(), BDG_F64}; break;
        case BDG_MESH_ELEMENTS: view2(out, m.get_Elements(), 3); break;
        case BDG_MESH_ETOE: view2(out, m.get_EToE(), 3); break;
        case BDG_MESH_ETOF: view2(out, m.get_EToF(), 3); break;
        case BDG_MESH_BCTYPE: view2(out, m.get_BCType(), 3); break;
        case BDG_MESH_EPART: view(out, m.get_ElementPartitionMap()); break;
        case BDG_MESH_NPART: view(out, m.get_VertexPartitionMap()); break;
        default: throw bdg_detail::arg_error("bdg_mesh_table: unknown table id");
        }
    });
}

// ------------------------------------------------------------------ triangle nodes

int bdg_trinodes_create(int order, const bdg_mesh* mesh, bdg_trinodes** out) {
    return guard([&] {
        if (!mesh || !out) throw bdg_detail::arg_error("bdg_trinodes_create: NULL argument");
        if (mesh->mgr.get_NumElements() < 1) throw bdg_detail::arg_error("bdg_trinodes_create: mesh is empty");
        *out = new bdg_trinodes{TriangleNodesProvisioner(order, mesh->mgr)};
    });
}

void bdg_trinodes_destroy(bdg_trinodes* nodes) { delete nodes; }

int bdg_trinodes_build_filter(bdg_trinodes* nodes, double Nc, int s) {
    return guard([&] {
        if (!nodes) throw bdg_detail::arg_error("bdg_trinodes_build_filter: nodes is NULL");
        nodes->prov.buildFilter(Nc, s);
        nodes->hasFilter = true;
    });
}

int bdg_trinodes_build_bchash(bdg_trinodes* nodes, const int* bctype, int n) {
    return guard([&] {
        if (!nodes || !bctype) throw bdg_detail::arg_error("bdg_trinodes_build_bchash: NULL argument");
        index_vector_type bc(n);
        std::copy(bctype, bctype + n, bc.begin());
        nodes->prov.buildBCHash(bc);
    });
}

// Hx, Hy of the variant-B driver (reference src/sw2d/main.cpp:128-133): physical gradient of H, then the
// dealiasing filter applied to each component.
int bdg_trinodes_bed_slopes(const bdg_trinodes* nodes, const double* H, double* Hx, double* Hy) {
    return guard([&] {
        if (!nodes || !H || !Hx || !Hy) throw bdg_detail::arg_error("bdg_trinodes_bed_slopes: NULL argument");
        const auto& p = nodes->prov;
        const int Np = p.get_NumLocalPoints(), K = p.get_NumElements();
        const real_matrix_type& Dr = p.get_Dr(), &Ds = p.get_Ds(), &F = p.get_Filter();
        if (F.rows() != Np) throw bdg_detail::arg_error("bdg_trinodes_bed_slopes: call buildFilter first");
        const real_matrix_type& rx = p.get_rx(), &sx = p.get_sx(), &ry = p.get_ry(), &sy = p.get_sy();
        blitzdg::detail::parallelChunks(K, [&](int kBegin, int kEnd) {
        std::vector<double> gx(Np), gy(Np);
        for (int k = kBegin; k < kEnd; ++k) {
            for (int i = 0; i < Np; ++i) {
                double dr = 0.0, ds = 0.0;
                for (int m = 0; m < Np; ++m) {
                    dr += Dr(i, m) * H[static_cast<size_t>(m) * K + k];
                    ds += Ds(i, m) * H[static_cast<size_t>(m) * K + k];
                }
                gx[i] = rx(i, k) * dr + sx(i, k) * ds;
                gy[i] = ry(i, k) * dr + sy(i, k) * ds;
            }
            for (int i = 0; i < Np; ++i) {
                double ax = 0.0, ay = 0.0;
                for (int m = 0; m < Np; ++m) {
                    ax += F(i, m) * gx[m];
                    ay += F(i, m) * gy[m];
                }
                Hx[static_cast<size_t>(i) * K + k] = ax;
                Hy[static_cast<size_t>(i) * K + k] = ay;
            }
        }
        });
    });
}

// buildSpongeCoeff (reference src/sw2d/main.cpp:516-556): strength*(1 - d/radius) where d < radius is
// the distance to the closest open-boundary node, 0 elsewhere.
int bdg_trinodes_sponge_coeff(const bdg_trinodes* nodes, const int* mapO, int num_out, double strength, double radius,
                              double* coeff) {
    return guard([&] {
        if (!nodes || !coeff || num_out < 0 || (num_out > 0 && !mapO))
            throw bdg_detail::arg_error("bdg_trinodes_sponge_coeff: bad argument");
        const auto& p = nodes->prov;
        const int Np = p.get_NumLocalPoints(), K = p.get_NumElements(), NFN = 3 * p.get_NumFacePoints();
        const real_matrix_type& x = p.get_xGrid(), &y = p.get_yGrid();
        const index_vector_type& vmapM = p.get_vmapM();
        std::vector<double> xo(num_out), yo(num_out);
        for (int i = 0; i < num_out; ++i) {
            if (mapO[i] < 0 || mapO[i] >= NFN * K) throw bdg_detail::arg_error("bdg_trinodes_sponge_coeff: node index out of range");
            const int v = vmapM(mapO[i]);
            xo[i] = x(v % Np, v / Np);
            yo[i] = y(v % Np, v / Np);
        }
        blitzdg::detail::parallelFor(K, [&](int k) {
            for (int n = 0; n < Np; ++n) {
                double closest = 1.0e12;
                for (int i = 0; i < num_out; ++i) {
                    const double dist = std::hypot(x(n, k) - xo[i], y(n, k) - yo[i]);
                    if (dist < radius && dist < closest) closest = dist;
                }
                coeff[static_cast<size_t>(n) * K + k] = closest < 1.0e12 ? strength * (1.0 - closest / radius) : 0.0;
            }
        }, 16);
    });
}

// ---- output step (reference splitElements + VtkOutputter)
int bdg_trinodes_split_count(const bdg_trinodes* nodes) {
    if (!nodes) return -1;
    const int N = nodes->prov.get_NOrder();
    return N * N;
}

int bdg_trinodes_split_operators(const bdg_trinodes* nodes, double* IM, int* local_triangles) {
    return guard([&] {
        if (!nodes || !IM || !local_triangles) throw bdg_detail::arg_error("bdg_trinodes_split_operators: NULL argument");
        real_matrix_type im;
        std::vector<index_type> tri;
        nodes->prov.splitOperators(im, tri);
        std::copy(im.data(), im.data() + static_cast<size_t>(im.rows()) * im.cols(), IM);
        std::copy(tri.begin(), tri.end(), local_triangles);
    });
}

int bdg_trinodes_split_elements(const bdg_trinodes* nodes, const double* field, double* xnew, double* ynew,
                                double* fieldnew) {
    return guard([&] {
        if (!nodes || !field || !xnew || !ynew || !fieldnew) throw bdg_detail::arg_error("bdg_trinodes_split_elements: NULL argument");
        const auto& p = nodes->prov;
        const int Np = p.get_NumLocalPoints(), K = p.get_NumElements();
        real_matrix_type f(Np, K), xn, yn, fn;
        std::copy(field, field + static_cast<size_t>(Np) * K, f.data());
        p.splitElements(p.get_xGrid(), p.get_yGrid(), f, xn, yn, fn);
        const size_t n = static_cast<size_t>(3) * xn.cols();
        std::copy(xn.data(), xn.data() + n, xnew);
        std::copy(yn.data(), yn.data() + n, ynew);
        std::copy(fn.data(), fn.data() + n, fieldnew);
    });
}

int bdg_trinodes_write_vtu(const bdg_trinodes* nodes, const char* path, const double* field, const char* field_name) {
    return guard([&] {
        if (!nodes || !path || !field || !field_name) throw bdg_detail::arg_error("bdg_trinodes_write_vtu: NULL argument");
        const auto& p = nodes->prov;
        const int Np = p.get_NumLocalPoints(), K = p.get_NumElements();
        real_matrix_type f(Np, K);
        std::copy(field, field + static_cast<size_t>(Np) * K, f.data());
        blitzdg::VtkOutputter(p).writeFieldToFile(path, f, field_name);
    });
}

int bdg_write_vtu_triangles(const char* path, const double* x, const double* y, const double* field, int num_triangles,
                            const char* field_name) {
    return guard([&] {
        if (!path || !x || !y || !field || !field_name || num_triangles < 0)
            throw bdg_detail::arg_error("bdg_write_vtu_triangles: bad argument");
        real_matrix_type xm(3, num_triangles), ym(3, num_triangles), fm(3, num_triangles);
        const size_t n = static_cast<size_t>(3) * num_triangles;
        std::copy(x, x + n, xm.data());
        std::copy(y, y + n, ym.data());
        std::copy(field, field + n, fm.data());
        blitzdg::VtkOutputter::writeTriangles(path, xm, ym, fm, field_name);
    });
}

int bdg_trinodes_set_coordinates(bdg_trinodes* nodes, const double* x, const double* y) {
    return guard([&] {
        if (!nodes || !x || !y) throw bdg_detail::arg_error("bdg_trinodes_set_coordinates: NULL argument");
        nodes->prov.setCoordinates(x, y);
    });
}

int bdg_trinodes_dims(const bdg_trinodes* nodes, int* order, int* np, int* nfp, int* K) {
    return guard([&] {
        if (!nodes) throw bdg_detail::arg_error("bdg_trinodes_dims: nodes is NULL");
        if (order) *order = nodes->prov.get_NOrder();
        if (np) *np = nodes->prov.get_NumLocalPoints();
        if (nfp) *nfp = nodes->prov.get_NumFacePoints();
        if (K) *K = nodes->prov.get_NumElements();
    });
}

int bdg_trinodes_table(const bdg_trinodes* nodes, int which, bdg_table* out) {
    return guard([&] {
        if (!nodes || !out) throw bdg_detail::arg_error("bdg_trinodes_table: NULL argument");
        const TriangleNodesProvisioner& p = nodes->prov;
        switch (which) {
        case BDG_TRI_R: view(out, p.get_rGrid()); break;
        case BDG_TRI_S: view(out, p.get_sGrid()); break;
        case BDG_TRI_X: view(out, p.get_xGrid()); break;
        case BDG_TRI_Y: view(out, p.get_yGrid()); break;
        case BDG_TRI_V: view(out, p.get_V()); break;
        case BDG_TRI_VINV: view(out, p.get_Vinv()); break;
        case BDG_TRI_DR: view(out, p.get_Dr()); break;
        case BDG_TRI_DS: view(out, p.get_Ds()); break;
        case BDG_TRI_DRW: view(out, p.get_Drw()); break;
        case BDG_TRI_DSW: view(out, p.get_Dsw()); break;
        case BDG_TRI_LIFT: view(out, p.get_Lift()); break;
        case BDG_TRI_FILTER: view(out, p.get_Filter()); break;
        case BDG_TRI_J: view(out, p.get_J()); break;
        case BDG_TRI_RX: view(out, p.get_rx()); break;
        case BDG_TRI_RY: view(out, p.get_ry()); break;
        case BDG_TRI_SX: view(out, p.get_sx()); break;
        case BDG_TRI_SY: view(out, p.get_sy()); break;
        case BDG_TRI_NX: view(out, p.get_nx()); break;
        case BDG_TRI_NY: view(out, p.get_ny()); break;
        case BDG_TRI_FSCALE: view(out, p.get_Fscale()); break;
        case BDG_TRI_FMASK: view(out, p.get_Fmask()); break;
        case BDG_TRI_FX: view(out, p.get_Fx()); break;
        case BDG_TRI_FY: view(out, p.get_Fy()); break;
        case BDG_TRI_VMAPM: view(out, p.get_vmapM()); break;
        case BDG_TRI_VMAPP: view(out, p.get_vmapP()); break;
        case BDG_TRI_MAPP: view(out, p.get_mapP()); break;
        case BDG_TRI_VMAPB: view(out, p.get_vmapB()); break;
        case BDG_TRI_MAPB: view(out, p.get_mapB()); break;
        case BDG_TRI_GATHER: view(out, p.get_gather()); break;
        case BDG_TRI_SCATTER: view(out, p.get_scatter()); break;
        default: throw bdg_detail::arg_error("bdg_trinodes_table: unknown table id");
        }
    });
}

int bdg_trinodes_bcmap_num_tags(const bdg_trinodes* nodes) {
    return nodes ? static_cast<int>(nodes->prov.get_bcMap().size()) : -1;
}

int bdg_trinodes_bcmap_tags(const bdg_trinodes* nodes, int* tags, int capacity) {
    return guard([&] {
        if (!nodes || !tags) throw bdg_detail::arg_error("bdg_trinodes_bcmap_tags: NULL argument");
        std::vector<int> keys;
        for (const auto& kv : nodes->prov.get_bcMap()) keys.push_back(kv.first);
        std::sort(keys.begin(), keys.end());
        if (static_cast<int>(keys.size()) > capacity) throw bdg_detail::arg_error("bdg_trinodes_bcmap_tags: capacity too small");
        std::copy(keys.begin(), keys.end(), tags);
    });
}

int bdg_trinodes_bcmap_nodes(const bdg_trinodes* nodes, int tag, const int** out, int* count) {
    return guard([&] {
        if (!nodes || !out || !count) throw bdg_detail::arg_error("bdg_trinodes_bcmap_nodes: NULL argument");
        const auto& map = nodes->prov.get_bcMap();
        const auto it = map.find(tag);
        if (it == map.end()) { *out = nullptr; *count = 0; return; }
        *out = it->second.data();
        *count = static_cast<int>(it->second.size());
    });
}

// ------------------------------------------------------------------ Gauss face / cubature contexts

int bdg_trinodes_build_gauss_face_nodes(bdg_trinodes* nodes, int NGauss, bdg_gaussctx** out) {
    return guard([&] {
        if (!nodes || !out) throw bdg_detail::arg_error("bdg_trinodes_build_gauss_face_nodes: NULL argument");
        if (NGauss < 1) throw bdg_detail::arg_error("bdg_trinodes_build_gauss_face_nodes: NGauss must be >= 1");
        *out = new bdg_gaussctx{nodes->prov.buildGaussFaceNodes(NGauss)};
    });
}

void bdg_gaussctx_destroy(bdg_gaussctx* ctx) { delete ctx; }
int bdg_gaussctx_ngauss(const bdg_gaussctx* ctx) { return ctx ? ctx->ctx.NGauss() : -1; }

int bdg_gaussctx_table(const bdg_gaussctx* ctx, int which, bdg_table* out) {
    return guard([&] {
        if (!ctx || !out) throw bdg_detail::arg_error("bdg_gaussctx_table: NULL argument");
        const GaussFaceContext2D& g = ctx->ctx;
        switch (which) {
        case BDG_GAUSS_NX: view(out, g.nx()); break;
        case BDG_GAUSS_NY: view(out, g.ny()); break;
        case BDG_GAUSS_SJ: view(out, g.sJ()); break;
        case BDG_GAUSS_J: view(out, g.Jac()); break;
        case BDG_GAUSS_RX: view(out, g.rx()); break;
        case BDG_GAUSS_RY: view(out, g.ry()); break;
        case BDG_GAUSS_SX: view(out, g.sx()); break;
        case BDG_GAUSS_SY: view(out, g.sy()); break;
        case BDG_GAUSS_X: view(out, g.x()); break;
        case BDG_GAUSS_Y: view(out, g.y()); break;
        case BDG_GAUSS_W: view(out, g.W()); break;
        case BDG_GAUSS_INTERP: view(out, g.Interp()); break;
        case BDG_GAUSS_MAPM: view(out, g.mapM()); break;
        case BDG_GAUSS_MAPP: view(out, g.mapP()); break;
        default: throw bdg_detail::arg_error("bdg_gaussctx_table: unknown table id");
        }
    });
}

int bdg_gaussctx_bcmap_num_tags(const bdg_gaussctx* ctx) { return ctx ? static_cast<int>(ctx->ctx.bcMap().size()) : -1; }

int bdg_gaussctx_bcmap_tags(const bdg_gaussctx* ctx, int* tags, int capacity) {
    return guard([&] {
        if (!ctx || (!tags && capacity > 0)) throw bdg_detail::arg_error("bdg_gaussctx_bcmap_tags: NULL argument");
        std::vector<int> keys;
        for (const auto& kv : ctx->ctx.bcMap()) keys.push_back(kv.first);
        std::sort(keys.begin(), keys.end());
        if (static_cast<int>(keys.size()) > capacity) throw bdg_detail::arg_error("bdg_gaussctx_bcmap_tags: capacity too small");
        std::copy(keys.begin(), keys.end(), tags);
    });
}

int bdg_gaussctx_bcmap_nodes(const bdg_gaussctx* ctx, int tag, const int** out, int* count) {
    return guard([&] {
        if (!ctx || !out || !count) throw bdg_detail::arg_error("bdg_gaussctx_bcmap_nodes: NULL argument");
        const auto it = ctx->ctx.bcMap().find(tag);
        if (it == ctx->ctx.bcMap().end()) { *out = nullptr; *count = 0; return; }
        *out = it->second.data();
        *count = static_cast<int>(it->second.size());
    });
}

int bdg_trinodes_build_cubature_volume_mesh(bdg_trinodes* nodes, int NCubature, bdg_cubctx** out) {
    return guard([&] {
        if (!nodes || !out) throw bdg_detail::arg_error("bdg_trinodes_build_cubature_volume_mesh: NULL argument");
        if (NCubature < 1) throw bdg_detail::arg_error("bdg_trinodes_build_cubature_volume_mesh: NCubature must be >= 1");
        *out = new bdg_cubctx{nodes->prov.buildCubatureVolumeMesh(NCubature)};
    });
}

int bdg_cubature_rule_num_points(int NCubature) {
    int n = -1;
    guard([&] { n = TriangleCubatureRules(NCubature).NumCubaturePoints(); });
    return n;
}

int bdg_cubature_rule(int NCubature, double* r, double* s, double* w) {
    return guard([&] {
        if (!r || !s || !w) throw bdg_detail::arg_error("bdg_cubature_rule: NULL argument");
        const TriangleCubatureRules rule(NCubature);
        const real_vector_type rr = rule.rCoord(), ss = rule.sCoord(), ww = rule.weights();
        for (index_type i = 0; i < rule.NumCubaturePoints(); ++i) { r[i] = rr(i); s[i] = ss(i); w[i] = ww(i); }
    });
}

void bdg_cubctx_destroy(bdg_cubctx* ctx) { delete ctx; }
int bdg_cubctx_num_points(const bdg_cubctx* ctx) { return ctx ? ctx->ctx.NumCubaturePoints() : -1; }
int bdg_cubctx_order(const bdg_cubctx* ctx) { return ctx ? ctx->ctx.NCubature() : -1; }

int bdg_cubctx_table(const bdg_cubctx* ctx, int which, bdg_table* out) {
    return guard([&] {
        if (!ctx || !out) throw bdg_detail::arg_error("bdg_cubctx_table: NULL argument");
        const CubatureContext2D& c = ctx->ctx;
        auto view3 = [&](const real_tensor3_type& t) {
            *out = {t.data(), t.length(0) * t.length(1), t.length(2), BDG_F64};
        };
        switch (which) {
        case BDG_CUB_R: view(out, c.r()); break;
        case BDG_CUB_S: view(out, c.s()); break;
        case BDG_CUB_WEIGHTS: view(out, c.w()); break;
        case BDG_CUB_V: view(out, c.V()); break;
        case BDG_CUB_RX: view(out, c.rx()); break;
        case BDG_CUB_RY: view(out, c.ry()); break;
        case BDG_CUB_SX: view(out, c.sx()); break;
        case BDG_CUB_SY: view(out, c.sy()); break;
        case BDG_CUB_J: view(out, c.Jac()); break;
        case BDG_CUB_DR: view(out, c.Dr()); break;
        case BDG_CUB_DS: view(out, c.Ds()); break;
        case BDG_CUB_MM: view3(c.MM()); break;
        case BDG_CUB_MMCHOL: view3(c.MMChol()); break;
        case BDG_CUB_X: view(out, c.x()); break;
        case BDG_CUB_Y: view(out, c.y()); break;
        case BDG_CUB_W: view(out, c.W()); break;
        default: throw bdg_detail::arg_error("bdg_cubctx_table: unknown table id");
        }
    });
}

// ------------------------------------------------------------------ 1-D nodes

int bdg_nodes1d_create(int order, int K, double xmin, double xmax, bdg_nodes1d** out) {
    return guard([&] {
        if (!out || order < 1 || K < 1 || !(xmax > xmin)) throw bdg_detail::arg_error("bdg_nodes1d_create: bad argument");
        *out = new bdg_nodes1d{Nodes1DProvisioner(order, K, xmin, xmax)};
    });
}

void bdg_nodes1d_destroy(bdg_nodes1d* nodes) { delete nodes; }

int bdg_nodes1d_build_nodes(bdg_nodes1d* nodes) {
    return guard([&] {
        if (!nodes) throw bdg_detail::arg_error("bdg_nodes1d_build_nodes: nodes is NULL");
        nodes->prov.buildNodes();
    });
}

int bdg_nodes1d_compute_jacobian(bdg_nodes1d* nodes) {
    return guard([&] {
        if (!nodes) throw bdg_detail::arg_error("bdg_nodes1d_compute_jacobian: nodes is NULL");
        nodes->prov.computeJacobian();
    });
}

int bdg_nodes1d_map_i(const bdg_nodes1d* nodes) { return nodes ? nodes->prov.get_mapI() : -1; }
int bdg_nodes1d_map_o(const bdg_nodes1d* nodes) { return nodes ? nodes->prov.get_mapO() : -1; }

int bdg_nodes1d_table(const bdg_nodes1d* nodes, int which, bdg_table* out) {
    return guard([&] {
        if (!nodes || !out) throw bdg_detail::arg_error("bdg_nodes1d_table: NULL argument");
        const Nodes1DProvisioner& p = nodes->prov;
        switch (which) {
        case BDG_N1D_R: view(out, p.get_rGrid()); break;
        case BDG_N1D_X: view(out, p.get_xGrid()); break;
        case BDG_N1D_V: view(out, p.get_V()); break;
        case BDG_N1D_VINV: view(out, p.get_Vinv()); break;
        case BDG_N1D_DR: view(out, p.get_Dr()); break;
        case BDG_N1D_LIFT: view(out, p.get_Lift()); break;
        case BDG_N1D_J: view(out, p.get_J()); break;
        case BDG_N1D_RX: view(out, p.get_rx()); break;
        case BDG_N1D_NX: view(out, p.get_nx()); break;
        case BDG_N1D_FMASK: view(out, p.get_Fmask()); break;
        case BDG_N1D_FX: view(out, p.get_Fx()); break;
        case BDG_N1D_FSCALE: view(out, p.get_Fscale()); break;
        case BDG_N1D_ETOV: view(out, p.get_EToV()); break;
        case BDG_N1D_ETOE: view(out, p.get_EToE()); break;
        case BDG_N1D_ETOF: view(out, p.get_EToF()); break;
        case BDG_N1D_VMAPM: view(out, p.get_vmapM()); break;
        case BDG_N1D_VMAPP: view(out, p.get_vmapP()); break;
        default: throw bdg_detail::arg_error("bdg_nodes1d_table: unknown table id");
        }
    });
}

int bdg_lserk4_num_stages(void) { return LSERK4::numStages; }
const double* bdg_lserk4_a(void) { return LSERK4::rk4a; }
const double* bdg_lserk4_b(void) { return LSERK4::rk4b; }

int bdg_vandermonde1d(const double* r, int num_points, int num_cols, double* V, double* Vinv) {
    return guard([&] {
        if (!r || !V || num_points < 1 || num_cols < 1) throw bdg_detail::arg_error("bdg_vandermonde1d: bad argument");
        if (Vinv && num_points != num_cols)
            throw bdg_detail::arg_error("bdg_vandermonde1d: the inverse needs a square matrix (num_points == num_cols)");
        real_vector_type rv(num_points);
        std::copy(r, r + num_points, rv.data());
        real_matrix_type Vm(num_points, num_cols), Vi(num_cols, num_cols);
        blitzdg::VandermondeBuilders().computeVandermondeMatrix(rv, Vm, Vi, Vinv != nullptr);
        std::copy(Vm.data(), Vm.data() + static_cast<size_t>(num_points) * num_cols, V);
        if (Vinv) std::copy(Vi.data(), Vi.data() + static_cast<size_t>(num_cols) * num_cols, Vinv);
    });
}

int bdg_nodes1d_advec_rhs(bdg_nodes1d* nodes, const double* u, double c, double* rhs) {
    return guard([&] {
        if (!nodes || !u || !rhs) throw bdg_detail::arg_error("bdg_nodes1d_advec_rhs: NULL argument");
        auto& p = nodes->prov;
        const int Np = p.get_NumLocalPoints(), K = p.get_NumElements();
        real_matrix_type um(Np, K), out(Np, K);
        std::copy(u, u + static_cast<size_t>(Np) * K, um.data());
        blitzdg::advec1d::computeRHS(um, c, p, out);
        std::copy(out.data(), out.data() + static_cast<size_t>(Np) * K, rhs);
    });
}

int bdg_advec1d_run(int order, int K, double xmin, double xmax, double c, double cfl, double final_time,
                    double* max_error, int* num_steps) {
    return guard([&] {
        if (!max_error || order < 1 || K < 1 || c == 0.0) throw bdg_detail::arg_error("bdg_advec1d_run: bad argument");
        index_type steps = 0;
        *max_error = advec1d::run(order, K, xmin, xmax, c, cfl, final_time, &steps);
        if (num_steps) *num_steps = steps;
    });
}

int bdg_nodes1d_burgers_rhs(bdg_nodes1d* nodes, const double* u, double t, double c, double alpha, double nu, double* rhs) {
    return guard([&] {
        if (!nodes || !u || !rhs) throw bdg_detail::arg_error("bdg_nodes1d_burgers_rhs: NULL argument");
        if (alpha == 0.0 || nu <= 0.0) throw bdg_detail::arg_error("bdg_nodes1d_burgers_rhs: need alpha != 0 and nu > 0");
        auto& p = nodes->prov;
        const int Np = p.get_NumLocalPoints(), K = p.get_NumElements();
        real_matrix_type um(Np, K), out(Np, K);
        std::copy(u, u + static_cast<size_t>(Np) * K, um.data());
        blitzdg::burgers1d::computeRHS(um, p.get_xGrid(), t, c, alpha, nu, p, out);
        std::copy(out.data(), out.data() + static_cast<size_t>(Np) * K, rhs);
    });
}

int bdg_burgers1d_run(int order, int K, double xmin, double xmax, double alpha, double nu, double c, double cfl, double final_time,
                      double* max_error, int* num_steps) {
    return guard([&] {
        if (!max_error || order < 1 || K < 1 || c == 0.0 || alpha == 0.0 || nu <= 0.0)
            throw bdg_detail::arg_error("bdg_burgers1d_run: bad argument");
        index_type steps = 0;
        *max_error = burgers1d::run(order, K, xmin, xmax, alpha, nu, c, cfl, final_time, &steps);
        if (num_steps) *num_steps = steps;
    });
}

} // extern "C"
