/* oracle_sw2d.c -- CPU restatement of the reference's sw2d / advec1d hot path.
 *
 * TEST INFRASTRUCTURE ONLY. This file is the checker the HIP path is compared
 * against (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg). Nothing
 * in blitzdg_amd/ imports, links or calls it; the product path has no CPU
 * fallback.
 *
 * Pinning: the reference's C++ cannot be built here (blitz++, boost, LAPACK,
 * METIS, VTK absent), so this restatement is pinned by
 *   (1) the reference's importable NumPy RHS, swhelpers/rhs.py:178-311, run in
 *       the build container by tests/golden/make_golden.py (fixtures: tests/golden, .npz files)
 *       (variant D with f = CD = zx = zy = 0 equals variant A up to round-off),
 *   (2) analytic checks in tests/ (lake at rest, mass conservation, advec1d
 *       convergence, LSERK4 order conditions).
 *
 * It keeps the reference's algorithmic STRUCTURE, pass by pass, with whole-array
 * temporaries, so that timing it is timing "what blitzdg does":
 *   sw2d RHS (variant A)   src/sw2d-simple/main.cpp:181-356
 *   midpoint RK2 + filter  src/sw2d-simple/main.cpp:132-151
 *   dt / blow-up check     src/sw2d-simple/main.cpp:98-109,153-167
 *   LSERK4 stage update    src/advec1d/main.cpp:92-102, include/LSERK4.hpp:15-29
 *   advec1d RHS            src/advec1d/main.cpp:126-188
 * Layout as in the reference: (rows, K) row-major, K contiguous; vmap values
 * number nodes column-wise (n + Np*k). All sums run in ascending index order
 * (blitz `sum(A(ii,kk)*B(kk,jj),kk)`).
 *
 * threads <= 1: strictly single-threaded (how blitzdg runs). threads > 1: the
 * same passes with OpenMP over the long index.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define PARFOR _Pragma("omp parallel for schedule(static) if (threads > 1) num_threads(threads > 1 ? threads : 1)")

static const double rk4a[5] = {0.0, -567301805773.0 / 1357537059087.0, -2404267990393.0 / 2016746695238.0,
                               -3550918686646.0 / 2091501179385.0, -1275806237668.0 / 842570457699.0};
static const double rk4b[5] = {1432997174477.0 / 9575080441755.0, 5161836677717.0 / 13612068292357.0,
                               1720146321549.0 / 2090206949498.0, 3134564353537.0 / 4481467310338.0,
                               2277821191437.0 / 14882151754819.0};

const double* oracle_lserk4_a(void) { return rk4a; }
const double* oracle_lserk4_b(void) { return rk4b; }

typedef struct {
    int Np, Nfp, K;
    const double *Dr, *Ds, *Lift, *Filt;       /* (Np,Np) (Np,Np) (Np,3Nfp) (Np,Np)|NULL */
    const double *rx, *sx, *ry, *sy;           /* (Np,K) */
    const double *nx, *ny, *Fscale;            /* (3Nfp,K) */
    const int *vmapM, *vmapP, *mapW;           /* 3Nfp*K, 3Nfp*K, nW */
    int nW;
    double g;
    int threads;
} oracle_sw2d_ctx;

/* fullToVector(mat, vec, byRows=false): vec[i + rows*j] = mat(i,j)   (BlitzHelpers.hpp:215-232) */
static void full_to_vector(const double* mat, double* vec, int rows, int K, int threads) {
    PARFOR
    for (int j = 0; j < K; ++j)
        for (int i = 0; i < rows; ++i) vec[i + (size_t)rows * j] = mat[(size_t)i * K + j];
}
static void vector_to_full(const double* vec, double* mat, int rows, int K, int threads) {
    PARFOR
    for (int j = 0; j < K; ++j)
        for (int i = 0; i < rows; ++i) mat[(size_t)i * K + j] = vec[i + (size_t)rows * j];
}
/* C(ii,jj) = sum(A(ii,kk)*B(kk,jj),kk): A (m x n), B (n x K) */
static void contract(const double* A, const double* B, double* C, int m, int n, int K, int threads) {
    PARFOR
    for (int j = 0; j < K; ++j)
        for (int i = 0; i < m; ++i) {
            double s = 0.0;
            for (int k = 0; k < n; ++k) s += A[i * n + k] * B[(size_t)k * K + j];
            C[(size_t)i * K + j] = s;
        }
}

/* blitzdg::sw2d::computeRHS (variant A), src/sw2d-simple/main.cpp:181-356 */
int oracle_sw2d_rhs(const oracle_sw2d_ctx* c, const double* h, const double* hu, const double* hv, double* RHS1,
                    double* RHS2, double* RHS3) {
    const int Np = c->Np, Nfp = c->Nfp, K = c->K, threads = c->threads;
    const int nfl = 3 * Nfp;                       /* face nodes per element */
    const size_t nF = (size_t)nfl * K, nT = (size_t)Np * K;
    const double g = c->g;
    /* temporaries, one per named array of the reference (:217-234, :269-296, :314, :321, :345) */
    enum { NFV = 28 };
    double* buf = (double*)malloc(sizeof(double) * (NFV * nF + 3 * nT + 3 * nT + 12 * nT));
    if (!buf) return 1;
    double* p = buf;
#define TAKE(n) (p += (n), p - (n))
    double *nxVec = TAKE(nF), *nyVec = TAKE(nF), *hM = TAKE(nF), *hP = TAKE(nF), *huM = TAKE(nF), *huP = TAKE(nF),
           *hvM = TAKE(nF), *hvP = TAKE(nF), *dh = TAKE(nF), *dhu = TAKE(nF), *dhv = TAKE(nF), *F2M = TAKE(nF),
           *G2M = TAKE(nF), *G3M = TAKE(nF), *F2P = TAKE(nF), *G2P = TAKE(nF), *G3P = TAKE(nF), *spdMax = TAKE(nF),
           *lam = TAKE(nF), *dFlux1 = TAKE(nF), *dFlux2 = TAKE(nF), *dFlux3 = TAKE(nF), *dF1m = TAKE(nF),
           *dF2m = TAKE(nF), *dF3m = TAKE(nF), *s1 = TAKE(nF), *s2 = TAKE(nF), *s3 = TAKE(nF);
    double *hVec = TAKE(nT), *huVec = TAKE(nT), *hvVec = TAKE(nT);
    double *F2 = TAKE(nT), *G2 = TAKE(nT), *G3 = TAKE(nT);
    double* D[12];
    for (int i = 0; i < 12; ++i) D[i] = TAKE(nT);
#undef TAKE

    /* :239-253 column-wise flatten + trace gathers */
    full_to_vector(c->nx, nxVec, nfl, K, threads);
    full_to_vector(c->ny, nyVec, nfl, K, threads);
    full_to_vector(h, hVec, Np, K, threads);
    full_to_vector(hu, huVec, Np, K, threads);
    full_to_vector(hv, hvVec, Np, K, threads);
    PARFOR
    for (size_t i = 0; i < nF; ++i) {
        hM[i] = hVec[c->vmapM[i]];   hP[i] = hVec[c->vmapP[i]];
        huM[i] = huVec[c->vmapM[i]]; huP[i] = huVec[c->vmapP[i]];
        hvM[i] = hvVec[c->vmapM[i]]; hvP[i] = hvVec[c->vmapP[i]];
    }
    /* :256-260 reflective walls */
    for (int i = 0; i < c->nW; ++i) {
        const int w = c->mapW[i];
        huP[w] = huM[w] - 2 * nxVec[w] * (huM[w] * nxVec[w] + hvM[w] * nyVec[w]);
        hvP[w] = hvM[w] - 2 * nyVec[w] * (huM[w] * nxVec[w] + hvM[w] * nyVec[w]);
    }
    /* :263-303 jumps, trace fluxes, wave speeds */
    PARFOR
    for (size_t i = 0; i < nF; ++i) {
        dh[i] = hM[i] - hP[i];
        dhu[i] = huM[i] - huP[i];
        dhv[i] = hvM[i] - hvP[i];
        F2M[i] = (huM[i] * huM[i]) / hM[i] + 0.5 * g * hM[i] * hM[i];
        G2M[i] = (huM[i] * hvM[i]) / hM[i];
        G3M[i] = (hvM[i] * hvM[i]) / hM[i] + 0.5 * g * hM[i] * hM[i];
        F2P[i] = (huP[i] * huP[i]) / hP[i] + 0.5 * g * hP[i] * hP[i];
        G2P[i] = (huP[i] * hvP[i]) / hP[i];
        G3P[i] = (hvP[i] * hvP[i]) / hP[i] + 0.5 * g * hP[i] * hP[i];
        const double uM = huM[i] / hM[i], vM = hvM[i] / hM[i], uP = huP[i] / hP[i], vP = hvP[i] / hP[i];
        const double spdM = sqrt(uM * uM + vM * vM) + sqrt(g * hM[i]);
        const double spdP = sqrt(uP * uP + vP * vP) + sqrt(g * hP[i]);
        spdMax[i] = spdM > spdP ? spdM : spdP; /* std::max(spdM, spdP) */
    }
    /* :282-286 volume fluxes (F1 = hu, G1 = hv, F3 aliases G2) */
    PARFOR
    for (size_t i = 0; i < nT; ++i) {
        F2[i] = (hu[i] * hu[i]) / h[i] + 0.5 * g * h[i] * h[i];
        G2[i] = (hu[i] * hv[i]) / h[i];
        G3[i] = (hv[i] * hv[i]) / h[i] + 0.5 * g * h[i] * h[i];
    }
    /* :306-312 per-face maximum of the trace speed, broadcast to the face's nodes */
    PARFOR
    for (int face = 0; face < 3 * K; ++face) {
        double m = spdMax[(size_t)face * Nfp];
        for (int n = 1; n < Nfp; ++n)
            if (spdMax[(size_t)face * Nfp + n] > m) m = spdMax[(size_t)face * Nfp + n];
        for (int n = 0; n < Nfp; ++n) lam[(size_t)face * Nfp + n] = m;
    }
    /* :317-319 strong-form flux jumps */
    PARFOR
    for (size_t i = 0; i < nF; ++i) {
        dFlux1[i] = 0.5 * ((huM[i] - huP[i]) * nxVec[i] + (hvM[i] - hvP[i]) * nyVec[i] - lam[i] * dh[i]);
        dFlux2[i] = 0.5 * ((F2M[i] - F2P[i]) * nxVec[i] + (G2M[i] - G2P[i]) * nyVec[i] - lam[i] * dhu[i]);
        dFlux3[i] = 0.5 * ((G2M[i] - G2P[i]) * nxVec[i] + (G3M[i] - G3P[i]) * nyVec[i] - lam[i] * dhv[i]);
    }
    vector_to_full(dFlux1, dF1m, nfl, K, threads);
    vector_to_full(dFlux2, dF2m, nfl, K, threads);
    vector_to_full(dFlux3, dF3m, nfl, K, threads);

    /* :332-339 twelve contractions (Dr*G2 and Ds*G2 are computed twice, as in the reference) */
    const double* srcs[6] = {hu, hv, F2, G2, G2, G3}; /* F1 G1 F2 G2 F3 G3 */
    for (int q = 0; q < 6; ++q) {
        contract(c->Dr, srcs[q], D[2 * q], Np, Np, K, threads);
        contract(c->Ds, srcs[q], D[2 * q + 1], Np, Np, K, threads);
    }
    PARFOR
    for (size_t i = 0; i < nT; ++i) {
        RHS1[i] = -(c->rx[i] * D[0][i] + c->sx[i] * D[1][i]);
        RHS1[i] += -(c->ry[i] * D[2][i] + c->sy[i] * D[3][i]);
        RHS2[i] = -(c->rx[i] * D[4][i] + c->sx[i] * D[5][i]);
        RHS2[i] += -(c->ry[i] * D[6][i] + c->sy[i] * D[7][i]);
        RHS3[i] = -(c->rx[i] * D[8][i] + c->sx[i] * D[9][i]);
        RHS3[i] += -(c->ry[i] * D[10][i] + c->sy[i] * D[11][i]);
    }
    /* :348-355 Jacobian scaling and lift */
    PARFOR
    for (size_t i = 0; i < nF; ++i) {
        s1[i] = c->Fscale[i] * dF1m[i];
        s2[i] = c->Fscale[i] * dF2m[i];
        s3[i] = c->Fscale[i] * dF3m[i];
    }
    contract(c->Lift, s1, D[0], Np, nfl, K, threads);
    contract(c->Lift, s2, D[1], Np, nfl, K, threads);
    contract(c->Lift, s3, D[2], Np, nfl, K, threads);
    PARFOR
    for (size_t i = 0; i < nT; ++i) {
        RHS1[i] += D[0][i];
        RHS2[i] += D[1][i];
        RHS3[i] += D[2][i];
    }
    free(buf);
    return 0;
}

/* RHS_i = sum(Filt(ii,kk)*RHS_i(kk,jj),kk)   (src/sw2d-simple/main.cpp:134-136) */
static int apply_filter(const oracle_sw2d_ctx* c, double* R) {
    const size_t nT = (size_t)c->Np * c->K;
    double* t = (double*)malloc(sizeof(double) * nT);
    if (!t) return 1;
    contract(c->Filt, R, t, c->Np, c->Np, c->K, c->threads);
    memcpy(R, t, sizeof(double) * nT);
    free(t);
    return 0;
}

int oracle_sw2d_rhs_filtered(const oracle_sw2d_ctx* c, const double* h, const double* hu, const double* hv,
                             double* R1, double* R2, double* R3) {
    if (!c->Filt) return 2;
    int rc = oracle_sw2d_rhs(c, h, hu, hv, R1, R2, R3);
    if (!rc) rc = apply_filter(c, R1) || apply_filter(c, R2) || apply_filter(c, R3);
    return rc;
}

/* One midpoint-RK2 step, in place (src/sw2d-simple/main.cpp:132-151). */
int oracle_sw2d_step_rk2(const oracle_sw2d_ctx* c, double* h, double* hu, double* hv, double dt, int filter) {
    const size_t nT = (size_t)c->Np * c->K;
    const int threads = c->threads;
    double* w = (double*)malloc(sizeof(double) * 6 * nT);
    if (!w) return 1;
    double *R1 = w, *R2 = w + nT, *R3 = w + 2 * nT, *h1 = w + 3 * nT, *hu1 = w + 4 * nT, *hv1 = w + 5 * nT;
    int rc = filter ? oracle_sw2d_rhs_filtered(c, h, hu, hv, R1, R2, R3) : oracle_sw2d_rhs(c, h, hu, hv, R1, R2, R3);
    if (!rc) {
        PARFOR
        for (size_t i = 0; i < nT; ++i) {
            h1[i] = h[i] + 0.5 * dt * R1[i];
            hu1[i] = hu[i] + 0.5 * dt * R2[i];
            hv1[i] = hv[i] + 0.5 * dt * R3[i];
        }
        rc = filter ? oracle_sw2d_rhs_filtered(c, h1, hu1, hv1, R1, R2, R3)
                    : oracle_sw2d_rhs(c, h1, hu1, hv1, R1, R2, R3);
    }
    if (!rc) {
        PARFOR
        for (size_t i = 0; i < nT; ++i) {
            h[i] += dt * R1[i];
            hu[i] += dt * R2[i];
            hv[i] += dt * R3[i];
        }
    }
    free(w);
    return rc;
}

/* One SSP-RK2 (Heun) step with the sponge relaxation of the variant-B driver, in place
 * (src/sw2d/main.cpp:211-235): q1 = q + dt R(q); hu1,hv1 /= 1 + s x^2; q = (q + q1 + dt R(q1))/2; sponge. */
int oracle_sw2d_step_ssprk2(const oracle_sw2d_ctx* c, double* h, double* hu, double* hv, double dt, int filter,
                            double sponge) {
    const size_t nT = (size_t)c->Np * c->K;
    double* w = (double*)malloc(sizeof(double) * 6 * nT);
    if (!w) return 1;
    double *R1 = w, *R2 = w + nT, *R3 = w + 2 * nT, *h1 = w + 3 * nT, *hu1 = w + 4 * nT, *hv1 = w + 5 * nT;
    int rc = filter ? oracle_sw2d_rhs_filtered(c, h, hu, hv, R1, R2, R3) : oracle_sw2d_rhs(c, h, hu, hv, R1, R2, R3);
    if (!rc) {
        for (size_t i = 0; i < nT; ++i) {
            h1[i] = h[i] + dt * R1[i];
            hu1[i] = hu[i] + dt * R2[i];
            hv1[i] = hv[i] + dt * R3[i];
            hu1[i] /= (1.0 + sponge * hu1[i] * hu1[i]);
            hv1[i] /= (1.0 + sponge * hv1[i] * hv1[i]);
        }
        rc = filter ? oracle_sw2d_rhs_filtered(c, h1, hu1, hv1, R1, R2, R3)
                    : oracle_sw2d_rhs(c, h1, hu1, hv1, R1, R2, R3);
    }
    if (!rc)
        for (size_t i = 0; i < nT; ++i) {
            h[i] = 0.5 * (h[i] + h1[i] + dt * R1[i]);
            hu[i] = 0.5 * (hu[i] + hu1[i] + dt * R2[i]);
            hv[i] = 0.5 * (hv[i] + hv1[i] + dt * R3[i]);
            hu[i] /= (1.0 + sponge * hu[i] * hu[i]);
            hv[i] /= (1.0 + sponge * hv[i] * hv[i]);
        }
    free(w);
    return rc;
}

/* num_stages LSERK4 stages starting at stage index `first` (mod 5), in place:
 *   res = a_i res + dt RHS(q);  q += b_i res        (src/advec1d/main.cpp:92-102) */
int oracle_sw2d_lserk4_stages(const oracle_sw2d_ctx* c, double* h, double* hu, double* hv, double* res1, double* res2,
                              double* res3, double dt, int first, int num_stages) {
    const size_t nT = (size_t)c->Np * c->K;
    const int threads = c->threads;
    double* w = (double*)malloc(sizeof(double) * 3 * nT);
    if (!w) return 1;
    double *R1 = w, *R2 = w + nT, *R3 = w + 2 * nT;
    int rc = 0;
    for (int s = 0; s < num_stages && !rc; ++s) {
        const double a = rk4a[(first + s) % 5], b = rk4b[(first + s) % 5];
        rc = oracle_sw2d_rhs(c, h, hu, hv, R1, R2, R3);
        if (rc) break;
        PARFOR
        for (size_t i = 0; i < nT; ++i) {
            res1[i] = a * res1[i] + dt * R1[i];
            res2[i] = a * res2[i] + dt * R2[i];
            res3[i] = a * res3[i] + dt * R3[i];
            h[i] += b * res1[i];
            hu[i] += b * res2[i];
            hv[i] += b * res3[i];
        }
    }
    free(w);
    return rc;
}

/* Fsc_max = max(|Fscale| * spd[vmapM]) and max |h - H|   (src/sw2d-simple/main.cpp:153-167).
 * dt = CFL / ((N+1)^2 * 0.5 * Fsc_max) is formed by the caller. H may be NULL (then max |h|). */
int oracle_sw2d_dt(const oracle_sw2d_ctx* c, const double* h, const double* hu, const double* hv, const double* H,
                   double* fsc_max, double* eta_max) {
    const int Np = c->Np, K = c->K, nfl = 3 * c->Nfp;
    const size_t nT = (size_t)Np * K, nF = (size_t)nfl * K;
    double* spd = (double*)malloc(sizeof(double) * (2 * nT + nF));
    if (!spd) return 1;
    double *spdVec = spd + nT, *fsVec = spd + 2 * nT;
    double em = 0.0;
    int nan = 0;
    for (size_t i = 0; i < nT; ++i) {
        const double u = hu[i] / h[i], v = hv[i] / h[i];
        spd[i] = sqrt(u * u + v * v) + sqrt(c->g * h[i]);
        const double e = fabs(H ? h[i] - H[i] : h[i]);
        if (e != e) nan = 1;
        if (e > em) em = e;
    }
    full_to_vector(spd, spdVec, Np, K, 1);
    full_to_vector(c->Fscale, fsVec, nfl, K, 1);
    double m = 0.0;
    for (size_t i = 0; i < nF; ++i) {
        const double val = fabs(fsVec[i]) * spdVec[c->vmapM[i]];
        if (val != val) nan = 1;
        if (val > m) m = val;
    }
    *fsc_max = nan ? NAN : m;
    *eta_max = nan ? NAN : em;
    free(spd);
    return 0;
}

/* blitzdg::advec1d::computeRHS, src/advec1d/main.cpp:126-188 (alpha = 0: upwind) */
int oracle_advec1d_rhs(int Np, int K, const double* Dr, const double* Lift, const double* rx, const double* Fscale,
                       const double* nx, const int* vmapM, const int* vmapP, int mapI, int mapO, double cvel,
                       const double* u, double* RHS) {
    const int nfl = 2;
    const size_t nF = (size_t)nfl * K, nT = (size_t)Np * K;
    double* buf = (double*)malloc(sizeof(double) * (5 * nF + 2 * nT));
    if (!buf) return 1;
    double *nxVec = buf, *uM = buf + nF, *uP = buf + 2 * nF, *du = buf + 3 * nF, *duMat = buf + 4 * nF;
    double *uVec = buf + 5 * nF, *Dru = uVec + nT;
    const double alpha = 0.0;
    full_to_vector(nx, nxVec, nfl, K, 1);
    full_to_vector(u, uVec, Np, K, 1);
    for (size_t i = 0; i < nF; ++i) {
        uM[i] = uVec[vmapM[i]];
        uP[i] = uVec[vmapP[i]];
    }
    uP[mapO] = uM[mapO];
    uP[mapI] = 0;
    for (size_t i = 0; i < nF; ++i) du[i] = (uM[i] - uP[i]) * 0.5 * (cvel * nxVec[i] - (1 - alpha) * fabs(cvel * nxVec[i]));
    vector_to_full(du, duMat, nfl, K, 1);
    contract(Dr, u, Dru, Np, Np, K, 1);
    for (size_t i = 0; i < nT; ++i) RHS[i] = -cvel * rx[i] * Dru[i];
    for (size_t i = 0; i < nF; ++i) du[i] = Fscale[i] * duMat[i]; /* surfaceRHS */
    contract(Lift, du, Dru, Np, nfl, K, 1);
    for (size_t i = 0; i < nT; ++i) RHS[i] += Dru[i];
    free(buf);
    return 0;
}

/* LSERK4 loop of src/advec1d/main.cpp:86-111 for num_steps steps, in place. */
int oracle_advec1d_steps(int Np, int K, const double* Dr, const double* Lift, const double* rx, const double* Fscale,
                         const double* nx, const int* vmapM, const int* vmapP, int mapI, int mapO, double cvel,
                         double dt, int num_steps, double* u) {
    const size_t nT = (size_t)Np * K;
    double* w = (double*)calloc(2 * nT, sizeof(double));
    if (!w) return 1;
    double *RHS = w, *res = w + nT;
    int rc = 0;
    for (int step = 0; step < num_steps && !rc; ++step)
        for (int s = 0; s < 5; ++s) {
            rc = oracle_advec1d_rhs(Np, K, Dr, Lift, rx, Fscale, nx, vmapM, vmapP, mapI, mapO, cvel, u, RHS);
            if (rc) break;
            for (size_t i = 0; i < nT; ++i) {
                res[i] = rk4a[s] * res[i] + dt * RHS[i];
                u[i] += rk4b[s] * res[i];
            }
        }
    free(w);
    return rc;
}
