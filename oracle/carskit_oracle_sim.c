/*
 * carskit_oracle_sim.c -- CPU restatement (fp64, single thread, order-exact) of the remaining SGD recommenders of
 * SVD++ and the similarity-based CAMF family.  TEST INFRASTRUCTURE ONLY, PARITY UNPINNED (see
 * carskit_oracle.h: no reference tests, no JVM; pinned by the independent restatement oracle/oracle_np_sim.py and by
 * hand-computed known answers in tests/test_oracle_sim.py).
 *
 * Follows, line by line (paths relative to the reference root):
 *   SVD++     src/carskit/alg/baseline/cf/SVDPlusPlus.java:46-146      (2-D train matrix, implicit-feedback rows Y)
 *   CAMF_ICS  src/carskit/alg/cars/adaptation/dependent/sim/CAMF_ICS.java:33-133
 *   CAMF_LCS  .../sim/CAMF_LCS.java:34-147
 *   CAMF_MCS  .../sim/CAMF_MCS.java:38-166
 * Java evaluates a*b*c as (a*b)*c and a*b/c as (a*b)/c, one rounding per operator, no FMA (-ffp-contract=off).
 * Math.pow(x, 2) is x*x (fdlibm e_pow.c, y == 2); Math.sqrt is correctly rounded, as C's sqrt.
 *
 * In all three CAMF_*CS models the i-th condition of a context is paired with EmptyContextConditions.get(i), the i-th ":na"
 * condition in header order (ContextRecommender.java:43, DataDAO.java:213-214); the per-tuple updates go through a guava
 * HashBasedTable<index1, index2, value> whose iteration order is irrelevant here: the pairs of one tuple touch pairwise
 * distinct cells / rows (a condition belongs to exactly one dimension), so every order gives the same result.
 */
#include "carskit_oracle.h"

#include <math.h>
#include <stddef.h>
#include <stdlib.h>

static double row_mult(const double *a, const double *b, int k) { /* librec DenseMatrix.rowMult: left to right */
    double s = 0.0;
    for (int f = 0; f < k; ++f) s += a[f] * b[f];
    return s;
}

/* SymmMatrix(numConditions) of CAMF_ICS is stored here as a full n x n array kept symmetric: get(i,j) == get(j,i) */
static double sym_get(const orc_sim_problem *p, int a, int b) { return p->ccMatrix[(size_t)a * p->n_conds + b]; }
static void sym_set(const orc_sim_problem *p, int a, int b, double v) {
    p->ccMatrix[(size_t)a * p->n_conds + b] = v;
    p->ccMatrix[(size_t)b * p->n_conds + a] = v;
}

double orc_sim_predict(const orc_sim_problem *p, int32_t u, int32_t j, int32_t ctx) {
    const int k = p->k;
    const double *pu = p->P + (size_t)u * k, *qj = p->Q + (size_t)j * k;
    if (p->model == ORC_SVDPP) { /* SVDPlusPlus.java:138-146 */
        double pred = p->globalMean + p->userBias[u] + p->itemBias[j] + row_mult(pu, qj, k);
        const int32_t b = p->ui_ptr[u], e = p->ui_ptr[u + 1];
        const double w = sqrt((double)(e - b));
        for (int32_t t = b; t < e; ++t) pred += row_mult(p->Y + (size_t)p->ui_items[t] * k, qj, k) / w;
        return pred;
    }
    double pred = row_mult(pu, qj, k);
    const int32_t b = p->ctx_ptr[ctx], e = p->ctx_ptr[ctx + 1];
    if (p->model == ORC_CAMF_ICS) { /* CAMF_ICS.java:53-59 */
        for (int32_t i = 0; i < e - b; ++i) pred = pred * sym_get(p, p->ctx_conds[b + i], p->empty_conds[i]);
        return pred;
    }
    if (p->model == ORC_CAMF_LCS) { /* CAMF_LCS.java:43-61 (the ranking branch: plain dot product of the two vectors) */
        for (int32_t i = 0; i < e - b; ++i)
            pred = pred * row_mult(p->cfMatrix + (size_t)p->ctx_conds[b + i] * p->numF, p->cfMatrix + (size_t)p->empty_conds[i] * p->numF, p->numF);
        return pred;
    }
    /* CAMF_MCS.java:53-68 */
    double dist = 0.0;
    for (int32_t i = 0; i < e - b; ++i) {
        const double d = p->cVector[p->ctx_conds[b + i]] - p->cVector[p->empty_conds[i]];
        dist += d * d;
    }
    dist = sqrt(dist);
    return pred * (1 - dist);
}

static double svdpp_epoch(const orc_sim_problem *p, double lRate) { /* SVDPlusPlus.java:58-128 */
    const int k = p->k;
    const double regU = p->regU, regI = p->regI, regB = p->regB;
    double loss = 0.0;
    double *sum_ys = (double *)malloc(sizeof(double) * (size_t)k);
    for (int64_t t = 0; t < p->n; ++t) {
        const int32_t u = p->u[t], j = p->j[t];
        const double ruj = p->r[t];
        const double pred = orc_sim_predict(p, u, j, -1);
        const double euj = ruj - pred;
        loss += euj * euj;
        const int32_t b = p->ui_ptr[u], e = p->ui_ptr[u + 1];
        const double w = sqrt((double)(e - b));
        double bu = p->userBias[u];
        double sgd = euj - regB * bu;
        p->userBias[u] += lRate * sgd;
        loss += (regB * bu) * bu;
        double bj = p->itemBias[j];
        sgd = euj - regB * bj;
        p->itemBias[j] += lRate * sgd;
        loss += (regB * bj) * bj;
        for (int f = 0; f < k; ++f) {
            double sum_f = 0;
            for (int32_t q = b; q < e; ++q) sum_f += p->Y[(size_t)p->ui_items[q] * k + f];
            sum_ys[f] = w > 0 ? sum_f / w : sum_f;
        }
        double *pu = p->P + (size_t)u * k, *qj = p->Q + (size_t)j * k;
        for (int f = 0; f < k; ++f) {
            const double puf = pu[f], qjf = qj[f];
            const double sgd_u = euj * qjf - regU * puf;
            const double sgd_j = euj * (puf + sum_ys[f]) - regI * qjf;
            pu[f] += lRate * sgd_u;
            qj[f] += lRate * sgd_j;
            loss += (regU * puf) * puf + (regI * qjf) * qjf;
            for (int32_t q = b; q < e; ++q) {
                double *y = p->Y + (size_t)p->ui_items[q] * k + f;
                const double ykf = *y;
                const double delta_y = (euj * qjf) / w - regU * ykf;
                *y += lRate * delta_y;
                loss += (regU * ykf) * ykf;
            }
        }
    }
    free(sum_ys);
    return loss * 0.5;
}

#define ORC_MAX_DIMS 64

static double sim_epoch(const orc_sim_problem *p, double lRate) {
    const int k = p->k;
    const double regU = p->regU, regI = p->regI, regC = p->regC;
    double loss = 0.0;
    for (int64_t t = 0; t < p->n; ++t) {
        const int32_t u = p->u[t], j = p->j[t], ctx = p->ctx[t];
        const double rujc = p->r[t];
        double *pu = p->P + (size_t)u * k, *qj = p->Q + (size_t)j * k;
        const int32_t b = p->ctx_ptr[ctx], e = p->ctx_ptr[ctx + 1];
        const int nc = e - b;
        int32_t i1[ORC_MAX_DIMS], i2[ORC_MAX_DIMS];
        double val[ORC_MAX_DIMS];
        int nupd = 0;
        double simc = 1.0;
        const double dotRating = row_mult(pu, qj, k);
        double pred = dotRating;
        double scale; /* the factor the P/Q gradient is multiplied with: simc (ICS, LCS) or 1 - dist (MCS) */

        if (p->model == ORC_CAMF_MCS) { /* CAMF_MCS.java:86-150 */
            double dist = 0.0;
            for (int i = 0; i < nc; ++i) {
                const int32_t index1 = p->ctx_conds[b + i], index2 = p->empty_conds[i];
                const double pos1 = p->cVector[index1], pos2 = p->cVector[index2];
                const double diff = pos1 - pos2;
                dist += diff * diff;
                if (index1 != index2) {
                    i1[nupd] = index1, i2[nupd] = index2, val[nupd] = diff;
                    ++nupd;
                }
                loss += (regC * pos1) * pos1 + (regC * pos2) * pos2;
            }
            dist = sqrt(dist);
            const double sim = 1 - dist;
            pred *= sim;
            const double euj = rujc - pred;
            loss += euj * euj;
            for (int q = 0; q < nupd; ++q) {
                const double pos1 = p->cVector[i1[q]], pos2 = p->cVector[i2[q]];
                if (dist == 0) dist = p->lowbound; /* sticks for the rest of this tuple, incl. the factor loop below */
                double pos1_update = pos1 + lRate * (((euj * dotRating) * val[q]) / dist - regC * pos1);
                double pos2_update = pos2 - lRate * (((euj * dotRating) * val[q]) / dist + regC * pos2);
                pos1_update = (pos1_update < 0) ? p->lowbound : pos1_update;
                pos1_update = (pos1_update > p->upbound) ? p->upbound - p->lowbound : pos1_update;
                pos2_update = (pos2_update < 0) ? p->lowbound : pos2_update;
                pos2_update = (pos2_update > p->upbound) ? p->upbound - p->lowbound : pos2_update;
                p->cVector[i1[q]] = pos1_update;
                p->cVector[i2[q]] = pos2_update;
            }
            scale = 1 - dist;
            for (int f = 0; f < k; ++f) {
                const double puf = pu[f], qjf = qj[f];
                const double delta_u = (euj * qjf) * scale - regU * puf;
                const double delta_j = (euj * puf) * scale - regI * qjf;
                pu[f] += lRate * delta_u;
                qj[f] += lRate * delta_j;
                loss += (regU * puf) * puf + (regI * qjf) * qjf;
            }
            continue;
        }

        /* CAMF_ICS.java:80-96 / CAMF_LCS.java:82-97 */
        for (int i = 0; i < nc; ++i) {
            const int32_t index1 = p->ctx_conds[b + i], index2 = p->empty_conds[i];
            double sim = 1.0;
            if (index1 != index2) {
                sim = p->model == ORC_CAMF_ICS
                          ? sym_get(p, index1, index2)
                          : row_mult(p->cfMatrix + (size_t)index1 * p->numF, p->cfMatrix + (size_t)index2 * p->numF, p->numF);
                i1[nupd] = index1, i2[nupd] = index2, val[nupd] = sim;
                ++nupd;
                simc *= sim;
            }
            if (p->model == ORC_CAMF_ICS) loss += (regC * sim) * sim; /* commented out in CAMF_LCS.java:94 */
            pred = pred * sim;
        }
        const double euj = rujc - pred;
        loss += euj * euj;
        if (p->model == ORC_CAMF_ICS) { /* CAMF_ICS.java:101-110 */
            for (int q = 0; q < nupd; ++q) {
                double update = val[q];
                update += lRate * (((euj * dotRating) * simc) / update - regC * update);
                sym_set(p, i1[q], i2[q], update);
            }
        } else { /* CAMF_LCS.java:103-121 */
            for (int q = 0; q < nupd; ++q) {
                double *c1 = p->cfMatrix + (size_t)i1[q] * p->numF, *c2 = p->cfMatrix + (size_t)i2[q] * p->numF;
                const double sim = val[q];
                for (int f = 0; f < p->numF; ++f) {
                    const double c1f = c1[f], c2f = c2[f];
                    const double delta_c1 = (((euj * dotRating) * simc) * c2f) / sim - regC * c1f;
                    const double delta_c2 = (((euj * dotRating) * simc) * c1f) / sim - regC * c2f;
                    c1[f] += lRate * delta_c1;
                    c2[f] += lRate * delta_c2;
                    loss += (regC * c1f) * c1f + (regC * c2f) * c2f;
                }
            }
        }
        scale = simc;
        for (int f = 0; f < k; ++f) { /* CAMF_ICS.java:114-125 */
            const double puf = pu[f], qjf = qj[f];
            const double delta_u = (euj * qjf) * scale - regU * puf;
            const double delta_j = (euj * puf) * scale - regI * qjf;
            pu[f] += lRate * delta_u;
            qj[f] += lRate * delta_j;
            loss += (regU * puf) * puf + (regI * qjf) * qjf;
        }
    }
    return loss * (p->model == ORC_CAMF_MCS ? 0.05 : 0.5); /* CAMF_MCS.java:158 really says 0.05 */
}

double orc_sim_epoch(const orc_sim_problem *p, double lRate) {
    return p->model == ORC_SVDPP ? svdpp_epoch(p, lRate) : sim_epoch(p, lRate);
}
