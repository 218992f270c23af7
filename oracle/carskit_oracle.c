/*
 * carskit_oracle.c -- see carskit_oracle.h.  TEST INFRASTRUCTURE ONLY, PARITY UNPINNED
 * (no reference tests / no JVM; pinned by oracle_np.py + hand-computed known answers).
 *
 * Plain C99, fp64, one thread, exactly the reference's visiting order and operator order.
 * Java evaluates a*b*c as (a*b)*c and a+b+c as (a+b)+c with one rounding each and never
 * fuses multiply-add; the expressions below are parenthesised to say so and the Makefile
 * passes -ffp-contract=off.  Math.pow(x,2) is x*x in fdlibm (e_pow.c special case y==2).
 */
#include "carskit_oracle.h"

#include <math.h>
#include <stddef.h>

/* librec DenseMatrix.rowMult (lib/librec-v1.4-alpha.jar): s=0; s+=m[f]*n[f] left to right */
static double row_mult(const double *a, const double *b, int k) {
    double s = 0.0;
    for (int f = 0; f < k; ++f) s += a[f] * b[f];
    return s;
}

/* shared tail of every buildModel(): the per-factor loop, e.g. CAMF_CI.java:108-120,
 * BiasedMF.java:84-95.  puf and qjf are both read BEFORE either row is written. */
static double factor_step(double *pu, double *qj, int k, double euj, double lRate, double regU, double regI,
                          double loss) {
    for (int f = 0; f < k; ++f) {
        double puf = pu[f];
        double qjf = qj[f];
        double delta_u = euj * qjf - regU * puf;
        double delta_j = euj * puf - regI * qjf;
        pu[f] += lRate * delta_u;
        qj[f] += lRate * delta_j;
        loss += (regU * puf) * puf + (regI * qjf) * qjf;
    }
    return loss;
}

/* ---- predict() of each model ------------------------------------------------------------- */

double orc_predict(const orc_problem *p, int32_t u, int32_t j, int32_t ctx) {
    const int k = p->k;
    const double dot = row_mult(p->P + (size_t)u * k, p->Q + (size_t)j * k, k);
    double pred;
    int32_t b = 0, e = 0;
    if (p->model != ORC_BIASEDMF && p->model != ORC_PMF) {
        b = p->ctx_ptr[ctx];
        e = p->ctx_ptr[ctx + 1];
    }
    switch (p->model) {
    case ORC_PMF: /* IterativeRecommender.java:126-128 via PMF.java:85-91: the bare dot product */
        return dot;
    case ORC_BIASEDMF: /* BiasedMF.java:111-114 */
        return p->globalMean + p->userBias[u] + p->itemBias[j] + dot;
    case ORC_CAMF_C: /* CAMF_C.java:66-72 */
        pred = p->globalMean + p->userBias[u] + p->itemBias[j] + dot;
        for (int32_t t = b; t < e; ++t) pred += p->condBias[p->ctx_conds[t]];
        return pred;
    case ORC_CAMF_CI: /* CAMF_CI.java:66-72 */
        pred = p->globalMean + p->userBias[u] + dot;
        for (int32_t t = b; t < e; ++t) pred += p->icBias[(size_t)j * p->n_conds + p->ctx_conds[t]];
        return pred;
    case ORC_CAMF_CU: /* CAMF_CU.java:62-69 */
        pred = p->globalMean + p->itemBias[j] + dot;
        for (int32_t t = b; t < e; ++t) pred += p->ucBias[(size_t)u * p->n_conds + p->ctx_conds[t]];
        return pred;
    case ORC_CAMF_CUCI: /* CAMF_CUCI.java:69-76: pred += icBias + ucBias (inner add first) */
        pred = p->globalMean + dot;
        for (int32_t t = b; t < e; ++t) {
            int32_t c = p->ctx_conds[t];
            pred += p->icBias[(size_t)j * p->n_conds + c] + p->ucBias[(size_t)u * p->n_conds + c];
        }
        return pred;
    default:
        return NAN;
    }
}

void orc_predict_items(const orc_problem *p, int32_t u, int32_t ctx, int32_t n, const int32_t *items, double *out) {
    for (int32_t i = 0; i < n; ++i) out[i] = orc_predict(p, u, items[i], ctx);
}

/* ---- one epoch of buildModel() ------------------------------------------------------------ */

double orc_sgd_epoch(const orc_problem *p, double lRate) {
    const int k = p->k;
    const double regU = p->regU, regI = p->regI, regB = p->regB, regC = p->regC;
    double loss = 0.0;
    for (int64_t t = 0; t < p->n; ++t) {
        const int32_t u = p->u[t], j = p->j[t];
        const int32_t ctx = p->ctx ? p->ctx[t] : -1;
        const double rujc = p->r[t];
        const double pred = orc_predict(p, u, j, ctx); /* predict(u,j,ctx,false): unbounded */
        const double euj = rujc - pred;
        double sgd;
        int32_t cb = 0, ce = 0;
        if (p->model != ORC_BIASEDMF && p->model != ORC_PMF) {
            cb = p->ctx_ptr[ctx];
            ce = p->ctx_ptr[ctx + 1];
        }

        loss += euj * euj;

        switch (p->model) {
        case ORC_PMF: /* PMF.java:60-71: no bias terms */
            break;
        case ORC_BIASEDMF: { /* BiasedMF.java:70-82 */
            double bu = p->userBias[u];
            sgd = euj - regB * bu;
            p->userBias[u] += lRate * sgd;
            loss += (regB * bu) * bu;
            double bj = p->itemBias[j];
            sgd = euj - regB * bj;
            p->itemBias[j] += lRate * sgd;
            loss += (regB * bj) * bj;
            break;
        }
        case ORC_CAMF_C: { /* CAMF_C.java:92-115 */
            double bu = p->userBias[u];
            sgd = euj - regB * bu;
            p->userBias[u] += lRate * sgd;
            loss += (regB * bu) * bu;
            double bj = p->itemBias[j];
            sgd = euj - regB * bj;
            p->itemBias[j] += lRate * sgd;
            loss += (regB * bj) * bj;
            double bc_sum = 0.0;
            for (int32_t t2 = cb; t2 < ce; ++t2) {
                int32_t cond = p->ctx_conds[t2];
                double bc = p->condBias[cond];
                bc_sum += bc; /* reference quirk: NOT squared (CAMF_C.java:110) */
                sgd = euj - regC * bc;
                p->condBias[cond] += lRate * sgd;
            }
            loss += regB * bc_sum; /* reference quirk: regB, not regC (CAMF_C.java:115) */
            break;
        }
        case ORC_CAMF_CI: { /* CAMF_CI.java:92-106 */
            double bu = p->userBias[u];
            sgd = euj - regB * bu;
            p->userBias[u] += lRate * sgd;
            loss += (regB * bu) * bu;
            double Bic_sum = 0.0;
            for (int32_t t2 = cb; t2 < ce; ++t2) {
                double *cell = p->icBias + (size_t)j * p->n_conds + p->ctx_conds[t2];
                double Bic = *cell;
                Bic_sum += Bic * Bic;
                sgd = euj - regC * Bic;
                *cell = Bic + lRate * sgd; /* set(), not add() */
            }
            loss += regC * Bic_sum;
            break;
        }
        case ORC_CAMF_CU: { /* CAMF_CU.java:89-103 */
            double bj = p->itemBias[j];
            sgd = euj - regB * bj;
            p->itemBias[j] += lRate * sgd;
            loss += (regB * bj) * bj;
            double Buc_sum = 0.0;
            for (int32_t t2 = cb; t2 < ce; ++t2) {
                double *cell = p->ucBias + (size_t)u * p->n_conds + p->ctx_conds[t2];
                double Buc = *cell;
                Buc_sum += Buc * Buc;
                sgd = euj - regC * Buc;
                *cell = Buc + lRate * sgd;
            }
            loss += regC * Buc_sum;
            break;
        }
        case ORC_CAMF_CUCI: { /* CAMF_CUCI.java:96-111 */
            double Buc_sum = 0.0, Bic_sum = 0.0;
            for (int32_t t2 = cb; t2 < ce; ++t2) {
                int32_t cond = p->ctx_conds[t2];
                double *ucell = p->ucBias + (size_t)u * p->n_conds + cond;
                double *icell = p->icBias + (size_t)j * p->n_conds + cond;
                double Buc = *ucell, Bic = *icell;
                Buc_sum += Buc * Buc;
                Bic_sum += Bic * Bic;
                double sgdu = euj - regC * Buc;
                double sgdj = euj - regC * Bic;
                *ucell = Buc + lRate * sgdu;
                *icell = Bic + lRate * sgdj;
            }
            loss += regC * Bic_sum + regC * Buc_sum;
            break;
        }
        default:
            return NAN;
        }

        loss = factor_step(p->P + (size_t)u * k, p->Q + (size_t)j * k, k, euj, lRate, regU, regI, loss);
    }
    return loss * 0.5;
}

/* ---- isConverged / updateLRate ------------------------------------------------------------ */

static void update_lrate(orc_schedule *s, int iter) { /* IterativeRecommender.java:216-229 */
    if (s->lRate <= 0) return;
    if (s->boldDriver && iter > 1)
        s->lRate = fabs(s->last_loss) > fabs(s->loss) ? s->lRate * 1.05 : s->lRate * 0.5;
    else if (s->decay > 0 && s->decay < 1)
        s->lRate *= s->decay;
    if (s->maxLRate > 0 && s->lRate > s->maxLRate) s->lRate = s->maxLRate;
}

int orc_is_converged(orc_schedule *s, int iter, int use_measure) { /* IterativeRecommender.java:145-199 */
    if (s->earlyStop == 1) { /* Loss */
        s->measure = s->loss;
        s->last_measure = s->last_loss;
    } else if (!use_measure) {
        /* earlyStopMeasure == null: measure and last_measure both stay 0 */
    }
    float delta_measure = (float)(s->last_measure - s->measure);
    if (isnan(s->loss) || isinf(s->loss)) return -1;
    int cond1 = fabs(s->loss) < 1e-5;
    int cond2 = (delta_measure > 0) && (delta_measure < 1e-5);
    int converged = cond1 || cond2;
    if (!converged) update_lrate(s, iter);
    s->last_loss = s->loss;
    s->last_measure = s->measure;
    return converged;
}

int orc_build_model(const orc_problem *p, orc_schedule *s, int numIters, double *losses, double *lrates) {
    int iter;
    for (iter = 1; iter <= numIters; ++iter) {
        if (lrates) lrates[iter - 1] = s->lRate;
        s->loss = orc_sgd_epoch(p, s->lRate);
        if (losses) losses[iter - 1] = s->loss;
        int c = orc_is_converged(s, iter, 0);
        if (c != 0) return iter;
    }
    return numIters;
}

/* ---- evalRatings --------------------------------------------------------------------------- */

int64_t orc_eval_ratings(const orc_problem *p, int64_t n_test, const int32_t *tu, const int32_t *tj,
                         const int32_t *tctx, const double *tr, double minRate, double maxRate, double *out,
                         double *preds) {
    double sum_maes = 0, sum_mses = 0, sum_r_maes = 0, sum_r_rmses = 0;
    int64_t numCount = 0;
    for (int64_t t = 0; t < n_test; ++t) {
        double rate = tr[t];
        double pred = orc_predict(p, tu[t], tj[t], tctx ? tctx[t] : -1);
        /* predict(u,j,c,true): Recommender.java:306-317 */
        if (pred > maxRate) pred = maxRate;
        if (pred < minRate) pred = minRate;
        if (preds) preds[t] = pred;
        if (isnan(pred)) continue;
        /* Math.round(x) = floor(x + 0.5) for finite x in long range */
        double rPred = (double)(int64_t)floor(pred / minRate + 0.5) * minRate;
        double err = fabs(rate - pred);
        double r_err = fabs(rate - rPred);
        sum_maes += err;
        sum_mses += err * err;
        sum_r_maes += r_err;
        sum_r_rmses += r_err * r_err;
        numCount++;
    }
    double mae = sum_maes / (double)numCount;
    out[0] = mae;
    out[1] = sqrt(sum_mses / (double)numCount);
    out[2] = mae / (maxRate - minRate);
    out[3] = sum_r_maes / (double)numCount;
    out[4] = sqrt(sum_r_rmses / (double)numCount);
    return numCount;
}

double orc_global_mean(const double *r, int64_t n) {
    /* librec SparseMatrix.sum() = Stats.sum(rowData) sequential; size() counts non-zero values
     * (lib/librec-v1.4-alpha.jar) */
    double s = 0.0;
    int64_t cnt = 0;
    for (int64_t t = 0; t < n; ++t) {
        s += r[t];
        if (r[t] != 0.0) cnt++;
    }
    return s / (double)cnt;
}

/* ---- java.util.Random (public algorithm: 48-bit LCG, polar-method nextGaussian) ------------- */

/* StrictMath.log = fdlibm __ieee754_log (public algorithm: argument reduction x = 2^k (1+f),
 * s = f/(2+f), degree-14 minimax polynomial in s).  glibc's log() is correctly rounded and differs
 * from fdlibm's in the last ulp for some arguments (e.g. the first nextGaussian() of seed 42), so
 * the Java stream needs fdlibm's rounding behaviour, restated here for finite positive normal x. */
static double fdlibm_log(double x) {
    static const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                        Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
                        Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
                        Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                        Lg7 = 1.479819860511658591e-01;
    union { double d; uint64_t u; } w;
    w.d = x;
    int32_t hx = (int32_t)(w.u >> 32);
    int32_t k = 0, i, j;
    if (hx < 0x00100000 || hx >= 0x7ff00000) return log(x); /* zero/subnormal/neg/inf/nan: not reached by nextGaussian */
    k += (hx >> 20) - 1023;
    hx &= 0x000fffff;
    i = (hx + 0x95f64) & 0x100000;
    w.u = ((uint64_t)(uint32_t)(hx | (i ^ 0x3ff00000)) << 32) | (w.u & 0xffffffffULL);
    x = w.d;
    k += (i >> 20);
    double f = x - 1.0, dk, R;
    if ((0x000fffff & (2 + hx)) < 3) { /* |f| < 2^-20 */
        if (f == 0.0) {
            if (k == 0) return 0.0;
            dk = (double)k;
            return dk * ln2_hi + dk * ln2_lo;
        }
        R = f * f * (0.5 - 0.33333333333333333 * f);
        if (k == 0) return f - R;
        dk = (double)k;
        return dk * ln2_hi - ((R - dk * ln2_lo) - f);
    }
    double s = f / (2.0 + f);
    dk = (double)k;
    double z = s * s;
    i = hx - 0x6147a;
    double ww = z * z;
    j = 0x6b851 - hx;
    double t1 = ww * (Lg2 + ww * (Lg4 + ww * Lg6));
    double t2 = z * (Lg1 + ww * (Lg3 + ww * (Lg5 + ww * Lg7)));
    i |= j;
    R = t2 + t1;
    if (i > 0) {
        double hfsq = 0.5 * f * f;
        if (k == 0) return f - (hfsq - s * (hfsq + R));
        return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
    }
    if (k == 0) return f - s * (f - R);
    return dk * ln2_hi - ((s * (f - R) - dk * ln2_lo) - f);
}

#define JR_MULT 0x5DEECE66DULL
#define JR_MASK ((1ULL << 48) - 1)

void orc_jrandom_seed(orc_jrandom *g, int64_t seed) {
    g->seed = ((uint64_t)seed ^ JR_MULT) & JR_MASK;
    g->haveNextNextGaussian = 0;
    g->nextNextGaussian = 0.0;
}

int32_t orc_jrandom_next(orc_jrandom *g, int bits) {
    g->seed = (g->seed * JR_MULT + 0xBULL) & JR_MASK;
    return (int32_t)((int64_t)g->seed >> (48 - bits)); /* seed < 2^48 so the shift is logical */
}

double orc_jrandom_next_double(orc_jrandom *g) {
    int64_t hi = (int64_t)orc_jrandom_next(g, 26);
    int64_t lo = (int64_t)orc_jrandom_next(g, 27);
    return (double)((hi << 27) + lo) * 0x1.0p-53;
}

double orc_jrandom_next_gaussian(orc_jrandom *g) {
    if (g->haveNextNextGaussian) {
        g->haveNextNextGaussian = 0;
        return g->nextNextGaussian;
    }
    double v1, v2, s;
    do {
        v1 = 2 * orc_jrandom_next_double(g) - 1;
        v2 = 2 * orc_jrandom_next_double(g) - 1;
        s = v1 * v1 + v2 * v2;
    } while (s >= 1 || s == 0);
    /* StrictMath.sqrt(-2 * StrictMath.log(s) / s) */
    double multiplier = sqrt(-2 * fdlibm_log(s) / s);
    g->nextNextGaussian = v2 * multiplier;
    g->haveNextNextGaussian = 1;
    return v1 * multiplier;
}

void orc_init_gaussian(orc_jrandom *g, double *a, int64_t n, double mean, double sigma) {
    for (int64_t i = 0; i < n; ++i) a[i] = mean + sigma * orc_jrandom_next_gaussian(g);
}

void orc_init_uniform(orc_jrandom *g, double *a, int64_t n, double range) {
    for (int64_t i = 0; i < n; ++i) a[i] = 0.0 + (range - 0.0) * orc_jrandom_next_double(g);
}
