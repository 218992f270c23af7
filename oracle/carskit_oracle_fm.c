/*
 * carskit_oracle_fm.c -- CPU restatement of the reference's FM recommender.  TEST INFRASTRUCTURE ONLY,
 * PARITY UNPINNED (see carskit_oracle.h).
 *
 * Follows src/carskit/alg/cars/adaptation/dependent/FM.java line by line, INCLUDING its dense O(size*p)
 * loops over absent (zero) features: adding x*0 terms and `sum += 0 + reg` one rating at a time is what
 * fixes the floating-point value of every accumulator, so the dense walk is kept (small sizes only).
 * Quirks restated as they are:
 *   - the context feature index is numUsers+numItems+c where c is the CONTEXT-COMBINATION id, compared
 *     against p = numUsers+numItems+numConditions (FM.java:62,81,85-86): contexts with id >= numConditions
 *     contribute no feature;
 *   - the ALS denominators add the regulariser once per rating (FM.java:181,201);
 *   - the error/Q update of a factor uses x_il, not h_lf (FM.java:209-210);
 *   - signs as written (`update = 0 - update/sum`).
 */
#include <math.h>
#include <stdlib.h>

#include "carskit_oracle.h"

/* FM.getFeatureVector (FM.java:76-91) as a sparse triple: indices ascending (u < nu+j < nu+ni+c) */
static int features(const orc_fm *m, int64_t i, int32_t idx[3], double val[3]) {
    int n = 0;
    idx[n] = m->u[i]; val[n++] = 1.0;
    idx[n] = m->n_users + m->j[i]; val[n++] = 1.0;
    int32_t ic = m->n_users + m->n_items + m->ctx[i];
    if (ic < m->p) { idx[n] = ic; val[n++] = 1.0 / (double)m->n_ctx_dims; }
    return n;
}

static double feature_of(const orc_fm *m, int64_t i, int32_t l) { /* fvalues.get(i, l) */
    int32_t idx[3];
    double val[3];
    int n = features(m, i, idx, val);
    for (int q = 0; q < n; ++q)
        if (idx[q] == l) return val[q];
    return 0.0;
}

/* FM.predict (FM.java:93-113) for an arbitrary (u, j, c); zero features add exact zeros and are skipped */
double orc_fm_predict(const orc_fm *m, int32_t u, int32_t j, int32_t c) {
    int32_t idx[3];
    double val[3];
    int n = 0;
    idx[n] = u; val[n++] = 1.0;
    idx[n] = m->n_users + j; val[n++] = 1.0;
    int32_t ic = m->n_users + m->n_items + c;
    if (ic < m->p) { idx[n] = ic; val[n++] = 1.0 / (double)m->n_ctx_dims; }
    double pred = m->w0;
    for (int q = 0; q < n; ++q) pred += m->w[idx[q]] * val[q];
    double sum = 0.0;
    for (int f = 0; f < m->k; ++f) {
        double sum1 = 0.0, sum2 = 0.0;
        for (int q = 0; q < n; ++q) {
            double dot = m->V[(size_t)idx[q] * m->k + f] * val[q];
            sum1 += dot;
            sum2 += dot * dot;
        }
        sum += sum1 * sum1 - sum2;
    }
    return pred + 0.5 * sum;
}

/* the pre-pass of buildModel (FM.java:117-146): errors[] and Q[][] */
void orc_fm_init(orc_fm *m) {
    for (int64_t i = 0; i < m->size; ++i) {
        m->errors[i] = m->r[i] - orc_fm_predict(m, m->u[i], m->j[i], m->ctx[i]);
        int32_t idx[3];
        double val[3];
        int n = features(m, i, idx, val);
        for (int f = 0; f < m->k; ++f) {
            double value = 0.0;
            for (int q = 0; q < n; ++q) value += m->V[(size_t)idx[q] * m->k + f] * val[q];
            m->Q[(size_t)i * m->k + f] = value;
        }
    }
}

/* one iteration of the sweep (FM.java:148-218); returns loss (after *= 0.05) */
double orc_fm_sweep(orc_fm *m) {
    const int64_t size = m->size;
    const double regLw = m->regLw, regLf = m->regLf;
    double loss = 0.0;
    /* w0 (FM.java:153-169) */
    double update_w0 = 0.0;
    for (int64_t i = 0; i < size; ++i) {
        double err = m->errors[i];
        update_w0 += err - m->w0;
        loss += err * err;
    }
    /* FM.java:161 `update_w0/(size + regLw)`: size is an int and regLw a float FIELD (FM.java:47), so Java's binary numeric promotion
     * computes the sum in FLOAT -- found by executing the reference's source (oracle/mint_reference_src.py), not by reading it */
    update_w0 = update_w0 / (double)((float)(int)size + (float)regLw);
    update_w0 = 0 - update_w0;
    for (int64_t i = 0; i < size; ++i) m->errors[i] = m->errors[i] + update_w0 - m->w0;
    loss += regLw * m->w0 * m->w0;
    m->w0 = update_w0;
    /* w (FM.java:172-191) */
    for (int32_t l = 0; l < m->p; ++l) {
        double update_wl = 0.0, sum = 0.0;
        for (int64_t i = 0; i < size; ++i) {
            double fl = feature_of(m, i, l);
            update_wl += (m->errors[i] - m->w[l] * fl) * fl;
            sum += fl * fl + regLw;
        }
        update_wl = 0 - update_wl / sum;
        for (int64_t i = 0; i < size; ++i) m->errors[i] = m->errors[i] + (update_wl - m->w[l]) * feature_of(m, i, l);
        loss += regLw * m->w[l] * m->w[l];
        m->w[l] = update_wl;
    }
    /* V (FM.java:194-217) */
    for (int f = 0; f < m->k; ++f)
        for (int32_t l = 0; l < m->p; ++l) {
            double update_Vlf = 0.0, sum = 0.0;
            double *Vlf = &m->V[(size_t)l * m->k + f];
            for (int64_t i = 0; i < size; ++i) {
                double fl = feature_of(m, i, l);
                double qif = m->Q[(size_t)i * m->k + f];
                double hlf = fl * qif - fl * fl * (*Vlf);
                update_Vlf += (m->errors[i] - (*Vlf) * hlf) * hlf;
                sum += hlf * hlf + regLf;
                loss += regLf * (qif * qif);
            }
            update_Vlf = 0 - update_Vlf / sum;
            for (int64_t i = 0; i < size; ++i) {
                double fl = feature_of(m, i, l);
                m->errors[i] = m->errors[i] + (update_Vlf - *Vlf) * fl;
                m->Q[(size_t)i * m->k + f] = m->Q[(size_t)i * m->k + f] + (update_Vlf - *Vlf) * fl;
            }
            *Vlf = update_Vlf;
        }
    loss *= 0.05;
    return loss;
}
