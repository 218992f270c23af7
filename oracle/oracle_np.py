"""Second, independently written restatement of the CARSKit SGD path (pure Python floats = IEEE
doubles, one rounding per operator).  TEST INFRASTRUCTURE ONLY -- PARITY UNPINNED (the reference
has no tests and cannot run here); its job is to cross-check oracle/carskit_oracle.c and to mint
the small golden fixtures under tests/golden/.

Written from the reference's formulas, model by model, in a different shape from the C file
(one class per recommender, the way the reference lays them out) so that a transcription slip
in either shows up as a disagreement.

Reference (paths relative to the reference root):
  BiasedMF   src/carskit/alg/baseline/cf/BiasedMF.java:57-114
  CAMF_C     src/carskit/alg/cars/adaptation/dependent/dev/CAMF_C.java:65-138
  CAMF_CI    .../dev/CAMF_CI.java:65-131
  CAMF_CU    .../dev/CAMF_CU.java:62-128
  CAMF_CUCI  .../dev/CAMF_CUCI.java:69-134
  schedule   src/carskit/generic/IterativeRecommender.java:145-229
  eval       src/carskit/generic/Recommender.java:306-317,504-594
"""
import math
import struct


def _f32(x):
    """Java (float) cast of a double."""
    return struct.unpack("f", struct.pack("f", x))[0]


def fdlibm_log(x):
    """StrictMath.log: fdlibm's __ieee754_log restated from the published algorithm (finite positive
    normal x only).  math.log is correctly rounded and is 1 ulp off fdlibm for some arguments."""
    ln2_hi, ln2_lo = 6.93147180369123816490e-01, 1.90821492927058770002e-10
    Lg = (6.666666666666735130e-01, 3.999999999940941908e-01, 2.857142874366239149e-01,
          2.222219843214978396e-01, 1.818357216161805012e-01, 1.531383769920937332e-01,
          1.479819860511658591e-01)
    bits = struct.unpack("<q", struct.pack("<d", x))[0]
    hx, lx = (bits >> 32) & 0xFFFFFFFF, bits & 0xFFFFFFFF
    if hx < 0x00100000 or hx >= 0x7FF00000:
        return math.log(x)
    k = (hx >> 20) - 1023
    hx &= 0x000FFFFF
    i = (hx + 0x95F64) & 0x100000
    x = struct.unpack("<d", struct.pack("<q", ((hx | (i ^ 0x3FF00000)) << 32) | lx))[0]
    k += i >> 20
    f = x - 1.0
    dk = float(k)
    if (0x000FFFFF & (2 + hx)) < 3:
        if f == 0.0:
            return 0.0 if k == 0 else dk * ln2_hi + dk * ln2_lo
        R = f * f * (0.5 - 0.33333333333333333 * f)
        return f - R if k == 0 else dk * ln2_hi - ((R - dk * ln2_lo) - f)
    s = f / (2.0 + f)
    z = s * s
    w = z * z
    t1 = w * (Lg[1] + w * (Lg[3] + w * Lg[5]))
    t2 = z * (Lg[0] + w * (Lg[2] + w * (Lg[4] + w * Lg[6])))
    R = t2 + t1
    i, j = hx - 0x6147A, 0x6B851 - hx
    if i >= 0 and j >= 0 and (i | j) > 0:   # fdlibm: (i | j) > 0 on int32s
        hfsq = 0.5 * f * f
        if k == 0:
            return f - (hfsq - s * (hfsq + R))
        return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f)
    if k == 0:
        return f - s * (f - R)
    return dk * ln2_hi - ((s * (f - R) - dk * ln2_lo) - f)


def _f32(x):
    """round to binary32 (Java float arithmetic: the operands and the result of a float operation)"""
    return struct.unpack("f", struct.pack("f", x))[0]


class JavaRandom:
    """java.util.Random from its published algorithm."""

    def __init__(self, seed):
        self.seed = (seed ^ 0x5DEECE66D) & ((1 << 48) - 1)
        self._next_gauss = None

    def next(self, bits):
        self.seed = (self.seed * 0x5DEECE66D + 0xB) & ((1 << 48) - 1)
        v = self.seed >> (48 - bits)
        if v >= 1 << 31:  # (int) cast
            v -= 1 << 32
        return v

    def next_int(self):
        return self.next(32)

    def next_double(self):
        return ((self.next(26) << 27) + self.next(27)) * (1.0 / (1 << 53))

    def next_gaussian(self):
        if self._next_gauss is not None:
            g, self._next_gauss = self._next_gauss, None
            return g
        while True:
            v1 = 2 * self.next_double() - 1
            v2 = 2 * self.next_double() - 1
            s = v1 * v1 + v2 * v2
            if 0 < s < 1:
                break
        mul = math.sqrt(-2 * fdlibm_log(s) / s)
        self._next_gauss = v2 * mul
        return v1 * mul


class Model:
    """State holder shaped like IterativeRecommender's fields (lists of Python floats)."""

    name = None

    def __init__(self, k, n_users, n_items, n_conds, ctx_conds, gm, regU, regI, regB, regC):
        self.k, self.nu, self.ni, self.nc = k, n_users, n_items, n_conds
        self.ctx_conds = ctx_conds  # list of lists: getConditions(ctx)
        self.gm = gm
        self.regU, self.regI, self.regB, self.regC = regU, regI, regB, regC
        self.P = self.Q = self.userBias = self.itemBias = self.condBias = self.ucBias = self.icBias = None

    def dot(self, u, j):
        s = 0.0
        pu, qj = self.P[u], self.Q[j]
        for f in range(self.k):
            s += pu[f] * qj[f]
        return s

    def factors(self, u, j, e, lr):
        pu, qj = self.P[u], self.Q[j]
        acc = []
        for f in range(self.k):
            p, q = pu[f], qj[f]
            du = e * q - self.regU * p
            dj = e * p - self.regI * q
            pu[f] = pu[f] + lr * du
            qj[f] = qj[f] + lr * dj
            acc.append(self.regU * p * p + self.regI * q * q)
        return acc

    def epoch(self, tuples, lr):
        loss = 0.0
        for (u, j, c, r) in tuples:
            e = r - self.predict(u, j, c)
            loss += e * e
            for term in self.biases(u, j, c, e, lr):
                loss += term
            for term in self.factors(u, j, e, lr):
                loss += term
        return loss * 0.5


class BiasedMF(Model):
    name = "BiasedMF"

    def predict(self, u, j, c):
        return self.gm + self.userBias[u] + self.itemBias[j] + self.dot(u, j)

    def biases(self, u, j, c, e, lr):
        bu = self.userBias[u]
        self.userBias[u] = self.userBias[u] + lr * (e - self.regB * bu)
        bj = self.itemBias[j]
        self.itemBias[j] = self.itemBias[j] + lr * (e - self.regB * bj)
        return [self.regB * bu * bu, self.regB * bj * bj]


class CAMF_C(Model):
    name = "CAMF_C"

    def predict(self, u, j, c):
        pred = self.gm + self.userBias[u] + self.itemBias[j] + self.dot(u, j)
        for cond in self.ctx_conds[c]:
            pred += self.condBias[cond]
        return pred

    def biases(self, u, j, c, e, lr):
        bu = self.userBias[u]
        self.userBias[u] = self.userBias[u] + lr * (e - self.regB * bu)
        bj = self.itemBias[j]
        self.itemBias[j] = self.itemBias[j] + lr * (e - self.regB * bj)
        bc_sum = 0.0
        for cond in self.ctx_conds[c]:
            bc = self.condBias[cond]
            bc_sum += bc  # not squared in the reference
            self.condBias[cond] = self.condBias[cond] + lr * (e - self.regC * bc)
        return [self.regB * bu * bu, self.regB * bj * bj, self.regB * bc_sum]  # regB in the reference


class CAMF_CI(Model):
    name = "CAMF_CI"

    def predict(self, u, j, c):
        pred = self.gm + self.userBias[u] + self.dot(u, j)
        for cond in self.ctx_conds[c]:
            pred += self.icBias[j][cond]
        return pred

    def biases(self, u, j, c, e, lr):
        bu = self.userBias[u]
        self.userBias[u] = self.userBias[u] + lr * (e - self.regB * bu)
        s = 0.0
        for cond in self.ctx_conds[c]:
            b = self.icBias[j][cond]
            s += b * b
            self.icBias[j][cond] = b + lr * (e - self.regC * b)
        return [self.regB * bu * bu, self.regC * s]


class CAMF_CU(Model):
    name = "CAMF_CU"

    def predict(self, u, j, c):
        pred = self.gm + self.itemBias[j] + self.dot(u, j)
        for cond in self.ctx_conds[c]:
            pred += self.ucBias[u][cond]
        return pred

    def biases(self, u, j, c, e, lr):
        bj = self.itemBias[j]
        self.itemBias[j] = self.itemBias[j] + lr * (e - self.regB * bj)
        s = 0.0
        for cond in self.ctx_conds[c]:
            b = self.ucBias[u][cond]
            s += b * b
            self.ucBias[u][cond] = b + lr * (e - self.regC * b)
        return [self.regB * bj * bj, self.regC * s]


class CAMF_CUCI(Model):
    name = "CAMF_CUCI"

    def predict(self, u, j, c):
        pred = self.gm + self.dot(u, j)
        for cond in self.ctx_conds[c]:
            pred += self.icBias[j][cond] + self.ucBias[u][cond]
        return pred

    def biases(self, u, j, c, e, lr):
        su = si = 0.0
        for cond in self.ctx_conds[c]:
            bu, bi = self.ucBias[u][cond], self.icBias[j][cond]
            su += bu * bu
            si += bi * bi
            self.ucBias[u][cond] = bu + lr * (e - self.regC * bu)
            self.icBias[j][cond] = bi + lr * (e - self.regC * bi)
        return [self.regC * si + self.regC * su]


class PMF(Model):
    """src/carskit/alg/baseline/cf/PMF.java:47-91: no biases, predict = rowMult."""
    name = "PMF"

    def predict(self, u, j, c):
        return self.dot(u, j)

    def biases(self, u, j, c, e, lr):
        return []


MODELS = {m.name: m for m in (BiasedMF, CAMF_C, CAMF_CI, CAMF_CU, CAMF_CUCI, PMF)}
MODEL_IDS = {"BiasedMF": 0, "CAMF_C": 1, "CAMF_CI": 2, "CAMF_CU": 3, "CAMF_CUCI": 4, "PMF": 5}


class Schedule:
    """isConverged / updateLRate with earlyStopMeasure in {None, 'Loss'}."""

    def __init__(self, init_lrate, max_lrate=-1.0, bold_driver=False, decay=-1.0, early_stop=None):
        self.lr = init_lrate
        self.max_lr, self.bold, self.decay, self.early = max_lrate, bold_driver, decay, early_stop
        self.loss = self.last_loss = 0.0
        self.measure = self.last_measure = 0.0

    def step(self, it, loss):
        self.loss = loss
        if self.early == "Loss":
            self.measure, self.last_measure = self.loss, self.last_loss
        delta_measure = _f32(self.last_measure - self.measure)
        if math.isnan(loss) or math.isinf(loss):
            raise FloatingPointError("Loss = NaN or Infinity")
        converged = abs(loss) < 1e-5 or (0 < delta_measure < 1e-5)
        if not converged and self.lr > 0:
            if self.bold and it > 1:
                self.lr = self.lr * 1.05 if abs(self.last_loss) > abs(self.loss) else self.lr * 0.5
            elif 0 < self.decay < 1:
                self.lr *= self.decay
            if self.max_lr > 0 and self.lr > self.max_lr:
                self.lr = self.max_lr
        self.last_loss, self.last_measure = self.loss, self.measure
        return converged


def build_model(model, tuples, sched, num_iters):
    losses, lrs = [], []
    for it in range(1, num_iters + 1):
        lrs.append(sched.lr)
        loss = model.epoch(tuples, sched.lr)
        losses.append(loss)
        if sched.step(it, loss):
            break
    return losses, lrs


def eval_ratings(model, tuples, min_rate, max_rate):
    sa = ss = sra = srs = 0.0
    n = 0
    for (u, j, c, r) in tuples:
        pred = model.predict(u, j, c)
        if pred > max_rate:
            pred = max_rate
        if pred < min_rate:
            pred = min_rate
        if math.isnan(pred):
            continue
        rpred = math.floor(pred / min_rate + 0.5) * min_rate
        err, rerr = abs(r - pred), abs(r - rpred)
        sa += err
        ss += err * err
        sra += rerr
        srs += rerr * rerr
        n += 1
    mae = sa / n
    return {"MAE": mae, "RMSE": math.sqrt(ss / n), "NMAE": mae / (max_rate - min_rate), "rMAE": sra / n,
            "rRMSE": math.sqrt(srs / n), "n": n}


class FM:
    """Second restatement of the reference's FM (src/carskit/alg/cars/adaptation/dependent/FM.java:57-220):
    dense feature matrix, dense loops, exactly as the Java is written (tiny sizes only)."""

    def __init__(self, k, n_users, n_items, n_conds, n_ctx_dims, tuples, w0, w, V, regLw, regLf):
        self.k, self.nu, self.ni, self.nc, self.dims = k, n_users, n_items, n_conds, n_ctx_dims
        self.p = n_users + n_items + n_conds
        self.tuples = tuples
        self.size = len(tuples)
        self.w0, self.w, self.V = w0, list(w), [list(row) for row in V]
        self.regLw, self.regLf = regLw, regLf
        self.X = [self.feature_vector(u, j, c) for (u, j, c, _) in tuples]   # fvalues
        self.errors = [0.0] * self.size
        self.Q = [[0.0] * k for _ in range(self.size)]

    def feature_vector(self, u, j, c):
        fs = [0.0] * self.p
        iu, ij, ic = u, self.nu + j, self.nu + self.ni + c
        for i in range(self.p):
            if i == iu or i == ij:
                fs[i] = 1.0
            elif i == ic:
                fs[i] = 1.0 / self.dims
        return fs

    def predict(self, u, j, c):
        fs = self.feature_vector(u, j, c)
        pred = self.w0
        for i in range(self.p):
            pred += self.w[i] * fs[i]
        total = 0.0
        for f in range(self.k):
            s1 = s2 = 0.0
            for i in range(self.p):
                d = self.V[i][f] * fs[i]
                s1 += self.V[i][f] * fs[i]
                s2 += d * d
            total += s1 * s1 - s2
        return pred + 0.5 * total

    def init(self):
        for n, (u, j, c, r) in enumerate(self.tuples):
            self.errors[n] = r - self.predict(u, j, c)
            for f in range(self.k):
                v = 0.0
                for i in range(self.p):
                    v += self.V[i][f] * self.X[n][i]
                self.Q[n][f] = v

    def sweep(self):
        loss = 0.0
        upd = 0.0
        for i in range(self.size):
            upd += self.errors[i] - self.w0
            loss += self.errors[i] * self.errors[i]
        # `size + regLw` is int + float in the reference (FM.java:47,161): a FLOAT sum (executed source: oracle/mint_reference_src.py)
        upd = 0 - upd / _f32(_f32(float(self.size)) + _f32(self.regLw))
        for i in range(self.size):
            self.errors[i] = self.errors[i] + upd - self.w0
        loss += self.regLw * self.w0 * self.w0
        self.w0 = upd
        for l in range(self.p):
            upd = tot = 0.0
            for i in range(self.size):
                fl = self.X[i][l]
                upd += (self.errors[i] - self.w[l] * fl) * fl
                tot += fl * fl + self.regLw
            upd = 0 - upd / tot
            for i in range(self.size):
                self.errors[i] = self.errors[i] + (upd - self.w[l]) * self.X[i][l]
            loss += self.regLw * self.w[l] * self.w[l]
            self.w[l] = upd
        for f in range(self.k):
            for l in range(self.p):
                upd = tot = 0.0
                for i in range(self.size):
                    fl = self.X[i][l]
                    h = fl * self.Q[i][f] - fl * fl * self.V[l][f]
                    upd += (self.errors[i] - self.V[l][f] * h) * h
                    tot += h * h + self.regLf
                    loss += self.regLf * (self.Q[i][f] * self.Q[i][f])
                upd = 0 - upd / tot
                for i in range(self.size):
                    self.errors[i] = self.errors[i] + (upd - self.V[l][f]) * self.X[i][l]
                    self.Q[i][f] = self.Q[i][f] + (upd - self.V[l][f]) * self.X[i][l]
                self.V[l][f] = upd
        return loss * 0.05
