"""Second, independently structured restatement of SVD++ and CAMF_ICS / CAMF_LCS / CAMF_MCS (pure Python floats = IEEE
doubles, one rounding per operator, lists of lists) -- TEST INFRASTRUCTURE ONLY.  Its only job is to disagree with
oracle/carskit_oracle_sim.c whenever one of the two misreads the reference: tests/test_oracle_sim.py demands bit-for-bit
equality on random small problems.  Written from the update rules, class per recommender:

  SVD++    (src/carskit/alg/baseline/cf/SVDPlusPlus.java:58-146)
      N(u) = items of u in the 2-D train matrix (ascending), w = sqrt(|N(u)|)
      pred = gm + bu + bj + <P_u,Q_j> + sum_{k in N(u)} <Y_k,Q_j> / w              (one division per k, added in order)
      e = r - pred ; bu += lr(e - regB bu) ; bj += lr(e - regB bj)
      s_f = (sum_k Y_kf) / w  (all f, before anything moves)
      per f: P_uf += lr(e Q_jf - regU P_uf) ; Q_jf += lr(e (P_uf + s_f) - regI Q_jf)   (old P_uf, Q_jf on the right)
             per k in N(u): Y_kf += lr(e Q_jf / w - regU Y_kf)                          (old Q_jf)
      loss = 0.5 (sum e^2 + regB bu^2 + regB bj^2 + per f: regU P^2 + regI Q^2 + per k regU Y^2)
  CAMF_ICS (sim/CAMF_ICS.java:62-131): pred = <P_u,Q_j> * prod_i S[c_i, na_i], S symmetric, pairs with c_i == na_i count as 1
      S[c_i,na_i] += lr(e * dot * simc / S - regC S), simc = product over the pairs with c_i != na_i
      P_uf += lr(e Q_jf simc - regU P_uf), Q likewise; loss = 0.5 (e^2 + sum_i regC sim_i^2 + factor terms)
  CAMF_LCS (sim/CAMF_LCS.java:63-146): S[c,na] replaced by <C_c, C_na> (numF-vectors); both vectors move:
      C_cf += lr(e dot simc C_naf / sim - regC C_cf), C_naf += lr(e dot simc C_cf / sim - regC C_naf) (old values on the right)
      loss has regC C_cf^2 + regC C_naf^2 per f instead of the sim^2 term
  CAMF_MCS (sim/CAMF_MCS.java:70-165): positions x_c on a line; dist = sqrt(sum_i (x_ci - x_nai)^2); pred = dot * (1 - dist)
      x_c  = clip(x_c  + lr(e dot diff / dist - regC x_c)),  x_na = clip(x_na - lr(e dot diff / dist + regC x_na)),
      clip(v) = 1e-100 if v < 0 else (upbound - 1e-100 if v > upbound else v); dist == 0 is replaced by 1e-100 from the first pair on;
      loss = 0.05 (!) * (e^2 + sum_i regC x_ci^2 + regC x_nai^2 + factor terms)
"""
import math


def _dot(a, b):
    s = 0.0
    for x, y in zip(a, b):
        s += x * y
    return s


class _Base:
    def __init__(self, k, gm, regU, regI, regB, regC):
        self.k, self.gm, self.regU, self.regI, self.regB, self.regC = k, gm, regU, regI, regB, regC

    def _factors(self, pu, qj, e, scale, lr, loss):
        """the per-factor loop; `loss` is the reference's single running sum, continued in place"""
        for f in range(self.k):
            p, q = pu[f], qj[f]
            du = e * q * scale - self.regU * p if scale is not None else e * q - self.regU * p
            dj = e * p * scale - self.regI * q if scale is not None else e * p - self.regI * q
            pu[f] = p + lr * du
            qj[f] = q + lr * dj
            loss += self.regU * p * p + self.regI * q * q
        return loss


class SVDPP(_Base):
    def __init__(self, k, n_users, u, j, r, P, Q, bu, bj, Y, gm, regU, regI, regB):
        super().__init__(k, gm, regU, regI, regB, 0.0)
        self.u, self.j, self.r, self.P, self.Q, self.bu, self.bj, self.Y = u, j, r, P, Q, bu, bj, Y
        self.N = [[] for _ in range(n_users)]
        for uu, jj in sorted(zip(u, j)):
            self.N[uu].append(jj)

    def predict(self, u, j):
        pred = self.gm + self.bu[u] + self.bj[j] + _dot(self.P[u], self.Q[j])
        w = math.sqrt(len(self.N[u]))
        for k in self.N[u]:
            pred += _dot(self.Y[k], self.Q[j]) / w
        return pred

    def epoch(self, lr):
        loss = 0.0
        for u, j, r in zip(self.u, self.j, self.r):
            e = r - self.predict(u, j)
            loss += e * e
            items = self.N[u]
            w = math.sqrt(len(items))
            b = self.bu[u]
            self.bu[u] = b + lr * (e - self.regB * b)
            loss += self.regB * b * b
            b = self.bj[j]
            self.bj[j] = b + lr * (e - self.regB * b)
            loss += self.regB * b * b
            s = []
            for f in range(self.k):
                t = 0.0
                for k in items:
                    t += self.Y[k][f]
                s.append(t / w if w > 0 else t)
            pu, qj = self.P[u], self.Q[j]
            for f in range(self.k):
                p, q = pu[f], qj[f]
                pu[f] = p + lr * (e * q - self.regU * p)
                qj[f] = q + lr * (e * (p + s[f]) - self.regI * q)
                loss += self.regU * p * p + self.regI * q * q
                for k in items:
                    y = self.Y[k][f]
                    self.Y[k][f] = y + lr * (e * q / w - self.regU * y)
                    loss += self.regU * y * y
        return loss * 0.5


class _Ctx(_Base):
    def __init__(self, k, u, j, ctx, r, conds, empty, P, Q, gm, regU, regI, regC):
        super().__init__(k, gm, regU, regI, 0.0, regC)
        self.u, self.j, self.ctx, self.r, self.conds, self.empty, self.P, self.Q = u, j, ctx, r, conds, empty, P, Q

    def pairs(self, c):
        return list(zip(self.conds[c], self.empty))   # zip stops at the shorter list, like the reference's index loop


class ICS(_Ctx):
    def __init__(self, *a, S=None):
        super().__init__(*a)
        self.S = S

    def predict(self, u, j, c):
        pred = _dot(self.P[u], self.Q[j])
        for a, b in self.pairs(c):
            pred = pred * self.S[a][b]
        return pred

    def epoch(self, lr):
        loss = 0.0
        for u, j, c, r in zip(self.u, self.j, self.ctx, self.r):
            dot = _dot(self.P[u], self.Q[j])
            pred, simc, upd = dot, 1.0, []
            for a, b in self.pairs(c):
                sim = 1.0
                if a != b:
                    sim = self.S[a][b]
                    upd.append((a, b, sim))
                    simc *= sim
                loss += self.regC * sim * sim
                pred = pred * sim
            e = r - pred
            loss += e * e
            for a, b, sim in upd:
                v = sim + lr * (e * dot * simc / sim - self.regC * sim)
                self.S[a][b] = v
                self.S[b][a] = v
            loss = self._factors(self.P[u], self.Q[j], e, simc, lr, loss)
        return loss * 0.5


class LCS(_Ctx):
    def __init__(self, *a, C=None):
        super().__init__(*a)
        self.C = C

    def predict(self, u, j, c):
        pred = _dot(self.P[u], self.Q[j])
        for a, b in self.pairs(c):
            pred = pred * _dot(self.C[a], self.C[b])
        return pred

    def epoch(self, lr):
        loss = 0.0
        for u, j, c, r in zip(self.u, self.j, self.ctx, self.r):
            dot = _dot(self.P[u], self.Q[j])
            pred, simc, upd = dot, 1.0, []
            for a, b in self.pairs(c):
                sim = 1.0
                if a != b:
                    sim = _dot(self.C[a], self.C[b])
                    upd.append((a, b, sim))
                    simc *= sim
                pred = pred * sim
            e = r - pred
            loss += e * e
            for a, b, sim in upd:
                ca, cb = self.C[a], self.C[b]
                for f in range(len(ca)):
                    x, y = ca[f], cb[f]
                    ca[f] = x + lr * (e * dot * simc * y / sim - self.regC * x)
                    cb[f] = y + lr * (e * dot * simc * x / sim - self.regC * y)
                    loss += self.regC * x * x + self.regC * y * y
            loss = self._factors(self.P[u], self.Q[j], e, simc, lr, loss)
        return loss * 0.5


class MCS(_Ctx):
    LOW = 1.0 / math.pow(10, 100)

    def __init__(self, *a, x=None, n_dims=1):
        super().__init__(*a)
        self.x = x
        self.up = 1.0 / math.sqrt(n_dims)

    def _dist(self, c):
        d = 0.0
        for a, b in self.pairs(c):
            t = self.x[a] - self.x[b]
            d += t * t
        return math.sqrt(d)

    def predict(self, u, j, c):
        return _dot(self.P[u], self.Q[j]) * (1 - self._dist(c))

    def _clip(self, v):
        v = self.LOW if v < 0 else v
        return self.up - self.LOW if v > self.up else v

    def epoch(self, lr):
        loss = 0.0
        for u, j, c, r in zip(self.u, self.j, self.ctx, self.r):
            dot = _dot(self.P[u], self.Q[j])
            upd, d = [], 0.0
            for a, b in self.pairs(c):
                xa, xb = self.x[a], self.x[b]
                t = xa - xb
                d += t * t
                if a != b:
                    upd.append((a, b, t))
                loss += self.regC * xa * xa + self.regC * xb * xb
            dist = math.sqrt(d)
            e = r - dot * (1 - dist)
            loss += e * e
            for a, b, t in upd:
                if dist == 0:
                    dist = self.LOW
                xa, xb = self.x[a], self.x[b]
                self.x[a] = self._clip(xa + lr * (e * dot * t / dist - self.regC * xa))
                self.x[b] = self._clip(xb - lr * (e * dot * t / dist + self.regC * xb))
            loss = self._factors(self.P[u], self.Q[j], e, 1 - dist, lr, loss)
        return loss * 0.05
