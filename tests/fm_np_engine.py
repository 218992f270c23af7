"""A NumPy engine with the FM phase interface (tests only): the same sparse field-parallel formulation the GPU
kernels use (carskit_amd/csrc/fm_kernels.hip), written with np.add.at, so the multi-rank exchange logic in
carskit_amd.dist.ShardedFMRunner can be exercised on CPU under gloo."""
import numpy as np
import torch


class NumpyFMEngine:
    def __init__(self, k, n_users, n_items, n_conds, n_dims, u, j, ctx, r, w0, w, V, regLw, regLf, global_size):
        self.k, self.nu, self.ni, self.nc = k, n_users, n_items, n_conds
        self.u, self.j, self.c, self.r = u.astype(np.int64), j.astype(np.int64), ctx.astype(np.int64), r.astype(np.float64)
        self.xc = 1.0 / n_dims
        self.w0, self.w, self.V = float(w0), np.array(w, np.float64), np.array(V, np.float64)
        self.regLw, self.regLf, self.size = regLw, regLf, global_size
        self.part = torch.zeros(2 * max(n_users, n_items, n_conds, 2), dtype=torch.float64)
        self.has_c = self.c < n_conds
        self._init()

    def _feat(self, field):
        if field == 0:
            return self.u, np.ones(len(self.u)), np.ones(len(self.u), bool), 0, self.nu
        if field == 1:
            return self.j, np.ones(len(self.u)), np.ones(len(self.u), bool), self.nu, self.ni
        return np.where(self.has_c, self.c, 0), np.full(len(self.u), self.xc), self.has_c, self.nu + self.ni, self.nc

    def _init(self):
        X = []
        self.Q = np.zeros((len(self.u), self.k))
        lin = np.full(len(self.u), self.w0)
        sq = np.zeros((len(self.u), self.k))
        for f in range(3):
            idx, x, m, base, _ = self._feat(f)
            contrib = self.V[base + idx] * (x * m)[:, None]
            self.Q += contrib
            sq += contrib ** 2
            lin += self.w[base + idx] * x * m
        self.err = self.r - (lin + 0.5 * ((self.Q ** 2) - sq).sum(axis=1))

    def num_phases(self):
        return 4 + 3 * self.k

    def _decode(self, ph):
        if ph == 0:
            return -1, -1
        if ph < 4:
            return ph - 1, -1
        return (ph - 4) % 3, (ph - 4) // 3

    def phase_tensor(self, ph):
        field, _ = self._decode(ph)
        cnt = 2 if field < 0 else 2 * (self.nu, self.ni, self.nc)[field]
        return self.part[:cnt]

    def phase_reduce(self, ph):
        field, f = self._decode(ph)
        t = self.phase_tensor(ph).numpy()
        t[:] = 0.0
        if field < 0:
            t[0] = (self.err - self.w0).sum()
            return
        idx, x, m, base, cnt = self._feat(field)
        theta = (self.w if f < 0 else self.V[:, f])[base + idx]
        h = x if f < 0 else x * self.Q[:, f] - x * x * theta
        np.add.at(t[:cnt], idx[m], ((self.err - theta * h) * h)[m])
        np.add.at(t[cnt:], idx[m], (h * h)[m])

    def phase_apply(self, ph):
        field, f = self._decode(ph)
        t = self.phase_tensor(ph).numpy()
        if field < 0:
            upd = 0.0 - t[0] / float(np.float32(self.size) + np.float32(self.regLw))   # int + float: a float sum (FM.java:161)
            self.err = self.err + upd - self.w0
            self.w0 = upd
            return
        idx, x, m, base, cnt = self._feat(field)
        reg = self.regLw if f < 0 else self.regLf
        cur = self.w[base:base + cnt] if f < 0 else self.V[base:base + cnt, f]
        upd = 0.0 - t[:cnt] / (t[cnt:] + self.size * reg)
        delta = (upd - cur)[idx] * x * m
        self.err = self.err + delta
        if f >= 0:
            self.Q[:, f] += delta
        cur[:] = upd
