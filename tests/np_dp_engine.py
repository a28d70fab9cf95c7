"""NumPy engine for ``dist.fit_data_parallel`` (test infrastructure; the arithmetic is the CPU
oracle's).  It speaks the same protocol as ``rfm_fm_fit_dp`` -- shards, the two exchanges with
the g_w0 record owned by the last rank, loss SUMS combined once per run -- through the HOST API
of ``HostStagedTransport``, so the CPU (gloo) tests exercise the product's driver, sharding,
transport and loss combination without a GPU."""
import numpy as np

from oracle import cpu_ref
from relevance_factorizationmachine_amd.dist import owner_ranges, shard_bounds


class NumpyDpEngine:
    def __init__(self, model, train, val, world, rank, exchange, transport):
        self.m, self.world, self.rank, self.exchange, self.t = model, world, rank, exchange, transport
        self.X, self.y, self.p = train["features"], train["labels"], train["pscores"]
        self.val = val
        n = self.X.shape[1]
        self.w0, self.w, self.V = cpu_ref.fm_init(model.seed, n, model.n_factors)
        self.n, self.k = n, model.n_factors
        self.sums = np.zeros((2, model.n_epochs))
        self.lo_cols = owner_ranges(n, world)

    def chunks(self):
        E, B = self.m.n_epochs, self.m.batch_size
        ids = np.stack([cpu_ref.batch_ids(self.X.shape[0], B, e) for e in range(E)])
        cut = max(1, E // 2)  # two chunks: the per-run loss combination is exercised twice
        yield 0, cut, ids[:cut]
        if E > cut:
            yield cut, E - cut, ids[cut:]

    # ---- one iteration ----------------------------------------------------
    def _grads(self, rows):
        n, k = self.n, self.k
        if len(rows) == 0:
            return np.zeros((n, k)), np.zeros(n), 0.0, np.zeros(0, dtype=np.int64)
        Xb = self.X[rows]
        _, g0, gw, GV = cpu_ref.fm_gradients(Xb, self.y[rows], self.p[rows], self.w0, self.w, self.V)
        return GV, gw, float(g0), np.unique(Xb.indices)

    def _step_dense(self, rows):
        GV, gw, g0, _ = self._grads(rows)
        g = self.t.all_reduce_sum_host(np.concatenate([GV.ravel(), gw, [g0]])) if self.world > 1 else \
            np.concatenate([GV.ravel(), gw, [g0]])
        n, k, lr = self.n, self.k, self.m.lr
        self.V -= lr * g[: n * k].reshape(n, k)
        self.w -= lr * g[n * k: n * k + n]
        self.w0 -= lr * g[-1]

    def _step_rows(self, rows):
        n, k, lr, W = self.n, self.k, self.m.lr, self.world
        GV, gw, g0, cols = self._grads(rows)
        rec = np.concatenate([cols[:, None].astype(np.float64), GV[cols], gw[cols, None]], axis=1)
        rec = np.concatenate([rec, [[float(n)] + [0.0] * k + [g0]]])  # the g_w0 record, column n
        bounds = np.concatenate([np.searchsorted(rec[:, 0], self.lo_cols), [len(rec)]]).astype(np.int64)
        allb = self.t.all_gather_host(bounds)  # [W][W+1]
        send = [rec[bounds[r]: bounds[r + 1]].copy().view(np.uint8).reshape(-1) for r in range(W)]
        wb = (k + 2) * 8
        got = self.t.all_to_all_host(send, [(allb[s, self.rank + 1] - allb[s, self.rank]) * wb for s in range(W)])
        segs = [g.view(np.float64).reshape(-1, k + 2) for g in got]
        # owner: records of a column added in RANK order, row updated once
        acc, order = {}, []
        for seg in segs:
            for r in seg:
                c = int(r[0])
                if c not in acc:
                    acc[c] = 0.0 + r[1:].copy()
                    order.append(c)
                else:
                    acc[c] = acc[c] + r[1:]
        out = np.zeros((len(order), k + 2))
        for i, c in enumerate(order):
            out[i, 0] = c
            if c == n:
                out[i, -1] = self.w0[0] - lr * acc[c][-1]
            else:
                out[i, 1:-1] = self.V[c] - lr * acc[c][:-1]
                out[i, -1] = self.w[c] - lr * acc[c][-1]
        cnts = self.t.all_gather_host(np.array([out.size * 8], dtype=np.int64)).reshape(-1)
        everything = self.t.all_to_all_host([out.copy().view(np.uint8).reshape(-1)] * W, [int(c) for c in cnts])
        for part in everything:
            for r in part.view(np.float64).reshape(-1, k + 2):
                c = int(r[0])
                if c == n:
                    self.w0[0] = r[-1]
                else:
                    self.V[c] = r[1:-1]
                    self.w[c] = r[-1]

    def run(self, first, count, chunk_first, ids):
        B = self.m.batch_size
        lo, hi = shard_bounds(B, self.world, self.rank)
        vlo, vhi = shard_bounds(self.val["features"].shape[0], self.world, self.rank)
        for it in range(first, first + count):
            rows = ids[it - chunk_first, lo:hi]
            if self.exchange == "rows" and self.world > 1:
                self._step_rows(rows)
            else:
                self._step_dense(rows)
            if len(rows):
                pred = cpu_ref.fm_predict(self.X[rows], self.w0, self.w, self.V)
                self.sums[0, it] = -len(rows) * cpu_ref.ips_logloss(self.y[rows], pred, self.p[rows])
            if vhi > vlo:
                Xv = self.val["features"][vlo:vhi]
                pred = cpu_ref.fm_predict(Xv, self.w0, self.w, self.V)
                self.sums[1, it] = -(vhi - vlo) * cpu_ref.ips_logloss(self.val["labels"][vlo:vhi], pred,
                                                                       self.val["pscores"][vlo:vhi])
        if self.world > 1:  # one combination per run of iterations
            s = self.t.all_reduce_sum_host(self.sums[:, first:first + count].reshape(-1))
            self.sums[:, first:first + count] = s.reshape(2, count)

    def predict(self, X):
        return cpu_ref.fm_predict(X, self.w0, self.w, self.V)

    def losses(self):
        nv = self.val["features"].shape[0]
        return (list(-self.sums[0] / self.m.batch_size),
                list(-self.sums[1] / nv) if nv else [float("nan")] * self.m.n_epochs)

    def close(self):
        pass
