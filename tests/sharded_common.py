"""Test-only stand-in for the HIP shard backend: the same four phases on numpy row panels.
It keeps ONLY the rows a rank owns (its features + the camera rows) valid and poisons every
other row of its Sigma / W / V copies with NaN after each phase, so an orchestration that reads a
panel before the matching all-gather fails loudly."""
import numpy as np
import torch

import ekf_oracle as o


class OracleShardBackend:
    def __init__(self, filt: o.StructuredFilter, rank, world):
        self.f = filt
        self.rank, self.world = rank, world
        self.N = filt.num_features()
        assert self.N % world == 0
        nf = self.N // world
        self.f0, self.f1 = rank * nf, (rank + 1) * nf
        self.camera_dim = filt.camera_dim
        self.rows_per_rank = 6 * nf
        self.r0 = self.camera_dim + 6 * self.f0
        self.r1 = self.r0 + self.rows_per_rank
        T = filt.T
        n = filt.n
        tdt = torch.float64 if T == np.float64 else torch.float32
        self._t = {"h": torch.zeros((self.N, 2), dtype=tdt), "Hc": torch.zeros((self.N, 14), dtype=tdt),
                   "Hf": torch.zeros((self.N, 12), dtype=tdt), "flags": torch.zeros((self.N, 1), dtype=torch.uint8),
                   "S": torch.zeros((2 * self.N, 2 * self.N), dtype=tdt), "V": torch.zeros((n, 2 * self.N), dtype=tdt)}
        self.W = np.full((n, 2 * self.N), np.nan, dtype=T)
        self._poison()

    def tensors(self):
        return self._t

    def own_rows(self):
        return np.r_[0:self.camera_dim, self.r0:self.r1]

    def _poison(self):
        mask = np.ones(self.f.n, bool)
        mask[self.own_rows()] = False
        self.f.Sigma[mask, :] = np.nan

    def predict(self):
        f = self.f
        Ft, Q = f._motion((0, 0, 0), (0, 0, 0), False)
        f.predict_covariance(Ft, Q)
        f.mu[0:13] = o.predict_state(f.mu[0:13], (0, 0, 0), (0, 0, 0), f.dT, f.T)
        t = self._t
        for name in ("h", "Hc", "Hf"):
            t[name].fill_(float("nan"))
        for i in range(self.f0, self.f1):
            hi, Hc, Hf, vis, rem = f.measure_feature(f.features[i])
            t["h"][i] = torch.from_numpy(hi)
            t["Hc"][i] = torch.from_numpy(Hc.reshape(-1))
            t["Hf"][i] = torch.from_numpy(Hf.reshape(-1))
            t["flags"][i, 0] = int(vis) | (int(rem) << 1)

    def innovation(self, z, M):
        f = self.f
        T = f.T
        t = self._t
        h = t["h"].numpy().astype(T)
        Hc = t["Hc"].numpy().astype(T).reshape(self.N, 2, 7)
        Hf = t["Hf"].numpy().astype(T).reshape(self.N, 2, 6)
        assert np.all(np.isfinite(h)) and np.all(np.isfinite(Hc))       # gathered before use
        self.nu = np.asarray(z, T).reshape(-1) - h.reshape(-1)
        rows = self.own_rows()
        self.W[:] = np.nan
        for k, ft in enumerate(f.features):
            p = ft.position_in_state
            self.W[rows, 2 * k:2 * k + 2] = f.Sigma[rows, 0:7] @ Hc[k].T + f.Sigma[rows, p:p + 6] @ Hf[k].T
        S = t["S"]
        S.fill_(float("nan"))
        for k in range(self.f0, self.f1):
            p = f.features[k].position_in_state
            blk = Hc[k] @ self.W[0:7, :] + Hf[k] @ self.W[p:p + 6, :]
            blk[0, 2 * k] += T(f.sigma_pixel_2)
            blk[1, 2 * k + 1] += T(f.sigma_pixel_2)
            S[2 * k:2 * k + 2] = torch.from_numpy(blk)

    def factor_solve(self):
        f = self.f
        T = f.T
        S = self._t["S"].numpy().astype(T)
        assert np.all(np.isfinite(S))
        L = np.linalg.cholesky(S)
        Linv = np.linalg.inv(L)
        rows = self.own_rows()
        V = self._t["V"]
        V.fill_(float("nan"))
        V[rows] = torch.from_numpy(self.W[rows] @ Linv.T)
        self.y = Linv @ self.nu

    def downdate(self):
        f = self.f
        T = f.T
        V = self._t["V"].numpy().astype(T)
        assert np.all(np.isfinite(V))
        f.mu = f.mu + V @ self.y
        rows = self.own_rows()
        f.Sigma[rows, :] = f.Sigma[rows, :] - V[rows] @ V.T
        f.normalize_quaternion()
        self._poison()
