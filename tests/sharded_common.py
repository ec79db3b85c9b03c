"""Test-only backend of `shard_protocol.ShardProtocol`: the arithmetic of ONE rank on the structured oracle.
It keeps ONLY what the rank owns valid -- the rows of Sigma of its features plus the camera rows, the W / V rows it
computed, the per-feature records it measured -- and poisons everything else with NaN after every operation, so a
protocol that reads a panel before the matching all-gather (or forgets one) produces NaN and fails loudly."""
import numpy as np

import ekf_oracle as o


class OracleShardBackend:
    def __init__(self, filt: o.StructuredFilter):
        self.f = filt
        self.camera_dim = filt.camera_dim
        self.own_f = range(0)
        self.own_r = range(0)

    # -- layout / ownership -----------------------------------------------------------------------
    @property
    def n(self):
        return self.f.n

    def positions(self):
        return [ft.position_in_state for ft in self.f.features]

    def set_owner(self, features, rows):
        self.own_f, self.own_r = features, rows
        self._poison()

    def valid_rows(self):
        return np.r_[0:self.camera_dim, self.own_r.start:self.own_r.stop].astype(np.int64)

    def _poison(self):
        mask = np.ones(self.f.n, bool)
        mask[self.valid_rows()] = False
        self.f.Sigma[mask, :] = np.nan

    # -- resize (every rank runs these) -------------------------------------------------------------
    def add_feature(self, u, v):
        return self.f.add_feature(u, v)

    def remove_features(self, indices):
        for i in reversed(indices):                      # descending, vR.cpp:1296-1299
            self.f.remove_feature(i)

    def linearity_flags(self, own):
        f = self.f
        out = np.zeros(len(own), bool)
        for j, i in enumerate(own):
            ft = f.features[i]
            if ft.coding == o.INV:
                p = ft.position_in_state
                out[j] = f.inverse_depth_to_xyz_world(f.mu[p:p + 6], 2, p)[2]
        return out

    def convert(self, which):
        """Convert the listed features in index order, as convert2XYZ_ifLinearAll walks them (vR.cpp:776-780): the
        decision was taken by the owners; the Jacobian needs mu only (replicated)."""
        f = self.f
        T = f.T
        for i in which:
            ft = f.features[i]
            pos = ft.position_in_state
            y, J_y, _ = f.inverse_depth_to_xyz_world(f.mu[pos:pos + 6], 1)
            f._convert_covariance(pos, J_y)
            f.mu = np.concatenate([f.mu[:pos], y.astype(T), f.mu[pos + 6:]])
            ft.coding = o.XYZ
            for g in f.features[i + 1:]:
                g.position_in_state -= 3
        return len(which)

    def sigma_rows(self, rows):
        return self.f.Sigma[rows.start:rows.stop, :].copy()

    def set_sigma_rows(self, rows, arr):
        self.f.Sigma[rows.start:rows.stop, :] = arr

    # -- predict ------------------------------------------------------------------------------------
    def predict_camera_and_strips(self):
        f = self.f
        Ft, Q = f._motion((0, 0, 0), (0, 0, 0), False)
        f.predict_covariance(Ft, Q)
        f.mu[0:13] = o.predict_state(f.mu[0:13], (0, 0, 0), (0, 0, 0), f.dT, f.T)
        for ft in f.features:                            # nothing of the per-feature records survives a predict
            ft.h = ft.Hc = ft.Hf = None
            ft.is_in_innovation = False

    def measure(self, own):
        f = self.f
        rec = np.zeros((len(own), 29), f.T)
        for j, i in enumerate(own):
            ft = f.features[i]
            hi, Hc, Hf, vis, rem = o.DenseFilter.measure_feature(f, ft)
            ft.h, ft.Hc, ft.Hf, ft.is_in_innovation = hi, Hc, Hf, vis
            if rem:
                ft.remove_flag = True
            rec[j, 0:2] = hi
            rec[j, 2:16] = Hc.reshape(-1)
            rec[j, 16:16 + ft.size], rec[j, 22:22 + ft.size] = Hf[0], Hf[1]   # 2 x 6 block, an XYZ feature fills 2 x 3
            rec[j, 28] = int(vis) | (int(rem) << 1)
        return rec

    def set_records(self, frange, rec):
        f = self.f
        for j, i in enumerate(frange):
            ft = f.features[i]
            s = ft.size
            ft.h = rec[j, 0:2].copy()
            ft.Hc = rec[j, 2:16].reshape(2, 7).copy()
            ft.Hf = np.stack([rec[j, 16:16 + s], rec[j, 22:22 + s]]).copy()
            fl = int(rec[j, 28])
            ft.is_in_innovation = bool(fl & 1)
            if fl & 2:
                ft.remove_flag = True

    def visible_indices(self):
        return self.f.visible_indices()

    # -- update -------------------------------------------------------------------------------------
    def begin_update(self, z, indices, plane):
        f = self.f
        T = f.T
        for i in indices:
            assert f.features[i].h is not None and np.all(np.isfinite(f.features[i].Hc))      # gathered before use
        self.indices, self.plane = list(indices), bool(plane)
        M = len(indices)
        self.m = 2 * M + (3 if plane else 0)
        h = f.stacked_h(self.indices)
        z = np.asarray(z, T).reshape(-1)
        if plane:
            h = np.concatenate([h, np.array([f.mu[1], f.mu[4], f.mu[6]], T)])
            z = np.concatenate([z, np.zeros(3, T)])
        self.nu = z - h
        W = f.sigma_Ht(self.indices, plane)               # rows of foreign features are NaN (their Sigma rows are)
        keep = np.zeros(f.n, bool)
        keep[self.valid_rows()] = True
        W[~keep] = np.nan
        self.W = W
        self.S = np.full((self.m, self.m), np.nan, T)
        self.V = np.full((f.n, self.m), np.nan, T)

    def innovation_rows(self, k0, k1):
        f = self.f
        T = f.T
        sub = self.indices[k0:k1]
        rows = f.H_times(self.W, sub, False) if sub else np.zeros((0, self.m), T)
        for j in range(len(sub)):
            rows[2 * j, 2 * (k0 + j)] += T(f.sigma_pixel_2)
            rows[2 * j + 1, 2 * (k0 + j) + 1] += T(f.sigma_pixel_2)
        self.S[2 * k0:2 * k1] = rows
        if self.plane:                                    # the plane rows read camera rows of W: every rank forms them
            M = len(self.indices)
            for e, r in enumerate((1, 4, 6)):
                self.S[2 * M + e] = self.W[r]
                self.S[2 * M + e, 2 * M + e] += T(0.00001)
        return rows

    def set_S_rows(self, row0, arr):
        self.S[row0:row0 + arr.shape[0]] = arr

    def factor(self):
        assert np.all(np.isfinite(self.S))
        L = np.linalg.cholesky(self.S.astype(np.float64)).astype(self.f.T)
        self.Linv_T = np.linalg.inv(L).T.astype(self.f.T)  # Z = L^-T
        self.y = (self.Linv_T.T @ self.nu).astype(self.f.T)

    def factor_distributed(self, rank, world, gather, nb=2):
        """The DISTRIBUTED chain (Filter::dist_chain_steps) on this rank's numpy copy of S, block size
        `nb` (the library: 128): the rank owns the row blocks I with I % world == rank; every other row of its copy is
        poisoned with NaN except the diagonal blocks, which every rank keeps up to date for itself.  Per block step:
        factor of the diagonal block (every rank), panel of the OWN blocks, one all-gather of the panel through equal
        padded slots, trailing update of the own rows and of every remaining diagonal block.  A step that reads a row
        nobody handed over fails on the NaN; at the end L must be complete on every rank and equal the factor."""
        assert np.all(np.isfinite(self.S))
        m = self.m
        nblk = (m + nb - 1) // nb
        mp_ = nblk * nb
        A = np.eye(mp_)
        A[:m, :m] = self.S.astype(np.float64)
        A[np.triu_indices(mp_, 1)] = np.nan                       # only the lower triangle exists
        blk = lambda I: slice(I * nb, (I + 1) * nb)
        for I in range(nblk):
            if I % world != rank:                                 # somebody else's rows: stale in the library, NaN here
                keep = A[blk(I), blk(I)].copy()
                A[blk(I), :] = np.nan
                A[blk(I), blk(I)] = keep
        for j in range(nblk):
            D = A[blk(j), blk(j)]
            D = np.tril(D) + np.tril(D, -1).T
            assert np.all(np.isfinite(D)), f"rank {rank}: diagonal block {j} is not up to date"
            Ljj = np.linalg.cholesky(D)
            A[blk(j), blk(j)] = Ljj
            Linv = np.linalg.inv(Ljj)
            mine = [I for I in range(j + 1, nblk) if I % world == rank]
            P_own = np.zeros((len(mine), nb, nb))
            for b_, I in enumerate(mine):
                P_own[b_] = A[blk(I), blk(j)] @ Linv.T
            assert np.all(np.isfinite(P_own)), f"rank {rank}: own panel rows of step {j} are not up to date"
            counts = [len([I for I in range(j + 1, nblk) if I % world == g]) for g in range(world)]
            if max(counts) > 0:
                parts = gather(P_own, counts)
                for g in range(world):
                    for b_, I in enumerate([I for I in range(j + 1, nblk) if I % world == g]):
                        A[blk(I), blk(j)] = parts[g][b_]
            assert np.all(np.isfinite(A[(j + 1) * nb:, blk(j)]))
            for I in range(j + 1, nblk):
                PI = A[blk(I), blk(j)]
                if I % world == rank:
                    for K in range(j + 1, I + 1):
                        A[blk(I), blk(K)] -= PI @ A[blk(K), blk(j)].T
                else:
                    A[blk(I), blk(I)] -= PI @ PI.T
        L = np.tril(A[:m, :m])
        assert np.all(np.isfinite(L))
        ref = np.linalg.cholesky(self.S.astype(np.float64))
        assert np.max(np.abs(L - ref)) <= 1e-9 * max(1.0, np.max(np.abs(ref))), "distributed factor differs"
        L = L.astype(self.f.T)
        self.Linv_T = np.linalg.inv(L.astype(np.float64)).T.astype(self.f.T)
        self.y = (self.Linv_T.T @ self.nu).astype(self.f.T)

    def solve_chunk(self, c0, c1):
        rows = self.valid_rows()
        self.V[rows, c0:c1] = self.W[rows] @ self.Linv_T[:, c0:c1]
        return self.V[self.own_r.start:self.own_r.stop, c0:c1].copy()

    def set_V_rows(self, rows, c0, c1, arr):
        self.V[rows.start:rows.stop, c0:c1] = arr

    def downdate_chunk(self, c0, c1):
        Vg = self.V[:, c0:c1]
        assert np.all(np.isfinite(Vg))
        rows = self.valid_rows()
        self.f.Sigma[rows, :] = self.f.Sigma[rows, :] - Vg[rows] @ Vg.T

    def finish_update(self):
        f = self.f
        assert np.all(np.isfinite(self.V))
        f.mu = f.mu + self.V @ self.y
        f.normalize_quaternion()
        self._poison()
