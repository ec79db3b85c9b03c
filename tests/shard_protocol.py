"""Test-only numpy model of the row-panel sharding protocol the library implements (csrc/ekf_capi.hip shard_*): same
ownership rules, same exchanges, padded equal slots, over a backend that does the arithmetic of ONE rank on numpy arrays
(tests/sharded_common.py: the structured oracle with everything a rank does not own poisoned with NaN, so a protocol that
reads a panel before the matching all-gather fails loudly).  CPU tests, gloo, world 2 / 4 (tests/test_sharded.py)."""
import numpy as np

IMBALANCE_LIMIT = 1.125          # Filter::sh_imbalance_limit


def partition_by_rows(pos, n, camera_dim, world):
    """Feature boundaries (world + 1 entries) that balance the state ROWS: Filter::partition_by_rows."""
    N = len(pos)
    fb = [N] * (world + 1)
    fb[0] = 0
    rows = n - camera_dim
    f = 0
    for g in range(1, world):
        target = rows * g // world
        while f < N and pos[f] - camera_dim < target:
            f += 1
        fb[g] = f
    return fb


def _all_gather_padded(own, counts, rank, world):
    """All-gather of per-rank arrays with different leading sizes through equal, padded slots (what the library's
    staging buffers are): returns the list of every rank's array."""
    import torch
    import torch.distributed as dist
    mx = max(counts)
    tail = own.shape[1:]
    slot = np.zeros((mx,) + tail, own.dtype)
    slot[:own.shape[0]] = own
    if world == 1:
        return [own]
    send = torch.from_numpy(slot)
    parts = [torch.empty_like(send) for _ in range(world)]
    dist.all_gather(parts, send)
    return [parts[g].numpy()[:counts[g]] for g in range(world)]


class ShardProtocol:
    """The sharded step over a backend that does the arithmetic of ONE rank on numpy arrays (tests/sharded_common.py:
    the structured oracle with everything a rank does not own poisoned with NaN)."""

    def __init__(self, backend, rank, world, dist_chain=False):
        self.b, self.rank, self.world = backend, rank, world
        self.dist_chain = dist_chain                          # the factorisation distributed over the ranks (round 6)
        self.fb = partition_by_rows(backend.positions(), backend.n, backend.camera_dim, world)
        self.rebalances = 0
        backend.set_owner(self.own_features(), self.own_rows())

    # -- ownership ------------------------------------------------------------------------------
    def own_features(self):
        return range(self.fb[self.rank], self.fb[self.rank + 1])

    def _row_of(self, f):
        pos = self.b.positions()
        return pos[f] if f < len(pos) else self.b.n

    def rows_of_rank(self, g):
        return range(self._row_of(self.fb[g]), self._row_of(self.fb[g + 1]))

    def own_rows(self):
        return self.rows_of_rank(self.rank)

    def _retag(self):
        self.b.set_owner(self.own_features(), self.own_rows())

    def needs_rebalance(self):
        if self.world == 1 or not len(self.b.positions()):
            return False
        mx = max(len(self.rows_of_rank(g)) for g in range(self.world))
        mean = (self.b.n - self.b.camera_dim) / self.world
        return mx > IMBALANCE_LIMIT * mean + 6.0

    # -- resize: every rank runs the operation, ownership follows (Filter::shard_after_*) -------------
    def add_feature(self, u, v):
        ok = self.b.add_feature(u, v)
        if ok:
            self.fb[self.world] = len(self.b.positions())
            self._retag()
        return ok

    def remove_features(self, indices):
        N = len(self.b.positions())
        rm = np.zeros(N, bool)
        rm[list(indices)] = True
        kept_before = np.concatenate([[0], np.cumsum(~rm)])
        self.b.remove_features(sorted(indices))
        self.fb = [int(kept_before[min(f, N)]) for f in self.fb]
        self._retag()

    def convert_all(self):
        own = np.array(list(self.own_features()), dtype=np.int64)
        flags = self.b.linearity_flags(own).astype(np.uint8)
        counts = [self.fb[g + 1] - self.fb[g] for g in range(self.world)]
        parts = _all_gather_padded(flags.reshape(-1, 1), counts, self.rank, self.world)
        allf = np.concatenate([p.reshape(-1) for p in parts]).astype(bool)
        cnt = self.b.convert(np.nonzero(allf)[0].tolist())
        self._retag()
        return cnt

    def rebalance(self):
        counts = [len(self.rows_of_rank(g)) for g in range(self.world)]
        rows = self.b.sigma_rows(self.own_rows())
        parts = _all_gather_padded(rows, counts, self.rank, self.world)
        for g in range(self.world):
            if g != self.rank:
                self.b.set_sigma_rows(self.rows_of_rank(g), parts[g])
        self.fb = partition_by_rows(self.b.positions(), self.b.n, self.b.camera_dim, self.world)
        self.rebalances += 1
        self._retag()

    # -- the step -----------------------------------------------------------------------------------
    def predict(self):
        if self.needs_rebalance():
            self.rebalance()
        b = self.b
        b.predict_camera_and_strips()
        own = list(self.own_features())
        rec = b.measure(own)                                  # (count, 29): h | Hc | Hf | flag
        counts = [self.fb[g + 1] - self.fb[g] for g in range(self.world)]
        parts = _all_gather_padded(rec, counts, self.rank, self.world)
        for g in range(self.world):
            if g != self.rank:
                b.set_records(range(self.fb[g], self.fb[g + 1]), parts[g])

    def update(self, z, indices, plane=False, chunks=2):
        b = self.b
        indices = list(indices)
        M = len(indices)
        if M == 0 and not plane:
            return
        # list positions of every rank's measured features (ascending list, contiguous ownership)
        kr = []
        k = 0
        for g in range(self.world):
            while k < M and indices[k] < self.fb[g]:
                k += 1
            e = k
            while e < M and indices[e] < self.fb[g + 1]:
                e += 1
            kr.append((k, e))
            k = e
        k0, k1 = kr[self.rank]
        b.begin_update(z, indices, plane)                     # nu (replicated), W rows {camera, own}
        S_own = b.innovation_rows(k0, k1)                     # rows 2 k0 .. 2 k1 of S
        parts = _all_gather_padded(S_own, [2 * (e - s) for s, e in kr], self.rank, self.world)
        for g in range(self.world):
            if g != self.rank:
                b.set_S_rows(2 * kr[g][0], parts[g])
        if self.dist_chain:                                   # one all-gather of the panel per block step
            b.factor_distributed(self.rank, self.world,
                                 lambda own, counts: _all_gather_padded(own, counts, self.rank, self.world))
        else:
            b.factor()                                        # replicated chain (incl. the plane / tail rows)
        m = b.m
        ends = sorted(set([m * (c + 1) // chunks for c in range(chunks)]))
        c0 = 0
        row_counts = [len(self.rows_of_rank(g)) for g in range(self.world)]
        for c1 in ends:
            if c1 == c0:
                continue
            V_own = b.solve_chunk(c0, c1)                     # own rows of V[:, c0:c1] (+ camera rows, + y, replicated)
            parts = _all_gather_padded(V_own, row_counts, self.rank, self.world)
            for g in range(self.world):
                if g != self.rank:
                    b.set_V_rows(self.rows_of_rank(g), c0, c1, parts[g])
            b.downdate_chunk(c0, c1)                          # Sigma[{camera, own}, :] -= V_g[rows] V_g^T
            c0 = c1
        b.finish_update()                                     # mu += V y, quaternion normalisation
