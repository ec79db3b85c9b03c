"""Does any kernel read device memory it never wrote?  The same short N = 4000 run (predict / update / one resize) in a fresh
process on fresh device memory, and in a process that first fills and frees 30 GB of device memory with NaN patterns, so
that the library's hipMallocs get poisoned blocks.  Identical mu / Sigma rows -> nothing uninitialised is read.
usage: python3 tools/dirty_heap_probe.py            (spawns the two runs itself)"""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]


def run(dirty, out):
    import torch
    if dirty:
        blocks = [torch.full((1 << 28,), float("nan"), device="cuda") for _ in range(30)]     # 30 x 1 GiB of NaN
        torch.cuda.synchronize()
        del blocks
        torch.cuda.empty_cache()                                                               # back to the driver: hipMalloc reuses it
    from __graft_entry__ import load_package
    pkg = load_package()
    from ekf_monoslam_amd import synthetic
    N, frames, every = int(os.environ.get("PROBE_N", "4000")), 9, 4
    cfg = pkg.kinect_config()
    px0, z = synthetic.measurement_stream(cfg, N, frames, sigma_px=0.5)
    f = pkg.VSlamFilter(cfg, capacity_features=N + 64, dtype=np.float32)
    f.setDt(1 / 30.0)
    for (u, v) in px0:
        assert f.addFeature((u, v)) == 1
    rng = np.random.default_rng(1236)
    sid = np.arange(N)
    for k in range(frames):
        f.predict()
        h, vis, rem, _ = f.predictions()
        sel = np.nonzero(vis.astype(bool) & (sid >= 0))[0].astype(np.int32)
        f.update(z[k][sid[sel]].reshape(-1), sel)
        if (k + 1) % every == 0:
            n_now = f.numOfFeatures()
            drop = sorted(rng.choice(n_now, size=n_now // 100, replace=False).tolist())
            f.removeFeatures(drop)
            sid = np.delete(sid, drop)
            for _ in range(len(drop)):
                assert f.addFeature((float(rng.uniform(20, 300)), float(rng.uniform(20, 220)))) == 1
            sid = np.concatenate([sid, -np.ones(len(drop), np.int64)])
    f.synchronize()
    mu = f.getFullState()
    n = f.stateDim()
    rows = np.r_[0:14, 1000:1100, n - 200:n]
    S = np.vstack([f.getSigmaBlock(int(r), 0, 1, n) for r in rows])
    np.savez(out, mu=mu, S=S)
    f.close()


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(sys.argv[1] == "dirty", sys.argv[2])
        sys.exit(0)
    outs = []
    for mode in ("clean", "dirty"):
        out = f"/tmp/dirty_heap_{mode}.npz"
        subprocess.run([sys.executable, os.path.abspath(__file__), mode, out], check=True)
        outs.append(np.load(out))
    a, b = outs
    dm = np.flatnonzero(a["mu"] != b["mu"])
    ds = np.argwhere(a["S"] != b["S"])
    print(f"mu: {dm.size} differing entries of {a['mu'].size}" + (f" (first: index {dm[0]}, {a['mu'][dm[0]]!r} vs {b['mu'][dm[0]]!r})" if dm.size else ""))
    print(f"Sigma rows: {len(ds)} differing entries of {a['S'].size}; NaN in the dirty run: {int(np.isnan(b['S']).sum())}" +
          (f" (first: {ds[0].tolist()})" if len(ds) else ""))
