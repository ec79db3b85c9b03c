"""The sharded step on ONE rank (world size 1, RCCL, EKF_SHARD_FORCE_COLLECTIVE=1: every exchange still goes through
pack -> all_gather_into_tensor (nccl) -> unpack on the library's streams): the collective path end to end on the one
GPU of the box, and its compute-only time, i.e. the chain-bound floor a multi-GPU run approaches when the collectives
are free.  python tools/shard_world1.py [features] [steps]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import argparse
import torch
import torch.distributed as dist
from __graft_entry__ import load_package
pkg = load_package()
from ekf_monoslam_amd import sharded, synthetic
import bench

os.environ.setdefault("EKF_SHARD_FORCE_COLLECTIVE", "1")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
n_feat = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
cfg = pkg.kinect_config()
args = argparse.Namespace(steps=steps, warmup=5)
px0, z = synthetic.measurement_stream(cfg, n_feat, steps + 5, sigma_px=bench.SIGMA_Z_PX)
res = sharded.bench(pkg, cfg, n_feat, px0, z, args, 0, 1, torch.device("cuda", 0))
dist.destroy_process_group()
print(json.dumps({k: res[k] for k in ("value", "ms_per_step", "per_rank_kernel_ms", "allgather_ms_per_step", "roofline")}))
