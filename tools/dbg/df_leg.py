import sys, json, time, types
sys.argv = ["bench.py"]
import bench
import torch
from nbed_amd.backend import HipBackend
be = HipBackend()
args = types.SimpleNamespace(df_naux=4000)
out = bench.df_jk_leg(be, args, torch.cuda.synchronize)
print(json.dumps(out, indent=1))
