"""The super-resolution sampling leg of bench.py (configs[4] on one GPU) alone, for rocprofv3 --kernel-trace --stats."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from wdno_amd import _lib
_lib.load()
os.environ['WDNO_SAMPLE_GRAPH'] = '0'
print(bench.sr_leg(torch.device('cuda', 0), batch=2, steps=4))
