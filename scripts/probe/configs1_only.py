"""BASELINE configs[1] alone (for rocprofv3 --kernel-trace: which kernels the 106 ms per image are)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
print(bench.side_config1(torch.device("cuda:0")))
