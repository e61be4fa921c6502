#!/usr/bin/env python3
"""The torch.distributed calls of bench.py (init_process_group("nccl", device_id=...), barrier, all_reduce MAX on a device tensor,
destroy) with one rank on one MI355X: what a one-GPU box can check of the RCCL branch (two ranks on one GPU are refused by RCCL;
the two-rank protocol itself is covered with gloo in tests/test_dist_cpu.py)."""
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29511")
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda",0))
torch.cuda.set_device(0)
dist.barrier()
t=torch.tensor([1.5],device="cuda:0",dtype=torch.float64); dist.all_reduce(t,op=dist.ReduceOp.MAX); print("nccl single-rank ok", t.item())
dist.barrier(); dist.destroy_process_group()
