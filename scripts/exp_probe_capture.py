import os, torch, torch.distributed as dist, sys
sys.path.insert(0, '/root/repo')
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29544")
dev=torch.device("cuda",0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from seervideoldm_amd import parallel
print("probe_capture:", parallel.probe_capture(dev))
dist.destroy_process_group()
