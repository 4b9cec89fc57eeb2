import sys, os
sys.argv = [sys.argv[0], "500"]
import torch
torch.cuda.set_device(0)
x = torch.zeros(8, device="cuda"); torch.cuda.synchronize()
print("torch hip", torch.version.hip)
exec(open(os.path.join(os.environ["GRAFT_REPO_ROOT"], "tools/probe_bench_phases.py")).read())
