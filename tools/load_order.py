import sys
sys.path.insert(0, '/root/repo')
import mpc_jellyfish_amd as mj
from importlib import import_module
mlib = import_module("mpc-jellyfish_amd.lib")
L = mj.load()
print("loaded", L.mzk_version())
if len(sys.argv) > 1:
    import torch
    print("torch avail", torch.cuda.is_available())
rc = L.mzk_init(0)
print("init rc", rc, L.mzk_last_error())
