"""GPU box check: does MIOpen on this box give batch-size-dependent convolution results on the small test network
(the executor then runs one image at a time)?  Exit code 0: yes (the 'odd' kind of box), 1: no."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dipoorlet_amd import models
g = models.resnet18(seed=11, image=64)
s = g.make_session()
ok = s.batched_ok()
print("batched_ok:", ok)
sys.exit(1 if ok else 0)
