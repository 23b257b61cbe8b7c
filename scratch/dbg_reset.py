import sys, faulthandler
faulthandler.enable()
sys.path.insert(0, '.')
if len(sys.argv) > 1 and sys.argv[1] == 'torch':
    import torch
    torch.cuda.set_device(0)
    x = torch.empty(10, device='cuda')
from nimpress_amd import capi
sc = capi.Scorer(1000, capi.make_params())
print("created", flush=True)
sc.reset()
print("reset ok", flush=True)
sc.push_locus(capi.ROW_ABSENT, False, 0.1, 0.1)
s, n = sc.finish(0.0)
print("finish ok", n, flush=True)
sc.reset()
print("reset2 ok", flush=True)
