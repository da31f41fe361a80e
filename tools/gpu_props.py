import torch
p = torch.cuda.get_device_properties(0)
print(p)
for k in dir(p):
    if 'shared' in k.lower() or 'lds' in k.lower(): print(k, getattr(p,k))
