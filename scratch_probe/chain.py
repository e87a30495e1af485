import torch, ctypes, os
print("torch hip", torch.version.hip)
torch.cuda.init(); torch.zeros(1, device='cuda')
lib = ctypes.CDLL(os.path.join(os.path.dirname(__file__), "libchain.so"))
for g in (0,1):
    for big in (0,1):
        lib.run_chain(2000, g, big)
