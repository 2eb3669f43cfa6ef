import torch, time, numpy as np
n = 3840*2160*3
d = torch.zeros(n, dtype=torch.uint8, device="cuda")
def t(f, reps=5):
    ts=[]
    for _ in range(reps):
        torch.cuda.synchronize(); t0=time.perf_counter(); f(); torch.cuda.synchronize(); ts.append((time.perf_counter()-t0)*1e3)
    return min(ts), sorted(ts)[len(ts)//2]
pinned = torch.empty(n, dtype=torch.uint8).pin_memory()
print("D2H pinned 24.9MB (min, median ms):", t(lambda: pinned.copy_(d, non_blocking=True)))
touched = torch.zeros(n, dtype=torch.uint8); touched += 1
print("D2H pageable touched:", t(lambda: touched.copy_(d)))
def fresh():
    h = torch.empty(n, dtype=torch.uint8)  # untouched pages
    h.copy_(d)
print("D2H pageable fresh torch.empty:", t(fresh))
def fresh_np():
    h = np.zeros(n, dtype=np.uint8)
    torch.from_numpy(h).copy_(d)
print("D2H pageable fresh np.zeros:", t(fresh_np))
src = pinned.numpy(); dst = np.zeros(n, dtype=np.uint8); dst += 1
def mc(): np.copyto(dst, src)
print("host memcpy pinned->touched pageable, 1 thread:", t(mc))
def mc_fresh():
    dd = np.empty(n, dtype=np.uint8); np.copyto(dd, src)
print("host memcpy pinned->fresh pageable, 1 thread:", t(mc_fresh))
n8 = n*8
d64 = torch.zeros(n, dtype=torch.float64, device="cuda")
p64 = torch.empty(n, dtype=torch.float64).pin_memory()
print("D2H pinned 199MB:", t(lambda: p64.copy_(d64, non_blocking=True), 3))
t64 = torch.zeros(n, dtype=torch.float64); t64 += 1
print("D2H pageable touched 199MB:", t(lambda: t64.copy_(d64), 3))
