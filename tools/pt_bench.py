import sys
sys.path.insert(0, "/root/repo/tools"); sys.path.insert(0, "hpc-ops_amd"); sys.path.insert(0, ".")
import torch, bench, hpc
dev = torch.device("cuda:0"); F8 = torch.float8_e4m3fn
E, k, H, I = 64, 8, 4096, 11008
torch.manual_seed(41)
guw = torch.randint(-80, 80, (E, 2 * I, H), dtype=torch.int8, device=dev).view(F8)
dw = torch.randint(-80, 80, (E, H, I), dtype=torch.int8, device=dev).view(F8)
gus, ds, ams = torch.rand(E, device=dev) * 0.01, torch.rand(E, device=dev) * 0.01, torch.ones(1, device=dev)
for T in (1024, 4096, 16384):
    ids = torch.sort(torch.multinomial(torch.ones(T, E, device=dev), k).to(torch.int32), dim=1)[0]
    sc = torch.rand(T, k, device=dev)
    x = (torch.randn(T, H, device=dev) / 100).to(F8)
    us = bench.timed(lambda: hpc.fuse_moe_pertensor_fp8(x, guw, dw, gus, ds, ams, ids, sc, 0, E), iters=10, warm=2)
    print("pertensor T", T, round(us, 1), "us", round(2.0 * T * k * 3 * I * H / us / 1e6, 1), "TFLOPS", flush=True)
