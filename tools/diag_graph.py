"""diagnostic: does hipGraph capture work for the new ops, and what is the kernel-only time?"""
import sys, time
sys.path.insert(0, "hpc-ops_amd"); sys.path.insert(0, ".")
import torch, hpc
dev = torch.device("cuda:0")
n, k = 256, 4096
w = torch.randn(n, k, device=dev); wh = w.bfloat16(); wl = ((w - wh.float()) * 256).bfloat16()
flag = hpc.get_gemm_bf16xfp32_workspace(n, 8192)
import os
hpc._C.lib.hpc_tuning_set(4, int(os.environ.get('ROUTER_MODE', '0')))
for n in (256, 2048):
  w = torch.randn(n, k, device=dev); wh = w.bfloat16(); wl = ((w - wh.float()) * 256).bfloat16()
  flag = hpc.get_gemm_bf16xfp32_workspace(n, 8192)
  for m in (16, 64, 256, 1024):
      x = torch.randn(m, k, device=dev).bfloat16()
      fn = lambda: hpc.gemm_bf16xfp32(x, wh, wl, 1 / 256, True, True, flag)
      for _ in range(3): fn()
      torch.cuda.synchronize()
      try:
          g = torch.cuda.CUDAGraph()
          with torch.cuda.graph(g):
              fn()
          ok = True
      except Exception as e:
          ok = False; print("capture failed:", repr(e)[:300])
      if ok:
          for _ in range(3): g.replay()
          torch.cuda.synchronize()
          # back-to-back replays, total time / N  (no per-replay events)
          N = 200
          s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
          s.record()
          for _ in range(N): g.replay()
          e.record(); torch.cuda.synchronize()
          print(f"m={m}: graph replay back-to-back {s.elapsed_time(e) / N * 1e3:.1f} us")
      N = 200
      s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
      s.record()
      for _ in range(N): fn()
      e.record(); torch.cuda.synchronize()
      print(f"m={m}: eager back-to-back {s.elapsed_time(e) / N * 1e3:.1f} us")
