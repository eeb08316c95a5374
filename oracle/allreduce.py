"""AllReduce + residual + RMSNorm oracle — TEST INFRASTRUCTURE (see oracle/__init__.py).
Restates reference tests/test_fuse_allreduce_rmsnorm_low_latency.py:16-29 (identical in
test_fuse_allreduce_rmsnorm_high_throughput.py:15-28): bf16 sequential sum of the rank inputs,
bf16 residual add, RMSNorm in fp32 rounded to bf16, bf16 multiply by the weight."""
import torch


def rmsnorm(x, w, rms_norm_eps):
    mean_square = x.float().pow(2).mean(-1, keepdim=True)
    return (x.float() * torch.rsqrt(mean_square + rms_norm_eps)).to(torch.bfloat16) * w.reshape(1, -1)


def ref_allreduce_rmsnorm(input_list, residual, weight, rms_norm_eps):
    input_sum = torch.zeros_like(input_list[0])
    for x in input_list:
        input_sum += x
    output_residual = input_sum + residual
    return output_residual, rmsnorm(output_residual, weight, rms_norm_eps)
