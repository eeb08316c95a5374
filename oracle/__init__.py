"""oracle/ — CPU restatements of the reference algorithms for the decode-step hot path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke() and bench.py's
`cpu_baseline` leg may import anything from here, and only as the checker / the timed CPU
baseline.  The product (hpc-ops_amd/) never imports it and has no CPU fallback.

Each function restates one PyTorch-eager reference embedded in the reference's tests/ (the
reference's own parity oracle) and cites it file:line.  Pinning:

* fp paths (attention, MoE, all-reduce+RMSNorm, RMSNorm): the restatements are compared with the
  reference's own in-file oracle functions, cut out of the reference test files and executed on the
  CPU by tests/golden/ref_extract.py + make_golden.py (build container only); the resulting
  input/output vectors are committed as tests/golden/fp_golden.npz and replayed by
  tests/test_oracle_golden.py (CPU: pins the oracles; GPU: HIP path vs the reference outputs).
* scheduler (integer): oracle/sched_oracle.c restates assign_attention_decode_task_sync
  (reference src/attention/decode/assign_task.cu:362-492); oracle/Makefile also compiles that
  very function from the reference sources into oracle/_ref/ (when /root/reference exists) and
  tests/golden/make_golden.py records its outputs as golden task maps.
"""
