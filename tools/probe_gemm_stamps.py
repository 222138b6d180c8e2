#!/usr/bin/env python3
"""Where a gemm256_kernel workgroup spends its cycles (tuning build: `make -C hive_amd/csrc stamps`, run with
HIVE_AMD_LIB=hive_amd/lib/libhive_stamps.so HIVE_GEMM_TILE=256 HIVE_GEMM_PERSIST=0).  Per shape: median over workgroups of
prologue fill | K loop (of which waiting for the next stage) | drain | epilogue issue | store drain, in clock64() ticks."""
import ctypes, os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hive_amd import _lib
DT = int(__import__("os").environ.get("HIVE_PROBE_DTYPE", "2"))  # hive_dtype of the operands: 2 = bf16 (default), 1 = f16
ctx = _lib.default_context(0); lib = ctx.lib
raw = ctypes.CDLL(_lib.LIB_PATH)
for (M, N, K, epi) in [(29184, 3072, 768, 1), (29184, 1536, 768, 0), (19456, 768, 3072, 2), (4096, 4096, 4096, 0)]:
    A = torch.randn(M, K, device="cuda").bfloat16(); W = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16(); b = torch.randn(N, device="cuda") * 0.1
    R = torch.randn(M, N, device="cuda").bfloat16(); C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    run = lambda: ctx.check(lib.hive_vit_linear(ctx.handle, A.data_ptr(), DT, W.data_ptr(), b.data_ptr(), R.data_ptr() if epi == 2 else None, C.data_ptr(), M, N, K, epi))
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    n_wg = min(8192, ((M + 255) // 256) * (N // 256))
    st = np.zeros((8192, 8), dtype=np.uint64)
    assert raw.hive_debug_read_stamps(st.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(st.nbytes)) == 0
    st = st[:n_wg].astype(np.int64)
    t0 = st[:, 0].min()
    med = lambda x: float(np.median(x))
    print(f"M={M} N={N} K={K} epi={epi}: {us:.1f} us, {n_wg} workgroups, K stages {K // 64}")
    print(f"   prologue fill {med(st[:,1]-st[:,0]):8.0f} | K loop {med(st[:,2]-st[:,1]):8.0f} (waiting {med(st[:,6]):8.0f}, of it vmcnt {med(st[:,7]):8.0f}) | drain {med(st[:,3]-st[:,2]):6.0f} | "
          f"epilogue issue {med(st[:,4]-st[:,3]):8.0f} | stores land {med(st[:,5]-st[:,4]):8.0f} | whole {med(st[:,5]-st[:,0]):8.0f} ticks")
    print(f"   kernel span {st[:,5].max()-t0} ticks for {us:.1f} us -> {(st[:,5].max()-t0)/us:.1f} ticks/us; start spread: p50 {med(st[:,0]-t0):.0f} p99 {np.percentile(st[:,0]-t0, 99):.0f}")
