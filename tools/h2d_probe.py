"""Host-to-device copy rates on this box: page-locked and pageable sources, one and several streams."""
import time

import numpy as np
import torch

dev = torch.device("cuda", 0)
for mb in (20, 200):
    n = mb << 20
    pin = torch.empty(n, dtype=torch.uint8).pin_memory()
    pag = torch.from_numpy(np.ones(n, np.uint8))
    dst = torch.empty(n, dtype=torch.uint8, device=dev)
    back = torch.empty(n, dtype=torch.uint8).pin_memory()
    for name, src in (("page-locked", pin), ("pageable", pag)):
        for _ in range(2):
            dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            dst.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        print("H2D %4d MB %-11s %6.2f ms  %6.1f GB/s" % (mb, name, dt * 1e3, n / dt / 1e9))
    # four streams, a quarter each
    ss = [torch.cuda.Stream() for _ in range(4)]
    q = n // 4
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        for k, s in enumerate(ss):
            with torch.cuda.stream(s):
                dst[k * q:(k + 1) * q].copy_(pin[k * q:(k + 1) * q], non_blocking=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print("H2D %4d MB page-locked, 4 streams %6.2f ms  %6.1f GB/s" % (mb, dt * 1e3, n / dt / 1e9))
    t0 = time.perf_counter()
    for _ in range(10):
        back.copy_(dst, non_blocking=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print("D2H %4d MB page-locked %6.2f ms  %6.1f GB/s" % (mb, dt * 1e3, n / dt / 1e9))
    a = np.ones(n, np.uint8)
    b = np.empty_like(a)
    t0 = time.perf_counter()
    for _ in range(5):
        np.copyto(b, a)
    dt = (time.perf_counter() - t0) / 5
    print("host memcpy %4d MB, one thread %6.2f ms  %6.1f GB/s" % (mb, dt * 1e3, n / dt / 1e9))
