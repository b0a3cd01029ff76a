"""PfbArbResampler alone: Msamples/s in at three rates (the serial phase chain of k_arb_plan).  tools/arb_time.py"""
import importlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
pkg = importlib.import_module("gr4-packet-modem_amd")
n = 1 << 22
x = (torch.randn(n, device="cuda") + 1j * torch.randn(n, device="cuda")).to(torch.complex64)
for rate in (1.0 + 50e-6, 0.7, 1.3):
    r = pkg.PfbArbResampler(rate)
    r.process_bulk(x[:4096])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    y = r.process_bulk(x)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    yy = y[0] if isinstance(y, tuple) else y
    print(f"rate {rate}: {n} in -> {yy.numel()} out in {dt*1e3:.1f} ms = {n/dt/1e6:.1f} Msps in")
