"""PCIe-inclusive rate of the host-pointer entry point (zke_verify_batch): H2D of the raw e-mails, the device
pipeline, D2H of the records.  Reported in DESIGN.md §5; never bench.py's `value`."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zkemail_rs_amd as z
from zkemail_rs_amd import _abi as A
import zkemail_rs_amd  # noqa: F401  (import shim for the dotted package directory)
import synth
for name, cfg in (("c2", dict(n=1024, body_len=4096)), ("8192x4KB", dict(n=8192, body_len=4096)), ("1024x64KB", dict(n=1024, body_len=65536))):
    wl = synth.make_workload(name, seed=3, **cfg)
    packed = A.PackedBatch(wl.emails)
    eng = z.Engine(0)
    for _ in range(3):
        r = eng.verify_batch(packed)
    assert (r["status"] == 0).all()
    reps = 20
    t0 = time.perf_counter()
    for _ in range(reps):
        eng.verify_batch(packed)
    dt = (time.perf_counter() - t0) / reps
    eng.set_timing(True); eng.verify_batch(packed); t = eng.timings()
    print(f"{name}: {dt*1e3:.3f} ms per call, {packed.n/dt:,.0f} e-mails/s, {wl.raw_bytes/dt/1e9:.2f} GB/s of raw e-mail; "
          f"h2d {t['h2d_us']:.0f} us, device {t['total_us']:.0f} us, d2h {t['d2h_us']:.0f} us")
