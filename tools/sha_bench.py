"""SHA-256 kernel micro-benchmark: nm messages x ml bytes resident in HBM (tuning aid)."""
import sys, os, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import zkemail_rs_amd as z
nm = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 18
ml = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
dev = torch.device("cuda", 0)
eng = z.Engine(0)
blob = torch.randint(0, 256, (nm * ml + 64,), dtype=torch.uint8, device=dev)
off = torch.arange(nm + 1, dtype=torch.int64, device=dev) * ml
dig = torch.zeros(nm * 32, dtype=torch.uint8, device=dev)
st = torch.cuda.Stream()
lib = eng.lib
torch.cuda.synchronize()
for _ in range(2):
    lib.zke_sha256_batch_device(eng.h, blob.data_ptr(), off.data_ptr(), nm, dig.data_ptr(), st.cuda_stream)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(st)
reps = 5
for _ in range(reps):
    lib.zke_sha256_batch_device(eng.h, blob.data_ptr(), off.data_ptr(), nm, dig.data_ptr(), st.cuda_stream)
e1.record(st)
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
for i in (0, nm - 1):
    assert bytes(dig[32 * i:32 * i + 32].cpu().numpy()) == hashlib.sha256(bytes(blob[i * ml:(i + 1) * ml].cpu().numpy())).digest()
print(f"tile={os.environ.get('ZKE_SHA_TILE','256')} nm={nm} ml={ml}: {ms:.3f} ms  {nm*(ml+32)/ms/1e6:.1f} GB/s")
