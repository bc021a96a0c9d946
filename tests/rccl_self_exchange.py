"""Run by tests/test_gpu_apply.py::test_rccl_transport_on_one_gpu in a child process: the product's TorchDistTransport
(torch.distributed batch_isend_irecv, backend nccl = RCCL) with the only peer a 1-GPU box offers -- the rank itself.  RCCL
carries a send and a receive to the own rank inside one group call as a device copy, so the calls the N > 1 path makes
(process-group initialisation, P2POp lists, one batch per phase, request waits, stream ordering against the kernels
that produce and consume the buffers) are exercised; what is not is the xGMI transfer between two GPUs."""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
os.environ["MASTER_ADDR"] = "127.0.0.1"
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from l3ster_amd.distributed import TorchDistTransport
tr = TorchDistTransport()
a = torch.arange(1000, dtype=torch.float64, device="cuda"); b = torch.zeros_like(a)
c = torch.ones(77, dtype=torch.float64, device="cuda") * 3; d = torch.zeros_like(c)
reqs = tr.post([(0, a), (0, c)], [(0, b), (0, d)])
tr.wait(reqs); torch.cuda.synchronize()
ok = bool(torch.equal(a, b)) and bool(torch.equal(c, d))
# stream ordering: a buffer produced by a kernel right before the post, consumed by a kernel right after the wait
for it in range(20):
    src = torch.full((4096,), float(it), dtype=torch.float64, device="cuda")
    snd = src * 2.0 + 1.0  # producer kernel
    rcv = torch.empty_like(snd)
    tr.wait(tr.post([(0, snd)], [(0, rcv)]))
    ok = ok and bool((rcv.sum() == (2.0 * it + 1.0) * 4096).item())  # consumer kernel
print("RCCL self exchange", "ok" if ok else "FAILED")
dist.destroy_process_group()
