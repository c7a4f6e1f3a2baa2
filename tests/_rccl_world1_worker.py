"""Worker of tests/test_gpu_rccl.py: backend "nccl" (= RCCL on ROCm) with a world of ONE on cuda:0 -- the library leg of the
tensor-parallel decode step (RowParallelLinear's all-reduce, vllm/plugin.py:33-50 consumer side) exercised as far as one GPU
allows: process-group init with device_id, eager all-reduce, the all-reduce captured in a HIP graph with
capture_error_mode="thread_local" (what bench.py and ParoDecoderLM.capture use) and replayed, and tp.make_allreduce's
library closure.  Prints one JSON line."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", sys.argv[1] if len(sys.argv) > 1 else "29533")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    out = {"backend": dist.get_backend()}
    x = torch.randn(1, 8192, device=dev, dtype=torch.float16)
    ref = x.clone()
    dist.all_reduce(x)                                   # eager: creates the communicator
    torch.cuda.synchronize(dev)
    out["eager_ok"] = bool(torch.equal(x, ref))
    from paroquant_amd import tp as ptp
    fn, name = ptp.make_allreduce(dev, 8192)             # world of one: the library closure
    out["make_allreduce"] = name
    res = torch.randn_like(x)
    y = fn(x.clone(), residual=res)
    out["closure_ok"] = bool(torch.allclose(y.float(), ref.float() + res.float(), atol=2e-3)) and getattr(fn, "group", "missing") is None
    # the all-reduce inside a captured decode step
    buf = ref.clone()
    s = torch.cuda.Stream(dev)
    s.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(s):
        dist.all_reduce(buf)
    torch.cuda.current_stream(dev).wait_stream(s)
    g = torch.cuda.CUDAGraph()
    try:
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            buf.mul_(2.0)
            dist.all_reduce(buf)
        buf.copy_(ref)
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize(dev)
        out["graph_ok"] = bool(torch.allclose(buf.float(), ref.float() * 8.0, rtol=1e-3, atol=1e-3))
        out["graph_error"] = None
    except Exception as e:                               # reported: bench.py falls back to eager timing in that case
        out["graph_ok"] = False
        out["graph_error"] = f"{type(e).__name__}: {e}"
    print(json.dumps(out), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
