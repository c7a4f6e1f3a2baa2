"""Worker of tests/test_gpu_parity.py::test_oneshot_allreduce_two_ranks_one_gpu: `world` processes share cuda:0 (gloo as the
control channel), build paroquant_amd.tp.OneShotAllReduce over CUDA-IPC-mapped buffers and compare it with gloo's
all-reduce on seeded data -- eager calls, then the same calls replayed from a captured HIP graph."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    from paroquant_amd import tp
    ar = tp.OneShotAllReduce(dev, 8192)
    assert ar.self_test(), "self-test failed"
    for dt in (torch.float16, torch.bfloat16):
        for n in (8, 2560, 8192):
            g = torch.Generator().manual_seed(100 * rank + n)
            x = torch.randn(n, generator=g).to(dt)
            ref = x.float().clone()
            dist.all_reduce(ref)
            got = ar(x.to(dev).clone())
            assert got.dtype == dt
            tol = (2e-2 if dt == torch.bfloat16 else 2e-3) * max(1.0, float(ref.abs().max()))
            assert float((got.float().cpu() - ref).abs().max()) <= tol, (dt, n)
            outs = [torch.empty(n, dtype=dt) for _ in range(world)]
            dist.all_gather(outs, got.cpu())
            assert all(torch.equal(o, outs[0]) for o in outs), "ranks disagree bitwise"
    # captured in a HIP graph: 6 calls per replay, replayed 5 times, every rank at its own pace
    y = torch.zeros(4096, dtype=torch.float16, device=dev)
    src = (torch.arange(4096, dtype=torch.float32) % 13 - 6).to(torch.float16).to(dev) * (rank + 1)
    s = torch.cuda.Stream(dev)
    with torch.cuda.stream(s):
        y.copy_(src); ar(y)
    torch.cuda.current_stream(dev).wait_stream(s)
    torch.cuda.synchronize(dev)
    dist.barrier()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(6):
            y.copy_(src)
            ar(y)
    for _ in range(5):
        graph.replay()
    torch.cuda.synchronize(dev)
    expect = (torch.arange(4096, dtype=torch.float32) % 13 - 6) * sum(r + 1 for r in range(world))
    assert torch.equal(y.float().cpu(), expect), "graph replay result"
    assert not ar.gave_up()
    # latency of one call (ranks sharing ONE GPU here: the kernel's own cost, not xGMI's): 200 calls in a graph
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2):
        for _ in range(200):
            ar(y)
    dist.barrier()
    g2.replay(); torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g2.replay(); e1.record(); torch.cuda.synchronize(dev)
    if rank == 0:
        print(f"ONESHOT_US_PER_CALL {e0.elapsed_time(e1) * 1e3 / 200:.2f} world {world}", flush=True)
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print("ONESHOT_OK", flush=True)


if __name__ == "__main__":
    main()
