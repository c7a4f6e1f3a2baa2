#!/usr/bin/env python3
"""First-multi-GPU-box checklist.  NOTHING here has run on more than one GPU: the tensor-parallel path of this repo
(paroquant_amd/tp.py, csrc/allreduce.hip, the GEMV's all-reduce epilogue, bench.py --gpus N) is covered by gloo world-2/4 tests on CPU
and by N processes sharing ONE device; RCCL has only ever seen a world of one.  This script is what to run FIRST on an N-GPU MI355X node,
in order, failing loudly and naming the step:

  1. environment      GPUs visible >= N, HSA_ENABLE_IPC_MODE_LEGACY=0 (dmabuf IPC: hipIpcGetMemHandle fails without it), library loads
  2. oneshot          N ranks over RCCL: OneShotAllReduce setup (IPC handles of fine-grained buffers -- the step a node can refuse) and
                      its self-test against dist.all_reduce; a refusal is REPORTED and the run continues on RCCL (make_allreduce's rule)
  3. rccl             eager all-reduce + the all-reduce captured in a HIP graph and replayed (what ParoDecoderLM.capture / bench.py do)
  4. bench            bench.py --gpus {1,2,4,..} --workload <wl> for the tensor-parallel workloads of BASELINE config 5 / the 70B class;
                      every N > 1 line must carry config.allreduce_ab (one-shot vs RCCL timed on the same shards)

Reference for what is being sharded: vllm/plugin.py:33-50 (the rotation weight loader narrows theta / pairs / channel_scales to the
rank's K slice of a RowParallelLinear; the all-reduce behind it is vLLM's).

    python tools/tp_first_run.py --gpus 4                       # everything
    python tools/tp_first_run.py --gpus 4 --dry-run             # print the plan (JSON), run nothing -- the CPU test checks this
    python tools/tp_first_run.py --stage worker ...             # internal: one rank of steps 2 / 3 (spawned through torch.distributed.run)
Outputs: gpurun_out/tp_first_run/{steps.jsonl, bench_<workload>_n<N>.json}; exit code 0 only if every step passed or fell back as designed."""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKLOADS = ("qwen3.5-27b-class-tp", "llama3-70b-tp")


def parse_args(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--gpus", type=int, default=4, help="ranks of the widest run (1, 2, 4 .. up to this are benched)")
    ap.add_argument("--workloads", default=",".join(WORKLOADS), help="comma-separated bench.py workloads (each must end in -tp)")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--layers", type=int, default=8, help="decoder layers per bench run (0 = the whole model)")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "tp_first_run"))
    ap.add_argument("--port", type=int, default=29611)
    ap.add_argument("--dry-run", action="store_true", help="print the plan as JSON and exit (no GPU needed)")
    ap.add_argument("--skip", default="", help="comma-separated steps to skip: environment, oneshot, rccl, bench")
    ap.add_argument("--stage", default="driver", choices=["driver", "worker"])
    ap.add_argument("--probe", default="oneshot", choices=["oneshot", "rccl"], help="worker: which probe this rank runs")
    return ap.parse_args(argv)


def rank_counts(n: int):
    """1, 2, 4, .. up to n (n itself last when it is not a power of two)."""
    out, k = [], 1
    while k < n:
        out.append(k)
        k *= 2
    out.append(n)
    return out


def plan(args):
    """The commands of every step, in order: [{step, n, cmd, env, must_have}]."""
    skip = {s for s in args.skip.split(",") if s}
    wls = [w for w in args.workloads.split(",") if w]
    for w in wls:
        if not w.endswith("-tp"):
            raise SystemExit(f"tp_first_run: workload {w!r} is not tensor-parallel (bench.py shards only <model>-tp)")
    if args.gpus < 2:
        raise SystemExit("tp_first_run: --gpus must be >= 2 (this is the multi-GPU checklist; one GPU is what every other tool covers)")
    env = {"HSA_ENABLE_IPC_MODE_LEGACY": "0", "MASTER_ADDR": "127.0.0.1"}
    torchrun = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1"]
    me = os.path.abspath(__file__)
    steps = []
    if "environment" not in skip:
        steps.append({"step": "environment", "n": args.gpus, "cmd": None, "env": env, "must_have": ["gpus_visible", "ipc_mode_legacy_0", "library"]})
    port = args.port
    for probe in ("oneshot", "rccl"):
        if probe in skip:
            continue
        steps.append({"step": probe, "n": args.gpus, "env": env, "must_have": ["ok"],
                      "cmd": torchrun + ["--master-port", str(port), me, "--stage", "worker", "--probe", probe]})
        port += 1
    if "bench" not in skip:
        for w in wls:
            for n in rank_counts(args.gpus):
                cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", str(args.steps), "--warmup", str(args.warmup),
                       "--workload", w, "--no-cpu-baseline", "--no-e2e", "--no-extra", "--no-north-star"]
                if args.layers:
                    cmd += ["--layers", str(args.layers)]
                if n > 1:
                    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
                           "--master-port", str(port)] + cmd[1:]
                    port += 1
                steps.append({"step": "bench", "n": n, "workload": w, "cmd": cmd, "env": env,
                              "must_have": ["value", "roofline"] + (["config.allreduce_ab"] if n > 1 else []),
                              "out": os.path.join(args.out, f"bench_{w}_n{n}.json")})
    return steps


# ---------------------------------------------------------------------------------------------------- the per-rank probes
def worker(args):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    out = {"step": args.probe, "world": world}
    if args.probe == "oneshot":
        from paroquant_amd import tp as ptp
        try:
            ar = ptp.OneShotAllReduce(dev, 8192)          # IPC handles of fine-grained buffers: the step a node can refuse
            out["setup"] = "ok"
            out["self_test"] = bool(ar.self_test())
            out["gave_up"] = bool(ar.gave_up())
        except Exception as e:                            # refusal (IPC handle, fine-grained memory): reported, the product falls back to RCCL
            out["setup"] = f"refused: {type(e).__name__}: {e}"
            out["self_test"] = False
        fn, name = ptp.make_allreduce(dev, 8192)          # what a decode step would get on this node
        out["make_allreduce"] = name
        x = torch.full((1, 4096), float(rank + 1), device=dev, dtype=torch.float16)
        y = fn(x.clone())
        out["sum_ok"] = bool(torch.all(y == world * (world + 1) / 2))
        out["ok"] = out["sum_ok"] and (out["self_test"] or name != "oneshot")   # a refusal with a working RCCL fallback passes, loudly
        out["fell_back_to_rccl"] = name != "oneshot"
    else:
        x = torch.full((1, 8192), float(rank + 1), device=dev, dtype=torch.float16)
        want = world * (world + 1) / 2
        dist.all_reduce(x)
        torch.cuda.synchronize(dev)
        out["eager_ok"] = bool(torch.all(x == want))
        buf = torch.empty_like(x)
        s = torch.cuda.Stream(dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s):
            buf.fill_(float(rank + 1)); dist.all_reduce(buf)
        torch.cuda.current_stream(dev).wait_stream(s)
        g = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                buf.fill_(float(rank + 1))
                dist.all_reduce(buf)
            for _ in range(3):
                g.replay()
            torch.cuda.synchronize(dev)
            out["graph_ok"] = bool(torch.all(buf == want))
            out["graph_error"] = None
        except Exception as e:                            # bench.py times eagerly in that case
            out["graph_ok"], out["graph_error"] = False, f"{type(e).__name__}: {e}"
        out["ok"] = out["eager_ok"]                       # (a graph refusal is reported, not fatal)
    flag = torch.tensor([1.0 if out["ok"] else 0.0], device=dev)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)           # every rank must pass
    out["ok"] = bool(flag.item() > 0.5)
    if rank == 0:
        print("TP_FIRST_RUN " + json.dumps(out), flush=True)
    dist.destroy_process_group()
    sys.exit(0 if out["ok"] else 1)


# ---------------------------------------------------------------------------------------------------- the driver
def _get(d, dotted):
    for k in dotted.split("."):
        if not isinstance(d, dict) or k not in d or d[k] is None:
            return None
        d = d[k]
    return d


def check_environment(n):
    import torch
    sys.path.insert(0, ROOT)
    out = {"step": "environment", "gpus_visible": torch.cuda.device_count(), "want": n,
           "ipc_mode_legacy_0": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY") == "0"}
    try:
        from paroquant_amd import _native
        out["library"] = _native.lib_path()
        out["abi"] = int(_native.load().paro_abi_version())
    except Exception as e:
        out["library"] = None
        out["error"] = f"{type(e).__name__}: {e}"
    out["ok"] = out["gpus_visible"] >= n and out["library"] is not None
    if not out["ipc_mode_legacy_0"]:
        out["warning"] = "HSA_ENABLE_IPC_MODE_LEGACY is not 0 in this shell: the steps below export it themselves"
    return out


def main(argv=None):
    args = parse_args(argv)
    if args.stage == "worker":
        worker(args)
        return
    steps = plan(args)
    if args.dry_run:
        print(json.dumps({"gpus": args.gpus, "steps": steps}, indent=1))
        return
    os.makedirs(args.out, exist_ok=True)
    log = open(os.path.join(args.out, "steps.jsonl"), "w")
    failed = []
    for st in steps:
        env = dict(os.environ, **st["env"])
        if st["step"] == "environment":
            res = check_environment(st["n"])
        else:
            print("+ " + " ".join(st["cmd"]), flush=True)
            p = subprocess.run(st["cmd"], env=env, cwd=ROOT, capture_output=True, text=True)
            res = {"step": st["step"], "n": st["n"], "rc": p.returncode}
            line = None
            for ln in p.stdout.splitlines():
                if st["step"] == "bench" and ln.startswith("{"):
                    line = ln
                elif ln.startswith("TP_FIRST_RUN "):
                    line = ln[len("TP_FIRST_RUN "):]
            if line:
                try:
                    parsed = json.loads(line)
                except Exception:
                    parsed = None
                if parsed is not None and st["step"] == "bench":
                    with open(st["out"], "w") as f:
                        f.write(line + "\n")
                    res.update(workload=st["workload"], value=parsed.get("value"), ms_per_step=parsed.get("ms_per_step"),
                               allreduce_ab=_get(parsed, "config.allreduce_ab"))
                    res["ok"] = p.returncode == 0 and all(_get(parsed, k) is not None for k in st["must_have"])
                elif parsed is not None:
                    res.update(parsed)
                    res["ok"] = p.returncode == 0 and bool(parsed.get("ok"))
            if "ok" not in res:
                res["ok"] = False
                res["stderr_tail"] = p.stderr[-2000:]
        print(json.dumps(res), flush=True)
        log.write(json.dumps(res) + "\n")
        log.flush()
        if not res["ok"]:
            failed.append(f"{st['step']} (n = {st['n']})")
            if st["step"] in ("environment", "rccl"):      # nothing below can work without these
                break
    log.close()
    if failed:
        raise SystemExit("tp_first_run: FAILED at " + ", ".join(failed) + f"  (details: {os.path.join(args.out, 'steps.jsonl')})")
    print("tp_first_run: every step passed", flush=True)


if __name__ == "__main__":
    main()
