"""Which RCCL collectives survive hipGraph capture on this stack (one rank, one GPU)?  Each case runs in its own process
(a crash in one does not take the others down) with faulthandler on.
    python tools/probes/rccl_graph_probe.py            # all cases
    python tools/probes/rccl_graph_probe.py <case>     # one case, in this process"""
import faulthandler
import os
import socket
import subprocess
import sys

CASES = ["all_reduce", "all_reduce_async", "all_gather", "all_to_all_single", "all_to_all_single_async", "batch_isend_irecv",
         "autograd_hook_callback", "all_to_all_list", "all_to_all_list_async"]


def run(case):
    faulthandler.enable()
    import torch
    import torch.distributed as dist
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    x = torch.arange(1024.0, device=dev)
    out = torch.zeros(1024, device=dev)
    p = torch.nn.Parameter(torch.ones(1024, device=dev))

    def body():
        if case == "all_reduce":
            y = x * 2
            dist.all_reduce(y)
            out.copy_(y)
        elif case == "all_reduce_async":
            y = x * 2
            w = dist.all_reduce(y, async_op=True)
            w.wait()
            out.copy_(y)
        elif case == "all_gather":
            y = x * 2
            g = [torch.empty_like(y)]
            dist.all_gather(g, y)
            out.copy_(g[0])
        elif case == "all_to_all_list":
            y = x * 2
            r = [torch.empty_like(y)]
            dist.all_to_all(r, [y])
            out.copy_(r[0])
        elif case == "all_to_all_list_async":
            y = x * 2
            r = [torch.empty_like(y)]
            dist.all_to_all(r, [y], async_op=True).wait()
            out.copy_(r[0])
        elif case == "all_to_all_single":
            y = x * 2
            r = torch.empty_like(y)
            dist.all_to_all_single(r, y)
            out.copy_(r)
        elif case == "all_to_all_single_async":
            y = x * 2
            r = torch.empty_like(y)
            dist.all_to_all_single(r, y, output_split_sizes=[1024], input_split_sizes=[1024], async_op=True).wait()
            out.copy_(r)
        elif case == "batch_isend_irecv":
            y = x * 2
            r = torch.empty_like(y)
            for w_ in dist.batch_isend_irecv([dist.P2POp(dist.isend, y, 0), dist.P2POp(dist.irecv, r, 0)]):
                w_.wait()
            out.copy_(r)
        elif case == "autograd_hook_callback":
            p.grad = None
            (p * x).sum().backward()
            out.copy_(p.grad * 2)
        elif case == "thread_local_mode_only":
            out.copy_(x * 2)

    if case == "autograd_hook_callback":
        from torch.autograd import Variable
        state = {"armed": False}

        def fin():
            state["armed"] = False
            dist.all_reduce(p.grad)

        def hook(q):
            if not state["armed"]:
                state["armed"] = True
                Variable._execution_engine.queue_callback(fin)
        p.register_post_accumulate_grad_hook(hook)
    for _ in range(2):
        body()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        body()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    print(f"[{case}] capturing", flush=True)
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        body()
    torch.cuda.synchronize()
    print(f"[{case}] captured", flush=True)
    out.zero_()
    x.add_(1.0)
    g.replay()
    torch.cuda.synchronize()
    ok = torch.equal(out, (torch.arange(1024.0, device=dev) + 1.0) * 2)
    print(f"[{case}] replayed, result {'ok' if ok else 'WRONG'}", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(sys.argv[1])
    else:
        for c in CASES:
            try:
                r = subprocess.run([sys.executable, os.path.abspath(__file__), c], capture_output=True, text=True, timeout=45)
                rc, text = r.returncode, r.stdout + r.stderr
            except subprocess.TimeoutExpired as e:
                rc, text = "TIMEOUT (hang)", (e.stdout or b"").decode() + (e.stderr or b"").decode()
            tail = [l for l in text.splitlines() if l.strip() and "amdgpu.ids" not in l and "socket.cpp" not in l
                    and not l.startswith(("RCCL version", "HIP version", "ROCm version", "Hostname", "Librccl"))][-6:]
            print(f"=== {c}: rc {rc}")
            print("\n".join("    " + l[:300] for l in tail), flush=True)
