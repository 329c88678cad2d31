"""Which multi-stream fork / join patterns survive hipStreamEndCapture on this ROCm?  Each pattern runs in a child process
(a crash in the HIP runtime must not take the driver down):  python3 tools/capture_probe.py"""
import subprocess
import sys

PATTERNS = ["pingpong", "pingpong_phased", "lanes_autograd", "cross_wait", "cross_wait_nofork", "join_to_side", "flat2", "nested", "nested_prefork", "nested_direct_join", "reuse", "nested_autograd", "flat3_autograd", "nested_autograd_prefork"]


def run(pattern):
    import torch
    dev = torch.device("cuda", 0)
    a = torch.randn(256, 256, device=dev, requires_grad=("autograd" in pattern))
    w = torch.randn(256, 256, device=dev, requires_grad=("autograd" in pattern))
    B, C, D = torch.cuda.Stream(dev), torch.cuda.Stream(dev), torch.cuda.Stream(dev)

    def fork(child, parent, *ts):
        child.wait_stream(parent)
        for t in ts:
            t.record_stream(child)

    def body():
        main = torch.cuda.current_stream(dev)
        x = a @ w
        if pattern == "flat2":
            fork(B, main, x)
            with torch.cuda.stream(B):
                y = x.relu() @ w
            main.wait_stream(B)
            return (x + y).sum()
        if pattern.startswith("flat3"):
            fork(B, main, x); fork(C, main, x)
            with torch.cuda.stream(B):
                y = x.relu() @ w
            with torch.cuda.stream(C):
                z = x.tanh() @ w
            main.wait_stream(B); main.wait_stream(C)
            return (x + y + z).sum()
        if pattern == "reuse":
            fork(B, main, x)
            with torch.cuda.stream(B):
                y = x.relu() @ w
            main.wait_stream(B)
            x2 = x + y
            fork(B, main, x2)
            with torch.cuda.stream(B):
                y2 = x2.relu() @ w
            main.wait_stream(B)
            return (x2 + y2).sum()
        if pattern.startswith("cross_wait"):   # C depends on B's work; both join main directly
            if pattern == "cross_wait":
                fork(C, main, x)
            fork(B, main, x)
            with torch.cuda.stream(B):
                y = x.relu() @ w
            fork(C, B, y)
            with torch.cuda.stream(C):
                z = y.tanh() @ w
            with torch.cuda.stream(B):
                q = y @ w
            main.wait_stream(B); main.wait_stream(C)
            return (x + q + z).sum()
        if pattern == "pingpong":             # B -> C -> B inside one fork
            fork(B, main, x); fork(C, main, x)
            with torch.cuda.stream(B):
                y = x.relu() @ w
            fork(C, B, y)
            with torch.cuda.stream(C):
                z = y.tanh() @ w
            with torch.cuda.stream(B):
                B.wait_stream(C)
                z.record_stream(B)
                r = y + z
            main.wait_stream(B); main.wait_stream(C)
            return (x + r).sum()
        if pattern == "pingpong_phased":      # B -> C, join main, then C -> B
            fork(B, main, x); fork(C, main, x)
            with torch.cuda.stream(B):
                y = x.relu() @ w
            fork(C, B, y)
            with torch.cuda.stream(C):
                z = y.tanh() @ w
            main.wait_stream(B); main.wait_stream(C)
            m = x + y + z
            fork(B, main, m); fork(C, main, m)
            with torch.cuda.stream(C):
                z2 = m.tanh() @ w
            with torch.cuda.stream(B):
                B.wait_stream(C)
                z2.record_stream(B)
                r = m + z2
            main.wait_stream(B); main.wait_stream(C)
            return r.sum()
        if pattern == "lanes_autograd":       # the aligner shape: B forks from main, C from B; C joins main only; backward by autograd
            fork(B, main, x); fork(C, main, x)
            with torch.cuda.stream(B):
                y = x.relu() @ w
                fork(C, B, y)
                with torch.cuda.stream(C):
                    z = y.tanh() @ w
                q = y @ w
            main.wait_stream(B); main.wait_stream(C)
            return (x + q + z).sum()
        if pattern == "join_to_side":         # B and C fork from main; C's result is consumed on B
            fork(B, main, x); fork(C, main, x)
            with torch.cuda.stream(C):
                z = x.tanh() @ w
            with torch.cuda.stream(B):
                y = x.relu() @ w
                B.wait_stream(C)
                z.record_stream(B)
                r = y + z
            main.wait_stream(B); main.wait_stream(C)
            return (x + r).sum()
        # nested: main -> B -> C
        if "prefork" in pattern:
            C.wait_stream(main)
        fork(B, main, x)
        with torch.cuda.stream(B):
            y = x.relu() @ w
            fork(C, B, y)
            with torch.cuda.stream(C):
                z = y.tanh() @ w
            q = y @ w
            B.wait_stream(C)
            z.record_stream(B)
            r = q + z
        main.wait_stream(B)
        if "direct_join" in pattern or "prefork" in pattern:
            main.wait_stream(C)
        r.record_stream(main)
        return (x + r).sum()

    def step():
        out = body()
        if "autograd" in pattern:
            a.grad = w.grad = None
            out.backward()
        return out

    s = torch.cuda.Stream(dev)
    s.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(s):
        for _ in range(2):
            ref = step()
    torch.cuda.current_stream(dev).wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = step()
    g.replay()
    torch.cuda.synchronize()
    print(pattern, "ok", float(out), float(ref))


if __name__ == "__main__":
    if len(sys.argv) > 1:
        run(sys.argv[1])
    else:
        for p in PATTERNS:
            r = subprocess.run([sys.executable, __file__, p], capture_output=True, text=True)
            tail = (r.stdout.strip().splitlines() or [""])[-1]
            print(f"{p:28s} rc={r.returncode} {tail}")
