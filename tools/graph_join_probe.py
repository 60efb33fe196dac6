"""Does a captured two-stream step whose LAST captured operation is the join (main.wait_stream(side)) finish its side
branch before work enqueued on the launch stream after graph.replay() starts?  (r03 step traces show the drug chain of
replay k still running while replay k+1's protein chain has started.)  The side branch runs a long chain of kernels
into `y`; right after the replay a copy of `y` is enqueued on the SAME stream; only that stream is synchronised."""
import torch
dev = torch.device("cuda:0")
a = torch.randn(2048, 2048, device=dev)
x = torch.zeros(1 << 20, device=dev)
y = torch.zeros(1 << 20, device=dev)
snap = torch.empty_like(y)
side = torch.cuda.Stream()


def step(sentinel):
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    x.add_(1.0)                                   # short main branch
    with torch.cuda.stream(side):                 # long side branch: ~50 matmuls, then the marker write
        t = a
        for _ in range(50):
            t = (t @ a) * 1e-3
        y.fill_(1.0)
        y.mul_(2.0)
    main.wait_stream(side)
    if sentinel:
        x.add_(0.0)                               # a node on main AFTER the join


for sentinel in (False, True):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        step(sentinel)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        step(sentinel)
    bad = 0
    for rep in range(20):
        y.zero_()
        torch.cuda.synchronize()
        with torch.cuda.stream(s):
            g.replay()
            snap.copy_(y)                         # same stream, right behind the replay
            s.synchronize()                       # ONLY this stream
        bad += int((snap != 2.0).any())
        torch.cuda.synchronize()
    print(f"sentinel node after the join: {sentinel}: side-branch result incomplete behind the replay in {bad} of 20 replays")
