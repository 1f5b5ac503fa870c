"""GPU micro-benchmark: what a HIP graph of K trivial kernel nodes costs to replay -- the dispatch floor under the forward's
launch structure (two forked streams per graph, two graphs in flight).
    python tools/bench_dispatch.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import tdeed_amd  # noqa: F401  (sets the hardware-queue count like the engine does)

dev = "cuda"


def build(K, streams):
    xs = [torch.zeros(64, device=dev) for _ in range(streams)]
    g = torch.cuda.CUDAGraph()
    main = torch.cuda.Stream()
    side = [torch.cuda.Stream() for _ in range(streams - 1)]
    with torch.cuda.stream(main):
        for x in xs:
            x.add_(1.0)
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=main):
            for s in side:
                s.wait_stream(main)
            for i, st in enumerate([main] + side):
                with torch.cuda.stream(st):
                    for _ in range(K // streams):
                        xs[i].add_(1.0)
            for s in side:
                main.wait_stream(s)
    return g, main


for K in (64, 128, 256, 512):
    for streams in (1, 2):
        g1, m1 = build(K, streams)
        g2, m2 = build(K, streams)
        for inflight in (1, 2):
            torch.cuda.synchronize()
            reps = 50
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
            torch.cuda.synchronize()
            a.record(sa)
            if inflight == 1:
                with torch.cuda.stream(sa):
                    for _ in range(reps):
                        g1.replay()
                b.record(sa)
            else:
                for _ in range(reps // 2):
                    with torch.cuda.stream(sa):
                        g1.replay()
                    with torch.cuda.stream(sb):
                        g2.replay()
                sa.wait_stream(sb)
                b.record(sa)
            torch.cuda.synchronize()
            t = a.elapsed_time(b) / reps * 1e3
            print(f"K={K:4d} nodes, {streams} stream(s) per graph, {inflight} graph(s) in flight: {t:8.1f} us per replay = "
                  f"{t / K:5.2f} us per node", flush=True)

print("--- separate single-chain graphs on S streams at once (K nodes each)")
for K in (128, 256):
    for S in (1, 2, 4, 6):
        gs = [build(K, 1) for _ in range(S)]
        sts = [torch.cuda.Stream() for _ in range(S)]
        torch.cuda.synchronize()
        reps = 30
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(sts[0])
        for s in sts[1:]:
            s.wait_stream(sts[0])
        for _ in range(reps):
            for (g, _m), s in zip(gs, sts):
                with torch.cuda.stream(s):
                    g.replay()
        for s in sts[1:]:
            sts[0].wait_stream(s)
        b.record(sts[0])
        torch.cuda.synchronize()
        t = a.elapsed_time(b) / reps * 1e3
        print(f"K={K:4d} nodes per graph, {S} streams: {t:8.1f} us per round of {S} graphs = {t / (K * S):5.2f} us per node", flush=True)
