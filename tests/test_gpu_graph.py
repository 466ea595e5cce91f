"""The forward is HIP-graph capturable (the library only enqueues on the caller's stream: no allocation, no
synchronisation, no host-side state besides one-time kernel attributes): capture with torch.cuda.CUDAGraph,
replay on new inputs, compare bit-for-bit with the eager launch sequence."""
import json
import os
import time

import pytest
import torch

from tokenpacker_amd import TokenPacker, synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B", [1, 10])
def test_forward_graph_capture_and_replay(B):
    dtype, D = torch.bfloat16, 4096
    m = TokenPacker(hidden_size=D, scale_factor=2).to(device="cuda", dtype=dtype).eval().requires_grad_(False)
    x, xm = synth.make_inputs(3, B, dtype)
    x2, xm2 = synth.make_inputs(4, B, dtype)
    sx, sxm = x.cuda(), xm.cuda()
    with torch.no_grad():
        y_eager = m((sx, sxm)).clone()
        y2_eager = m((x2.cuda(), xm2.cuda())).clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                      # warm-up on the capture stream (workspace, kernel attributes)
            m((sx, sxm))
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            y_static = m((sx, sxm))
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(y_static, y_eager)
        sx.copy_(x2.cuda()); sxm.copy_(xm2.cuda())
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(y_static, y2_eager)

        def timed(fn, n=200):
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n * 1e6
        rec = {"B": B, "eager_launch_us": round(timed(lambda: m((sx, sxm))), 1), "graph_replay_us": round(timed(graph.replay), 1)}
    print("\n[graph]", json.dumps(rec))
    os.makedirs("gpurun_out", exist_ok=True)
    with open(f"gpurun_out/graph_latency_B{B}.json", "w") as f:
        json.dump(rec, f)
