"""One-off probe: B=32 forward timing, fresh inputs vs a slice of the B=256 inputs, with and without the saturation poll."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from tokenpacker_amd import _capi

dev = torch.device("cuda", 0)
dtype = torch.bfloat16
model = bench.build_model(4096, 2, dtype, dev)
x256, xm256 = bench.make_device_inputs(256, dtype, "tower", dev, 1234)
x32, xm32 = bench.make_device_inputs(32, dtype, "tower", dev, 1234)
with torch.no_grad():
    for name, fn in (("fresh32", lambda: model((x32, xm32))), ("slice32", lambda: model((x256[:32], xm256[:32]))),
                     ("fresh32 again", lambda: model((x32, xm32))), ("full256", lambda: model((x256, xm256))),
                     ("fresh32 after 256", lambda: model((x32, xm32)))):
        print(name, round(bench._time_forward(fn, dev, 10, 100), 4), "ms", flush=True)
    model._sat_warned = True          # no polling
    print("fresh32, poll off", round(bench._time_forward(lambda: model((x32, xm32)), dev, 10, 100), 4), "ms")
