import sys, os
sys.path.insert(0, "python-super_amd")
import torch
from super_amd import synth
from super_amd.engine import DeviceFrame, Engine
dev = torch.device("cuda", 0)
for wl in ("C2", "C1", "C4"):
    eng = Engine(dev, max_frames=8)
    frames = [DeviceFrame.from_scene(synth.make_scene(seed=s, **synth.WORKLOADS[wl]), dev) for s in range(8 if wl != "C4" else 2)]
    eng.bind_batch(frames)
    eng.run(len(frames))
    for i in range(len(frames)):
        r = eng.records(i)
        print(wl, i, "".join("T" if x["accepted"] else "F" for x in r), [f"{x['u']:.2e}" for x in r][-3:])
    eng.close()
