"""Study: does a host wait burn a CPU?  Long GPU work (~60 ms of matmuls), then wait for it in different ways and print
the CPU seconds (user + system, all threads and per thread) consumed during the wait."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from bench import thread_cpu

dev = torch.device("cuda", 0)
a = torch.randn(8192, 8192, device=dev)
torch.cuda.synchronize()


def work():
    for _ in range(40):
        a @ a


def measure(name, wait):
    work(); torch.cuda.synchronize()
    t0c, thr0, t0 = os.times(), thread_cpu(), time.perf_counter()
    work()
    wait()
    dt = time.perf_counter() - t0
    t1c, thr1 = os.times(), thread_cpu()
    cpu = (t1c.user - t0c.user) + (t1c.system - t0c.system)
    use = sorted(((thr1[t][0] - thr0.get(t, (0.0, ""))[0], t) for t in thr1), reverse=True)[:4]
    print(f"{name:34s} wall {1e3 * dt:7.1f} ms  cpu {1e3 * cpu:7.1f} ms  threads " + ", ".join(f"{t}:{1e3 * u:.0f}" for u, t in use if u > 0.001))


measure("torch.cuda.synchronize", torch.cuda.synchronize)
ev = torch.cuda.Event(blocking=True)
measure("Event(blocking=True).synchronize", lambda: (ev.record(), ev.synchronize()))
ev2 = torch.cuda.Event()
measure("Event().synchronize", lambda: (ev2.record(), ev2.synchronize()))
def poll():
    e = torch.cuda.Event(); e.record()
    while not e.query():
        time.sleep(0.0005)
measure("Event.query + sleep(0.5 ms)", poll)
x = torch.zeros(1, device=dev)
measure("tensor.item()", lambda: x.item())
